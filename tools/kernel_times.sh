#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace --stats of an arbitrary command, prints per-kernel calls / avg us.
# Usage: tools/kernel_times.sh <tag> <command...>
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/kt_$TAG
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o k -- "$@" > $OUT.log 2>&1
tail -2 $OUT.log | cut -c1-300
python3 - <<PY
import sqlite3, glob
db = glob.glob("$OUT/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
for n, c, t, a, p in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    n = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if not n.startswith("__amd"): print("%-28s calls %5d  avg %9.1f us  %5.1f%%" % (n, c, a, p))
PY
