#!/bin/bash
# Runs on the GPU box: rocprofv3 --hip-trace --stats of a command, prints per-HIP-API calls / avg us / total (host side of a latency-bound path).
# Usage: tools/hip_api_times.sh <tag> <command...>
set -u
TAG=$1; shift
OUT=/tmp/ha_$TAG
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --hip-runtime-trace --stats -d $OUT -o k -- "$@" > $OUT.log 2>&1
tail -2 $OUT.log | cut -c1-300
python3 - <<PY
import sqlite3, glob
db = glob.glob("$OUT/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
cand = [t for t in tabs if "top" in t.lower() or "hip" in t.lower()]
done = False
for t in ("top_hip_api", "top"):
    if t in tabs:
        cols = [c[1] for c in con.execute("pragma table_info(%s)" % t)]
        print(t, cols)
        for row in con.execute("select * from %s" % t):
            print(row)
        done = True
        break
if not done:
    print("tables:", cand)
    # fall back: aggregate the regions table
    for t in tabs:
        if t.startswith("regions") and not t.startswith("regions_and"):
            pass
    try:
        q = "select name, count(*), avg(end-start)/1e3, sum(end-start)/1e3 from regions_and_samples group by name order by 4 desc limit 25"
        for n, c, a, s in con.execute(q):
            print("%-40s calls %6d  avg %8.2f us  total %10.1f us" % (n, c, a, s))
    except Exception as e:
        print("no aggregate:", e)
PY
