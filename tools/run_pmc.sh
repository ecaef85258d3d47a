#!/bin/bash
# Runs on the GPU box (via gpurun): one rocprofv3 --pmc pass of the default bench with the given counters.
# Usage: tools/run_pmc.sh <tag> <counter> [<counter> ...]   -> gpurun_out/pmc_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc "$@" -d $OUT -o bench -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-200
