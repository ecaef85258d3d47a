#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + two PMC passes of the default bench.
# Usage: tools/run_profiles.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
find $OUT -name '*.csv' | head -20
tail -2 $OUT/trace.log
