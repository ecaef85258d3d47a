#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 --kernel-trace --stats of an arbitrary bench.py command line.
# Usage: tools/run_trace.sh <tag> <bench.py arguments...>   -> gpurun_out/trace_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $ROOT/bench.py "$@" > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-300
