#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + the PMC passes of the default bench, and the kernel trace of the
# one-stream run (kernels alone).   Usage: tools/run_profiles.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace1 -o bench -- $BENCH --streams 1 > $OUT/trace1.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES -d $OUT/pmc_sq2 -o bench -- $BENCH > $OUT/pmc_sq2.log 2>&1
find $OUT -name '*.db' | head -20
tail -1 $OUT/trace.log | cut -c1-200
