#!/bin/bash
# Runs on the GPU box (via gpurun): FETCH_SIZE / WRITE_SIZE (+ the raw TCC request counters when this rocprofv3 exposes them) of
# tools/ubench_hbm_calib, one counter set per pass.   -> gpurun_out/calib/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/calib
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$ROOT/tools/ubench_hbm_calib > $OUT/known.txt
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RD_UNCACHED_32B_sum TCC_BUBBLE_sum"; do
  T=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/$T -o calib -- $ROOT/tools/ubench_hbm_calib > $OUT/$T.log 2>&1 || echo "pass $T failed"
done
find $OUT -name '*counter_collection.csv' | head
