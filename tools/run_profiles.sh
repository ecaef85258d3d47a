#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel traces and PMC passes of bench.py, summarised ON the box (the result
# databases are large and stay there); only small CSV / JSON summaries come back under gpurun_out/prof_<tag>/summary/, to be
# copied into profiles/.
#   tools/run_profiles.sh <tag> [default] [alone] [pmc] [sq] [stereo] [stereo_pmc] [lba]      (no selector = all)
# Every summary is stamped with the hash of csrc/ it was measured on (bench.py refuses counters of other sources).
set -u
TAG=${1:-r03}; shift || true
SEL=" ${*:-default alone pmc sq stereo stereo_pmc lba} "
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
RAW=/tmp/prof_raw_$TAG
mkdir -p $OUT/summary $RAW
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-workloads"
run() {   # run <name> <rocprof args...> -- <command...>
    local name=$1; shift
    timeout 900 rocprofv3 "$@" > $OUT/$name.log 2>&1 || echo "[$name] rc $?" >> $OUT/errors.txt
}
want() { [[ "$SEL" == *" $1 "* ]]; }
want default    && run trace        --kernel-trace --stats -d $RAW/trace -o bench -- $B --steps 10 --warmup 2
want alone      && run alone        --kernel-trace --stats -d $RAW/alone -o bench -- $B --steps 6 --warmup 2 --alone --batches-per-step 2
want pmc        && run pmc_fetch    --pmc FETCH_SIZE -d $RAW/pmc_fetch -o bench -- $B --steps 3 --warmup 1 --batches-per-step 2
want pmc        && run pmc_write    --pmc WRITE_SIZE -d $RAW/pmc_write -o bench -- $B --steps 3 --warmup 1 --batches-per-step 2
want sq         && run pmc_sq       --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $RAW/pmc_sq -o bench -- $B --steps 3 --warmup 1 --batches-per-step 2
want sq         && run pmc_sq2      --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES -d $RAW/pmc_sq2 -o bench -- $B --steps 3 --warmup 1 --batches-per-step 2
want stereo     && run st_trace     --kernel-trace --stats -d $RAW/st_trace -o bench -- $B --workload stereo --steps 12 --warmup 3
want stereo     && run st_alone     --kernel-trace --stats -d $RAW/st_alone -o bench -- $B --workload stereo --steps 6 --warmup 2 --alone
want stereo_pmc && run st_pmc_fetch --pmc FETCH_SIZE -d $RAW/st_pmc_fetch -o bench -- $B --workload stereo --steps 4 --warmup 1
want stereo_pmc && run st_pmc_write --pmc WRITE_SIZE -d $RAW/st_pmc_write -o bench -- $B --workload stereo --steps 4 --warmup 1
want lba        && run lba_trace    --kernel-trace --stats -d $RAW/lba_trace -o bench -- $B --workload lba --steps 10 --warmup 2 --streams 1
python $ROOT/tools/summarize_all.py $RAW $OUT/summary $TAG > $OUT/summarize.log 2>&1 || echo "[summarize] rc $?" >> $OUT/errors.txt
tail -30 $OUT/summarize.log
cat $OUT/errors.txt 2>/dev/null
ls $OUT/summary
