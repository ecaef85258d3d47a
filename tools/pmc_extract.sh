#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 --pmc passes over the extractor alone (tools/stage_timing.py, 256 x 640x480), one pass per counter group.
#   tools/pmc_extract.sh <tag> "<counters of pass 1>" "<counters of pass 2>" ...     -> gpurun_out/pmcx_<tag>.txt
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
RAW=/tmp/pmcx_$TAG
mkdir -p $RAW $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
DIRS=""
for grp in "$@"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp -d $RAW/p$i -o bench -- python $ROOT/tools/stage_timing.py --steps 2 > $RAW/p$i.log 2>&1 || { echo "[pass $i: $grp] failed"; tail -3 $RAW/p$i.log; }
    DIRS="$DIRS $RAW/p$i"
done
python $ROOT/tools/summarize_pmc.py $DIRS | tee $ROOT/gpurun_out/pmcx_$TAG.txt
