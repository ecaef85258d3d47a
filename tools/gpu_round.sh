#!/bin/bash
# One gpurun call's worth of measurements: GPU tests, the default bench line, the RCCL (nccl) run with one rank, profiles.
#   tools/gpu_round.sh <tag> [run_profiles selectors...]
set -u
TAG=${1:-r03}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out
mkdir -p $O
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu_$TAG.log
# profiles first: the bench line below reads the counters of THIS build (profiles/latest_*.json, stamped with the hash of csrc/)
if [ $# -gt 0 ]; then
    bash tools/run_profiles.sh $TAG "$@"
    cp $O/prof_$TAG/summary/latest_*.json $ROOT/profiles/ 2>/dev/null
fi
# the drop-in surface as the reference calls it (C++ loops): one thread, 1..16 threads, the stereo Frame constructor; every single-call latency; the VALU issue table
ORBX_SHIM_BENCH_STATS=1 timeout 400 python tools/latency_shim.py > $O/latency_shim_$TAG.jsonl 2> $O/latency_shim_$TAG.err; echo "latency_shim rc $?"
timeout 300 python tools/latency_calls.py > $O/latency_calls_$TAG.txt 2>&1; echo "latency_calls rc $?"
# the reference's own metric for the drop-in: per-frame tracking time of the call chain (both libraries), where its host time goes, the kernels of one tracked frame
timeout 400 python tools/latency_track.py 4 > $O/latency_track_$TAG.jsonl 2> $O/latency_track_$TAG.err; echo "latency_track rc $?"
timeout 300 python tools/latency_track.py --trace > $O/track_trace_$TAG.txt 2>&1; echo "track trace rc $?"
bash tools/kernel_times.sh trk_$TAG python $ROOT/tools/latency_track.py 4 --no-ref > $O/track_kernels_$TAG.txt 2>&1; echo "track kernels rc $?"
# phase stamps of the two latency-bound kernels (PROF instantiations), the fixed cost of a small host call
(timeout 200 python tools/pose_opt_phases.py 400 800; timeout 200 python tools/pose_opt_phases.py --sequence) > $O/pose_opt_phases_$TAG.txt 2>&1; echo "pose_opt_phases rc $?"
timeout 200 python tools/chol_phases.py > $O/chol_phases_$TAG.txt 2>&1; echo "chol_phases rc $?"
if [ -x tools/_build/ubench_call ]; then timeout 100 tools/_build/ubench_call > $O/call_floor_$TAG.txt 2>&1; fi
# the stereo Frame constructor of the drop-in library: timeline of one constructor, quantiles of 300, per-kernel device times
(timeout 200 python tools/latency_shim.py --trace; ORBSLAM_BENCH_QUANTILES=1 timeout 200 python tools/prof_stereo_ctor.py 300; ORBX_SHIM_EARLY=0 ORBSLAM_BENCH_QUANTILES=1 timeout 200 python tools/prof_stereo_ctor.py 300; bash tools/kernel_times.sh sc_$TAG python $ROOT/tools/prof_stereo_ctor.py 300) > $O/stereo_ctor_$TAG.txt 2>&1; echo "stereo ctor rc $?"
if [ -x tools/_build/ubench_valu ]; then (echo "# rocm-smi before:"; rocm-smi --showclocks 2>/dev/null | grep -i sclk | head -2; tools/_build/ubench_valu; echo "# rocm-smi after:"; rocm-smi --showclocks 2>/dev/null | grep -i sclk | head -2) > $O/valu_issue_$TAG.txt 2>&1; fi
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc $?"; cut -c1-400 $O/bench_$TAG.json
# RCCL for real: one rank under the launcher the driver uses, backend nccl (init with device_id, all-reduce, all-gather, barriers)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 \
    --no-workloads --no-cpu-baseline > $O/bench_${TAG}_nccl_world1.json 2> $O/bench_${TAG}_nccl_world1.err; echo "nccl rc $?"
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_${TAG}_nccl_world1.json").read().strip().splitlines()[-1])
    print("nccl world1:", d["value"], d["ranks"])
except Exception as e:
    print("nccl world1 line unreadable:", e)
PY
