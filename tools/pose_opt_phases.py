#!/usr/bin/env python3
"""Per-phase shader-cycle breakdown of k_pose_opt (developer tool; output committed as profiles/r06*_pose_opt_phases.txt).

The PROF instantiation of the kernel (csrc/orbx_lba.hip: PO_STAMP) reads s_memtime on thread 0 at the phase boundaries of the Levenberg loop and
adds the WALL cycles between consecutive stamps (barrier waits included: thread 0 sees what the workgroup's critical path sees) to a device array.
One frame of n correspondences per call, as Optimizer::PoseOptimization runs it (src/Optimizer.cc:363-605).
   python tools/pose_opt_phases.py [n ...]"""
import ctypes
import importlib
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
orbx = importlib.import_module("self_commit_orb-slam2_amd")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from test_pose_optimization import make_frame  # noqa: E402

NAMES = ["(stamp cost)", "round setup: estimate reset, active-edge count", "build: errors, Jacobians, 28 sums per thread", "reduce28 (transposed LDS tile, 2 barriers)",
         "H / b to LDS, lambda init (thread 0)", "6x6 LDL^T solve (thread 0)", "pose oplus = SE3 exp (thread 0)", "barrier behind the solve",
         "error pass at the trial pose", "reduce chi2 (DPP + 1 barrier)       ", "decision (thread 0) + barrier", "end of iteration: stall test + barrier",
         "classification of the round", "flags + pose out"]
L = orbx.load_library()
L.orbx_debug_pose_opt_profile.argtypes = [ctypes.c_void_p]
L.orbx_debug_pose_opt_profile.restype = None
if "--sequence" in sys.argv:
    # the PoseOptimization calls of the tracked-frame loop (tools/latency_shim.py: tracking), all of them through the PROF instantiation
    sys.path.insert(0, str(ROOT / "tools"))
    import latency_shim as ls
    ls.tracking(orbx, 1, 0)
    buf = torch.zeros(32, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    L.orbx_debug_pose_opt_profile(ctypes.c_void_p(buf.data_ptr()))
    rows = ls.tracking(orbx, 1, 0)
    L.orbx_debug_pose_opt_profile(None)
    torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(float)
    calls = 2 * 29 * 2      # two runs (warm-up + one) x 29 tracked frames x 2 calls
    cyc, cnt = b[:16] / calls, b[16:] / calls
    tot = cyc.sum()
    print("k_pose_opt in the tracked-frame loop, %d calls: %.0f wall cycles per call (stamps included)" % (calls, tot))
    for i, nm in enumerate(NAMES):
        if cnt[i] > 0:
            print("  %-58s x %5.1f  %8.0f cycles  %5.1f %%   %7.1f per pass" % (nm, cnt[i], cyc[i], 100 * cyc[i] / tot, cyc[i] / cnt[i]))
    sys.exit(0)
for n in [int(x) for x in sys.argv[1:]] or [400, 800]:
    fr = make_frame(5, n=n, stereo_frac=0.0)
    po = orbx.PoseOptimizer(max_frames=1, max_features=4096)
    for _ in range(3):
        res = po.PoseOptimization([fr])
    buf = torch.zeros(32, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    L.orbx_debug_pose_opt_profile(ctypes.c_void_p(buf.data_ptr()))
    reps = 20
    for _ in range(reps):
        res = po.PoseOptimization([fr])
    L.orbx_debug_pose_opt_profile(None)
    torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(float)
    cyc, cnt = b[:16] / reps, b[16:] / reps
    tot = cyc.sum()
    print("k_pose_opt<%d>, n = %d correspondences, iterations per round %s: %.0f wall cycles of s_memtime (100 MHz: %.1f us) per call" %
          (2 if n <= 512 else 4, n, [int(x) for x in res[0]["stats"][0::2]], tot, tot / 100.0))
    for i, nm in enumerate(NAMES):
        if cnt[i] > 0:
            print("  %-58s x %5.1f  %8.0f cycles  %5.1f %%   %7.1f per pass" % (nm, cnt[i], cyc[i], 100 * cyc[i] / tot, cyc[i] / cnt[i]))
