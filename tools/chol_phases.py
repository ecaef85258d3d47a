#!/usr/bin/env python3
"""Per-phase shader-cycle breakdown of k_chol_step, the panel step of the reduced system's dense Cholesky factorisation in LocalBundleAdjustment
(developer tool; output committed as profiles/r06_chol_phases.txt).

The PROF instantiation (csrc/orbx_lba.hip: CH_STAMP) reads s_memtime on thread 0 of the FIRST panel workgroup at its phase boundaries and adds the WALL
cycles between consecutive stamps to a device array: staging (S / L of this block column and the previous panel -> LDS), the previous panel's update on
the matrix cores, the 32 x 32 diagonal block on one wave (8 blocks of 4 pivots), the wait for the panel rows' wave, the stores.  Window = BASELINE config 5
(50 keyframes / 5000 points).
   python tools/chol_phases.py"""
import ctypes
import importlib
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
orbx = importlib.import_module("self_commit_orb-slam2_amd")
import numpy as np  # noqa: E402
import torch  # noqa: E402

NAMES = ["staging: block column + previous panel -> LDS (one round trip of loads in flight)", "previous panel's update of this column (v_mfma_f64_16x16x4_f64)",
         "diagonal block: 8 x (4 pivots, publish 4 columns through LDS, rank-4 update) on wave 0 - REST (load to registers, final store; its parts below)",
         "wait for the panel rows (wave 1, one block behind)", "stores: L21, L11, solved right-hand side",
         "  diagonal block, per launch: the pivot chains (8 x: readlanes of the 4 x 4 part, 4 x (v_rsq_f64 + Newton), its elimination)",
         "  diagonal block, per launch: own entries against the 4 x 4 part + publishing the four columns through LDS (8 x)",
         "  diagonal block, per launch: rank-4 update of the rest of the block (7 x)"]
L = orbx.load_library()
L.orbx_debug_chol_step_profile.argtypes = [ctypes.c_void_p]
L.orbx_debug_chol_step_profile.restype = None
w = orbx.lba_synth.make_window(K=50, P=5000, seed=12345)
opt = orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000)
for _ in range(3):
    res = opt.LocalBundleAdjustment(w)
buf = torch.zeros(16, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
L.orbx_debug_chol_step_profile(ctypes.c_void_p(buf.data_ptr()))
reps = 10
for _ in range(reps):
    res = opt.LocalBundleAdjustment(w)
L.orbx_debug_chol_step_profile(None)
torch.cuda.synchronize()
b = buf.cpu().numpy().astype(float)
cyc, cnt = b[:8], b[8:]
launches = cnt[0] / reps
print("k_chol_step, window of %d keyframes / %d points / %d edges, trials per window %d + %d: %.0f panel launches per window" % (w["K"], w["P"], w["E"], res["stats"][1], res["stats"][5], launches))
per = [cyc[i] / max(cnt[0], 1) for i in range(8)]      # cycles per LAUNCH (the sub-phases of the diagonal block are sums over its blocks)
tot = sum(per)
print("per launch (first panel workgroup, wall cycles of s_memtime incl. ~2 x 200 per stamp): %.0f cycles" % tot)
for i, nm in enumerate(NAMES):
    if cnt[i] > 0:
        print("  %-132s %7.0f cycles  %5.1f %%   (%.1f stamps per launch)" % (nm, per[i], 100 * per[i] / tot, cnt[i] / cnt[0]))
print("diagonal block in all: %.0f cycles per launch = %.0f per block of four pivots" % (per[2] + per[5] + per[6] + per[7], (per[2] + per[5] + per[6] + per[7]) / 8))
