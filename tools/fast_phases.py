#!/usr/bin/env python3
"""Per-phase shader-cycle breakdown of k_fast_cells (developer tool; output committed as profiles/r05*_fast_phases.txt).

The PROF instantiation of the kernel (csrc/orbx_kernels.hip: FC_STAMP) reads s_memtime at its phase boundaries and adds, per cell (= per wave),
the cycles between consecutive stamps to a device array.  The cycles are WALL cycles of a wave (they include the time the wave waits for the
SIMD's other waves and for memory), so the shares say where a wave's lifetime goes; the stamps themselves (an SMEM round trip each) are
measured and listed.  Same frames as bench.py's headline batch (synth_sequence, every 16th frame low texture).
   python tools/fast_phases.py [--batch 256] [--width 640 --height 480]"""
import argparse
import ctypes
import importlib
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
orbx = importlib.import_module("self_commit_orb-slam2_amd")
import numpy as np  # noqa: E402
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--nfeatures", type=int, default=1000)
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()

L = orbx.load_library()
L.orbx_debug_fast_cells_profile.argtypes = [ctypes.c_void_p]
L.orbx_debug_fast_cells_profile.restype = None
ext = orbx.ORBextractor(a.nfeatures, 1.2, 8, 20, 7, max_width=a.width, max_height=a.height, max_batch=a.batch)
frames = orbx.synth_sequence(1, a.batch, a.width, a.height)
dev = ext.upload(frames)
for _ in range(2):
    ext.run_device(*dev)
ext.sync()
cells_per_frame = 0      # the launch grid: sum over the levels of nCols * nRows (src/ORBextractor.cc:1060-1066)
for l in range(8):
    s = 1.2 ** l
    lw, lh = int(round(a.width / s)), int(round(a.height / s))
    cells_per_frame += ((lw - 32) // 30) * ((lh - 32) // 30)
nrec = a.batch * cells_per_frame
buf = torch.zeros(32 + 16 * nrec, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
L.orbx_debug_fast_cells_profile(ctypes.c_void_p(buf.data_ptr()))
ext.run_device(*dev)
ext.sync()
L.orbx_debug_fast_cells_profile(None)
torch.cuda.synchronize()
rec = buf.cpu().numpy()[32:].reshape(nrec, 16).astype(float)
rec = rec[rec[:, :11].sum(1) > 0]      # (cells the geometry skips write nothing)
p = np.zeros(32)
p[:11] = rec[:, :11].sum(0); p[22] = rec[:, 11].sum(); p[16] = len(rec); p[17] = nrec - len(rec)
p[18] = rec[:, 12].sum(); p[19] = rec[:, 13].sum(); p[20] = rec[:, 14].sum(); p[21] = rec[:, 15].sum()
names = ["prologue (block -> level / cell, bounds)", "staging (window -> LDS) + zeroing the score tile", "A: window rows + compass pre-test (iniThFAST)",
         "A: wave scan + list append", "B: exact score of the survivors", "C: 3x3 maxima + ini / min flags",
         "retry A: pre-test at minThFAST", "retry A: scan + append", "retry B", "retry C", "emission (ordered ballot compaction)"]
cells, skipped, retried = p[16], p[17], p[18]
tot = p[:11].sum()
print("k_fast_cells per-phase shader cycles (s_memtime), batch %d x %dx%d, %d features; %d cells + %d skipped per launch" %
      (a.batch, a.width, a.height, a.nfeatures, cells, skipped))
print("%-52s %12s %8s" % ("phase", "cycles/cell", "share"))
for i, n in enumerate(names):
    print("%-52s %12.1f %7.1f%%" % (n, p[i] / cells, 100.0 * p[i] / tot))
print("%-52s %12.1f" % ("sum (wave lifetime between first and last stamp)", tot / cells))
print("%-52s %12.1f   (the s_memtime round trips themselves, excluded from the rows above)" % ("stamp cost", p[22] / cells))
life = rec[:, :11].sum(1)
print("wave lifetime quantiles (cycles): p10 %.0f  p50 %.0f  p90 %.0f  p99 %.0f" % tuple(np.percentile(life, [10, 50, 90, 99])))
nr = rec[rec[:, 12] == 0]
print("cells NOT retried: %.0f cycles mean; retried: %.0f cycles mean" % (nr[:, :11].sum(1).mean(), rec[rec[:, 12] > 0][:, :11].sum(1).mean() if (rec[:, 12] > 0).any() else 0))
h = np.bincount(np.minimum(np.ceil(rec[:, 13] / 64.0).astype(int), 8), minlength=9)
print("phase-B passes at iniThFAST (64 survivors each): " + "  ".join("%d:%.1f%%" % (i, 100.0 * h[i] / len(rec)) for i in range(9)))
print("cells retried at minThFAST: %.1f %%  (their retry phases: %.1f cycles per RETRIED cell)" % (100.0 * retried / cells, p[6:10].sum() / max(retried, 1)))
print("pre-test survivors per cell: %.1f at iniThFAST; %.1f per retried cell at minThFAST; candidates emitted per cell %.2f" %
      (p[19] / cells, p[20] / max(retried, 1), p[21] / cells))
