#!/usr/bin/env python3
"""Experiment: LocalBundleAdjustment on the synthetic 50-KF window, for rocprofv3 kernel stats."""
import importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
orbx = importlib.import_module("self_commit_orb-slam2_amd")
w = orbx.lba_synth.make_window(K=50, P=5000, seed=12345)
opt = orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000)
for _ in range(2):
    opt.LocalBundleAdjustment(w)
t0 = time.perf_counter()
for _ in range(5):
    r = opt.LocalBundleAdjustment(w)
print("LBA wall %.2f ms, kernel %.2f ms, stats %s" % ((time.perf_counter() - t0) / 5 * 1e3, opt.last_timing()[0], r["stats"]))
