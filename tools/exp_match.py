#!/usr/bin/env python3
"""Experiment: matcher alone on resident features (no extraction running concurrently)."""
import importlib, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
orbx = importlib.import_module("self_commit_orb-slam2_amd")
W, H, B, nf = 640, 480, 256, 1000
frames = orbx.synth_sequence(1, B, W, H)
ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
dev = ext.upload(frames)
ext.run_device(*dev)
ext.sync()
mt = orbx.ORBmatcher(0.7, True, max_features=ext.capacity, max_pairs=B)
pa = np.arange(B, dtype=np.int32); pb = (pa + 1) % B
fs = orbx.ORBmatcher.features_of(ext, B)
for _ in range(3):
    mt.search_by_bow_device(fs, fs, pa, pb, mode=0)
mt.sync()
t0 = time.perf_counter()
for _ in range(20):
    mt.search_by_bow_device(fs, fs, pa, pb, mode=0)
mt.sync()
print("match alone: %.3f ms/call" % ((time.perf_counter() - t0) / 20 * 1e3))
