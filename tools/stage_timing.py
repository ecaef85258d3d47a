#!/usr/bin/env python3
"""Per-stage HIP-event timing of the extractor on one GPU (developer tool)."""
import argparse
import importlib
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
orbx = importlib.import_module("self_commit_orb-slam2_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--nfeatures", type=int, default=1000)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
ext = orbx.ORBextractor(a.nfeatures, 1.2, 8, 20, 7, max_width=a.width, max_height=a.height, max_batch=a.batch)
frames = [orbx.synth_frame(i, a.width, a.height, orbx.SYNTH_LOW_TEXTURE if i % 16 == 15 else 0) for i in range(a.batch)]
dev = ext.upload(frames)
for _ in range(3):
    ext.run_device(*dev)
ext.sync()
ext.set_profiling(True)
acc = {}
tot = 0.0
for _ in range(a.steps):
    ext.run_device(*dev)
    ext.sync()
    t, st = ext.last_timing()
    tot += t
    for k, v in st.items():
        acc[k] = acc.get(k, 0.0) + v
ext.set_profiling(False)
t0 = time.time()
for _ in range(a.steps):
    ext.run_device(*dev)
ext.sync()
wall = (time.time() - t0) / a.steps
print("batch %d  %dx%d  nfeatures %d" % (a.batch, a.width, a.height, a.nfeatures))
for k, v in acc.items():
    print("  %-12s %8.3f ms/batch" % (k, v / a.steps))
print("  events total %8.3f ms/batch ; wall %8.3f ms/batch -> %.0f frames/s" % (tot / a.steps, wall * 1e3, a.batch / wall))
k, d, c = ext.download(a.batch)
print("  keypoints/frame: mean %.1f min %d max %d" % (c.mean(), c.min(), c.max()))
