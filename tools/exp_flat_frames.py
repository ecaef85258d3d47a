#!/usr/bin/env python3
"""Extraction time on LOW-TEXTURE frames (smooth gradients + weak noise): most cells find no iniThFAST corner and run the detector's
second pass at minThFAST.  python tools/exp_flat_frames.py"""
import importlib, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
orbx = importlib.import_module("self_commit_orb-slam2_amd")

def frame(seed, W=640, H=480):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    img = 120 + 40 * np.sin(x / 97.0 + seed) * np.cos(y / 71.0) + rng.normal(0, 2.5, (H, W))
    k = np.ones(5, np.float32) / 5
    for _ in range(2):
        img = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, img)
        img = np.apply_along_axis(lambda c: np.convolve(c, k, mode="same"), 0, img)
    img += rng.normal(0, 3.0, (H, W))
    return np.clip(img, 0, 255).astype(np.uint8)

B = 64
frames = [frame(s) for s in range(B)]
ext = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=B)
k, d, c = ext.extract_batch(frames)
print("keypoints per frame: mean %.0f" % np.mean(c))
ext.set_profiling(True)
acc = {}
for _ in range(10):
    ext.extract_batch(frames)
    tot, st = ext.last_timing()
    for k_, v in st.items():
        acc[k_] = acc.get(k_, 0.0) + v / 10
print("device stages of %d low-texture frames (ms):" % B, {k_: round(v, 3) for k_, v in acc.items()})
