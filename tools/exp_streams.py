#!/usr/bin/env python3
"""Experiment: does splitting the 256-frame batch over K extractor handles (= K HIP streams)
overlap the latency-bound stages (quadtree, small pyramid levels) with the VALU-bound ones?"""
import importlib, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
orbx = importlib.import_module("self_commit_orb-slam2_amd")
W, H, B, nf = 640, 480, 256, 1000
frames = orbx.synth_sequence(1, B, W, H)
for K in (1, 2, 4, 8):
    per = B // K
    exts = [orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=per) for _ in range(K)]
    devs = [e.upload(frames[i * per:(i + 1) * per]) for i, e in enumerate(exts)]
    for _ in range(3):
        for e, d in zip(exts, devs):
            e.run_device(*d)
    for e in exts:
        e.sync()
    t0 = time.perf_counter()
    steps = 20
    for _ in range(steps):
        for e, d in zip(exts, devs):
            e.run_device(*d)
    for e in exts:
        e.sync()
    dt = time.perf_counter() - t0
    print("K=%d  %.3f ms/step  %.0f frames/s" % (K, dt / steps * 1e3, B * steps / dt), flush=True)
    for e in exts:
        e.close()
