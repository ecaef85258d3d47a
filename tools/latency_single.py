#!/usr/bin/env python3
"""Single-frame latency of the drop-in call: host image in, host keypoints + descriptors out
(ORBextractor::operator(), reference src/ORBextractor.cc:1604), batch of one."""
import importlib, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
orbx = importlib.import_module("self_commit_orb-slam2_amd")
for (W, H, nf) in ((640, 480, 1000), (1241, 376, 2000)):
    fr = orbx.synth_sequence(7, 8, W, H)
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1)
    for f in fr[:3]:
        ext(f)
    ts = []
    for i in range(40):
        t0 = time.perf_counter()
        kps, desc = ext(fr[i % 8])
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print("%dx%d / %d features: operator() median %.3f ms, min %.3f ms, %d keypoints" % (W, H, nf, np.median(ts), ts.min(), len(kps)))
    ext.set_profiling(True)
    for i in range(8):
        ext(fr[i % 8])
    tot, st = ext.last_timing()
    print("   device stages (ms):", {k: round(v, 3) for k, v in st.items()}, "sum %.3f" % tot)
