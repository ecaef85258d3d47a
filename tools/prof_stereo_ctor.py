#!/usr/bin/env python3
"""The drop-in stereo Frame constructor alone (for rocprofv3 --kernel-trace --stats): Frame::Frame(imLeft, imRight, ...) of the reference
in oracle/_ref/liborbslam_hip.so, 1241x376 / 2000 features.   python tools/prof_stereo_ctor.py [iters=300]"""
import importlib, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
orbx = importlib.import_module("self_commit_orb-slam2_amd")
import latency_shim
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for r in latency_shim.stereo_frame(orbx, iters, 0):
    print(r)
