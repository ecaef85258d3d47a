#!/usr/bin/env python3
"""The multi-threaded drop-in call alone (for rocprofv3 --kernel-trace --stats): N threads, one shim ORBextractor each, 640x480 / 1000 features.
   python tools/prof_dropin.py [threads=16] [iters=300] [keep_host_pyramid=0]"""
import ctypes, importlib, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
orbx = importlib.import_module("self_commit_orb-slam2_amd")
import latency_shim
L = latency_shim._shim_lib(orbx)
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
keep = int(sys.argv[3]) if len(sys.argv) > 3 else 0
frames = orbx.synth_sequence(7, 8, 640, 480)
arr = (ctypes.c_void_p * 8)(*[f.ctypes.data for f in frames])
print("threads %d: %.1f frames/s" % (nt, L.shim_bench_threads(nt, 1000, arr, 8, 640, 480, 640, iters, keep)))
