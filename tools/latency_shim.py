#!/usr/bin/env python3
"""Latency of the REAL drop-in call: ORB_SLAM2::ORBextractor::operator() of shim/ORBextractor.cc (C++, the reference's call shape
(*extractor)(im, cv::Mat(), keys, desc), src/Frame.cc:503), one frame per call on one thread, host image in, std::vector<cv::KeyPoint>
+ cv::Mat descriptors out.  Timed inside C++ (tests/shim_wrap.cc: shim_bench).  Prints one JSON line per configuration."""
import ctypes, importlib, json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
orbx = importlib.import_module("self_commit_orb-slam2_amd")
orbx.load_library()
import test_shim_dropin as tsd
tsd.build_shim()
L = ctypes.CDLL(str(tsd.SO))
L.shim_create.restype = ctypes.c_void_p
L.shim_create.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int]
for (W, H, nf) in ((640, 480, 1000), (1241, 376, 2000)):
    h = ctypes.c_void_p(L.shim_create(nf, 1.2, 8, 20, 7))
    frames = orbx.synth_sequence(7, 8, W, H)
    arr = (ctypes.c_void_p * 8)(*[f.ctypes.data for f in frames])
    for keep in (0, 1):
        mean, med, nk = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        L.shim_bench(h, arr, 8, W, H, W, 300, keep, ctypes.byref(mean), ctypes.byref(med), ctypes.byref(nk))
        print(json.dumps({"call": "ORBextractor::operator() via shim (C++)", "size": "%dx%d" % (W, H), "nfeatures": nf, "host_pyramid": bool(keep),
                          "mean_us": round(mean.value, 1), "median_us": round(med.value, 1), "frames_per_s_one_thread": round(1e6 / mean.value, 1),
                          "keypoints": nk.value}))
    L.shim_destroy(h)

L.shim_bench_threads.restype = ctypes.c_double
L.shim_bench_threads.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
frames = orbx.synth_sequence(7, 8, 640, 480)
arr = (ctypes.c_void_p * 8)(*[f.ctypes.data for f in frames])
for keep in (0, 1):      # 0: the full drop-in (shim/Frame_hip.cc linked, the stereo matcher reads the device pyramid); 1: extractor-only swap (mvImagePyramid refilled per call)
    for nt in (1, 2, 4, 8, 16):
        fps = L.shim_bench_threads(nt, 1000, arr, 8, 640, 480, 640, 400, keep)
        print(json.dumps({"call": "ORBextractor::operator() via shim (C++), one extractor per thread", "size": "640x480", "nfeatures": 1000, "threads": nt,
                          "host_pyramid": bool(keep), "frames_per_s": round(fps, 1)}))
