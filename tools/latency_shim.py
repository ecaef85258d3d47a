#!/usr/bin/env python3
"""Latency / throughput of the REAL drop-in surface, timed inside C++:

  * ORB_SLAM2::ORBextractor::operator() of shim/ORBextractor.cc in the reference's call shape (*extractor)(im, cv::Mat(), keys, desc)
    (src/Frame.cc:503): host image in, std::vector<cv::KeyPoint> + cv::Mat descriptors out (tests/shim_wrap.cc: shim_bench) - one thread,
    with and without the host pyramid (mbKeepHostPyramid, the default of a build that swaps only the extractor);
  * the same call from 1..16 threads, one extractor per thread (shim_bench_threads): concurrent calls are combined into one launch set
    per geometry inside liborbx (csrc/orbx_extractor.hip, "the combiner"); ORBX_COMBINE=0 = every handle its own graph (round 3);
  * the reference's stereo Frame constructor (src/Frame.cc:100-199: two extractor threads + ComputeStereoMatches) in
    oracle/_ref/liborbslam_hip.so (drop-in) and, a few iterations, in oracle/_ref/liborbslam.so (the reference's own CPU path on cvshim).

`measure(orbx, quick)` returns the numbers as a dict (bench.py puts a digest of it into the driver-visible line); run as a script it
prints one JSON line per configuration (profiles/*_latency_shim.jsonl)."""
import ctypes
import importlib
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))


def _shim_lib(orbx):
    orbx.load_library()
    import test_shim_dropin as tsd
    tsd.build_shim()
    L = ctypes.CDLL(str(tsd.SO))
    L.shim_create.restype = ctypes.c_void_p
    L.shim_create.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.shim_destroy.argtypes = [ctypes.c_void_p]
    L.shim_bench.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                             ctypes.c_void_p, ctypes.c_void_p]
    L.shim_bench_threads.restype = ctypes.c_double
    L.shim_bench_threads.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return L


def one_thread(orbx, L, W, H, nf, iters):
    rows = []
    h = ctypes.c_void_p(L.shim_create(nf, 1.2, 8, 20, 7))
    frames = orbx.synth_sequence(7, 8, W, H)
    arr = (ctypes.c_void_p * 8)(*[f.ctypes.data for f in frames])
    for keep in (0, 1, 2, 3):      # 0: no host pyramid; 1: mbKeepHostPyramid and every level READ after every call (src/Frame.cc:1044, 1248); 2: kept, never read (lazy: nothing moves); 3: as 1 with mbViewHostPyramid
        mean, med, nk = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        L.shim_bench(h, arr, 8, W, H, W, iters, keep, ctypes.byref(mean), ctypes.byref(med), ctypes.byref(nk))
        rows.append({"call": "ORBextractor::operator() via shim (C++)", "size": "%dx%d" % (W, H), "nfeatures": nf, "host_pyramid": keep in (1, 3), "host_pyramid_unread": keep == 2, "host_pyramid_views": keep == 3,
                     "combiner": os.environ.get("ORBX_COMBINE", "1") != "0", "mean_us": round(mean.value, 1), "median_us": round(med.value, 1),
                     "frames_per_s_one_thread": round(1e6 / mean.value, 1), "keypoints": nk.value})
    L.shim_destroy(h)
    return rows


def threads(orbx, L, nts, iters, keeps=(0, 1)):
    rows = []
    frames = orbx.synth_sequence(7, 8, 640, 480)
    arr = (ctypes.c_void_p * 8)(*[f.ctypes.data for f in frames])
    for keep in keeps:      # 0: the full drop-in (shim/Frame_hip.cc linked, the stereo matcher reads the device pyramid); 1: extractor-only swap (mvImagePyramid per call)
        for nt in nts:
            fps = L.shim_bench_threads(nt, 1000, arr, 8, 640, 480, 640, iters, keep)
            rows.append({"call": "ORBextractor::operator() via shim (C++), one extractor per thread", "size": "640x480", "nfeatures": 1000, "threads": nt,
                         "host_pyramid": keep in (1, 3), "host_pyramid_unread": keep == 2, "host_pyramid_views": keep == 3, "combiner": os.environ.get("ORBX_COMBINE", "1") != "0",
                         "frames_per_s": round(fps, 1)})
    return rows


def stereo_frame(orbx, iters_hip, iters_ref):
    """Frame::Frame(imLeft, imRight, ...) of the reference, 1241x376 / 2000 features, in the drop-in library and in the all-reference one."""
    import oracle_lib
    rows = []
    W, H, nf = 1241, 376, 2000
    fr = orbx.synth_sequence(9, 8, W, H, views_per_scene=2, step=(6, 0))      # pairs: view 0 = left, view 1 = right (6 px disparity)
    aL = (ctypes.c_void_p * 4)(*[fr[2 * i].ctypes.data for i in range(4)])
    aR = (ctypes.c_void_p * 4)(*[fr[2 * i + 1].ctypes.data for i in range(4)])
    for name, lib, iters in (("liborbslam_hip.so (drop-in)", oracle_lib.slam_hip_lib(), iters_hip), ("liborbslam.so (reference CPU path on cvshim)", oracle_lib.slam_lib(), iters_ref)):
        if lib is None or iters <= 0:
            continue
        lib.orbslam_stereo_frame_bench.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int] + \
            [ctypes.c_float] * 6 + [ctypes.c_int] + [ctypes.c_void_p] * 4
        mean, med, nl, nm = ctypes.c_double(), ctypes.c_double(), ctypes.c_int(), ctypes.c_int()
        prof = hasattr(lib, "orbx_shim_profile")
        if prof:
            lib.orbx_shim_profile.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            for i in range(5):
                lib.orbx_shim_profile(i, 1, None, None)
        lib.orbslam_stereo_frame_bench(aL, aR, 4, W, H, W, nf, 1.2, 8, 20, 7, 718.856, 718.856, 607.19, 185.2, 386.1448, 35.0, iters, ctypes.byref(mean), ctypes.byref(med),
                                       ctypes.byref(nl), ctypes.byref(nm))
        rows.append({"call": "Frame::Frame(imLeft, imRight, ...) = 2 x ExtractORB on two threads + ComputeStereoMatches", "library": name, "size": "%dx%d" % (W, H),
                     "nfeatures": nf, "combiner": os.environ.get("ORBX_COMBINE", "1") != "0", "mean_us": round(mean.value, 1), "median_us": round(med.value, 1),
                     "keypoints_left": nl.value, "stereo_matches": nm.value,
                     "allocator_note": "this constructor loop runs WITHOUT the CallerArena of the parity tests: the quadtree's tie between equally large nodes is decided by "
                                       "their addresses in the reference (src/ORBextractor.cc:948), i.e. by the allocator - on plain malloc the all-reference library keeps ~0.1 % other "
                                       "keypoints (and a few other stereo matches) than under the bump allocator the drop-in's rule restates (tests/test_tie_rule.py); inside the arena both "
                                       "libraries agree to the bit (tests/test_dropin_slam.py)"})
        if prof:      # mean microseconds per constructor inside every replaced member function (ExtractORB: per call, two concurrent calls per frame)
            br = {}
            for i, nm_ in enumerate(("ExtractORB_per_call", "UndistortKeyPoints", "ComputeStereoMatches", "ComputeImageBounds", "AssignFeaturesToGrid")):
                us, calls = ctypes.c_double(), ctypes.c_ulong()
                lib.orbx_shim_profile(i, 1, ctypes.byref(us), ctypes.byref(calls))
                if calls.value:
                    br[nm_] = round(us.value / calls.value, 1)
            rows[-1]["breakdown_us"] = br
    return rows


TRACK_CALLS = ("Frame::Frame(mono) = ExtractORB + UndistortKeyPoints + AssignFeaturesToGrid", "Frame::ComputeBoW", "ORBmatcher::SearchByBoW(KF, F)",
               "Optimizer::PoseOptimization #1", "Tracking::SearchLocalPoints (isInFrustum + SearchByProjection)", "Optimizer::PoseOptimization #2",
               "KeyFrame insertion (local mapper)", "Optimizer::LocalBundleAdjustment (local mapper)")


def tracking(orbx, runs_hip=3, runs_ref=1, nframes=30, kf_every=5):
    """The reference's OWN metric for the drop-in: per-frame tracking time (Examples/Monocular/mono_tum.cc:81-95, 114-122: steady_clock around the
    frame's work, median and mean over the sequence, no pacing) of the call chain src/Tracking.cc makes for a tracked frame - the monocular Frame
    constructor (src/Frame.cc:394), ComputeBoW, SearchByBoW (src/Tracking.cc:1195), PoseOptimization, SearchLocalPoints (:1765-1829), PoseOptimization -
    on real Frame / KeyFrame / MapPoint / Map objects (oracle/refslam_wrap.cc: orbslam_sequence, the loop tests/test_sequence_dropin.py proves
    identical in both libraries), every kf_every-th frame a KeyFrame + LocalBundleAdjustment (timed apart: the local mapper's thread).  640x480,
    1000 features, 30 translating views of one plane; the first run of a library warms it up (graphs, engines, statics) and is not counted."""
    import numpy as np
    import oracle_lib
    import tempfile
    W, H, NF = 640, 480, 1000
    MAXKF, MAXPT = 16, 16384
    frames = orbx.synth_sequence(4242, nframes, W, H, views_per_scene=nframes, step=(3, 1), low_texture_every=0)
    rows = []
    with tempfile.TemporaryDirectory() as td:
        voc = orbx.voc_synth.make_vocabulary(10, 4, 5)
        path = Path(td) / "voc.txt"
        orbx.voc_synth.write_text(voc, path)
        for name, lib, runs in (("liborbslam_hip.so (drop-in)", oracle_lib.slam_hip_lib(), runs_hip), ("liborbslam.so (reference CPU path on cvshim)", oracle_lib.slam_lib(), runs_ref)):
            if lib is None or runs <= 0 or not hasattr(lib, "orbslam_sequence_timing"):
                continue
            V = oracle_lib.RefVocabulary(path, lib)
            arr = (ctypes.c_void_p * nframes)(*[f.ctypes.data for f in frames])
            rec = np.zeros((nframes, 64), np.float64)
            kf = np.zeros((64, MAXKF, 17), np.float32)
            pt = np.zeros((64, MAXPT, 4), np.float32)
            ne = ctypes.c_int(0)
            P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
            lib.orbslam_sequence.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p] + [ctypes.c_float] * 5 + [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int] * 2 + [ctypes.c_void_p]
            lib.orbslam_sequence_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
            times = []
            for run in range(runs + (1 if "hip" in name else 0)):
                t = np.zeros((nframes, 8), np.float64)
                lib.orbslam_sequence_timing(P(t), nframes)
                rc = lib.orbslam_sequence(arr, nframes, W, H, W, NF, V.h, 500.0, 500.0, 320.0, 240.0, 2.0, kf_every, None, None, None, P(rec), P(kf), P(pt), MAXKF, MAXPT, ctypes.byref(ne))
                lib.orbslam_sequence_timing(None, 0)
                if rc != 0:
                    raise RuntimeError("orbslam_sequence rc %d" % rc)
                if "hip" in name and run == 0:
                    continue
                times.append(t[1:])                      # frame 0 builds the initial map: not a tracked frame
            t = np.concatenate(times)
            track = t[:, :6].sum(1)
            kfr = t[:, 7] > 0
            rows.append({"kind": "track", "call": "tracked frame: Frame ctor -> ComputeBoW -> SearchByBoW -> PoseOptimization -> SearchLocalPoints -> PoseOptimization", "library": name,
                         "size": "%dx%d" % (W, H), "nfeatures": NF, "frames": int(len(track)), "track_ms_median": round(float(np.median(track)), 4),
                         "track_ms_mean": round(float(track.mean()), 4), "track_ms_p90": round(float(np.quantile(track, 0.9)), 4),
                         "matches_bow_mean": round(float(rec[1:, 5].mean()), 1), "inliers_final_mean": round(float(rec[1:, 29].mean()), 1), "map_points_mean": round(float(rec[1:, 28].mean()), 1),
                         "breakdown_ms_median": {TRACK_CALLS[i]: round(float(np.median(t[:, i])), 4) for i in range(6)},
                         "local_mapper_ms_median": {TRACK_CALLS[i]: round(float(np.median(t[kfr, i])), 4) for i in (6, 7)} if kfr.any() else {},
                         "allocator_note": "the sequence loop builds every Frame inside a CallerArena (the bump allocator the parity tests use): the quadtree's pointer tie (src/ORBextractor.cc:948) "
                                           "resolves the same way in both libraries, keypoint counts are identical"})
    return rows


def trace_stereo(orbx):
    """Timeline of the drop-in stereo constructor (marks set by shim/Frame_hip.cc and oracle/refslam_wrap.cc), last frames of a short run."""
    import oracle_lib
    lib = oracle_lib.slam_hip_lib()
    if lib is None or not hasattr(lib, "orbx_shim_trace"):
        return ""
    stereo_frame(orbx, 50, 0)                                   # engines, graphs and statics exist
    W, H, nf = 1241, 376, 2000
    fr = orbx.synth_sequence(9, 8, W, H, views_per_scene=2, step=(6, 0))
    aL = (ctypes.c_void_p * 4)(*[fr[2 * i].ctypes.data for i in range(4)])
    aR = (ctypes.c_void_p * 4)(*[fr[2 * i + 1].ctypes.data for i in range(4)])
    mean, med, nl, nm = ctypes.c_double(), ctypes.c_double(), ctypes.c_int(), ctypes.c_int()
    lib.orbx_shim_trace.argtypes = [ctypes.c_int]
    lib.orbx_shim_trace(1)
    lib.orbslam_stereo_frame_bench(aL, aR, 4, W, H, W, nf, 1.2, 8, 20, 7, 718.856, 718.856, 607.19, 185.2, 386.1448, 35.0, 3, ctypes.byref(mean), ctypes.byref(med),
                                   ctypes.byref(nl), ctypes.byref(nm))
    lib.orbx_shim_trace(0) if False else None
    buf = ctypes.create_string_buffer(1 << 16)
    lib.orbx_shim_trace_dump.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.orbx_shim_trace_dump(buf, len(buf))
    lines = buf.value.decode().splitlines()
    # the last constructor only (the statics' first-frame path and the warm-up calls are above it)
    starts = [i for i, l in enumerate(lines) if "Frame::Frame(stereo) called" in l]
    return "\n".join(lines[starts[-1]:]) if starts else "\n".join(lines)


def measure(orbx, quick=False):
    """-> {"rows": [...], "digest": {...}}; quick = the few numbers bench.py's digest needs (a couple of seconds)."""
    L = _shim_lib(orbx)
    rows = []
    it1, itn = (150, 250) if quick else (300, 400)
    rows += one_thread(orbx, L, 640, 480, 1000, it1)
    if not quick:
        rows += one_thread(orbx, L, 1241, 376, 2000, it1)
    rows += threads(orbx, L, (8, 16) if quick else (1, 2, 4, 8, 16), itn, keeps=(0,) if quick else (0, 1))
    rows += threads(orbx, L, (16,), itn, keeps=(1, 3) if quick else (3,))
    try:
        rows += stereo_frame(orbx, 100 if quick else 300, 0 if quick else 8)
    except Exception as e:      # noqa: BLE001  (the drop-in library is only there when oracle/_ref was built)
        rows.append({"call": "stereo Frame constructor", "error": repr(e)})
    try:
        rows += tracking(orbx, 2 if quick else 4, 1)
    except Exception as e:      # noqa: BLE001
        rows.append({"call": "tracked frame", "error": repr(e)})
    if not quick:
        os.environ["ORBX_COMBINE"] = "0"      # read when a handle is created: the round-3 behaviour (one graph per handle) beside the combiner's
        try:
            rows += one_thread(orbx, L, 640, 480, 1000, it1)
            rows += threads(orbx, L, (1, 8, 16), itn, keeps=(0,))
            rows += stereo_frame(orbx, 200, 0)
        finally:
            del os.environ["ORBX_COMBINE"]

    def pick(**kw):
        for r in rows:
            if all(r.get(k) == v for k, v in kw.items()):
                return r
        return {}
    dig = {"us_1thread": pick(size="640x480", host_pyramid=False, host_pyramid_unread=False, combiner=True).get("mean_us"),
           "us_1thread_hostpyr": pick(size="640x480", host_pyramid=True, host_pyramid_views=False, combiner=True).get("mean_us"),
           "us_1thread_hostpyr_views": pick(size="640x480", host_pyramid=True, host_pyramid_views=True, combiner=True).get("mean_us"),
           "fps_8threads": pick(threads=8, host_pyramid=False, host_pyramid_unread=False, combiner=True).get("frames_per_s"),
           "fps_16threads": pick(threads=16, host_pyramid=False, host_pyramid_unread=False, combiner=True).get("frames_per_s"),
           "fps_16threads_hostpyr": pick(threads=16, host_pyramid=True, host_pyramid_views=False, combiner=True).get("frames_per_s"),
           "fps_16threads_hostpyr_views": pick(threads=16, host_pyramid=True, host_pyramid_views=True, combiner=True).get("frames_per_s"),
           "stereo_frame_ctor_us": pick(library="liborbslam_hip.so (drop-in)", combiner=True).get("mean_us"),
           "stereo_frame_ctor_median_us": pick(library="liborbslam_hip.so (drop-in)", combiner=True).get("median_us"),
           "track_ms_median": pick(kind="track", library="liborbslam_hip.so (drop-in)").get("track_ms_median"),
           "track_ms_mean": pick(kind="track", library="liborbslam_hip.so (drop-in)").get("track_ms_mean"),
           "ref_track_ms_median": pick(kind="track", library="liborbslam.so (reference CPU path on cvshim)").get("track_ms_median"),
           "ref_track_ms_mean": pick(kind="track", library="liborbslam.so (reference CPU path on cvshim)").get("track_ms_mean"),
           "track_breakdown_ms": pick(kind="track", library="liborbslam_hip.so (drop-in)").get("breakdown_ms_median"),
           "lba_in_loop_ms": (pick(kind="track", library="liborbslam_hip.so (drop-in)").get("local_mapper_ms_median") or {}).get(TRACK_CALLS[7]),
           "source": "tools/latency_shim.py: C++ loops over the reference's call shapes (tests/shim_wrap.cc, oracle/refslam_wrap.cc)"}
    return {"rows": rows, "digest": dig}


if __name__ == "__main__":
    orbx = importlib.import_module("self_commit_orb-slam2_amd")
    if "--trace" in sys.argv:
        print(trace_stereo(orbx))
        sys.exit(0)
    out = measure(orbx, quick="--quick" in sys.argv)
    for r in out["rows"]:
        print(json.dumps(r), flush=True)
    print(json.dumps({"digest": out["digest"]}), flush=True)
