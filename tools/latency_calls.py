#!/usr/bin/env python3
"""Latency of the single-frame, host-array forms of the drop-in calls (what Tracking.cc / LocalMapping.cc
would see per frame through the shim): each timed call includes its uploads and downloads."""
import importlib, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
orbx = importlib.import_module("self_commit_orb-slam2_amd")


def bench(name, fn, n=30, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print("%-58s median %.3f ms  min %.3f ms" % (name, np.median(ts), ts.min()), flush=True)


W, H, nf = 640, 480, 1000
fr = orbx.synth_sequence(3, 4, W, H)
ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1)
bench("ORBextractor::operator() 640x480/1000", lambda: ext(fr[0]))
kA, dA = ext(fr[0])
kB, dB = ext(fr[1])
mt = orbx.ORBmatcher(0.7, True, max_features=4096, max_pairs=1)
bench("ORBmatcher::SearchByBoW (one node, %d x %d)" % (len(kA), len(kB)), lambda: mt.SearchByBoW(kA, dA, kB, dB))
grp = (np.arange(len(kA)) % 50).astype(np.int32)
grpB = (np.arange(len(kB)) % 50).astype(np.int32)
bench("ORBmatcher::SearchByBoW (50 nodes)", lambda: mt.SearchByBoW(kA, dA, kB, dB, groupsA=grp, groupsB=grpB))

rng = np.random.default_rng(5)
n, m = len(kB), 4000
T = np.eye(4, dtype=np.float32)
P = np.stack([rng.uniform(-4, 4, m), rng.uniform(-3, 3, m), rng.uniform(1, 10, m)], 1).astype(np.float32)
nrm = (-P / np.linalg.norm(P, axis=1, keepdims=True)).astype(np.float32)
d = np.linalg.norm(P, axis=1).astype(np.float32)
pts = dict(pos=P, normal=-nrm, max_distance=d * 1.5, min_distance=d * 0.4)
lsf = float(np.float32(np.log(np.float32(1.2))))
sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
mdesc = rng.integers(0, 256, (m, 32), dtype=np.uint8)
frus = lambda: mt.isInFrustum(T, (500.0, 500.0, 320.0, 240.0, 40.0), (0.0, 640.0, 0.0, 480.0), lsf, 8, pts, 0.5)
bench("Frame::isInFrustum x %d map points" % m, frus)
r = frus()
frame = dict(kps=kB, desc=dB, u_right=np.full(n, -1, np.float32), occupied=np.zeros(n, np.uint8), scale_factors=sf, width=W, height=H)
points = dict(proj_x=r["proj_x"], proj_y=r["proj_y"], proj_xr=r["proj_xr"], level=r["level"], view_cos=r["view_cos"], in_view=r["in_view"],
              has_obs=np.ones(m, np.uint8), desc=mdesc)
bench("ORBmatcher::SearchByProjection(F, %d points, th=3)" % m, lambda: mt.SearchByProjection(frame, points, 3.0))

try:
    fo = orbx.FrameOps(517.3, 516.5, 318.6, 255.3, (0.2624, -0.9531, -0.0054, 0.0026, 1.1633))
    b = fo.ComputeImageBounds(W, H)
    g = orbx.FrameGrid.from_bounds(b)
    bench("Frame::UndistortKeyPoints (%d)" % len(kB), lambda: fo.UndistortKeyPoints(kB))
    ku = fo.UndistortKeyPoints(kB)
    bench("Frame::AssignFeaturesToGrid", lambda: fo.AssignFeaturesToGrid(ku, g))
except Exception as e:   # signature drift in this helper script must not hide the other numbers
    print("FrameOps skipped:", repr(e))
try:
    V = orbx.Vocabulary(orbx.voc_synth.make_vocabulary(10, 4, 1))
    bench("ORBVocabulary::transform (k=10, L=4, %d descriptors)" % len(dB), lambda: V.transform(dB, 4))
except Exception as e:
    print("Vocabulary skipped:", repr(e))
try:
    WS, HS = 1241, 376
    frs = orbx.synth_sequence(9, 2, WS, HS)
    exs = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=WS, max_height=HS, max_batch=2)
    mts = orbx.ORBmatcher(0.7, True, max_features=exs.capacity, max_pairs=1)
    def stereo():
        dv = exs.upload(frs)
        exs.run_device(*dv)
        mts.compute_stereo_matches_device(exs, exs, [0], [1], 386.1448, 0.0)
        return mts.download_stereo(1)
    bench("stereo Frame: extract L+R 1241x376/2000 + ComputeStereoMatches", stereo)
except Exception as e:
    print("stereo skipped:", repr(e))

# ---- the remaining single-frame calls, on the inputs of the parity tests ----
sys.path.insert(0, str(ROOT / "tests"))
try:
    import test_projection as tp, test_search_init as tsi, test_pose_optimization as tpo
    frl, last = tp.make_last_case(21, 0.0)
    mtl = orbx.ORBmatcher(0.9, True, max_features=2048)
    framel = dict(frl, kps=tp._struct_kps(orbx, frl["kps7"]))
    lastd = dict(last, kps=tp._struct_kps(orbx, last["kps7"]), valid=(last["valid"] == 1).astype(np.uint8))
    bench("ORBmatcher::SearchByProjection(Current, Last) %d x %d" % (len(framel["kps"]), len(lastd["kps"])), lambda: mtl.SearchByProjectionLast(framel, lastd, 7.0, 0))
    f1, f2, prev = tsi._frames(orbx, 11, n=1500)
    mti = orbx.ORBmatcher(0.9, True, max_features=1500)
    bench("ORBmatcher::SearchForInitialization 1500 x 1500, window 100", lambda: mti.SearchForInitialization(f1, f2, prev, 100))
    pf = [tpo.make_frame(10, n=600)]
    po = orbx.PoseOptimizer(max_frames=8, max_features=2048)
    bench("Optimizer::PoseOptimization, one frame, 600 correspondences", lambda: po.PoseOptimization(pf))
except Exception as e:
    print("skipped:", repr(e))

# ---- LocalMapping / LoopClosing side ----
try:
    import test_fuse as tf, test_triangulation as tt
    SF = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    kf, Tt, sk, Ts, P, cdesc, rng = tf._scene(orbx, 13, nc=3000, nextra=1500)
    nc = len(P)
    Pc = P @ Tt[:3, :3].T + Tt[:3, 3]
    z = np.where(np.abs(Pc[:, 2]) < 1e-3, 1e-3, Pc[:, 2])
    lvl = rng.integers(0, 8, nc).astype(np.int32)
    pts = dict(u=(Pc[:, 0] / z * 500 + 320).astype(np.float32), v=(Pc[:, 1] / z * 500 + 240).astype(np.float32),
               ur=(Pc[:, 0] / z * 500 + 320 - 40.0 / z).astype(np.float32), level=lvl, radius=(3.0 * SF[lvl]).astype(np.float32),
               active=(rng.random(nc) < 0.9).astype(np.uint8), desc=cdesc)
    mtf = orbx.ORBmatcher(0.6, True, max_features=max(len(kf["kps"]), nc, 64))
    bench("ORBmatcher::Fuse search, %d points x %d features" % (nc, len(kf["kps"])), lambda: mtf.FuseSearch(kf, pts, True))
    q = dict(u=pts["u"], v=pts["v"], radius=(6.0 * SF[lvl]).astype(np.float32), min_level=lvl - 1, max_level=(lvl + 1).astype(np.int32),
             active=pts["active"], desc=cdesc, window_int_bounds=False)
    frame = dict(kps=kf["kps"], desc=kf["desc"], blocked=np.zeros(len(kf["kps"]), np.uint8), width=640, height=480)
    bench("SearchByProjection (loop / reloc overloads) greedy area search", lambda: mtf.AreaSearchGreedy(frame, q, 100))
    k1, k2, T1, T2, F12 = tt._scene(orbx, 12, n=2000, forward=False, stereo_frac=0.3)
    mtt = orbx.ORBmatcher(0.6, False, max_features=2048)
    epi = np.array([5000.0, 300.0], np.float32)
    bench("ORBmatcher::SearchForTriangulation 2000 x 2000", lambda: mtt.SearchForTriangulation(k1, k2, F12, epi, tt.SF, tt.SIGMA2, False))
except Exception as e:
    print("LocalMapping part skipped:", repr(e))
