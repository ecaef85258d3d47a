#!/usr/bin/env python3
"""Measured numbers for the BASELINE.json configs that are not bench.py's headline line:
  config 2  batch of 256 synthetic 640x480 frames, extract only
  config 2b the same batch through the device-resident tracking front end (undistort/grid, BoW transform, SearchByBoW in nodes)
  config 3  KITTI-shaped stereo 1241x376, 2000 features: extract L+R + complete ComputeStereoMatches,
            and + brute-force SearchByBoW L<->R
  config 5  LocalBundleAdjustment on the synthetic 50-KF / 5000-point window (GPU vs the CPU oracle)
Prints one JSON line per config.  GPU box only (python tools/bench_configs.py)."""
import importlib
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
orbx = importlib.import_module("self_commit_orb-slam2_amd")


def timed(fn, sync, steps, warmup=3):
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


def config2(B=256, W=640, H=480, nf=1000, parts=2):
    Bs = B // parts
    exts = [orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=Bs) for _ in range(parts)]
    frames = orbx.synth_sequence(1, B, W, H)
    devs = [e.upload(frames[k * Bs:(k + 1) * Bs]) for k, e in enumerate(exts)]
    dt = timed(lambda: [e.run_device(*d) for e, d in zip(exts, devs)], lambda: [e.sync() for e in exts], 20)
    return {"config": "2: 256 x 640x480 extract only", "frames_per_s": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 3)}


def config2b(B=256, W=640, H=480, nf=1000, parts=2):
    """The tracking front end, device resident: extract -> UndistortKeyPoints + AssignFeaturesToGrid (TUM1 camera) ->
    DBoW2 transform (synthetic k=10, L=4 vocabulary, levelsup=2: 100 FeatureVector nodes) -> SearchByBoW(KeyFrame, Frame)
    of consecutive frames inside their vocabulary nodes.  No host round trip between the stages."""
    Bs = B // parts
    voc_np = orbx.voc_synth.make_vocabulary(10, 4, 7, ragged=False)
    frames = orbx.synth_sequence(1, B, W, H)
    exts, ops, vocs, mts, devs, grids = [], [], [], [], [], []
    for k in range(parts):
        e = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=Bs)
        o = orbx.FrameOps(517.306408, 516.469215, 318.643040, 255.313989, [0.262383, -0.953104, -0.005358, 0.002628, 1.163314])
        exts.append(e); ops.append(o); vocs.append(orbx.Vocabulary(voc_np))
        mts.append(orbx.ORBmatcher(0.7, True, max_features=e.capacity, max_pairs=Bs))
        devs.append(e.upload(frames[k * Bs:(k + 1) * Bs]))
        grids.append(orbx.FrameGrid.from_bounds(o.ComputeImageBounds(W, H)))
    pa = np.arange(Bs, dtype=np.int32)
    pb = (pa + 1) % Bs

    def step():
        for e, o, v, m, d, g in zip(exts, ops, vocs, mts, devs, grids):
            e.run_device(*d)
            o.finish_device(e, g)
            v.transform_device(e, 2)
            kd, dd, cd, cap = e.results_device()
            ku, _ = o.keypoints_un_device()
            nd, _ = v.groups_device()
            fs = orbx.FeatureSet(ku.value, dd.value, cd.value, nd.value, None, cap, Bs)
            m.search_by_bow_device(fs, fs, pa, pb, mode=0, after=e)

    dt = timed(step, lambda: [x.sync() for x in exts + mts], 20)
    _, _, nm = mts[0].download(Bs)
    return {"config": "2b: 256 x 640x480 extract + undistort/grid + BoW transform (k=10,L=4) + SearchByBoW in vocabulary nodes",
            "frames_per_s": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "matches_per_pair": round(float(np.mean(nm)), 1)}


def config3(pairs=64, W=1241, H=376, nf=2000, bf=386.1448):
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2 * pairs)
    lefts = [orbx.synth_frame(1000 + i, W, H) for i in range(pairs)]
    rights = [orbx.synth_frame(1000 + i, W, H, orbx.SYNTH_STEREO_RIGHT) for i in range(pairs)]
    dev = ext.upload(lefts + rights)
    mt = orbx.ORBmatcher(0.7, True, max_features=ext.capacity, max_pairs=pairs)
    fl, fr = np.arange(pairs, dtype=np.int32), np.arange(pairs, 2 * pairs, dtype=np.int32)

    def step_stereo():
        ext.run_device(*dev)
        mt.compute_stereo_matches_device(ext, ext, fl, fr, bf, 0.0)

    def step_bow():
        ext.run_device(*dev)
        fs = orbx.ORBmatcher.features_of(ext, 2 * pairs)
        mt.search_by_bow_device(fs, fs, fl, fr, mode=0, after=ext)

    sync = lambda: (ext.sync(), mt.sync())
    dt1 = timed(step_stereo, sync, 10)
    u, z = mt.download_stereo(pairs)
    _, _, counts = ext.download(2 * pairs)
    dt2 = timed(step_bow, sync, 10)
    return {"config": "3: KITTI-shaped stereo 1241x376, 2000 feat", "stereo_pairs_per_s_extract_plus_ComputeStereoMatches": round(pairs / dt1, 1),
            "stereo_pairs_per_s_extract_plus_bruteforce_SearchByBoW": round(pairs / dt2, 1), "keypoints_per_image": round(float(counts.mean()), 1),
            "depths_per_pair": round(float((u >= 0).sum() / pairs), 1)}


def config5():
    import oracle_lib
    w = orbx.lba_synth.make_window(K=50, P=5000, seed=12345)
    opt = orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000)
    for _ in range(2):
        got = opt.LocalBundleAdjustment(w)
    t0 = time.perf_counter()
    for _ in range(5):
        got = opt.LocalBundleAdjustment(w)
    wall = (time.perf_counter() - t0) / 5
    ms, flops = opt.last_timing()
    orc = oracle_lib.Oracle()
    t0 = time.perf_counter()
    want = oracle_lib.local_bundle_adjustment(orc, w)
    cpu = time.perf_counter() - t0
    return {"config": "5: LBA 50 KF / 5000 points / %d edges" % w["E"], "gpu_wall_ms": round(wall * 1e3, 2), "gpu_kernel_ms": round(ms, 2),
            "fp64_gflops": round(flops / (ms * 1e-3) / 1e9, 1), "cpu_oracle_ms_1_core": round(cpu * 1e3, 1),
            "max_abs_pose_diff": float(np.abs(got["poses"].astype(np.float64) - want["poses"]).max()),
            "max_abs_point_diff": float(np.abs(got["points"].astype(np.float64) - want["points"]).max())}


def config6(B=256, n=400):
    """Optimizer::PoseOptimization for a batch of independent frames (one kernel, one workgroup per frame) vs the CPU restatement."""
    import oracle_lib
    from test_pose_optimization import make_frame
    frames = [make_frame(100 + i, n=n) for i in range(B)]
    po = orbx.PoseOptimizer(max_frames=B, max_features=n)
    for _ in range(2):
        got = po.PoseOptimization(frames)
    t0 = time.perf_counter()
    for _ in range(5):
        got = po.PoseOptimization(frames)
    wall = (time.perf_counter() - t0) / 5
    orc = oracle_lib.Oracle()
    t0 = time.perf_counter()
    want = [oracle_lib.pose_optimization(orc, f) for f in frames[:32]]
    cpu = (time.perf_counter() - t0) / 32
    dmax = max(float(np.abs(g["pose"].astype(np.float64) - w["pose"]).max()) for g, w in zip(got[:32], want))
    return {"config": "6: PoseOptimization, batch of %d frames x %d correspondences (host arrays in / out)" % (B, n), "frames_per_s": round(B / wall, 1),
            "ms_per_batch": round(wall * 1e3, 3), "cpu_oracle_ms_per_frame_1_core": round(cpu * 1e3, 3), "max_abs_pose_diff": dmax}


def config7(n=2000, m=4000):
    """Tracking back end on one frame (host-array forms, upload included): Frame::isInFrustum over m map points, then
    ORBmatcher::SearchByProjection(F, vpMapPoints, th) of the visible ones against n features."""
    rng = np.random.default_rng(5)
    mt = orbx.ORBmatcher(0.8, True, max_features=max(n, m))
    T = np.eye(4, dtype=np.float32)
    P = np.stack([rng.uniform(-4, 4, m), rng.uniform(-3, 3, m), rng.uniform(1, 10, m)], 1).astype(np.float32)
    nrm = (-P / np.linalg.norm(P, axis=1, keepdims=True)).astype(np.float32)
    d = np.linalg.norm(P, axis=1).astype(np.float32)
    pts = dict(pos=P, normal=-nrm, max_distance=d * 1.5, min_distance=d * 0.4)
    lsf = float(np.float32(np.log(np.float32(1.2))))
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    k = np.zeros(n, orbx.KEYPOINT_DTYPE)
    k["x"], k["y"], k["octave"], k["size"], k["class_id"] = rng.uniform(0, 640, n), rng.uniform(0, 480, n), rng.integers(0, 8, n), 31, -1
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    mdesc = rng.integers(0, 256, (m, 32), dtype=np.uint8)

    def step():
        r = mt.isInFrustum(T, (500.0, 500.0, 320.0, 240.0, 40.0), (0.0, 640.0, 0.0, 480.0), lsf, 8, pts, 0.5)
        frame = dict(kps=k, desc=desc, u_right=np.full(n, -1, np.float32), occupied=np.zeros(n, np.uint8), scale_factors=sf, width=640, height=480)
        points = dict(proj_x=r["proj_x"], proj_y=r["proj_y"], proj_xr=r["proj_xr"], level=r["level"], view_cos=r["view_cos"], in_view=r["in_view"],
                      has_obs=np.ones(m, np.uint8), desc=mdesc)
        return r, mt.SearchByProjection(frame, points, 3.0)

    for _ in range(3):
        r, (nm, _) = step()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    dt = (time.perf_counter() - t0) / 10
    return {"config": "7: isInFrustum (%d map points) + SearchByProjection (%d features), one frame, host arrays" % (m, n), "ms_per_frame": round(dt * 1e3, 3),
            "points_in_view": int(r["in_view"].sum())}


if __name__ == "__main__":
    for fn in (config2, config2b, config3, config5, config6, config7):
        print(json.dumps(fn()), flush=True)
