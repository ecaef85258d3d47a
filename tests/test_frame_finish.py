"""Frame::UndistortKeyPoints + ComputeImageBounds + AssignFeaturesToGrid (reference src/Frame.cc:899-1004,
460-491): restatement vs the compiled reference (CPU), HIP vs both and vs the golden fixtures of the
reference's monocular Frame constructor (gpu).  Bit-exact floats, index-exact grid."""
from pathlib import Path

import ctypes

import numpy as np
import pytest

import oracle_lib

GOLDEN = sorted((Path(__file__).resolve().parent / "golden" / "slam").glob("mono_*.npz"))
CAMS = {  # Examples/Monocular/TUM1.yaml, TUM2.yaml, EuRoC.yaml, TUM3.yaml
    "tum1": (640, 480, [517.306408, 516.469215, 318.643040, 255.313989], [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]),
    "tum2": (640, 480, [520.908620, 521.007327, 325.141442, 249.701764], [0.231222, -0.784899, -0.003257, -0.000105, 0.917205]),
    "euroc": (752, 480, [458.654, 457.296, 367.215, 248.375], [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]),
    "tum3": (640, 480, [535.4, 539.2, 320.1, 247.6], [0.0, 0.0, 0.0, 0.0]),
}


def _kps7(rng, n, W, H):
    k = np.zeros((n, 7), np.float32)
    k[:, 0] = rng.uniform(0, W, n)
    k[:, 1] = rng.uniform(0, H, n)
    k[:, 2] = 31.0
    k[:, 3] = rng.uniform(0, 360, n)
    k[:, 4] = rng.integers(7, 200, n)
    k[:, 5] = rng.integers(0, 8, n)
    k[:, 6] = -1
    return k


def _struct(orbx, k7):
    k = np.zeros(len(k7), orbx.KEYPOINT_DTYPE)
    for j, c in enumerate(("x", "y", "size", "angle", "response")):
        k[c] = k7[:, j]
    k["octave"] = k7[:, 5].astype(np.int32)
    k["class_id"] = k7[:, 6].astype(np.int32)
    return k


def _same(got, want):
    assert (got["bounds"].view(np.uint32) == want["bounds"].view(np.uint32)).all()
    assert (got["gridInv"].view(np.uint32) == want["gridInv"].view(np.uint32)).all()
    assert (np.asarray(got["kpsUn"], np.float32).view(np.uint32) == np.asarray(want["kpsUn"], np.float32).view(np.uint32)).all()
    assert (got["gridOff"] == want["gridOff"]).all()
    assert (got["gridIdx"] == want["gridIdx"]).all()


def test_golden_present():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: p.stem)
def test_restatement_matches_reference_golden(orbx, oracle, path):
    z = np.load(path)
    k, _ = oracle.restatement(int(z["nfeatures"])).extract(orbx.synth_frame(int(z["seed"]), int(z["W"]), int(z["H"])))
    assert k.shape == z["kps"].shape and (k.view(np.uint32) == z["kps"].view(np.uint32)).all()   # mvKeys of the reference's constructor
    got = oracle_lib.frame_finish(oracle, z["kps"], z["cam"], z["dist"], int(z["W"]), int(z["H"]))
    _same(got, z)
    if z["dist"][0] != 0:
        assert np.abs(z["kpsUn"][:, :2] - z["kps"][:, :2]).max() > 0.5       # the distortion does move points
        assert abs(z["bounds"][0]) > 5 and abs(z["bounds"][1] - int(z["W"])) > 5   # and so do the image corners
    assert z["gridOff"][-1] <= len(z["kps"])


@pytest.mark.skipif(oracle_lib.slam_lib() is None, reason="oracle/_ref/liborbslam.so not built (needs /root/reference)")
@pytest.mark.parametrize("cam", ["tum1", "euroc", "tum3"])
def test_restatement_equals_reference_frame(orbx, oracle, cam):
    W, H, K, dist = CAMS[cam]
    im = orbx.synth_frame(70 + len(cam), W, H)
    want = oracle_lib.ref_mono_frame(im, 1500, K[0], K[1], K[2], K[3], dist)
    got = oracle_lib.frame_finish(oracle, want["kps"], K, dist, W, H)
    _same(got, want)


def _hip(orbx, k7, W, H, K, dist):
    ops = orbx.FrameOps(K[0], K[1], K[2], K[3], dist)
    b = ops.ComputeImageBounds(W, H)
    grid = orbx.FrameGrid.from_bounds(b)
    g = np.array([grid.width_inv, grid.height_inv], np.float32)
    un = ops.UndistortKeyPoints(_struct(orbx, k7))
    off, idx = ops.AssignFeaturesToGrid(un, grid)
    ops.close()
    un7 = np.stack([un[c].astype(np.float32) for c in ("x", "y", "size", "angle", "response", "octave", "class_id")], 1) if len(un) else np.zeros((0, 7), np.float32)
    return dict(kpsUn=un7, bounds=b, gridInv=g, gridOff=off, gridIdx=idx)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: p.stem)
def test_hip_matches_reference_golden(orbx, path):
    z = np.load(path)
    _same(_hip(orbx, z["kps"], int(z["W"]), int(z["H"]), z["cam"], z["dist"]), z)


@pytest.mark.gpu
@pytest.mark.parametrize("cam", sorted(CAMS))
@pytest.mark.parametrize("n", [0, 1, 257, 1000, 4000, 8000])
def test_hip_equals_restatement(orbx, oracle, cam, n):
    W, H, K, dist = CAMS[cam]
    rng = np.random.default_rng(n + len(cam))
    k7 = _kps7(rng, n, W, H)
    if n >= 257:
        k7[:40, :2] = k7[40:80, :2]                       # crowded cells: several features per cell, order matters
        k7[100:140, 0], k7[100:140, 1] = 3.0, rng.uniform(0, H, 40)
    _same(_hip(orbx, k7, W, H, K, dist), oracle_lib.frame_finish(oracle, k7, K, dist, W, H))


@pytest.mark.gpu
def test_hip_finish_device_on_extractor_batch(orbx, oracle):
    W, H, K, dist = CAMS["tum1"]
    frames = np.stack([orbx.synth_frame(s, W, H) for s in (5, 6, 7)])
    ext = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=len(frames))
    kps, desc, counts = ext.extract_batch(frames)
    ops = orbx.FrameOps(K[0], K[1], K[2], K[3], dist)
    ops.finish_device(ext, orbx.FrameGrid.from_bounds(ops.ComputeImageBounds(W, H)))
    un, off, idx = ops.download(ext, len(frames))
    for f in range(len(frames)):
        n = int(counts[f])
        k7 = np.stack([kps[f][c][:n].astype(np.float32) for c in ("x", "y", "size", "angle", "response", "octave", "class_id")], 1)
        want = oracle_lib.frame_finish(oracle, k7, K, dist, W, H)
        got7 = np.stack([un[f][c][:n].astype(np.float32) for c in ("x", "y", "size", "angle", "response", "octave", "class_id")], 1)
        assert (got7.view(np.uint32) == want["kpsUn"].view(np.uint32)).all()
        assert (off[f] == want["gridOff"]).all()
        assert (idx[f][:off[f][-1]] == want["gridIdx"]).all()
    ops.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cam", ["tum1", "tum3"])      # distorted / rectified (mvKeysUn = mvKeys: no undistorted keypoints come back)
def test_hip_finish_frame_latency_form(orbx, oracle, cam):
    """orbx_frame_finish_begin / _end: the frame a single-frame extractor call left on the device, results in pinned memory (what the drop-in
    Frame constructors use), against the restatement; also without a grid, and the state error of an _end without a _begin."""
    W, H, K, dist = CAMS[cam]
    ext = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=W, max_height=H)
    ops = orbx.FrameOps(K[0], K[1], K[2], K[3], dist)
    grid = orbx.FrameGrid.from_bounds(ops.ComputeImageBounds(W, H))
    for seed in (31, 32, 33):
        kps, desc = ext.extract_with_pyramid(orbx.synth_frame(seed, W, H))[:2]
        n = len(kps)
        k7 = np.stack([kps[c].astype(np.float32) for c in ("x", "y", "size", "angle", "response", "octave", "class_id")], 1)
        want = oracle_lib.frame_finish(oracle, k7, K, dist, W, H)
        un, off, idx, cnt = ops.finish_frame(ext, grid)
        assert cnt == n and n > 800
        if float(dist[0]) == 0.0:
            assert un is None
        else:
            got7 = np.stack([un[c].astype(np.float32) for c in ("x", "y", "size", "angle", "response", "octave", "class_id")], 1)
            assert (got7.view(np.uint32) == want["kpsUn"].view(np.uint32)).all()
        assert (off == want["gridOff"]).all() and (idx == want["gridIdx"]).all()
        un2, off2, idx2, cnt2 = ops.finish_frame(ext, None)      # undistortion only
        assert off2 is None and idx2 is None and cnt2 == n
        if un is not None:
            assert (un2.view(np.uint8) == un.view(np.uint8)).all()
    with pytest.raises(orbx.OrbxError):
        ops._L.orbx_frame_finish_end.argtypes = [ctypes.c_void_p] * 5
        orbx._check(ops._L.orbx_frame_finish_end(ops._h, None, None, None, None))      # nothing begun
    ops.close(); ext.close()


@pytest.mark.gpu
def test_bad_arguments(orbx):
    with pytest.raises(Exception):
        orbx.FrameOps(500, 500, 320, 240, [0.1, 0, 0])      # 3 coefficients: not a form mDistCoef takes
    with pytest.raises(Exception):
        orbx.FrameOps(0, 500, 320, 240, [0.1, 0, 0, 0])


@pytest.mark.parametrize("cam", ["tum1", "tum2", "euroc"])
def test_restated_undistort_inverts_the_distortion_model(oracle, cam):
    """cv::undistortPoints is restated (OpenCV is not available: parity unpinned at that level).  Independent check by the definition:
    pushing the undistorted points through the forward Brown-Conrady model (k1 k2 p1 p2 k3) must give the measured pixels back - the
    five fixed-point iterations of the inverse converge to well under a hundredth of a pixel for these cameras inside the image."""
    W, H, K, dist = CAMS[cam]
    rng = np.random.default_rng(11)
    k7 = _kps7(rng, 800, W, H)
    un = oracle_lib.frame_finish(oracle, k7, K, dist, W, H)["kpsUn"]
    fx, fy, cx, cy = K
    d = list(dist) + [0.0] * (5 - len(dist))
    k1, k2, p1, p2, k3 = d
    x = (un[:, 0].astype(np.float64) - cx) / fx
    y = (un[:, 1].astype(np.float64) - cy) / fy
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    u, v = fx * xd + cx, fy * yd + cy
    err = np.hypot(u - k7[:, 0], v - k7[:, 1])
    # five iterations, like OpenCV 2.4 / 3.x: converged for the TUM cameras; the strongly distorted EuRoC camera (k1 = -0.28) keeps up to 0.3 px
    # at the image corners - a property of the reference's OpenCV call, reproduced here
    assert err.max() < (0.5 if cam == "euroc" else 0.02), err.max()
    assert np.median(err) < 0.01
