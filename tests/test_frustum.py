"""Frame::isInFrustum + MapPoint::PredictScale (reference src/Frame.cc:608-742, src/MapPoint.cc:571-586), the loop of
Tracking::SearchLocalPoints that prepares SearchByProjection.  CPU: the host-tabulated PredictScale thresholds reproduce
ceil(log(ratio)/logScaleFactor) exactly; gpu: the device op against the reference run on real Frame / MapPoint objects, bit-exact."""
import math

import numpy as np
import pytest

import oracle_lib
from test_fuse import _pose


def test_predict_scale_thresholds_reproduce_the_formula(orbx):
    lsf = float(np.float32(math.log(np.float32(1.2))))        # Frame::mfLogScaleFactor = log(mfScaleFactor) stored as float
    for nlevels, ls in ((8, lsf), (12, lsf), (5, float(np.float32(math.log(2.0)))), (1, lsf)):
        th = orbx.predict_scale_thresholds(ls, nlevels)
        assert len(th) == nlevels - 1 and (np.diff(th) > 0).all()
        rng = np.random.default_rng(nlevels)
        ratios = np.concatenate([np.exp(rng.uniform(-3, 4, 20000)).astype(np.float32), th, np.nextafter(th, np.float32(np.inf)), np.nextafter(th, np.float32(0)),
                                 np.array([1.0, 1e-30, 1e30], np.float32)])
        want = np.array([min(max(math.ceil(math.log(float(r)) / ls), 0), nlevels - 1) for r in ratios])
        got = (ratios[:, None] > th[None, :]).sum(1) if nlevels > 1 else np.zeros(len(ratios), int)
        assert (got == want).all()


def _setup(orbx, seed, n=3000):
    rng = np.random.default_rng(seed)
    Ts, T = _pose(rng), _pose(rng, 3.0)
    P = np.stack([rng.uniform(-6, 6, n), rng.uniform(-4, 4, n), rng.uniform(-2, 12, n)], 1).astype(np.float32)
    sk = np.zeros(n, orbx.KEYPOINT_DTYPE)
    sk["octave"], sk["size"], sk["class_id"] = rng.integers(0, 8, n), 31, -1
    return T, Ts, sk, P


@pytest.mark.gpu
@pytest.mark.skipif(oracle_lib.slam_lib() is None, reason="oracle/_ref/liborbslam.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed,cos_limit", [(1, 0.5), (2, 0.5), (3, 0.9)])
def test_hip_equals_reference(orbx, seed, cos_limit):
    T, Ts, sk, P = _setup(orbx, seed)
    r = oracle_lib.ref_is_in_frustum(T, Ts, sk, P, cos_limit)
    mt = orbx.ORBmatcher(0.8, True, max_features=len(P))
    got = mt.isInFrustum(T, (500.0, 500.0, 320.0, 240.0, 40.0), (0.0, 640.0, 0.0, 480.0), r["log_scale_factor"], 8,
                         dict(pos=P, normal=r["normal"], max_distance=r["max_distance"], min_distance=r["min_distance"]), cos_limit)
    assert (got["in_view"] == r["in_view"]).all()
    ok = r["in_view"] > 0
    assert 0.1 * len(P) < ok.sum() < 0.9 * len(P)
    for k in ("proj_x", "proj_y", "proj_xr", "view_cos"):
        assert (got[k][ok].view(np.uint32) == r[k][ok].view(np.uint32)).all(), k
    assert (got["level"][ok] == r["level"][ok]).all()
    assert len(np.unique(r["level"][ok])) >= 5
