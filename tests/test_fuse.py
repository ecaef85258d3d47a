"""ORBmatcher::Fuse, both overloads (reference src/ORBmatcher.cc:1020-1312; LocalMapping::SearchInNeighbors,
LoopClosing::SearchAndFuse): the per-point search on the device.  CPU: the restatement against the reference run
point by point on a real KeyFrame; gpu: HIP against the restatement and the reference."""
import numpy as np
import pytest

import oracle_lib
from test_matcher import _rand_desc

SF = np.array([1.0, 1.2, 1.44, 1.728, 2.0736, 2.48832, 2.985984, 3.5831808], np.float32)
INV_SIGMA2 = (np.float32(1.0) / (SF * SF)).astype(np.float32)
TH_LOW = 50


def _pose(rng, scale=1.0):
    a = rng.normal(0, 0.02, 3)
    ca, sa = np.cos(a), np.sin(a)
    Rx = np.array([[1, 0, 0], [0, ca[0], -sa[0]], [0, sa[0], ca[0]]])
    Ry = np.array([[ca[1], 0, sa[1]], [0, 1, 0], [-sa[1], 0, ca[1]]])
    Rz = np.array([[ca[2], -sa[2], 0], [sa[2], ca[2], 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = rng.normal(0, 0.15, 3) * scale
    return T.astype(np.float32)


def _scene(orbx, seed, nc=700, nextra=300, stereo_frac=0.4):
    """A source KeyFrame that created nc map points, and a target KeyFrame that sees most of them again (with
    measurement noise, sometimes in the wrong octave, sometimes with a very different descriptor) plus unrelated features."""
    rng = np.random.default_rng(seed)
    Ts, Tt = np.eye(4, dtype=np.float32), _pose(rng)
    P = np.stack([rng.uniform(-3, 3, nc), rng.uniform(-2, 2, nc), rng.uniform(2.5, 9, nc)], 1).astype(np.float32)
    P[rng.random(nc) < 0.03, 2] = -1.0                                   # behind the target camera
    src_oct = rng.integers(0, 8, nc)
    sk = np.zeros(nc, orbx.KEYPOINT_DTYPE)
    sk["x"], sk["y"] = P[:, 0] / np.abs(P[:, 2]) * 500 + 320, P[:, 1] / np.abs(P[:, 2]) * 500 + 240
    sk["size"], sk["octave"], sk["class_id"] = 31, src_oct, -1
    cand_desc = _rand_desc(rng, max(nc // 2, 1))[rng.integers(0, max(nc // 2, 1), nc)]    # descriptors repeat: equal distances happen
    Pc = P @ Tt[:3, :3].T + Tt[:3, 3]
    z = np.where(np.abs(Pc[:, 2]) < 1e-3, 1e-3, Pc[:, 2])
    uv = Pc[:, :2] / z[:, None] * 500 + np.array([320, 240])
    dist_s, dist_t = np.linalg.norm(P, axis=1), np.linalg.norm(P - (-Tt[:3, :3].T @ Tt[:3, 3]), axis=1)
    lvl = np.clip(np.ceil(np.log(dist_s * SF[src_oct] / dist_t) / np.log(1.2)), 0, 7).astype(int)
    n = nc + nextra
    k = np.zeros(n, orbx.KEYPOINT_DTYPE)
    k["size"], k["class_id"], k["response"] = 31, -1, 40
    k["x"][:nc] = uv[:, 0] + rng.normal(0, 0.7, nc) * SF[lvl]
    k["y"][:nc] = uv[:, 1] + rng.normal(0, 0.7, nc) * SF[lvl]
    k["octave"][:nc] = np.clip(lvl - rng.integers(0, 2, nc) + (rng.random(nc) < 0.1) * 2, 0, 7)
    k["x"][nc:], k["y"][nc:] = rng.uniform(0, 640, nextra), rng.uniform(0, 480, nextra)
    k["octave"][nc:] = rng.integers(0, 8, nextra)
    desc = np.zeros((n, 32), np.uint8)
    desc[:nc] = cand_desc
    flips = rng.integers(0, 256, (nc, 40))
    nflip = np.where(rng.random(nc) < 0.15, 40, rng.integers(0, 12, nc))  # 15 %: far beyond TH_LOW
    for i in range(nc):
        for b in flips[i][: nflip[i]]:
            desc[i, b >> 3] ^= 1 << (b & 7)
    desc[nc:] = _rand_desc(rng, nextra)
    ur = np.full(n, -1.0, np.float32)
    st = rng.random(nc) < stereo_frac
    ur[:nc][st] = (k["x"][:nc] - 40.0 / z)[st] + rng.normal(0, 0.7, int(st.sum()))
    perm = rng.permutation(n)
    kf = dict(kps=k[perm], desc=desc[perm], u_right=ur[perm], inv_level_sigma2=INV_SIGMA2, width=640, height=480)
    return kf, Tt, sk, Ts, P, cand_desc, rng


def _sim3(Tt, s=1.1):
    S = Tt.copy()
    S[:3, :] *= np.float32(s)
    return S


@pytest.mark.skipif(oracle_lib.slam_lib() is None, reason="oracle/_ref/liborbslam.so not built (needs /root/reference)")
@pytest.mark.parametrize("overload,seed", [(1, 1), (1, 2), (2, 3)])
def test_restatement_equals_reference_point_by_point(orbx, oracle, overload, seed):
    kf, Tt, sk, Ts, P, cdesc, rng = _scene(orbx, seed)
    nc = len(P)
    obs = np.full(nc, 4, np.int32)                               # enough observations to survive the undo between probes
    r = oracle_lib.ref_fuse(overload, kf, Tt, _sim3(Tt), np.full(len(kf["kps"]), -1), [], sk, Ts, P, cdesc, obs, [], 3.0, True)
    bi, bd = oracle_lib.fuse_best(oracle, kf, r["points"], overload == 1)
    got = np.where((r["points"]["active"] > 0) & (bd <= TH_LOW), bi, -1)
    assert (got == r["probe_idx"]).all()
    assert (r["probe_idx"] >= 0).sum() > 0.4 * nc and (r["probe_idx"] < 0).sum() > 0.1 * nc
    assert r["points"]["active"].sum() < nc                       # some points fail the visibility gates


@pytest.mark.gpu
@pytest.mark.parametrize("chi2,seed,nc", [(True, 11, 700), (False, 12, 700), (True, 13, 3000), (True, 14, 1), (False, 15, 65)])
def test_hip_equals_restatement(orbx, oracle, chi2, seed, nc):
    kf, Tt, sk, Ts, P, cdesc, rng = _scene(orbx, seed, nc=nc, nextra=max(nc // 2, 5))
    # the preparation of step 1 in float32 (any values are legal inputs of the search)
    Pc = P @ Tt[:3, :3].T + Tt[:3, 3]
    z = np.where(np.abs(Pc[:, 2]) < 1e-3, 1e-3, Pc[:, 2])
    lvl = rng.integers(0, 8, nc).astype(np.int32)
    pts = dict(u=(Pc[:, 0] / z * 500 + 320).astype(np.float32), v=(Pc[:, 1] / z * 500 + 240).astype(np.float32),
               ur=(Pc[:, 0] / z * 500 + 320 - 40.0 / z).astype(np.float32), level=lvl, radius=(3.0 * SF[lvl]).astype(np.float32),
               active=(rng.random(nc) < 0.9).astype(np.uint8), desc=cdesc)
    for bounds in ({}, dict(min_x=-17.6, min_y=-9.3, max_x=661.2, max_y=492.8)):      # undistorted bounds: the KeyFrame window uses their int part
        kf2 = dict(kf, **bounds)
        want_i, want_d = oracle_lib.fuse_best(oracle, kf2, pts, chi2)
        mt = orbx.ORBmatcher(0.6, True, max_features=max(len(kf["kps"]), nc, 64))
        got_i, got_d = mt.FuseSearch(kf2, pts, chi2)
        assert (got_i == want_i).all() and (got_d == want_d).all()
        assert nc < 100 or (want_d <= TH_LOW).sum() > 0.1 * nc            # (levels are random here: most windows miss the right octave)


@pytest.mark.gpu
@pytest.mark.skipif(oracle_lib.slam_lib() is None, reason="oracle/_ref/liborbslam.so not built (needs /root/reference)")
@pytest.mark.parametrize("overload,seed", [(1, 21), (2, 22)])
def test_hip_equals_reference_point_by_point(orbx, overload, seed):
    kf, Tt, sk, Ts, P, cdesc, rng = _scene(orbx, seed)
    nc = len(P)
    r = oracle_lib.ref_fuse(overload, kf, Tt, _sim3(Tt), np.full(len(kf["kps"]), -1), [], sk, Ts, P, cdesc, np.full(nc, 4, np.int32), [], 3.0, True)
    mt = orbx.ORBmatcher(0.6, True, max_features=max(len(kf["kps"]), nc))
    bi, bd = mt.FuseSearch(kf, r["points"], overload == 1)
    got = np.where((r["points"]["active"] > 0) & (bd <= TH_LOW), bi, -1)
    assert (got == r["probe_idx"]).all()
