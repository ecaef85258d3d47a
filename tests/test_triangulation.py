"""ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:810-1017, called by
LocalMapping::CreateNewMapPoints): restatement vs the compiled reference on real KeyFrames (CPU),
HIP vs both (gpu).  Index-exact."""
import numpy as np
import pytest

import oracle_lib
from test_matcher import _rand_desc

SF = np.array([1.2 ** l for l in range(8)], np.float32)     # what the reference driver's KeyFrames carry
SF[:] = [1.0, 1.2, 1.44, 1.728, 2.0736, 2.48832, 2.985984, 3.5831808]
SIGMA2 = (SF * SF).astype(np.float32)
K = np.array([[500, 0, 320], [0, 500, 240], [0, 0, 1]], np.float64)


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _scene(orbx, seed, n=900, forward=False, stereo_frac=0.0):
    """Two KeyFrames seeing the same random points: true matches lie on each other's epipolar lines, a share of
    them is displaced (fails CheckDistEpipolarLine), descriptors repeat (ties, competition for the same KF2 feature)."""
    rng = np.random.default_rng(seed)
    R2 = _rot(*(rng.normal(0, 0.03, 3)))
    t2 = np.array([0.05, 0.02, 0.6]) if forward else np.array([0.4, 0.05, 0.02])      # forward motion puts the epipole inside the image
    T1, T2 = np.eye(4), np.eye(4)
    T2[:3, :3], T2[:3, 3] = R2, t2
    P = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(3, 10, n)], 1)
    def proj(T):
        Pc = P @ T[:3, :3].T + T[:3, 3]
        return (Pc[:, :2] / Pc[:, 2:3]) * 500 + np.array([320, 240])
    uv1, uv2 = proj(T1), proj(T2)
    uv2 = uv2 + rng.normal(0, 0.4, uv2.shape)                       # measurement noise: some land near the chi2 bound
    off = rng.random(n) < 0.2
    uv2[off] += rng.normal(0, 12, (int(off.sum()), 2))              # these violate the epipolar constraint
    base = _rand_desc(rng, n // 3)
    d1 = base[rng.integers(0, len(base), n)]                        # every descriptor about three times
    d2 = d1.copy()
    flip = rng.integers(0, 256, (n, 6))
    for i in range(n):
        for b in flip[i][: rng.integers(0, 7)]:
            d2[i, b >> 3] ^= 1 << (b & 7)
    perm = rng.permutation(n)
    uv2, d2 = uv2[perm], d2[perm]
    def kps(uv, ang):
        k = np.zeros(n, orbx.KEYPOINT_DTYPE)
        k["x"], k["y"], k["size"], k["angle"], k["response"] = uv[:, 0], uv[:, 1], 31, ang, 50
        k["octave"] = rng.integers(0, 8, n)
        k["class_id"] = -1
        return k
    ang = rng.uniform(0, 360, n).astype(np.float32)
    ang2 = (ang + rng.normal(0, 4, n).astype(np.float32))[perm] % 360
    wild = rng.random(n) < 0.1
    ang2[wild] = rng.uniform(0, 360, int(wild.sum()))
    groups1 = rng.integers(0, 9, n).astype(np.int32) * 5
    groups2 = groups1[perm].copy()
    groups2[rng.random(n) < 0.1] = rng.integers(0, 9) * 5          # some true matches fall into different nodes
    groups1[rng.random(n) < 0.03] = -1
    kf1 = dict(kps=kps(uv1, ang), desc=d1, groups=groups1, has_mp=(rng.random(n) < 0.3).astype(np.uint8),
               u_right=np.where(rng.random(n) < stereo_frac, 100.0, -1.0).astype(np.float32))
    kf2 = dict(kps=kps(uv2, ang2), desc=d2, groups=groups2, has_mp=(rng.random(n) < 0.3).astype(np.uint8),
               u_right=np.where(rng.random(n) < stereo_frac, 100.0, -1.0).astype(np.float32))
    # LocalMapping::ComputeF12 (src/LocalMapping.cc:600-620): F12 = K1^-T [t12]x R12 K2^-1
    R12 = T1[:3, :3] @ T2[:3, :3].T
    t12 = -R12 @ T2[:3, 3] + T1[:3, 3]
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    F12 = (np.linalg.inv(K).T @ tx @ R12 @ np.linalg.inv(K)).astype(np.float32)
    return kf1, kf2, T1.astype(np.float32), T2.astype(np.float32), F12


CASES = [(1, False, 0.0, False, True), (2, True, 0.0, False, True), (3, True, 0.5, False, True), (4, False, 0.5, True, True), (5, True, 0.0, False, False)]


@pytest.mark.skipif(oracle_lib.slam_lib() is None, reason="oracle/_ref/liborbslam.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed,forward,stereo,only_stereo,ori", CASES)
def test_restatement_equals_reference(orbx, oracle, seed, forward, stereo, only_stereo, ori):
    kf1, kf2, T1, T2, F12 = _scene(orbx, seed, forward=forward, stereo_frac=stereo)
    want_n, want, epi = oracle_lib.ref_search_for_triangulation(kf1, kf2, T1, T2, F12, only_stereo, ori)
    got_n, got = oracle_lib.search_for_triangulation(oracle, kf1, kf2, F12, epi, SF, SIGMA2, only_stereo, ori)
    assert got_n == want_n and (got == want).all()
    assert want_n > 40                                            # the scene does produce matches ...
    ok1 = (kf1["has_mp"] == 0) & (kf1["groups"] >= 0)
    assert want_n < ok1.sum() * 0.8                               # ... and the gates do reject many
    if forward:
        assert 0 < epi[0] < 640 and 0 < epi[1] < 480              # the epipole gate is in play


def _hip(orbx, kf1, kf2, F12, epi, only_stereo, ori):
    mt = orbx.ORBmatcher(0.6, ori, max_features=max(len(kf1["kps"]), len(kf2["kps"]), 64))
    return mt.SearchForTriangulation(kf1, kf2, F12, epi, SF, SIGMA2, only_stereo)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,forward,stereo,only_stereo,ori", CASES)
def test_hip_equals_restatement(orbx, oracle, seed, forward, stereo, only_stereo, ori):
    kf1, kf2, T1, T2, F12 = _scene(orbx, 10 + seed, n=[900, 2000, 300, 1500, 64][seed - 1], forward=forward, stereo_frac=stereo)
    epi = np.array([300.0, 200.0], np.float32) if forward else np.array([5000.0, 300.0], np.float32)
    want_n, want = oracle_lib.search_for_triangulation(oracle, kf1, kf2, F12, epi, SF, SIGMA2, only_stereo, ori)
    got_n, got = _hip(orbx, kf1, kf2, F12, epi, only_stereo, ori)
    assert got_n == want_n and (got == want).all()


@pytest.mark.gpu
@pytest.mark.skipif(oracle_lib.slam_lib() is None, reason="oracle/_ref/liborbslam.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed,forward,stereo,only_stereo,ori", CASES)
def test_hip_equals_reference(orbx, seed, forward, stereo, only_stereo, ori):
    kf1, kf2, T1, T2, F12 = _scene(orbx, 20 + seed, forward=forward, stereo_frac=stereo)
    want_n, want, epi = oracle_lib.ref_search_for_triangulation(kf1, kf2, T1, T2, F12, only_stereo, ori)
    got_n, got = _hip(orbx, kf1, kf2, F12, epi, only_stereo, ori)
    assert got_n == want_n and (got == want).all()


@pytest.mark.gpu
def test_hip_exhausted_lists_and_ties(orbx, oracle):
    """Identical descriptors everywhere: every KF1 feature has dozens of distance-0 candidates, so the candidate
    lists overflow, the last-of-equals rule decides and the exact rescan path runs."""
    rng = np.random.default_rng(7)
    kf1, kf2, T1, T2, F12 = _scene(orbx, 31, n=400)
    base = _rand_desc(rng, 4)
    kf1["desc"] = base[rng.integers(0, 4, 400)]
    kf2["desc"] = base[rng.integers(0, 4, 400)]
    kf1["groups"][:] = 0
    kf2["groups"][:] = 0
    kf1["has_mp"][:] = 0
    kf2["has_mp"][:] = 0
    F0 = np.zeros(9, np.float32)
    F0[5], F0[7] = -1e-3, 1e-3                       # a = 0, b = 1e-3, c = -1e-3*y1: the line y = y1, passes for |dy| small
    kf2["kps"]["y"] = kf1["kps"]["y"][rng.permutation(400)]
    epi = np.array([9000.0, 9000.0], np.float32)
    want_n, want = oracle_lib.search_for_triangulation(oracle, kf1, kf2, F0, epi, SF, SIGMA2, False, False)
    got_n, got = _hip(orbx, kf1, kf2, F0, epi, False, False)
    assert got_n == want_n and (got == want).all()
    assert want_n > 100
