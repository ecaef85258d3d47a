"""ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:515-654, Tracking::MonocularInitialization):
order-dependent window matching of the level-0 features with the vMatchedDistance / override bookkeeping.
CPU: restatement vs the compiled reference on real Frames; gpu: HIP vs both.  Index-exact."""
import numpy as np
import pytest

import oracle_lib
from test_matcher import _rand_desc


def _frames(orbx, seed, n=1500):
    """F2 = F1 moved by a few pixels with descriptor noise; descriptors repeat so that several F1 features compete for
    the same F2 feature with different distances (override and vMatchedDistance paths), plus unrelated features."""
    rng = np.random.default_rng(seed)
    def kps(x, y, octave, ang):
        k = np.zeros(len(x), orbx.KEYPOINT_DTYPE)
        k["x"], k["y"], k["octave"], k["angle"], k["size"], k["class_id"] = x, y, octave, ang, 31, -1
        return k
    x, y = rng.uniform(20, 620, n), rng.uniform(20, 460, n)
    cl = rng.random(n) < 0.3                                           # clusters: neighbours within the window
    x[cl], y[cl] = x[rng.integers(0, n, int(cl.sum()))] + rng.normal(0, 3, int(cl.sum())), y[rng.integers(0, n, int(cl.sum()))] + rng.normal(0, 3, int(cl.sum()))
    octave = np.where(rng.random(n) < 0.6, 0, rng.integers(1, 8, n))
    ang = rng.uniform(0, 360, n)
    base = _rand_desc(rng, n // 4)
    d1 = base[rng.integers(0, len(base), n)]
    d2 = d1.copy()
    fl = rng.integers(0, 256, (n, 20))
    for i in range(n):
        for b in fl[i][: rng.integers(0, 20)]:
            d2[i, b >> 3] ^= 1 << (b & 7)
    perm = rng.permutation(n)
    ang2 = (ang + np.where(rng.random(n) < 0.85, 25.0 + rng.normal(0, 3, n), rng.uniform(0, 360, n))) % 360
    f1 = dict(kps=kps(x, y, octave, ang), desc=d1)
    f2 = dict(kps=kps((x + rng.normal(2, 3, n))[perm], (y + rng.normal(-1, 3, n))[perm], octave[perm], ang2[perm]), desc=d2[perm], width=640, height=480)
    prev = np.stack([x, y], 1).astype(np.float32)
    return f1, f2, prev


CASES = [(1, 10, 0.9, True), (2, 30, 0.9, True), (3, 100, 0.8, False), (4, 15, 0.6, True)]


@pytest.mark.skipif(oracle_lib.slam_lib() is None, reason="oracle/_ref/liborbslam.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed,window,ratio,ori", CASES)
def test_restatement_equals_reference(orbx, oracle, seed, window, ratio, ori):
    f1, f2, prev = _frames(orbx, seed)
    want_n, want, want_prev = oracle_lib.ref_search_for_initialization(f1, f2, prev, window, ratio, ori)
    got_n, got = oracle_lib.search_for_initialization(oracle, f1, f2, prev, window, ratio, ori)
    assert got_n == want_n and (got == want).all()
    assert want_n > 100 and (want >= 0).sum() == want_n
    ok = want >= 0
    assert (want_prev[ok, 0] == f2["kps"]["x"][want[ok]]).all() and (want_prev[~ok] == prev[~ok]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,window,ratio,ori", CASES)
def test_hip_equals_restatement(orbx, oracle, seed, window, ratio, ori):
    f1, f2, prev = _frames(orbx, 10 + seed, n=[1500, 4000, 300, 2500][seed - 1])
    want_n, want = oracle_lib.search_for_initialization(oracle, f1, f2, prev, window, ratio, ori)
    n = len(f1["kps"])
    got_n, got, got_prev = orbx.ORBmatcher(ratio, ori, max_features=n).SearchForInitialization(f1, f2, prev, window)
    assert got_n == want_n and (got == want).all()
    assert want_n > 20


@pytest.mark.gpu
@pytest.mark.skipif(oracle_lib.slam_lib() is None, reason="oracle/_ref/liborbslam.so not built (needs /root/reference)")
def test_dropin_equals_reference(orbx):
    import ctypes
    orbx.load_library()
    hip, ref = oracle_lib.slam_hip_lib(), oracle_lib.slam_lib()
    hip.orbx_shim_search_for_initialization_calls.restype = ctypes.c_ulong
    before = hip.orbx_shim_search_for_initialization_calls()
    for seed, window, ratio, ori in CASES:
        f1, f2, prev = _frames(orbx, 20 + seed)
        want = oracle_lib.ref_search_for_initialization(f1, f2, prev, window, ratio, ori, lib=ref)
        got = oracle_lib.ref_search_for_initialization(f1, f2, prev, window, ratio, ori, lib=hip)
        assert got[0] == want[0] and (got[1] == want[1]).all() and (got[2].view(np.uint32) == want[2].view(np.uint32)).all()
    assert hip.orbx_shim_search_for_initialization_calls() - before == len(CASES), "the HIP body was not the one linked"
