"""The two remaining ORBmatcher::SearchByProjection overloads (reference src/ORBmatcher.cc:388-513 loop closing,
:1731-1864 relocalisation): an order-dependent search (a feature taken by one map point is blocked for the next).
CPU: the restatement of the search against the reference's full functions on real objects; gpu: HIP against the
restatement, including exhausted candidate lists."""
import numpy as np
import pytest

import oracle_lib
from test_fuse import SF, TH_LOW, _scene, _sim3
from test_matcher import _rand_desc


def _setup(orbx, seed, overload):
    kf, Tt, sk, Ts, P, cdesc, rng = _scene(orbx, seed, nc=900, nextra=200)
    n, nc = len(kf["kps"]), len(P)
    sk["angle"] = rng.uniform(0, 360, nc)
    kf["kps"]["angle"] = rng.uniform(0, 360, n)
    holder = np.full(n, -1, np.int32)
    holder[rng.choice(n, 150, replace=False)] = np.arange(150)            # features that already hold some other map point
    lst = rng.permutation(nc).astype(np.int32) if overload == 3 else rng.choice(nc, 80, replace=False).astype(np.int32)
    if overload == 3:
        pre = rng.choice(np.nonzero(holder < 0)[0], 30, replace=False)
        holder[pre] = -2 - rng.choice(nc, 30, replace=False)             # candidates that are already matched: skipped (spAlreadyFound)
    return kf, Tt, sk, Ts, P, cdesc, holder, lst, rng


def _expected(orc_search, kf, r, holder, lst, overload, nc):
    """the reference's bookkeeping around the search, from the prepared queries"""
    pts = r["points"]
    already = set(int(-2 - h) for h in holder if h <= -2) if overload == 3 else set(int(c) for c in lst)
    order = [int(c) for c in lst] if overload == 3 else list(range(nc))
    act = pts["active"].copy()
    for c in already:
        act[c] = 0
    q = dict(u=pts["u"][order], v=pts["v"][order], radius=pts["radius"][order], min_level=pts["level"][order] - 1,
             max_level=pts["level"][order] + (0 if overload == 3 else 1), active=act[order], desc=pts["desc"][order], window_int_bounds=overload == 3)
    frame = dict(kps=kf["kps"], desc=kf["desc"], blocked=(holder != -1).astype(np.uint8), width=640, height=480)
    nm, asg, dst = orc_search(frame, q, TH_LOW if overload == 3 else 100)
    want = np.where(holder >= 0, 1000000 + holder, np.where(holder <= -2, -2 - holder, -1))
    for j, c in enumerate(order):
        if asg[j] >= 0:
            want[asg[j]] = c
    return nm, want, frame, q


@pytest.mark.skipif(oracle_lib.slam_lib() is None, reason="oracle/_ref/liborbslam.so not built (needs /root/reference)")
@pytest.mark.parametrize("overload,seed", [(3, 1), (3, 2), (5, 3), (5, 4)])
def test_restatement_equals_reference(orbx, oracle, overload, seed):
    kf, Tt, sk, Ts, P, cdesc, holder, lst, rng = _setup(orbx, seed, 3 if overload == 3 else 4)
    r = oracle_lib.ref_fuse(overload, kf, Tt, _sim3(Tt), holder, np.ones(150, np.int32), sk, Ts, P, cdesc, np.full(len(P), 1, np.int32), lst, 8.0, False)
    nm, want, _, _ = _expected(lambda f, q, d: oracle_lib.area_search_greedy(oracle, f, q, d), kf, r, holder, lst, 3 if overload == 3 else 4, len(P))
    assert r["nfused"] == nm and (r["holder"] == want).all()
    assert nm > 150


@pytest.mark.gpu
@pytest.mark.parametrize("seed,window_int", [(11, False), (12, True), (13, False)])
def test_hip_equals_restatement(orbx, oracle, seed, window_int):
    kf, Tt, sk, Ts, P, cdesc, rng = _scene(orbx, seed, nc=1500, nextra=300)
    nc, n = len(P), len(kf["kps"])
    Pc = P @ Tt[:3, :3].T + Tt[:3, 3]
    z = np.where(np.abs(Pc[:, 2]) < 1e-3, 1e-3, Pc[:, 2])
    lvl = rng.integers(0, 8, nc).astype(np.int32)
    q = dict(u=(Pc[:, 0] / z * 500 + 320).astype(np.float32), v=(Pc[:, 1] / z * 500 + 240).astype(np.float32), radius=(6.0 * SF[lvl]).astype(np.float32),
             min_level=lvl - 1, max_level=np.where(rng.random(nc) < 0.2, -1, lvl + 1).astype(np.int32), active=(rng.random(nc) < 0.9).astype(np.uint8), desc=cdesc,
             window_int_bounds=window_int)
    for bounds in ({}, dict(min_x=-17.6, min_y=-9.3, max_x=661.2, max_y=492.8)):
        frame = dict(kps=kf["kps"], desc=kf["desc"], blocked=(rng.random(n) < 0.2).astype(np.uint8), width=640, height=480, **bounds)
        want = oracle_lib.area_search_greedy(oracle, frame, q, 100)
        mt = orbx.ORBmatcher(0.9, True, max_features=max(n, nc))
        got = mt.AreaSearchGreedy(frame, q, 100)
        assert got[0] == want[0] and (got[1] == want[1]).all() and (got[2] == want[2]).all()
        assert want[0] > 0.25 * nc


@pytest.mark.gpu
def test_hip_exhausted_lists(orbx, oracle):
    """Hundreds of queries aimed at the same 40 features: every candidate list is full and taken after a while, the
    exact rescan decides."""
    rng = np.random.default_rng(5)
    n, m = 40, 600
    k = np.zeros(n, orbx.KEYPOINT_DTYPE)
    k["x"], k["y"], k["octave"], k["size"], k["class_id"] = rng.uniform(300, 340, n), rng.uniform(220, 260, n), rng.integers(0, 3, n), 31, -1
    base = _rand_desc(rng, 3)
    frame = dict(kps=k, desc=base[rng.integers(0, 3, n)], blocked=np.zeros(n, np.uint8), width=640, height=480)
    q = dict(u=np.full(m, 320, np.float32), v=np.full(m, 240, np.float32), radius=np.full(m, 60, np.float32), min_level=np.zeros(m, np.int32),
             max_level=np.full(m, 2, np.int32), active=np.ones(m, np.uint8), desc=base[rng.integers(0, 3, m)])
    want = oracle_lib.area_search_greedy(oracle, frame, q, 50)
    got = orbx.ORBmatcher(0.9, True, max_features=m).AreaSearchGreedy(frame, q, 50)
    assert got[0] == want[0] == n and (got[1] == want[1]).all() and (got[2] == want[2]).all()
