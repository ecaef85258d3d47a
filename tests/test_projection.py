"""ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) - the matcher of
Tracking::SearchLocalPoints (reference src/ORBmatcher.cc:70-175, src/Tracking.cc:1616) - with
Frame::GetFeaturesInArea and the feature grid: restatement and HIP kernels against the compiled
reference driven through a real Frame and real MapPoints."""
import numpy as np
import pytest

import oracle_lib

HAVE_REF = oracle_lib.slam_lib() is not None
SCALES = np.float32(1.2) ** np.arange(8, dtype=np.float32)


def make_case(seed, n=1500, m=1300, W=640, H=480, crowded=False):
    rng = np.random.default_rng(seed)
    k7 = np.zeros((n, 7), np.float32)
    span = 0.25 if crowded else 1.0           # crowded: many features per search window -> long candidate lists, ties
    k7[:, 0] = rng.uniform(5, 5 + (W - 10) * span, n)
    k7[:, 1] = rng.uniform(5, 5 + (H - 10) * span, n)
    k7[:, 2] = 31
    k7[:, 3] = rng.uniform(0, 360, n)
    k7[:, 5] = rng.integers(0, 8, n)
    k7[:, 6] = -1
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if crowded:                                # few distinct descriptors: equal distances everywhere
        desc = desc[rng.integers(0, 12, n)]
    u_right = np.where(rng.random(n) < 0.5, k7[:, 0] - rng.uniform(1, 40, n), -1).astype(np.float32)
    occupied = (rng.random(n) < 0.1).astype(np.uint8)
    src = rng.integers(0, n, m)                # every map point observes (a noisy version of) some feature
    md = desc[src].copy()
    for i in range(m):
        for b in rng.integers(0, 256, rng.integers(0, 40)):
            md[i, b >> 3] ^= np.uint8(1 << (b & 7))
    rnd = rng.random(m) < 0.1
    md[rnd] = rng.integers(0, 256, (int(rnd.sum()), 32), dtype=np.uint8)
    lvl = np.clip(k7[src, 5].astype(np.int32) + rng.integers(0, 2, m), 0, 7)
    px = (k7[src, 0] + rng.normal(0, 1.5, m)).astype(np.float32)
    py = (k7[src, 1] + rng.normal(0, 1.5, m)).astype(np.float32)
    pxr = np.where(u_right[src] > 0, u_right[src] + rng.normal(0, 2.0, m), px - 10).astype(np.float32)
    fr = dict(kps7=k7, desc=desc, u_right=u_right, occupied=occupied, scale_factors=SCALES, width=W, height=H)
    pts = dict(proj_x=px, proj_y=py, proj_xr=pxr, level=lvl, view_cos=rng.uniform(0.99, 1.0, m).astype(np.float32),
               in_view=(rng.random(m) < 0.9).astype(np.uint8), has_obs=(rng.random(m) < 0.9).astype(np.uint8), desc=md)
    return fr, pts


CASES = [(1, False, 1.0, 0.8), (2, False, 3.0, 0.8), (3, True, 1.0, 0.8), (4, True, 5.0, 0.9), (5, False, 1.0, 0.6)]


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/liborbslam.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed,crowded,th,ratio", CASES)
def test_restatement_equals_reference(oracle, seed, crowded, th, ratio):
    fr, pts = make_case(seed, crowded=crowded)
    want_n, want = oracle_lib.ref_search_by_projection(fr, pts, th, ratio)
    got_n, got = oracle_lib.search_by_projection(oracle, fr, pts, th, ratio)
    assert got_n == want_n and (got == want).all()
    assert want_n > 200


def _struct_kps(orbx, k7):
    k = np.zeros(len(k7), orbx.KEYPOINT_DTYPE)
    for j, c in enumerate(("x", "y", "size", "angle", "response")):
        k[c] = k7[:, j]
    k["octave"] = k7[:, 5].astype(np.int32)
    k["class_id"] = k7[:, 6].astype(np.int32)
    return k


@pytest.mark.gpu
@pytest.mark.parametrize("seed,crowded,th,ratio", CASES + [(6, True, 8.0, 0.8)])
def test_hip_equals_restatement_and_reference(orbx, oracle, seed, crowded, th, ratio):
    fr, pts = make_case(seed, crowded=crowded)
    want_n, want = oracle_lib.search_by_projection(oracle, fr, pts, th, ratio)
    mt = orbx.ORBmatcher(ratio, True, max_features=2048)
    frame = dict(fr, kps=_struct_kps(orbx, fr["kps7"]))
    got_n, got = mt.SearchByProjection(frame, pts, th)
    assert got_n == want_n and (got == want).all()
    if HAVE_REF:
        ref_n, ref = oracle_lib.ref_search_by_projection(fr, pts, th, ratio)
        assert got_n == ref_n and (got == ref).all()
    # empty inputs
    e_n, e = mt.SearchByProjection(frame, {k: v[:0] for k, v in pts.items()}, th)
    assert e_n == 0 and (e == -1).all()
    mt.close()
