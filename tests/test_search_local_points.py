"""Tracking::SearchLocalPoints (reference src/Tracking.cc:1760-1830) end to end on the device: Frame::isInFrustum over the local map points
(src/Frame.cc:608-742) feeding ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th) (src/ORBmatcher.cc:70-175) in ONE call
(orbx_search_local_points; shim/SearchLocalPoints.h is the body a maintainer puts into the Tracking function).  The all-reference library
runs the function's own three steps on a real Frame and real MapPoints (oracle/refslam_wrap.cc: orbslam_search_local_points); the drop-in
library runs the shim.  Everything the reference function leaves behind must be identical: F.mvpMapPoints, and per MapPoint
mbTrackInView, mTrackProjX / Y / XR, mnTrackScaleLevel, mTrackViewCos and the visibility counter."""
import ctypes

import numpy as np
import pytest

import oracle_lib
from test_frustum import _setup

HAVE_REF = oracle_lib.SLAM_SO.exists() and oracle_lib.SLAM_HIP_SO.exists()      # (libraries are loaded inside the tests, liborbx.so first)


def _call(lib, fr, sc, th):
    n, m = len(fr["k7"]), len(sc["pos"])
    assigned = np.full(max(n, 1), -9, np.int32)
    in_view, visible = np.zeros(m, np.uint8), np.zeros(m, np.int32)
    px, py, pxr, vc = (np.zeros(m, np.float32) for _ in range(4))
    lvl = np.zeros(m, np.int32)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.orbslam_search_local_points.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, ci] + [vp] * 8
    P = lambda a: a.ctypes.data_as(vp)
    keep = [np.ascontiguousarray(x) for x in (fr["k7"], fr["desc"], fr["u_right"], sc["T"].astype(np.float32).reshape(16), sc["Ts"].astype(np.float32).reshape(16),
                                               sc["src_k7"], sc["src_desc"], sc["pos"], fr["pre"], sc["bad"], sc["has_obs"])]
    nm = lib.orbslam_search_local_points(P(keep[0]), P(keep[1]), P(keep[2]), n, P(keep[3]), P(keep[4]), P(keep[5]), P(keep[6]), P(keep[7]), m, P(keep[8]), P(keep[9]),
                                         P(keep[10]), int(th), P(assigned), P(in_view), P(px), P(py), P(pxr), P(lvl), P(vc), P(visible))
    return dict(nm=nm, assigned=assigned[:n], in_view=in_view, proj_x=px, proj_y=py, proj_xr=pxr, level=lvl, view_cos=vc, visible=visible)


def _scene(orbx, seed, m=2500, clutter=600):
    """Map points seen from a source keyframe; a current frame whose features are noisy re-observations of the visible points + clutter."""
    rng = np.random.default_rng(seed)
    T, Ts, sk, pos = _setup(orbx, seed, m)
    src_k7 = np.zeros((m, 7), np.float32)
    src_k7[:, 2], src_k7[:, 5], src_k7[:, 6] = 31, sk["octave"], -1
    src_desc = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    r = oracle_lib.ref_is_in_frustum(T, Ts, sk, pos, 0.5)                     # where the points land in the current frame (scene construction only)
    vis = np.flatnonzero(r["in_view"])
    obs = vis[rng.random(len(vis)) < 0.8]
    n = len(obs) + clutter
    k7 = np.zeros((n, 7), np.float32)
    k7[:len(obs), 0] = r["proj_x"][obs] + rng.normal(0, 1.2, len(obs))
    k7[:len(obs), 1] = r["proj_y"][obs] + rng.normal(0, 1.2, len(obs))
    k7[:len(obs), 5] = np.clip(r["level"][obs] - rng.integers(0, 2, len(obs)), 0, 7)
    k7[len(obs):, 0], k7[len(obs):, 1] = rng.uniform(1, 639, clutter), rng.uniform(1, 479, clutter)
    k7[len(obs):, 5] = rng.integers(0, 8, clutter)
    k7[:, 0], k7[:, 1] = np.clip(k7[:, 0], 0.5, 639.4), np.clip(k7[:, 1], 0.5, 479.4)
    k7[:, 2], k7[:, 3], k7[:, 6] = 31, rng.uniform(0, 360, n), -1
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    desc[:len(obs)] = src_desc[obs]
    for i in range(len(obs)):
        for b in rng.integers(0, 256, rng.integers(0, 45)):
            desc[i, b >> 3] ^= np.uint8(1 << (b & 7))
    perm = rng.permutation(n)
    k7, desc = k7[perm], desc[perm]
    u_right = np.where(rng.random(n) < 0.4, k7[:, 0] - rng.uniform(1, 30, n), -1).astype(np.float32)
    pre = np.full(n, -1, np.int32)                                            # features that already hold a MapPoint (TrackWithMotionModel's matches)
    held = rng.choice(n, min(n // 8, m), replace=False)
    pre[held] = rng.choice(m, len(held), replace=False)
    bad = (rng.random(m) < 0.05).astype(np.uint8)
    has_obs = (rng.random(m) < 0.9).astype(np.uint8)
    return dict(k7=k7, desc=desc, u_right=u_right, pre=pre), dict(T=T, Ts=Ts, src_k7=src_k7, src_desc=src_desc, pos=pos, bad=bad, has_obs=has_obs)


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref libraries not built (need /root/reference)")
@pytest.mark.parametrize("seed,th", [(11, 1), (12, 3), (13, 5)])
def test_search_local_points_dropin_equals_reference(orbx, seed, th):
    orbx.load_library()
    ref, hip = oracle_lib.slam_lib(), oracle_lib.slam_hip_lib()
    hip.orbx_shim_search_local_points_calls.restype = ctypes.c_ulong
    before = hip.orbx_shim_search_local_points_calls()
    fr, sc = _scene(orbx, seed)
    want, got = _call(ref, fr, sc, th), _call(hip, fr, sc, th)
    assert hip.orbx_shim_search_local_points_calls() - before == 1
    assert got["nm"] == want["nm"] and want["nm"] > 150
    assert (got["assigned"] == want["assigned"]).all()
    asked = sc["bad"] == 0                # (a bad point is never looked at: its mbTrackInView is whatever the constructor left in memory)
    assert (got["in_view"][asked] == want["in_view"][asked]).all() and (got["visible"] == want["visible"]).all()
    ok = (want["in_view"] == 1) & asked
    assert 200 < ok.sum() < len(ok)
    for k in ("proj_x", "proj_y", "proj_xr", "view_cos"):
        assert (got[k][ok].view(np.uint32) == want[k][ok].view(np.uint32)).all(), k
    assert (got["level"][ok] == want["level"][ok]).all()
    assert ((got["assigned"] >= 0) & (got["assigned"] != fr["pre"])).sum() > 150          # the call really assigned new MapPoints


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref libraries not built (need /root/reference)")
def test_search_local_points_edge_cases(orbx):
    orbx.load_library()
    ref, hip = oracle_lib.slam_lib(), oracle_lib.slam_hip_lib()
    fr, sc = _scene(orbx, 21, m=400, clutter=50)
    # every point already seen or bad: nothing to project, nothing to match
    fr2 = dict(fr, pre=np.arange(len(fr["k7"]), dtype=np.int32) % 400)
    sc2 = dict(sc, bad=np.ones(400, np.uint8))
    want, got = _call(ref, fr2, sc2, 1), _call(hip, fr2, sc2, 1)
    assert got["nm"] == want["nm"] == 0 and (got["assigned"] == want["assigned"]).all() and (got["visible"] == want["visible"]).all()
    # a frame without features: the frustum test still runs over the points
    fr3 = dict(k7=np.zeros((0, 7), np.float32), desc=np.zeros((0, 32), np.uint8), u_right=np.zeros(0, np.float32), pre=np.zeros(0, np.int32))
    want, got = _call(ref, fr3, sc, 1), _call(hip, fr3, sc, 1)
    asked = sc["bad"] == 0
    assert got["nm"] == want["nm"] == 0 and (got["in_view"][asked] == want["in_view"][asked]).all() and (got["visible"] == want["visible"]).all()
    assert (want["in_view"][asked] == 1).sum() > 20


def _struct_kps(orbx, k7):
    k = np.zeros(len(k7), orbx.KEYPOINT_DTYPE)
    for j, c in enumerate(("x", "y", "size", "angle", "response")):
        k[c] = k7[:, j]
    k["octave"], k["class_id"] = k7[:, 5].astype(np.int32), k7[:, 6].astype(np.int32)
    return k


@pytest.mark.gpu
@pytest.mark.skipif(oracle_lib.slam_lib() is None, reason="oracle/_ref/liborbslam.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed,th", [(31, 1.0), (32, 4.0)])
def test_chained_call_equals_the_two_calls(orbx, seed, th):
    """orbx_search_local_points (C ABI) = orbx_is_in_frustum followed by orbx_search_by_projection on its outputs, without the round trip."""
    fr, sc = _scene(orbx, seed)
    from test_frustum import _setup
    _, _, sk, _ = _setup(orbx, seed, len(sc["pos"]))
    r = oracle_lib.ref_is_in_frustum(sc["T"], sc["Ts"], sk, sc["pos"], 0.5)       # normals / distance ranges of MapPoints created from the source frame
    mt = orbx.ORBmatcher(0.8, True, max_features=4096)
    frame = dict(kps=_struct_kps(orbx, fr["k7"]), desc=fr["desc"], u_right=fr["u_right"], occupied=(fr["pre"] >= 0).astype(np.uint8),
                 scale_factors=oracle_lib.SCALE_FACTORS, width=640, height=480)
    pts = dict(pos=sc["pos"], normal=r["normal"], max_distance=r["max_distance"], min_distance=r["min_distance"], desc=sc["src_desc"], has_obs=sc["has_obs"])
    cam = (500.0, 500.0, 320.0, 240.0, 40.0)
    for rep in range(3):                                                           # (staging buffers are reused across calls)
        nm, assigned, fv = mt.SearchLocalPoints(frame, sc["T"], cam, r["log_scale_factor"], pts, th)
        two = mt.isInFrustum(sc["T"], cam, (0.0, 640.0, 0.0, 480.0), r["log_scale_factor"], 8, pts, 0.5)
        assert (fv["in_view"] == two["in_view"]).all() and (two["in_view"] == r["in_view"]).all()
        ok = two["in_view"] > 0
        for k in ("proj_x", "proj_y", "proj_xr", "view_cos"):
            assert (fv[k][ok].view(np.uint32) == two[k][ok].view(np.uint32)).all(), k
        assert (fv["level"][ok] == two["level"][ok]).all()
        n2, a2 = mt.SearchByProjection(frame, dict(two, desc=sc["src_desc"], has_obs=sc["has_obs"]), th)
        assert nm == n2 and (assigned == a2).all() and nm > 100


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref libraries not built (need /root/reference)")
@pytest.mark.parametrize("seed,m,clutter,th", [(31, 1, 0, 1), (32, 3, 5, 3), (33, 65, 64, 1), (34, 300, 3000, 8), (35, 900, 2500, 5), (36, 4000, 100, 1), (37, 2500, 600, 8),
                                               (38, 64, 0, 3), (39, 1200, 1200, 2), (40, 2000, 3500, 8)])
def test_search_local_points_sizes_and_crowded_windows(orbx, seed, m, clutter, th):
    """The same comparison over shapes the three seeds above do not reach: one or a few map points, a frame without clutter, more map points than features and the
    reverse, and search windows (th = 5, 8: a radius of up to 114 px) that hold hundreds of features - the candidate walk of k_frustum_topk queues more than a wave's
    worth (rank counting does not apply) and more than its LDS queue takes (worked off in the middle of the scan), the replay meets full lists and rescans."""
    orbx.load_library()
    ref, hip = oracle_lib.slam_lib(), oracle_lib.slam_hip_lib()
    fr, sc = _scene(orbx, seed, m=m, clutter=clutter)
    if len(fr["k7"]) == 0:
        pytest.skip("the scene has no feature at all")
    want, got = _call(ref, fr, sc, th), _call(hip, fr, sc, th)
    assert got["nm"] == want["nm"], (seed, m, clutter, th)
    assert (got["assigned"] == want["assigned"]).all()
    asked = sc["bad"] == 0
    assert (got["in_view"][asked] == want["in_view"][asked]).all() and (got["visible"] == want["visible"]).all()
    ok = (want["in_view"] == 1) & asked
    for k in ("proj_x", "proj_y", "proj_xr", "view_cos"):
        assert (got[k][ok].view(np.uint32) == want[k][ok].view(np.uint32)).all(), k
    assert (got["level"][ok] == want["level"][ok]).all()
