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


# ---------------- SearchByProjection(CurrentFrame, LastFrame, th, bMono) ----------------
def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_last_case(seed, motion_z=0.0, n=1500, nl=1400, W=640, H=480, crowded=False):
    rng = np.random.default_rng(seed)
    fr, _ = make_case(seed, n=n, m=4, W=W, H=H, crowded=crowded)
    cam = (517.3, 516.5, 318.6, 255.3, 40.0)
    fx, fy, cx, cy, bf = cam
    Tc = np.eye(4)
    Tc[:3, :3] = _rot(0.01, -0.02, 0.015)
    Tc[:3, 3] = [0.05, -0.02, 0.1]
    Tl = np.eye(4)
    Tl[:3, :3] = _rot(0.0, 0.01, 0.0)
    # tlc.z = (Rlw*twc + tlw).z decides forward / backward against mb = bf/fx (~0.077)
    twc = -Tc[:3, :3].T @ Tc[:3, 3]
    Tl[:3, 3] = -Tl[:3, :3] @ twc + np.array([0.0, 0.0, motion_z])
    k7 = fr["kps7"]
    src = rng.integers(0, n, nl)
    z = rng.uniform(1.0, 8.0, nl)
    u = k7[src, 0] + rng.normal(0, 2.0, nl)
    v = k7[src, 1] + rng.normal(0, 2.0, nl)
    Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    behind = rng.random(nl) < 0.03
    Xc[behind, 2] *= -1                                       # invzc < 0
    Xw = (Tc[:3, :3].T @ (Xc - Tc[:3, 3]).T).T
    ld = fr["desc"][src].copy()
    for i in range(nl):
        for b in rng.integers(0, 256, rng.integers(0, 40)):
            ld[i, b >> 3] ^= np.uint8(1 << (b & 7))
    lk = np.zeros((nl, 7), np.float32)
    lk[:, 0], lk[:, 1] = u, v
    lk[:, 2] = 31
    lk[:, 3] = (k7[src, 3] + rng.normal(0, 5, nl)) % 360
    wild = rng.random(nl) < 0.15                               # inconsistent rotations: pruned by the histogram
    lk[wild, 3] = rng.uniform(0, 360, int(wild.sum()))
    lk[:, 5] = np.clip(k7[src, 5] + rng.integers(-1, 2, nl), 0, 7)
    lk[:, 6] = -1
    valid = np.ones(nl, np.uint8)
    valid[rng.random(nl) < 0.15] = 0                           # no MapPoint
    valid[rng.random(nl) < 0.05] = 2                           # outlier of the last frame
    fr = dict(fr, Tcw=Tc.astype(np.float32), cam=cam)
    last = dict(Tcw=Tl.astype(np.float32), valid=valid, pos=Xw.astype(np.float32), desc=ld, has_obs=(rng.random(nl) < 0.9).astype(np.uint8), kps7=lk)
    return fr, last


LAST_CASES = [(21, 0.0, False, 7.0, 0, 1), (22, 0.5, False, 15.0, 0, 1), (23, -0.5, False, 7.0, 0, 1), (24, 0.5, False, 15.0, 1, 1),
              (25, 0.0, True, 7.0, 0, 0), (26, 0.5, True, 15.0, 0, 1)]


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/liborbslam.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed,mz,crowded,th,mono,ori", LAST_CASES)
def test_last_frame_restatement_equals_reference(oracle, seed, mz, crowded, th, mono, ori):
    fr, last = make_last_case(seed, mz, crowded=crowded)
    want_n, want = oracle_lib.ref_search_by_projection_last(fr, last, th, mono, ori)
    got_n, got = oracle_lib.search_by_projection_last(oracle, fr, last, th, mono, ori)
    assert got_n == want_n and (np.maximum(got, -1) == want).all()      # -2 (assigned, then pruned) reads back as NULL from the Frame
    assert want_n > 150 and (not ori or (got == -2).any())


@pytest.mark.gpu
@pytest.mark.parametrize("seed,mz,crowded,th,mono,ori", LAST_CASES + [(27, -0.5, True, 15.0, 0, 1)])
def test_last_frame_hip_equals_restatement_and_reference(orbx, oracle, seed, mz, crowded, th, mono, ori):
    fr, last = make_last_case(seed, mz, crowded=crowded)
    want_n, want = oracle_lib.search_by_projection_last(oracle, fr, last, th, mono, ori)
    mt = orbx.ORBmatcher(0.9, bool(ori), max_features=2048)
    frame = dict(fr, kps=_struct_kps(orbx, fr["kps7"]))
    lastd = dict(last, kps=_struct_kps(orbx, last["kps7"]), valid=(last["valid"] == 1).astype(np.uint8))
    got_n, got = mt.SearchByProjectionLast(frame, lastd, th, mono)
    assert got_n == want_n and (got == want).all()
    if HAVE_REF:
        ref_n, ref = oracle_lib.ref_search_by_projection_last(fr, last, th, mono, ori)
        assert got_n == ref_n and (np.maximum(got, -1) == ref).all()
    mt.close()


# ---------------- replays under contention (k_proj_greedy / k_proj_last_greedy are parallel fixed points) ----------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", range(10))
def test_hip_replays_under_contention(orbx, oracle, case):
    """Many map points per feature, few distinct descriptors, points without observations (which do not block a feature
    and are overwritten later) and occupied features: the regime where the order of the reference's loop decides."""
    rng = np.random.default_rng(7000 + case)
    n = int(rng.choice([40, 150, 900]))
    m = int(rng.choice([500, 3000, 6000]))
    th = float(rng.choice([3.0, 8.0, 15.0]))
    fr, pts = make_case(7100 + case, n=n, m=m, crowded=bool(case & 1))
    pts["has_obs"] = (rng.random(m) < [0.0, 0.5, 1.0][case % 3]).astype(np.uint8)
    fr["occupied"] = (rng.random(n) < [0.0, 0.1, 0.6][(case // 3) % 3]).astype(np.uint8)
    want_n, want = oracle_lib.search_by_projection(oracle, fr, pts, th, 0.8)
    mt = orbx.ORBmatcher(0.8, True, max_features=2048)
    got_n, got = mt.SearchByProjection(dict(fr, kps=_struct_kps(orbx, fr["kps7"])), pts, th)
    assert got_n == want_n and (got == want).all()
    mt.close()
    nl = int(rng.choice([800, 1900]))
    fr2, last = make_last_case(7200 + case, float(rng.choice([0.0, 0.5, -0.5])), n=max(n, 60), nl=nl, crowded=bool(case & 1))
    last["has_obs"] = (rng.random(nl) < [0.0, 0.5, 1.0][case % 3]).astype(np.uint8)
    fr2["occupied"] = (rng.random(len(fr2["kps7"])) < 0.2).astype(np.uint8)
    mono, ori = case & 1, (case >> 1) & 1
    want_n, want = oracle_lib.search_by_projection_last(oracle, fr2, last, 15.0, mono, ori)
    mt = orbx.ORBmatcher(0.9, bool(ori), max_features=2048)
    got_n, got = mt.SearchByProjectionLast(dict(fr2, kps=_struct_kps(orbx, fr2["kps7"])),
                                           dict(last, kps=_struct_kps(orbx, last["kps7"]), valid=(last["valid"] == 1).astype(np.uint8)), 15.0, mono)
    assert got_n == want_n and (got == want).all()
    mt.close()


@pytest.mark.gpu
def test_hip_device_form_batch_of_frames(orbx, oracle):
    """orbx_search_by_projection_device with nframes = 3 (device pointers, frames of different sizes padded to the capacity):
    every frame must equal its single-frame result."""
    import ctypes
    import torch
    cases = [make_case(8001, n=900, m=2500, crowded=True), make_case(8002, n=300, m=4000), make_case(8003, n=1500, m=100)]
    B, capF, capP, th, ratio = 3, 1600, 4096, 5.0, 0.8
    mt = orbx.ORBmatcher(ratio, True, max_features=capF, max_pairs=B)
    k = np.zeros((B, capF), orbx.KEYPOINT_DTYPE)
    desc = np.zeros((B, capF, 32), np.uint8); ur = np.zeros((B, capF), np.float32); occ = np.zeros((B, capF), np.uint8)
    px = np.zeros((B, capP), np.float32); py = px.copy(); pxr = px.copy(); vc = px.copy()
    lvl = np.zeros((B, capP), np.int32); inv = np.zeros((B, capP), np.uint8); obs = inv.copy(); md = np.zeros((B, capP, 32), np.uint8)
    cn, cm = np.zeros(B, np.int32), np.zeros(B, np.int32)
    for f, (fr, pts) in enumerate(cases):
        n, m = len(fr["kps7"]), len(pts["proj_x"])
        cn[f], cm[f] = n, m
        k[f, :n] = _struct_kps(orbx, fr["kps7"]); desc[f, :n] = fr["desc"]; ur[f, :n] = fr["u_right"]; occ[f, :n] = fr["occupied"]
        px[f, :m], py[f, :m], pxr[f, :m], vc[f, :m] = pts["proj_x"], pts["proj_y"], pts["proj_xr"], pts["view_cos"]
        lvl[f, :m], inv[f, :m], obs[f, :m], md[f, :m] = pts["level"], pts["in_view"], pts["has_obs"], pts["desc"]
    keep = []
    def dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()
        keep.append(t)
        return t.data_ptr()
    W, H = cases[0][0]["width"], cases[0][0]["height"]
    F = orbx.ProjectionFrame(dev(k), dev(desc), dev(ur), dev(occ), dev(cn), capF, B, 0.0, 0.0, float(np.float32(64) / np.float32(W)), float(np.float32(48) / np.float32(H)))
    P = orbx.ProjectionPoints(dev(px), dev(py), dev(pxr), dev(lvl), dev(vc), dev(inv), dev(obs), dev(md), dev(cm), capP)
    torch.cuda.synchronize()
    sf = np.ascontiguousarray(SCALES, np.float32)
    rc = mt._L.orbx_search_by_projection_device(mt._h, ctypes.byref(F), ctypes.byref(P), sf.ctypes.data_as(ctypes.c_void_p), len(sf), ctypes.c_float(th), ctypes.c_float(ratio))
    assert rc == 0
    got, _, got_n = mt.download(B)
    for f, (fr, pts) in enumerate(cases):
        want_n, want = oracle_lib.search_by_projection(oracle, fr, pts, th, ratio)
        n = len(fr["kps7"])
        assert got_n[f] == want_n and (got[f, :n] == want).all(), f
        assert (got[f, n:] == -1).all()
    mt.close()
