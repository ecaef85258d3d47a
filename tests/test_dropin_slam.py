"""Drop-in proof for the matcher / stereo surfaces.  oracle/_ref/liborbslam_hip.so is the
reference's own Frame.cc / KeyFrame.cc / MapPoint.cc / Map.cc / ORBmatcher.cc / DBoW2, built
UNMODIFIED, with exactly the change INTEGRATION.md describes: shim/ORBextractor.{h,cc} instead of
the reference's extractor and the HIP bodies of ORBmatcher::SearchByBoW (both overloads),
ORBmatcher::DescriptorDistance, Frame::ComputeStereoMatches, UndistortKeyPoints, ComputeImageBounds
and AssignFeaturesToGrid (shim/ORBmatcher_hip.cc, shim/Frame_hip.cc) linked over the reference's.  The same C driver (oracle/refslam_wrap.cc) then
builds real KeyFrame / Frame / MapPoint objects in both libraries; every output must be identical
to the all-reference build (liborbslam.so)."""
import ctypes

import numpy as np
import pytest

import oracle_lib
from test_matcher import _noisy_pair, _rand_desc

pytestmark = pytest.mark.skipif(oracle_lib.slam_lib() is None or not oracle_lib.SLAM_HIP_SO.exists(),
                                reason="oracle/_ref/liborbslam{,_hip}.so not built (needs /root/reference)")


def test_dropin_library_links_and_exports(orbx):
    orbx.load_library()
    L = oracle_lib.slam_hip_lib()
    for sym in ("orbslam_search_by_bow", "orbslam_stereo_frame", "orbslam_descriptor_distance", "orbx_shim_search_by_bow_calls",
                "orbx_shim_compute_stereo_matches_calls"):
        assert hasattr(L, sym)
    rng = np.random.default_rng(0)
    a, b = _rand_desc(rng, 1)[0], _rand_desc(rng, 1)[0]     # host helper of liborbx: runs without a GPU
    assert oracle_lib.ref_descriptor_distance(a, b, lib=L) == oracle_lib.ref_descriptor_distance(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_search_by_bow_dropin_equals_reference(orbx, mode):
    orbx.load_library()
    hip, ref = oracle_lib.slam_hip_lib(), oracle_lib.slam_lib()
    hip.orbx_shim_search_by_bow_calls.restype = ctypes.c_ulong
    before = hip.orbx_shim_search_by_bow_calls()
    rng = np.random.default_rng(300 + mode)
    ncalls = 0
    for trial, n in enumerate([40, 500, 1500, 2500]):
        kA, dA, kB, dB = _noisy_pair(rng, n, orbx)
        for groups in (False, True):
            gA = gB = vA = vB = None
            if groups:
                gA = rng.integers(0, 12, len(kA)).astype(np.int32) * 7
                gB = rng.integers(0, 14, len(kB)).astype(np.int32) * 7
                gA[rng.random(len(kA)) < 0.05] = -1
                gB[rng.random(len(kB)) < 0.05] = -1
                vA = (rng.random(len(kA)) < 0.8).astype(np.uint8)
                vB = (rng.random(len(kB)) < 0.9).astype(np.uint8) if mode == 1 else None
            want_n, want = oracle_lib.ref_search_by_bow(mode, kA, dA, kB, dB, 0.7, True, gA, gB, vA, vB, lib=ref)
            got_n, got = oracle_lib.ref_search_by_bow(mode, kA, dA, kB, dB, 0.7, True, gA, gB, vA, vB, lib=hip)
            ncalls += 1
            assert got_n == want_n and (got == want).all(), (trial, groups)
    assert hip.orbx_shim_search_by_bow_calls() - before == ncalls, "the HIP bodies were not the ones linked"


@pytest.mark.gpu
@pytest.mark.parametrize("early", [True, False])
@pytest.mark.parametrize("W,H,nf,seed,bf", [(1241, 376, 2000, 61, 386.1448), (752, 480, 1200, 62, 47.9064)])
def test_stereo_frame_dropin_equals_reference(orbx, monkeypatch, W, H, nf, seed, bf, early):
    """Frame::Frame(imLeft, imRight, ...): two shim extractors on the reference's two threads +
    the HIP ComputeStereoMatches, vs the all-reference constructor.  early: the match is launched by the extractor thread that
    finishes second and only collected by ComputeStereoMatches (shim/Frame_hip.cc); ORBX_SHIM_EARLY=0: launched by the member function."""
    orbx.load_library()
    hip, ref = oracle_lib.slam_hip_lib(), oracle_lib.slam_lib()
    hip.orbx_shim_compute_stereo_matches_calls.restype = ctypes.c_ulong
    hip.orbx_shim_early_stereo.restype = ctypes.c_ulong
    if not early:
        monkeypatch.setenv("ORBX_SHIM_EARLY", "0")
    before = hip.orbx_shim_compute_stereo_matches_calls()
    before_early = hip.orbx_shim_early_stereo()
    imL = orbx.synth_frame(seed, W, H)
    imR = orbx.synth_frame(seed, W, H, orbx.SYNTH_STEREO_RIGHT)
    want = oracle_lib.ref_stereo_frame(imL, imR, nf, 500.0, 500.0, W / 2, H / 2, bf, lib=ref)
    got = oracle_lib.ref_stereo_frame(imL, imR, nf, 500.0, 500.0, W / 2, H / 2, bf, lib=hip)
    assert hip.orbx_shim_compute_stereo_matches_calls() - before == 1
    assert hip.orbx_shim_early_stereo() - before_early == (1 if early else 0)
    for k in ("kpsL", "kpsR", "uRight", "depth"):
        assert got[k].shape == want[k].shape and (got[k].view(np.uint32) == want[k].view(np.uint32)).all(), k
    for k in ("descL", "descR"):
        assert (got[k] == want[k]).all(), k
    assert (want["uRight"] >= 0).sum() > 100


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,K,dist,seed", [
    (640, 480, (517.306408, 516.469215, 318.643040, 255.313989), (0.262383, -0.953104, -0.005358, 0.002628, 1.163314), 71),   # TUM1.yaml
    (752, 480, (458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05), 72),            # EuRoC.yaml
    (640, 480, (535.4, 539.2, 320.1, 247.6), (0.0, 0.0, 0.0, 0.0), 73)])                                                     # TUM3.yaml (rectified)
def test_mono_frame_dropin_equals_reference(orbx, W, H, K, dist, seed):
    """Frame::Frame(imGray, ...): shim extractor + the HIP UndistortKeyPoints / ComputeImageBounds /
    AssignFeaturesToGrid, vs the all-reference constructor: mvKeys, mvKeysUn, bounds, mGrid."""
    orbx.load_library()
    hip, ref = oracle_lib.slam_hip_lib(), oracle_lib.slam_lib()
    names = ("orbx_shim_undistort_calls", "orbx_shim_image_bounds_calls", "orbx_shim_assign_grid_calls")
    for n in names:
        getattr(hip, n).restype = ctypes.c_ulong
    before = [getattr(hip, n)() for n in names]
    im = orbx.synth_frame(seed, W, H)
    want = oracle_lib.ref_mono_frame(im, 1000, K[0], K[1], K[2], K[3], dist, lib=ref)
    got = oracle_lib.ref_mono_frame(im, 1000, K[0], K[1], K[2], K[3], dist, lib=hip)
    assert [getattr(hip, n)() - b for n, b in zip(names, before)] == [1, 1, 1], "the HIP bodies were not the ones linked"
    for k in ("kps", "kpsUn", "bounds", "gridInv"):
        assert got[k].shape == want[k].shape and (got[k].view(np.uint32) == want[k].view(np.uint32)).all(), k
    for k in ("desc", "gridOff", "gridIdx"):
        assert (got[k] == want[k]).all(), k
    assert want["gridOff"][-1] > 900


@pytest.mark.gpu
@pytest.mark.parametrize("early", [True, False])
@pytest.mark.parametrize("W,H,K,dist,seed", [
    (640, 480, (517.306408, 516.469215, 318.643040, 255.313989), (0.262383, -0.953104, -0.005358, 0.002628, 1.163314), 74),   # TUM1.yaml
    (1241, 376, (718.856, 718.856, 607.1928, 185.2157), (0.0, 0.0, 0.0, 0.0), 75)])                                          # KITTI00-02.yaml (rectified)
def test_later_mono_frames_dropin_equal_reference(orbx, monkeypatch, W, H, K, dist, seed, early):
    """Every frame but the first of a run (Frame::mbInitialComputations false: bounds and grid statics exist, src/Frame.cc:203-221).  In the
    drop-in library the left image's mvKeysUn and mGrid are then computed from inside the extractor call and filled by Frame::ExtractORB
    (shim/Frame_hip.cc); UndistortKeyPoints / AssignFeaturesToGrid are still called by the constructor and find them done.
    ORBX_SHIM_EARLY=0: the member functions do the work themselves.  Both against the all-reference constructor, frame by frame."""
    orbx.load_library()
    hip, ref = oracle_lib.slam_hip_lib(), oracle_lib.slam_lib()
    names = ("orbx_shim_undistort_calls", "orbx_shim_assign_grid_calls", "orbx_shim_early_fills")
    for n in names:
        getattr(hip, n).restype = ctypes.c_ulong
    if not early:
        monkeypatch.setenv("ORBX_SHIM_EARLY", "0")
    ims = [orbx.synth_frame(seed + 10 * i, W, H) for i in range(3)]
    try:
        for i, im in enumerate(ims):
            for lib in (ref, hip):
                lib.orbslam_keep_frame_statics(1 if i else 0)
            before = [getattr(hip, n)() for n in names]
            want = oracle_lib.ref_mono_frame(im, 1500, K[0], K[1], K[2], K[3], dist, lib=ref)
            got = oracle_lib.ref_mono_frame(im, 1500, K[0], K[1], K[2], K[3], dist, lib=hip)
            assert [getattr(hip, n)() - b for n, b in zip(names, before)] == [1, 1, 1 if (early and i) else 0], i
            for k in ("kps", "kpsUn", "bounds", "gridInv"):
                assert got[k].shape == want[k].shape and (got[k].view(np.uint32) == want[k].view(np.uint32)).all(), (i, k)
            for k in ("desc", "gridOff", "gridIdx"):
                assert (got[k] == want[k]).all(), (i, k)
            assert want["gridOff"][-1] > 900
    finally:
        for lib in (ref, hip):
            lib.orbslam_keep_frame_statics(0)


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1])
def test_compute_bow_dropin_equals_reference(orbx, tmp_path, which):
    """Frame::ComputeBoW / KeyFrame::ComputeBoW with the HIP tree descent vs DBoW2's own transform,
    both through the reference's ORBVocabulary loaded from the same text file."""
    from test_bow_transform import _descs
    orbx.load_library()
    hip, ref = oracle_lib.slam_hip_lib(), oracle_lib.slam_lib()
    hip.orbx_shim_compute_bow_calls.restype = ctypes.c_ulong
    voc = orbx.voc_synth.make_vocabulary(10, 5, 31)
    path = tmp_path / "voc.txt"
    orbx.voc_synth.write_text(voc, path)
    vr, vh = oracle_lib.RefVocabulary(path, lib=ref), oracle_lib.RefVocabulary(path, lib=hip)
    before = hip.orbx_shim_compute_bow_calls()
    for seed in (1, 2):
        d = _descs(orbx, voc, 2000, seed)
        want, got = vr.compute_bow(d, which), vh.compute_bow(d, which)
        assert (got["fv_node"] == want["fv_node"]).all() and (want["fv_node"] >= 0).sum() > 1000
        assert (got["bow_ids"] == want["bow_ids"]).all()
        assert (got["bow_vals"].view(np.uint64) == want["bow_vals"].view(np.uint64)).all()
    assert hip.orbx_shim_compute_bow_calls() - before == 2


@pytest.mark.gpu
def test_compute_bow_two_threads_share_one_vocabulary(orbx, tmp_path):
    """Tracking (Frame::ComputeBoW) and LocalMapping (KeyFrame::ComputeBoW) share the ORBVocabulary, hence the one device
    vocabulary handle, which is not re-entrant: two threads hammering it with different feature counts must each get the
    reference's result (shim/BoW_hip.cc serialises the orbx_bow_transform calls per vocabulary)."""
    import threading
    from test_bow_transform import _descs
    orbx.load_library()
    hip, ref = oracle_lib.slam_hip_lib(), oracle_lib.slam_lib()
    voc = orbx.voc_synth.make_vocabulary(10, 5, 33)
    path = tmp_path / "voc.txt"
    orbx.voc_synth.write_text(voc, path)
    vr, vh = oracle_lib.RefVocabulary(path, lib=ref), oracle_lib.RefVocabulary(path, lib=hip)
    sets = [_descs(orbx, voc, n, 40 + i) for i, n in enumerate((2000, 300, 1500, 37))]
    want = [[vr.compute_bow(d, which) for d in sets] for which in (0, 1)]
    errors = []

    def work(which):
        try:
            for rep in range(25):
                for i, d in enumerate(sets):
                    got = vh.compute_bow(d, which)
                    w = want[which][i]
                    if not ((got["fv_node"] == w["fv_node"]).all() and (got["bow_ids"] == w["bow_ids"]).all()
                            and (got["bow_vals"].view(np.uint64) == w["bow_vals"].view(np.uint64)).all()):
                        errors.append((which, rep, i))
        except Exception as e:                      # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=work, args=(w,)) for w in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:5]


@pytest.mark.gpu
def test_search_by_projection_dropin_equals_reference(orbx):
    """ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th) with the HIP body vs the
    reference body, both on a real Frame and real MapPoints."""
    from test_projection import make_case
    orbx.load_library()
    hip, ref = oracle_lib.slam_hip_lib(), oracle_lib.slam_lib()
    hip.orbx_shim_search_by_projection_calls.restype = ctypes.c_ulong
    before = hip.orbx_shim_search_by_projection_calls()
    for seed, crowded, th in ((11, False, 1.0), (12, True, 3.0), (13, False, 5.0)):
        fr, pts = make_case(seed, crowded=crowded)
        want_n, want = oracle_lib.ref_search_by_projection(fr, pts, th, 0.8, lib=ref)
        got_n, got = oracle_lib.ref_search_by_projection(fr, pts, th, 0.8, lib=hip)
        assert got_n == want_n and (got == want).all() and want_n > 200
    assert hip.orbx_shim_search_by_projection_calls() - before == 3


@pytest.mark.gpu
def test_search_by_projection_last_frame_dropin_equals_reference(orbx):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) with the HIP body vs the
    reference body on two real Frames whose MapPoints are real objects."""
    from test_projection import make_last_case
    orbx.load_library()
    hip, ref = oracle_lib.slam_hip_lib(), oracle_lib.slam_lib()
    hip.orbx_shim_search_by_projection_calls.restype = ctypes.c_ulong
    before = hip.orbx_shim_search_by_projection_calls()
    for seed, mz, crowded, th, mono in ((41, 0.0, False, 7.0, 0), (42, 0.5, True, 15.0, 0), (43, -0.5, False, 7.0, 0), (44, 0.5, False, 15.0, 1)):
        fr, last = make_last_case(seed, mz, crowded=crowded)
        want_n, want = oracle_lib.ref_search_by_projection_last(fr, last, th, mono, 1, lib=ref)
        got_n, got = oracle_lib.ref_search_by_projection_last(fr, last, th, mono, 1, lib=hip)
        assert got_n == want_n and (got == want).all() and want_n > 150
    assert hip.orbx_shim_search_by_projection_calls() - before == 4


def _octaves(inv_s2):
    return np.rint(np.log(1.0 / inv_s2.astype(np.float64)) / (2 * np.log(1.2))).astype(np.int32)


@pytest.mark.gpu
@pytest.mark.parametrize("overload,seed", [(1, 51), (1, 52), (2, 53)])
def test_fuse_dropin_equals_reference(orbx, overload, seed):
    """ORBmatcher::Fuse on a real target KeyFrame that already holds MapPoints (with more or fewer observations than the
    candidates: both Replace directions), a list with NULL entries and duplicates: identical pointer surgery."""
    from test_fuse import _scene, _sim3
    orbx.load_library()
    hip, ref = oracle_lib.slam_hip_lib(), oracle_lib.slam_lib()
    hip.orbx_shim_fuse_calls.restype = ctypes.c_ulong
    before = hip.orbx_shim_fuse_calls()
    kf, Tt, sk, Ts, P, cdesc, rng = _scene(orbx, seed)
    n, nc = len(kf["kps"]), len(P)
    n_exist = 120
    holder = np.full(n, -1, np.int32)
    holder[rng.choice(n, n_exist, replace=False)] = np.arange(n_exist)
    exist_obs = rng.integers(0, 5, n_exist).astype(np.int32)
    cand_obs = rng.integers(0, 4, nc).astype(np.int32)
    lst = np.concatenate([rng.permutation(nc), rng.integers(0, nc, 60)]).astype(np.int32)     # duplicates at the end
    lst[rng.choice(len(lst), 25, replace=False)] = -1 if overload == 1 else lst[0]            # NULL entries (overload 2 dereferences every entry)
    out = [oracle_lib.ref_fuse(overload, kf, Tt, _sim3(Tt), holder, exist_obs, sk, Ts, P, cdesc, cand_obs, lst, 3.0, False, lib=L) for L in (ref, hip)]
    assert hip.orbx_shim_fuse_calls() - before == 1, "the HIP body was not the one linked"
    want, got = out
    assert got["nfused"] == want["nfused"] and want["nfused"] > 200
    for k in ("holder", "bad", "replaced", "replace_point"):
        assert (got[k] == want[k]).all(), k
    if overload == 1:
        assert want["bad"].sum() > 5 and (want["holder"] >= 1000000).sum() < n_exist          # candidates and existing points were both replaced
    else:
        assert (want["replace_point"] >= 0).sum() > 5


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [61, 62])
def test_search_by_sim3_dropin_equals_reference(orbx, seed):
    """ORBmatcher::SearchBySim3: two KeyFrames that see the same structure through their own MapPoints, related by a
    Sim3; the mutual-consistency matches must be the reference's."""
    from test_fuse import SF, _pose
    orbx.load_library()
    hip, ref = oracle_lib.slam_hip_lib(), oracle_lib.slam_lib()
    hip.orbx_shim_search_by_sim3_calls.restype = ctypes.c_ulong
    before = hip.orbx_shim_search_by_sim3_calls()
    rng = np.random.default_rng(seed)
    n = 800
    T1, T2 = _pose(rng), _pose(rng)
    P = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(3, 9, n)], 1).astype(np.float32)
    def view(T, perm, noise):
        Pc = P @ T[:3, :3].T + T[:3, 3]
        uv = Pc[:, :2] / Pc[:, 2:3] * 500 + np.array([320, 240])
        k = np.zeros(n, orbx.KEYPOINT_DTYPE)
        k["x"], k["y"] = uv[:, 0] + rng.normal(0, noise, n), uv[:, 1] + rng.normal(0, noise, n)
        k["size"], k["class_id"], k["octave"] = 31, -1, np.clip(np.round(np.log(np.linalg.norm(Pc, axis=1) / 3.0) / np.log(1.2)), 0, 7)
        return k[perm]
    perm2 = rng.permutation(n)
    from test_matcher import _rand_desc
    base = _rand_desc(rng, n)
    d2 = base.copy()
    fl = rng.integers(0, 256, (n, 30))
    for i in range(n):
        for b in fl[i][: rng.integers(0, 30)]:
            d2[i, b >> 3] ^= 1 << (b & 7)
    kf1 = dict(kps=view(T1, np.arange(n), 0.5), desc=base)
    kf2 = dict(kps=view(T2, perm2, 0.5), desc=d2[perm2])
    pos2 = (P + rng.normal(0, 0.01, P.shape).astype(np.float32))[perm2]
    R12 = T1[:3, :3] @ T2[:3, :3].T
    t12 = T1[:3, 3] - R12 @ T2[:3, 3]
    pre = np.full(n, -1, np.int32)
    inv2 = np.argsort(perm2)
    pre[:40] = inv2[:40]                                           # matches SearchByBoW already found
    args = (kf1, P, T1, kf2, pos2, T2, pre, 1.01, R12, t12 * 1.01, 7.5)
    want_n, want = oracle_lib.ref_search_by_sim3(*args, lib=ref)
    got_n, got = oracle_lib.ref_search_by_sim3(*args, lib=hip)
    assert hip.orbx_shim_search_by_sim3_calls() - before == 1, "the HIP body was not the one linked"
    assert got_n == want_n and (got == want).all()
    assert want_n > 200 and (want[40:] < 0).sum() > 50


@pytest.mark.gpu
@pytest.mark.parametrize("overload,seed", [(3, 71), (4, 72), (5, 73)])
def test_remaining_search_by_projection_dropin_equals_reference(orbx, overload, seed):
    """SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (loop closing, 3) and SearchByProjection(CurrentFrame, pKF,
    sAlreadyFound, th, ORBdist) (relocalisation; 4 with, 5 without the rotation histogram) on real objects."""
    from test_area_search import _setup
    from test_fuse import _sim3
    orbx.load_library()
    hip, ref = oracle_lib.slam_hip_lib(), oracle_lib.slam_lib()
    hip.orbx_shim_search_by_projection_calls.restype = ctypes.c_ulong
    before = hip.orbx_shim_search_by_projection_calls()
    kf, Tt, sk, Ts, P, cdesc, holder, lst, rng = _setup(orbx, seed, 3 if overload == 3 else 4)
    if overload == 4:
        n, nc = len(kf["kps"]), len(P)
        kf["kps"]["angle"][:] = 10.0                                  # a dominant rotation bin plus outliers: the pruning removes matches
        sk["angle"] = np.where(rng.random(nc) < 0.8, 40.0, rng.uniform(0, 360, nc))
    out = [oracle_lib.ref_fuse(overload, kf, Tt, _sim3(Tt), holder, np.ones(150, np.int32), sk, Ts, P, cdesc, np.full(len(P), 1, np.int32), lst, 8.0, False, lib=L)
           for L in (ref, hip)]
    assert hip.orbx_shim_search_by_projection_calls() - before == 1, "the HIP body was not the one linked"
    want, got = out
    assert got["nfused"] == want["nfused"] and (got["holder"] == want["holder"]).all()
    assert want["nfused"] > 100


@pytest.mark.gpu
def test_search_for_triangulation_dropin_equals_reference(orbx):
    """ORBmatcher::SearchForTriangulation on two real KeyFrames (poses, mFeatVec, MapPoints, mvuRight): the shim
    computes the epipole with the reference's cv::Mat expressions and runs the matching on the device."""
    from test_triangulation import CASES, _scene
    orbx.load_library()
    hip, ref = oracle_lib.slam_hip_lib(), oracle_lib.slam_lib()
    hip.orbx_shim_search_for_triangulation_calls.restype = ctypes.c_ulong
    before = hip.orbx_shim_search_for_triangulation_calls()
    for seed, forward, stereo, only_stereo, ori in CASES:
        kf1, kf2, T1, T2, F12 = _scene(orbx, 40 + seed, forward=forward, stereo_frac=stereo)
        want_n, want, epi_w = oracle_lib.ref_search_for_triangulation(kf1, kf2, T1, T2, F12, only_stereo, ori, lib=ref)
        got_n, got, epi_g = oracle_lib.ref_search_for_triangulation(kf1, kf2, T1, T2, F12, only_stereo, ori, lib=hip)
        assert got_n == want_n and (got == want).all() and want_n > 30
    assert hip.orbx_shim_search_for_triangulation_calls() - before == len(CASES), "the HIP body was not the one linked"


@pytest.mark.gpu
@pytest.mark.parametrize("seed,stereo", [(5, 0.0), (6, 0.4)])
def test_local_bundle_adjustment_dropin_on_a_real_map(orbx, oracle, seed, stereo):
    """Optimizer::LocalBundleAdjustment (shim/Optimizer_hip.cc) on a real Map: KeyFrames, MapPoints,
    observations and the covisibility graph are the reference's objects; the result is compared with the
    CPU restatement run on the same window flattened independently here (which keyframes are local /
    fixed follows from the reference's covisibility rule, re-derived below)."""
    orbx.load_library()
    hip = oracle_lib.slam_hip_lib()
    hip.orbx_shim_lba_calls.restype = ctypes.c_ulong
    before = hip.orbx_shim_lba_calls()
    w = orbx.lba_synth.make_window(K=16, P=1200, seed=seed, max_obs=5, n_fixed=0, stereo_frac=stereo)
    K, P, E = w["K"], w["P"], w["E"]
    sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    octv = _octaves(w["edge_inv_sigma2"])
    obs6 = np.ascontiguousarray(np.stack([w["edge_point"], w["edge_kf"], w["edge_obs"][:, 0], w["edge_obs"][:, 1], w["edge_obs"][:, 2], octv], 1), np.float32)
    cam5 = np.ascontiguousarray(w["intr"][0], np.float32)
    ref_kf = K - 1
    poses_out, points_out = np.zeros((K, 16), np.float32), np.zeros((P, 3), np.float32)
    erased, role = np.zeros(E, np.uint8), np.zeros(K, np.uint8)
    _p = oracle_lib._p
    hip.orbslam_local_ba.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
    hip.orbslam_local_ba(K, _p(w["poses"]), _p(cam5), P, _p(w["points"]), E, _p(obs6), ref_kf, _p(sf), 8, 640, 480, _p(poses_out), _p(points_out), _p(erased), _p(role))
    assert hip.orbx_shim_lba_calls() - before == 1
    # ---- the reference's window rule, re-derived: neighbours = keyframes sharing >= 15 points with ref_kf (KeyFrame::UpdateConnections, th = 15)
    seen = np.zeros((K, P), bool)
    seen[w["edge_kf"], w["edge_point"]] = True
    shared = (seen & seen[ref_kf]).sum(1)
    shared[ref_kf] = 0
    neigh = shared >= 15 if (shared >= 15).any() else (shared == shared.max())
    local = neigh.copy(); local[ref_kf] = True
    local_pts = seen[local].any(0)
    fixed_kf = seen[:, local_pts].any(1) & ~local
    want_role = np.where(local, 1, np.where(fixed_kf, 2, 0))
    assert (role == want_role).all(), (role, want_role)
    # ---- flatten the same window for the restatement
    kf_list = list(np.flatnonzero(local)) + list(np.flatnonzero(fixed_kf))
    kf_new = {int(k): i for i, k in enumerate(kf_list)}
    pt_list = np.flatnonzero(local_pts)
    pt_new = -np.ones(P, np.int64); pt_new[pt_list] = np.arange(len(pt_list))
    sel = local_pts[w["edge_point"]]
    inv = (np.float32(1.0) / (sf[octv] * sf[octv])).astype(np.float32)       # Frame::mvInvLevelSigma2 as fill_frame builds it
    fx = np.array([1 if (k == 0 or fixed_kf[k]) else 0 for k in kf_list], np.uint8)
    prob = dict(K=len(kf_list), P=len(pt_list), E=int(sel.sum()), poses=np.ascontiguousarray(w["poses"][kf_list]), fixed=fx,
                intr=np.ascontiguousarray(w["intr"][kf_list]), points=np.ascontiguousarray(w["points"][pt_list]),
                edge_point=np.ascontiguousarray(pt_new[w["edge_point"][sel]].astype(np.int32)),
                edge_kf=np.array([kf_new[int(k)] for k in w["edge_kf"][sel]], np.int32),
                edge_obs=np.ascontiguousarray(w["edge_obs"][sel]), edge_inv_sigma2=np.ascontiguousarray(inv[sel]))
    want = oracle_lib.local_bundle_adjustment(oracle, prob)
    assert np.abs(poses_out[kf_list].astype(np.float64) - want["poses"]).max() <= 1e-5
    assert np.abs(points_out[pt_list].astype(np.float64) - want["points"]).max() <= 1e-5
    untouched = np.setdiff1d(np.arange(K), kf_list)
    assert (poses_out[untouched] == w["poses"][untouched]).all() and (points_out[~local_pts] == w["points"][~local_pts]).all()
    # every outlier observation is erased (:966-975).  MapPoint::EraseObservation additionally turns a point with <= 2 remaining
    # observations bad, which removes ALL its observations (src/MapPoint.cc:176-215): those extra erasures must belong to
    # points that lost an outlier observation.
    er, out = erased[sel].astype(bool), want["outlier"].astype(bool)
    th = np.where(prob["edge_obs"][:, 2] < 0, 5.991, 7.815)
    miss = out & ~er
    assert (np.abs(want["chi2"][miss] - th[miss]) < 1e-3).all()
    pts_with_outlier = np.unique(prob["edge_point"][out])
    extra = er & ~out
    assert np.isin(prob["edge_point"][extra], pts_with_outlier).all()
    assert erased[~sel].sum() == 0 and out.sum() > 10
    # the optimisation really moved towards the truth
    err0 = np.abs(w["poses"][kf_list].astype(np.float64) - w["true_poses"][kf_list]).max()
    err1 = np.abs(poses_out[kf_list].astype(np.float64) - w["true_poses"][kf_list]).max()
    assert err1 < 0.5 * err0


@pytest.mark.gpu
@pytest.mark.parametrize("seed,stereo,iters,robust,loop_kf", [(8, 0.0, 20, True, 0), (9, 0.4, 10, False, 7)])
def test_global_bundle_adjustment_dropin_on_a_real_map(orbx, oracle, seed, stereo, iters, robust, loop_kf):
    """Optimizer::GlobalBundleAdjustemnt / BundleAdjustment (shim/Optimizer_hip.cc) on a real Map, against the CPU restatement of the
    same graph; with nLoopKF != 0 the results go to mTcwGBA / mPosGBA and the live estimates stay untouched (src/Optimizer.cc:262-300)."""
    orbx.load_library()
    hip = oracle_lib.slam_hip_lib()
    hip.orbx_shim_bundle_adjustment_calls.restype = ctypes.c_ulong
    before = hip.orbx_shim_bundle_adjustment_calls()
    w = orbx.lba_synth.make_window(K=14, P=900, seed=seed, max_obs=6, n_fixed=0, stereo_frac=stereo)
    K, P, E = w["K"], w["P"], w["E"]
    sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    octv = _octaves(w["edge_inv_sigma2"])
    obs6 = np.ascontiguousarray(np.stack([w["edge_point"], w["edge_kf"], w["edge_obs"][:, 0], w["edge_obs"][:, 1], w["edge_obs"][:, 2], octv], 1), np.float32)
    cam5 = np.ascontiguousarray(w["intr"][0], np.float32)
    poses_out, points_out = np.zeros((K, 16), np.float32), np.zeros((P, 3), np.float32)
    untouched = ctypes.c_int(0)
    _p = oracle_lib._p
    ci, vp = ctypes.c_int, ctypes.c_void_p
    hip.orbslam_global_ba.argtypes = [ci, vp, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp]
    hip.orbslam_global_ba(K, _p(w["poses"]), _p(cam5), P, _p(w["points"]), E, _p(obs6), _p(sf), 8, 640, 480, iters, 1 if robust else 0, loop_kf, _p(poses_out),
                          _p(points_out), ctypes.byref(untouched))
    assert hip.orbx_shim_bundle_adjustment_calls() - before == 1 and untouched.value == 1
    # the same graph for the restatement: keyframe 0 fixed (mnId == 0), float32 information as fill_frame builds it
    w2 = dict(w)
    fixed = np.zeros(K, np.uint8); fixed[0] = 1
    w2["fixed"] = fixed
    w2["edge_inv_sigma2"] = (np.float32(1.0) / (sf[octv] * sf[octv])).astype(np.float32)
    want = oracle_lib.bundle_adjustment(oracle, w2, iters, robust)
    seen = np.zeros(P, bool); seen[w["edge_point"]] = True
    assert np.abs(poses_out.astype(np.float64) - want["poses"]).max() <= 1e-5
    assert np.abs(points_out[seen].astype(np.float64) - want["points"][seen]).max() <= 1e-5
    assert np.abs(want["poses"] - w["poses"]).max() > 1e-3                      # the optimisation did move things


@pytest.mark.gpu
def test_pose_optimization_dropin_on_a_real_frame(orbx, oracle):
    from test_pose_optimization import make_frame
    orbx.load_library()
    hip = oracle_lib.slam_hip_lib()
    hip.orbx_shim_pose_optimization_calls.restype = ctypes.c_ulong
    before = hip.orbx_shim_pose_optimization_calls()
    rng = np.random.default_rng(3)
    sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    _p = oracle_lib._p
    for seed in (31, 32):
        fr = make_frame(seed, n=700)
        n = len(fr["Xw"])
        octv = _octaves(fr["inv_sigma2"])
        fr["inv_sigma2"] = (np.float32(1.0) / (sf[octv] * sf[octv])).astype(np.float32)    # what the Frame holds
        kobs = np.ascontiguousarray(np.concatenate([fr["obs"], octv[:, None].astype(np.float32)], 1), np.float32)
        pose = np.ascontiguousarray(fr["pose"].reshape(16), np.float32)
        cam5 = np.ascontiguousarray(fr["cam"], np.float32)
        out, outl = np.zeros(16, np.float32), np.zeros(n, np.uint8)
        hip.orbslam_pose_optimization.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        ret = hip.orbslam_pose_optimization(_p(pose), _p(cam5), n, _p(np.ascontiguousarray(fr["Xw"], np.float32)), _p(kobs), _p(sf), 8, 640, 480, _p(out), _p(outl))
        want = oracle_lib.pose_optimization(oracle, fr)
        assert ret == want["inliers"] and (outl == want["outlier"]).all()
        assert np.abs(out.reshape(4, 4).astype(np.float64) - want["pose"]).max() <= 1e-5
    assert hip.orbx_shim_pose_optimization_calls() - before == 2
