"""Drop-in proof for the matcher / stereo surfaces.  oracle/_ref/liborbslam_hip.so is the
reference's own Frame.cc / KeyFrame.cc / MapPoint.cc / Map.cc / ORBmatcher.cc / DBoW2, built
UNMODIFIED, with exactly the change INTEGRATION.md describes: shim/ORBextractor.{h,cc} instead of
the reference's extractor and the HIP bodies of ORBmatcher::SearchByBoW (both overloads),
ORBmatcher::DescriptorDistance and Frame::ComputeStereoMatches (shim/ORBmatcher_hip.cc,
shim/Frame_hip.cc) linked over the reference's.  The same C driver (oracle/refslam_wrap.cc) then
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
@pytest.mark.parametrize("W,H,nf,seed,bf", [(1241, 376, 2000, 61, 386.1448), (752, 480, 1200, 62, 47.9064)])
def test_stereo_frame_dropin_equals_reference(orbx, W, H, nf, seed, bf):
    """Frame::Frame(imLeft, imRight, ...): two shim extractors on the reference's two threads +
    the HIP ComputeStereoMatches, vs the all-reference constructor."""
    orbx.load_library()
    hip, ref = oracle_lib.slam_hip_lib(), oracle_lib.slam_lib()
    hip.orbx_shim_compute_stereo_matches_calls.restype = ctypes.c_ulong
    before = hip.orbx_shim_compute_stereo_matches_calls()
    imL = orbx.synth_frame(seed, W, H)
    imR = orbx.synth_frame(seed, W, H, orbx.SYNTH_STEREO_RIGHT)
    want = oracle_lib.ref_stereo_frame(imL, imR, nf, 500.0, 500.0, W / 2, H / 2, bf, lib=ref)
    got = oracle_lib.ref_stereo_frame(imL, imR, nf, 500.0, 500.0, W / 2, H / 2, bf, lib=hip)
    assert hip.orbx_shim_compute_stereo_matches_calls() - before == 1
    for k in ("kpsL", "kpsR", "uRight", "depth"):
        assert got[k].shape == want[k].shape and (got[k].view(np.uint32) == want[k].view(np.uint32)).all(), k
    for k in ("descL", "descR"):
        assert (got[k] == want[k]).all(), k
    assert (want["uRight"] >= 0).sum() > 100


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
