"""The restated matchers (oracle/match_oracle.cc) against the UNMODIFIED reference sources
(src/ORBmatcher.cc, Frame.cc, KeyFrame.cc, MapPoint.cc, Map.cc + vendored DBoW2) compiled on
oracle/cvshim into oracle/_ref/liborbslam.so and driven through real KeyFrame / Frame /
MapPoint objects.  This is what pins the matcher oracle to the reference's own code."""
import numpy as np
import pytest

import oracle_lib
from test_matcher import _kps, _noisy_pair, _rand_desc

pytestmark = pytest.mark.skipif(oracle_lib.slam_lib() is None, reason="oracle/_ref/liborbslam.so not built (needs /root/reference)")


def test_descriptor_distance_equals_reference(oracle):
    rng = np.random.default_rng(2)
    for _ in range(200):
        a, b = _rand_desc(rng, 1)[0], _rand_desc(rng, 1)[0]
        want = oracle_lib.ref_descriptor_distance(a, b)
        assert want == int(np.unpackbits(a ^ b).sum())
        assert oracle_lib.descriptor_distance(oracle, a, b) == want


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("groups", [False, True])
@pytest.mark.parametrize("ratio,ori", [(0.7, True), (0.6, False), (0.9, True)])
def test_search_by_bow_restatement_equals_reference(orbx, oracle, mode, groups, ratio, ori):
    rng = np.random.default_rng(100 + mode + 2 * groups + int(ratio * 10))
    for trial in range(6):
        n = [50, 300, 1000, 2000, 1, 700][trial]
        kA, dA, kB, dB = _noisy_pair(rng, n, orbx)
        gA = gB = vA = vB = None
        if groups:
            gA = rng.integers(0, 12, len(kA)).astype(np.int32) * 7
            gB = rng.integers(0, 14, len(kB)).astype(np.int32) * 7
            gA[rng.random(len(kA)) < 0.05] = -1          # features the vocabulary did not place
            gB[rng.random(len(kB)) < 0.05] = -1
            vA = (rng.random(len(kA)) < 0.8).astype(np.uint8)
            vB = (rng.random(len(kB)) < 0.9).astype(np.uint8)
        gA2, gB2 = gA, gB     # negative node id = not filed in the FeatureVector (both in the C ABI and in the restatement)
        want_n, want = oracle_lib.ref_search_by_bow(mode, kA, dA, kB, dB, ratio, ori, gA, gB, vA, vB if mode == 1 else None)
        got_n, got = oracle_lib.search_by_bow(oracle, mode, kA, dA, kB, dB, ratio, ori, gA2, gB2, vA, vB if mode == 1 else None)
        assert got_n == want_n, (trial, got_n, want_n)
        assert (got == want).all(), trial


def test_search_by_bow_ties_equal_reference(orbx, oracle):
    rng = np.random.default_rng(5)
    base = _rand_desc(rng, 8)
    dA = np.repeat(base, 40, axis=0)
    dB = np.repeat(base, 50, axis=0)[rng.permutation(400)]
    kA, kB = _kps(rng, len(dA), orbx), _kps(rng, len(dB), orbx)
    for mode in (0, 1):
        for ratio, ori in ((0.95, False), (0.95, True)):
            want_n, want = oracle_lib.ref_search_by_bow(mode, kA, dA, kB, dB, ratio, ori)
            got_n, got = oracle_lib.search_by_bow(oracle, mode, kA, dA, kB, dB, ratio, ori)
            assert got_n == want_n and (got == want).all()


STEREO_CASES = [(1241, 376, 2000, 31, 718.856, 718.856, 607.1928, 185.2157, 386.1448),    # KITTI00-02.yaml
                (752, 480, 1200, 33, 435.2047, 435.2047, 367.4517, 252.2008, 47.9064),    # EuRoC.yaml
                (640, 480, 1000, 35, 517.3, 516.5, 318.6, 255.3, 40.0)]


@pytest.mark.parametrize("W,H,nf,seed,fx,fy,cx,cy,bf", STEREO_CASES)
def test_stereo_frame_restatement_equals_reference(orbx, oracle, W, H, nf, seed, fx, fy, cx, cy, bf):
    """Frame::Frame(imLeft, imRight, ...) of the reference (two extractor threads +
    ComputeStereoMatches) vs restated extractor + restated ComputeStereoMatches."""
    imL = orbx.synth_frame(seed, W, H)
    imR = orbx.synth_frame(seed, W, H, orbx.SYNTH_STEREO_RIGHT)
    ref = oracle_lib.ref_stereo_frame(imL, imR, nf, fx, fy, cx, cy, bf)
    rst = oracle.restatement(nf)
    kL, dL = rst.extract(imL)
    kR, dR = rst.extract(imR)
    assert len(kL) == len(ref["kpsL"]) and (kL.view(np.uint32) == ref["kpsL"].view(np.uint32)).all() and (dL == ref["descL"]).all()
    assert len(kR) == len(ref["kpsR"]) and (kR.view(np.uint32) == ref["kpsR"].view(np.uint32)).all() and (dR == ref["descR"]).all()
    pyrL, pyrR = oracle.pyramid(rst, imL), oracle.pyramid(rst, imR)
    t, _, _ = rst.tables()
    uR, dep, sad = oracle_lib.compute_stereo_matches(oracle, kL, dL, kR, dR, pyrL, pyrR, t[0], t[1], bf, 0.0)
    assert (uR.view(np.uint32) == ref["uRight"].view(np.uint32)).all()
    assert (dep.view(np.uint32) == ref["depth"].view(np.uint32)).all()
    assert (uR >= 0).sum() > 100                          # the synthetic pair really matches
    # the array-driven form of the reference (default-constructed Frame) agrees with its constructor form
    uR2, dep2 = oracle_lib.ref_compute_stereo_matches(kL, dL, kR, dR, pyrL, pyrR, t[0], bf)
    assert (uR2.view(np.uint32) == uR.view(np.uint32)).all() and (dep2.view(np.uint32) == dep.view(np.uint32)).all()
