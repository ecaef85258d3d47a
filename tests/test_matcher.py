"""Hamming matchers: known-answer tests on CPU (oracle + host helper), index-exact parity of
the HIP path against the oracle on the GPU."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib


def _rand_desc(rng, n):
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def test_descriptor_distance_kat(orbx, oracle):
    z = np.zeros(32, np.uint8)
    f = np.full(32, 255, np.uint8)
    one = z.copy(); one[5] = 0x10
    for fn in (lambda a, b: oracle_lib.descriptor_distance(oracle, a, b), orbx.DescriptorDistance):
        assert fn(z, z) == 0
        assert fn(z, f) == 256
        assert fn(z, one) == 1
        assert fn(f, one) == 255
    rng = np.random.default_rng(1)
    for _ in range(50):
        a, b = _rand_desc(rng, 1)[0], _rand_desc(rng, 1)[0]
        want = int(np.unpackbits(a ^ b).sum())
        assert oracle_lib.descriptor_distance(oracle, a, b) == want
        assert orbx.DescriptorDistance(a, b) == want


def _kps(rng, n, orbx):
    k = np.zeros(n, orbx.KEYPOINT_DTYPE)
    k["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    k["x"] = rng.uniform(20, 600, n).astype(np.float32)
    k["y"] = rng.uniform(20, 440, n).astype(np.float32)
    k["octave"] = rng.integers(0, 8, n)
    return k


def test_search_by_bow_oracle_kat(orbx, oracle):
    """Hand-checkable: identical descriptors match 1:1, rotation-inconsistent ones are pruned."""
    rng = np.random.default_rng(3)
    n = 40
    dA = _rand_desc(rng, n)
    kA, kB = _kps(rng, n, orbx), _kps(rng, n, orbx)
    kB["angle"] = kA["angle"]          # rot = 0 for every true match -> one dominant bin
    perm = rng.permutation(n)
    dB = dA[perm]
    kB = kB[perm]
    nm, m = oracle_lib.search_by_bow(oracle, 0, kA, dA, kB, dB, 0.7, True)
    assert nm == n and (m == perm).all()      # m[j] = index of the A feature matched to B feature j
    nm1, m1 = oracle_lib.search_by_bow(oracle, 1, kA, dA, kB, dB, 0.7, True)
    assert nm1 == n and (perm[m1] == np.arange(n)).all()
    # one match with an inconsistent rotation is removed by ComputeThreeMaxima pruning
    kB2 = kB.copy()
    kB2["angle"][0] = (kB2["angle"][0] + 170) % 360
    nm2, m2 = oracle_lib.search_by_bow(oracle, 0, kA, dA, kB2, dB, 0.7, True)
    assert nm2 == n - 1 and m2[0] == -1


def _noisy_pair(rng, n, orbx, flip_bits=18, extra=200):
    dA = _rand_desc(rng, n)
    kA = _kps(rng, n, orbx)
    dB = dA.copy()
    for i in range(n):                        # flip a few bits: realistic distances 0..40
        bits = rng.integers(0, 256, size=rng.integers(0, flip_bits))
        for b in bits:
            dB[i, b >> 3] ^= np.uint8(1 << (b & 7))
    kB = kA.copy()
    kB["angle"] = (kA["angle"] + rng.normal(0, 4, n).astype(np.float32)) % 360
    # duplicates + distractors so that second-best / ties / greedy skipping all occur
    dB = np.concatenate([dB, dB[: n // 5], _rand_desc(rng, extra)])
    kB = np.concatenate([kB, kB[: n // 5], _kps(rng, extra, orbx)])
    p = rng.permutation(len(dB))
    return kA, dA, kB[p], dB[p]


@pytest.mark.gpu
@pytest.mark.parametrize("mfma", [False, True])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("groups", [False, True])
def test_search_by_bow_hip_index_exact(orbx, oracle, monkeypatch, mode, groups, mfma):
    """mfma: the unfiltered (brute-force) candidate lists from k_bow_topk_mfma (ORBX_MATCH_MFMA=1, a measured alternative, not the
    default; with groups the switch changes nothing: the filtered kernel is the popcount one)."""
    if mfma:
        monkeypatch.setenv("ORBX_MATCH_MFMA", "1")
    rng = np.random.default_rng(10 + mode + 2 * groups)
    mt = orbx.ORBmatcher(0.7, True, max_features=2600)
    for trial in range(6):
        n = [50, 300, 1000, 2000, 1, 700][trial]
        kA, dA, kB, dB = _noisy_pair(rng, n, orbx)
        gA = gB = vA = vB = None
        if groups:
            gA = rng.integers(0, 12, len(kA)).astype(np.int32) * 7 - 3
            gB = rng.integers(0, 14, len(kB)).astype(np.int32) * 7 - 3
            vA = (rng.random(len(kA)) < 0.8).astype(np.uint8)
            vB = (rng.random(len(kB)) < 0.9).astype(np.uint8)
        want_n, want = oracle_lib.search_by_bow(oracle, mode, kA, dA, kB, dB, 0.7, True, gA, gB, vA, vB)
        got_n, got = mt.SearchByBoW(kA, dA, kB, dB, gA, gB, vA, vB, mode=mode)
        assert got_n == want_n, (trial, got_n, want_n)
        assert (got == want).all(), trial
    mt.close()


@pytest.mark.gpu
def test_search_by_bow_degenerate_ties(orbx, oracle):
    """Many identical descriptors: every tie must resolve to the first index and the top-4
    short lists overflow, forcing the exact rescan path."""
    rng = np.random.default_rng(5)
    base = _rand_desc(rng, 8)
    dA = np.repeat(base, 40, axis=0)             # 320 features, 8 distinct descriptors
    dB = np.repeat(base, 50, axis=0)[rng.permutation(400)]
    kA, kB = _kps(rng, len(dA), orbx), _kps(rng, len(dB), orbx)
    mt = orbx.ORBmatcher(0.95, False, max_features=512)
    for mfma in ("0", "1"):          # (both candidate-list kernels: popcount and matrix cores)
        os.environ["ORBX_MATCH_MFMA"] = mfma
        try:
            for mode in (0, 1):
                want_n, want = oracle_lib.search_by_bow(oracle, mode, kA, dA, kB, dB, 0.95, False)
                got_n, got = mt.SearchByBoW(kA, dA, kB, dB, mode=mode)
                assert got_n == want_n and (got == want).all(), (mfma, mode)
        finally:
            del os.environ["ORBX_MATCH_MFMA"]
    mt.close()


@pytest.mark.gpu
def test_stereo_hamming_and_frame_match_on_real_features(orbx, oracle):
    """KITTI-shaped stereo pair through the HIP extractor, then (a) the ComputeStereoMatches
    Hamming stage and (b) brute-force SearchByBoW, device-resident, vs the oracle."""
    W, H, nf = 1241, 376, 2000
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=4)
    frames = [orbx.synth_frame(31, W, H), orbx.synth_frame(31, W, H, orbx.SYNTH_STEREO_RIGHT),
              orbx.synth_frame(32, W, H), orbx.synth_frame(32, W, H, orbx.SYNTH_STEREO_RIGHT)]
    dev = ext.upload(frames)
    ext.run_device(*dev)
    fs = orbx.ORBmatcher.features_of(ext, 4)
    mt = orbx.ORBmatcher(0.7, True, max_features=ext.capacity, max_pairs=2)
    sf = ext.GetScaleFactors()
    mt.stereo_match_device(fs, fs, [0, 2], [1, 3], sf, float("inf"), after=ext)
    bi, bd, nm = mt.download(2)
    kps, desc, counts = ext.download(4)
    for p, (l, r) in enumerate([(0, 1), (2, 3)]):
        nl, nr = counts[l], counts[r]
        wd, wi = oracle_lib.stereo_hamming(oracle, kps[l, :nl], desc[l, :nl], kps[r, :nr], desc[r, :nr], sf, H, float("inf"))
        assert (bd[p, :nl] == wd).all() and (bi[p, :nl] == wi).all()
        assert nm[p] == int((wd < 75).sum()) and nm[p] > 50      # the synthetic pair really matches
    # finite disparity gate (bf/b of KITTI00-02.yaml: 386.1448/0.5371...) through the host form
    nl, nr = counts[0], counts[1]
    wd, wi = oracle_lib.stereo_hamming(oracle, kps[0, :nl], desc[0, :nl], kps[1, :nr], desc[1, :nr], sf, H, 40.0)
    gd, gi = mt.StereoHamming(kps[0, :nl], desc[0, :nl], kps[1, :nr], desc[1, :nr], sf, 40.0)
    assert (gd == wd).all() and (gi == wi).all()
    # frame-to-frame brute force (one vocabulary node) on the same device-resident features
    mt.search_by_bow_device(fs, fs, [0, 2], [1, 3], mode=0, after=ext)
    m, d, nm = mt.download(2)
    for p, (a, b) in enumerate([(0, 1), (2, 3)]):
        na, nb = counts[a], counts[b]
        wn, wm = oracle_lib.search_by_bow(oracle, 0, kps[a, :na], desc[a, :na], kps[b, :nb], desc[b, :nb], 0.7, True)
        assert nm[p] == wn and (m[p, :nb] == wm).all()
    mt.close()
    ext.close()


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,nf,bf,two_handles", [(1241, 376, 2000, 386.1448, False), (752, 480, 1200, 47.9064, True), (640, 480, 1000, 40.0, False)])
def test_compute_stereo_matches_hip_bit_exact(orbx, oracle, W, H, nf, bf, two_handles):
    """Complete Frame::ComputeStereoMatches on the device (Hamming + SAD + sub-pixel + median cut)
    vs the restatement and, when oracle/_ref/liborbslam.so travelled with the snapshot, vs the
    reference's own stereo Frame constructor.  mvuRight / mvDepth compared as bit patterns."""
    seeds = [31, 32, 47]
    lefts = [orbx.synth_frame(s, W, H) for s in seeds]
    rights = [orbx.synth_frame(s, W, H, orbx.SYNTH_STEREO_RIGHT) for s in seeds]
    n = len(seeds)
    if two_handles:      # the reference's model: one extractor per eye (src/Tracking.cc:179-187)
        eL = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=n)
        eR = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=n)
        dl, dr = eL.upload(lefts), eR.upload(rights)
        eL.run_device(*dl)
        eR.run_device(*dr)
        fl, fr = list(range(n)), list(range(n))
    else:                # both eyes in one batch: frames 0..n-1 left, n..2n-1 right
        eL = eR = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2 * n)
        d = eL.upload(lefts + rights)
        eL.run_device(*d)
        fl, fr = list(range(n)), list(range(n, 2 * n))
    mt = orbx.ORBmatcher(0.7, True, max_features=eL.capacity, max_pairs=n)
    mt.compute_stereo_matches_device(eL, eR, fl, fr, bf, 0.0)
    uR, dep = mt.download_stereo(n)
    bi, bd, nm = mt.download(n)
    kL, dL, cL = eL.download(eL_b := (n if two_handles else 2 * n))
    kR, dR, cR = (eR.download(n) if two_handles else (kL, dL, cL))
    rst = oracle.restatement(nf)
    t, _, _ = rst.tables()
    slam = oracle_lib.slam_lib()
    for p in range(n):
        a, b = fl[p], fr[p]
        na, nb = int(cL[a]), int(cR[b])
        pyrL, pyrR = oracle.pyramid(rst, lefts[p]), oracle.pyramid(rst, rights[p])
        wu, wz, _ = oracle_lib.compute_stereo_matches(oracle, kL[a, :na], dL[a, :na], kR[b, :nb], dR[b, :nb], pyrL, pyrR, t[0], t[1], bf, 0.0)
        assert (uR[p, :na].view(np.uint32) == wu.view(np.uint32)).all(), p
        assert (dep[p, :na].view(np.uint32) == wz.view(np.uint32)).all(), p
        assert nm[p] == int((wu >= 0).sum()) and nm[p] > 100
        if slam is not None:
            ref = oracle_lib.ref_stereo_frame(lefts[p], rights[p], nf, 500.0, 500.0, W / 2, H / 2, bf)
            assert len(ref["uRight"]) == na
            assert (uR[p, :na].view(np.uint32) == ref["uRight"].view(np.uint32)).all()
            assert (dep[p, :na].view(np.uint32) == ref["depth"].view(np.uint32)).all()
    mt.close()
    eL.close()
    if two_handles:
        eR.close()


@pytest.mark.gpu
def test_stereo_frame_of_two_single_frame_calls(orbx, oracle):
    """The stereo Frame constructor's shape (src/Frame.cc:159-168): two one-frame extractors on two threads - their calls announce each other
    and run as one launch set -, then orbx_stereo_frame (the latency form of ComputeStereoMatches: no events, pinned results).  mvuRight /
    mvDepth as bit patterns against the restatement, for several frames in a row on the same handles (double-buffered member arenas, the
    cached scale tables and the pinned result buffer all carry over from frame to frame)."""
    import threading
    W, H, nf, bf = 1241, 376, 2000, 386.1448
    eL = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H)
    eR = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H)
    mt = orbx.ORBmatcher(0.7, True, max_features=eL.capacity, max_pairs=1)
    rst = oracle.restatement(nf)
    t, _, _ = rst.tables()
    for seed in (31, 32, 47, 31):
        left, right = orbx.synth_frame(seed, W, H), orbx.synth_frame(seed, W, H, orbx.SYNTH_STEREO_RIGHT)
        out = {}

        def run(name, ext, other, im):
            ext.expect_partner(other)
            out[name] = ext(im)
        th = [threading.Thread(target=run, args=("l", eL, eR, left)), threading.Thread(target=run, args=("r", eR, eL, right))]
        for x in th:
            x.start()
        for x in th:
            x.join()
        (kL, dL), (kR, dR) = out["l"], out["r"]
        u, z = mt.stereo_frame(eL, eR, bf, 0.0, n=len(kL))
        pyrL, pyrR = oracle.pyramid(rst, left), oracle.pyramid(rst, right)
        wu, wz, _ = oracle_lib.compute_stereo_matches(oracle, kL, dL, kR, dR, pyrL, pyrR, t[0], t[1], bf, 0.0)
        assert (u.view(np.uint32) == wu.view(np.uint32)).all() and (z.view(np.uint32) == wz.view(np.uint32)).all(), seed
        assert int((wu >= 0).sum()) > 100
        # the general entry point on the same two handles gives the same answer (and leaves the fast path's cached tables alone)
        mt.compute_stereo_matches_device(eL, eR, [0], [0], bf, 0.0)
        u2, z2 = mt.download_stereo(1)
        assert (u2[0, :len(kL)].view(np.uint32) == u.view(np.uint32)).all() and (z2[0, :len(kL)].view(np.uint32) == z.view(np.uint32)).all()
        # the two halves apart (what the drop-in constructor does: begin on an extractor thread, end in ComputeStereoMatches), and with the
        # three kernels of the batched path instead of the two of the latency form: the same bits
        L = mt._L
        L.orbx_stereo_frame_begin.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
        L.orbx_stereo_frame_end.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        for one in ("1", "0"):
            os.environ["ORBX_STEREO_ONE"] = one
            try:
                assert L.orbx_stereo_frame_begin(mt._h, eL._h, eR._h, bf, 0.0) == 0
                u3, z3 = np.full(len(kL), -7.0, np.float32), np.full(len(kL), -7.0, np.float32)
                assert L.orbx_stereo_frame_end(mt._h, u3.ctypes.data, z3.ctypes.data, len(kL)) == 0
            finally:
                del os.environ["ORBX_STEREO_ONE"]
            assert (u3.view(np.uint32) == u.view(np.uint32)).all() and (z3.view(np.uint32) == z.view(np.uint32)).all(), one
        assert L.orbx_stereo_frame_end(mt._h, u3.ctypes.data, z3.ctypes.data, len(kL)) != 0      # nothing begun: a state error, not stale results
    mt.close(); eL.close(); eR.close()
