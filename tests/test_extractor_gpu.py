"""GPU parity of the HIP extractor against the CPU oracle, stage by stage and end to end.

Bit-exact everywhere: pyramid bytes, FAST score bytes, candidate lists (order included),
quadtree selection and order, angles as float bit patterns, blurred bytes, descriptors,
final cv::KeyPoint fields.
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONFIGS = [
    # (W, H, nfeatures) - TUM1.yaml, KITTI00-02.yaml, EuRoC.yaml of the reference
    (640, 480, 1000),
    (1241, 376, 2000),
    (752, 480, 1200),
]


def kp_matrix(k):
    return np.stack([k["x"], k["y"], k["size"], k["angle"], k["response"], k["octave"].astype(np.float32),
                     k["class_id"].astype(np.float32)], 1)


@pytest.mark.parametrize("W,H,nf", CONFIGS)
def test_stages_bit_exact(orbx, oracle, W, H, nf):
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2)
    ext.set_debug_taps(True)     # keep the FAST score map (the fused detector does not write it otherwise)
    rst = oracle.restatement(nf)
    t, q, u = rst.tables()
    assert (ext.GetScaleFactors().view(np.uint32) == t[0].view(np.uint32)).all()
    assert (ext.GetInverseScaleFactors().view(np.uint32) == t[1].view(np.uint32)).all()
    assert (ext.GetScaleSigmaSquares().view(np.uint32) == t[2].view(np.uint32)).all()
    assert (ext.GetInverseScaleSigmaSquares().view(np.uint32) == t[3].view(np.uint32)).all()
    assert (ext.features_per_level() == q).all()
    frames = [orbx.synth_frame(11, W, H), orbx.synth_frame(12, W, H, orbx.SYNTH_LOW_TEXTURE)]
    kps, desc, counts = ext.extract_batch(frames)
    for f, im in enumerate(frames):
        pyr = oracle.pyramid(rst, im)
        for l in range(8):
            lv = ext.mvImagePyramid(l, frame=f)
            assert lv.shape == pyr[l].shape
            assert (lv == pyr[l]).all(), "pyramid level %d" % l
            S = oracle.score_map(pyr[l], 7)
            Sd = ext.debug_scores(l, frame=f)
            h, w = S.shape
            assert (Sd[19:h - 19, 19:w - 19] == S[19:h - 19, 19:w - 19]).all(), "score map level %d" % l
            cand = oracle.cell_candidates(S, 20)
            cd, n = ext.debug_candidates(l, frame=f)
            assert n == len(cand) and (cd == cand).all(), "candidates level %d" % l
            sel = oracle.octree(cand, w, h, int(q[l]))
            lk = ext.debug_level_keypoints(l, frame=f)
            assert len(lk) == len(sel), "octree count level %d: %d vs %d" % (l, len(lk), len(sel))
            packed = ((lk["x"].astype(np.uint32) - 16) | ((lk["y"].astype(np.uint32) - 16) << 12) |
                      (lk["response"].astype(np.uint32) << 24))
            assert (packed == sel).all(), "octree selection/order level %d" % l
            ang = np.array([oracle.ic_angle(rst, pyr[l], x, y) for x, y in zip(lk["x"], lk["y"])], np.float32)
            assert (ang.view(np.uint32) == lk["angle"].view(np.uint32)).all(), "angles level %d" % l
            if len(lk):
                B = oracle.blur(pyr[l])
                Bd = ext.mvImagePyramid(l, frame=f, blurred=True)
                assert (B == Bd).all(), "blur level %d" % l
        ko, do = rst.extract(im)
        n = int(counts[f])
        assert n == len(ko)
        assert (kp_matrix(kps[f, :n]).view(np.uint32) == ko.view(np.uint32)).all(), "final keypoints"
        assert (desc[f, :n] == do).all(), "descriptors"
    ext.close()


def test_vs_compiled_reference(orbx, oracle):
    """End to end against oracle/_ref (the unmodified reference source), when it was built."""
    ref = oracle.reference(1000)
    if ref is None:
        pytest.skip("oracle/_ref not built")
    ext = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=8)
    frames = [orbx.synth_frame(100 + i, 640, 480, orbx.SYNTH_LOW_TEXTURE if i % 4 == 3 else 0) for i in range(8)]
    kps, desc, counts = ext.extract_batch(frames)
    for f, im in enumerate(frames):
        ko, do = ref.extract(im)
        n = int(counts[f])
        assert n == len(ko)
        assert (kp_matrix(kps[f, :n]).view(np.uint32) == ko.view(np.uint32)).all()
        assert (desc[f, :n] == do).all()
    ext.close()


def test_single_image_and_empty(orbx, oracle):
    ext = orbx.ORBextractor(500, 1.2, 8, 20, 7, max_width=640, max_height=480)
    k, d = ext(None)
    assert len(k) == 0 and d.shape == (0, 32)
    im = orbx.synth_frame(5, 640, 480)
    k, d = ext(im)
    ko, do = oracle.restatement(500).extract(im)
    assert len(k) == len(ko) and (d == do).all()
    # a flat image yields no keypoints (reference: descriptors.release())
    k, d = ext(np.full((480, 640), 90, np.uint8))
    assert len(k) == 0
    ext.close()


# Ragged / extreme geometries: widths and heights that are not multiples of anything, the smallest image the
# 8-level pyramid accepts, a large frame, other pyramid settings, tight (unpadded) caller strides.  End to end
# (keypoints as bit patterns + descriptors) against the CPU restatement, which is itself pinned to the compiled reference.
ODD = [
    # W, H, nfeatures, scale, levels, iniTh, minTh
    (641, 479, 1000, 1.2, 8, 20, 7),
    (333, 251, 500, 1.2, 8, 20, 7),       # smallest level is 93x70: single-column cell grids, wide cells
    (227, 227, 300, 1.2, 8, 20, 7),       # near the minimum the 8-level pyramid accepts
    (1920, 1080, 2000, 1.2, 8, 20, 7),
    (1023, 769, 1500, 1.2, 8, 12, 5),
    (640, 480, 2000, 1.2, 8, 20, 7),      # the 2*nFeatures monocular-initialisation extractor (src/Tracking.cc:192)
    (640, 480, 1000, 1.5, 4, 20, 7),
    (500, 375, 800, 1.1, 12, 20, 7),
    (1920, 1080, 5000, 1.2, 8, 20, 7),    # level quotas above 1024: the 2048-node quadtree
    (640, 480, 30, 1.2, 8, 20, 7),        # quotas of 2 .. 7 keypoints per level
    (640, 480, 800, 1.2, 1, 20, 7),       # a single level
]


@pytest.mark.parametrize("W,H,nf,sf,nl,ini,mn", ODD)
def test_ragged_geometries_bit_exact(orbx, oracle, W, H, nf, sf, nl, ini, mn):
    ext = orbx.ORBextractor(nf, sf, nl, ini, mn, max_width=W, max_height=H, max_batch=3)
    rst = oracle.restatement(nf, sf, nl, ini, mn)
    frames = [orbx.synth_frame(100 + W + i, W, H, orbx.SYNTH_LOW_TEXTURE if i == 2 else 0) for i in range(3)]
    kps, desc, counts = ext.extract_batch(frames)
    for f, im in enumerate(frames):
        ko, do = rst.extract(im)
        n = int(counts[f])
        assert n == len(ko), (f, n, len(ko))
        assert (kp_matrix(kps[f, :n]).view(np.uint32) == ko.view(np.uint32)).all(), f
        assert (desc[f, :n] == do).all(), f
    ext.close()
    # the same geometry through the single-frame call of a one-frame handle (one hipGraph launch per frame), twice: both result buffers
    one = orbx.ORBextractor(nf, sf, nl, ini, mn, max_width=W, max_height=H)
    for f in (0, 1):
        k, d = one(frames[f])
        ko, do = rst.extract(frames[f])
        assert len(k) == len(ko) and (kp_matrix(k).view(np.uint32) == ko.view(np.uint32)).all() and (d == do).all(), f
        # ... and every byte of its pyramid (one k_pyramid_tiles launch: all levels of the frame tile by tile inside LDS)
        pyr = oracle.pyramid(rst, frames[f])
        for l in range(nl):
            assert (one.mvImagePyramid(l) == pyr[l]).all(), "single-frame pyramid, level %d" % l
    one.close()


def test_tight_device_stride(orbx, oracle):
    """orbx_extract_batch_device on the caller's own device buffer with stride == width (no row padding):
    the pyramid kernel must take its safe path and still be bit-exact."""
    import torch
    W, H, nf, B = 641, 480, 1000, 2
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    frames = [orbx.synth_frame(300 + i, W, H) for i in range(B)]
    buf = torch.from_numpy(np.stack(frames)).cuda()              # tight: stride W, frame pitch W*H
    ext.run_device(ctypes.c_void_p(buf.data_ptr()), W, W * H, (B, W, H))
    kps, desc, counts = ext.download(B)
    rst = oracle.restatement(nf)
    for f, im in enumerate(frames):
        ko, do = rst.extract(im)
        n = int(counts[f])
        assert n == len(ko) and (kp_matrix(kps[f, :n]).view(np.uint32) == ko.view(np.uint32)).all() and (desc[f, :n] == do).all()
    ext.close()


def test_single_frame_pyramid_equals_batch_pyramid_on_random_geometries(orbx):
    """k_pyramid_tiles (the single-frame call: a host-planned tile grid taken through all levels inside LDS) against the per-level
    k_resize launches of the batch path (itself checked against the oracle above) on random image sizes, level counts and scale factors:
    every byte of every level - tile boundaries, halos, right / bottom borders, levels whose tiles own nothing.  Also the results, which
    read the pyramid and its blurred copy."""
    rng = np.random.default_rng(20260926)
    done = 0
    for trial in range(40):
        W, H = int(rng.integers(120, 1500)), int(rng.integers(120, 1100))
        nl = int(rng.integers(2, 11))
        sf = float(rng.choice([1.1, 1.15, 1.2, 1.25, 1.3, 1.4, 1.5, 1.6]))
        if min(W, H) / sf ** (nl - 1) < 64 or not (0.26 < (W / sf ** (nl - 1) - 32) / max(H / sf ** (nl - 1) - 32, 1) < 4.4):
            continue            # (level too small for a cell grid / aspect ratio outside 1..4 initial quadtree nodes: rejected at handle creation)
        nf = int(rng.integers(100, 1500))
        try:
            one = orbx.ORBextractor(nf, sf, nl, 20, 7, max_width=W, max_height=H)
            two = orbx.ORBextractor(nf, sf, nl, 20, 7, max_width=W, max_height=H, max_batch=2)
            im = orbx.synth_frame(900 + trial, W, H)
            k1, d1 = one(im)
            k2, d2, c2 = two.extract_batch([im, im])
        except RuntimeError:
            continue            # geometry outside the documented limits
        n = int(c2[0])
        assert len(k1) == n and (kp_matrix(k1).view(np.uint32) == kp_matrix(k2[0, :n]).view(np.uint32)).all() and (d1 == d2[0, :n]).all(), (W, H, nl, sf)
        for l in range(nl):
            assert (one.mvImagePyramid(l) == two.mvImagePyramid(l)).all(), ("pyramid", W, H, nl, sf, l)
        # (the blurred copy of a single-frame call stays on the shared engine; the descriptors above were sampled from it)
        one.close(); two.close()
        done += 1
    assert done >= 12


def test_single_frame_calls_from_sixteen_threads_are_combined(orbx):
    """Sixteen threads, one one-frame extractor each (the reference's rule: one ORBextractor per thread, include/ORBextractor.h:161; its stereo
    constructor uses two), three image geometries interleaved: calls that are inside the library at the same moment are COMBINED into one
    launch set per geometry (csrc/orbx_extractor.hip, "the combiner"); the engines and their per-size graphs are built concurrently on first
    use.  Every call - keypoints, descriptors AND the host pyramid that comes back with them - must equal what the batch path (no combiner:
    a max_batch = 2 handle, itself checked against the oracle above) gives for the same frame."""
    import threading
    sizes = [(640, 480, 1000), (752, 480, 1200), (1241, 376, 2000)]
    frames = {sz: [orbx.synth_frame(500 + i, sz[0], sz[1]) for i in range(6)] for sz in sizes}
    want, wantPyr = {}, {}
    for sz in sizes:
        ref = orbx.ORBextractor(sz[2], 1.2, 8, 20, 7, max_width=sz[0], max_height=sz[1], max_batch=2)
        want[sz], wantPyr[sz] = [], []
        for im in frames[sz]:
            k, d, c = ref.extract_batch([im, im])
            want[sz].append((k[0, :c[0]].copy(), d[0, :c[0]].copy()))
            wantPyr[sz].append([ref.mvImagePyramid(l).copy() for l in range(8)])
        ref.close()
    errors, stats = [], {}
    NT, REPS = 16, 5
    start = threading.Barrier(NT)

    def work(t):
        sz = sizes[t % 3] if t >= 4 else sizes[0]       # 8 threads on 640x480, 4 + 4 on the other two
        try:
            ext = orbx.ORBextractor(sz[2], 1.2, 8, 20, 7, max_width=sz[0], max_height=sz[1])
            start.wait()
            for rep in range(REPS):
                for i, im in enumerate(frames[sz]):
                    if (rep + t) % 2:
                        k, d, pyr = ext.extract_with_pyramid(im)
                    else:
                        (k, d), pyr = ext(im), None
                    kw, dw = want[sz][i]
                    if len(k) != len(kw) or not (kp_matrix(k).view(np.uint32) == kp_matrix(kw).view(np.uint32)).all() or not (d == dw).all():
                        errors.append((t, rep, i, "results"))
                    if pyr is not None and not all(p.shape == w.shape and (p == w).all() for p, w in zip(pyr, wantPyr[sz][i])):
                        errors.append((t, rep, i, "pyramid"))
                    # the device-resident state of the member is the frame's too (what a matcher chained behind the call reads)
                    if rep == REPS - 1 and i == 5 and not (ext.mvImagePyramid(3) == wantPyr[sz][i][3]).all():
                        errors.append((t, rep, i, "device pyramid"))
            stats[sz] = ext.combiner_stats()
            ext.close()
        except Exception as e:      # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(NT)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:5]
    for sz, (batches, nframes, engines) in stats.items():
        assert nframes >= 4 * REPS * 6 and 1 <= batches <= nframes and 1 <= engines <= 2, (sz, batches, nframes, engines)
    b0, f0, _ = stats[sizes[0]]
    assert f0 > b0, "eight threads on one geometry never met in a launch set: %d frames in %d sets" % (f0, b0)


def test_a_partner_hint_makes_one_launch_set_of_two(orbx, monkeypatch):
    """The stereo Frame constructor's two extractor threads (src/Frame.cc:159-167): each call announces the other
    (orbx_extractor_expect_partner, set by the HIP body of Frame::ExtractORB); with ORBX_COMBINE_PARTNER_US set, the pair runs as ONE set of two
    frames - bit-identical to two separate calls.  (The wait is OFF by default: at the reference's thread-start skew it costs more than the
    second launch set it saves - csrc/orbx_extractor.hip.)  Python threads leave a barrier with some jitter; most (not all) trials must merge."""
    import threading
    monkeypatch.setenv("ORBX_COMBINE_PARTNER_US", "300")
    W, H, nf = 1241, 376, 2000
    left = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H)
    right = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H)
    imgs = [orbx.synth_frame(700 + i, W, H) for i in range(4)]
    want = [left(im) for im in imgs]
    merged, trials = 0, 12
    for trial in range(trials):
        b0, f0, _ = left.combiner_stats()
        out = {}
        bar = threading.Barrier(2)

        def run(name, ext, other, im):
            ext.expect_partner(other)
            bar.wait()
            out[name] = ext(im)
        ta = threading.Thread(target=run, args=("l", left, right, imgs[trial % 4]))
        tb = threading.Thread(target=run, args=("r", right, left, imgs[(trial + 1) % 4]))
        ta.start(); tb.start(); ta.join(); tb.join()
        b1, f1, _ = left.combiner_stats()
        assert f1 - f0 == 2
        merged += (b1 - b0) == 1
        for name, idx in (("l", trial % 4), ("r", (trial + 1) % 4)):
            k, d = out[name]
            assert len(k) == len(want[idx][0]) and (kp_matrix(k).view(np.uint32) == kp_matrix(want[idx][0]).view(np.uint32)).all() and (d == want[idx][1]).all()
    assert merged >= trials // 2, "only %d of %d announced pairs ran as one launch set" % (merged, trials)
    # a hint whose partner never calls costs a bounded wait and nothing else
    left.expect_partner(right)
    k, d = left(imgs[0])
    assert len(k) == len(want[0][0]) and (d == want[0][1]).all()
    left.close(); right.close()


def test_the_handles_own_graph_still_serves_single_frames(orbx, monkeypatch):
    """ORBX_COMBINE=0: the one-frame call on the handle's own graph (the path every call took before the combiner existed) - same results,
    same host pyramid."""
    monkeypatch.setenv("ORBX_COMBINE", "0")
    W, H = 752, 480
    ext = orbx.ORBextractor(1200, 1.2, 8, 20, 7, max_width=W, max_height=H)
    ref = orbx.ORBextractor(1200, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2)
    for seed in (3, 4):
        im = orbx.synth_frame(seed, W, H)
        k, d, pyr = ext.extract_with_pyramid(im)
        kb, db, cb = ref.extract_batch([im, im])
        assert len(k) == cb[0] and (kp_matrix(k).view(np.uint32) == kp_matrix(kb[0, :cb[0]]).view(np.uint32)).all() and (d == db[0, :cb[0]]).all()
        assert all((pyr[l] == ref.mvImagePyramid(l)).all() for l in range(8))
    assert ext.combiner_stats()[1] == 0
    ext.close(); ref.close()


@pytest.mark.parametrize("W,H,nini,nl", [(1000, 230, 5, 4), (1241, 230, 6, 4), (1400, 200, 8, 2)])
def test_wide_images_with_five_to_eight_initial_quadtree_nodes(orbx, oracle, W, H, nini, nl):
    """DistributeOctTree starts from round(width / height) root nodes (src/ORBextractor.cc:719-766): panoramic images give more than the four of
    the usual camera formats.  Batch and single-frame paths against the restatement, bit for bit."""
    assert round((W - 32) / (H - 32)) == nini
    nf = 800
    rst = oracle.restatement(nf, 1.2, nl)
    ext = orbx.ORBextractor(nf, 1.2, nl, 20, 7, max_width=W, max_height=H, max_batch=2)
    frames = [orbx.synth_frame(300 + nini, W, H), orbx.synth_frame(310 + nini, W, H, orbx.SYNTH_LOW_TEXTURE)]
    kps, desc, counts = ext.extract_batch(frames)
    one = orbx.ORBextractor(nf, 1.2, nl, 20, 7, max_width=W, max_height=H)
    for f, im in enumerate(frames):
        ko, do = rst.extract(im)
        n = int(counts[f])
        assert n == len(ko) and n > 100
        assert (kp_matrix(kps[f, :n]).view(np.uint32) == ko.view(np.uint32)).all() and (desc[f, :n] == do).all()
        k1, d1 = one(im)
        assert len(k1) == n and (kp_matrix(k1).view(np.uint32) == ko.view(np.uint32)).all() and (d1 == do).all()
    ext.close(); one.close()


@pytest.mark.gpu
@pytest.mark.parametrize("taps", [(19, 34, 48, 56, 48, 34, 18), (10, 30, 50, 76, 50, 30, 10), (4, 20, 60, 88, 52, 24, 8), (0, 0, 0, 255, 2, 0, 0)])
def test_blur_with_configured_taps(orbx, oracle, taps):
    """orbx_extractor_config::gauss_taps (include/orbx.h): the 7x7 blur in 8-bit fixed point with the caller's taps - a set that sums to 257 (the clamping
    instantiation of k_blur / k_octree_blur: a white patch reaches 257 before the clamp), one that sums to 256, an asymmetric one (even and odd output rows of
    the vertical pass use different tap pairs) and a near-identity.  Batch path and combined single-frame path against op_gauss7_u8 of the restatement."""
    W, H, nf = 640, 480, 1000
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2, gauss_taps=taps)
    ext.set_debug_taps(True)
    rst = oracle.restatement(nf)
    white = np.full((H, W), 255, np.uint8)
    white[::37, ::41] = 0                 # (a few corners, so that every level has keypoints)
    frames = [orbx.synth_frame(61, W, H), white]
    ext.extract_batch(frames)
    for f, im in enumerate(frames):
        pyr = oracle.pyramid(rst, im)
        for l in range(8):
            want = oracle.blur(pyr[l], taps)
            got = ext.mvImagePyramid(l, frame=f, blurred=True)
            assert (got == want).all(), "batch path: frame %d level %d" % (f, l)
    kps, desc, counts = ext.extract_batch(frames)
    # the single-frame paths: a one-frame handle with the taps on (its own graph: k_octree + k_blur, the blurred levels can be read back), and without
    # them (the combined launch set: k_octree_blur, whose blurred pyramid stays on the shared engine - its descriptors must be the batch path's)
    one = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1, gauss_taps=taps)
    for f, im in enumerate(frames):
        k1, d1 = one(im)
        n = int(counts[f])
        assert len(k1) == n and (kp_matrix(k1).view(np.uint32) == kp_matrix(kps[f, :n]).view(np.uint32)).all() and (d1 == desc[f, :n]).all(), "combined single-frame path, frame %d" % f
    one.set_debug_taps(True)
    for im in frames:
        one(im)
        pyr = oracle.pyramid(rst, im)
        for l in range(8):
            assert (one.mvImagePyramid(l, frame=0, blurred=True) == oracle.blur(pyr[l], taps)).all(), "single-frame path: level %d" % l
    one.close()
    ext.close()


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,nf", [(640, 480, 1000), (752, 480, 2000)])
def test_white_noise_and_saturated_frames(orbx, oracle, W, H, nf):
    """Inputs at the ends of the detector's range: uniform white noise (a third of all pixels pass the compass pre-test: the exact-score phase runs its
    maximum number of passes per cell, the candidate lists are at their densest), a two-level noise image (every arc maximal), a frame of isolated
    single-pixel spikes (each one the only keypoint of its cell) - batch path and single-frame path against the restatement."""
    rng = np.random.Generator(np.random.PCG64(77))
    noise = rng.integers(0, 256, (H, W), dtype=np.uint8)
    two = np.where(rng.integers(0, 2, (H, W)) > 0, 255, 0).astype(np.uint8)
    spikes = np.full((H, W), 90, np.uint8)
    spikes[20::31, 23::29] = 200
    frames = [noise, two, spikes]
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=len(frames))
    rst = oracle.restatement(nf)
    kps, desc, counts = ext.extract_batch(frames)
    for f, im in enumerate(frames):
        ko, do = rst.extract(im)
        n = int(counts[f])
        assert n == len(ko), "frame %d: %d keypoints, the restatement finds %d" % (f, n, len(ko))
        assert (kp_matrix(kps[f, :n]).view(np.uint32) == ko.view(np.uint32)).all() and (desc[f, :n] == do).all(), "frame %d" % f
    one = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1)      # (a one-frame handle: the combined single-frame launch set)
    for f, im in enumerate(frames):
        ko, do = rst.extract(im)
        k1, d1 = one(im)
        assert len(k1) == len(ko) and (kp_matrix(k1).view(np.uint32) == ko.view(np.uint32)).all() and (d1 == do).all(), "single-frame path, frame %d" % f
    one.close()
    ext.close()
