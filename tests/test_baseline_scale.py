"""GPU tests at the sizes BASELINE.json quotes (the other GPU tests use batches of <= 8 frames):
 * configs[1]: a batch of 256 synthetic 640x480 frames, 1000 features, bit-exact keypoints + descriptors vs the CPU restatement
   for EVERY frame (buffer arithmetic at B = 256: 1.7 GB of scratch, size_t pitches);
 * configs[2]: a batch of KITTI-shaped 1241x376 stereo pairs, 2000 features, extract + ORBmatcher::SearchByBoW L<->R on the
   device-resident path, index-exact vs the restated matcher on the same features;
 * the device status word: a resident pipeline (no orbx_batch_download) learns about a capacity overflow."""
import concurrent.futures as cf
import os

import numpy as np
import pytest

import oracle_lib


def _kp_array(k):
    return np.stack([k[c].astype(np.float32) for c in ("x", "y", "size", "angle", "response", "octave", "class_id")], 1)


@pytest.mark.gpu
def test_config2_batch_256_bit_exact(orbx, oracle):
    W, H, nf, B = 640, 480, 1000, 256
    frames = orbx.synth_sequence(777, B, W, H)
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    ext.run_device(*ext.upload(frames))
    assert ext.status() == 0
    kps, desc, counts = ext.download(B)

    def cpu(chunk):
        rst = oracle.restatement(nf)            # one instance per thread (not re-entrant, like the reference's)
        return [rst.extract(frames[i]) for i in chunk]
    nthr = min(16, os.cpu_count() or 1)
    chunks = [list(range(t, B, nthr)) for t in range(nthr)]
    with cf.ThreadPoolExecutor(nthr) as ex:
        res = list(ex.map(cpu, chunks))
    total = 0
    for chunk, out in zip(chunks, res):
        for i, (ko, do) in zip(chunk, out):
            n = int(counts[i])
            assert n == len(ko), (i, n, len(ko))
            assert (_kp_array(kps[i, :n]).view(np.uint32) == ko.view(np.uint32)).all(), i
            assert (desc[i, :n] == do).all(), i
            total += n
    assert total > 200 * B


@pytest.mark.gpu
def test_config3_stereo_batch_extract_and_search_by_bow_left_right(orbx, oracle):
    W, H, nf, B = 1241, 376, 2000, 16
    seeds = [500 + i for i in range(B)]
    frames = [orbx.synth_frame(s, W, H) for s in seeds] + [orbx.synth_frame(s, W, H, orbx.SYNTH_STEREO_RIGHT) for s in seeds]
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2 * B)    # KITTI00-02.yaml
    mt = orbx.ORBmatcher(0.7, True, max_features=ext.capacity, max_pairs=B)
    ext.run_device(*ext.upload(frames))
    fs = orbx.ORBmatcher.features_of(ext, 2 * B)
    fl, fr = np.arange(B, dtype=np.int32), np.arange(B, 2 * B, dtype=np.int32)
    mt.search_by_bow_device(fs, fs, fl, fr, mode=0, after=ext)         # left image as "KeyFrame", right image as "Frame"
    m, d, nm = mt.download(B)
    kps, desc, counts = ext.download(2 * B)
    rst = oracle.restatement(nf)
    for i in (0, 5, B - 1):                                            # extraction of this geometry, bit-exact
        ko, do = rst.extract(frames[i])
        n = int(counts[i])
        assert n == len(ko) and (_kp_array(kps[i, :n]).view(np.uint32) == ko.view(np.uint32)).all() and (desc[i, :n] == do).all()
    for p in range(B):                                                 # every pair, index-exact
        nl, nr = int(counts[p]), int(counts[B + p])
        wn, wm = oracle_lib.search_by_bow(oracle, 0, kps[p, :nl], desc[p, :nl], kps[B + p, :nr], desc[B + p, :nr], 0.7, True)
        assert int(nm[p]) == wn and (m[p, :len(wm)] == wm).all(), p   # mode 0: one entry per Frame (right image) feature
    assert nm.mean() > 8          # brute force over 2000 x 2000 repeated-texture features: the 0.7 ratio test keeps few (ComputeStereoMatches' row bands keep ~500)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,nf,B", [(640, 480, 1000, 64), (1241, 376, 2000, 8)])
def test_matrix_core_candidate_lists_equal_the_popcount_ones(orbx, monkeypatch, W, H, nf, B):
    """k_bow_topk_mfma (ORBX_MATCH_MFMA=1: Hamming distances as v_mfma_i32_32x32x32_i8 products) against k_bow_topk (the default) on extracted
    frames of one scene - consecutive views, the headline's workload: every match index, distance and count identical, both modes."""
    frames = orbx.synth_sequence(900, B, W, H)
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    mt = orbx.ORBmatcher(0.7, True, max_features=ext.capacity, max_pairs=B - 1)
    ext.run_device(*ext.upload(frames))
    ext.sync()                         # (the matcher below is not chained to the extractor: its stream knows nothing of this batch)
    fs = orbx.ORBmatcher.features_of(ext, B)
    fa, fb = np.arange(B - 1, dtype=np.int32), np.arange(1, B, dtype=np.int32)
    for mode in (0, 1):
        res = {}
        for sw in ("0", "1"):
            monkeypatch.setenv("ORBX_MATCH_MFMA", sw)
            mt.search_by_bow_device(fs, fs, fa, fb, mode=mode)
            res[sw] = [x.copy() for x in mt.download(B - 1)]
        for x, y in zip(res["0"], res["1"]):
            assert x.shape == y.shape and (x == y).all(), mode
        assert res["0"][2].mean() > 100
    mt.close(); ext.close()


@pytest.mark.gpu
def test_white_noise_is_extracted_like_the_reference(orbx):
    """White noise gives > 100k FAST candidates in level 0 of a 1241x376 frame.  The reference's std::vectors just grow
    (src/ORBextractor.cc:1075 only reserves); the quadtree's point arrays are sized for the worst case - every second pixel in both
    directions a maximum -, so there is no capacity error any more: results equal the compiled reference's bit for bit and the
    device status word that resident consumers inherit stays 0."""
    import oracle_lib
    W, H = 1241, 376
    rng = np.random.default_rng(0)
    noise = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(2)]
    ext = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2)
    mt = orbx.ORBmatcher(0.7, True, max_features=ext.capacity, max_pairs=1)
    ext.run_device(*ext.upload(noise))
    fs = orbx.ORBmatcher.features_of(ext, 2)
    mt.search_by_bow_device(fs, fs, np.array([0], np.int32), np.array([1], np.int32), mode=0, after=ext)
    assert ext.status() == 0
    mt.download(1)
    kps, desc, counts = ext.download(2)
    assert min(ext.debug_candidates(0, f)[1] for f in range(2)) > 32768              # beyond the former per-level buffer of the quadtree
    orc = oracle_lib.Oracle()
    ref = orc.reference(2000) if orc.ref is not None else orc.restatement(2000)
    for f in range(2):
        ko, do = ref.extract(noise[f], cap=16384)
        n = int(counts[f])
        got = np.stack([kps[f, :n][c].astype(np.float32) for c in ("x", "y", "size", "angle", "response", "octave", "class_id")], 1)
        assert n == len(ko) and (got.view(np.uint32) == ko.view(np.uint32)).all()
        assert (desc[f, :n] == do).all()
