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
def test_device_status_word_reaches_a_resident_consumer(orbx):
    """White noise drives a level past the 32768-candidate buffer (INTEGRATION.md limits).  A device-resident pipeline never calls
    orbx_batch_download; it must still fail loudly: orbx_extractor_status reports the bits, and the matcher chained behind the
    extractor refuses to hand out its results."""
    W, H = 1241, 376
    rng = np.random.default_rng(0)
    noise = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(2)]
    ext = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2)
    mt = orbx.ORBmatcher(0.7, True, max_features=ext.capacity, max_pairs=1)
    ext.run_device(*ext.upload(noise))
    fs = orbx.ORBmatcher.features_of(ext, 2)
    mt.search_by_bow_device(fs, fs, np.array([0], np.int32), np.array([1], np.int32), mode=0, after=ext)
    assert ext.status() & 1
    with pytest.raises(orbx.OrbxError) as e:
        mt.download(1)
    assert e.value.code == -3 and "overflowed" in str(e.value)          # ORBX_ERR_CAPACITY
    with pytest.raises(orbx.OrbxError):
        ext.download(2)
    # the handles recover with the next (ordinary) batch
    ok = [orbx.synth_frame(1, W, H), orbx.synth_frame(2, W, H)]
    ext.run_device(*ext.upload(ok))
    fs = orbx.ORBmatcher.features_of(ext, 2)
    mt.search_by_bow_device(fs, fs, np.array([0], np.int32), np.array([1], np.int32), mode=0, after=ext)
    assert ext.status() == 0
    mt.download(1)
