"""DBoW2 vocabulary transform (Frame::ComputeBoW): restatement and HIP kernel against the
reference's own TemplatedVocabulary (vendored DBoW2 compiled into oracle/_ref/liborbslam.so),
which loads the same synthetic tree through its own loadFromTextFile."""
import numpy as np
import pytest

import oracle_lib

HAVE_REF = oracle_lib.slam_lib() is not None
CASES = [(10, 3, 11, 4), (10, 4, 12, 2), (6, 5, 13, 4), (10, 3, 14, 1)]   # k, L, seed, levelsup


def _descs(orbx, voc, n, seed):
    """descriptors near random tree nodes + pure noise"""
    rng = np.random.default_rng(seed)
    leaves = np.flatnonzero(voc["is_leaf"])
    d = voc["desc"][rng.choice(leaves, n)].copy()
    for i in range(n):
        for b in rng.integers(0, 256, rng.integers(0, 30)):
            d[i, b >> 3] ^= np.uint8(1 << (b & 7))
    d[: n // 10] = rng.integers(0, 256, (n // 10, 32), dtype=np.uint8)
    return d


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/liborbslam.so not built (needs /root/reference)")
@pytest.mark.parametrize("k,L,seed,levelsup", CASES)
def test_restatement_equals_reference_vocabulary(orbx, oracle, tmp_path, k, L, seed, levelsup):
    voc = orbx.voc_synth.make_vocabulary(k, L, seed)
    path = tmp_path / "voc.txt"
    orbx.voc_synth.write_text(voc, path)
    ref = oracle_lib.RefVocabulary(path)
    assert ref.size() == int(voc["is_leaf"].sum())
    d = _descs(orbx, voc, 1500, seed)
    want = ref.transform(d, levelsup)
    word, node, weight = oracle_lib.voc_transform(oracle, voc, d, levelsup)
    assert (word == want["word"]).all() and (weight == want["weight"]).all()
    if L - levelsup >= 1:
        assert (node == want["node"]).all()
    else:
        assert (node == 0).all() and (want["fv_node"][weight > 0] == 0).all()
    # FeatureVector: features with a zero-weight word are not filed
    fv = np.where(weight > 0, node, -1)
    assert (fv == want["fv_node"]).all() and (weight == 0).any()
    ids, vals = oracle_lib.bow_vector(word, weight)
    assert (ids == want["bow_ids"]).all() and (vals.view(np.uint64) == want["bow_vals"].view(np.uint64)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("k,L,seed,levelsup", CASES)
def test_hip_transform_equals_oracle_and_reference(orbx, oracle, tmp_path, k, L, seed, levelsup):
    voc = orbx.voc_synth.make_vocabulary(k, L, seed)
    V = orbx.Vocabulary(voc)
    assert V.size() == int(voc["is_leaf"].sum())
    d = _descs(orbx, voc, 3000, seed)
    word, node, weight = V.transform(d, levelsup)
    w2, n2, wt2 = oracle_lib.voc_transform(oracle, voc, d, levelsup)
    assert (word == w2).all() and (weight.view(np.uint64) == wt2.view(np.uint64)).all()
    assert (node == np.where(wt2 > 0, n2, -1)).all()
    if HAVE_REF:
        path = tmp_path / "voc.txt"
        orbx.voc_synth.write_text(voc, path)
        want = oracle_lib.RefVocabulary(path).transform(d, levelsup)
        assert (word == want["word"]).all() and (weight == want["weight"]).all() and (node == want["fv_node"]).all()
    V.close()


@pytest.mark.gpu
def test_transform_feeds_search_by_bow_on_device(orbx, oracle):
    """extract -> transform (node ids stay on the device) -> SearchByBoW gated by those node ids,
    vs the oracle run on the downloaded node ids."""
    W, H, nf, B = 640, 480, 1000, 4
    voc = orbx.voc_synth.make_vocabulary(10, 3, 21)
    V = orbx.Vocabulary(voc)
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    frames = orbx.synth_sequence(77, B, W, H)
    ext.run_device(*ext.upload(frames))
    V.transform_device(ext, 1)                       # L=3, levelsup=1: 100-ish nodes at depth 2
    groups, cap = V.groups_device()
    fs = orbx.ORBmatcher.features_of(ext, B)
    assert cap == fs.capacity
    fs.groups = groups.value
    mt = orbx.ORBmatcher(0.7, True, max_features=ext.capacity, max_pairs=B - 1)
    pa, pb = np.arange(B - 1, dtype=np.int32), np.arange(1, B, dtype=np.int32)
    mt.search_by_bow_device(fs, fs, pa, pb, mode=0, after=ext)
    m, dd, nm = mt.download(B - 1)
    kps, desc, counts = ext.download(B)
    word, node, weight = V.download(ext, B)
    total = 0
    for p in range(B - 1):
        a, b = int(pa[p]), int(pb[p])
        na, nb = int(counts[a]), int(counts[b])
        w2, n2, wt2 = oracle_lib.voc_transform(oracle, voc, desc[a, :na], 1)
        assert (word[a, :na] == w2).all() and (node[a, :na] == np.where(wt2 > 0, n2, -1)).all()
        gA, gB = node[a, :na], node[b, :nb]          # -1 = not filed in the FeatureVector: never matched
        wn, wm = oracle_lib.search_by_bow(oracle, 0, kps[a, :na], desc[a, :na], kps[b, :nb], desc[b, :nb], 0.7, True, gA, gB)
        assert nm[p] == wn and (m[p, :nb] == wm).all()
        total += int(nm[p])
    assert total > 50
    mt.close(); ext.close(); V.close()
