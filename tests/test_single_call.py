"""The single-call host entry points of the tracking thread (round 6: mapped pinned inputs / results + a polled sequence word instead of a copy engine and
a stream synchronisation, csrc/orbx_internal.h: OrbxCallBox): shapes the batch tests do not reach, the round-5 path as a cross-check, the new entry
points (orbx_bow_transform_sorted, orbx_bow_job_*), and many different calls on ONE handle (they share its box and its device arena)."""
import numpy as np
import pytest

import oracle_lib
from test_bow_transform import _descs
from test_matcher import _kps, _noisy_pair, _rand_desc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 1])
def test_pair_kernel_on_ragged_shapes(orbx, oracle, mode):
    """k_bow_topk_pair: 16 KeyFrame features per workgroup x 16 Frame slices - feature counts that are not multiples of either, one feature on a side, more
    Frame features than the LDS tile takes (the call falls back to the staged batch kernels), node ids with unfiled (-1) features and validity masks."""
    rng = np.random.default_rng(100 + mode)
    mt = orbx.ORBmatcher(0.7, True, max_features=4600)
    for nA, nB, groups in [(1, 1, False), (1, 37, True), (17, 1, False), (15, 16, True), (16, 17, False), (31, 255, True), (33, 257, True), (1005, 1011, False),
                           (1005, 1011, True), (2000, 3999, True), (700, 4000, False), (300, 4500, True)]:
        n = max(nA, nB)
        kA, dA, kB, dB = _noisy_pair(rng, n, orbx, extra=max(0, nB - n - n // 5))
        kA, dA, kB, dB = kA[:nA], dA[:nA], kB[:nB], dB[:nB]
        gA = gB = vA = vB = None
        if groups:
            gA = rng.integers(-1, 9, nA).astype(np.int32) * 11          # -11 = not filed in the FeatureVector
            gB = rng.integers(-1, 9, nB).astype(np.int32) * 11
            vA = (rng.random(nA) < 0.85).astype(np.uint8)
            vB = (rng.random(nB) < 0.9).astype(np.uint8)
        want_n, want = oracle_lib.search_by_bow(oracle, mode, kA, dA, kB, dB, 0.7, True, gA, gB, vA, vB)
        got_n, got = mt.SearchByBoW(kA, dA, kB, dB, gA, gB, vA, vB, mode=mode)
        assert got_n == want_n and (got == want).all(), (nA, nB, groups)
    mt.close()


def test_pair_kernel_equals_the_round5_path(orbx, monkeypatch):
    """ORBX_BOW_SINGLE_SPLIT is read once per process: the cross-check runs the round-5 path in a child process and compares the match lists."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    code = r'''
import importlib, json, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
orbx = importlib.import_module("self_commit_orb-slam2_amd")
from test_matcher import _noisy_pair
rng = np.random.default_rng(77)
mt = orbx.ORBmatcher(0.8, True, max_features=2600)
out = []
for n in (60, 999, 1500):
    kA, dA, kB, dB = _noisy_pair(rng, n, orbx)
    g = rng.integers(0, 3, len(kA)).astype(np.int32) if n != 999 else None
    gb = rng.integers(0, 3, len(kB)).astype(np.int32) if n != 999 else None
    for mode in (0, 1):
        nm, m = mt.SearchByBoW(kA, dA, kB, dB, g, gb, mode=mode)
        out.append([int(nm), [int(x) for x in m]])
print(json.dumps(out))
''' % (str(Path(__file__).resolve().parent.parent), str(Path(__file__).resolve().parent))
    import os
    res = []
    for split in ("1", "0"):
        env = dict(os.environ, ORBX_BOW_SINGLE_SPLIT=split)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert res[0] == res[1]
    assert all(nm > 5 for nm, _ in res[0]), [nm for nm, _ in res[0]]


def test_sorted_transform_orders_and_the_job_form(orbx, oracle):
    """orbx_bow_transform_sorted: word / node / weight as orbx_bow_transform, by_word / by_node = the filed features in ascending (key, index) order - the
    order in which TemplatedVocabulary::transform's loop meets the keys of its two maps; orbx_bow_job_* = the same five arrays from the features an
    extractor call left on the device."""
    voc = orbx.voc_synth.make_vocabulary(10, 4, 21)
    V = orbx.Vocabulary(voc)
    for n in (1, 31, 33, 1000, 2049, 3000):
        d = _descs(orbx, voc, n, 500 + n)
        w0, n0, wt0 = V.transform(d, 2)
        w, nd, wt, bw, bn = V.transform_sorted(d, 2)
        assert (w == w0).all() and (nd == n0).all() and (wt.view(np.uint64) == wt0.view(np.uint64)).all()
        filed = np.flatnonzero(nd >= 0)
        assert len(bw) == len(filed) == len(bn) and (wt[filed] > 0).all()
        assert (bw == filed[np.lexsort((filed, w[filed]))]).all(), n
        assert (bn == filed[np.lexsort((filed, nd[filed]))]).all(), n
    # the job form on a real frame's descriptors
    W, H = 640, 480
    ext = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1)
    for seed in (3, 4):
        kps, desc = ext(orbx.synth_frame(seed, W, H))
        got = V.job_transform(ext, 2)
        want = V.transform_sorted(desc, 2)
        assert len(got[0]) == len(desc)
        for a, b in zip(got, want):
            assert a.dtype == b.dtype and (a.view(np.uint8) == b.view(np.uint8)).all()
    ext.close(); V.close()


def test_many_kinds_of_calls_share_one_matcher_handle(orbx, oracle):
    """SearchByBoW, SearchByProjection(F, points) and SearchByProjection(Current, Last) interleaved on ONE handle, sizes going up and down: every call
    re-sizes the shared box / arena and must leave nothing behind for the next kind."""
    from test_projection import _struct_kps, make_case, make_last_case
    rng = np.random.default_rng(9)
    mt = orbx.ORBmatcher(0.8, True, max_features=3000)
    cache = {}
    for rnd in range(3):
        for n in (800, 60, 2000):
            if n not in cache:
                kA, dA, kB, dB = _noisy_pair(rng, n, orbx)
                fr, pts = make_case(40 + n % 7, n=min(n, 1500), m=2 * n, crowded=n == 60)
                fl, last = make_last_case(50 + n % 5, n=min(n, 1500), nl=n)
                cache[n] = (kA, dA, kB, dB, oracle_lib.search_by_bow(oracle, 0, kA, dA, kB, dB, 0.8, True), fr, pts, oracle_lib.search_by_projection(oracle, fr, pts, 3.0, 0.8),
                            fl, last, oracle_lib.search_by_projection_last(oracle, fl, last, 7.0, True, True))
            kA, dA, kB, dB, (wn, wm), fr, pts, (pn, pm), fl, last, (ln, lm) = cache[n]
            gn, gm = mt.SearchByBoW(kA, dA, kB, dB)
            assert gn == wn and (gm == wm).all(), (rnd, n, "bow")
            got = mt.SearchByProjection(dict(fr, kps=_struct_kps(orbx, fr["kps7"])), pts, 3.0)
            assert got[0] == pn and (got[1] == pm).all(), (rnd, n, "projection")
            lastd = dict(last, kps=_struct_kps(orbx, last["kps7"]), valid=(last["valid"] == 1).astype(np.uint8))
            got = mt.SearchByProjectionLast(dict(fl, kps=_struct_kps(orbx, fl["kps7"])), lastd, 7.0, True)
            assert got[0] == ln and (got[1] == lm).all(), (rnd, n, "projection last")
    mt.close()


@pytest.mark.parametrize("layout", ["four_way_tie", "tie_for_third", "second_below_a_tenth", "third_below_a_tenth", "one_bin", "wrap_bin"])
def test_three_maxima_ties_and_the_tenth_rule(orbx, oracle, layout):
    """ComputeThreeMaxima (src/ORBmatcher.cc:1866-1908) inside the replay kernels is one wave with a bin per lane: the three largest counts, EQUAL counts in bin
    order (the sequential scan's strict comparisons), bins below a tenth of the largest dropped.  Distinct descriptors match 1:1, so the histogram is whatever
    the angles say: exact ties among the top bins, a tie for the third place, both branches of the tenth rule, a single bin, and the bin that wraps (30 -> 0)."""
    rng = np.random.default_rng(11)
    bins = {"four_way_tie": {3: 30, 7: 30, 12: 30, 20: 30}, "tie_for_third": {1: 50, 5: 40, 9: 20, 14: 20, 22: 20}, "second_below_a_tenth": {4: 100, 8: 9, 15: 9},
            "third_below_a_tenth": {2: 100, 6: 50, 11: 9, 13: 9}, "one_bin": {17: 64}, "wrap_bin": {0: 40, 29: 40, 10: 40, 11: 39}}[layout]
    n = sum(bins.values())
    dA = _rand_desc(rng, n)
    kA = _kps(rng, n, orbx)
    kB = kA.copy()
    rot = np.concatenate([np.full(c, 12.0 * b + (354.5 if (layout == "wrap_bin" and b == 0) else 0.0), np.float32) for b, c in bins.items()])   # bin = round(rot / 12); 354.5 / 12 rounds to 30 -> 0
    rot = rot[rng.permutation(n)]
    kA["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    kB["angle"] = ((kA["angle"].astype(np.float64) - rot) % 360).astype(np.float32)
    mt = orbx.ORBmatcher(0.9, True, max_features=1200)
    for mode in (0, 1):
        want_n, want = oracle_lib.search_by_bow(oracle, mode, kA, dA, kB, dA, 0.9, True)
        got_n, got = mt.SearchByBoW(kA, dA, kB, dA, mode=mode)
        assert got_n == want_n and (got == want).all(), (layout, mode)
        assert 0 < want_n < n or layout in ("one_bin",), (layout, want_n)      # the pruning did something (a single bin keeps everything)
    mt.close()
