"""Golden vectors produced by the COMPILED REFERENCE matcher / Frame sources
(tools/gen_golden_slam.py, oracle/_ref/liborbslam.so): mvuRight / mvDepth of the reference's
stereo Frame constructor and the results of both ORBmatcher::SearchByBoW overloads.  The CPU
restatement (CPU tests) and the HIP path (gpu tests) must reproduce them bit-for-bit /
index-for-index.  The fixtures travel to the GPU box; /root/reference does not."""
import importlib
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle_lib

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
GOLDEN = Path(__file__).resolve().parent / "golden" / "slam"
STEREO = sorted(GOLDEN.glob("stereo_*.npz"))
BOW = sorted(GOLDEN.glob("bow_*.npz"))


def test_golden_present():
    assert len(STEREO) >= 2 and len(BOW) >= 2


def _gen():
    return importlib.import_module("gen_golden_slam")


@pytest.mark.parametrize("path", STEREO, ids=lambda p: p.stem)
def test_stereo_restatement_matches_reference_golden(orbx, oracle, path):
    z = np.load(path)
    W, H, nf, bf = int(z["W"]), int(z["H"]), int(z["nfeatures"]), float(z["bf"])
    rst = oracle.restatement(nf)
    t, _, _ = rst.tables()
    for i, s in enumerate(z["seeds"]):
        imL = orbx.synth_frame(int(s), W, H)
        imR = orbx.synth_frame(int(s), W, H, orbx.SYNTH_STEREO_RIGHT)
        kL, dL = rst.extract(imL)
        kR, dR = rst.extract(imR)
        uR, dep, _ = oracle_lib.compute_stereo_matches(oracle, kL, dL, kR, dR, oracle.pyramid(rst, imL), oracle.pyramid(rst, imR), t[0], t[1], bf, 0.0)
        assert len(kR) == int(z["nR_%d" % i])
        assert (uR.view(np.uint32) == z["uRight_%d" % i].view(np.uint32)).all()
        assert (dep.view(np.uint32) == z["depth_%d" % i].view(np.uint32)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", STEREO, ids=lambda p: p.stem)
def test_stereo_hip_matches_reference_golden(orbx, path):
    z = np.load(path)
    W, H, nf, bf = int(z["W"]), int(z["H"]), int(z["nfeatures"]), float(z["bf"])
    seeds = [int(s) for s in z["seeds"]]
    n = len(seeds)
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2 * n)
    frames = [orbx.synth_frame(s, W, H) for s in seeds] + [orbx.synth_frame(s, W, H, orbx.SYNTH_STEREO_RIGHT) for s in seeds]
    ext.run_device(*ext.upload(frames))
    mt = orbx.ORBmatcher(0.7, True, max_features=ext.capacity, max_pairs=n)
    mt.compute_stereo_matches_device(ext, ext, list(range(n)), list(range(n, 2 * n)), bf, 0.0)
    uR, dep = mt.download_stereo(n)
    _, _, counts = ext.download(2 * n)
    for i in range(n):
        g = z["uRight_%d" % i]
        assert counts[i] == len(g) and counts[n + i] == int(z["nR_%d" % i])
        assert (uR[i, :len(g)].view(np.uint32) == g.view(np.uint32)).all()
        assert (dep[i, :len(g)].view(np.uint32) == z["depth_%d" % i].view(np.uint32)).all()
    mt.close()
    ext.close()


def _bow_inputs(orbx, z, extract):
    g = _gen()
    W, H, seed = int(z["W"]), int(z["H"]), int(z["seed"])
    imA = orbx.synth_frame(seed, W, H, 0, 0, 0, 0)
    imB = orbx.synth_frame(seed, W, H, 0, 1, 3, 1)
    (kA, dA), (kB, dB) = extract(imA), extract(imB)
    return kA, dA, kB, dB, g.synth_groups(dA), g.synth_groups(dB), g.synth_valid(dA, 5), g.synth_valid(dB, 7)


@pytest.mark.parametrize("path", BOW, ids=lambda p: p.stem)
def test_bow_restatement_matches_reference_golden(orbx, oracle, path):
    z = np.load(path)
    g = _gen()
    rst = oracle.restatement(int(z["nfeatures"]))

    def extract(im):
        k, d = rst.extract(im)
        return g.kps_struct(k), d
    kA, dA, kB, dB, gA, gB, vA, vB = _bow_inputs(orbx, z, extract)
    for mode in (0, 1):
        n, m = oracle_lib.search_by_bow(oracle, mode, kA, dA, kB, dB, 0.7, True)
        assert n == int(z["brute_n_%d" % mode]) and (m == z["brute_m_%d" % mode]).all()
        n, m = oracle_lib.search_by_bow(oracle, mode, kA, dA, kB, dB, 0.7, True, gA, gB, vA, vB if mode == 1 else None)
        assert n == int(z["nodes_n_%d" % mode]) and (m == z["nodes_m_%d" % mode]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", BOW, ids=lambda p: p.stem)
def test_bow_hip_matches_reference_golden(orbx, path):
    z = np.load(path)
    W, H, nf = int(z["W"]), int(z["H"]), int(z["nfeatures"])
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H)
    kA, dA, kB, dB, gA, gB, vA, vB = _bow_inputs(orbx, z, lambda im: ext(im))
    mt = orbx.ORBmatcher(0.7, True, max_features=ext.capacity)
    for mode in (0, 1):
        n, m = mt.SearchByBoW(kA, dA, kB, dB, mode=mode)
        assert n == int(z["brute_n_%d" % mode]) and (m == z["brute_m_%d" % mode]).all()
        n, m = mt.SearchByBoW(kA, dA, kB, dB, gA, gB, vA, vB if mode == 1 else None, mode=mode)
        assert n == int(z["nodes_n_%d" % mode]) and (m == z["nodes_m_%d" % mode]).all()
    mt.close()
    ext.close()
