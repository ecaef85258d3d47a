"""Golden vectors of the widened rows, produced by the COMPILED REFERENCE (tools/gen_golden_widen.py, oracle/_ref/liborbslam.so:
real KeyFrame / Frame / MapPoint objects): SearchForTriangulation, the Fuse search, the two remaining SearchByProjection
overloads, SearchForInitialization and Frame::isInFrustum.  The restatement (CPU tests) and the HIP path (gpu tests) must
reproduce them; inputs are regenerated from the stored seeds.  The fixtures travel to the GPU box; /root/reference does not."""
from pathlib import Path

import numpy as np
import pytest

import oracle_lib
from test_fuse import SF, TH_LOW
from test_triangulation import SIGMA2

G = Path(__file__).resolve().parent / "golden" / "slam"


def _z(name):
    return np.load(G / ("widen_%s.npz" % name))


def test_fixtures_present():
    for n in ("triangulation", "fuse", "area_search", "initialization", "frustum"):
        assert (G / ("widen_%s.npz" % n)).exists(), n


def _tri_cases(orbx):
    from test_triangulation import _scene
    z = _z("triangulation")
    for seed, forward, stereo, only, ori in z["cases"]:
        seed = int(seed)
        kf1, kf2, T1, T2, F12 = _scene(orbx, seed, forward=bool(forward), stereo_frac=float(stereo))
        yield kf1, kf2, F12, z["tri_%d_epi" % seed], bool(only), bool(ori), int(z["tri_%d_n" % seed]), z["tri_%d_m" % seed]


def test_triangulation_restatement(orbx, oracle):
    for kf1, kf2, F12, epi, only, ori, wn, wm in _tri_cases(orbx):
        n, m = oracle_lib.search_for_triangulation(oracle, kf1, kf2, F12, epi, SF, SIGMA2, only, ori)
        assert n == wn and (m == wm).all()


@pytest.mark.gpu
def test_triangulation_hip(orbx):
    for kf1, kf2, F12, epi, only, ori, wn, wm in _tri_cases(orbx):
        n, m = orbx.ORBmatcher(0.6, ori, max_features=len(kf1["kps"])).SearchForTriangulation(kf1, kf2, F12, epi, SF, SIGMA2, only)
        assert n == wn and (m == wm).all()


def _fuse_cases(orbx):
    from test_fuse import _scene
    z = _z("fuse")
    for overload, seed in z["cases"]:
        kf, Tt, sk, Ts, P, cdesc, rng = _scene(orbx, int(seed))
        pts = dict(desc=cdesc, **{k: z["fuse_%d_%s" % (seed, k)] for k in ("u", "v", "ur", "level", "radius", "active")})
        yield int(overload), kf, pts, z["fuse_%d_probe" % seed]


def test_fuse_restatement(orbx, oracle):
    for overload, kf, pts, probe in _fuse_cases(orbx):
        bi, bd = oracle_lib.fuse_best(oracle, kf, pts, overload == 1)
        assert (np.where((pts["active"] > 0) & (bd <= TH_LOW), bi, -1) == probe).all()


@pytest.mark.gpu
def test_fuse_hip(orbx):
    for overload, kf, pts, probe in _fuse_cases(orbx):
        bi, bd = orbx.ORBmatcher(0.6, True, max_features=max(len(kf["kps"]), len(probe))).FuseSearch(kf, pts, overload == 1)
        assert (np.where((pts["active"] > 0) & (bd <= TH_LOW), bi, -1) == probe).all()


def _area_cases(orbx):
    from test_area_search import _expected, _setup
    z = _z("area_search")
    for overload, seed in z["cases"]:
        overload, seed = int(overload), int(seed)
        ov = 3 if overload == 3 else 4
        kf, Tt, sk, Ts, P, cdesc, holder, lst, rng = _setup(orbx, seed, ov)
        pts = dict(desc=cdesc, ur=np.zeros(len(P), np.float32), **{k: z["area_%d_%s" % (seed, k)] for k in ("u", "v", "level", "radius", "active")})
        yield ov, kf, dict(points=pts), holder, lst, len(P), int(z["area_%d_n" % seed]), z["area_%d_holder" % seed], _expected


def test_area_search_restatement(orbx, oracle):
    for ov, kf, r, holder, lst, nc, wn, wholder, expected in _area_cases(orbx):
        nm, want, _, _ = expected(lambda f, q, d: oracle_lib.area_search_greedy(oracle, f, q, d), kf, r, holder, lst, ov, nc)
        assert nm == wn and (want == wholder).all()


@pytest.mark.gpu
def test_area_search_hip(orbx):
    for ov, kf, r, holder, lst, nc, wn, wholder, expected in _area_cases(orbx):
        mt = orbx.ORBmatcher(0.9, True, max_features=max(len(kf["kps"]), nc))
        nm, got, _, _ = expected(lambda f, q, d: mt.AreaSearchGreedy(f, q, d), kf, r, holder, lst, ov, nc)
        assert nm == wn and (got == wholder).all()


def _init_cases(orbx):
    from test_search_init import _frames
    z = _z("initialization")
    for seed, window, ratio, ori in z["cases"]:
        f1, f2, prev = _frames(orbx, int(seed))
        yield f1, f2, prev, int(window), float(ratio), bool(ori), int(z["init_%d_n" % int(seed)]), z["init_%d_m" % int(seed)]


def test_initialization_restatement(orbx, oracle):
    for f1, f2, prev, window, ratio, ori, wn, wm in _init_cases(orbx):
        n, m = oracle_lib.search_for_initialization(oracle, f1, f2, prev, window, ratio, ori)
        assert n == wn and (m == wm).all()


@pytest.mark.gpu
def test_initialization_hip(orbx):
    for f1, f2, prev, window, ratio, ori, wn, wm in _init_cases(orbx):
        n, m, _ = orbx.ORBmatcher(ratio, ori, max_features=len(f1["kps"])).SearchForInitialization(f1, f2, prev, window)
        assert n == wn and (m == wm).all()


@pytest.mark.gpu
def test_frustum_hip(orbx):
    from test_frustum import _setup
    z = _z("frustum")
    for seed, cosl in z["cases"]:
        seed = int(seed)
        T, Ts, sk, P = _setup(orbx, seed)
        g = lambda k: z["fr_%d_%s" % (seed, k)]
        got = orbx.ORBmatcher(0.8, True, max_features=len(P)).isInFrustum(
            T, (500.0, 500.0, 320.0, 240.0, 40.0), (0.0, 640.0, 0.0, 480.0), float(g("lsf")), 8,
            dict(pos=P, normal=g("normal"), max_distance=g("max_distance"), min_distance=g("min_distance")), float(cosl))
        ok = g("in_view") > 0
        assert (got["in_view"] == g("in_view")).all() and ok.sum() > 300
        for k in ("proj_x", "proj_y", "proj_xr", "view_cos"):
            assert (got[k][ok].view(np.uint32) == g(k)[ok].view(np.uint32)).all(), k
        assert (got["level"][ok] == g("level")[ok]).all()
