#!/usr/bin/env python3
"""Randomised HIP-vs-restatement sweep over the matcher entry points (GPU box): sizes, group layouts, masks and
thresholds drawn at random; every result must be index-exact.  python tools/stress_matchers.py [cases]"""
import importlib
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
orbx = importlib.import_module("self_commit_orb-slam2_amd")
import oracle_lib  # noqa: E402
from test_matcher import _kps, _noisy_pair, _rand_desc  # noqa: E402
from test_fuse import SF, _scene  # noqa: E402
from test_triangulation import SIGMA2  # noqa: E402
from test_triangulation import _scene as tri_scene  # noqa: E402
from test_search_init import _frames  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    orc = oracle_lib.Oracle()
    rng = np.random.default_rng(2026)
    t0 = time.time()
    bad = 0
    for it in range(cases):
        dbg = lambda *a: (print(*a, flush=True) if len(sys.argv) > 2 else None)
        n = int(rng.choice([1, 2, 63, 64, 65, 300, 1000, 2047, 3000]))
        # SearchByBoW
        kA, dA, kB, dB = _noisy_pair(rng, n, orbx)
        if rng.random() < 0.3:      # heavy ties
            base = _rand_desc(rng, int(rng.integers(1, 9)))
            dA, dB = base[rng.integers(0, len(base), len(dA))], base[rng.integers(0, len(base), len(dB))]
        use_groups = rng.random() < 0.5
        gA = (rng.integers(-1, 20, len(kA)).astype(np.int32) * 3) if use_groups else None
        gB = (rng.integers(-1, 20, len(kB)).astype(np.int32) * 3) if use_groups else None
        vA = (rng.random(len(kA)) < 0.8).astype(np.uint8) if rng.random() < 0.5 else None
        vB = (rng.random(len(kB)) < 0.8).astype(np.uint8) if rng.random() < 0.5 else None
        mode = int(rng.integers(0, 2))
        ratio, ori = float(rng.choice([0.6, 0.7, 0.75, 0.9, 0.95])), bool(rng.integers(0, 2))
        dbg('bow', it, n, mode, use_groups)
        want = oracle_lib.search_by_bow(orc, mode, kA, dA, kB, dB, ratio, ori, gA, gB, vA, vB if mode == 1 else None)
        mt = orbx.ORBmatcher(ratio, ori, max_features=max(len(kA), len(kB), 64))
        got = mt.SearchByBoW(kA, dA, kB, dB, gA, gB, vA, vB if mode == 1 else None, mode=mode)
        if got[0] != want[0] or not (got[1] == want[1]).all():
            bad += 1; print("SearchByBoW mismatch", it, n, mode, ratio, ori, use_groups)
        # SearchForTriangulation
        m = int(rng.choice([64, 300, 900, 2000]))
        dbg('tri', it, m)
        kf1, kf2, T1, T2, F12 = tri_scene(orbx, 1000 + it, n=m, forward=bool(rng.integers(0, 2)), stereo_frac=float(rng.choice([0.0, 0.5])))
        epi = np.array([rng.uniform(0, 640), rng.uniform(0, 480)], np.float32)
        only = bool(rng.integers(0, 2))
        want = oracle_lib.search_for_triangulation(orc, kf1, kf2, F12, epi, SF, SIGMA2, only, ori)
        got = orbx.ORBmatcher(0.6, ori, max_features=m).SearchForTriangulation(kf1, kf2, F12, epi, SF, SIGMA2, only)
        if got[0] != want[0] or not (got[1] == want[1]).all():
            bad += 1; print("SearchForTriangulation mismatch", it, m)
        # Fuse search + greedy area search on the same scene
        nc = int(rng.choice([1, 65, 700, 2500]))
        dbg('fuse', it, nc)
        kf, Tt, sk, Ts, P, cdesc, r2 = _scene(orbx, 2000 + it, nc=nc, nextra=max(nc // 2, 5))
        Pc = P @ Tt[:3, :3].T + Tt[:3, 3]
        z = np.where(np.abs(Pc[:, 2]) < 1e-3, 1e-3, Pc[:, 2])
        lvl = rng.integers(0, 8, nc).astype(np.int32)
        pts = dict(u=(Pc[:, 0] / z * 500 + 320).astype(np.float32), v=(Pc[:, 1] / z * 500 + 240).astype(np.float32),
                   ur=(Pc[:, 0] / z * 500 + 320 - 40.0 / z).astype(np.float32), level=lvl, radius=(float(rng.choice([3.0, 6.0, 20.0])) * SF[lvl]).astype(np.float32),
                   active=(rng.random(nc) < 0.9).astype(np.uint8), desc=cdesc)
        chi2 = bool(rng.integers(0, 2))
        mt = orbx.ORBmatcher(0.8, True, max_features=max(len(kf["kps"]), nc, 64))
        wi, wd = oracle_lib.fuse_best(orc, kf, pts, chi2)
        gi, gd = mt.FuseSearch(kf, pts, chi2)
        if not ((gi == wi).all() and (gd == wd).all()):
            bad += 1; print("FuseSearch mismatch", it, nc)
        frame = dict(kps=kf["kps"], desc=kf["desc"], blocked=(rng.random(len(kf["kps"])) < 0.3).astype(np.uint8), width=640, height=480)
        q = dict(u=pts["u"], v=pts["v"], radius=pts["radius"], min_level=lvl - 1, max_level=np.where(rng.random(nc) < 0.3, -1, lvl + 1).astype(np.int32),
                 active=pts["active"], desc=cdesc, window_int_bounds=bool(rng.integers(0, 2)))
        md = int(rng.choice([50, 100, 255]))
        dbg('area', it, nc, md)
        want = oracle_lib.area_search_greedy(orc, frame, q, md)
        got = mt.AreaSearchGreedy(frame, q, md)
        if not (got[0] == want[0] and (got[1] == want[1]).all() and (got[2] == want[2]).all()):
            bad += 1; print("AreaSearchGreedy mismatch", it, nc)
        # SearchForInitialization
        dbg('init', it)
        f1, f2, prev = _frames(orbx, 3000 + it, n=int(rng.choice([50, 700, 2500])))
        win = int(rng.choice([5, 10, 50, 200]))
        want = oracle_lib.search_for_initialization(orc, f1, f2, prev, win, ratio, ori)
        got = orbx.ORBmatcher(ratio, ori, max_features=len(f1["kps"])).SearchForInitialization(f1, f2, prev, win)
        if got[0] != want[0] or not (got[1] == want[1]).all():
            bad += 1; print("SearchForInitialization mismatch", it)
    print("stress: %d cases x 5 entry points, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
