#!/usr/bin/env python3
"""High-contention sweep of ORBmatcher::SearchByProjection(Frame, MapPoints, th): many map points per feature,
few distinct descriptors, half of the points without observations - the parallel replay (k_proj_greedy) against the
CPU restatement.  GPU only; `python tools/stress_projection.py [cases]`."""
import importlib, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib
import test_projection as tp
orbx = importlib.import_module("self_commit_orb-slam2_amd")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
orc = oracle_lib.Oracle()
mt = orbx.ORBmatcher(0.8, True, max_features=4096)
bad = 0
t0 = time.time()
for c in range(cases):
    rng = np.random.default_rng(1000 + c)
    n = int(rng.choice([40, 150, 300, 900, 2500]))
    m = int(rng.choice([10, 500, 3000, 6000]))
    crowded = bool(rng.integers(0, 2))
    th = float(rng.choice([1.0, 3.0, 8.0, 15.0]))
    ratio = float(rng.choice([0.6, 0.8, 0.9]))
    fr, pts = tp.make_case(2000 + c, n=n, m=m, crowded=crowded)
    pts["has_obs"] = (rng.random(m) < rng.choice([0.0, 0.5, 1.0])).astype(np.uint8)
    fr["occupied"] = (rng.random(n) < rng.choice([0.0, 0.1, 0.6])).astype(np.uint8)
    want_n, want = oracle_lib.search_by_projection(orc, fr, pts, th, ratio)
    mt2 = mt if ratio == 0.8 else orbx.ORBmatcher(ratio, True, max_features=4096)
    got_n, got = mt2.SearchByProjection(dict(fr, kps=tp._struct_kps(orbx, fr["kps7"])), pts, th)
    if got_n != want_n or not (got == want).all():
        bad += 1
        print("MISMATCH case", c, n, m, crowded, th, ratio, got_n, want_n, int((got != want).sum()))
    if mt2 is not mt:
        mt2.close()
# ---- SearchByProjection(CurrentFrame, LastFrame, th, bMono): k_proj_last_greedy ----
for c in range(cases):
    rng = np.random.default_rng(5000 + c)
    n = int(rng.choice([60, 300, 1500]))
    nl = int(rng.choice([20, 800, 1400, 1900]))
    crowded = bool(rng.integers(0, 2))
    th = float(rng.choice([7.0, 15.0, 30.0]))
    mono, ori = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    fr, last = tp.make_last_case(3000 + c, float(rng.choice([0.0, 0.5, -0.5])), n=n, nl=nl, crowded=crowded)
    last["has_obs"] = (rng.random(nl) < rng.choice([0.0, 0.5, 1.0])).astype(np.uint8)
    fr["occupied"] = (rng.random(n) < rng.choice([0.0, 0.1, 0.6])).astype(np.uint8)
    want_n, want = oracle_lib.search_by_projection_last(orc, fr, last, th, mono, ori)
    mtl = orbx.ORBmatcher(0.9, bool(ori), max_features=2048)
    got_n, got = mtl.SearchByProjectionLast(dict(fr, kps=tp._struct_kps(orbx, fr["kps7"])),
                                            dict(last, kps=tp._struct_kps(orbx, last["kps7"]), valid=(last["valid"] == 1).astype(np.uint8)), th, mono)
    if got_n != want_n or not (got == want).all():
        bad += 1
        print("MISMATCH last case", c, n, nl, crowded, th, mono, ori, got_n, want_n, int((got != want).sum()))
    mtl.close()
print("stress_projection: 2 x %d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
