#!/usr/bin/env python3
"""Golden vectors for the widened rows, produced by the COMPILED REFERENCE (oracle/_ref/liborbslam.so: the unmodified
src/ORBmatcher.cc / Frame.cc / KeyFrame.cc / MapPoint.cc on oracle/cvshim, driven through real objects by
oracle/refslam_wrap.cc).  Inputs are regenerated from seeds by the generators in tests/ (numpy Generator streams), only the
reference's outputs (and the few reference-computed inputs of the device functions) are stored.

    make -f oracle/Makefile all && python tools/gen_golden_widen.py      ->  tests/golden/slam/widen_*.npz
"""
import importlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib  # noqa: E402

orbx = importlib.import_module("self_commit_orb-slam2_amd")

TRI = [(101, False, 0.0, False, True), (102, True, 0.0, False, True), (103, True, 0.5, True, False)]
FUSE = [(1, 111), (2, 112)]
AREA = [(3, 121), (5, 122)]
INIT = [(131, 10, 0.9, True), (132, 50, 0.8, False)]
FRUSTUM = [(141, 0.5), (142, 0.9)]


def main():
    assert oracle_lib.slam_lib() is not None, "build oracle/_ref first (needs /root/reference)"
    from test_triangulation import _scene as tri_scene
    from test_fuse import _scene as fuse_scene, _sim3
    from test_area_search import _setup as area_setup
    from test_search_init import _frames
    from test_frustum import _setup as frustum_setup
    out = ROOT / "tests" / "golden" / "slam"
    data = {}
    for seed, forward, stereo, only, ori in TRI:
        kf1, kf2, T1, T2, F12 = tri_scene(orbx, seed, forward=forward, stereo_frac=stereo)
        n, m, epi = oracle_lib.ref_search_for_triangulation(kf1, kf2, T1, T2, F12, only, ori)
        data["tri_%d_n" % seed], data["tri_%d_m" % seed], data["tri_%d_epi" % seed] = np.int32(n), m, epi
    np.savez_compressed(out / "widen_triangulation.npz", cases=np.array(TRI, np.float64), **data)
    print("triangulation", [int(data["tri_%d_n" % c[0]]) for c in TRI])

    data = {}
    for overload, seed in FUSE:
        kf, Tt, sk, Ts, P, cdesc, rng = fuse_scene(orbx, seed)
        r = oracle_lib.ref_fuse(overload, kf, Tt, _sim3(Tt), np.full(len(kf["kps"]), -1), [], sk, Ts, P, cdesc, np.full(len(P), 4, np.int32), [], 3.0, True)
        pts = r["points"]
        data["fuse_%d_probe" % seed] = r["probe_idx"]
        for k in ("u", "v", "ur", "level", "radius", "active"):
            data["fuse_%d_%s" % (seed, k)] = pts[k]
    np.savez_compressed(out / "widen_fuse.npz", cases=np.array(FUSE, np.int64), **data)
    print("fuse", [int((data["fuse_%d_probe" % s] >= 0).sum()) for _, s in FUSE])

    data = {}
    for overload, seed in AREA:
        kf, Tt, sk, Ts, P, cdesc, holder, lst, rng = area_setup(orbx, seed, 3 if overload == 3 else 4)
        r = oracle_lib.ref_fuse(overload, kf, Tt, _sim3(Tt), holder, np.ones(150, np.int32), sk, Ts, P, cdesc, np.full(len(P), 1, np.int32), lst, 8.0, False)
        pts = r["points"]
        data["area_%d_n" % seed], data["area_%d_holder" % seed] = np.int32(r["nfused"]), r["holder"]
        for k in ("u", "v", "level", "radius", "active"):
            data["area_%d_%s" % (seed, k)] = pts[k]
    np.savez_compressed(out / "widen_area_search.npz", cases=np.array(AREA, np.int64), **data)
    print("area", [int(data["area_%d_n" % s]) for _, s in AREA])

    data = {}
    for seed, window, ratio, ori in INIT:
        f1, f2, prev = _frames(orbx, seed)
        n, m, p2 = oracle_lib.ref_search_for_initialization(f1, f2, prev, window, ratio, ori)
        data["init_%d_n" % seed], data["init_%d_m" % seed] = np.int32(n), m
    np.savez_compressed(out / "widen_initialization.npz", cases=np.array(INIT, np.float64), **data)
    print("init", [int(data["init_%d_n" % c[0]]) for c in INIT])

    data = {}
    for seed, cosl in FRUSTUM:
        T, Ts, sk, P = frustum_setup(orbx, seed)
        r = oracle_lib.ref_is_in_frustum(T, Ts, sk, P, cosl)
        for k in ("in_view", "proj_x", "proj_y", "proj_xr", "level", "view_cos", "normal", "max_distance", "min_distance"):
            data["fr_%d_%s" % (seed, k)] = r[k]
        data["fr_%d_lsf" % seed] = np.float32(r["log_scale_factor"])
    np.savez_compressed(out / "widen_frustum.npz", cases=np.array(FRUSTUM, np.float64), **data)
    print("frustum", [int(data["fr_%d_in_view" % c[0]].sum()) for c in FRUSTUM])


if __name__ == "__main__":
    main()
