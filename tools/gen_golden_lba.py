#!/usr/bin/env python3
"""Generate tests/golden/lba/*.npz from the COMPILED REFERENCE optimizer: src/Optimizer.cc + src/Converter.cc + the vendored g2o,
unmodified, on oracle/eigenshim (oracle/_ref/liborbslam.so).

    make -f oracle/Makefile all && python tools/gen_golden_lba.py

lba_map_*:  Optimizer::LocalBundleAdjustment on a real Map (KeyFrames / MapPoints / covisibility graph): float32 poses / points written
            back by the reference, erased observations, keyframe roles.
gba_map_*:  Optimizer::GlobalBundleAdjustemnt on a real Map.
pose_*:     Optimizer::PoseOptimization on a real Frame: pose, outlier flags, return value.
g2o_*:      the vendored g2o driven directly on the flattened window: FP64 poses (R, t) / points, per-edge chi2, outlier flags.
Inputs are regenerated from the seeds by self_commit_orb-slam2_amd/lba_synth.py; every file carries a checksum of them.
"""
import importlib
import json
import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib  # noqa: E402

orbx = importlib.import_module("self_commit_orb-slam2_amd")
OUT = ROOT / "tests" / "golden" / "lba"

LBA_MAP = [("lba_map_mono_16kf", dict(K=16, P=1200, seed=5, max_obs=5, n_fixed=0)),
           ("lba_map_mixed_16kf", dict(K=16, P=1200, seed=6, max_obs=5, n_fixed=0, stereo_frac=0.4)),
           ("lba_map_config5_50kf", dict(K=50, P=5000, seed=12345, n_fixed=0))]
GBA_MAP = [("gba_map_robust", dict(K=14, P=900, seed=8, max_obs=6, n_fixed=0), 20, True, 0),
           ("gba_map_loop", dict(K=14, P=900, seed=9, max_obs=6, n_fixed=0, stereo_frac=0.4), 10, False, 7)]
G2O = [("g2o_config5_50kf", dict(K=50, P=5000, seed=12345), (5, True, True)),
       ("g2o_mixed_20kf", dict(K=20, P=1500, seed=7, stereo_frac=0.5), (5, True, True)),
       ("g2o_stereo_8kf", dict(K=8, P=200, seed=2, stereo_frac=1.0, n_fixed=1), (5, True, True)),
       ("g2o_ba_config5_50kf", dict(K=50, P=5000, seed=12345, n_fixed=1), (10, True, False)),
       ("g2o_ba_mixed_20kf", dict(K=20, P=1500, seed=7, stereo_frac=0.5, n_fixed=1), (20, False, False)),
       ("g2o_ba_stereo_8kf", dict(K=8, P=200, seed=2, stereo_frac=1.0, n_fixed=1), (20, True, False)),
       # badly initialised windows: Levenberg trials are rejected and retried (pop() / restore path)
       ("g2o_rejected_12kf_a", dict(K=12, P=400, seed=1, n_fixed=2, pose_noise=[0.10471975511965977, 0.25], point_noise=0.25, stereo_frac=0.3), (5, True, True)),
       ("g2o_rejected_12kf_b", dict(K=12, P=400, seed=6, n_fixed=2, pose_noise=[0.20943951023931953, 0.5], point_noise=0.4, stereo_frac=0.3), (5, True, True))]
POSE = [("pose_%d" % i, 10 + i, n, st) for i, (n, st) in enumerate([(600, 0.5), (1500, 0.0), (40, 1.0), (8, 0.5), (2, 0.5), (300, 0.3), (3500, 0.4)])]


def window_crc(w):
    c = 0
    for k in ("poses", "points", "intr", "edge_point", "edge_kf", "edge_obs", "edge_inv_sigma2", "fixed"):
        c = zlib.crc32(np.ascontiguousarray(w[k]).tobytes(), c)
    return np.uint32(c)


def frame_crc(fr):
    c = 0
    for k in ("pose", "Xw", "obs", "inv_sigma2"):
        c = zlib.crc32(np.ascontiguousarray(fr[k]).tobytes(), c)
    return np.uint32(c)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    only = sys.argv[sys.argv.index("--only") + 1:] if "--only" in sys.argv else None      # regenerate just these files (npz archives carry timestamps)
    global LBA_MAP, GBA_MAP, G2O, POSE
    if only:
        LBA_MAP = [e for e in LBA_MAP if e[0] in only]; GBA_MAP = [e for e in GBA_MAP if e[0] in only]
        G2O = [e for e in G2O if e[0] in only]; POSE = [e for e in POSE if e[0] in only]
    for name, cfg in LBA_MAP:
        w = orbx.lba_synth.make_window(**cfg)
        r = oracle_lib.ref_local_ba_on_map(w, w["K"] - 1)
        np.savez_compressed(OUT / (name + ".npz"), cfg=json.dumps(cfg), crc=window_crc(w), ref_kf=w["K"] - 1, **r)
        print(name, "erased", int(r["erased"].sum()))
    for name, cfg, iters, robust, loop_kf in GBA_MAP:
        w = orbx.lba_synth.make_window(**cfg)
        r = oracle_lib.ref_global_ba_on_map(w, iters, robust, loop_kf)
        np.savez_compressed(OUT / (name + ".npz"), cfg=json.dumps(cfg), crc=window_crc(w), iters=iters, robust=int(robust), loop_kf=loop_kf, poses=r["poses"], points=r["points"])
        print(name)
    for name, cfg, sched in G2O:
        w = orbx.lba_synth.make_window(**cfg)
        r = oracle_lib.g2o_ba_f64(w, *sched)
        np.savez_compressed(OUT / (name + ".npz"), cfg=json.dumps(cfg), crc=window_crc(w), sched=np.array([int(s) for s in sched]), **r)
        print(name, "iters", r["iters"], "outliers", int(r["outlier"].sum()))
    from test_pose_optimization import make_frame
    for name, seed, n, st in POSE:
        fr = make_frame(seed, n=n, stereo_frac=st)
        r = oracle_lib.ref_pose_optimization_on_frame(fr)
        np.savez_compressed(OUT / (name + ".npz"), seed=seed, n=n, stereo_frac=st, crc=frame_crc(fr), pose=r["pose"], outlier=r["outlier"], inliers=r["inliers"])
        print(name, "inliers", r["inliers"])


if __name__ == "__main__":
    main()
