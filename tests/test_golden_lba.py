"""Golden vectors of the optimizer rows, produced by the COMPILED REFERENCE (src/Optimizer.cc + src/Converter.cc + vendored g2o,
unmodified, on oracle/eigenshim; tools/gen_golden_lba.py).  The CPU restatement (CPU tests) must reproduce them to 1e-9 (FP64 state,
per-edge chi2) / to the last float (what the reference writes back); the HIP path (gpu tests, through the C ABI and through the
drop-in shim on a real Map) to BASELINE.json's 1e-5 on poses, landmarks AND residuals.  The fixtures travel to the GPU box;
/root/reference does not."""
import ctypes
import importlib
import json
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle_lib

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
GOLDEN = Path(__file__).resolve().parent / "golden" / "lba"
G2O = sorted(GOLDEN.glob("g2o_*.npz"))
LBA_MAP = sorted(GOLDEN.glob("lba_map_*.npz"))
GBA_MAP = sorted(GOLDEN.glob("gba_map_*.npz"))
POSE = sorted(GOLDEN.glob("pose_*.npz"))
TOL = 1e-5


def _gen():
    return importlib.import_module("gen_golden_lba")


def test_golden_present():
    assert len(G2O) >= 8 and len(LBA_MAP) >= 3 and len(GBA_MAP) >= 2 and len(POSE) >= 7


def _window(orbx, z):
    w = orbx.lba_synth.make_window(**json.loads(str(z["cfg"])))
    assert _gen().window_crc(w) == z["crc"], "lba_synth no longer regenerates the inputs these goldens were made from"
    return w


def _frame(z):
    from test_pose_optimization import make_frame
    return make_frame(int(z["seed"]), n=int(z["n"]), stereo_frac=float(z["stereo_frac"]))


def _pose12(p16):
    p = np.asarray(p16, np.float64).reshape(-1, 4, 4)
    return np.concatenate([p[:, :3, :3].reshape(-1, 9), p[:, :3, 3]], 1)


def _ulp(got32, want32):
    got32, want32 = np.ascontiguousarray(got32, np.float32), np.ascontiguousarray(want32, np.float32)
    return (np.abs(got32.astype(np.float64) - want32.astype(np.float64)) / np.maximum(np.spacing(np.abs(want32)).astype(np.float64), 1e-45)).max()


# ------------------------------------------------------------------------------------------------ CPU: restatement == reference
@pytest.mark.parametrize("path", G2O, ids=lambda p: p.stem)
def test_restatement_matches_g2o_golden(orbx, oracle, path):
    z = np.load(path)
    w = _window(orbx, z)
    r = oracle_lib.ba_f64(oracle, w, int(z["sched"][0]), bool(z["sched"][1]), bool(z["sched"][2]))
    assert (r["iters"] == z["iters"]).all()
    assert np.abs(r["poses"] - z["poses"]).max() <= 1e-9 and np.abs(r["points"] - z["points"]).max() <= 1e-9
    assert (np.abs(r["chi2"] - z["chi2"]) / np.maximum(1.0, z["chi2"])).max() <= 1e-9
    assert (r["outlier"] == z["outlier"]).all()


@pytest.mark.parametrize("path", LBA_MAP, ids=lambda p: p.stem)
def test_restatement_matches_local_ba_golden(orbx, oracle, path):
    z = np.load(path)
    w = _window(orbx, z)
    role, prob, kf_list, pt_list, sel = oracle_lib.local_window_of(w, int(z["ref_kf"]))
    assert (role == z["role"]).all()
    r = oracle_lib.local_bundle_adjustment(oracle, prob)
    assert _ulp(r["poses"], z["poses"][kf_list]) <= 1 and _ulp(r["points"], z["points"][pt_list]) <= 1
    assert not (r["outlier"].astype(bool) & ~z["erased"][sel].astype(bool)).any()


@pytest.mark.parametrize("path", GBA_MAP, ids=lambda p: p.stem)
def test_restatement_matches_global_ba_golden(orbx, oracle, path):
    z = np.load(path)
    w = _window(orbx, z)
    r = oracle_lib.bundle_adjustment(oracle, oracle_lib.global_problem_of(w), int(z["iters"]), bool(z["robust"]))
    seen = np.zeros(w["P"], bool); seen[w["edge_point"]] = True
    assert _ulp(r["poses"], z["poses"]) <= 1 and _ulp(r["points"][seen], z["points"][seen]) <= 1


@pytest.mark.parametrize("path", POSE, ids=lambda p: p.stem)
def test_restatement_matches_pose_optimization_golden(oracle, path):
    z = np.load(path)
    fr = _frame(z)
    octv = oracle_lib.octaves_of(fr["inv_sigma2"])
    fr["inv_sigma2"] = (np.float32(1.0) / (oracle_lib.SCALE_FACTORS[octv] * oracle_lib.SCALE_FACTORS[octv])).astype(np.float32)
    assert _gen().frame_crc(fr) == z["crc"]
    r = oracle_lib.pose_optimization(oracle, fr)
    assert r["inliers"] == int(z["inliers"]) and (r["outlier"] == z["outlier"]).all()
    assert _ulp(r["pose"], z["pose"]) <= 1


# ------------------------------------------------------------------------------------------------ GPU: HIP == reference, 1e-5
def _check_against_g2o(got, z, w):
    assert np.abs(_pose12(got["poses"]) - z["poses"]).max() <= TOL
    assert np.abs(got["points"].astype(np.float64) - z["points"]).max() <= TOL
    assert (got["stats"][[0, 4]].astype(np.int64) == z["iters"]).all(), (got["stats"], z["iters"])       # same LM path
    dchi = np.abs(got["chi2"] - z["chi2"]) / np.maximum(1.0, z["chi2"])
    assert dchi.max() <= TOL, dchi.max()                                                                     # residuals within 1e-5
    diff = got["outlier"] != z["outlier"]                                                                    # flags: only edges sitting on the threshold may flip
    th = np.where(w["edge_obs"][:, 2] < 0, 5.991, 7.815)
    assert (np.abs(z["chi2"][diff] - th[diff]) <= TOL * th[diff]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", G2O, ids=lambda p: p.stem)
def test_hip_matches_g2o_golden(orbx, path):
    z = np.load(path)
    w = _window(orbx, z)
    opt = orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000)
    it1, robust, second = int(z["sched"][0]), bool(z["sched"][1]), bool(z["sched"][2])
    got = opt.LocalBundleAdjustment(w) if second else opt.BundleAdjustment(w, it1, robust)
    _check_against_g2o(got, z, w)
    opt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", LBA_MAP, ids=lambda p: p.stem)
def test_hip_dropin_matches_local_ba_golden(orbx, path):
    """shim/Optimizer_hip.cc::LocalBundleAdjustment on a real Map vs the reference's own LocalBundleAdjustment on the same Map."""
    orbx.load_library()
    hip = oracle_lib.slam_hip_lib()
    z = np.load(path)
    w = _window(orbx, z)
    got = oracle_lib.ref_local_ba_on_map(w, int(z["ref_kf"]), lib=hip)
    assert (got["role"] == z["role"]).all()
    assert np.abs(got["poses"].astype(np.float64) - z["poses"]).max() <= TOL
    assert np.abs(got["points"].astype(np.float64) - z["points"]).max() <= TOL
    # the erased observations: identical except for edges whose chi2 rides the threshold, and what MapPoint::EraseObservation cascades from
    # them (a point left with <= 2 observations is set bad and drops ALL of its observations, src/MapPoint.cc:168-198).  The chi2 of every
    # window edge comes from the restatement, which is pinned to the reference's g2o to 1e-12 (tests/test_optimizer_ref.py).
    diff = got["erased"] != z["erased"]
    if diff.any():
        role, prob, kf_list, pt_list, sel = oracle_lib.local_window_of(w, int(z["ref_kf"]))
        r = oracle_lib.local_bundle_adjustment(oracle_lib.Oracle(), prob)
        chi2 = np.full(w["E"], np.nan)
        chi2[np.flatnonzero(sel)] = r["chi2"]
        th = np.where(w["edge_obs"][:, 2] < 0, 5.991, 7.815)
        rides = np.abs(chi2 - th) <= TOL * th                                        # (NaN outside the window: False)
        pts_riding = np.unique(w["edge_point"][diff & rides])
        explained = rides | np.isin(w["edge_point"], pts_riding)
        assert (explained[diff]).all(), (int(diff.sum()), int((diff & ~explained).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("path", GBA_MAP, ids=lambda p: p.stem)
def test_hip_dropin_matches_global_ba_golden(orbx, path):
    orbx.load_library()
    hip = oracle_lib.slam_hip_lib()
    z = np.load(path)
    w = _window(orbx, z)
    got = oracle_lib.ref_global_ba_on_map(w, int(z["iters"]), bool(z["robust"]), int(z["loop_kf"]), lib=hip)
    seen = np.zeros(w["P"], bool); seen[w["edge_point"]] = True
    assert got["untouched"] == 1
    assert np.abs(got["poses"].astype(np.float64) - z["poses"]).max() <= TOL
    assert np.abs(got["points"][seen].astype(np.float64) - z["points"][seen]).max() <= TOL


@pytest.mark.gpu
def test_hip_matches_pose_optimization_golden(orbx):
    zs = [np.load(p) for p in POSE]
    frames = [_frame(z) for z in zs]
    for fr in frames:
        octv = oracle_lib.octaves_of(fr["inv_sigma2"])
        fr["inv_sigma2"] = (np.float32(1.0) / (oracle_lib.SCALE_FACTORS[octv] * oracle_lib.SCALE_FACTORS[octv])).astype(np.float32)
    opt = orbx.PoseOptimizer(max_frames=8, max_features=4096)
    got = opt.PoseOptimization(frames)
    for i, z in enumerate(zs):
        assert np.abs(got[i]["pose"].astype(np.float64) - z["pose"]).max() <= TOL, i
        assert got[i]["inliers"] == int(z["inliers"]) and (got[i]["outlier"] == z["outlier"]).all(), i
    # one frame per call: every register variant of the kernel (2 / 4 / 8 / 16 correspondences per thread) against the reference
    for i, z in enumerate(zs):
        g1 = opt.PoseOptimization([frames[i]])[0]
        assert np.abs(g1["pose"].astype(np.float64) - z["pose"]).max() <= TOL, i
        assert g1["inliers"] == int(z["inliers"]) and (g1["outlier"] == z["outlier"]).all(), i
    opt.close()
    # and through the drop-in shim on a real Frame
    hip = oracle_lib.slam_hip_lib()
    for z in zs[:3]:
        r = oracle_lib.ref_pose_optimization_on_frame(_frame(z), lib=hip)
        assert r["inliers"] == int(z["inliers"]) and (r["outlier"] == z["outlier"]).all()
        assert np.abs(r["pose"].astype(np.float64) - z["pose"]).max() <= TOL
