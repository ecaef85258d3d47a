"""The optimizer rows pinned to the reference: src/Optimizer.cc, src/Converter.cc and the vendored g2o are compiled UNMODIFIED
against oracle/eigenshim (an Eigen API stand-in, like oracle/cvshim for OpenCV) into oracle/_ref/liborbslam.so, and

 * Optimizer::LocalBundleAdjustment / GlobalBundleAdjustemnt / PoseOptimization run on real Map / KeyFrame / MapPoint / Frame
   objects: the CPU restatement (oracle/lba_oracle.cc) must reproduce their float32 results (what the reference writes back
   through Converter::toCvMat) - bit for bit on all but a handful of values that sit on a float rounding boundary;
 * the same g2o driven directly (orbslam_g2o_ba) exposes the FP64 state and e->chi2(): the restatement agrees to 1e-9.

CPU only (the reference sources exist in the build container; on the GPU box the prebuilt library travels with the snapshot)."""
import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.skipif(oracle_lib.slam_lib() is None or not hasattr(oracle_lib.slam_lib(), "orbslam_g2o_ba"),
                                reason="oracle/_ref/liborbslam.so (reference + g2o) not built")


def _ulp_report(got32, want32):
    """(fraction of bit-identical floats, max difference in units of the last place)"""
    got32, want32 = np.ascontiguousarray(got32, np.float32), np.ascontiguousarray(want32, np.float32)
    same = got32.view(np.uint32) == want32.view(np.uint32)
    ulp = np.abs(got32.astype(np.float64) - want32.astype(np.float64)) / np.maximum(np.spacing(np.abs(want32)).astype(np.float64), 1e-45)
    return same.mean(), ulp.max()


LBA_WINDOWS = [dict(K=16, P=1200, seed=5, max_obs=5, n_fixed=0), dict(K=16, P=1200, seed=6, max_obs=5, n_fixed=0, stereo_frac=0.4),
               dict(K=10, P=300, seed=21, max_obs=6, n_fixed=0, stereo_frac=1.0), dict(K=50, P=5000, seed=12345, n_fixed=0)]


@pytest.mark.parametrize("cfg", LBA_WINDOWS)
def test_restatement_equals_reference_local_bundle_adjustment(orbx, oracle, cfg):
    w = orbx.lba_synth.make_window(**cfg)
    ref_kf = w["K"] - 1
    got = oracle_lib.ref_local_ba_on_map(w, ref_kf)
    role, prob, kf_list, pt_list, sel = oracle_lib.local_window_of(w, ref_kf)
    assert (got["role"] == role).all()
    want = oracle_lib.local_bundle_adjustment(oracle, prob)
    same_p, ulp_p = _ulp_report(want["poses"], got["poses"][kf_list])
    same_x, ulp_x = _ulp_report(want["points"], got["points"][pt_list])
    assert same_p >= 0.995 and ulp_p <= 1 and same_x >= 0.995 and ulp_x <= 1, (same_p, ulp_p, same_x, ulp_x)
    assert np.abs(got["poses"][kf_list] - w["poses"][kf_list]).max() > 1e-3               # it did optimise
    untouched = np.setdiff1d(np.arange(w["K"]), kf_list)
    assert (got["poses"][untouched] == w["poses"][untouched]).all()
    # every outlier of the restatement was erased by the reference; extra erasures are points turned bad by
    # MapPoint::EraseObservation (<= 2 observations left, src/MapPoint.cc:176-215)
    er, out = got["erased"][sel].astype(bool), want["outlier"].astype(bool)
    assert not (out & ~er).any()
    assert np.isin(prob["edge_point"][er & ~out], np.unique(prob["edge_point"][out])).all()
    assert got["erased"][~sel].sum() == 0


@pytest.mark.parametrize("cfg,iters,robust,loop_kf", [(dict(K=14, P=900, seed=8, max_obs=6, n_fixed=0), 20, True, 0),
                                                      (dict(K=14, P=900, seed=9, max_obs=6, n_fixed=0, stereo_frac=0.4), 10, False, 7),
                                                      (dict(K=30, P=2500, seed=10, n_fixed=0, stereo_frac=0.2), 20, True, 0)])
def test_restatement_equals_reference_global_bundle_adjustment(orbx, oracle, cfg, iters, robust, loop_kf):
    w = orbx.lba_synth.make_window(**cfg)
    got = oracle_lib.ref_global_ba_on_map(w, iters, robust, loop_kf)
    assert got["untouched"] == 1
    want = oracle_lib.bundle_adjustment(oracle, oracle_lib.global_problem_of(w), iters, robust)
    seen = np.zeros(w["P"], bool); seen[w["edge_point"]] = True
    same_p, ulp_p = _ulp_report(want["poses"], got["poses"])
    same_x, ulp_x = _ulp_report(want["points"][seen], got["points"][seen])
    assert same_p >= 0.995 and ulp_p <= 1 and same_x >= 0.995 and ulp_x <= 1, (same_p, ulp_p, same_x, ulp_x)
    assert np.abs(got["poses"] - w["poses"]).max() > 1e-3


def test_restatement_equals_reference_pose_optimization(oracle):
    from test_pose_optimization import make_frame
    for i in range(8):
        fr = make_frame(10 + i, n=[600, 1500, 40, 8, 300, 700, 700, 1000][i], stereo_frac=[0.5, 0.0, 1.0, 0.5, 0.3, 0.5, 0.0, 1.0][i])
        got = oracle_lib.ref_pose_optimization_on_frame(fr)          # also puts the Frame's float32 information into fr
        want = oracle_lib.pose_optimization(oracle, fr)
        assert got["inliers"] == want["inliers"] and (got["outlier"] == want["outlier"]).all(), i
        same, ulp = _ulp_report(want["pose"], got["pose"])
        assert ulp <= 1 and same >= 0.9, (i, same, ulp)


@pytest.mark.parametrize("cfg,sched", [(dict(K=50, P=5000, seed=12345), (5, True, True)), (dict(K=20, P=1500, seed=7, stereo_frac=0.5), (5, True, True)),
                                       (dict(K=8, P=200, seed=2, stereo_frac=1.0, n_fixed=1), (5, True, True)),
                                       (dict(K=20, P=1500, seed=7, stereo_frac=0.5, n_fixed=1), (20, False, False)),
                                       (dict(K=12, P=400, seed=9, n_fixed=1), (20, True, False))])
def test_restatement_equals_g2o_in_double_precision(orbx, oracle, cfg, sched):
    """FP64 state and per-edge chi2 of the vendored g2o (LinearSolverEigen + BlockSolver_6_3 + Levenberg) vs the restatement: 1e-9."""
    w = orbx.lba_synth.make_window(**cfg)
    got = oracle_lib.g2o_ba_f64(w, *sched)
    want = oracle_lib.ba_f64(oracle, w, *sched)
    assert (got["iters"] == want["iters"]).all(), (got["iters"], want["iters"])
    assert np.abs(got["poses"] - want["poses"]).max() <= 1e-9
    assert np.abs(got["points"] - want["points"]).max() <= 1e-9
    assert (np.abs(got["chi2"] - want["chi2"]) / np.maximum(1.0, want["chi2"])).max() <= 1e-9
    assert (got["outlier"] == want["outlier"]).all()
