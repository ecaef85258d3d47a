"""Local bundle adjustment: the CPU oracle is pinned by (a) analytic Jacobians vs central
differences and (b) an independent scipy solve of a small window; the HIP path is compared
to the oracle to 1e-5 (poses, points, residual chi2), as BASELINE.json's north_star states."""
import ctypes

import numpy as np
import pytest

import oracle_lib

TOL = 1e-5


def _P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _edge_eval(orc, pose, intr, X, obs):
    err = np.zeros(3)
    A = np.zeros(9)
    B = np.zeros(18)
    orc.lib.lo_edge_eval(_P(pose), _P(intr), _P(X), _P(obs), _P(err), _P(A), _P(B))
    return err, A.reshape(3, 3), B.reshape(3, 6)


def _edge_perturbed(orc, pose, intr, X, obs, d6, dx3):
    err = np.zeros(3)
    orc.lib.lo_edge_error_perturbed(_P(pose), _P(intr), _P(X), _P(obs), _P(np.ascontiguousarray(d6)), _P(np.ascontiguousarray(dx3)), _P(err))
    return err


@pytest.mark.parametrize("stereo", [False, True])
def test_oracle_jacobians_match_central_differences(orbx, oracle, stereo):
    """g2o ships the numeric fallback (base_binary_edge.hpp:130-199); the analytic Jacobians
    restated from types_six_dof_expmap.cpp must agree with it."""
    w = orbx.lba_synth.make_window(K=6, P=40, seed=3, stereo_frac=1.0 if stereo else 0.0, n_fixed=1)
    rng = np.random.default_rng(0)
    for e in rng.choice(w["E"], 20, replace=False):
        pose = w["poses"][w["edge_kf"][e]].copy()
        intr = w["intr"][w["edge_kf"][e]].copy()
        X = w["points"][w["edge_point"][e]].astype(np.float64)
        obs = w["edge_obs"][e].copy()
        err, A, B = _edge_eval(oracle, pose, intr, X, obs)
        D = 3 if stereo else 2
        h = 1e-3 if stereo else 1e-6   # the stereo edge rounds 1/z to float32 (types_six_dof_expmap.cpp:151): needs a coarse step
        for j in range(3):
            dx = np.zeros(3); dx[j] = h
            num = (_edge_perturbed(oracle, pose, intr, X, obs, np.zeros(6), dx) - _edge_perturbed(oracle, pose, intr, X, obs, np.zeros(6), -dx)) / (2 * h)
            assert np.allclose(num[:D], A[:D, j], rtol=1e-3 if stereo else 1e-5, atol=5e-2 if stereo else 1e-4)
        for j in range(6):
            d = np.zeros(6); d[j] = h
            num = (_edge_perturbed(oracle, pose, intr, X, obs, d, np.zeros(3)) - _edge_perturbed(oracle, pose, intr, X, obs, -d, np.zeros(3))) / (2 * h)
            assert np.allclose(num[:D], B[:D, j], rtol=1e-3 if stereo else 1e-5, atol=2e-1 if stereo else 1e-3)


def test_oracle_reaches_the_scipy_optimum(orbx, oracle):
    """No outliers, no robust kernel effect: the LM restatement must land on the same
    least-squares optimum as an independent scipy solve of the same residuals."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation
    w = orbx.lba_synth.make_window(K=6, P=60, seed=5, outlier_frac=0.0, n_fixed=2, max_obs=6)
    r = oracle_lib.local_bundle_adjustment(oracle, w)
    assert r["outlier"].sum() <= 0.05 * w["E"]
    K, P = w["K"], w["P"]
    free = [k for k in range(K) if not w["fixed"][k]]
    keep = r["outlier"] == 0           # stage 2 of the reference only uses the inliers

    def unpack(x):
        poses = w["poses"].reshape(K, 4, 4).astype(np.float64).copy()
        for i, k in enumerate(free):
            rv, t = x[6 * i:6 * i + 3], x[6 * i + 3:6 * i + 6]
            poses[k, :3, :3] = Rotation.from_rotvec(rv).as_matrix()
            poses[k, :3, 3] = t
        pts = x[6 * len(free):].reshape(P, 3)
        return poses, pts

    def resid(x):
        poses, pts = unpack(x)
        k, l = w["edge_kf"][keep], w["edge_point"][keep]
        Xc = np.einsum("eij,ej->ei", poses[k, :3, :3], pts[l]) + poses[k, :3, 3]
        intr = w["intr"][k].astype(np.float64)
        u = intr[:, 0] * Xc[:, 0] / Xc[:, 2] + intr[:, 2]
        v = intr[:, 1] * Xc[:, 1] / Xc[:, 2] + intr[:, 3]
        s = np.sqrt(w["edge_inv_sigma2"][keep].astype(np.float64))
        return np.concatenate([(w["edge_obs"][keep, 0] - u) * s, (w["edge_obs"][keep, 1] - v) * s])

    x0 = []
    P0 = r["poses"].reshape(K, 4, 4).astype(np.float64)
    for k in free:
        x0 += list(Rotation.from_matrix(P0[k, :3, :3]).as_rotvec()) + list(P0[k, :3, 3])
    x0 = np.array(x0 + list(r["points"].astype(np.float64).ravel()))
    c0 = (resid(x0) ** 2).sum()
    sol = least_squares(resid, x0, method="trf", xtol=1e-12, ftol=1e-12, gtol=1e-12, max_nfev=200)
    c1 = (sol.fun ** 2).sum()
    # the oracle's 10 LM iterations are already at the optimum scipy converges to
    assert c1 <= c0 * (1 + 1e-9)
    assert (c0 - c1) / c0 < 2e-3, (c0, c1)
    # and chi2 reported by the oracle is the same objective
    assert abs(r["chi2"][keep].sum() - c0) / c0 < 1e-4


def test_oracle_stage_protocol(orbx, oracle):
    w = orbx.lba_synth.make_window(K=12, P=400, seed=9)
    r = oracle_lib.local_bundle_adjustment(oracle, w)
    s = r["stats"]
    assert 1 <= s[0] <= 5 and 1 <= s[4] <= 10          # optimize(5) then optimize(10), src/Optimizer.cc:863-917
    assert s[3] < s[2] and s[7] <= s[6]                # robust chi2 decreases in both stages
    gross = 0.05 * w["E"]
    assert 0.5 * gross < r["outlier"].sum() < 4 * gross
    # fixed keyframes come back unchanged, free ones moved towards the truth
    fx = w["fixed"].astype(bool)
    assert np.allclose(r["poses"][fx], w["poses"][fx], atol=1e-6)
    assert np.abs(r["poses"][~fx] - w["true_poses"][~fx]).max() < np.abs(w["poses"][~fx] - w["true_poses"][~fx]).max()
    # stop flag set before the call: nothing is optimised
    stop = np.array([1], np.uint8)
    r2 = oracle_lib.local_bundle_adjustment(oracle, w, stop=stop)
    assert np.allclose(r2["points"], w["points"]) and r2["stats"][0] == 0


def _compare(got, want, w):
    dp = np.abs(got["poses"].astype(np.float64) - want["poses"]).max()
    dx = np.abs(got["points"].astype(np.float64) - want["points"]).max()
    assert dp <= TOL, "pose delta %g" % dp
    assert dx <= TOL, "point delta %g" % dx
    assert (got["stats"][[0, 1, 4, 5]] == want["stats"][[0, 1, 4, 5]]).all(), (got["stats"], want["stats"])   # same LM path
    rel = np.abs(got["stats"][[2, 3, 6, 7]] - want["stats"][[2, 3, 6, 7]]) / np.maximum(want["stats"][[2, 3, 6, 7]], 1.0)
    assert rel.max() <= TOL, rel
    dchi = np.abs(got["chi2"] - want["chi2"]) / np.maximum(1.0, want["chi2"])
    assert dchi.max() <= TOL, dchi.max()         # residual chi2 per edge within 1e-5 (north_star)
    # outlier flags may only differ on edges sitting numerically on the threshold
    diff = got["outlier"] != want["outlier"]
    th = np.where(w["edge_obs"][:, 2] < 0, 5.991, 7.815)
    assert (np.abs(want["chi2"][diff] - th[diff]) <= TOL * th[diff]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(K=50, P=5000, seed=12345), dict(K=20, P=1500, seed=7, stereo_frac=0.5),
                                 dict(K=8, P=200, seed=2, stereo_frac=1.0, n_fixed=1)])
def test_lba_hip_matches_oracle(orbx, oracle, cfg):
    w = orbx.lba_synth.make_window(**cfg)
    want = oracle_lib.local_bundle_adjustment(oracle, w)
    opt = orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000)
    got = opt.LocalBundleAdjustment(w)
    _compare(got, want, w)
    ms, flops = opt.last_timing()
    assert ms > 0 and flops > 0
    # stop flag honoured before the start (src/Optimizer.cc:858-860)
    stop = np.array([1], np.uint8)
    g2 = opt.LocalBundleAdjustment(w, stop_flag=stop)
    assert g2["stats"][0] == 0 and np.allclose(g2["points"], w["points"])
    opt.close()


def test_oracle_bundle_adjustment_protocol(orbx, oracle):
    """Optimizer::BundleAdjustment = one optimize(nIterations), Huber kernels iff bRobust, no second stage
    (reference src/Optimizer.cc:86-360)."""
    w = orbx.lba_synth.make_window(K=12, P=400, seed=9, n_fixed=1)
    r = oracle_lib.bundle_adjustment(oracle, w, 20, True)
    s = r["stats"]
    assert 1 <= s[0] <= 20 and s[4] == 0 and s[5] == 0 and s[3] < s[2]
    r5 = oracle_lib.bundle_adjustment(oracle, w, 5, True)
    lba = oracle_lib.local_bundle_adjustment(oracle, w)
    # LBA's first stage is the same schedule up to the monocular Huber delta (sqrt(5.99) here, sqrt(5.991) there: src/Optimizer.cc:141 / :764)
    assert r5["stats"][0] == lba["stats"][0] and np.isclose(r5["stats"][3], lba["stats"][3], rtol=1e-3) and r5["stats"][3] != lba["stats"][3]
    nr = oracle_lib.bundle_adjustment(oracle, w, 10, False)
    assert nr["stats"][2] > r["stats"][2]            # without kernels the gross outliers count fully in chi2
    r0 = oracle_lib.bundle_adjustment(oracle, w, 0, True)
    assert np.allclose(r0["points"], w["points"])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(K=12, P=400, seed=1, n_fixed=2, pose_noise=(np.deg2rad(6.0), 0.25), point_noise=0.25, stereo_frac=0.3),
                                 dict(K=12, P=400, seed=6, n_fixed=2, pose_noise=(np.deg2rad(12.0), 0.5), point_noise=0.4, stereo_frac=0.3)])
def test_lba_with_rejected_trials(orbx, oracle, cfg):
    """Badly initialised windows: Levenberg trials are rejected (more trials than iterations), i.e. the pop() / retry path - and, in the
    HIP driver, the rebuild of H and b after a speculative linearisation on a state that was rejected."""
    w = orbx.lba_synth.make_window(**cfg)
    want = oracle_lib.local_bundle_adjustment(oracle, w)
    assert want["stats"][1] + want["stats"][5] > want["stats"][0] + want["stats"][4], want["stats"]
    opt = orbx.Optimizer(max_keyframes=16, max_points=512, max_edges=8192)
    _compare(opt.LocalBundleAdjustment(w), want, w)
    opt.close()


def _with_unused_vertices(w):
    """The same window with one more keyframe and two more landmarks that no edge references (rows of length 0 in the adjacency lists,
    vertices that initializeOptimization leaves out)."""
    w = dict(w)
    K, P = w["K"], w["P"]
    w["poses"] = np.concatenate([w["poses"], w["poses"][-1:]]).copy()
    w["intr"] = np.concatenate([w["intr"], w["intr"][-1:]]).copy()
    w["fixed"] = np.concatenate([w["fixed"], np.zeros(1, np.uint8)])
    w["points"] = np.concatenate([w["points"], np.array([[0.3, 0.1, 0.2], [9.0, 9.0, 9.0]], w["points"].dtype)]).copy()
    w["K"], w["P"] = K + 1, P + 2
    return w


@pytest.mark.gpu
def test_lba_edge_cases(orbx, oracle):
    """Shapes the device-side preparation (adjacency lists, index mapping) must get right: every keyframe fixed (structure-only, no
    reduced system), vertices without edges, fewer edges than one workgroup."""
    opt = orbx.Optimizer(max_keyframes=16, max_points=512, max_edges=8192)
    w = orbx.lba_synth.make_window(K=6, P=120, seed=21, n_fixed=6, stereo_frac=0.5)
    assert w["fixed"].all()
    _compare(opt.LocalBundleAdjustment(w), oracle_lib.local_bundle_adjustment(oracle, w), w)
    w = _with_unused_vertices(orbx.lba_synth.make_window(K=7, P=150, seed=22, n_fixed=1, stereo_frac=0.3))
    got, want = opt.LocalBundleAdjustment(w), oracle_lib.local_bundle_adjustment(oracle, w)
    _compare(got, want, w)
    assert np.allclose(got["poses"][-1].ravel(), w["poses"][-1].ravel(), atol=1e-6) and np.allclose(got["points"][-2:], w["points"][-2:])
    w = orbx.lba_synth.make_window(K=3, P=12, seed=23, n_fixed=1, max_obs=3)
    assert w["E"] < 64
    _compare(opt.LocalBundleAdjustment(w), oracle_lib.local_bundle_adjustment(oracle, w), w)
    opt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,iters,robust", [(dict(K=50, P=5000, seed=12345, n_fixed=1), 10, True), (dict(K=20, P=1500, seed=7, stereo_frac=0.5, n_fixed=1), 20, False),
                                              (dict(K=8, P=200, seed=2, stereo_frac=1.0, n_fixed=1), 20, True)])
def test_bundle_adjustment_hip_matches_oracle(orbx, oracle, cfg, iters, robust):
    w = orbx.lba_synth.make_window(**cfg)
    want = oracle_lib.bundle_adjustment(oracle, w, iters, robust)
    opt = orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000)
    got = opt.BundleAdjustment(w, iters, robust)
    _compare(got, want, w)
    opt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("K,iters", [(360, 3), (560, 2)])
def test_bundle_adjustment_beyond_the_lds_limits(orbx, oracle, K, iters):
    """GlobalBundleAdjustemnt of a larger map: more than 341 free keyframes (the back-substitution vector leaves LDS) and more than
    530 (the Schur block rows are accumulated in HBM).  The reference (g2o sparse) has no such limits; neither may the drop-in."""
    w = orbx.lba_synth.make_window(K=K, P=4000, seed=77, n_fixed=1, max_obs=12)
    assert (w["fixed"] == 0).sum() > (341 if K == 360 else 530)
    want = oracle_lib.bundle_adjustment(oracle, w, iters, True)
    opt = orbx.Optimizer(max_keyframes=K, max_points=4096, max_edges=w["E"] + 1)
    got = opt.BundleAdjustment(w, iters, True)
    _compare(got, want, w)
    opt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(K=50, P=5000, seed=12345), dict(K=12, P=400, seed=6, n_fixed=2, pose_noise=(np.deg2rad(12.0), 0.5), point_noise=0.4, stereo_frac=0.3)])
def test_fused_linearisation_equals_the_split_one(orbx, cfg, monkeypatch):
    """k_lin_sums (the Jacobian blocks computed inside the landmark / keyframe sums) against k_linearize + k_sum_points + k_sum_poses (every block written
    out per edge; ORBX_LBA_SPLIT=1, read when the handle is created).  H and b come out of the same expressions in the same order; the reduced system is
    summed with FP64 atomics in an order that varies from run to run, so the two solves are compared like two runs of one path: identical iteration and
    trial counts in both stages, identical outlier flags, estimates and residuals equal to 1e-10 - on a well conditioned window and on one with rejected
    trials."""
    w = orbx.lba_synth.make_window(**cfg)
    fused = orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000)
    monkeypatch.setenv("ORBX_LBA_SPLIT", "1")
    split = orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000)
    monkeypatch.delenv("ORBX_LBA_SPLIT")
    a, b = fused.LocalBundleAdjustment(w), split.LocalBundleAdjustment(w)
    sa, sb = np.asarray(a["stats"]), np.asarray(b["stats"])
    assert (sa[[0, 1, 4, 5]] == sb[[0, 1, 4, 5]]).all(), (sa, sb)
    assert (np.asarray(a["outlier"]) == np.asarray(b["outlier"])).all()
    for key in ("poses", "points", "chi2"):
        assert np.allclose(a[key], b[key], rtol=1e-10, atol=1e-10), key
    fused.close(); split.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,runs", [(dict(K=50, P=5000, seed=12345), 100),
                                      (dict(K=12, P=400, seed=6, n_fixed=2, pose_noise=(np.deg2rad(12.0), 0.5), point_noise=0.4, stereo_frac=0.3), 40)])
def test_lba_is_reproducible(orbx, cfg, runs):
    """The same window again and again, on one handle and on a fresh one: identical LM path (iterations, trials, chi2 to the last bit) and
    identical float32 outputs, per-edge chi2 and outlier flags.  The Schur complement is accumulated in 64-bit fixed point (integer sums do
    not depend on the order in which workgroups and waves arrive - the former FP64 atomics did) and its right-hand side in ordered partial
    sums (csrc/orbx_lba.hip: k_schur_rows / k_schur_fin); everything else was ordered before.  The second window goes through rejected trials."""
    w = orbx.lba_synth.make_window(**cfg)
    opt = orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000)
    first = opt.LocalBundleAdjustment(w)
    for run in range(1, runs):
        if run == runs // 2:      # a fresh handle (fresh allocations, another stream) must land on the same bits as well
            opt.close()
            opt = orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000)
        got = opt.LocalBundleAdjustment(w)
        assert (np.asarray(got["stats"]).view(np.uint64) == np.asarray(first["stats"]).view(np.uint64)).all(), (run, got["stats"], first["stats"])
        assert (got["poses"].view(np.uint32) == first["poses"].view(np.uint32)).all() and (got["points"].view(np.uint32) == first["points"].view(np.uint32)).all(), run
        assert (np.asarray(got["chi2"], np.float64).view(np.uint64) == np.asarray(first["chi2"], np.float64).view(np.uint64)).all() and (got["outlier"] == first["outlier"]).all(), run
    opt.close()


# ---- the stop flag raised DURING a solve (src/Optimizer.cc:858-871, optimization_algorithm_levenberg.cpp:98-146, sparse_optimizer.cpp:370) ----
STOP_WINDOWS = [dict(K=50, P=5000, seed=12345),
                dict(K=12, P=400, seed=6, n_fixed=2, pose_noise=(np.deg2rad(12.0), 0.5), point_noise=0.4, stereo_frac=0.3)]      # (the second one rejects trials)


@pytest.mark.parametrize("cfg", STOP_WINDOWS)
def test_oracle_stop_after_trial_k(orbx, oracle, cfg):
    """The oracle's test input: trials are counted over both stages; a flag raised after trial k ends the iteration and the stage there (the rejected
    trial's estimates restored), skips the second stage when it comes during the first (:869-871), still classifies and writes back; k at or beyond
    the natural number of trials changes nothing."""
    w = orbx.lba_synth.make_window(**cfg)
    free = oracle_lib.local_bundle_adjustment(oracle, w)
    t1, t2 = int(free["stats"][1]), int(free["stats"][5])
    assert t1 >= 2 and t2 >= 2
    prev = None
    for k in range(1, t1 + t2 + 2):
        r = oracle_lib.local_bundle_adjustment(oracle, w, stop_after_trials=k)
        s = r["stats"]
        if k <= t1:
            assert s[1] == k and s[4] == 0 and s[5] == 0, (k, s)          # stopped inside (or at the end of) stage one: stage two never starts
        elif k < t1 + t2:
            assert s[1] == t1 and s[5] == k - t1, (k, s)
        else:
            for key in ("poses", "points", "chi2", "outlier", "stats"):
                assert (np.asarray(r[key]) == np.asarray(free[key])).all(), (k, key)
        assert np.isfinite(r["poses"]).all() and r["chi2"].sum() > 0                 # the outlier pass and the write-back ran
        if prev is not None and k <= t1 + t2:
            assert not (np.array_equal(prev["points"], r["points"]) and np.array_equal(prev["stats"], r["stats"])), k      # every trial leaves a trace
        prev = r
    # the flag set before the call still means "nothing happens" with the counter armed elsewhere (thread-local, reset)
    r0 = oracle_lib.local_bundle_adjustment(oracle, w)
    assert (r0["stats"] == free["stats"]).all()


def _same_bits(a, b):
    return all((np.ascontiguousarray(a[k]).view(np.uint8) == np.ascontiguousarray(b[k]).view(np.uint8)).all() for k in ("poses", "points", "chi2", "outlier")) and \
        (np.asarray(a["stats"]) == np.asarray(b["stats"])).all()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", STOP_WINDOWS)
def test_lba_stop_flag_raised_during_the_solve(orbx, oracle, cfg, monkeypatch):
    """The caller's flag goes up while the host waits for trial decisions (test hook: when the host has seen the j-th decision of the call, the
    CALLER's byte is written, as the tracking thread's InsertKeyFrame would, src/LocalMapping.cc:123 / mbAbortBA).  From there the product's own
    mechanism carries it: wait_seq mirrors the byte into the pinned word, k_lm_decide reads it with the next decision or the one after.  The
    result must be the oracle's, stopped after the trial the product reports it stopped at (<= 1e-5, same trial counts); that trial is j + 1 or
    j + 2 (or the stage's natural end); stage two is skipped when the stop lands in stage one; the outlier pass and the write-back run; and the
    next call on the same handle is bit-identical to a fresh handle's."""
    w = orbx.lba_synth.make_window(**cfg)
    free_want = oracle_lib.local_bundle_adjustment(oracle, w)
    t1, t2 = int(free_want["stats"][1]), int(free_want["stats"][5])
    fresh = orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000)
    free_got = fresh.LocalBundleAdjustment(w)
    fresh.close()
    _compare(free_got, free_want, w)
    opt = orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000)
    stopped_somewhere = 0
    for j in list(range(1, t1 + t2)):
        monkeypatch.setenv("ORBX_LBA_TEST_STOP_AFTER_DECISIONS", str(j))
        flag = np.zeros(1, np.uint8)
        got = opt.LocalBundleAdjustment(w, stop_flag=flag)
        monkeypatch.delenv("ORBX_LBA_TEST_STOP_AFTER_DECISIONS")
        assert flag[0] == 1, j
        s = got["stats"]
        T = int(s[1] + s[5])
        assert j <= T <= min(j + 2, t1 + t2), (j, s)
        if j < t1:
            assert T <= t1 and s[4] == 0 and s[5] == 0, (j, s)         # raised in stage one with a trial to go: stage two must not start (:869-871)
        want = oracle_lib.local_bundle_adjustment(oracle, w, stop_after_trials=T)
        _compare(got, want, w)
        assert np.isfinite(got["poses"]).all() and got["chi2"].sum() > 0
        stopped_somewhere += T < t1 + t2
        # the handle is clean again: an un-stopped call equals a fresh handle's to the bit (hostStop reset, LM state, accumulators)
        flag[0] = 0
        again = opt.LocalBundleAdjustment(w, stop_flag=flag)
        assert _same_bits(again, free_got), j
    assert stopped_somewhere >= t1 + t2 - 3
    opt.close()


@pytest.mark.gpu
def test_lba_stop_flag_raised_by_another_thread(orbx, oracle):
    """No hook: a second thread raises the flag some hundred microseconds into the call (ctypes releases the GIL).  Wherever it lands, the result is
    the oracle's stopped after exactly the number of trials the product reports."""
    import threading
    w = orbx.lba_synth.make_window(K=50, P=5000, seed=12345)
    free_want = oracle_lib.local_bundle_adjustment(oracle, w)
    total = int(free_want["stats"][1] + free_want["stats"][5])
    opt = orbx.Optimizer(max_keyframes=64, max_points=6000, max_edges=80000)
    opt.LocalBundleAdjustment(w)
    landed = set()
    libc = ctypes.CDLL("libc.so.6")
    for delay_us in (200, 400, 600, 800, 1000, 1200, 1400, 1700, 2000, 2400, 2800, 3300):
        flag = np.zeros(1, np.uint8)
        go = threading.Event()

        def raiser():
            go.wait()
            libc.usleep(delay_us)      # (a foreign call: the GIL is free while it sleeps, the main thread marshals and enters orbx_lba_solve meanwhile)
            flag[0] = 1
        th = threading.Thread(target=raiser)
        th.start()
        go.set()
        got = opt.LocalBundleAdjustment(w, stop_flag=flag)
        th.join()
        s = got["stats"]
        T = int(s[1] + s[5])
        landed.add(T)
        if T == 0:      # before the first decision: the reference's own result is undefined there (e->chi2() of errors never computed); the estimates must be untouched
            assert np.allclose(got["points"], w["points"])
            continue
        want = oracle_lib.local_bundle_adjustment(oracle, w, stop_after_trials=T) if T < total else free_want
        _compare(got, want, w)
    assert len(landed) >= 2, landed      # (the delays span the call: the flag must have landed at different trials)
    opt.close()
