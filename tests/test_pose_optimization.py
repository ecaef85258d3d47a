"""Optimizer::PoseOptimization (reference src/Optimizer.cc:363-605): motion-only BA.  The CPU restatement (oracle/lba_oracle.cc)
is pinned to the reference's own PoseOptimization + g2o, compiled unmodified on oracle/eigenshim (tests/test_optimizer_ref.py); here
it is also checked against ground truth and an independent scipy solve, and the HIP kernel against the restatement (|delta pose|
<= 1e-5, identical outlier flags)."""
import numpy as np
import pytest

import oracle_lib
from test_projection import _rot


def make_frame(seed, n=600, stereo_frac=0.5, outlier_frac=0.15, noise=0.7):
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy, bf = 517.3, 516.5, 318.6, 255.3, 40.0
    T = np.eye(4)
    T[:3, :3] = _rot(*rng.normal(0, 0.05, 3))
    T[:3, 3] = rng.normal(0, 0.3, 3)
    u, v, z = rng.uniform(20, 620, n), rng.uniform(20, 460, n), rng.uniform(1.0, 10.0, n)
    Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    Xw = (T[:3, :3].T @ (Xc - T[:3, 3]).T).T
    octave = rng.integers(0, 8, n)
    sig = 1.2 ** octave
    obs = np.stack([u + rng.normal(0, noise, n) * sig, v + rng.normal(0, noise, n) * sig, u - bf / z + rng.normal(0, noise, n) * sig], 1)
    mono = rng.random(n) >= stereo_frac
    obs[mono, 2] = -1
    bad = rng.random(n) < outlier_frac
    obs[bad, :2] += rng.normal(0, 30, (int(bad.sum()), 2))
    T0 = np.eye(4)
    T0[:3, :3] = _rot(*rng.normal(0, 0.01, 3)) @ T[:3, :3]
    T0[:3, 3] = T[:3, 3] + rng.normal(0, 0.03, 3)
    return dict(pose=T0.astype(np.float32), cam=(fx, fy, cx, cy, bf), Xw=Xw.astype(np.float32), obs=obs.astype(np.float32),
                inv_sigma2=(1.0 / sig ** 2).astype(np.float32), truth=T, gross=bad)


def _reproj(T, fr, idx):
    fx, fy, cx, cy, bf = fr["cam"]
    X = fr["Xw"][idx].astype(np.float64)
    Xc = (T[:3, :3] @ X.T).T + T[:3, 3]
    u, v = fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy
    o = fr["obs"][idx].astype(np.float64)
    w = np.sqrt(fr["inv_sigma2"][idx].astype(np.float64))
    r = [(o[:, 0] - u) * w, (o[:, 1] - v) * w]
    st = o[:, 2] >= 0
    r.append(np.where(st, (o[:, 2] - (u - bf / Xc[:, 2])) * w, 0.0))
    return np.concatenate(r)


@pytest.mark.parametrize("seed,stereo", [(1, 0.5), (2, 0.0), (3, 1.0)])
def test_oracle_recovers_pose_and_agrees_with_scipy(oracle, seed, stereo):
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation
    fr = make_frame(seed, stereo_frac=stereo)
    res = oracle_lib.pose_optimization(oracle, fr)
    T = res["pose"].astype(np.float64)
    # close to the ground truth, and the gross outliers are (mostly) flagged
    assert np.abs(T[:3, 3] - fr["truth"][:3, 3]).max() < 0.05
    assert np.abs(T[:3, :3] - fr["truth"][:3, :3]).max() < 0.01
    assert res["outlier"][fr["gross"]].mean() > 0.9 and res["outlier"][~fr["gross"]].mean() < 0.1
    assert res["inliers"] == int((res["outlier"] == 0).sum())
    # the last round is a plain (non-robust) least-squares problem on the inliers of round three; its optimum on the final
    # inlier set must be next to the returned pose
    idx = np.flatnonzero(res["outlier"] == 0)

    def fun(p):
        Tn = np.eye(4)
        Tn[:3, :3] = Rotation.from_rotvec(p[:3]).as_matrix() @ T[:3, :3]
        Tn[:3, 3] = T[:3, 3] + p[3:]
        return _reproj(Tn, fr, idx)
    sol = least_squares(fun, np.zeros(6), method="lm", xtol=1e-14, ftol=1e-14)
    assert np.abs(sol.x).max() < 2e-3


@pytest.mark.gpu
def test_pose_optimization_hip_matches_oracle(orbx, oracle):
    frames = [make_frame(10 + i, n=[600, 1500, 40, 8, 2, 300][i], stereo_frac=[0.5, 0.0, 1.0, 0.5, 0.5, 0.3][i]) for i in range(6)]
    opt = orbx.PoseOptimizer(max_frames=8, max_features=2048)
    got = opt.PoseOptimization(frames)
    for i, fr in enumerate(frames):
        want = oracle_lib.pose_optimization(oracle, fr)
        assert np.abs(got[i]["pose"].astype(np.float64) - want["pose"]).max() <= 1e-5, i
        assert got[i]["inliers"] == want["inliers"], i
        diff = got[i]["outlier"] != want["outlier"]
        assert diff.sum() == 0, (i, int(diff.sum()))
        # same LM path; the 3-strikes stop rule ((chi_start - chi_end)*1e3 < chi_start) sits on rounding noise once converged,
        # so the iteration count of a round may differ by one
        assert np.abs(got[i]["stats"][0::2] - want["stats"][0::2]).max() <= 1, (i, got[i]["stats"], want["stats"])
    opt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nmax", [300, 900, 2000, 3500, 6000])
def test_pose_optimization_every_kernel_variant(orbx, oracle, nmax):
    """k_pose_opt keeps a frame's correspondences in registers, 2 / 4 / 8 / 16 / 32 per thread by the largest frame of the batch (the last
    variant, 4097 .. 8192 correspondences, spills part of them: a slow path the reference's unlimited std::vector asks for): one batch per
    variant, each with a small frame next to the largest one."""
    frames = [make_frame(70 + nmax, n=nmax, stereo_frac=0.4), make_frame(71 + nmax, n=max(12, nmax // 7), stereo_frac=0.6)]
    opt = orbx.PoseOptimizer(max_frames=2, max_features=8192)
    got = opt.PoseOptimization(frames)
    for i, fr in enumerate(frames):
        want = oracle_lib.pose_optimization(oracle, fr)
        assert np.abs(got[i]["pose"].astype(np.float64) - want["pose"]).max() <= 1e-5, i
        assert got[i]["inliers"] == want["inliers"], i
        assert (got[i]["outlier"] != want["outlier"]).sum() == 0, i
    opt.close()


@pytest.mark.gpu
def test_pose_optimization_rejects_more_than_8192_correspondences(orbx):
    fr = make_frame(5, n=8300)
    opt = orbx.PoseOptimizer(max_frames=1, max_features=8300)
    with pytest.raises(Exception):
        opt.PoseOptimization([fr])
    opt.close()


@pytest.mark.gpu
def test_pose_optimization_many_random_problems(orbx, oracle):
    """Round 6 put fused multiply-adds, reciprocal-square-root forms and DPP sums into k_pose_opt (csrc/orbx_lba.hip; rounded differently from the
    restatement in the last bits): 80 problems over the sizes around the kernel's instantiation boundaries (256 / 512 / 768 / 1024 correspondences),
    mono / mixed / stereo, with and without gross outliers - pose within 1e-5, identical outlier flags and inlier counts, iteration counts within one."""
    rng = np.random.default_rng(2026)
    opt = orbx.PoseOptimizer(max_frames=2, max_features=2048)
    worst = 0.0
    for t in range(80):
        n = int(rng.choice([12, 40, 120, 256, 257, 310, 512, 513, 720, 768, 769, 1024, 1025, 1500, 2000]))
        fr = make_frame(1000 + t, n=n, stereo_frac=float(rng.choice([0.0, 0.3, 0.5, 1.0])), outlier_frac=float(rng.choice([0.0, 0.1, 0.3])),
                        noise=float(rng.choice([0.3, 0.7, 1.5])))
        got = opt.PoseOptimization([fr])[0]
        want = oracle_lib.pose_optimization(oracle, fr)
        d = float(np.abs(got["pose"].astype(np.float64) - want["pose"]).max())
        worst = max(worst, d)
        assert d <= 1e-5, (t, n, d)
        assert got["inliers"] == want["inliers"] and (got["outlier"] != want["outlier"]).sum() == 0, (t, n)
        assert np.abs(got["stats"][0::2] - want["stats"][0::2]).max() <= 1, (t, n, got["stats"], want["stats"])
    opt.close()
    assert worst < 1e-6, worst      # (observed: 2e-10)
