"""Synthetic local-BA window (BASELINE.json configs[4]).

K keyframes on a circle of radius 5 m looking at a cloud of P points uniform in a 4 m cube,
TUM1 intrinsics, every point observed by the keyframes where it projects inside 640x480 at
z > 0.1 (at most `max_obs` observations), observation noise N(0, 1 px * scale^octave) with
octave uniform in 0..7 (=> invSigma2 = 1/1.44^octave), `outlier_frac` gross outliers (+30 px),
the `n_fixed` oldest keyframes fixed (+ keyframe id 0), poses perturbed 1 deg / 2 cm, points
2 cm.  Everything is rounded to float32: the reference's boundary precision
(src/Converter.cc:57-70, 96-107).  Pure numpy; used by tests and bench.
"""
import numpy as np

TUM1 = dict(fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989)   # Examples/Monocular/TUM1.yaml:9-12
KITTI_BF = 386.1448                                                      # Examples/Stereo/KITTI00-02.yaml:25 (used for stereo edges)


def _rot(axis, ang):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def make_window(K=50, P=5000, seed=12345, max_obs=12, n_fixed=10, outlier_frac=0.05, stereo_frac=0.0, W=640, H=480,
                pose_noise=(np.deg2rad(1.0), 0.02), point_noise=0.02):
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = TUM1["fx"], TUM1["fy"], TUM1["cx"], TUM1["cy"]
    bf = 40.0   # small baseline*fx so that u_R stays positive for the synthetic depths
    pts = rng.uniform(-2.0, 2.0, size=(P, 3))
    Tcw = np.zeros((K, 4, 4))
    for k in range(K):
        a = 2 * np.pi * k / K
        C = np.array([5.0 * np.cos(a), 0.3 * np.sin(3 * a), 5.0 * np.sin(a)])      # camera centre on a (wobbly) circle
        z = -C / np.linalg.norm(C)                                                  # look at the origin
        x = np.cross(np.array([0.0, 1.0, 0.0]), z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])                                                     # world -> camera
        Tcw[k, :3, :3] = R
        Tcw[k, :3, 3] = -R @ C
        Tcw[k, 3, 3] = 1
    edge_p, edge_k, obs, inv_s2 = [], [], [], []
    for l in range(P):
        ks = rng.permutation(K)
        n = 0
        for k in ks:
            Xc = Tcw[k, :3, :3] @ pts[l] + Tcw[k, :3, 3]
            if Xc[2] <= 0.1:
                continue
            u, v = fx * Xc[0] / Xc[2] + cx, fy * Xc[1] / Xc[2] + cy
            if not (0 <= u < W and 0 <= v < H):
                continue
            octave = int(rng.integers(0, 8))
            sig = 1.2 ** octave
            du, dv = rng.normal(0, sig, 2)
            if rng.random() < outlier_frac:
                du += 30.0 * rng.choice([-1, 1]); dv += 30.0 * rng.choice([-1, 1])
            ur = -1.0
            if rng.random() < stereo_frac:
                ur = (u + du) - bf / Xc[2] + rng.normal(0, sig)
            edge_p.append(l); edge_k.append(k); obs.append((u + du, v + dv, ur)); inv_s2.append(1.0 / (1.2 ** (2 * octave)))
            n += 1
            if n >= max_obs:
                break
    fixed = np.zeros(K, np.uint8)
    fixed[:n_fixed] = 1
    fixed[0] = 1                                            # vSE3->setFixed(pKFi->mnId==0), src/Optimizer.cc:722
    poses0 = Tcw.copy()
    for k in range(K):
        if fixed[k]:
            continue
        dR = _rot(rng.normal(size=3), rng.normal(0, pose_noise[0]))
        poses0[k, :3, :3] = dR @ Tcw[k, :3, :3]
        poses0[k, :3, 3] = dR @ Tcw[k, :3, 3] + rng.normal(0, pose_noise[1], 3)
    pts0 = pts + rng.normal(0, point_noise, size=pts.shape)
    intr = np.tile(np.array([fx, fy, cx, cy, bf], np.float32), (K, 1))
    return dict(K=K, P=P, E=len(edge_p),
                poses=np.ascontiguousarray(poses0.reshape(K, 16), np.float32), fixed=fixed,
                intr=np.ascontiguousarray(intr, np.float32), points=np.ascontiguousarray(pts0, np.float32),
                edge_point=np.array(edge_p, np.int32), edge_kf=np.array(edge_k, np.int32),
                edge_obs=np.ascontiguousarray(np.array(obs), np.float32), edge_inv_sigma2=np.array(inv_s2, np.float32),
                true_poses=Tcw.reshape(K, 16), true_points=pts)
