"""ctypes access to the CPU checkers under oracle/ (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_SO = ROOT / "oracle" / "liborb_oracle.so"
REF_SO = ROOT / "oracle" / "_ref" / "liborbref.so"


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def build():
    subprocess.run(["make", "-s", "-f", str(ROOT / "oracle" / "Makefile"), "all"], check=True)


def unpack_cands(p):
    p = np.asarray(p, np.uint32)
    return np.stack([p & 0xfff, (p >> 12) & 0xfff, p >> 24], 1).astype(np.int32)


class _Ext:
    """One extractor instance of a checker library (prefix orbo_ or orbref_)."""

    def __init__(self, lib, prefix, args):
        self.lib, self.prefix, self.args = lib, prefix, args
        create = getattr(lib, prefix + "create")
        create.restype = ctypes.c_void_p
        create.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        self.h = ctypes.c_void_p(create(*args))
        self.nlevels = args[2]

    def __del__(self):
        try:
            getattr(self.lib, self.prefix + "destroy")(self.h)
        except Exception:
            pass

    def tables(self):
        nl = self.nlevels
        t = np.zeros(4 * nl, np.float32)
        q = np.zeros(nl, np.int32)
        u = np.zeros(16, np.int32)
        getattr(self.lib, self.prefix + "tables")(self.h, _p(t), _p(q), _p(u))
        return t.reshape(4, nl), q, u

    def extract(self, im, cap=8192):
        H, W = im.shape
        im = np.ascontiguousarray(im)
        k = np.zeros((cap, 7), np.float32)
        d = np.zeros((cap, 32), np.uint8)
        if self.prefix == "orbo_":
            lc = np.zeros(self.nlevels, np.int32)
            n = self.lib.orbo_extract(self.h, _p(im), W, H, W, _p(k), _p(d), cap, _p(lc))
        else:
            n = self.lib.orbref_extract(self.h, _p(im), W, H, W, _p(k), _p(d), cap)
        assert n <= cap
        return k[:n].copy(), d[:n].copy()


class Oracle:
    def __init__(self):
        if not ORACLE_SO.exists():
            build()
        self.lib = ctypes.CDLL(str(ORACLE_SO))
        self.lib.orbo_ic_angle.restype = ctypes.c_float
        self.lib.orbo_ic_angle.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        self.lib.orbo_descriptor.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        self.lib.orbo_sincos_exhaustive.restype = ctypes.c_long
        self.lib.orbo_sincos_exhaustive.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        self.ref = ctypes.CDLL(str(REF_SO)) if REF_SO.exists() else None

    def restatement(self, nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7):
        return _Ext(self.lib, "orbo_", (nfeatures, scale, nlevels, ini, mn))

    def reference(self, nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7):
        if self.ref is None:
            return None
        return _Ext(self.ref, "orbref_", (nfeatures, scale, nlevels, ini, mn))

    # ---- stage functions of the restatement ----
    def level_size(self, ext, W, H, level):
        w, h = ctypes.c_int(), ctypes.c_int()
        self.lib.orbo_level_size(ext.h, W, H, level, ctypes.byref(w), ctypes.byref(h))
        return w.value, h.value

    def pyramid(self, ext, im):
        H, W = im.shape
        levels = [np.ascontiguousarray(im)]
        for l in range(1, ext.nlevels):
            w, h = self.level_size(ext, W, H, l)
            dst = np.zeros((h, w), np.uint8)
            src = levels[-1]
            self.lib.orbo_resize(_p(src), src.shape[1], src.shape[0], _p(dst), w, h)
            levels.append(dst)
        return levels

    def score_map(self, img, min_th):
        h, w = img.shape
        out = np.zeros((h, w), np.uint8)
        self.lib.orbo_score_map(_p(np.ascontiguousarray(img)), w, h, w, min_th, _p(out))
        return out

    def cell_candidates(self, scores, ini_th, cap=1 << 17):
        h, w = scores.shape
        out = np.zeros(cap, np.uint32)
        n = self.lib.orbo_cell_candidates(_p(np.ascontiguousarray(scores)), w, h, ini_th, _p(out), cap)
        assert n <= cap
        return out[:n].copy()

    def octree(self, packed, w, h, N, cap=8192):
        out = np.zeros(cap, np.uint32)
        packed = np.ascontiguousarray(packed, np.uint32)
        n = self.lib.orbo_octree(_p(packed), len(packed), 16, w - 16, 16, h - 16, N, _p(out), cap)
        return out[:n].copy()

    def ref_octree(self, ext, packed, w, h, N, cap=8192):
        c = unpack_cands(packed).astype(np.float32)
        out = np.zeros((cap, 3), np.float32)
        n = self.ref.orbref_octree(ext.h, _p(np.ascontiguousarray(c)), len(c), 16, w - 16, 16, h - 16, N, _p(out), cap)
        o = out[:n].astype(np.uint32)
        return (o[:, 0] | (o[:, 1] << 12) | (o[:, 2] << 24)).astype(np.uint32)

    def ic_angle(self, ext, img, x, y):
        return self.lib.orbo_ic_angle(ext.h, _p(img), img.shape[1], int(x), int(y))

    def blur(self, img, taps=None):
        h, w = img.shape
        out = np.zeros((h, w), np.uint8)
        t = None if taps is None else _p(np.asarray(taps, np.uint16))
        self.lib.orbo_blur(_p(np.ascontiguousarray(img)), w, h, w, _p(out), w, t)
        return out

    def descriptor(self, blur, x, y, angle):
        d = np.zeros(32, np.uint8)
        self.lib.orbo_descriptor(_p(blur), blur.shape[1], int(x), int(y), ctypes.c_float(angle), _p(d))
        return d


# ---- Hamming matcher restatements (oracle/match_oracle.cc) ----
def _kp7(k):
    """structured orbx keypoints -> 7-float rows as the oracles take them."""
    return np.ascontiguousarray(np.stack([k["x"], k["y"], k["size"], k["angle"], k["response"], k["octave"].astype(np.float32),
                                          k["class_id"].astype(np.float32)], 1), np.float32)


def _opt(a, dt):
    return None if a is None else np.ascontiguousarray(a, dt)


def search_by_bow(orc, mode, kpsA, descA, kpsB, descB, nnratio, check_ori, groupsA=None, groupsB=None, validA=None, validB=None):
    lib = orc.lib
    lib.mo_search_by_bow.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
    aA = np.ascontiguousarray(kpsA["angle"], np.float32)
    aB = np.ascontiguousarray(kpsB["angle"], np.float32)
    dA = np.ascontiguousarray(descA, np.uint8)
    dB = np.ascontiguousarray(descB, np.uint8)
    gA, gB, vA, vB = _opt(groupsA, np.int32), _opt(groupsB, np.int32), _opt(validA, np.uint8), _opt(validB, np.uint8)
    nout = len(kpsB) if mode == 0 else len(kpsA)
    out = np.full(max(nout, 1), -1, np.int32)
    P = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    n = lib.mo_search_by_bow(mode, P(dA), P(aA), P(gA), P(vA), len(kpsA), P(dB), P(aB), P(gB), P(vB), len(kpsB),
                             ctypes.c_float(nnratio), 1 if check_ori else 0, P(out))
    return n, out[:nout]


def stereo_hamming(orc, kpsL, descL, kpsR, descR, scale_factors, nrows, max_d):
    lib = orc.lib
    lib.mo_stereo_hamming.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    kl, kr = _kp7(kpsL), _kp7(kpsR)
    dl, dr = np.ascontiguousarray(descL, np.uint8), np.ascontiguousarray(descR, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    bd = np.zeros(max(len(kl), 1), np.int32)
    bi = np.zeros(max(len(kl), 1), np.int32)
    lib.mo_stereo_hamming(_p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr), _p(sf), nrows, ctypes.c_float(max_d), _p(bd), _p(bi))
    return bd[:len(kl)], bi[:len(kl)]


def descriptor_distance(orc, a, b):
    return orc.lib.mo_descriptor_distance(_p(np.ascontiguousarray(a, np.uint8)), _p(np.ascontiguousarray(b, np.uint8)))


# ---- LocalBundleAdjustment restatement (oracle/lba_oracle.cc) ----
def local_bundle_adjustment(orc, w, stop=None, stop_after_trials=0):
    """stop_after_trials = k > 0: the run sees its stop flag raised right after its k-th Levenberg trial (counted over both stages), i.e. before
    the first read of the flag that follows that trial (oracle/lba_oracle.cc: lo_set_stop_after_trials)."""
    lib = orc.lib
    if stop_after_trials:
        lib.lo_set_stop_after_trials(int(stop_after_trials))
        try:
            return local_bundle_adjustment(orc, w, stop=None)
        finally:
            lib.lo_set_stop_after_trials(0)
    K, P, E = w["K"], w["P"], w["E"]
    poses_out = np.zeros((K, 16), np.float32)
    points_out = np.zeros((P, 3), np.float32)
    chi2 = np.zeros(E, np.float64)
    outl = np.zeros(E, np.uint8)
    stats = np.zeros(8, np.float64)
    lib.lo_local_bundle_adjustment.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 10
    lib.lo_local_bundle_adjustment(K, _p(w["poses"]), _p(w["fixed"]), _p(w["intr"]), P, _p(w["points"]), E, _p(w["edge_point"]), _p(w["edge_kf"]),
                                   _p(w["edge_obs"]), _p(w["edge_inv_sigma2"]), None if stop is None else _p(stop), _p(poses_out), _p(points_out),
                                   _p(chi2), _p(outl), _p(stats))
    return dict(poses=poses_out, points=points_out, chi2=chi2, outlier=outl, stats=stats)


def bundle_adjustment(orc, w, iterations, robust, stop=None):
    """Restatement of Optimizer::BundleAdjustment (oracle/lba_oracle.cc:lo_bundle_adjustment)."""
    lib = orc.lib
    K, P, E = w["K"], w["P"], w["E"]
    poses_out, points_out = np.zeros((K, 16), np.float32), np.zeros((P, 3), np.float32)
    chi2, outl, stats = np.zeros(E, np.float64), np.zeros(E, np.uint8), np.zeros(8, np.float64)
    lib.lo_bundle_adjustment.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 5 + \
                                        [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
    lib.lo_bundle_adjustment(K, _p(w["poses"]), _p(w["fixed"]), _p(w["intr"]), P, _p(w["points"]), E, _p(w["edge_point"]), _p(w["edge_kf"]),
                             _p(w["edge_obs"]), _p(w["edge_inv_sigma2"]), None if stop is None else _p(stop), int(iterations), 1 if robust else 0,
                             _p(poses_out), _p(points_out), _p(chi2), _p(outl), _p(stats))
    return dict(poses=poses_out, points=points_out, chi2=chi2, outlier=outl, stats=stats)


# ---- the UNMODIFIED reference matcher / Frame / KeyFrame / MapPoint / DBoW2 sources compiled
# ---- against oracle/cvshim (oracle/refslam_wrap.cc -> oracle/_ref/liborbslam.so)
SLAM_SO = ROOT / "oracle" / "_ref" / "liborbslam.so"
_slam = None


SLAM_HIP_SO = ROOT / "oracle" / "_ref" / "liborbslam_hip.so"
_slam_hip = None


def slam_lib():
    """ctypes handle of oracle/_ref/liborbslam.so, or None when it was never built (it is built
    in the container that has /root/reference and travels to the GPU box as a binary)."""
    global _slam
    if _slam is None and SLAM_SO.exists():
        _slam = ctypes.CDLL(str(SLAM_SO))
    return _slam


def slam_hip_lib():
    """The DROP-IN build: the same reference sources with shim/ORBextractor + the HIP bodies of
    SearchByBoW / DescriptorDistance / ComputeStereoMatches linked in (oracle/Makefile)."""
    global _slam_hip
    if _slam_hip is None and SLAM_HIP_SO.exists():
        # liborbx.so (and with it the HIP / HSA runtime) is loaded on its own FIRST: brought in as a mere dependency of this library it
        # would share its lookup scope (see oracle/exports.map for what that did to operator new before the checker libraries hid theirs)
        import importlib
        importlib.import_module("self_commit_orb-slam2_amd").load_library()
        _slam_hip = ctypes.CDLL(str(SLAM_HIP_SO))
    return _slam_hip


def ref_descriptor_distance(a, b, lib=None):
    return (lib or slam_lib()).orbslam_descriptor_distance(_p(np.ascontiguousarray(a, np.uint8)), _p(np.ascontiguousarray(b, np.uint8)))


def ref_search_by_bow(mode, kpsA, descA, kpsB, descB, nnratio, check_ori, groupsA=None, groupsB=None, validA=None, validB=None, lib=None):
    """ORBmatcher::SearchByBoW of the reference, driven through real KeyFrame/Frame/MapPoint objects."""
    lib = lib or slam_lib()
    lib.orbslam_search_by_bow.argtypes = [ctypes.c_int] + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] * 2 + \
                                         [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
    kA, kB = _kp7(kpsA), _kp7(kpsB)
    dA, dB = np.ascontiguousarray(descA, np.uint8), np.ascontiguousarray(descB, np.uint8)
    # the reference takes DBoW2 FeatureVectors: no groups = every feature under one node
    gA = np.zeros(len(kA), np.int32) if groupsA is None else np.ascontiguousarray(groupsA, np.int32)
    gB = np.zeros(len(kB), np.int32) if groupsB is None else np.ascontiguousarray(groupsB, np.int32)
    vA, vB = _opt(validA, np.uint8), _opt(validB, np.uint8)
    nout = len(kB) if mode == 0 else len(kA)
    out = np.full(max(nout, 1), -1, np.int32)
    P = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    n = lib.orbslam_search_by_bow(mode, P(kA), P(dA), len(kA), P(gA), P(vA), P(kB), P(dB), len(kB), P(gB), P(vB),
                                  ctypes.c_float(nnratio), 1 if check_ori else 0, P(out))
    return n, out[:nout]


def ref_stereo_frame(imL, imR, nfeatures, fx, fy, cx, cy, bf, th_depth=35.0, scale=1.2, nlevels=8, ini=20, mn=7, cap=8192, lib=None):
    """Frame::Frame(imLeft, imRight, ...) of the reference: both extractions + ComputeStereoMatches."""
    lib = lib or slam_lib()
    H, W = imL.shape
    imL, imR = np.ascontiguousarray(imL), np.ascontiguousarray(imR)
    kL, kR = np.zeros((cap, 7), np.float32), np.zeros((cap, 7), np.float32)
    dL, dR = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
    uR, dep = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    nL, nR = ctypes.c_int(), ctypes.c_int()
    lib.orbslam_stereo_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int] + \
                                        [ctypes.c_float] * 6 + [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.orbslam_stereo_frame(_p(imL), _p(imR), W, H, W, nfeatures, scale, nlevels, ini, mn, fx, fy, cx, cy, bf, th_depth,
                             _p(kL), _p(dL), _p(kR), _p(dR), _p(uR), _p(dep), cap, ctypes.byref(nL), ctypes.byref(nR))
    a, b = nL.value, nR.value
    assert a <= cap and b <= cap
    return dict(kpsL=kL[:a].copy(), descL=dL[:a].copy(), kpsR=kR[:b].copy(), descR=dR[:b].copy(), uRight=uR[:a].copy(), depth=dep[:a].copy())


def ref_compute_stereo_matches(kpsL, descL, kpsR, descR, pyrL, pyrR, scale_factors, bf):
    """Frame::ComputeStereoMatches of the reference on given features and pyramids (lists of 2-D u8 arrays)."""
    lib = slam_lib()
    kl = kpsL if kpsL.dtype == np.float32 and kpsL.ndim == 2 else _kp7(kpsL)
    kr = kpsR if kpsR.dtype == np.float32 and kpsR.ndim == 2 else _kp7(kpsR)
    kl, kr = np.ascontiguousarray(kl), np.ascontiguousarray(kr)
    dl, dr = np.ascontiguousarray(descL, np.uint8), np.ascontiguousarray(descR, np.uint8)
    lw = np.array([p.shape[1] for p in pyrL], np.int32)
    lh = np.array([p.shape[0] for p in pyrL], np.int32)
    pl = np.concatenate([np.ascontiguousarray(p).ravel() for p in pyrL])
    pr = np.concatenate([np.ascontiguousarray(p).ravel() for p in pyrR])
    sf = np.ascontiguousarray(scale_factors, np.float32)
    uR, dep = np.zeros(max(len(kl), 1), np.float32), np.zeros(max(len(kl), 1), np.float32)
    lib.orbslam_compute_stereo_matches.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] * 2 + [ctypes.c_void_p] * 4 + \
                                                  [ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    lib.orbslam_compute_stereo_matches(_p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr), _p(pl), _p(pr), _p(lw), _p(lh), len(pyrL), _p(sf),
                                       ctypes.c_float(bf), _p(uR), _p(dep))
    return uR[:len(kl)], dep[:len(kl)]


def compute_stereo_matches(orc, kpsL, descL, kpsR, descR, pyrL, pyrR, scale_factors, inv_scale_factors, bf, mb=0.0):
    """Restatement of the complete Frame::ComputeStereoMatches (oracle/match_oracle.cc)."""
    lib = orc.lib
    kl = kpsL if kpsL.dtype == np.float32 and kpsL.ndim == 2 else _kp7(kpsL)
    kr = kpsR if kpsR.dtype == np.float32 and kpsR.ndim == 2 else _kp7(kpsR)
    kl, kr = np.ascontiguousarray(kl), np.ascontiguousarray(kr)
    dl, dr = np.ascontiguousarray(descL, np.uint8), np.ascontiguousarray(descR, np.uint8)
    lw = np.array([p.shape[1] for p in pyrL], np.int32)
    lh = np.array([p.shape[0] for p in pyrL], np.int32)
    pl = np.concatenate([np.ascontiguousarray(p).ravel() for p in pyrL])
    pr = np.concatenate([np.ascontiguousarray(p).ravel() for p in pyrR])
    sf = np.ascontiguousarray(scale_factors, np.float32)
    isf = np.ascontiguousarray(inv_scale_factors, np.float32)
    n = max(len(kl), 1)
    uR, dep, sad = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
    lib.mo_compute_stereo_matches.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] * 2 + [ctypes.c_void_p] * 4 + \
                                             [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float] + [ctypes.c_void_p] * 3
    with np.errstate(divide="ignore"):
        lib.mo_compute_stereo_matches(_p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr), _p(pl), _p(pr), _p(lw), _p(lh), len(pyrL), _p(sf), _p(isf),
                                      ctypes.c_float(bf), ctypes.c_float(mb), _p(uR), _p(dep), _p(sad))
    return uR[:len(kl)], dep[:len(kl)], sad[:len(kl)]


# ---- DBoW2 vocabulary transform: restatement (oracle/match_oracle.cc) and the compiled reference ----
def voc_transform(orc, voc, descriptors, levelsup):
    lib = orc.lib
    d = np.ascontiguousarray(descriptors, np.uint8)
    n = len(d)
    word, node, weight = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64)
    par, leaf = np.ascontiguousarray(voc["parent"], np.int32), np.ascontiguousarray(voc["is_leaf"], np.uint8)
    nd, wt = np.ascontiguousarray(voc["desc"], np.uint8), np.ascontiguousarray(voc["weight"], np.float64)
    wid = np.full(len(par), -1, np.int32)
    wid[leaf > 0] = np.arange(int((leaf > 0).sum()), dtype=np.int32)     # word ids count the leaves in node-id order
    lib.mo_voc_transform.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3
    lib.mo_voc_transform(len(par), _p(par), _p(leaf), _p(nd), _p(wt), _p(wid), int(voc["L"]), _p(d), n, levelsup, _p(word), _p(node), _p(weight))
    return word[:n], node[:n], weight[:n]


def bow_vector(word, weight):
    """BowVector of the reference's vector transform for TF_IDF/TF weighting + L1 scoring
    (TemplatedVocabulary.h:1146-1178, BowVector.cpp:62-84): per word the weight added once per
    occurrence in feature order, then divided by the L1 norm summed in ascending word-id order."""
    acc = {}
    for w, v in zip(word, weight):
        if v > 0:
            acc[int(w)] = acc.get(int(w), 0.0) + float(v)
    ids = sorted(acc)
    norm = 0.0
    for i in ids:
        norm += abs(acc[i])
    vals = [acc[i] / norm if norm > 0 else acc[i] for i in ids]
    return np.array(ids, np.int32), np.array(vals, np.float64)


class RefVocabulary:
    """ORBVocabulary of the compiled reference, loaded through its own loadFromTextFile."""

    def __init__(self, path, lib=None):
        self.lib = lib or slam_lib()
        self.lib.orbslam_voc_load.restype = ctypes.c_void_p
        self.lib.orbslam_voc_load.argtypes = [ctypes.c_char_p]
        self.lib.orbslam_voc_destroy.argtypes = [ctypes.c_void_p]
        self.lib.orbslam_voc_size.argtypes = [ctypes.c_void_p]
        self.h = ctypes.c_void_p(self.lib.orbslam_voc_load(str(path).encode()))
        assert self.h.value, "loadFromTextFile failed"

    def __del__(self):
        try:
            self.lib.orbslam_voc_destroy(self.h)
        except Exception:
            pass

    def size(self):
        return self.lib.orbslam_voc_size(self.h)

    def transform(self, descriptors, levelsup):
        d = np.ascontiguousarray(descriptors, np.uint8)
        n = len(d)
        word, node, fvn = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        weight = np.zeros(n, np.float64)
        bi, bv = np.zeros(n + 1, np.int32), np.zeros(n + 1, np.float64)
        self.lib.orbslam_voc_transform.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p]
        nb = self.lib.orbslam_voc_transform(self.h, _p(d), n, levelsup, _p(word), _p(node), _p(weight), _p(bi), _p(bv), n + 1, _p(fvn))
        return dict(word=word, node=node, weight=weight, bow_ids=bi[:nb].copy(), bow_vals=bv[:nb].copy(), fv_node=fvn)

    def compute_bow(self, descriptors, which=0):
        """Frame::ComputeBoW (which=0) / KeyFrame::ComputeBoW (which=1) on an object holding these descriptors."""
        d = np.ascontiguousarray(descriptors, np.uint8)
        n = len(d)
        fvn = np.zeros(n, np.int32)
        bi, bv = np.zeros(n + 1, np.int32), np.zeros(n + 1, np.float64)
        self.lib.orbslam_compute_bow.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        nb = self.lib.orbslam_compute_bow(self.h, which, _p(d), n, _p(bi), _p(bv), n + 1, _p(fvn))
        return dict(bow_ids=bi[:nb].copy(), bow_vals=bv[:nb].copy(), fv_node=fvn)


# ---- ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th): restatement and compiled reference ----
def _proj_args(fr, pts):
    k7 = np.ascontiguousarray(fr["kps7"], np.float32)
    return dict(k7=k7, desc=np.ascontiguousarray(fr["desc"], np.uint8), uR=np.ascontiguousarray(fr["u_right"], np.float32),
                occ=np.ascontiguousarray(fr["occupied"], np.uint8), sf=np.ascontiguousarray(fr["scale_factors"], np.float32),
                px=np.ascontiguousarray(pts["proj_x"], np.float32), py=np.ascontiguousarray(pts["proj_y"], np.float32),
                pxr=np.ascontiguousarray(pts["proj_xr"], np.float32), lvl=np.ascontiguousarray(pts["level"], np.int32),
                vc=np.ascontiguousarray(pts["view_cos"], np.float32), inv=np.ascontiguousarray(pts["in_view"], np.uint8),
                obs=np.ascontiguousarray(pts["has_obs"], np.uint8), md=np.ascontiguousarray(pts["desc"], np.uint8))


def search_by_projection(orc, fr, pts, th, nnratio):
    a = _proj_args(fr, pts)
    n, m = len(a["k7"]), len(a["px"])
    W, H = fr["width"], fr["height"]
    out = np.full(max(n, 1), -1, np.int32)
    lib = orc.lib
    lib.mo_search_by_projection.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_float] * 4 + [ctypes.c_void_p] * 9 + \
                                           [ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
    gw, gh = np.float32(64) / np.float32(W), np.float32(48) / np.float32(H)
    nm = lib.mo_search_by_projection(_p(a["k7"]), _p(a["desc"]), _p(a["uR"]), _p(a["occ"]), n, 0.0, 0.0, float(gw), float(gh), _p(a["sf"]), _p(a["px"]), _p(a["py"]),
                                     _p(a["pxr"]), _p(a["lvl"]), _p(a["vc"]), _p(a["inv"]), _p(a["obs"]), _p(a["md"]), m, th, nnratio, _p(out))
    return nm, out[:n]


def ref_search_by_projection(fr, pts, th, nnratio, lib=None):
    lib = lib or slam_lib()
    a = _proj_args(fr, pts)
    n, m = len(a["k7"]), len(a["px"])
    out = np.full(max(n, 1), -1, np.int32)
    lib.orbslam_search_by_projection.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 8 + \
                                                [ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
    nm = lib.orbslam_search_by_projection(_p(a["k7"]), _p(a["desc"]), _p(a["uR"]), _p(a["occ"]), n, fr["width"], fr["height"], _p(a["sf"]), len(a["sf"]),
                                          _p(a["px"]), _p(a["py"]), _p(a["pxr"]), _p(a["lvl"]), _p(a["vc"]), _p(a["inv"]), _p(a["obs"]), _p(a["md"]), m, th,
                                          nnratio, _p(out))
    return nm, out[:n]


# ---- ORBmatcher::SearchByProjection(Frame &Current, const Frame &Last, th, bMono): restatement and compiled reference ----
def _last_args(fr, last):
    a = dict(k7=np.ascontiguousarray(fr["kps7"], np.float32), desc=np.ascontiguousarray(fr["desc"], np.uint8),
             uR=np.ascontiguousarray(fr["u_right"], np.float32), occ=np.ascontiguousarray(fr["occupied"], np.uint8),
             sf=np.ascontiguousarray(fr["scale_factors"], np.float32), Tc=np.ascontiguousarray(fr["Tcw"], np.float32),
             Tl=np.ascontiguousarray(last["Tcw"], np.float32), lv=np.ascontiguousarray(last["valid"], np.uint8),
             lp=np.ascontiguousarray(last["pos"], np.float32), ld=np.ascontiguousarray(last["desc"], np.uint8),
             lo=np.ascontiguousarray(last["has_obs"], np.uint8), lk=np.ascontiguousarray(last["kps7"], np.float32))
    return a


def search_by_projection_last(orc, fr, last, th, mono, check_ori):
    a = _last_args(fr, last)
    n, nl = len(a["k7"]), len(a["lk"])
    W, H = np.float32(fr["width"]), np.float32(fr["height"])
    cam = fr["cam"]
    fx, fy, cx, cy, bf = [np.float32(v) for v in cam]
    mb = bf / fx
    out = np.full(max(n, 1), -1, np.int32)
    valid = (a["lv"] == 1).astype(np.uint8)                 # 2 = outlier of the last frame: skipped (:1608)
    octave = np.ascontiguousarray(a["lk"][:, 5].astype(np.int32))
    angle = np.ascontiguousarray(a["lk"][:, 3])
    lib = orc.lib
    F = ctypes.c_float
    lib.mo_search_by_projection_last.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] + [F] * 6 + [ctypes.c_void_p] * 3 + [F] * 6 + [ctypes.c_void_p] * 6 + \
                                                [ctypes.c_int, F, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    nm = lib.mo_search_by_projection_last(_p(a["k7"]), _p(a["desc"]), _p(a["uR"]), _p(a["occ"]), n, 0.0, 0.0, float(W), float(H), float(np.float32(64) / W),
                                          float(np.float32(48) / H), _p(a["sf"]), _p(a["Tc"]), _p(a["Tl"]), float(fx), float(fy), float(cx), float(cy), float(bf),
                                          float(mb), _p(valid), _p(a["lp"]), _p(a["ld"]), _p(a["lo"]), _p(octave), _p(angle), nl, th, int(mono), int(check_ori), _p(out))
    return nm, out[:n]


def ref_search_by_projection_last(fr, last, th, mono, check_ori, lib=None):
    lib = lib or slam_lib()
    a = _last_args(fr, last)
    n, nl = len(a["k7"]), len(a["lk"])
    fx, fy, cx, cy, bf = [float(np.float32(v)) for v in fr["cam"]]
    out = np.full(max(n, 1), -1, np.int32)
    F = ctypes.c_float
    lib.orbslam_search_by_projection_last.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + \
                                                     [F] * 5 + [ctypes.c_void_p] * 5 + [ctypes.c_int, F, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    nm = lib.orbslam_search_by_projection_last(_p(a["k7"]), _p(a["desc"]), _p(a["uR"]), _p(a["occ"]), n, fr["width"], fr["height"], _p(a["sf"]), len(a["sf"]),
                                               _p(a["Tc"]), _p(a["Tl"]), fx, fy, cx, cy, bf, _p(a["lv"]), _p(a["lp"]), _p(a["ld"]), _p(a["lo"]), _p(a["lk"]), nl,
                                               th, int(mono), int(check_ori), _p(out))
    return nm, out[:n]


# ---- Optimizer::PoseOptimization restatement (oracle/lba_oracle.cc) ----
def pose_optimization(orc, fr):
    lib = orc.lib
    pose = np.ascontiguousarray(np.asarray(fr["pose"], np.float32).reshape(16))
    cam = np.ascontiguousarray(fr["cam"], np.float32)
    Xw, obs, inv = np.ascontiguousarray(fr["Xw"], np.float32), np.ascontiguousarray(fr["obs"], np.float32), np.ascontiguousarray(fr["inv_sigma2"], np.float32)
    n = len(Xw)
    out, outl, st = np.zeros(16, np.float32), np.zeros(max(n, 1), np.uint8), np.zeros(8, np.float64)
    lib.lo_pose_optimization.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6
    ret = lib.lo_pose_optimization(_p(pose), _p(cam), n, _p(Xw), _p(obs), _p(inv), _p(out), _p(outl), _p(st))
    return dict(pose=out.reshape(4, 4), outlier=outl[:n], inliers=ret, stats=st)


NCELL = 64 * 48


def ref_mono_frame(im, nfeatures, fx, fy, cx, cy, dist, scale=1.2, nlevels=8, ini=20, mn=7, cap=8192, lib=None):
    """Frame::Frame(imGray, ...) of the reference: extraction + UndistortKeyPoints + ComputeImageBounds +
    AssignFeaturesToGrid.  Returns mvKeys, mvKeysUn (n x 7 float), descriptors, bounds, gridInv, grid CSR."""
    lib = lib or slam_lib()
    H, W = im.shape
    im = np.ascontiguousarray(im)
    k, ku = np.zeros((cap, 7), np.float32), np.zeros((cap, 7), np.float32)
    d = np.zeros((cap, 32), np.uint8)
    dist = np.ascontiguousarray(dist, np.float32)
    bounds, ginv = np.zeros(4, np.float32), np.zeros(2, np.float32)
    off, idx = np.zeros(NCELL + 1, np.int32), np.zeros(cap, np.int32)
    n = ctypes.c_int()
    lib.orbslam_mono_frame.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int] + \
                                      [ctypes.c_float] * 4 + [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 5
    lib.orbslam_mono_frame(_p(im), W, H, W, nfeatures, scale, nlevels, ini, mn, fx, fy, cx, cy, _p(dist), len(dist),
                           _p(k), _p(ku), _p(d), cap, _p(bounds), _p(ginv), _p(off), _p(idx), ctypes.byref(n))
    n = n.value
    assert n <= cap
    return dict(kps=k[:n].copy(), kpsUn=ku[:n].copy(), desc=d[:n].copy(), bounds=bounds, gridInv=ginv, gridOff=off, gridIdx=idx[:off[-1]].copy())


def frame_finish(orc, kps7, cam, dist, cols, rows):
    """Restatement of UndistortKeyPoints + ComputeImageBounds + AssignFeaturesToGrid (oracle/match_oracle.cc)."""
    lib = orc.lib
    k = np.ascontiguousarray(kps7, np.float32).reshape(-1, 7)
    n = len(k)
    cam = np.ascontiguousarray(cam, np.float32)
    dist = np.ascontiguousarray(dist, np.float32)
    ku = np.zeros((max(n, 1), 7), np.float32)
    bounds, ginv = np.zeros(4, np.float32), np.zeros(2, np.float32)
    off, idx = np.zeros(NCELL + 1, np.int32), np.zeros(max(n, 1), np.int32)
    lib.mo_frame_finish.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 5
    lib.mo_frame_finish.restype = None
    lib.mo_frame_finish(_p(k), n, _p(cam), _p(dist), len(dist), cols, rows, _p(ku), _p(bounds), _p(ginv), _p(off), _p(idx))
    return dict(kpsUn=ku[:n], bounds=bounds, gridInv=ginv, gridOff=off, gridIdx=idx[:off[-1]].copy())


def _tri_arrays(kf):
    k = np.ascontiguousarray(_kp7(kf["kps"]) if np.asarray(kf["kps"]).dtype.names else kf["kps"], np.float32)
    return (k, np.ascontiguousarray(kf["desc"], np.uint8), np.ascontiguousarray(kf["groups"], np.int32),
            np.ascontiguousarray(kf["has_mp"], np.uint8), np.ascontiguousarray(kf["u_right"], np.float32))


def search_for_triangulation(orc, kf1, kf2, F12, epipole, sf, sigma2, only_stereo, check_ori):
    """Restatement of ORBmatcher::SearchForTriangulation (oracle/match_oracle.cc)."""
    lib = orc.lib
    a, b = _tri_arrays(kf1), _tri_arrays(kf2)
    F = np.ascontiguousarray(F12, np.float32).reshape(9)
    sf, s2 = np.ascontiguousarray(sf, np.float32), np.ascontiguousarray(sigma2, np.float32)
    out = np.full(max(len(a[0]), 1), -1, np.int32)
    lib.mo_search_for_triangulation.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p,
                                                ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    n = lib.mo_search_for_triangulation(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), len(a[0]), _p(b[0]), _p(b[1]), _p(b[2]), _p(b[3]), _p(b[4]), len(b[0]),
                                        _p(F), float(epipole[0]), float(epipole[1]), _p(sf), _p(s2), 1 if only_stereo else 0, 1 if check_ori else 0, _p(out))
    return n, out[:len(a[0])]


def ref_search_for_triangulation(kf1, kf2, Tcw1, Tcw2, F12, only_stereo, check_ori, lib=None):
    """ORBmatcher::SearchForTriangulation of the reference on two real KeyFrames (default camera 500/500/320/240,
    scale factors 1.2^l).  Returns (nmatches, matches12, (ex, ey))."""
    lib = lib or slam_lib()
    a, b = _tri_arrays(kf1), _tri_arrays(kf2)
    F = np.ascontiguousarray(F12, np.float32).reshape(9)
    T1, T2 = np.ascontiguousarray(Tcw1, np.float32).reshape(16), np.ascontiguousarray(Tcw2, np.float32).reshape(16)
    out = np.full(max(len(a[0]), 1), -1, np.int32)
    epi = np.zeros(2, np.float32)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.orbslam_search_for_triangulation.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, ci, vp, vp]
    n = lib.orbslam_search_for_triangulation(_p(a[0]), _p(a[1]), len(a[0]), _p(a[2]), _p(a[3]), _p(a[4]), _p(T1), _p(b[0]), _p(b[1]), len(b[0]), _p(b[2]), _p(b[3]),
                                             _p(b[4]), _p(T2), _p(F), 1 if only_stereo else 0, 1 if check_ori else 0, _p(out), _p(epi))
    return n, out[:len(a[0])], epi


def fuse_best(orc, kf, pts, chi2_gate):
    """Restatement of steps 2-3 of ORBmatcher::Fuse (oracle/match_oracle.cc): (best_idx, best_dist) per map point."""
    lib = orc.lib
    k = np.ascontiguousarray(_kp7(kf["kps"]) if np.asarray(kf["kps"]).dtype.names else kf["kps"], np.float32)
    n = len(k)
    d = np.ascontiguousarray(kf["desc"], np.uint8)
    ur = np.ascontiguousarray(kf["u_right"], np.float32)
    s2 = np.ascontiguousarray(kf["inv_level_sigma2"], np.float32)
    minx, miny = np.float32(kf.get("min_x", 0.0)), np.float32(kf.get("min_y", 0.0))
    maxx, maxy = np.float32(kf.get("max_x", kf["width"])), np.float32(kf.get("max_y", kf["height"]))
    gw, gh = np.float32(64) / (maxx - minx), np.float32(48) / (maxy - miny)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    pu, pv, pur, prad = f32(pts["u"]), f32(pts["v"]), f32(pts["ur"]), f32(pts["radius"])
    m = len(pu)
    lvl = np.ascontiguousarray(pts["level"], np.int32)
    act = np.ascontiguousarray(pts["active"], np.uint8)
    pd = np.ascontiguousarray(pts["desc"], np.uint8)
    bi, bd = np.full(max(m, 1), -1, np.int32), np.full(max(m, 1), 256, np.int32)
    vp, cf, ci = ctypes.c_void_p, ctypes.c_float, ctypes.c_int
    lib.mo_fuse_best.argtypes = [vp, vp, vp, ci, cf, cf, cf, cf, cf, cf, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp, vp]
    lib.mo_fuse_best.restype = None
    lib.mo_fuse_best(_p(k), _p(d), _p(ur), n, minx, miny, float(np.float32(int(minx))), float(np.float32(int(miny))), gw, gh, _p(s2), _p(pu), _p(pv), _p(pur),
                     _p(lvl), _p(prad), _p(act), _p(pd), m, 1 if chi2_gate else 0, _p(bi), _p(bd))
    return bi[:m], bd[:m]


def ref_fuse(overload, kf, Tcw, Scw, holder, exist_obs, src_kps, Tcw_src, cand_pos, cand_desc, cand_obs, lst, th, probe, lib=None):
    """ORBmatcher::Fuse of the reference on a real target KeyFrame / real MapPoints (oracle/refslam_wrap.cc:orbslam_fuse)."""
    lib = lib or slam_lib()
    k = np.ascontiguousarray(_kp7(kf["kps"]) if np.asarray(kf["kps"]).dtype.names else kf["kps"], np.float32)
    n = len(k)
    d = np.ascontiguousarray(kf["desc"], np.uint8)
    ur = np.ascontiguousarray(kf["u_right"], np.float32)
    T, S, Ts = (np.ascontiguousarray(a, np.float32).reshape(16) for a in (Tcw, Scw, Tcw_src))
    holder = np.ascontiguousarray(holder, np.int32)
    exist_obs = np.ascontiguousarray(exist_obs, np.int32)
    sk = np.ascontiguousarray(_kp7(src_kps) if np.asarray(src_kps).dtype.names else src_kps, np.float32)
    cp, cd, co = np.ascontiguousarray(cand_pos, np.float32), np.ascontiguousarray(cand_desc, np.uint8), np.ascontiguousarray(cand_obs, np.int32)
    nc = len(cp)
    lst = np.ascontiguousarray(lst, np.int32)
    probe_idx = np.full(max(nc, 1), -1, np.int32)
    hold_out = np.full(max(n, 1), -1, np.int32)
    bad, repl = np.zeros(max(nc, 1), np.uint8), np.full(max(nc, 1), -1, np.int32)
    rep_pt = np.full(max(len(lst), 1), -1, np.int32)
    prep = np.zeros((max(nc, 1), 6), np.float32)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.orbslam_fuse.argtypes = [ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, ci, ctypes.c_float, ci, vp, vp, vp, vp, vp, vp]
    nf = lib.orbslam_fuse(overload, _p(k), _p(d), _p(ur), n, _p(T), _p(S), _p(holder), len(exist_obs), _p(exist_obs), _p(sk), _p(Ts), _p(cp), _p(cd), _p(co), nc,
                          _p(lst), len(lst), th, 1 if probe else 0, _p(probe_idx), _p(hold_out), _p(bad), _p(repl), _p(rep_pt), _p(prep))
    pts = dict(u=prep[:nc, 0].copy(), v=prep[:nc, 1].copy(), ur=prep[:nc, 2].copy(), level=prep[:nc, 3].astype(np.int32), radius=prep[:nc, 4].copy(),
               active=(prep[:nc, 5] > 0).astype(np.uint8), desc=cd)
    return dict(nfused=nf, probe_idx=probe_idx[:nc], holder=hold_out[:n], bad=bad[:nc], replaced=repl[:nc], replace_point=rep_pt[:len(lst)], points=pts)


def ref_search_by_sim3(kf1, pos1, Tcw1, kf2, pos2, Tcw2, pre12, s12, R12, t12, th, lib=None):
    """ORBmatcher::SearchBySim3 of the reference on two real KeyFrames whose features all hold MapPoints."""
    lib = lib or slam_lib()
    k1 = np.ascontiguousarray(_kp7(kf1["kps"]), np.float32)
    k2 = np.ascontiguousarray(_kp7(kf2["kps"]), np.float32)
    d1, d2 = np.ascontiguousarray(kf1["desc"], np.uint8), np.ascontiguousarray(kf2["desc"], np.uint8)
    p1, p2 = np.ascontiguousarray(pos1, np.float32), np.ascontiguousarray(pos2, np.float32)
    T1, T2 = np.ascontiguousarray(Tcw1, np.float32).reshape(16), np.ascontiguousarray(Tcw2, np.float32).reshape(16)
    pre = np.ascontiguousarray(pre12, np.int32)
    R, t = np.ascontiguousarray(R12, np.float32).reshape(9), np.ascontiguousarray(t12, np.float32).reshape(3)
    out = np.full(max(len(k1), 1), -1, np.int32)
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.orbslam_search_by_sim3.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, ci, vp, vp, cf, vp, vp, cf, vp]
    n = lib.orbslam_search_by_sim3(_p(k1), _p(d1), _p(p1), len(k1), _p(T1), _p(k2), _p(d2), _p(p2), len(k2), _p(T2), _p(pre), s12, _p(R), _p(t), th, _p(out))
    return n, out[:len(k1)]


def area_search_greedy(orc, frame, queries, max_dist):
    """Restatement of the greedy area search (oracle/match_oracle.cc:mo_area_search_greedy)."""
    lib = orc.lib
    k = np.ascontiguousarray(_kp7(frame["kps"]) if np.asarray(frame["kps"]).dtype.names else frame["kps"], np.float32)
    n = len(k)
    d = np.ascontiguousarray(frame["desc"], np.uint8)
    blk = np.ascontiguousarray(frame["blocked"], np.uint8)
    minx, miny = np.float32(frame.get("min_x", 0.0)), np.float32(frame.get("min_y", 0.0))
    maxx, maxy = np.float32(frame.get("max_x", frame["width"])), np.float32(frame.get("max_y", frame["height"]))
    gw, gh = np.float32(64) / (maxx - minx), np.float32(48) / (maxy - miny)
    wx, wy = (np.float32(int(minx)), np.float32(int(miny))) if queries.get("window_int_bounds") else (minx, miny)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    qu, qv, qr = f32(queries["u"]), f32(queries["v"]), f32(queries["radius"])
    m = len(qu)
    lo, hi = np.ascontiguousarray(queries["min_level"], np.int32), np.ascontiguousarray(queries["max_level"], np.int32)
    act = np.ascontiguousarray(queries["active"], np.uint8)
    qd = np.ascontiguousarray(queries["desc"], np.uint8)
    asg, dst = np.full(max(m, 1), -1, np.int32), np.full(max(m, 1), 256, np.int32)
    vp, cf, ci = ctypes.c_void_p, ctypes.c_float, ctypes.c_int
    lib.mo_area_search_greedy.argtypes = [vp, vp, vp, ci, cf, cf, cf, cf, cf, cf, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp, vp]
    nm = lib.mo_area_search_greedy(_p(k), _p(d), _p(blk), n, minx, miny, wx, wy, gw, gh, _p(qu), _p(qv), _p(qr), _p(lo), _p(hi), _p(act), _p(qd), m, int(max_dist),
                                   _p(asg), _p(dst))
    return nm, asg[:m], dst[:m]


def search_for_initialization(orc, f1, f2, prev, window, nnratio, check_ori):
    """Restatement of ORBmatcher::SearchForInitialization (oracle/match_oracle.cc); bounds 0..640 x 0..480."""
    lib = orc.lib
    k1, k2 = np.ascontiguousarray(_kp7(f1["kps"]), np.float32), np.ascontiguousarray(_kp7(f2["kps"]), np.float32)
    d1, d2 = np.ascontiguousarray(f1["desc"], np.uint8), np.ascontiguousarray(f2["desc"], np.uint8)
    prev = np.ascontiguousarray(prev, np.float32).reshape(-1, 2)
    out = np.full(max(len(k1), 1), -1, np.int32)
    vp, cf, ci = ctypes.c_void_p, ctypes.c_float, ctypes.c_int
    lib.mo_search_for_initialization.argtypes = [vp, vp, ci, vp, vp, ci, cf, cf, cf, cf, vp, ci, cf, ci, vp]
    n = lib.mo_search_for_initialization(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), 0.0, 0.0, np.float32(64) / np.float32(640), np.float32(48) / np.float32(480),
                                         _p(prev), window, nnratio, 1 if check_ori else 0, _p(out))
    return n, out[:len(k1)]


def ref_search_for_initialization(f1, f2, prev, window, nnratio, check_ori, lib=None):
    lib = lib or slam_lib()
    k1, k2 = np.ascontiguousarray(_kp7(f1["kps"]), np.float32), np.ascontiguousarray(_kp7(f2["kps"]), np.float32)
    d1, d2 = np.ascontiguousarray(f1["desc"], np.uint8), np.ascontiguousarray(f2["desc"], np.uint8)
    prev = np.ascontiguousarray(prev, np.float32).reshape(-1, 2).copy()
    out = np.full(max(len(k1), 1), -1, np.int32)
    vp, cf, ci = ctypes.c_void_p, ctypes.c_float, ctypes.c_int
    lib.orbslam_search_for_initialization.argtypes = [vp, vp, ci, vp, vp, ci, vp, ci, cf, ci, vp]
    n = lib.orbslam_search_for_initialization(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), _p(prev), window, nnratio, 1 if check_ori else 0, _p(out))
    return n, out[:len(k1)], prev


def ref_is_in_frustum(Tcw, Tcw_src, src_kps, pos, cos_limit, lib=None):
    """Frame::isInFrustum of the reference per map point (points created from a source frame)."""
    lib = lib or slam_lib()
    sk = np.ascontiguousarray(_kp7(src_kps) if np.asarray(src_kps).dtype.names else src_kps, np.float32)
    pos = np.ascontiguousarray(pos, np.float32)
    n = len(pos)
    T, Ts = np.ascontiguousarray(Tcw, np.float32).reshape(16), np.ascontiguousarray(Tcw_src, np.float32).reshape(16)
    iv = np.zeros(max(n, 1), np.uint8)
    f = lambda k=1: np.zeros((max(n, 1), k) if k > 1 else max(n, 1), np.float32)
    px, py, pxr, vc, nrm, mx, mn = f(), f(), f(), f(), f(3), f(), f()
    lvl = np.zeros(max(n, 1), np.int32)
    lsf = ctypes.c_float()
    vp = ctypes.c_void_p
    lib.orbslam_is_in_frustum.argtypes = [vp, vp, vp, vp, ctypes.c_int, ctypes.c_float] + [vp] * 10
    nin = lib.orbslam_is_in_frustum(_p(T), _p(Ts), _p(sk), _p(pos), n, cos_limit, _p(iv), _p(px), _p(py), _p(pxr), _p(lvl), _p(vc), _p(nrm), _p(mx), _p(mn), ctypes.byref(lsf))
    return dict(n_in=nin, in_view=iv[:n], proj_x=px[:n], proj_y=py[:n], proj_xr=pxr[:n], level=lvl[:n], view_cos=vc[:n], normal=nrm[:n], max_distance=mx[:n],
                min_distance=mn[:n], log_scale_factor=lsf.value)


# ---- the reference's Optimizer (src/Optimizer.cc + vendored g2o, unmodified, on oracle/eigenshim) on real Map / Frame objects,
# ---- or shim/Optimizer_hip.cc when `lib` is the drop-in build; both through oracle/refslam_wrap.cc
SCALE_FACTORS = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)


def octaves_of(inv_s2):
    return np.rint(np.log(1.0 / np.asarray(inv_s2, np.float64)) / (2 * np.log(1.2))).astype(np.int32)


def _map_arrays(w):
    octv = octaves_of(w["edge_inv_sigma2"])
    obs6 = np.ascontiguousarray(np.stack([w["edge_point"], w["edge_kf"], w["edge_obs"][:, 0], w["edge_obs"][:, 1], w["edge_obs"][:, 2], octv], 1), np.float32)
    return octv, obs6, np.ascontiguousarray(w["intr"][0], np.float32)


def ref_local_ba_on_map(w, ref_kf, lib=None):
    """Optimizer::LocalBundleAdjustment(kfs[ref_kf], &stop, &map) on a real Map built from the window `w`
    (oracle/refslam_wrap.cc: orbslam_local_ba).  Returns poses (K x 16 f32), points (P x 3 f32), erased (E), role (K)."""
    lib = lib or slam_lib()
    K, P, E = w["K"], w["P"], w["E"]
    octv, obs6, cam5 = _map_arrays(w)
    poses_out, points_out = np.zeros((K, 16), np.float32), np.zeros((P, 3), np.float32)
    erased, role = np.zeros(E, np.uint8), np.zeros(K, np.uint8)
    lib.orbslam_local_ba.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
    lib.orbslam_local_ba(K, _p(w["poses"]), _p(cam5), P, _p(w["points"]), E, _p(obs6), int(ref_kf), _p(SCALE_FACTORS), 8, 640, 480, _p(poses_out), _p(points_out),
                         _p(erased), _p(role))
    return dict(poses=poses_out, points=points_out, erased=erased, role=role)


def local_window_of(w, ref_kf):
    """The reference's window rule re-derived on the arrays (KeyFrame::UpdateConnections th = 15, src/Optimizer.cc:629-697): local keyframes =
    ref_kf + keyframes sharing >= 15 points with it (or the best one), local points = everything they see, fixed = other observers.
    Returns (role K, flattened problem for the restatement / the C ABI, kf_list, pt_list, sel = edges inside the window)."""
    K, P = w["K"], w["P"]
    seen = np.zeros((K, P), bool)
    seen[w["edge_kf"], w["edge_point"]] = True
    shared = (seen & seen[ref_kf]).sum(1)
    shared[ref_kf] = 0
    neigh = shared >= 15 if (shared >= 15).any() else (shared == shared.max())
    local = neigh.copy(); local[ref_kf] = True
    local_pts = seen[local].any(0)
    fixed_kf = seen[:, local_pts].any(1) & ~local
    role = np.where(local, 1, np.where(fixed_kf, 2, 0))
    kf_list = list(np.flatnonzero(local)) + list(np.flatnonzero(fixed_kf))
    kf_new = {int(k): i for i, k in enumerate(kf_list)}
    pt_list = np.flatnonzero(local_pts)
    pt_new = -np.ones(P, np.int64); pt_new[pt_list] = np.arange(len(pt_list))
    sel = local_pts[w["edge_point"]]
    octv = octaves_of(w["edge_inv_sigma2"])
    inv = (np.float32(1.0) / (SCALE_FACTORS[octv] * SCALE_FACTORS[octv])).astype(np.float32)       # Frame::mvInvLevelSigma2 as fill_frame builds it
    fx = np.array([1 if (k == 0 or fixed_kf[k]) else 0 for k in kf_list], np.uint8)
    prob = dict(K=len(kf_list), P=len(pt_list), E=int(sel.sum()), poses=np.ascontiguousarray(w["poses"][kf_list]), fixed=fx,
                intr=np.ascontiguousarray(w["intr"][kf_list]), points=np.ascontiguousarray(w["points"][pt_list]),
                edge_point=np.ascontiguousarray(pt_new[w["edge_point"][sel]].astype(np.int32)),
                edge_kf=np.array([kf_new[int(k)] for k in w["edge_kf"][sel]], np.int32),
                edge_obs=np.ascontiguousarray(w["edge_obs"][sel]), edge_inv_sigma2=np.ascontiguousarray(inv[sel]))
    return role, prob, kf_list, pt_list, sel


def ref_global_ba_on_map(w, iters, robust, loop_kf, lib=None):
    """Optimizer::GlobalBundleAdjustemnt(&map, iters, NULL, loop_kf, robust) on a real Map (orbslam_global_ba)."""
    lib = lib or slam_lib()
    K, P, E = w["K"], w["P"], w["E"]
    octv, obs6, cam5 = _map_arrays(w)
    poses_out, points_out = np.zeros((K, 16), np.float32), np.zeros((P, 3), np.float32)
    untouched = ctypes.c_int(0)
    ci, vp = ctypes.c_int, ctypes.c_void_p
    lib.orbslam_global_ba.argtypes = [ci, vp, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp]
    lib.orbslam_global_ba(K, _p(w["poses"]), _p(cam5), P, _p(w["points"]), E, _p(obs6), _p(SCALE_FACTORS), 8, 640, 480, int(iters), 1 if robust else 0, int(loop_kf),
                          _p(poses_out), _p(points_out), ctypes.byref(untouched))
    return dict(poses=poses_out, points=points_out, untouched=untouched.value)


def global_problem_of(w):
    """The graph GlobalBundleAdjustemnt builds from that Map, for the restatement: keyframe 0 fixed, float32 information of the octave."""
    w2 = dict(w)
    fixed = np.zeros(w["K"], np.uint8); fixed[0] = 1
    octv = octaves_of(w["edge_inv_sigma2"])
    w2["fixed"] = fixed
    w2["edge_inv_sigma2"] = (np.float32(1.0) / (SCALE_FACTORS[octv] * SCALE_FACTORS[octv])).astype(np.float32)
    return w2


def ref_pose_optimization_on_frame(fr, lib=None):
    """Optimizer::PoseOptimization(&F) on a real Frame whose features carry MapPoints (orbslam_pose_optimization).
    fr["inv_sigma2"] is replaced by what the Frame holds for the octave (float32 1 / sf^2)."""
    lib = lib or slam_lib()
    n = len(fr["Xw"])
    octv = octaves_of(fr["inv_sigma2"])
    fr["inv_sigma2"] = (np.float32(1.0) / (SCALE_FACTORS[octv] * SCALE_FACTORS[octv])).astype(np.float32)
    kobs = np.ascontiguousarray(np.concatenate([fr["obs"], octv[:, None].astype(np.float32)], 1), np.float32)
    pose = np.ascontiguousarray(np.asarray(fr["pose"], np.float32).reshape(16))
    cam5 = np.ascontiguousarray(fr["cam"], np.float32)
    out, outl = np.zeros(16, np.float32), np.zeros(max(n, 1), np.uint8)
    lib.orbslam_pose_optimization.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    ret = lib.orbslam_pose_optimization(_p(pose), _p(cam5), n, _p(np.ascontiguousarray(fr["Xw"], np.float32)), _p(kobs), _p(SCALE_FACTORS), 8, 640, 480, _p(out), _p(outl))
    return dict(pose=out.reshape(4, 4), outlier=outl[:n], inliers=ret)


def _ba_f64(fn, w, iters1, robust1, second_stage):
    K, P, E = w["K"], w["P"], w["E"]
    poses_d, points_d = np.zeros((K, 12), np.float64), np.zeros((P, 3), np.float64)
    chi2, outl = np.zeros(E, np.float64), np.zeros(E, np.uint8)
    return K, P, E, poses_d, points_d, chi2, outl


def g2o_ba_f64(w, iters1=5, robust1=True, second_stage=True):
    """The vendored g2o driven directly (oracle/refslam_wrap.cc: orbslam_g2o_ba): FP64 poses (R, t), points, per-edge chi2."""
    lib = slam_lib()
    K, P, E, poses_d, points_d, chi2, outl = _ba_f64(None, w, iters1, robust1, second_stage)
    iters = np.zeros(2, np.int32)
    ci, vp = ctypes.c_int, ctypes.c_void_p
    lib.orbslam_g2o_ba.argtypes = [ci, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp]
    lib.orbslam_g2o_ba(K, _p(w["poses"]), _p(w["fixed"]), _p(w["intr"]), P, _p(w["points"]), E, _p(w["edge_point"]), _p(w["edge_kf"]), _p(w["edge_obs"]),
                       _p(w["edge_inv_sigma2"]), int(iters1), 1 if robust1 else 0, 1 if second_stage else 0, _p(poses_d), _p(points_d), _p(chi2), _p(outl), _p(iters))
    return dict(poses=poses_d, points=points_d, chi2=chi2, outlier=outl, iters=iters)


def ba_f64(orc, w, iters1=5, robust1=True, second_stage=True):
    """The restatement with its FP64 state exposed (oracle/lba_oracle.cc: lo_ba_f64)."""
    lib = orc.lib
    K, P, E, poses_d, points_d, chi2, outl = _ba_f64(None, w, iters1, robust1, second_stage)
    stats = np.zeros(8, np.float64)
    ci, vp = ctypes.c_int, ctypes.c_void_p
    lib.lo_ba_f64.argtypes = [ci, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp]
    lib.lo_ba_f64(K, _p(w["poses"]), _p(w["fixed"]), _p(w["intr"]), P, _p(w["points"]), E, _p(w["edge_point"]), _p(w["edge_kf"]), _p(w["edge_obs"]),
                  _p(w["edge_inv_sigma2"]), int(iters1), 1 if robust1 else 0, 1 if second_stage else 0, _p(poses_d), _p(points_d), _p(chi2), _p(outl), _p(stats))
    return dict(poses=poses_d, points=points_d, chi2=chi2, outlier=outl, iters=stats[[0, 4]].astype(np.int32), stats=stats)
