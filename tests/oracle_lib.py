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
