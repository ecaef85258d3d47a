"""MI355X-native ORB-SLAM2 hot path: Python host side over the C ABI of liborbx.so.

The product is the C-ABI library (include/orbx.h, csrc/*.hip).  This module is the thin
ctypes mirror of the reference's class surfaces used by tests and bench.py:

    ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)   include/ORBextractor.h:92
        .__call__(image) -> (keypoints[n] structured, descriptors[n,32])    ORBextractor.h:110
        .GetLevels() / GetScaleFactors() / ...                               ORBextractor.h:118-158
        .mvImagePyramid(level)                                               ORBextractor.h:161

There is no CPU fallback: constructing an extractor without a HIP device raises.
"""
import ctypes
import importlib.util
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
_spec = importlib.util.spec_from_file_location("orbx_build", _PKG / "build.py")
build_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(build_mod)

ORBX_OK = 0
ERR_NAMES = {-1: "ORBX_ERR_ARG", -2: "ORBX_ERR_HIP", -3: "ORBX_ERR_CAPACITY", -4: "ORBX_ERR_NODEVICE", -5: "ORBX_ERR_STATE"}

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28

SYNTH_LOW_TEXTURE = 1
SYNTH_STEREO_RIGHT = 2


class OrbxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "?"), code, msg))
        self.code = code


class ExtractorConfig(ctypes.Structure):
    _fields_ = [("nfeatures", ctypes.c_int), ("scale_factor", ctypes.c_float), ("nlevels", ctypes.c_int),
                ("ini_th_fast", ctypes.c_int), ("min_th_fast", ctypes.c_int),
                ("max_width", ctypes.c_int), ("max_height", ctypes.c_int), ("max_batch", ctypes.c_int),
                ("device", ctypes.c_int), ("gauss_taps", ctypes.c_uint16 * 7), ("reserved_", ctypes.c_uint16)]


_lib = None


def lib_path():
    return build_mod.LIB


def load_library():
    """dlopen liborbx.so (building it first when a compiler is present)."""
    global _lib
    if _lib is not None:
        return _lib
    path = build_mod.build_liborbx(verbose=False)
    L = ctypes.CDLL(str(path))
    L.orbx_last_error.restype = ctypes.c_char_p
    L.orbx_stage_name.restype = ctypes.c_char_p
    L.orbx_stage_name.argtypes = [ctypes.c_int]
    L.orbx_synth_frame.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.orbx_extractor_create.argtypes = [ctypes.POINTER(ExtractorConfig), ctypes.POINTER(vp)]
    L.orbx_extractor_destroy.argtypes = [vp]
    L.orbx_extractor_destroy.restype = None
    L.orbx_extractor_tables.argtypes = [vp] + [vp] * 6
    L.orbx_extractor_capacity.argtypes = [vp]
    L.orbx_extract.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, vp]
    L.orbx_extract_batch.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, ci, vp]
    L.orbx_extract_batch_device.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.c_size_t]
    L.orbx_batch_results_device.argtypes = [vp, vp, vp, vp, vp]
    L.orbx_batch_download.argtypes = [vp, ci, vp, vp, ci, vp]
    L.orbx_upload_frames.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, vp]
    L.orbx_extractor_sync.argtypes = [vp]
    L.orbx_pyramid_level_size.argtypes = [vp, ci, ci, ci, vp, vp]
    L.orbx_download_pyramid.argtypes = [vp, ci, ci, ci, vp, ci]
    L.orbx_debug_download_scores.argtypes = [vp, ci, ci, vp, ci]
    L.orbx_debug_download_candidates.argtypes = [vp, ci, ci, vp, ci, vp]
    L.orbx_debug_download_level_keypoints.argtypes = [vp, ci, ci, vp, ci, vp]
    L.orbx_extractor_set_profiling.argtypes = [vp, ci]
    L.orbx_extractor_last_timing.argtypes = [vp, vp, vp, vp]
    _lib = L
    return L


def _check(rc):
    if rc != ORBX_OK:
        raise OrbxError(rc, load_library().orbx_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def synth_frame(seed, width, height, flags=0):
    """Deterministic synthetic grayscale frame (orbx_synth_frame)."""
    im = np.empty((height, width), np.uint8)
    _check(load_library().orbx_synth_frame(ctypes.c_uint64(seed), width, height, width, flags, _ptr(im)))
    return im


class ORBextractor:
    """Mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:92-161)."""

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_width=1280, max_height=1024,
                 max_batch=1, device=0, gauss_taps=None):
        self._L = load_library()
        cfg = ExtractorConfig(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_width, max_height, max_batch, device)
        if gauss_taps is not None:
            for i in range(7):
                cfg.gauss_taps[i] = int(gauss_taps[i])
        self._h = ctypes.c_void_p()
        _check(self._L.orbx_extractor_create(ctypes.byref(cfg), ctypes.byref(self._h)))
        self.nlevels = nlevels
        self.scaleFactor = float(np.float32(scaleFactor))
        self.max_batch = max_batch
        self.capacity = self._L.orbx_extractor_capacity(self._h)
        self._last_size = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbx_extractor_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- getters (ORBextractor.h:118-158) ---
    def _tables(self):
        nl = self.nlevels
        t = [np.zeros(nl, np.float32) for _ in range(4)]
        q = np.zeros(nl, np.int32)
        n = ctypes.c_int()
        _check(self._L.orbx_extractor_tables(self._h, ctypes.byref(n), _ptr(t[0]), _ptr(t[1]), _ptr(t[2]), _ptr(t[3]), _ptr(q)))
        return t, q

    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self.scaleFactor

    def GetScaleFactors(self):
        return self._tables()[0][0]

    def GetInverseScaleFactors(self):
        return self._tables()[0][1]

    def GetScaleSigmaSquares(self):
        return self._tables()[0][2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[0][3]

    def features_per_level(self):
        return self._tables()[1]

    # --- operator() ---
    def __call__(self, image, mask=None):
        """(keypoints, descriptors) of one CV_8UC1 image; `mask` is ignored like in the reference."""
        if image is None or image.size == 0:
            return np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2
        kps, desc, counts = self.extract_batch([image])
        n = int(counts[0])
        return kps[0, :n].copy(), desc[0, :n].copy()

    def extract_batch(self, images):
        B = len(images)
        H, W = images[0].shape
        imgs = [np.ascontiguousarray(im) for im in images]
        arr = (ctypes.c_void_p * B)(*[im.ctypes.data for im in imgs])
        cap = self.capacity
        kps = np.zeros((B, cap), KEYPOINT_DTYPE)
        desc = np.zeros((B, cap, 32), np.uint8)
        counts = np.zeros(B, np.int32)
        _check(self._L.orbx_extract_batch(self._h, arr, B, W, H, W, _ptr(kps), _ptr(desc), cap, _ptr(counts)))
        self._last_size = (W, H)
        return kps, desc, counts

    # --- device-resident path used by bench.py ---
    def upload(self, images):
        B = len(images)
        H, W = images[0].shape
        imgs = [np.ascontiguousarray(im) for im in images]
        arr = (ctypes.c_void_p * B)(*[im.ctypes.data for im in imgs])
        dev = ctypes.c_void_p()
        stride = ctypes.c_int()
        fp = ctypes.c_size_t()
        _check(self._L.orbx_upload_frames(self._h, arr, B, W, H, W, ctypes.byref(dev), ctypes.byref(stride), ctypes.byref(fp)))
        return dev, stride.value, fp.value, (B, W, H)

    def run_device(self, dev, stride, frame_pitch, shape):
        B, W, H = shape
        _check(self._L.orbx_extract_batch_device(self._h, dev, B, W, H, stride, frame_pitch))
        self._last_size = (W, H)

    def sync(self):
        _check(self._L.orbx_extractor_sync(self._h))

    def download(self, batch):
        cap = self.capacity
        kps = np.zeros((batch, cap), KEYPOINT_DTYPE)
        desc = np.zeros((batch, cap, 32), np.uint8)
        counts = np.zeros(batch, np.int32)
        _check(self._L.orbx_batch_download(self._h, batch, _ptr(kps), _ptr(desc), cap, _ptr(counts)))
        return kps, desc, counts

    def results_device(self):
        k, d, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        cap = ctypes.c_int()
        _check(self._L.orbx_batch_results_device(self._h, ctypes.byref(k), ctypes.byref(d), ctypes.byref(c), ctypes.byref(cap)))
        return k, d, c, cap.value

    def set_profiling(self, on):
        _check(self._L.orbx_extractor_set_profiling(self._h, 1 if on else 0))

    def last_timing(self):
        tot = ctypes.c_float()
        st = (ctypes.c_float * 16)()
        n = ctypes.c_int()
        _check(self._L.orbx_extractor_last_timing(self._h, ctypes.byref(tot), st, ctypes.byref(n)))
        return tot.value, {self._L.orbx_stage_name(i).decode(): st[i] for i in range(n.value)}

    # --- mvImagePyramid and the stage taps ---
    def level_size(self, level, size=None):
        W, H = size or self._last_size
        w, h = ctypes.c_int(), ctypes.c_int()
        _check(self._L.orbx_pyramid_level_size(self._h, W, H, level, ctypes.byref(w), ctypes.byref(h)))
        return w.value, h.value

    def mvImagePyramid(self, level, frame=0, blurred=False):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        _check(self._L.orbx_download_pyramid(self._h, frame, level, 1 if blurred else 0, _ptr(out), w))
        return out

    def debug_scores(self, level, frame=0):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        _check(self._L.orbx_debug_download_scores(self._h, frame, level, _ptr(out), w))
        return out

    def debug_candidates(self, level, frame=0, cap=1 << 16):
        out = np.zeros(cap, np.uint32)
        n = ctypes.c_int()
        _check(self._L.orbx_debug_download_candidates(self._h, frame, level, _ptr(out), cap, ctypes.byref(n)))
        return out[:min(n.value, cap)].copy(), n.value

    def debug_level_keypoints(self, level, frame=0):
        out = np.zeros(4096, KEYPOINT_DTYPE)
        n = ctypes.c_int()
        _check(self._L.orbx_debug_download_level_keypoints(self._h, frame, level, _ptr(out), 4096, ctypes.byref(n)))
        return out[:n.value].copy()
