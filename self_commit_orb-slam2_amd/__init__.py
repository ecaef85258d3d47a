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
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.orbx_extractor_create.argtypes = [ctypes.POINTER(ExtractorConfig), ctypes.POINTER(vp)]
    L.orbx_extractor_destroy.argtypes = [vp]
    L.orbx_extractor_destroy.restype = None
    L.orbx_extractor_tables.argtypes = [vp] + [vp] * 6
    L.orbx_extractor_capacity.argtypes = [vp]
    L.orbx_extract.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, vp]
    L.orbx_extract_batch.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, ci, vp]
    L.orbx_extract_batch_begin.argtypes = [vp, vp, ci, ci, ci, ci]
    L.orbx_extract_batch_end.argtypes = [vp, vp, vp, ci, vp]
    L.orbx_extract_view_pyramid.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp, vp]
    L.orbx_extractor_expect_partner.argtypes = [vp, vp]
    L.orbx_combiner_stats.argtypes = [vp, vp, vp, vp]
    L.orbx_extract_batch_device.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.c_size_t]
    L.orbx_batch_results_device.argtypes = [vp, vp, vp, vp, vp]
    L.orbx_batch_download.argtypes = [vp, ci, vp, vp, ci, vp]
    L.orbx_upload_frames.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, vp]
    L.orbx_extractor_sync.argtypes = [vp]
    L.orbx_extractor_status.argtypes = [vp, vp]
    L.orbx_batch_status_device.argtypes = [vp, vp, vp]
    L.orbx_pyramid_level_size.argtypes = [vp, ci, ci, ci, vp, vp]
    L.orbx_download_pyramid.argtypes = [vp, ci, ci, ci, vp, ci]
    L.orbx_debug_download_scores.argtypes = [vp, ci, ci, vp, ci]
    L.orbx_extractor_set_debug_taps.argtypes = [vp, ci]
    L.orbx_debug_download_candidates.argtypes = [vp, ci, ci, vp, ci, vp]
    L.orbx_debug_download_level_keypoints.argtypes = [vp, ci, ci, vp, ci, vp]
    L.orbx_extractor_set_profiling.argtypes = [vp, ci]
    L.orbx_extractor_last_timing.argtypes = [vp, vp, vp, vp]
    _lib = L
    return L


_synth = None


def _synth_lib():
    global _synth
    if _synth is None:
        _synth = ctypes.CDLL(str(build_mod.build_synth(verbose=False)))
        _synth.orbx_synth_frame_ex.argtypes = [ctypes.c_uint64] + [ctypes.c_int] * 7 + [ctypes.c_void_p]
    return _synth


def _check(rc):
    if rc != ORBX_OK:
        raise OrbxError(rc, load_library().orbx_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def synth_frame(seed, width, height, flags=0, view=0, dx=0, dy=0):
    """Deterministic synthetic grayscale frame (synth/orbx_synth.cc in liborbx_synth.so: a test / bench helper, not part of liborbx.so)."""
    im = np.empty((height, width), np.uint8)
    if _synth_lib().orbx_synth_frame_ex(ctypes.c_uint64(seed), view, dx, dy, width, height, width, flags, _ptr(im)) != ORBX_OK:
        raise ValueError("bad synthetic frame arguments")
    return im


def synth_sequence(first_seed, count, width, height, views_per_scene=16, step=(3, 1), low_texture_every=16):
    """`count` frames: scenes of `views_per_scene` consecutive views translating by `step` px/view
    (so consecutive frames share corners); every `low_texture_every`-th scene view is low texture."""
    out = []
    for i in range(count):
        scene, view = divmod(i, views_per_scene)
        flags = SYNTH_LOW_TEXTURE if (low_texture_every and i % low_texture_every == low_texture_every - 1) else 0
        out.append(synth_frame(first_seed + scene, width, height, flags, view, view * step[0], view * step[1]))
    return out


class HostPyramid(ctypes.Structure):
    """orbx_host_pyramid (include/orbx.h)."""
    _fields_ = [("level", ctypes.c_void_p * 12), ("width", ctypes.c_int * 12), ("height", ctypes.c_int * 12), ("stride", ctypes.c_int * 12), ("nlevels", ctypes.c_int)]


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

    def extract_with_pyramid(self, image):
        """operator() + the host pyramid of the same call (orbx_extract_view_pyramid): (keypoints, descriptors, [level 0, level 1, ...]);
        the arrays are copies of the handle's pinned views."""
        image = np.ascontiguousarray(image)
        H, W = image.shape
        kp, dp, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int()
        pyr = HostPyramid()
        _check(self._L.orbx_extract_view_pyramid(self._h, _ptr(image), W, H, W, ctypes.byref(kp), ctypes.byref(dp), ctypes.byref(n), ctypes.byref(pyr)))
        self._last_size = (W, H)
        cnt = n.value
        kps = np.ctypeslib.as_array(ctypes.cast(kp, ctypes.POINTER(ctypes.c_uint8)), shape=(cnt * KEYPOINT_DTYPE.itemsize,)).view(KEYPOINT_DTYPE).copy() if cnt else np.zeros(0, KEYPOINT_DTYPE)
        desc = np.ctypeslib.as_array(ctypes.cast(dp, ctypes.POINTER(ctypes.c_uint8)), shape=(cnt, 32)).copy() if cnt else np.zeros((0, 32), np.uint8)
        levels = []
        for l in range(pyr.nlevels):
            w, h, st = pyr.width[l], pyr.height[l], pyr.stride[l]
            buf = np.ctypeslib.as_array(ctypes.cast(pyr.level[l], ctypes.POINTER(ctypes.c_uint8)), shape=((h - 1) * st + w,))
            levels.append(np.lib.stride_tricks.as_strided(buf, shape=(h, w), strides=(st, 1)).copy())
        return kps, desc, levels

    def expect_partner(self, other):
        """One-shot hint: `other`'s single-frame call is about to arrive on another thread (orbx_extractor_expect_partner)."""
        _check(self._L.orbx_extractor_expect_partner(self._h, other._h if other is not None else None))

    def combiner_stats(self):
        """(launch sets, frames, engines) served so far for this handle's configuration and image size."""
        b, f, e = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        _check(self._L.orbx_combiner_stats(self._h, ctypes.byref(b), ctypes.byref(f), ctypes.byref(e)))
        return b.value, f.value, e.value

    def extract_batch(self, images, out=None):
        """out = (kps, desc, counts) of an earlier call: the result arrays are reused (a caller in a loop; allocating and zeroing 60 bytes x
        capacity x batch per call costs as much as the call)."""
        B = len(images)
        H, W = images[0].shape
        imgs = [np.ascontiguousarray(im) for im in images]
        arr = (ctypes.c_void_p * B)(*[im.ctypes.data for im in imgs])
        cap = self.capacity
        if out is not None and out[0].shape == (B, cap) and out[1].shape == (B, cap, 32) and out[2].shape == (B,):
            kps, desc, counts = out
        else:
            kps = np.zeros((B, cap), KEYPOINT_DTYPE)
            desc = np.zeros((B, cap, 32), np.uint8)
            counts = np.zeros(B, np.int32)
        _check(self._L.orbx_extract_batch(self._h, arr, B, W, H, W, _ptr(kps), _ptr(desc), cap, _ptr(counts)))
        self._last_size = (W, H)
        return kps, desc, counts

    # --- the two-deep pipeline of host batches (orbx_extract_batch_begin / _end) ---
    def extract_batch_begin(self, images):
        """Stage + upload + launch set + read-back of one batch, nothing waited for; at most two batches between a begin and its end."""
        B = len(images)
        H, W = images[0].shape
        arr = (ctypes.c_void_p * B)(*[im.ctypes.data for im in images])      # (the caller keeps `images` alive and contiguous until the call returns)
        _check(self._L.orbx_extract_batch_begin(self._h, arr, B, W, H, images[0].strides[0]))
        self._last_size = (W, H)
        self._pipe_batches = getattr(self, "_pipe_batches", []) + [B]

    def extract_batch_end(self, out=None):
        """Results of the OLDEST begun batch, as extract_batch returns them."""
        if not getattr(self, "_pipe_batches", None):
            raise OrbxError(-5, "no batch has been begun")
        B = self._pipe_batches.pop(0)
        cap = self.capacity
        if out is not None and out[0].shape == (B, cap) and out[1].shape == (B, cap, 32) and out[2].shape == (B,):
            kps, desc, counts = out
        else:
            kps = np.zeros((B, cap), KEYPOINT_DTYPE)
            desc = np.zeros((B, cap, 32), np.uint8)
            counts = np.zeros(B, np.int32)
        _check(self._L.orbx_extract_batch_end(self._h, _ptr(kps), _ptr(desc), cap, _ptr(counts)))
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

    def status(self):
        """Capacity bits of the last batch (0 = complete), without downloading it: orbx_extractor_status."""
        bits = ctypes.c_int32()
        _check(self._L.orbx_extractor_status(self._h, ctypes.byref(bits)))
        return bits.value

    def set_debug_taps(self, on):
        _check(self._L.orbx_extractor_set_debug_taps(self._h, 1 if on else 0))

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


# =====================================================================================
# ORBmatcher (Hamming paths) over the C ABI
# =====================================================================================
class FeatureSet(ctypes.Structure):
    _fields_ = [("keypoints", ctypes.c_void_p), ("descriptors", ctypes.c_void_p), ("counts", ctypes.c_void_p),
                ("groups", ctypes.c_void_p), ("valid", ctypes.c_void_p), ("capacity", ctypes.c_int), ("nframes", ctypes.c_int)]


class ProjectionFrame(ctypes.Structure):
    _fields_ = [("keypoints_un", ctypes.c_void_p), ("descriptors", ctypes.c_void_p), ("u_right", ctypes.c_void_p), ("occupied", ctypes.c_void_p),
                ("counts", ctypes.c_void_p), ("capacity", ctypes.c_int), ("nframes", ctypes.c_int), ("min_x", ctypes.c_float), ("min_y", ctypes.c_float),
                ("grid_width_inv", ctypes.c_float), ("grid_height_inv", ctypes.c_float)]


class ProjectionPoints(ctypes.Structure):
    _fields_ = [("proj_x", ctypes.c_void_p), ("proj_y", ctypes.c_void_p), ("proj_xr", ctypes.c_void_p), ("scale_level", ctypes.c_void_p),
                ("view_cos", ctypes.c_void_p), ("in_view", ctypes.c_void_p), ("has_observations", ctypes.c_void_p), ("descriptors", ctypes.c_void_p),
                ("counts", ctypes.c_void_p), ("capacity", ctypes.c_int)]


class ProjectionLast(ctypes.Structure):
    _fields_ = [("valid", ctypes.c_void_p), ("world_pos", ctypes.c_void_p), ("descriptors", ctypes.c_void_p), ("has_observations", ctypes.c_void_p),
                ("octave", ctypes.c_void_p), ("angle", ctypes.c_void_p), ("counts", ctypes.c_void_p), ("capacity", ctypes.c_int),
                ("tcw_current", ctypes.c_void_p), ("tcw_last", ctypes.c_void_p), ("fx", ctypes.c_float), ("fy", ctypes.c_float), ("cx", ctypes.c_float),
                ("cy", ctypes.c_float), ("mbf", ctypes.c_float), ("mb", ctypes.c_float), ("max_x", ctypes.c_float), ("max_y", ctypes.c_float)]


class BowParams(ctypes.Structure):
    _fields_ = [("nn_ratio", ctypes.c_float), ("check_orientation", ctypes.c_int), ("mode", ctypes.c_int)]


def _bind_matcher(L):
    if getattr(L, "_matcher_bound", False):
        return
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.orbx_descriptor_distance.argtypes = [vp, vp]
    L.orbx_matcher_create.argtypes = [ci, ci, ci, ctypes.POINTER(vp)]
    L.orbx_matcher_destroy.argtypes = [vp]
    L.orbx_matcher_destroy.restype = None
    L.orbx_search_by_bow_device.argtypes = [vp, ctypes.POINTER(FeatureSet), ctypes.POINTER(FeatureSet), vp, vp, ci, ctypes.POINTER(BowParams), vp]
    L.orbx_search_for_triangulation.argtypes = [vp, ctypes.POINTER(FeatureSet), ctypes.POINTER(FeatureSet), vp, vp, vp]
    L.orbx_stereo_match_device.argtypes = [vp, ctypes.POINTER(FeatureSet), ctypes.POINTER(FeatureSet), vp, vp, ci, vp, ci, ctypes.c_float, vp]
    L.orbx_matcher_results_device.argtypes = [vp, vp, vp, vp, vp]
    L.orbx_compute_stereo_matches_device.argtypes = [vp, vp, vp, vp, vp, ci, ctypes.c_float, ctypes.c_float]
    L.orbx_stereo_results_device.argtypes = [vp, vp, vp, vp]
    L.orbx_stereo_download.argtypes = [vp, ci, vp, vp, ci]
    L.orbx_stereo_frame.argtypes = [vp, vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, ci]
    L.orbx_matcher_download.argtypes = [vp, ci, vp, vp, ci, vp]
    L.orbx_matcher_sync.argtypes = [vp]
    L.orbx_search_by_bow.argtypes = [vp, ctypes.POINTER(FeatureSet), ctypes.POINTER(FeatureSet), ctypes.POINTER(BowParams), vp, vp]
    L.orbx_stereo_match.argtypes = [vp, ctypes.POINTER(FeatureSet), ctypes.POINTER(FeatureSet), vp, ci, ctypes.c_float, vp, vp]
    L.orbx_matcher_last_timing.argtypes = [vp, vp]
    L.orbx_search_by_projection_last.argtypes = [vp, ctypes.POINTER(ProjectionFrame), ctypes.POINTER(ProjectionLast), vp, ci, ctypes.c_float, ci, ci, vp, vp]
    L.orbx_search_by_projection.argtypes = [vp, ctypes.POINTER(ProjectionFrame), ctypes.POINTER(ProjectionPoints), vp, ci, ctypes.c_float, ctypes.c_float, vp, vp]
    L.orbx_matcher_last_kernel_timing.argtypes = [vp, vp, vp]
    L._matcher_bound = True


def DescriptorDistance(a, b):
    """ORBmatcher::DescriptorDistance (reference include/ORBmatcher.h:65)."""
    L = load_library()
    _bind_matcher(L)
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return L.orbx_descriptor_distance(_ptr(a), _ptr(b))


def _host_set(kps, desc, groups=None, valid=None):
    """One frame of host features -> (FeatureSet, keepalive)."""
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    n = np.array([len(kps)], np.int32)
    keep = [kps, desc, n]
    g = v = None
    if groups is not None:
        g = np.ascontiguousarray(groups, np.int32)
        keep.append(g)
    if valid is not None:
        v = np.ascontiguousarray(valid, np.uint8)
        keep.append(v)
    fs = FeatureSet(kps.ctypes.data, desc.ctypes.data, n.ctypes.data, g.ctypes.data if g is not None else None,
                    v.ctypes.data if v is not None else None, max(len(kps), 1), 1)
    return fs, keep


class FusePoints(ctypes.Structure):
    _fields_ = [("u", ctypes.c_void_p), ("v", ctypes.c_void_p), ("ur", ctypes.c_void_p), ("level", ctypes.c_void_p), ("radius", ctypes.c_void_p),
                ("active", ctypes.c_void_p), ("descriptors", ctypes.c_void_p), ("counts", ctypes.c_void_p), ("capacity", ctypes.c_int),
                ("kf_min_x", ctypes.c_float), ("kf_min_y", ctypes.c_float)]


class AreaQueries(ctypes.Structure):
    _fields_ = [("u", ctypes.c_void_p), ("v", ctypes.c_void_p), ("radius", ctypes.c_void_p), ("min_level", ctypes.c_void_p), ("max_level", ctypes.c_void_p),
                ("active", ctypes.c_void_p), ("descriptors", ctypes.c_void_p), ("counts", ctypes.c_void_p), ("capacity", ctypes.c_int),
                ("window_min_x", ctypes.c_float), ("window_min_y", ctypes.c_float)]


class FrustumFrame(ctypes.Structure):
    _fields_ = [("tcw", ctypes.c_void_p), ("fx", ctypes.c_float), ("fy", ctypes.c_float), ("cx", ctypes.c_float), ("cy", ctypes.c_float), ("mbf", ctypes.c_float),
                ("min_x", ctypes.c_float), ("max_x", ctypes.c_float), ("min_y", ctypes.c_float), ("max_y", ctypes.c_float), ("ratio_thresholds", ctypes.c_void_p),
                ("nlevels", ctypes.c_int), ("nframes", ctypes.c_int)]


class MapPoints(ctypes.Structure):
    _fields_ = [("world_pos", ctypes.c_void_p), ("normal", ctypes.c_void_p), ("max_distance", ctypes.c_void_p), ("min_distance", ctypes.c_void_p),
                ("counts", ctypes.c_void_p), ("capacity", ctypes.c_int)]


class LocalPoints(ctypes.Structure):
    _fields_ = [("world_pos", ctypes.c_void_p), ("normal", ctypes.c_void_p), ("max_distance", ctypes.c_void_p), ("min_distance", ctypes.c_void_p),
                ("descriptors", ctypes.c_void_p), ("has_observations", ctypes.c_void_p), ("count", ctypes.c_int)]


def predict_scale_thresholds(log_scale_factor, nlevels):
    """orbx_predict_scale_thresholds: the float ratios at which MapPoint::PredictScale changes level (host, libm log)."""
    L = load_library()
    out = np.zeros(max(nlevels - 1, 1), np.float32)
    L.orbx_predict_scale_thresholds.argtypes = [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
    _check(L.orbx_predict_scale_thresholds(ctypes.c_float(log_scale_factor), nlevels, _ptr(out)))
    return out[:nlevels - 1]


class TriangulationParams(ctypes.Structure):
    _fields_ = [("f12", ctypes.c_void_p), ("epipole", ctypes.c_void_p), ("stereo_a", ctypes.c_void_p), ("stereo_b", ctypes.c_void_p),
                ("scale_factors", ctypes.c_void_p), ("level_sigma2", ctypes.c_void_p), ("nlevels", ctypes.c_int), ("check_orientation", ctypes.c_int)]


class ORBmatcher:
    """Mirror of the Hamming paths of ORB_SLAM2::ORBmatcher (reference include/ORBmatcher.h:57-215)."""
    TH_LOW = 50
    TH_HIGH = 100
    HISTO_LENGTH = 30

    def __init__(self, nnratio=0.6, checkOri=True, max_features=4096, max_pairs=1, device=0):
        self._L = load_library()
        _bind_matcher(self._L)
        self.nnratio, self.checkOri = float(nnratio), bool(checkOri)
        self.max_features, self.max_pairs = max_features, max_pairs
        self._h = ctypes.c_void_p()
        _check(self._L.orbx_matcher_create(device, max_features, max_pairs, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbx_matcher_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- single pair, host arrays ----
    def SearchByBoW(self, kpsA, descA, kpsB, descB, groupsA=None, groupsB=None, validA=None, validB=None, mode=0):
        """mode 0: (KeyFrame, Frame) -> matches indexed by B;  mode 1: (KF1, KF2) -> indexed by A."""
        fa, ka = _host_set(kpsA, descA, groupsA, validA)
        fb, kb = _host_set(kpsB, descB, groupsB, validB)
        nout = len(kpsB) if mode == 0 else len(kpsA)
        out = np.full(max(nout, 1), -1, np.int32)
        nm = ctypes.c_int32()
        prm = BowParams(self.nnratio, 1 if self.checkOri else 0, mode)
        _check(self._L.orbx_search_by_bow(self._h, ctypes.byref(fa), ctypes.byref(fb), ctypes.byref(prm), _ptr(out), ctypes.byref(nm)))
        return nm.value, out[:nout]

    def SearchForTriangulation(self, kf1, kf2, F12, epipole, scale_factors, level_sigma2, only_stereo=False):
        """ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (reference
        src/ORBmatcher.cc:810-1017).  kf*: dict(kps (structured mvKeysUn), desc, groups (mFeatVec node ids),
        has_mp (feature holds a MapPoint), u_right (mvuRight)); epipole = (ex, ey) of :817-826;
        scale_factors / level_sigma2 of pKF2.  Returns (nmatches, matches12[n1]) with matches12[i] = KF2 feature or -1."""
        sets, keep, flags = [], [], []
        for kf in (kf1, kf2):
            ur = np.asarray(kf["u_right"], np.float32)
            st = np.ascontiguousarray(ur >= 0, np.uint8)
            ok = np.asarray(kf["has_mp"], np.uint8) == 0
            if only_stereo:
                ok = ok & (st != 0)
            fs, k = _host_set(kf["kps"], kf["desc"], kf["groups"], ok.astype(np.uint8))
            sets.append(fs); keep.append(k); flags.append(st)
        f12 = np.ascontiguousarray(F12, np.float32).reshape(9)
        epi = np.ascontiguousarray(epipole, np.float32).reshape(2)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        s2 = np.ascontiguousarray(level_sigma2, np.float32)
        prm = TriangulationParams(f12.ctypes.data, epi.ctypes.data, flags[0].ctypes.data, flags[1].ctypes.data, sf.ctypes.data, s2.ctypes.data,
                                   len(sf), 1 if self.checkOri else 0)
        n1 = len(kf1["kps"])
        out = np.full(max(n1, 1), -1, np.int32)
        nm = ctypes.c_int32()
        _check(self._L.orbx_search_for_triangulation(self._h, ctypes.byref(sets[0]), ctypes.byref(sets[1]), ctypes.byref(prm), _ptr(out), ctypes.byref(nm)))
        return nm.value, out[:n1]

    def FuseSearch(self, kf, points, chi2_gate=True):
        """Steps 2-3 of ORBmatcher::Fuse (reference src/ORBmatcher.cc:1093-1146 / 1258-1276): per map point the KeyFrame
        feature of minimum Hamming distance inside GetFeaturesInArea(u, v, radius) that passes the level gate and
        (first overload, chi2_gate) the reprojection gate.  kf: dict(kps (mvKeysUn), desc, u_right, inv_level_sigma2,
        width, height[, min_x, min_y, max_x, max_y]); points: dict(u, v, ur, level, radius, active, desc).
        Returns (best_idx[m], best_dist[m]); the caller fuses where best_dist <= TH_LOW."""
        k = np.ascontiguousarray(kf["kps"], KEYPOINT_DTYPE)
        n = len(k)
        d = np.ascontiguousarray(kf["desc"], np.uint8)
        ur = np.ascontiguousarray(kf["u_right"], np.float32)
        s2 = np.ascontiguousarray(kf["inv_level_sigma2"], np.float32)
        minx, miny = np.float32(kf.get("min_x", 0.0)), np.float32(kf.get("min_y", 0.0))
        maxx, maxy = np.float32(kf.get("max_x", kf["width"])), np.float32(kf.get("max_y", kf["height"]))
        gw, gh = np.float32(64) / (maxx - minx), np.float32(48) / (maxy - miny)
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        pu, pv, pur, prad = f32(points["u"]), f32(points["v"]), f32(points["ur"]), f32(points["radius"])
        m = len(pu)
        lvl = np.ascontiguousarray(points["level"], np.int32)
        act = np.ascontiguousarray(points["active"], np.uint8)
        pd = np.ascontiguousarray(points["desc"], np.uint8)
        cn, cm = np.array([n], np.int32), np.array([m], np.int32)
        fr = ProjectionFrame(k.ctypes.data, d.ctypes.data, ur.ctypes.data, None, cn.ctypes.data, max(n, 1), 1, minx, miny, gw, gh)
        pt = FusePoints(pu.ctypes.data, pv.ctypes.data, pur.ctypes.data, lvl.ctypes.data, prad.ctypes.data, act.ctypes.data, pd.ctypes.data, cm.ctypes.data,
                        max(m, 1), float(np.float32(int(minx))), float(np.float32(int(miny))))
        bi, bd = np.full(max(m, 1), -1, np.int32), np.full(max(m, 1), 256, np.int32)
        self._L.orbx_fuse_search.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _check(self._L.orbx_fuse_search(self._h, ctypes.byref(fr), ctypes.byref(pt), _ptr(s2), len(s2), 1 if chi2_gate else 0, _ptr(bi), _ptr(bd)))
        return bi[:m], bd[:m]

    def AreaSearchGreedy(self, frame, queries, max_dist):
        """The search loops of ORBmatcher::SearchByProjection(pKF, Scw, ...) (loop closing, reference src/ORBmatcher.cc:453-510)
        and SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (relocalisation, :1790-1826).
        frame: dict(kps (mvKeysUn), desc, blocked, width, height[, min_x, ...]); queries: dict(u, v, radius, min_level,
        max_level, active, desc[, window_int_bounds]).  Returns (nmatches, assigned[m], dists[m])."""
        k = np.ascontiguousarray(frame["kps"], KEYPOINT_DTYPE)
        n = len(k)
        d = np.ascontiguousarray(frame["desc"], np.uint8)
        blk = np.ascontiguousarray(frame["blocked"], np.uint8)
        minx, miny = np.float32(frame.get("min_x", 0.0)), np.float32(frame.get("min_y", 0.0))
        maxx, maxy = np.float32(frame.get("max_x", frame["width"])), np.float32(frame.get("max_y", frame["height"]))
        gw, gh = np.float32(64) / (maxx - minx), np.float32(48) / (maxy - miny)
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        qu, qv, qr = f32(queries["u"]), f32(queries["v"]), f32(queries["radius"])
        m = len(qu)
        lo, hi = np.ascontiguousarray(queries["min_level"], np.int32), np.ascontiguousarray(queries["max_level"], np.int32)
        act = np.ascontiguousarray(queries["active"], np.uint8)
        qd = np.ascontiguousarray(queries["desc"], np.uint8)
        cn, cm = np.array([n], np.int32), np.array([m], np.int32)
        wx, wy = (np.float32(int(minx)), np.float32(int(miny))) if queries.get("window_int_bounds") else (minx, miny)
        fr = ProjectionFrame(k.ctypes.data, d.ctypes.data, None, blk.ctypes.data, cn.ctypes.data, max(n, 1), 1, minx, miny, gw, gh)
        q = AreaQueries(qu.ctypes.data, qv.ctypes.data, qr.ctypes.data, lo.ctypes.data, hi.ctypes.data, act.ctypes.data, qd.ctypes.data, cm.ctypes.data,
                        max(m, 1), float(wx), float(wy))
        asg, dst = np.full(max(m, 1), -1, np.int32), np.full(max(m, 1), 256, np.int32)
        nm = ctypes.c_int32()
        self._L.orbx_area_search_greedy.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 3
        _check(self._L.orbx_area_search_greedy(self._h, ctypes.byref(fr), ctypes.byref(q), int(max_dist), _ptr(asg), _ptr(dst), ctypes.byref(nm)))
        return nm.value, asg[:m], dst[:m]

    def SearchForInitialization(self, f1, f2, prev_matched, window_size=10):
        """ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (reference src/ORBmatcher.cc:515-654).
        f1 / f2: dict(kps (mvKeysUn), desc); f2 also width, height[, min_x, ...]; prev_matched: (n1, 2) float array (updated copy returned).
        Returns (nmatches, vnMatches12, vbPrevMatched)."""
        fs1, keep1 = _host_set(f1["kps"], f1["desc"])
        k2 = np.ascontiguousarray(f2["kps"], KEYPOINT_DTYPE)
        d2 = np.ascontiguousarray(f2["desc"], np.uint8)
        n1, n2 = len(f1["kps"]), len(k2)
        minx, miny = np.float32(f2.get("min_x", 0.0)), np.float32(f2.get("min_y", 0.0))
        maxx, maxy = np.float32(f2.get("max_x", f2["width"])), np.float32(f2.get("max_y", f2["height"]))
        gw, gh = np.float32(64) / (maxx - minx), np.float32(48) / (maxy - miny)
        cn = np.array([n2], np.int32)
        fr = ProjectionFrame(k2.ctypes.data, d2.ctypes.data, None, None, cn.ctypes.data, max(n2, 1), 1, minx, miny, gw, gh)
        prev = np.ascontiguousarray(prev_matched, np.float32).reshape(-1, 2).copy()
        out = np.full(max(n1, 1), -1, np.int32)
        nm = ctypes.c_int32()
        self._L.orbx_search_for_initialization.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _check(self._L.orbx_search_for_initialization(self._h, ctypes.byref(fs1), ctypes.byref(fr), _ptr(prev), int(window_size), self.nnratio,
                                                      1 if self.checkOri else 0, _ptr(out), ctypes.byref(nm)))
        out = out[:n1]
        ok = out >= 0
        prev[ok, 0], prev[ok, 1] = k2["x"][out[ok]], k2["y"][out[ok]]          # :646-650
        return nm.value, out, prev

    def isInFrustum(self, Tcw, cam, bounds, log_scale_factor, nlevels, points, viewing_cos_limit):
        """Frame::isInFrustum (reference src/Frame.cc:608-742) for a list of map points.  cam = (fx, fy, cx, cy, mbf), bounds = (mnMinX, mnMaxX,
        mnMinY, mnMaxY), points: dict(pos (n,3), normal (n,3), max_distance, min_distance).  Returns dict(in_view, proj_x, proj_y, proj_xr, level, view_cos)."""
        T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        th = predict_scale_thresholds(log_scale_factor, nlevels)
        pos, nrm = np.ascontiguousarray(points["pos"], np.float32), np.ascontiguousarray(points["normal"], np.float32)
        mx, mn = np.ascontiguousarray(points["max_distance"], np.float32), np.ascontiguousarray(points["min_distance"], np.float32)
        n = len(mx)
        cnt = np.array([n], np.int32)
        fr = FrustumFrame(T.ctypes.data, cam[0], cam[1], cam[2], cam[3], cam[4], bounds[0], bounds[1], bounds[2], bounds[3], th.ctypes.data, nlevels, 1)
        mp = MapPoints(pos.ctypes.data, nrm.ctypes.data, mx.ctypes.data, mn.ctypes.data, cnt.ctypes.data, max(n, 1))
        f = lambda: np.zeros(max(n, 1), np.float32)
        px, py, pxr, vc, lvl, iv = f(), f(), f(), f(), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.uint8)
        self._L.orbx_is_in_frustum.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_float] + [ctypes.c_void_p] * 6
        _check(self._L.orbx_is_in_frustum(self._h, ctypes.byref(fr), ctypes.byref(mp), ctypes.c_float(viewing_cos_limit), _ptr(px), _ptr(py), _ptr(pxr), _ptr(lvl),
                                          _ptr(vc), _ptr(iv)))
        return dict(in_view=iv[:n], proj_x=px[:n], proj_y=py[:n], proj_xr=pxr[:n], level=lvl[:n], view_cos=vc[:n])

    def SearchByProjection(self, frame, points, th, nnratio=None):
        """ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (reference src/ORBmatcher.cc:70-175).
        frame: dict(kps (structured mvKeysUn), desc, u_right, occupied, scale_factors, width, height[, min_x, min_y]);
        points: dict(proj_x, proj_y, proj_xr, level, view_cos, in_view, has_obs, desc).
        Returns (nmatches, assigned[n]) with assigned[i] = index of the point put into mvpMapPoints[i] or -1."""
        k = np.ascontiguousarray(frame["kps"], KEYPOINT_DTYPE)
        n = len(k)
        d = np.ascontiguousarray(frame["desc"], np.uint8)
        ur = np.ascontiguousarray(frame["u_right"], np.float32)
        occ = np.ascontiguousarray(frame["occupied"], np.uint8)
        sf = np.ascontiguousarray(frame["scale_factors"], np.float32)
        minx, miny = np.float32(frame.get("min_x", 0.0)), np.float32(frame.get("min_y", 0.0))
        maxx, maxy = np.float32(frame.get("max_x", frame["width"])), np.float32(frame.get("max_y", frame["height"]))
        gw, gh = np.float32(64) / (maxx - minx), np.float32(48) / (maxy - miny)      # src/Frame.cc:181-182
        cn, cm = np.array([n], np.int32), np.array([len(points["proj_x"])], np.int32)
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        px, py, pxr, vc = f32(points["proj_x"]), f32(points["proj_y"]), f32(points["proj_xr"]), f32(points["view_cos"])
        lvl = np.ascontiguousarray(points["level"], np.int32)
        inv, obs = np.ascontiguousarray(points["in_view"], np.uint8), np.ascontiguousarray(points["has_obs"], np.uint8)
        md = np.ascontiguousarray(points["desc"], np.uint8)
        F = ProjectionFrame(_ptr(k).value, _ptr(d).value, _ptr(ur).value, _ptr(occ).value, _ptr(cn).value, n, 1, float(minx), float(miny), float(gw), float(gh))
        P = ProjectionPoints(_ptr(px).value, _ptr(py).value, _ptr(pxr).value, _ptr(lvl).value, _ptr(vc).value, _ptr(inv).value, _ptr(obs).value,
                             _ptr(md).value, _ptr(cm).value, int(cm[0]))
        out = np.full(max(n, 1), -1, np.int32)
        nm = ctypes.c_int32()
        _check(self._L.orbx_search_by_projection(self._h, ctypes.byref(F), ctypes.byref(P), _ptr(sf), len(sf), ctypes.c_float(th),
                                                 ctypes.c_float(self.nnratio if nnratio is None else nnratio), _ptr(out), ctypes.byref(nm)))
        return nm.value, out[:n]

    def SearchLocalPoints(self, frame, Tcw, cam, log_scale_factor, points, th, nnratio=None, viewing_cos_limit=0.5):
        """Tracking::SearchLocalPoints (reference src/Tracking.cc:1760-1830) as one device chain (orbx_search_local_points): Frame::isInFrustum over
        `points` (dict pos, normal, max_distance, min_distance, desc, has_obs) and SearchByProjection(F, points, th) on the frame (dict as in
        SearchByProjection); cam = (fx, fy, cx, cy, mbf).  Returns (nmatches, assigned[n], dict(in_view, proj_x, proj_y, proj_xr, level, view_cos))."""
        k = np.ascontiguousarray(frame["kps"], KEYPOINT_DTYPE)
        n = len(k)
        d = np.ascontiguousarray(frame["desc"], np.uint8)
        ur = np.ascontiguousarray(frame["u_right"], np.float32)
        occ = np.ascontiguousarray(frame["occupied"], np.uint8)
        sf = np.ascontiguousarray(frame["scale_factors"], np.float32)
        minx, miny = np.float32(frame.get("min_x", 0.0)), np.float32(frame.get("min_y", 0.0))
        maxx, maxy = np.float32(frame.get("max_x", frame["width"])), np.float32(frame.get("max_y", frame["height"]))
        gw, gh = np.float32(64) / (maxx - minx), np.float32(48) / (maxy - miny)
        cn = np.array([n], np.int32)
        T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        thr = predict_scale_thresholds(log_scale_factor, len(sf))
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        pos, nrm, mx, mn = f32(points["pos"]), f32(points["normal"]), f32(points["max_distance"]), f32(points["min_distance"])
        md, obs = np.ascontiguousarray(points["desc"], np.uint8), np.ascontiguousarray(points["has_obs"], np.uint8)
        m = len(mx)
        F = ProjectionFrame(_ptr(k).value, _ptr(d).value, _ptr(ur).value, _ptr(occ).value, _ptr(cn).value, max(n, 1), 1, float(minx), float(miny), float(gw), float(gh))
        fr = FrustumFrame(T.ctypes.data, cam[0], cam[1], cam[2], cam[3], cam[4], float(minx), float(maxx), float(miny), float(maxy), thr.ctypes.data, len(sf), 1)
        P = LocalPoints(pos.ctypes.data, nrm.ctypes.data, mx.ctypes.data, mn.ctypes.data, md.ctypes.data, obs.ctypes.data, m)
        out, nm = np.full(max(n, 1), -1, np.int32), ctypes.c_int32()
        z = lambda dt: np.zeros(max(m, 1), dt)
        iv, px, py, pxr, lvl, vc = z(np.uint8), z(np.float32), z(np.float32), z(np.float32), z(np.int32), z(np.float32)
        self._L.orbx_search_local_points.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float] + [ctypes.c_void_p] * 8
        _check(self._L.orbx_search_local_points(self._h, ctypes.byref(F), ctypes.byref(fr), ctypes.byref(P), _ptr(sf), len(sf), ctypes.c_float(viewing_cos_limit),
                                                ctypes.c_float(th), ctypes.c_float(self.nnratio if nnratio is None else nnratio), _ptr(out), ctypes.byref(nm),
                                                _ptr(iv), _ptr(px), _ptr(py), _ptr(pxr), _ptr(lvl), _ptr(vc)))
        return nm.value, out[:n], dict(in_view=iv[:m], proj_x=px[:m], proj_y=py[:m], proj_xr=pxr[:m], level=lvl[:m], view_cos=vc[:m])

    def SearchByProjectionLast(self, frame, last, th, mono):
        """ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) (reference
        src/ORBmatcher.cc:1569-1728).  frame: as SearchByProjection plus Tcw (4x4) and cam (fx, fy, cx, cy, bf);
        last: dict(Tcw, valid, pos, desc, has_obs, kps (structured; octave of mvKeys, angle of mvKeysUn)).
        Returns (nmatches, assigned[n]): assigned[i2] = last-frame feature whose MapPoint is in mvpMapPoints[i2], or -1."""
        k = np.ascontiguousarray(frame["kps"], KEYPOINT_DTYPE)
        n = len(k)
        d = np.ascontiguousarray(frame["desc"], np.uint8)
        ur = np.ascontiguousarray(frame["u_right"], np.float32)
        occ = np.ascontiguousarray(frame["occupied"], np.uint8)
        sf = np.ascontiguousarray(frame["scale_factors"], np.float32)
        minx, miny = np.float32(frame.get("min_x", 0.0)), np.float32(frame.get("min_y", 0.0))
        maxx, maxy = np.float32(frame.get("max_x", frame["width"])), np.float32(frame.get("max_y", frame["height"]))
        gw, gh = np.float32(64) / (maxx - minx), np.float32(48) / (maxy - miny)
        lk = np.ascontiguousarray(last["kps"], KEYPOINT_DTYPE)
        nl = len(lk)
        cn, cl = np.array([n], np.int32), np.array([nl], np.int32)
        valid = np.ascontiguousarray(last["valid"], np.uint8)
        pos = np.ascontiguousarray(last["pos"], np.float32)
        ld = np.ascontiguousarray(last["desc"], np.uint8)
        obs = np.ascontiguousarray(last["has_obs"], np.uint8)
        octv = np.ascontiguousarray(lk["octave"], np.int32)
        ang = np.ascontiguousarray(lk["angle"], np.float32)
        tc, tl = np.ascontiguousarray(frame["Tcw"], np.float32), np.ascontiguousarray(last["Tcw"], np.float32)
        fx, fy, cx, cy, bf = [np.float32(v) for v in frame["cam"]]
        F = ProjectionFrame(_ptr(k).value, _ptr(d).value, _ptr(ur).value, _ptr(occ).value, _ptr(cn).value, n, 1, float(minx), float(miny), float(gw), float(gh))
        Ls = ProjectionLast(_ptr(valid).value, _ptr(pos).value, _ptr(ld).value, _ptr(obs).value, _ptr(octv).value, _ptr(ang).value, _ptr(cl).value, nl,
                            _ptr(tc).value, _ptr(tl).value, float(fx), float(fy), float(cx), float(cy), float(bf), float(bf / fx), float(maxx), float(maxy))
        out = np.full(max(n, 1), -1, np.int32)
        nm = ctypes.c_int32()
        _check(self._L.orbx_search_by_projection_last(self._h, ctypes.byref(F), ctypes.byref(Ls), _ptr(sf), len(sf), ctypes.c_float(th), 1 if mono else 0,
                                                      1 if self.checkOri else 0, _ptr(out), ctypes.byref(nm)))
        return nm.value, out[:n]

    def StereoHamming(self, kpsL, descL, kpsR, descR, scale_factors, max_disparity=float("inf")):
        fl, kl = _host_set(kpsL, descL)
        fr, kr = _host_set(kpsR, descR)
        n = len(kpsL)
        bd = np.zeros(max(n, 1), np.int32)
        bi = np.zeros(max(n, 1), np.int32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        _check(self._L.orbx_stereo_match(self._h, ctypes.byref(fl), ctypes.byref(fr), _ptr(sf), len(sf), ctypes.c_float(max_disparity), _ptr(bd), _ptr(bi)))
        return bd[:n], bi[:n]

    # ---- batched, device resident: features straight from an extractor's last batch ----
    @staticmethod
    def features_of(extractor, nframes):
        k, d, c, cap = extractor.results_device()
        return FeatureSet(k.value, d.value, c.value, None, None, cap, nframes)

    def search_by_bow_device(self, fsA, fsB, pairsA, pairsB, mode=0, after=None):
        pa = np.ascontiguousarray(pairsA, np.int32)
        pb = np.ascontiguousarray(pairsB, np.int32)
        prm = BowParams(self.nnratio, 1 if self.checkOri else 0, mode)
        _check(self._L.orbx_search_by_bow_device(self._h, ctypes.byref(fsA), ctypes.byref(fsB), _ptr(pa), _ptr(pb), len(pa), ctypes.byref(prm),
                                                 after._h if after is not None else None))

    def stereo_match_device(self, fsL, fsR, pairsL, pairsR, scale_factors, max_disparity=float("inf"), after=None):
        pl = np.ascontiguousarray(pairsL, np.int32)
        pr = np.ascontiguousarray(pairsR, np.int32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        _check(self._L.orbx_stereo_match_device(self._h, ctypes.byref(fsL), ctypes.byref(fsR), _ptr(pl), _ptr(pr), len(pl), _ptr(sf), len(sf),
                                                ctypes.c_float(max_disparity), after._h if after is not None else None))

    def compute_stereo_matches_device(self, ext_left, ext_right, frames_l, frames_r, mbf, mb=0.0):
        """Frame::ComputeStereoMatches (reference src/Frame.cc:1026-1420), complete, on the last batches
        of two extractor handles (may be the same handle).  mb=0 is what the reference's stereo
        constructor has when the function runs (src/Frame.cc:125)."""
        fl = np.ascontiguousarray(frames_l, np.int32)
        fr = np.ascontiguousarray(frames_r, np.int32)
        _check(self._L.orbx_compute_stereo_matches_device(self._h, ext_left._h, ext_right._h, _ptr(fl), _ptr(fr), len(fl),
                                                          ctypes.c_float(mbf), ctypes.c_float(mb)))

    def stereo_frame(self, ext_left, ext_right, mbf, mb=0.0, n=None):
        """Frame::ComputeStereoMatches of ONE stereo frame, after the two extractors' single-frame calls (orbx_stereo_frame): (mvuRight, mvDepth)."""
        n = self.max_features if n is None else int(n)
        u = np.full(n, -1.0, np.float32)
        z = np.full(n, -1.0, np.float32)
        _check(self._L.orbx_stereo_frame(self._h, ext_left._h, ext_right._h, ctypes.c_float(mbf), ctypes.c_float(mb), _ptr(u), _ptr(z), n))
        return u, z

    def download_stereo(self, npairs, stride=None):
        """(mvuRight, mvDepth) per pair, -1 where the reference leaves -1."""
        stride = stride or self.max_features
        u = np.zeros((npairs, stride), np.float32)
        z = np.zeros((npairs, stride), np.float32)
        _check(self._L.orbx_stereo_download(self._h, npairs, _ptr(u), _ptr(z), stride))
        return u, z

    def sync(self):
        _check(self._L.orbx_matcher_sync(self._h))

    def download(self, npairs, stride=None):
        stride = stride or self.max_features
        m = np.zeros((npairs, stride), np.int32)
        d = np.zeros((npairs, stride), np.int32)
        n = np.zeros(npairs, np.int32)
        _check(self._L.orbx_matcher_download(self._h, npairs, _ptr(m), _ptr(d), stride, _ptr(n)))
        return m, d, n

    def last_timing(self):
        t = ctypes.c_float()
        _check(self._L.orbx_matcher_last_timing(self._h, ctypes.byref(t)))
        return t.value

    def last_kernel_timing(self):
        """(distance kernels ms, greedy replay ms) of the SearchByBoW calls averaged by the last last_timing()."""
        a, b = ctypes.c_float(), ctypes.c_float()
        _check(self._L.orbx_matcher_last_kernel_timing(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value


# =====================================================================================
# Optimizer::LocalBundleAdjustment numerical core over the C ABI
# =====================================================================================
class Vocabulary:
    """DBoW2 vocabulary tree on the device: transform() == TemplatedVocabulary::transform
    (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1262), i.e. Frame::ComputeBoW."""

    def __init__(self, voc, device=0):
        self._L = load_library()
        L = self._L
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.orbx_vocabulary_create.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, ctypes.POINTER(vp)]
        L.orbx_vocabulary_destroy.argtypes = [vp]
        L.orbx_vocabulary_destroy.restype = None
        L.orbx_vocabulary_words.argtypes = [vp]
        L.orbx_bow_transform_device.argtypes = [vp, vp, ci]
        L.orbx_bow_results_device.argtypes = [vp, vp, vp, vp, vp]
        L.orbx_bow_download.argtypes = [vp, vp, ci, vp, vp, vp]
        L.orbx_bow_transform.argtypes = [vp, vp, ci, ci, vp, vp, vp]
        self._h = vp()
        par = np.ascontiguousarray(voc["parent"], np.int32)
        leaf = np.ascontiguousarray(voc["is_leaf"], np.uint8)
        desc = np.ascontiguousarray(voc["desc"], np.uint8)
        wt = np.ascontiguousarray(voc["weight"], np.float64)
        _check(L.orbx_vocabulary_create(device, int(voc["k"]), int(voc["L"]), len(par), _ptr(par), _ptr(leaf), _ptr(desc), _ptr(wt), ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_job", None) is not None and self._job.value:      # (a job only reads the vocabulary and must go first)
            self._L.orbx_bow_job_destroy(self._job)
            self._job = None
        if getattr(self, "_h", None):
            self._L.orbx_vocabulary_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return self._L.orbx_vocabulary_words(self._h)

    def transform(self, descriptors, levelsup=4):
        """(word id, FeatureVector node id or -1, word weight) per descriptor."""
        d = np.ascontiguousarray(descriptors, np.uint8)
        n = len(d)
        w, nd, wt = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64)
        _check(self._L.orbx_bow_transform(self._h, _ptr(d), n, levelsup, _ptr(w), _ptr(nd), _ptr(wt)))
        return w[:n], nd[:n], wt[:n]

    def transform_sorted(self, descriptors, levelsup=4):
        """transform() plus the two orders of its std::map fills: (word, node, weight, by_word, by_node) - by_word[k] / by_node[k] = the feature that is
        k-th by (word id, index) / (node id, index) among the filed features (orbx_bow_transform_sorted)."""
        d = np.ascontiguousarray(descriptors, np.uint8)
        n = len(d)
        w, nd, wt = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64)
        bw, bn = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        filed = ctypes.c_int32()
        self._L.orbx_bow_transform_sorted.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6
        _check(self._L.orbx_bow_transform_sorted(self._h, _ptr(d), n, levelsup, _ptr(w), _ptr(nd), _ptr(wt), _ptr(bw), _ptr(bn), ctypes.byref(filed)))
        return w[:n], nd[:n], wt[:n], bw[:filed.value], bn[:filed.value]

    def job_transform(self, extractor, levelsup=4):
        """The latency form (orbx_bow_job_begin / _end) on the features `extractor`'s last single-frame call left on the device: same five arrays."""
        L = self._L
        vp = ctypes.c_void_p
        L.orbx_bow_job_create.argtypes = [vp, ctypes.POINTER(vp)]
        L.orbx_bow_job_destroy.argtypes = [vp]
        L.orbx_bow_job_destroy.restype = None
        L.orbx_bow_job_begin.argtypes = [vp, vp, ctypes.c_int]
        L.orbx_bow_job_end.argtypes = [vp] + [ctypes.POINTER(vp)] * 5 + [ctypes.POINTER(ctypes.c_int32)] * 2
        if getattr(self, "_job", None) is None:
            self._job = vp()
            _check(L.orbx_bow_job_create(self._h, ctypes.byref(self._job)))
        _check(L.orbx_bow_job_begin(self._job, extractor._h, levelsup))
        ptrs = [vp() for _ in range(5)]
        filed, n = ctypes.c_int32(), ctypes.c_int32()
        _check(L.orbx_bow_job_end(self._job, *[ctypes.byref(q) for q in ptrs], ctypes.byref(filed), ctypes.byref(n)))
        if n.value == 0:
            return tuple(np.zeros(0, t) for t in (np.int32, np.int32, np.float64, np.int32, np.int32))
        view = lambda q, t, k: np.ctypeslib.as_array(ctypes.cast(q, ctypes.POINTER(t)), (k,)).copy()
        return (view(ptrs[0], ctypes.c_int32, n.value), view(ptrs[1], ctypes.c_int32, n.value), view(ptrs[2], ctypes.c_double, n.value),
                view(ptrs[3], ctypes.c_int32, filed.value), view(ptrs[4], ctypes.c_int32, filed.value))

    def transform_device(self, extractor, levelsup=4):
        _check(self._L.orbx_bow_transform_device(self._h, extractor._h, levelsup))

    def groups_device(self):
        """device pointer of the FeatureVector node ids of the last transform_device (orbx_feature_set.groups)."""
        nd, cap = ctypes.c_void_p(), ctypes.c_int()
        _check(self._L.orbx_bow_results_device(self._h, None, ctypes.byref(nd), None, ctypes.byref(cap)))
        return nd, cap.value

    def download(self, extractor, batch):
        _, cap = self.groups_device()
        w, nd, wt = np.zeros((batch, cap), np.int32), np.zeros((batch, cap), np.int32), np.zeros((batch, cap), np.float64)
        _check(self._L.orbx_bow_download(self._h, extractor._h, batch, _ptr(w), _ptr(nd), _ptr(wt)))
        return w, nd, wt


class Camera(ctypes.Structure):
    """orbx_camera: mK and mDistCoef (k1 k2 p1 p2 [k3])."""
    _fields_ = [("fx", ctypes.c_float), ("fy", ctypes.c_float), ("cx", ctypes.c_float), ("cy", ctypes.c_float),
                ("dist", ctypes.c_float * 5), ("ndist", ctypes.c_int)]


FRAME_GRID_COLS, FRAME_GRID_ROWS = 64, 48


class FrameGrid(ctypes.Structure):
    """orbx_frame_grid: Frame::mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv."""
    _fields_ = [("min_x", ctypes.c_float), ("min_y", ctypes.c_float), ("width_inv", ctypes.c_float), ("height_inv", ctypes.c_float)]

    @classmethod
    def from_bounds(cls, bounds):
        """the grid constants as the Frame constructor derives them from the image bounds (src/Frame.cc:326-327)"""
        b = np.asarray(bounds, np.float32)
        return cls(float(b[0]), float(b[2]), float(np.float32(FRAME_GRID_COLS) / np.float32(b[1] - b[0])),
                   float(np.float32(FRAME_GRID_ROWS) / np.float32(b[3] - b[2])))


class FrameOps:
    """Frame::UndistortKeyPoints, ComputeImageBounds and AssignFeaturesToGrid on the device
    (reference src/Frame.cc:899-1004, 460-491) for one camera (mK, mDistCoef)."""

    def __init__(self, fx, fy, cx, cy, dist, device=0):
        self._L = load_library()
        L = self._L
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.orbx_frame_ops_create.argtypes = [ci, vp, ctypes.POINTER(vp)]
        L.orbx_frame_ops_destroy.argtypes = [vp]
        L.orbx_frame_ops_destroy.restype = None
        L.orbx_frame_image_bounds.argtypes = [vp, ci, ci, vp]
        L.orbx_frame_undistort.argtypes = [vp, vp, ci, vp]
        L.orbx_frame_assign_grid.argtypes = [vp, vp, vp, ci, vp, vp]
        L.orbx_frame_finish_device.argtypes = [vp, vp, vp]
        L.orbx_frame_results_device.argtypes = [vp, vp, vp, vp, vp]
        L.orbx_frame_download.argtypes = [vp, vp, ci, vp, vp, vp]
        cam = Camera(fx, fy, cx, cy)
        dist = [float(d) for d in dist]
        for i, d in enumerate(dist[:5]):
            cam.dist[i] = d
        cam.ndist = len(dist)
        self._h = vp()
        _check(L.orbx_frame_ops_create(device, ctypes.byref(cam), ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self._L.orbx_frame_ops_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ComputeImageBounds(self, cols, rows):
        """mnMinX, mnMaxX, mnMinY, mnMaxY."""
        b = np.zeros(4, np.float32)
        _check(self._L.orbx_frame_image_bounds(self._h, cols, rows, _ptr(b)))
        return b

    def UndistortKeyPoints(self, keypoints):
        kp = np.ascontiguousarray(keypoints, dtype=KEYPOINT_DTYPE)
        un = np.zeros(max(len(kp), 1), KEYPOINT_DTYPE)
        _check(self._L.orbx_frame_undistort(self._h, _ptr(kp), len(kp), _ptr(un)))
        return un[:len(kp)]

    def AssignFeaturesToGrid(self, keypoints_un, grid):
        """mGrid as CSR: offsets[64*48+1] over cell = x*48 + y, indices in push_back order."""
        kp = np.ascontiguousarray(keypoints_un, dtype=KEYPOINT_DTYPE)
        off = np.zeros(FRAME_GRID_COLS * FRAME_GRID_ROWS + 1, np.int32)
        idx = np.zeros(max(len(kp), 1), np.int32)
        _check(self._L.orbx_frame_assign_grid(self._h, ctypes.byref(grid), _ptr(kp), len(kp), _ptr(off), _ptr(idx)))
        return off, idx[:off[-1]]

    def finish_device(self, extractor, grid):
        """both, fused, on the extractor's last batch (device resident)"""
        _check(self._L.orbx_frame_finish_device(self._h, extractor._h, ctypes.byref(grid)))

    def finish_frame(self, extractor, grid=None):
        """Latency form for ONE frame that `extractor`'s last single-frame call extracted (orbx_frame_finish_begin + _end): the kernel reads
        the keypoints on the device and writes into pinned memory.  -> (mvKeysUn or None when the camera is not distorted, offsets, indices, n);
        offsets / indices are None without a grid."""
        L = self._L
        vp = ctypes.c_void_p
        L.orbx_frame_finish_begin.argtypes = [vp, vp, vp]
        L.orbx_frame_finish_end.argtypes = [vp, vp, vp, vp, vp]
        _check(L.orbx_frame_finish_begin(self._h, extractor._h, ctypes.byref(grid) if grid is not None else None))
        un, off, idx, n = vp(), vp(), vp(), ctypes.c_int()
        _check(L.orbx_frame_finish_end(self._h, ctypes.byref(un), ctypes.byref(off), ctypes.byref(idx), ctypes.byref(n)))
        n = n.value
        kun = np.ctypeslib.as_array(ctypes.cast(un, ctypes.POINTER(ctypes.c_uint8)), shape=(max(n, 1) * KEYPOINT_DTYPE.itemsize,)).view(KEYPOINT_DTYPE)[:n].copy() if un.value else None
        o = np.ctypeslib.as_array(ctypes.cast(off, ctypes.POINTER(ctypes.c_int32)), shape=(FRAME_GRID_COLS * FRAME_GRID_ROWS + 1,)).copy() if off.value else None
        i = np.ctypeslib.as_array(ctypes.cast(idx, ctypes.POINTER(ctypes.c_int32)), shape=(max(n, 1),))[:int(o[-1])].copy() if (idx.value and o is not None) else None
        return kun, o, i, n

    def keypoints_un_device(self):
        kp, cap = ctypes.c_void_p(), ctypes.c_int()
        _check(self._L.orbx_frame_results_device(self._h, ctypes.byref(kp), None, None, ctypes.byref(cap)))
        return kp, cap.value

    def download(self, extractor, batch):
        _, cap = self.keypoints_un_device()
        un = np.zeros((batch, cap), KEYPOINT_DTYPE)
        off = np.zeros((batch, FRAME_GRID_COLS * FRAME_GRID_ROWS + 1), np.int32)
        idx = np.zeros((batch, cap), np.int32)
        _check(self._L.orbx_frame_download(self._h, extractor._h, batch, _ptr(un), _ptr(off), _ptr(idx)))
        return un, off, idx


class LbaProblem(ctypes.Structure):
    _fields_ = [("num_keyframes", ctypes.c_int), ("poses", ctypes.c_void_p), ("fixed", ctypes.c_void_p), ("intrinsics", ctypes.c_void_p),
                ("num_points", ctypes.c_int), ("points", ctypes.c_void_p), ("num_edges", ctypes.c_int), ("edge_point", ctypes.c_void_p),
                ("edge_keyframe", ctypes.c_void_p), ("edge_obs", ctypes.c_void_p), ("edge_inv_sigma2", ctypes.c_void_p)]


class LbaResult(ctypes.Structure):
    _fields_ = [("poses", ctypes.c_void_p), ("points", ctypes.c_void_p), ("edge_chi2", ctypes.c_void_p), ("edge_outlier", ctypes.c_void_p),
                ("stats", ctypes.c_double * 8)]


def _load_lba_synth():
    spec = importlib.util.spec_from_file_location("orbx_lba_synth", _PKG / "lba_synth.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


lba_synth = _load_lba_synth()


def _load_sibling(name):
    spec = importlib.util.spec_from_file_location("orbx_" + name, _PKG / (name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


distributed = _load_sibling("distributed")
voc_synth = _load_sibling("voc_synth")


class PoseProblem(ctypes.Structure):
    _fields_ = [("num_frames", ctypes.c_int), ("capacity", ctypes.c_int), ("poses", ctypes.c_void_p), ("cameras", ctypes.c_void_p), ("counts", ctypes.c_void_p),
                ("world_points", ctypes.c_void_p), ("observations", ctypes.c_void_p), ("inv_sigma2", ctypes.c_void_p)]


class PoseOptimizer:
    """Optimizer::PoseOptimization(Frame*) (reference src/Optimizer.cc:363-605) for a batch of independent frames."""

    def __init__(self, max_frames=64, max_features=4096, device=0):
        self._L = load_library()
        L = self._L
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.orbx_pose_optimizer_create.argtypes = [ci, ci, ci, ctypes.POINTER(vp)]
        L.orbx_pose_optimizer_destroy.argtypes = [vp]
        L.orbx_pose_optimizer_destroy.restype = None
        L.orbx_pose_optimization.argtypes = [vp, ctypes.POINTER(PoseProblem), vp, vp, vp, vp]
        self._h = vp()
        _check(L.orbx_pose_optimizer_create(device, max_frames, max_features, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self._L.orbx_pose_optimizer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def PoseOptimization(self, frames):
        """frames: list of dict(pose (4x4), cam (fx,fy,cx,cy,bf), Xw (n,3), obs (n,3; uR<0 mono), inv_sigma2 (n)).
        Returns list of dict(pose, outlier, inliers, stats)."""
        B = len(frames)
        cap = max(1, max(len(f["Xw"]) for f in frames))
        poses = np.zeros((B, 16), np.float32)
        cams = np.zeros((B, 5), np.float32)
        counts = np.zeros(B, np.int32)
        Xw, obs, inv = np.zeros((B, cap, 3), np.float32), np.zeros((B, cap, 3), np.float32), np.zeros((B, cap), np.float32)
        for i, f in enumerate(frames):
            n = len(f["Xw"])
            poses[i] = np.asarray(f["pose"], np.float32).reshape(16)
            cams[i] = np.asarray(f["cam"], np.float32)
            counts[i] = n
            Xw[i, :n], obs[i, :n], inv[i, :n] = f["Xw"], f["obs"], f["inv_sigma2"]
        P = PoseProblem(B, cap, _ptr(poses).value, _ptr(cams).value, _ptr(counts).value, _ptr(Xw).value, _ptr(obs).value, _ptr(inv).value)
        po, outl, inl, st = np.zeros((B, 16), np.float32), np.zeros((B, cap), np.uint8), np.zeros(B, np.int32), np.zeros((B, 8), np.float64)
        _check(self._L.orbx_pose_optimization(self._h, ctypes.byref(P), _ptr(po), _ptr(outl), _ptr(inl), _ptr(st)))
        return [dict(pose=po[i].reshape(4, 4), outlier=outl[i, :counts[i]], inliers=int(inl[i]), stats=st[i]) for i in range(B)]


class Optimizer:
    """Mirror of the static ORB_SLAM2::Optimizer::LocalBundleAdjustment (reference include/Optimizer.h:112)
    on a flat window (dict as produced by lba_synth.make_window)."""

    def __init__(self, max_keyframes=256, max_points=20000, max_edges=400000, device=0):
        self._L = load_library()
        vp, ci = ctypes.c_void_p, ctypes.c_int
        self._L.orbx_lba_create.argtypes = [ci, ci, ci, ci, ctypes.POINTER(vp)]
        self._L.orbx_lba_destroy.argtypes = [vp]
        self._L.orbx_lba_destroy.restype = None
        self._L.orbx_lba_solve.argtypes = [vp, ctypes.POINTER(LbaProblem), vp, ctypes.POINTER(LbaResult)]
        self._L.orbx_lba_last_timing.argtypes = [vp, vp, vp]
        self._h = vp()
        _check(self._L.orbx_lba_create(device, max_keyframes, max_points, max_edges, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbx_lba_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def LocalBundleAdjustment(self, w, stop_flag=None, iterations=None, robust=True):
        K, P, E = w["K"], w["P"], w["E"]
        arrs = {k: np.ascontiguousarray(w[k]) for k in ("poses", "fixed", "intr", "points", "edge_point", "edge_kf", "edge_obs", "edge_inv_sigma2")}
        prob = LbaProblem(K, arrs["poses"].ctypes.data, arrs["fixed"].ctypes.data, arrs["intr"].ctypes.data, P, arrs["points"].ctypes.data, E,
                          arrs["edge_point"].ctypes.data, arrs["edge_kf"].ctypes.data, arrs["edge_obs"].ctypes.data, arrs["edge_inv_sigma2"].ctypes.data)
        poses = np.zeros((K, 16), np.float32)
        points = np.zeros((P, 3), np.float32)
        chi2 = np.zeros(E, np.float64)
        outl = np.zeros(E, np.uint8)
        res = LbaResult(poses.ctypes.data, points.ctypes.data, chi2.ctypes.data, outl.ctypes.data)
        stop = None if stop_flag is None else stop_flag.ctypes.data_as(ctypes.c_void_p)
        if iterations is None:
            _check(self._L.orbx_lba_solve(self._h, ctypes.byref(prob), stop, ctypes.byref(res)))
        else:
            self._L.orbx_bundle_adjustment.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            _check(self._L.orbx_bundle_adjustment(self._h, ctypes.byref(prob), int(iterations), 1 if robust else 0, stop, ctypes.byref(res)))
        return dict(poses=poses, points=points, chi2=chi2, outlier=outl, stats=np.array(list(res.stats)))

    def BundleAdjustment(self, w, iterations=5, robust=True, stop_flag=None):
        """Optimizer::BundleAdjustment / GlobalBundleAdjustemnt (reference src/Optimizer.cc:55-360) on the same flat problem layout."""
        return self.LocalBundleAdjustment(w, stop_flag, iterations=iterations, robust=robust)

    def last_timing(self):
        ms = ctypes.c_float()
        fl = ctypes.c_double()
        _check(self._L.orbx_lba_last_timing(self._h, ctypes.byref(ms), ctypes.byref(fl)))
        return ms.value, fl.value
