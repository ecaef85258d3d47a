"""The C-ABI library loads on a box without a GPU and exports every symbol that
include/orbx.h declares; constructing a handle without a device fails loudly (no CPU
fallback)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_functions():
    text = (ROOT / "include" / "orbx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", text)))


def test_all_declared_symbols_exported(orbx):
    L = orbx.load_library()
    names = declared_functions()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_no_cpu_fallback(orbx):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(orbx.OrbxError) as e:
        orbx.ORBextractor(1000, 1.2, 8, 20, 7)
    assert e.value.code == -4   # ORBX_ERR_NODEVICE
    for make in (lambda: orbx.ORBmatcher(0.7, True), lambda: orbx.Optimizer(), lambda: orbx.PoseOptimizer(),
                 lambda: orbx.Vocabulary(orbx.voc_synth.make_vocabulary(4, 2, 1)),
                 lambda: orbx.FrameOps(500.0, 500.0, 320.0, 240.0, [0.1, 0.0, 0.0, 0.0])):
        with pytest.raises(orbx.OrbxError) as e:
            make()
        assert e.value.code == -4


def test_synth_frame_deterministic(orbx):
    a = orbx.synth_frame(42, 640, 480)
    b = orbx.synth_frame(42, 640, 480)
    c = orbx.synth_frame(43, 640, 480)
    assert (a == b).all() and (a != c).any()
    lo = orbx.synth_frame(42, 640, 480, orbx.SYNTH_LOW_TEXTURE)
    assert lo.std() < a.std()


def test_null_handles_and_bad_arguments_are_errors_not_crashes(orbx):
    """Every entry point validates its handle / pointers before touching a device: negative status + message, no crash (runs without a GPU)."""
    L = orbx.load_library()
    vp = ctypes.c_void_p
    none = vp(None)
    calls = [
        (L.orbx_search_by_bow_device, [none] * 5 + [1, none, none]),
        (L.orbx_search_for_triangulation_device, [none] * 5 + [1, none, none]),
        (L.orbx_fuse_search_device, [none, none, none, none, 8, 1]),
        (L.orbx_area_search_greedy_device, [none, none, none, 50]),
        (L.orbx_search_for_initialization_device, [none, none, none, none, 10, ctypes.c_float(0.9), 1]),
        (L.orbx_is_in_frustum_device, [none, none, none, ctypes.c_float(0.5)]),
        (L.orbx_search_by_projection_device, [none, none, none, none, 8, ctypes.c_float(1.0), ctypes.c_float(0.8)]),
        (L.orbx_frame_finish_device, [none, none, none]),
        (L.orbx_bow_transform_device, [none, none, 4]),
        (L.orbx_lba_solve, [none, none, none, none]),
        (L.orbx_bundle_adjustment, [none, none, 5, 1, none, none]),
        (L.orbx_pose_optimization, [none] * 6),
        (L.orbx_matcher_sync, [none]),
    ]
    for fn, args in calls:
        fn.restype = ctypes.c_int
        fn.argtypes = None
        assert fn(*args) < 0, fn.__name__
        assert len(L.orbx_last_error()) > 0
    th = (ctypes.c_float * 8)()
    L.orbx_predict_scale_thresholds.argtypes = [ctypes.c_float, ctypes.c_int, vp]
    assert L.orbx_predict_scale_thresholds(ctypes.c_float(0.1823), 0, th) < 0          # nlevels out of range
    assert L.orbx_predict_scale_thresholds(ctypes.c_float(-1.0), 8, th) < 0            # log of a scale factor <= 1
    assert L.orbx_predict_scale_thresholds(ctypes.c_float(0.1823), 8, th) == 0 and th[0] == 1.0   # ratio <= 1 is level 0


def test_extractor_tables_without_a_device(orbx, oracle):
    """orbx_extractor_tables_for: ORBextractor's constructor tables (src/ORBextractor.cc:499-554) from the configuration alone - what the drop-in
    constructor serves its getters from even when no HIP device can be opened - equal the compiled reference's (the restatement's, where
    oracle/_ref is not built) bit for bit, on a box without a GPU."""
    import numpy as np
    L = orbx.load_library()
    for nf, sf, nl in ((1000, 1.2, 8), (2000, 1.2, 8), (1200, 1.2, 8), (500, 1.5, 5), (3000, 1.1, 12), (1, 2.0, 1)):
        cfg = orbx.ExtractorConfig(nf, sf, nl, 20, 7, 0, 0, 0, 0)
        t = [np.zeros(nl, np.float32) for _ in range(4)]
        q = np.zeros(nl, np.int32)
        L.orbx_extractor_tables_for.argtypes = [ctypes.c_void_p] * 6
        assert L.orbx_extractor_tables_for(ctypes.byref(cfg), *[a.ctypes.data_as(ctypes.c_void_p) for a in t], q.ctypes.data_as(ctypes.c_void_p)) == 0
        ext = oracle.reference(nf, sf, nl) if oracle.ref is not None else oracle.restatement(nf, sf, nl)
        wt, wq, _ = ext.tables()
        assert all((a.view(np.uint32) == b.view(np.uint32)).all() for a, b in zip(t, wt)) and (q == wq).all() and q.sum() >= nf
    bad = orbx.ExtractorConfig(1000, 1.0, 8, 20, 7, 0, 0, 0, 0)
    assert L.orbx_extractor_tables_for(ctypes.byref(bad), None, None, None, None, None) < 0
