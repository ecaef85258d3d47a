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
                 lambda: orbx.Vocabulary(orbx.voc_synth.make_vocabulary(4, 2, 1))):
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
