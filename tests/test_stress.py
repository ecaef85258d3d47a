"""The randomised sweeps of tools/stress_*.py as GPU tests (bounded): random image sizes, feature counts, scale factors, level
counts, FAST thresholds, noise / flat / low-texture images through ORBextractor::operator(), and random feature sets / high
contention cases through the matcher entry points - bit-exact / index-exact vs the CPU restatements (which are pinned to the
compiled reference).  Each sweep runs as its own process and exits non-zero on any mismatch."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _sweep(script, *args):
    r = subprocess.run([sys.executable, str(ROOT / "tools" / script)] + list(args), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, r.stdout[-3000:]
    return r.stdout


@pytest.mark.gpu
def test_extractor_random_geometries():
    out = _sweep("stress_extractor.py", "40")
    assert " 0 mismatches" in out


@pytest.mark.gpu
def test_matchers_random_cases():
    assert " 0 mismatches" in _sweep("stress_matchers.py", "40")


@pytest.mark.gpu
def test_projection_high_contention():
    assert " 0 mismatches" in _sweep("stress_projection.py", "30")
