"""The callers north_star names compile UNMODIFIED against the drop-in headers (oracle/Makefile: `callers`).

src/Tracking.cc constructs the extractors (:179-192) and calls the matcher / PoseOptimization (:1195, :2073); src/LocalMapping.cc calls
Optimizer::LocalBundleAdjustment (:123).  Both are compiled (-fsyntax-only: nothing linked, nothing run) with shim/ORBextractor.h in the
place of include/ORBextractor.h and the reference's own ORBmatcher.h / Optimizer.h.  Needs /root/reference (this container only)."""
import os
import subprocess
import tempfile
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
pytestmark = pytest.mark.skipif(not (REF / "src" / "Tracking.cc").exists(), reason="the reference sources are not mounted")


def _make(*extra):
    return subprocess.run(["make", "-s", "-f", str(ROOT / "oracle" / "Makefile"), "callers", *extra], capture_output=True, text=True)


def test_tracking_and_local_mapping_compile_against_the_shim_headers():
    r = _make()
    assert r.returncode == 0, r.stdout + r.stderr
    assert "compile against the shim headers" in r.stdout


def test_the_shim_header_is_the_one_the_callers_see():
    """Negative control: a header that takes the include guard but lacks the constructor Tracking.cc calls must break the build -
    i.e. the reference's include/ORBextractor.h is NOT what the callers were compiled against above."""
    with tempfile.TemporaryDirectory() as d:
        fake = Path(d) / "ORBextractor.h"
        fake.write_text("#ifndef ORBEXTRACTOR_H\n#define ORBEXTRACTOR_H\n#include <vector>\n#include <opencv/cv.h>\n"
                        "namespace ORB_SLAM2 { class ORBextractor { public: ORBextractor() {} std::vector<float> GetScaleFactors(); }; }\n#endif\n")
        r = _make("SHIM=%s" % d)
    assert r.returncode != 0, "a broken drop-in header went unnoticed"
    assert "ORBextractor" in (r.stdout + r.stderr)
