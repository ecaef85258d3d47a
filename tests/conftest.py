import importlib
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orbx():
    """The product package (ctypes over liborbx.so)."""
    return importlib.import_module("self_commit_orb-slam2_amd")


@pytest.fixture(scope="session")
def oracle():
    """CPU checkers (test infrastructure): restatement + compiled reference when present."""
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(autouse=True)
def _native_backtrace_on_crash():
    """Debug aid: ORBX_SEGV_BT=<path of a .so exporting segv_bt_install()> (re)installs a SIGSEGV / SIGABRT handler that prints the native
    backtrace before every test (the HIP runtime replaces handlers installed earlier)."""
    p = os.environ.get("ORBX_SEGV_BT")
    if p:
        import ctypes
        ctypes.CDLL(p).segv_bt_install()
    yield
