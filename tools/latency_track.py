#!/usr/bin/env python3
"""Per-frame tracking time of the drop-in call chain beside the reference's (tools/latency_shim.py: tracking) - one JSON row per library."""
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools")); sys.path.insert(0, str(ROOT / "tests"))
import latency_shim as ls      # noqa: E402

if __name__ == "__main__":
    orbx = importlib.import_module("self_commit_orb-slam2_amd")
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    for r in ls.tracking(orbx, runs, 0 if "--no-ref" in sys.argv else 1):
        print(json.dumps(r), flush=True)
