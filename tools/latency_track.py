#!/usr/bin/env python3
"""Per-frame tracking time of the drop-in call chain beside the reference's (tools/latency_shim.py: tracking) - one JSON row per library."""
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools")); sys.path.insert(0, str(ROOT / "tests"))
import latency_shim as ls      # noqa: E402

def trace(orbx, nframes=20):
    """Timeline marks of the shim (orbx_shim_trace) over one warm sequence: median microseconds between consecutive marks, i.e. where the host time of a
    tracked frame goes - marshalling of the object graph, the device call, the write-back."""
    import collections
    import ctypes
    import numpy as np
    import oracle_lib
    lib = oracle_lib.slam_hip_lib()
    ls.tracking(orbx, 1, 0, nframes=nframes)                     # warm
    lib.orbx_shim_trace.argtypes = [ctypes.c_int]
    lib.orbx_shim_trace(1)
    ls.tracking(orbx, 1, 0, nframes=nframes)
    buf = ctypes.create_string_buffer(1 << 20)
    lib.orbx_shim_trace_dump.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.orbx_shim_trace_dump(buf, len(buf))
    lib.orbx_shim_trace(0)
    ev = []
    for l in buf.value.decode().splitlines():
        parts = l.split("us  [thread")
        ev.append((float(parts[0]), parts[1].split("]  ", 1)[1]))
    spans = collections.OrderedDict()
    for (t0, a), (t1, b) in zip(ev, ev[1:]):
        spans.setdefault("%s -> %s" % (a, b), []).append(t1 - t0)
    for k, v in spans.items():
        if len(v) >= 3:
            print("%8.1f us median  x %3d   %s" % (float(np.median(v)), len(v), k))


if __name__ == "__main__":
    orbx = importlib.import_module("self_commit_orb-slam2_amd")
    if "--trace" in sys.argv:
        trace(orbx)
        sys.exit(0)
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    for r in ls.tracking(orbx, runs, 0 if "--no-ref" in sys.argv else 1):
        print(json.dumps(r), flush=True)
