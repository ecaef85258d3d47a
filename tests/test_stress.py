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


@pytest.mark.gpu
def test_combiner_soak_sixteen_threads_three_geometries(orbx, oracle):
    """Ten seconds of ORBextractor::operator()-shaped calls from sixteen threads through the combiner: three image geometries, handles created and
    destroyed on the way (the last handle of a geometry releases its engine set), a sample of every thread's results checked against the oracle,
    device memory before and after (what the engine sets held must be back)."""
    import threading
    import time

    import numpy as np
    import torch
    GEOMS = [(640, 480, 1000), (752, 480, 1200), (320, 240, 500)]
    rsts = {g: oracle.restatement(g[2]) for g in GEOMS}
    imgs = {g: [orbx.synth_frame(900 + 7 * i + g[0], g[0], g[1], orbx.SYNTH_LOW_TEXTURE if i == 3 else 0) for i in range(4)] for g in GEOMS}
    want = {g: [rsts[g].extract(im) for im in imgs[g]] for g in GEOMS}

    def kp_bits(k):
        return np.stack([k[c].astype(np.float32) for c in ("x", "y", "size", "angle", "response", "octave", "class_id")], 1).view(np.uint32)

    # one warm-up handle per geometry (the HIP context, the first engine sets), then everything is released again
    for g in GEOMS:
        e = orbx.ORBextractor(g[2], 1.2, 8, 20, 7, max_width=g[0], max_height=g[1], max_batch=1)
        e(imgs[g][0]); e.close()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    deadline = time.time() + 10.0
    errors, frames, handles = [], [0] * 16, [0] * 16

    def work(t):
        rng = np.random.default_rng(t)
        try:
            while time.time() < deadline:
                g = GEOMS[int(rng.integers(0, 3))]
                e = orbx.ORBextractor(g[2], 1.2, 8, 20, 7, max_width=g[0], max_height=g[1], max_batch=1)
                handles[t] += 1
                for it in range(int(rng.integers(20, 200))):
                    i = int(rng.integers(0, 4))
                    k, d = e(imgs[g][i])
                    frames[t] += 1
                    if it % 16 == 0:
                        ko, do = want[g][i]
                        if len(k) != len(ko) or not (kp_bits(k) == ko.view(np.uint32)).all() or not (d == do).all():
                            errors.append((t, g, i, len(k), len(ko)))
                            return
                    if time.time() >= deadline:
                        break
                e.close()
        except Exception as ex:      # noqa: BLE001
            errors.append((t, repr(ex)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(16)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:3]
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    print("combiner soak: %d frames, %d handles in 10 s on 16 threads; device memory free before %.1f MB, after %.1f MB" %
          (sum(frames), sum(handles), free0 / 2**20, free1 / 2**20))
    assert sum(frames) > 20000, sum(frames)
    assert free0 - free1 < 64 * 2**20, "engine sets were not released: %.1f MB still held" % ((free0 - free1) / 2**20)
