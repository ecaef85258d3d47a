"""Golden vectors produced by the COMPILED REFERENCE (tools/gen_golden.py, oracle/_ref):
the CPU restatement (CPU test) and the HIP path (gpu test) must both reproduce them
bit-for-bit.  The fixtures travel; /root/reference does not."""
from pathlib import Path

import numpy as np
import pytest

GOLDEN = sorted((Path(__file__).resolve().parent / "golden").glob("*.npz"))


def _cases():
    for p in GOLDEN:
        z = np.load(p)
        for i, (seed, flags) in enumerate(z["seeds"]):
            yield pytest.param(p, i, int(seed), int(flags), id="%s-%d" % (p.stem, i))


def _kpm(k):
    return np.stack([k["x"], k["y"], k["size"], k["angle"], k["response"], k["octave"].astype(np.float32),
                     k["class_id"].astype(np.float32)], 1)


def test_golden_present():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize("path,i,seed,flags", list(_cases()))
def test_restatement_matches_reference_golden(orbx, oracle, path, i, seed, flags):
    z = np.load(path)
    W, H, nf = int(z["W"]), int(z["H"]), int(z["nfeatures"])
    im = orbx.synth_frame(seed, W, H, flags)
    assert int(im.astype(np.int64).sum()) == int(z["imgsum_%d" % i]), "synthetic frame generator changed"
    k, d = oracle.restatement(nf).extract(im)
    assert k.shape == z["kps_%d" % i].shape
    assert (k.view(np.uint32) == z["kps_%d" % i].view(np.uint32)).all()
    assert (d == z["desc_%d" % i]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path,i,seed,flags", list(_cases()))
def test_hip_matches_reference_golden(orbx, path, i, seed, flags):
    z = np.load(path)
    W, H, nf = int(z["W"]), int(z["H"]), int(z["nfeatures"])
    im = orbx.synth_frame(seed, W, H, flags)
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H)
    k, d = ext(im)
    ext.close()
    g = z["kps_%d" % i]
    assert len(k) == len(g)
    assert (_kpm(k).view(np.uint32) == g.view(np.uint32)).all()
    assert (d == z["desc_%d" % i]).all()
