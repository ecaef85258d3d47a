"""Golden vectors produced by the COMPILED REFERENCE (tools/gen_golden.py, oracle/_ref):
the CPU restatement (CPU test) and the HIP path (gpu test) must both reproduce them
bit-for-bit.  The fixtures travel; /root/reference does not."""
from pathlib import Path

import numpy as np
import pytest

GOLDEN = sorted(p for p in (Path(__file__).resolve().parent / "golden").glob("*.npz") if not p.name.startswith("textures_"))


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


# ---- images that are not rectangles + triangles + noise (tests/texture_frames.py; goldens from the compiled reference, tools/gen_golden_textures.py)
TEXTURES = sorted((Path(__file__).resolve().parent / "golden").glob("textures_*.npz"))


def _texture_cases():
    for p in TEXTURES:
        z = np.load(p)
        for kind in z["kinds"]:
            yield pytest.param(p, str(kind), id="%s-%s" % (p.stem, kind))


@pytest.mark.parametrize("path,kind", list(_texture_cases()))
def test_restatement_matches_reference_on_textures(orbx, oracle, path, kind):
    import texture_frames as tf
    z = np.load(path)
    W, H, nf = int(z["W"]), int(z["H"]), int(z["nfeatures"])
    im = tf.texture_frame(kind, int(z["seed"]), W, H)
    assert tf.crc(im) == int(z["crc_" + kind]), "texture generator changed"
    k, d = oracle.restatement(nf).extract(im, cap=16384)
    assert k.shape == z["kps_" + kind].shape and (k.view(np.uint32) == z["kps_" + kind].view(np.uint32)).all()
    assert (d == z["desc_" + kind]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", TEXTURES, ids=lambda p: p.stem)
def test_hip_matches_reference_on_textures(orbx, path):
    """Smooth gradients (cells retry at minThFAST, src/ORBextractor.cc:1132-1139), 0 / 255 plateaus, 1/f-like texture, a checker at the
    density limit of the NMS, nearly empty frames (levels leave the quadtree far below their quota, :910): one frame alone and all kinds
    as one batch must equal the compiled reference bit for bit."""
    import texture_frames as tf
    z = np.load(path)
    W, H, nf = int(z["W"]), int(z["H"]), int(z["nfeatures"])
    kinds = [str(k) for k in z["kinds"]]
    ims = [tf.texture_frame(k, int(z["seed"]), W, H) for k in kinds]
    one = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H)
    for kind, im in zip(kinds, ims):
        assert tf.crc(im) == int(z["crc_" + kind])
        k, d = one(im)
        g = z["kps_" + kind]
        assert len(k) == len(g) and (_kpm(k).view(np.uint32) == g.view(np.uint32)).all(), kind
        assert (d == z["desc_" + kind]).all(), kind
    one.close()
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=len(ims))
    kps, desc, counts = ext.extract_batch(ims)
    for f, kind in enumerate(kinds):
        g, n = z["kps_" + kind], int(counts[f])
        assert n == len(g) and (_kpm(kps[f, :n]).view(np.uint32) == g.view(np.uint32)).all(), kind
        assert (desc[f, :n] == z["desc_" + kind]).all(), kind
    ext.close()
