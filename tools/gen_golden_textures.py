#!/usr/bin/env python3
"""tests/golden/textures_*.npz: outputs of the COMPILED REFERENCE (oracle/_ref/liborbref.so = unmodified src/ORBextractor.cc on
oracle/cvshim) on the images of tests/texture_frames.py - smooth gradients, 0 / 255 plateaus, 1/f-like texture, a dense checker,
a nearly empty frame.  Same role as tools/gen_golden.py; run where /root/reference is mounted:  python tools/gen_golden_textures.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib  # noqa: E402
import texture_frames as tf  # noqa: E402

CASES = [("textures_640x480_1000", 640, 480, 1000, 11), ("textures_1241x376_2000", 1241, 376, 2000, 12), ("textures_752x480_1200", 752, 480, 1200, 13)]


def main():
    orc = oracle_lib.Oracle()
    assert orc.ref is not None, "build oracle/_ref first (needs /root/reference)"
    for name, W, H, nf, seed in CASES:
        ref = orc.reference(nf)
        data = {"W": W, "H": H, "nfeatures": nf, "seed": seed, "kinds": np.array(tf.KINDS)}
        for kind in tf.KINDS:
            im = tf.texture_frame(kind, seed, W, H)
            k, d = ref.extract(im, cap=16384)
            data["kps_" + kind], data["desc_" + kind], data["crc_" + kind] = k, d, np.int64(tf.crc(im))
            per_level = np.bincount(k[:, 5].astype(int), minlength=8)
            print(name, kind, len(k), per_level.tolist(), "img mean %.1f, 0s %.3f, 255s %.3f" % (im.mean(), (im == 0).mean(), (im == 255).mean()))
        np.savez_compressed(ROOT / "tests" / "golden" / (name + ".npz"), **data)


if __name__ == "__main__":
    main()
