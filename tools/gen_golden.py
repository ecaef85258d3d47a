#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/liborbref.so =
unmodified /root/reference/src/ORBextractor.cc against oracle/cvshim).

Run in the container that has /root/reference mounted:
    make -f oracle/Makefile all && python tools/gen_golden.py
The fixtures pin (a) the CPU restatement (tests/test_golden.py, CPU) and (b) the HIP path
(tests/test_golden.py -m gpu) to outputs of the reference itself on the GPU box, where
/root/reference does not exist.
"""
import importlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib  # noqa: E402

orbx = importlib.import_module("self_commit_orb-slam2_amd")

CASES = [
    # name, W, H, nfeatures, seeds(flags)
    ("tum1_640x480_1000", 640, 480, 1000, [(1, 0), (2, 0), (3, 1)]),
    ("kitti_1241x376_2000", 1241, 376, 2000, [(4, 0), (5, 1)]),
    ("euroc_752x480_1200", 752, 480, 1200, [(6, 0)]),
]


def main():
    orc = oracle_lib.Oracle()
    assert orc.ref is not None, "build oracle/_ref first (needs /root/reference)"
    out = ROOT / "tests" / "golden"
    out.mkdir(parents=True, exist_ok=True)
    for name, W, H, nf, seeds in CASES:
        ref = orc.reference(nf)
        data = {"W": W, "H": H, "nfeatures": nf, "seeds": np.array(seeds, np.int64)}
        for i, (seed, flags) in enumerate(seeds):
            im = orbx.synth_frame(seed, W, H, flags)
            k, d = ref.extract(im)
            data["kps_%d" % i] = k
            data["desc_%d" % i] = np.packbits(np.unpackbits(d, axis=1), axis=1)  # = d, explicit u8
            data["imgsum_%d" % i] = np.int64(im.astype(np.int64).sum())
        np.savez_compressed(out / (name + ".npz"), **data)
        print(name, [len(data["kps_%d" % i]) for i in range(len(seeds))])


if __name__ == "__main__":
    main()
