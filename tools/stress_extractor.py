#!/usr/bin/env python3
"""Randomised HIP-vs-restatement sweep of ORBextractor::operator() (GPU box): image sizes, feature counts, scale factors,
level counts and FAST thresholds drawn at random, synthetic scenes, noise and flat images; keypoints and descriptors
must be bit-exact.  python tools/stress_extractor.py [cases]"""
import importlib
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
orbx = importlib.import_module("self_commit_orb-slam2_amd")
import oracle_lib  # noqa: E402


def kp_matrix(k):
    return np.stack([k[c].astype(np.float32) for c in ("x", "y", "size", "angle", "response", "octave", "class_id")], 1)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    orc = oracle_lib.Oracle()
    rng = np.random.default_rng(77)
    bad, t0, total_kp = 0, time.time(), 0
    for it in range(cases):
        W, H = int(rng.integers(140, 1300)), int(rng.integers(120, 800))
        nf = int(rng.choice([100, 500, 1000, 2000, 3000]))
        sf = float(rng.choice([1.1, 1.2, 1.3, 1.5]))
        nl = int(rng.integers(1, 11))
        # the reference needs every level to keep a positive detection window (w, h > 2*16 + 6)
        while nl > 1 and min(W, H) / sf ** (nl - 1) < 45:
            nl -= 1
        ini, mn = int(rng.choice([10, 20, 30])), int(rng.choice([3, 7]))
        kind = rng.integers(0, 4)
        frames = []
        for b in range(2):
            if kind == 0:
                im = orbx.synth_frame(int(rng.integers(1, 1 << 30)), W, H)
            elif kind == 1:
                im = orbx.synth_frame(int(rng.integers(1, 1 << 30)), W, H, orbx.SYNTH_LOW_TEXTURE)
            elif kind == 2:
                im = rng.integers(0, 256, (H, W), dtype=np.uint8)
            else:
                im = np.full((H, W), int(rng.integers(0, 256)), np.uint8)
                im[H // 3:H // 3 + 40, W // 3:W // 3 + 40] = 255 - im[0, 0]
            frames.append(np.ascontiguousarray(im))
        try:
            ext = orbx.ORBextractor(nf, sf, nl, ini, mn, max_width=W, max_height=H, max_batch=2)
        except orbx.OrbxError as e:
            print("skip (geometry rejected):", W, H, nf, sf, nl, str(e)[:80])
            continue
        rst = orc.restatement(nf, sf, nl, ini, mn)
        try:
            kps, desc, counts = ext.extract_batch(frames)
        except orbx.OrbxError as e:      # documented capacity limit (e.g. > 32768 FAST candidates in one level of a pure-noise image): loud, not wrong
            print("skip (capacity):", W, H, nf, sf, nl, kind, str(e)[:90])
            ext.close()
            continue
        for f, im in enumerate(frames):
            ko, do = rst.extract(im)
            n = int(counts[f])
            total_kp += n
            if n != len(ko) or not (kp_matrix(kps[f, :n]).view(np.uint32) == ko.view(np.uint32)).all() or not (desc[f, :n] == do).all():
                bad += 1
                print("MISMATCH", it, W, H, nf, sf, nl, ini, mn, kind, n, len(ko))
        ext.close()
    print("stress: %d geometries x 2 frames, %d keypoints, %d mismatches, %.1f s" % (cases, total_kp, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
