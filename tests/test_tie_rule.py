"""What "bit-exact vs the compiled reference" means against a stock ORB-SLAM2 binary.

DistributeOctTree sorts pair<int, ExtractorNode*> (reference src/ORBextractor.cc:948): nodes with equal point counts are split
in the order of their HEAP ADDRESSES.  The project pins that order ("the later-created node first" = the unmodified reference
under a monotone bump allocator); a stock build runs on glibc malloc, whose addresses follow the heap's
history.  This test runs the SAME unmodified reference source both ways and measures how far apart the outputs are, so the
number is on record and regressions of the harness are caught."""
import numpy as np
import pytest

FRAMES = 60


def _keyset(k):
    return set(map(tuple, np.ascontiguousarray(k[:, [0, 1, 5]]).astype(np.float32).tolist()))   # (x, y, octave)


def test_stock_allocator_divergence_is_small_and_measured(orbx, oracle):
    if oracle.ref is None or not hasattr(oracle.ref, "orbref_set_arena"):
        pytest.skip("oracle/_ref/liborbref.so not built")
    frames = [orbx.synth_frame(3000 + i, 640, 480, orbx.SYNTH_LOW_TEXTURE if i % 16 == 15 else 0) for i in range(FRAMES)]
    ext = oracle.reference(1000)
    pinned = [ext.extract(f) for f in frames]
    oracle.ref.orbref_set_arena(0)
    try:
        ext2 = oracle.reference(1000)
        stock = [ext2.extract(f) for f in frames]
    finally:
        oracle.ref.orbref_set_arena(1)
    again = [ext.extract(f) for f in frames]
    for (k0, d0), (k1, d1) in zip(pinned, again):                       # the pinned rule is reproducible ...
        assert k0.shape == k1.shape and (k0.view(np.uint32) == k1.view(np.uint32)).all() and (d0 == d1).all()
    differ, missing, total, dn = 0, 0, 0, 0
    for (kp, dp), (ks, ds) in zip(pinned, stock):
        a, b = _keyset(kp), _keyset(ks)
        total += len(b)
        missing += len(b - a)                                           # keypoints of the stock build the pinned rule does not produce
        dn += abs(len(a) - len(b))
        if kp.shape != ks.shape or not (kp.view(np.uint32) == ks.view(np.uint32)).all():
            differ += 1
    frac = missing / max(total, 1)
    print("\ntie rule: %d of %d frames differ, %d of %d keypoints (%.2f %%) of the stock-malloc build are not in the pinned output, "
          "sum |count difference| = %d" % (differ, FRAMES, missing, total, 100 * frac, dn))
    # ... and the allocator only permutes which of equally-populated nodes split first: a small fraction of the keypoints moves
    assert frac < 0.05
    # every keypoint either build selects is a cell-NMS FAST candidate with identical response in both (same detector, same images):
    # keypoints present in both carry identical angle / response / descriptor
    for (kp, dp), (ks, ds) in zip(pinned[:8], stock[:8]):
        ip = {tuple(r[[0, 1, 5]]): i for i, r in enumerate(kp.tolist() and np.asarray(kp))}
        for j, r in enumerate(np.asarray(ks)):
            i = ip.get(tuple(r[[0, 1, 5]]))
            if i is not None:
                assert (kp[i].view(np.uint32) == ks[j].view(np.uint32)).all() and (dp[i] == ds[j]).all()
