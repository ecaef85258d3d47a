"""The host-to-host batch entry point as a pipeline: orbx_extract_batch_begin / _end and the chunked orbx_extract_batch,
against the CPU restatement, bit for bit (keypoint fields as float bit patterns, descriptor bytes)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def kp_bits(k):
    return np.stack([k[c].astype(np.float32) for c in ("x", "y", "size", "angle", "response", "octave", "class_id")], 1).view(np.uint32)


def check(orc_rst, frames, kps, desc, counts):
    for f, im in enumerate(frames):
        ko, do = orc_rst.extract(im)
        n = int(counts[f])
        assert n == len(ko), "frame %d: %d keypoints, the oracle has %d" % (f, n, len(ko))
        assert (kp_bits(kps[f, :n]) == ko.view(np.uint32)).all(), "frame %d keypoints" % f
        assert (desc[f, :n] == do).all(), "frame %d descriptors" % f


def test_eight_interleaved_batches_equal_the_oracle(orbx, oracle):
    """begin(0), begin(1), end(0), begin(2), end(1), ... : eight batches of twelve distinct frames, two always in flight."""
    W, H, nf, B, NB = 320, 240, 500, 12, 8
    rst = oracle.restatement(nf)
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    batches = [[orbx.synth_frame(100 * b + i, W, H, orbx.SYNTH_LOW_TEXTURE if (b + i) % 7 == 3 else 0) for i in range(B)] for b in range(NB)]
    results = []
    ext.extract_batch_begin(batches[0])
    for b in range(1, NB):
        ext.extract_batch_begin(batches[b])
        results.append(ext.extract_batch_end())
    results.append(ext.extract_batch_end())
    for b in range(NB):
        check(rst, batches[b], *results[b])


def test_third_begin_and_lonely_end_are_state_errors(orbx):
    W, H, B = 320, 240, 4
    ext = orbx.ORBextractor(500, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    fr = [orbx.synth_frame(i, W, H) for i in range(B)]
    with pytest.raises(orbx.OrbxError):
        ext.extract_batch_end()
    ext.extract_batch_begin(fr)
    ext.extract_batch_begin(fr)
    with pytest.raises(orbx.OrbxError):
        ext.extract_batch_begin(fr)
    a = ext.extract_batch_end()
    b = ext.extract_batch_end()
    assert (a[2] == b[2]).all() and (a[1] == b[1]).all()
    with pytest.raises(orbx.OrbxError):
        ext.extract_batch_end()


def test_chunked_synchronous_call_equals_the_single_stage_call(orbx, oracle, monkeypatch):
    """orbx_extract_batch over 160 frames runs as three chunks (64 + 64 + 32) through the pipeline: identical to the oracle on a sample and to itself
    across calls that reuse the slots."""
    W, H, nf, B = 320, 240, 500, 160
    rst = oracle.restatement(nf)
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    frames = [orbx.synth_frame(7000 + i, W, H, orbx.SYNTH_LOW_TEXTURE if i % 16 == 15 else 0) for i in range(B)]
    k1, d1, c1 = ext.extract_batch(frames)
    k2, d2, c2 = ext.extract_batch(frames[::-1])
    assert (c1 == c2[::-1]).all() and (d1 == d2[::-1]).all()
    sample = [0, 1, 63, 64, 65, 127, 128, 159]
    check(rst, [frames[i] for i in sample], k1[sample], d1[sample], c1[sample])


def test_pinned_frames_are_read_in_place(orbx, oracle):
    """Frames in hipHostMalloc'ed memory skip the staging copy (hipPointerGetAttributes); ragged stride."""
    import torch
    W, H, nf, B = 322, 240, 500, 6
    stride = 352
    rst = oracle.restatement(nf)
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    pin = torch.empty((B, H, stride), dtype=torch.uint8).pin_memory()
    host = pin.numpy()
    frames = [orbx.synth_frame(40 + i, W, H) for i in range(B)]
    for i in range(B):
        host[i, :, :W] = frames[i]
    views = [host[i, :, :W] for i in range(B)]
    ext.extract_batch_begin(views)
    check(rst, frames, *ext.extract_batch_end())


def test_a_batch_of_another_geometry_may_be_begun_before_the_first_is_ended(orbx, oracle):
    """begin(320x240), begin(400x300), end, end on one handle: the second begin rebuilds the handle's geometry while the first batch's results wait
    in their slot (with the capacity they were produced at)."""
    rst = oracle.restatement(500)
    ext = orbx.ORBextractor(500, 1.2, 8, 20, 7, max_width=400, max_height=300, max_batch=4)
    a = [orbx.synth_frame(60 + i, 320, 240) for i in range(4)]
    b = [orbx.synth_frame(70 + i, 400, 300) for i in range(3)]
    ext.extract_batch_begin(a)
    ext.extract_batch_begin(b)
    check(rst, a, *ext.extract_batch_end())
    check(rst, b, *ext.extract_batch_end())


def test_random_begin_end_schedule(orbx, oracle):
    """Sixty batches of random size (1 .. 40 frames), geometry (two sizes), memory kind (pageable / pinned, packed or ragged stride) and schedule (one or
    two batches begun before an end) on ONE handle; every batch's counts and a sample of its frames against the oracle."""
    import torch
    rng = np.random.default_rng(5)
    rst = oracle.restatement(500)
    GEOM = [(320, 240), (402, 300)]
    ext = orbx.ORBextractor(500, 1.2, 8, 20, 7, max_width=402, max_height=300, max_batch=40)
    pool = {g: [orbx.synth_frame(500 + 10 * gi + i, g[0], g[1], orbx.SYNTH_LOW_TEXTURE if i == 5 else 0) for i in range(8)] for gi, g in enumerate(GEOM)}
    want = {g: [rst.extract(im) for im in pool[g]] for g in GEOM}
    keep = []          # pinned tensors stay alive until their batch is ended
    pending = []

    def begin():
        g = GEOM[int(rng.integers(0, 2))]
        B = int(rng.integers(1, 41))
        ids = rng.integers(0, 8, B)
        kind = int(rng.integers(0, 3))
        if kind == 0:
            views = [pool[g][i] for i in ids]
        else:
            stride = g[0] if kind == 1 else g[0] + int(rng.integers(1, 9)) * 4
            t = torch.empty((B, g[1], stride), dtype=torch.uint8).pin_memory()
            h = t.numpy()
            for k, i in enumerate(ids):
                h[k, :, :g[0]] = pool[g][i]
            keep.append(t)
            views = [h[k, :, :g[0]] for k in range(B)]
        ext.extract_batch_begin(views)
        pending.append((g, ids))

    def end():
        g, ids = pending.pop(0)
        kps, desc, counts = ext.extract_batch_end()
        assert len(counts) == len(ids)
        for k in {0, len(ids) - 1, len(ids) // 2}:
            ko, do = want[g][ids[k]]
            n = int(counts[k])
            assert n == len(ko) and (kp_bits(kps[k, :n]) == ko.view(np.uint32)).all() and (desc[k, :n] == do).all(), (g, k)
        assert all(int(counts[k]) == len(want[g][ids[k]][0]) for k in range(len(ids)))

    for _ in range(60):
        begin()
        if len(pending) == 2 or rng.random() < 0.4:
            end()
    while pending:
        end()


def test_mixed_pinned_and_pageable_frames_in_one_batch(orbx, oracle):
    """Pinned - pageable - pinned (and every other arrangement of two memory kinds over five frames that has a pageable frame strictly inside, at the
    front or at the back): the gather kernel / the DMA engines must never be handed the pageable frame's address (round-5 advice: the decision looked
    at the first byte of the first frame and the last byte of the last frame only).  Also a frame whose LAST rows leave a registered range."""
    import torch
    W, H, nf, B = 320, 240, 500, 5
    rst = oracle.restatement(nf)
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    frames = [orbx.synth_frame(900 + i, W, H) for i in range(B)]
    want = [rst.extract(im) for im in frames]
    pin = torch.empty((B, H, W), dtype=torch.uint8).pin_memory()
    ph = pin.numpy()
    for i in range(B):
        ph[i] = frames[i]
    for mask in (0b00100, 0b01010, 0b00001, 0b10000, 0b01110, 0b11111, 0b00000):
        views = [frames[i] if (mask >> i) & 1 else ph[i] for i in range(B)]
        ext.extract_batch_begin(views)
        kps, desc, counts = ext.extract_batch_end()
        for i in range(B):
            ko, do = want[i]
            n = int(counts[i])
            assert n == len(ko) and (kp_bits(kps[i, :n]) == ko.view(np.uint32)).all() and (desc[i, :n] == do).all(), (bin(mask), i)
    # two separate pinned allocations and a frame view that starts in pinned memory and ends in pageable memory cannot be built from numpy without
    # copying; what can: the frames of one batch spread over THREE pinned allocations (more distinct ranges than one query answers)
    pins = [torch.empty((H, W), dtype=torch.uint8).pin_memory() for _ in range(B)]
    for i in range(B):
        pins[i].numpy()[:] = frames[i]
    ext.extract_batch_begin([p.numpy() for p in pins])
    kps, desc, counts = ext.extract_batch_end()
    for i in range(B):
        ko, do = want[i]
        n = int(counts[i])
        assert n == len(ko) and (kp_bits(kps[i, :n]) == ko.view(np.uint32)).all() and (desc[i, :n] == do).all(), i


def test_registered_range_that_ends_inside_a_frame(orbx, oracle):
    """hipHostRegister over the first frame and a half of a pageable two-frame array: frame 1 starts in registered memory and ends in pageable memory -
    the batch must be staged (first-byte-only checks would hand the device an address it cannot read to the end)."""
    W, H, nf = 320, 240, 500
    rst = oracle.restatement(nf)
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2)
    hip = ctypes.CDLL("libamdhip64.so")
    page = 4096
    raw = np.zeros(2 * W * H + 2 * page, np.uint8)
    off = (-raw.ctypes.data) % page
    arr = raw[off:off + 2 * W * H].reshape(2, H, W)
    frames = [orbx.synth_frame(950 + i, W, H) for i in range(2)]
    arr[0] = frames[0]; arr[1] = frames[1]
    nreg = (W * H + W * H // 2) // page * page
    assert hip.hipHostRegister(ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(nreg), 0) == 0
    try:
        ext.extract_batch_begin([arr[0], arr[1]])
        check(rst, frames, *ext.extract_batch_end())
    finally:
        assert hip.hipHostUnregister(ctypes.c_void_p(arr.ctypes.data)) == 0
    # the same array, no longer registered: pageable, staged
    ext.extract_batch_begin([arr[0], arr[1]])
    check(rst, frames, *ext.extract_batch_end())


def test_device_views_after_a_chunked_host_batch_are_state_errors(orbx):
    """A host batch that ran in chunks leaves only its last chunk on the device: the "last batch" device views refuse (ORBX_ERR_STATE) instead of
    indexing frame f of the last chunk; a device-resident call afterwards makes them valid again."""
    W, H, nf, B = 320, 240, 300, 130
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    frames = [orbx.synth_frame(8000 + (i % 7), W, H) for i in range(B)]
    kps, desc, counts = ext.extract_batch(frames)
    assert (counts > 0).all()
    with pytest.raises(orbx.OrbxError) as e:
        ext.results_device()
    assert e.value.code == -5 and "chunk" in str(e.value)
    with pytest.raises(orbx.OrbxError):
        ext.mvImagePyramid(1, frame=0)
    dev = ext.upload(frames[:8])
    ext.run_device(*dev)
    k2, d2, c2 = ext.download(8)
    assert (c2 == counts[:8]).all() and (d2[:, :] == desc[:8]).all()
