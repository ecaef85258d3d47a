"""CPU tests of the checkers themselves: known-answer tests of the restated OpenCV
primitives, the restatement against the compiled reference (when oracle/_ref exists), the
array-form quadtree against the std::list original on random candidate sets, and the
sin/cos restatement against this box's libm over every float the path can produce."""
import ctypes
import zlib
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_pattern_table_crc():
    for p in (ROOT / "oracle" / "orb_pattern.inc", ROOT / "self_commit_orb-slam2_amd" / "csrc" / "orb_pattern.inc"):
        vals = []
        for line in p.read_text().splitlines():
            if line.startswith("//"):
                continue
            vals += [int(t) for t in line.strip().strip(",").split(",") if t]
        assert len(vals) == 1024
        assert zlib.crc32(bytes((v + 256) % 256 for v in vals)) == 0xD1A39030
        # first test pair of the ORB pattern (reference src/ORBextractor.cc:233)
        assert vals[:4] == [8, -3, 9, 5]


def test_tables_tum1(oracle):
    t, q, u = oracle.restatement(1000).tables()
    assert list(q) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(u) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert t[0][0] == 1.0 and abs(t[0][7] - 1.2 ** 7) < 1e-5
    t2, q2, _ = oracle.restatement(2000).tables()
    assert list(q2) == [434, 362, 302, 251, 209, 175, 145, 122]
    ext = oracle.restatement(1000)
    sizes = [oracle.level_size(ext, 640, 480, l) for l in range(8)]
    assert sizes == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]


def test_gaussian_impulse_and_constant(oracle):
    im = np.zeros((31, 31), np.uint8)
    im[15, 15] = 255
    b = oracle.blur(im).astype(np.int64)
    taps = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)
    want = (np.outer(taps, taps) * 255 + 32768) >> 16
    assert (b[12:19, 12:19] == want).all()
    assert b.sum() == want.sum()
    c = np.full((20, 33), 201, np.uint8)
    assert (oracle.blur(c) == 201).all()          # taps sum to exactly 256
    # REFLECT_101 at the border: a ramp stays symmetric
    r = np.tile(np.arange(40, dtype=np.uint8) * 5, (12, 1))
    br = oracle.blur(r)
    assert (br[:, 5:-5] == r[:, 5:-5]).all()      # linear ramp is reproduced exactly inside


def test_resize_constant_and_identity(oracle):
    ext = oracle.restatement(1000)
    im = np.full((480, 640), 77, np.uint8)
    for lv in oracle.pyramid(ext, im):
        assert (lv == 77).all()


def test_fast_step_corner(oracle):
    # a bright quadrant on dark background: the corner pixel has 12 contiguous darker
    # circle pixels (difference 100) -> it is a FAST-9 corner with score 99
    im = np.full((40, 40), 50, np.uint8)
    im[20:, 20:] = 150
    S = oracle.score_map(im, 7)
    assert S[20, 20] == 99
    assert S[10, 10] == 0 and S[30, 30] == 0      # flat areas
    assert S[20, 30] == 0                          # straight edge is not a corner
    # score definition: corner at t  <=>  score >= t
    S20 = oracle.score_map(im, 20)
    assert ((S >= 20) == (S20 > 0)).all() and (S[S >= 20] == S20[S20 > 0]).all()


def test_fast_atan2_axes(oracle):
    # orientation of simple patches: bright right half -> 0 deg, bright bottom -> 90 deg
    ext = oracle.restatement(1000)
    im = np.zeros((64, 64), np.uint8)
    im[:, 33:] = 200
    assert abs(oracle.ic_angle(ext, im, 32, 32)) < 1e-3
    im = np.zeros((64, 64), np.uint8)
    im[33:, :] = 200
    assert abs(oracle.ic_angle(ext, im, 32, 32) - 90.0) < 1e-2
    im = np.zeros((64, 64), np.uint8)
    im[:, :32] = 200
    assert abs(oracle.ic_angle(ext, im, 32, 32) - 180.0) < 1e-2
    im = np.zeros((64, 64), np.uint8)
    im[:32, :] = 200
    assert abs(oracle.ic_angle(ext, im, 32, 32) - 270.0) < 1e-2


def test_sincos_restatement_equals_libm(oracle):
    """Every float in [0, 6.5] (all angle*pi/180 values lie in [0, 2*pi]): the restated
    glibc algorithm (oracle/prims.h op_sincosf, copied on the device) == this box's libm."""
    lo = np.array([0.0], np.float32).view(np.uint32)[0]
    hi = np.array([6.5], np.float32).view(np.uint32)[0]
    assert oracle.lib.orbo_sincos_exhaustive(int(lo), int(hi)) == 0


@pytest.mark.parametrize("W,H,nf,seeds", [(640, 480, 1000, [21, 22, 23]), (1241, 376, 2000, [24]), (752, 480, 1200, [25])])
def test_restatement_equals_compiled_reference(orbx, oracle, W, H, nf, seeds):
    ref = oracle.reference(nf)
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rst = oracle.restatement(nf)
    assert all((a.view(np.uint32) == b.view(np.uint32)).all() for a, b in zip(ref.tables(), rst.tables()))
    for s in seeds:
        im = orbx.synth_frame(s, W, H, orbx.SYNTH_LOW_TEXTURE if s % 4 == 3 else 0)
        k1, d1 = ref.extract(im)
        k2, d2 = rst.extract(im)
        assert k1.shape == k2.shape and (k1.view(np.uint32) == k2.view(np.uint32)).all() and (d1 == d2).all()


def test_octree_array_form_equals_std_list(oracle):
    """Random candidate sets (ties at the careful-round break are the
    norm): array-form quadtree == DistributeOctTree of the compiled reference."""
    ref = oracle.reference(1000)
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(7)
    for trial in range(60):
        w, h = [(640, 480), (1241, 376), (309, 231), (179, 134)][trial % 4]
        ww, hh = w - 32, h - 32
        n = int(rng.integers(1, 4000))
        xy = rng.integers([3, 3], [ww - 3, hh - 3], size=(n * 2, 2))
        xy = np.unique(xy, axis=0)
        rng.shuffle(xy)
        xy = xy[:n]
        # candidates arrive in cell-raster order in the real pipeline; any order is legal input
        sc = rng.integers(7, 120, size=len(xy))
        packed = (xy[:, 0].astype(np.uint32) | (xy[:, 1].astype(np.uint32) << 12) | (sc.astype(np.uint32) << 24))
        N = int(rng.integers(1, 500))
        a = oracle.octree(packed, w, h, N)
        b = oracle.ref_octree(ref, packed, w, h, N)
        assert len(a) == len(b) and (a == b).all(), "trial %d" % trial
