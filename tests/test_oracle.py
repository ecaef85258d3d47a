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


@pytest.mark.parametrize("W,H,nl", [(1000, 230, 4), (1241, 230, 4), (1400, 200, 2)])
def test_restatement_equals_compiled_reference_on_wide_images(orbx, oracle, W, H, nl):
    """Five to eight initial quadtree nodes (round(width / height) of the detection window, src/ORBextractor.cc:719): the restatement the GPU
    tests of these geometries compare with is pinned to the compiled reference here."""
    ref = oracle.reference(800, 1.2, nl)
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rst = oracle.restatement(800, 1.2, nl)
    for s in (41, 43):
        im = orbx.synth_frame(s, W, H, orbx.SYNTH_LOW_TEXTURE if s % 4 == 3 else 0)
        k1, d1 = ref.extract(im)
        k2, d2 = rst.extract(im)
        assert len(k1) > 100 and k1.shape == k2.shape and (k1.view(np.uint32) == k2.view(np.uint32)).all() and (d1 == d2).all()


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


# ------------------------------------------------------------------------------------------------
# The three OpenCV primitives are restated (OpenCV is not vendored in the reference and not installed here: parity unpinned).  These
# checks hold the restatements to the primitives' MATHEMATICAL definitions, computed independently in numpy - not to OpenCV's code.
def _fast9_bruteforce(im, t):
    """FAST-9/16 by definition: corner iff 9 contiguous circle pixels are all > v + t or all < v - t; score = the largest t for which
    the pixel is still a corner (0 if it is none at threshold t)."""
    dx = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
    dy = [-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3]
    H, W = im.shape
    v = im[3:H - 3, 3:W - 3].astype(np.int32)
    ring = np.stack([im[3 + dy[k]:H - 3 + dy[k], 3 + dx[k]:W - 3 + dx[k]].astype(np.int32) for k in range(16)], 0)
    ring2 = np.concatenate([ring, ring[:8]], 0)
    best = np.full(v.shape, -1, np.int32)
    for s in range(16):                                    # arc of 9 starting at s
        arc = ring2[s:s + 9]
        best = np.maximum(best, np.maximum(arc.min(0) - v, v - arc.max(0)))      # bright arc: min(arc) - v > t ; dark arc: v - max(arc) > t
    score = np.where(best - 1 >= t, best - 1, 0)           # corner at t <=> best > t ; largest such t is best - 1
    out = np.zeros(im.shape, np.int32)
    out[3:H - 3, 3:W - 3] = score
    return out


@pytest.mark.parametrize("seed,t", [(1, 7), (2, 20), (3, 1), (4, 40)])
def test_fast_score_equals_the_definition(oracle, seed, t):
    rng = np.random.default_rng(seed)
    im = rng.integers(0, 256, (61, 83)).astype(np.uint8)
    im[20:40, 30:60] = (im[20:40, 30:60] // 8 + 100).astype(np.uint8)      # a low-contrast patch next to noise
    S = oracle.score_map(im, t).astype(np.int32)
    want = _fast9_bruteforce(im, t)
    assert (S[3:-3, 3:-3] == np.minimum(want, 255)[3:-3, 3:-3]).all()


def test_gaussian_blur_against_the_real_valued_filter(oracle):
    """cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101): the fixed-point result vs the float64 separable filter with
    getGaussianKernel's formula exp(-x^2 / (2 sigma^2)) normalised to 1."""
    rng = np.random.default_rng(5)
    im = rng.integers(0, 256, (45, 70)).astype(np.uint8)
    x = np.arange(-3, 4, dtype=np.float64)
    k = np.exp(-x * x / 8.0)
    k /= k.sum()
    p = np.pad(im.astype(np.float64), 3, mode="reflect")      # numpy 'reflect' == BORDER_REFLECT_101
    h = sum(k[i] * p[:, i:i + 70] for i in range(7))
    ref = sum(k[i] * h[i:i + 45, :] for i in range(7))
    got = oracle.blur(im).astype(np.float64)
    # the 8-bit taps (18 34 48 56 48 34 18) / 256 are the real kernel (17.96 33.56 48.82 55.32 ...) / 256 rounded so that they sum to 1:
    # up to 0.82 / 256 apart, i.e. at most ~1.2 grey levels on white noise; against the taps themselves only the final rounding remains
    assert np.abs(k * 256 - np.array([18, 34, 48, 56, 48, 34, 18])).max() < 0.85
    assert np.abs(got - ref).max() <= 1.5 and np.abs(got - ref).mean() < 0.4
    kt = np.array([18, 34, 48, 56, 48, 34, 18], np.float64) / 256
    h = sum(kt[i] * p[:, i:i + 70] for i in range(7))
    ref_taps = sum(kt[i] * h[i:i + 45, :] for i in range(7))
    assert np.abs(got - ref_taps).max() <= 0.5 + 1e-9


def test_resize_within_one_level_of_real_valued_bilinear(oracle):
    """cv::resize(INTER_LINEAR): source coordinate (dst + 0.5) * scale - 0.5 with scale = src / dst, clamped at the borders; the 11-bit
    fixed-point result vs float64 bilinear interpolation, level 0 -> level 1 of the 640x480 pyramid."""
    rng = np.random.default_rng(6)
    im = rng.integers(0, 256, (480, 640)).astype(np.uint8)
    ext = oracle.restatement(1000)
    lv1 = oracle.pyramid(ext, im)[1].astype(np.float64)
    h1, w1 = lv1.shape

    def axis(n_dst, n_src):
        s = (np.arange(n_dst) + 0.5) * (n_src / n_dst) - 0.5
        i0 = np.floor(s).astype(np.int64)
        f = s - i0
        f = np.where(i0 < 0, 0.0, f)
        i0 = np.clip(i0, 0, n_src - 1)
        i1 = np.clip(i0 + 1, 0, n_src - 1)
        return i0, i1, f

    y0, y1, fy = axis(h1, 480)
    x0, x1, fx = axis(w1, 640)
    a = im.astype(np.float64)
    top = a[y0][:, x0] * (1 - fx) + a[y0][:, x1] * fx
    bot = a[y1][:, x0] * (1 - fx) + a[y1][:, x1] * fx
    ref = top * (1 - fy)[:, None] + bot * fy[:, None]
    assert np.abs(lv1 - ref).max() <= 1.0 and np.abs(lv1 - ref).mean() < 0.3
