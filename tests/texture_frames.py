"""Seeded test images that are NOT rectangles + triangles + uniform noise (what csrc/orbx_synth.cc renders): smooth gradients, saturated
0 / 255 plateaus, 1/f-like texture, a dense checker.  Integer arithmetic only (numpy int64 on PCG64 integers), so the same bytes come out
on every machine; the goldens carry a CRC of each image.  Used by tools/gen_golden_textures.py (reference outputs) and tests/test_golden.py."""
import zlib

import numpy as np

KINDS = ("gradient", "plateaus", "pink", "checker", "flat", "weak")


def _value_noise(rng, W, H, cell, amp):
    """Bilinear interpolation (fixed point, 8 fractional bits) of a random grid with `cell`-pixel spacing, values in [-amp, amp]."""
    gw, gh = W // cell + 2, H // cell + 2
    g = rng.integers(-amp, amp + 1, (gh, gw)).astype(np.int64)
    y, x = np.arange(H, dtype=np.int64)[:, None], np.arange(W, dtype=np.int64)[None, :]
    gy, gx = y // cell, x // cell
    fy, fx = (y % cell) * 256 // cell, (x % cell) * 256 // cell
    a, b = g[gy, gx], g[gy, gx + 1]
    c, d = g[gy + 1, gx], g[gy + 1, gx + 1]
    top = a * (256 - fx) + b * fx
    bot = c * (256 - fx) + d * fx
    return (top * (256 - fy) + bot * fy) >> 16


def texture_frame(kind, seed, W, H):
    rng = np.random.Generator(np.random.PCG64(seed * 7919 + KINDS.index(kind)))
    y, x = np.arange(H, dtype=np.int64)[:, None], np.arange(W, dtype=np.int64)[None, :]
    if kind == "gradient":        # smooth ramps + weak noise: most cells find nothing at iniThFAST and retry at minThFAST (src/ORBextractor.cc:1132-1139)
        a, b = int(rng.integers(40, 200)), int(rng.integers(40, 200))
        im = 20 + (x * a) // W + (y * b) // H // 2 + _value_noise(rng, W, H, 64, 30) + rng.integers(-5, 6, (H, W))
    elif kind == "plateaus":      # large areas clipped to 0 and to 255 with textured rims
        im = 128 + 6 * _value_noise(rng, W, H, 48, 60) + _value_noise(rng, W, H, 6, 25)
    elif kind == "pink":          # octaves with halving amplitude: 1/f-like
        im = 128 + sum(_value_noise(rng, W, H, c, amp) for c, amp in ((64, 64), (32, 48), (16, 32), (8, 24), (4, 16), (2, 12))) + rng.integers(-4, 5, (H, W))
    elif kind == "checker":       # 2-pixel checker with jittered amplitudes: candidates at the density limit of the 3x3 NMS
        im = 128 + np.where(((x >> 1) + (y >> 1)) & 1, 1, -1) * rng.integers(30, 120, (H, W)) + rng.integers(-3, 4, (H, W))
    elif kind == "flat":          # almost nothing to find: levels leave the quadtree far below their quota (:910)
        im = 100 + _value_noise(rng, W, H, 96, 12) + rng.integers(-1, 2, (H, W))
        im[H // 3:H // 3 + 40, W // 4:W // 4 + 60] += 70          # one bright box: a few strong corners
    elif kind == "weak":          # low-contrast texture: corners mostly between minThFAST and iniThFAST, levels end below their quota
        im = 128 + _value_noise(rng, W, H, 5, 22) + _value_noise(rng, W, H, 40, 40) + rng.integers(-2, 3, (H, W))
    else:
        raise ValueError(kind)
    return np.clip(im, 0, 255).astype(np.uint8)


def crc(im):
    return zlib.crc32(np.ascontiguousarray(im).tobytes())
