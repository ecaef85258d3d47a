// orbx_synth.cc -- deterministic synthetic grayscale frames (host code, integer only).
//
// No dataset ships with the reference and there is no network,
// so every benchmark/parity input is produced here.  The generator is integer-only
// (xorshift64*), hence bit-identical on every host.  A frame is mid-gray canvas +
// random filled rectangles and triangles (true corners at several scales) + uniform
// noise; `low_texture` frames exercise the minThFAST fallback of the cell detector
// (reference src/ORBextractor.cc:1132-1139) and the "fewer than quota" octree exit
// (:910).  Stereo right images re-render the same shapes shifted left by a per-shape
// disparity with a fresh noise stream.
#include <stdint.h>
#include <string.h>

#include "orbx_synth.h"

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed ? seed : 0x2545F4914F6CDD1DULL) {}
    inline uint64_t next()
    {
        s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
        return s * 0x2545F4914F6CDD1DULL;
    }
    inline uint32_t below(uint32_t n) { return (uint32_t)((next() >> 33) % n); }
};

inline void fill_rect(uint8_t *img, int W, int H, int stride, int x0, int y0, int w, int h, uint8_t g)
{
    int x1 = x0 + w, y1 = y0 + h;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > W) x1 = W;
    if (y1 > H) y1 = H;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) img[(size_t)y * stride + x] = g;
}

inline int64_t edge(int ax, int ay, int bx, int by, int px, int py)
{
    return (int64_t)(bx - ax) * (py - ay) - (int64_t)(by - ay) * (px - ax);
}

inline void fill_tri(uint8_t *img, int W, int H, int stride, const int *vx, const int *vy, uint8_t g)
{
    int xmin = vx[0], xmax = vx[0], ymin = vy[0], ymax = vy[0];
    for (int i = 1; i < 3; i++) {
        if (vx[i] < xmin) xmin = vx[i];
        if (vx[i] > xmax) xmax = vx[i];
        if (vy[i] < ymin) ymin = vy[i];
        if (vy[i] > ymax) ymax = vy[i];
    }
    if (xmin < 0) xmin = 0;
    if (ymin < 0) ymin = 0;
    if (xmax >= W) xmax = W - 1;
    if (ymax >= H) ymax = H - 1;
    int64_t area = edge(vx[0], vy[0], vx[1], vy[1], vx[2], vy[2]);
    if (area == 0) return;
    for (int y = ymin; y <= ymax; y++)
        for (int x = xmin; x <= xmax; x++) {
            int64_t e0 = edge(vx[0], vy[0], vx[1], vy[1], x, y);
            int64_t e1 = edge(vx[1], vy[1], vx[2], vy[2], x, y);
            int64_t e2 = edge(vx[2], vy[2], vx[0], vy[0], x, y);
            bool in = area > 0 ? (e0 >= 0 && e1 >= 0 && e2 >= 0) : (e0 <= 0 && e1 <= 0 && e2 <= 0);
            if (in) img[(size_t)y * stride + x] = g;
        }
}

}  // namespace

extern "C" int orbx_synth_frame(uint64_t seed, int width, int height, int stride, int flags, uint8_t *dst)
{
    return orbx_synth_frame_ex(seed, 0, 0, 0, width, height, stride, flags, dst);
}

// Same scene `seed` seen `view` steps later: every shape is translated by (dx, dy) pixels and
// the noise stream is re-drawn per view, so consecutive views of one scene share most of
// their corners (a camera translating in front of a fronto-parallel scene).
extern "C" int orbx_synth_frame_ex(uint64_t seed, int view, int dx, int dy, int width, int height, int stride, int flags, uint8_t *dst)
{
    if (!dst || width <= 0 || height <= 0 || stride < width) return -1;
    const bool low = (flags & ORBX_SYNTH_LOW_TEXTURE) != 0;
    const bool right = (flags & ORBX_SYNTH_STEREO_RIGHT) != 0;
    Rng shapes(0x9E3779B97F4A7C15ULL ^ seed);
    Rng noise((0xD1B54A32D192ED03ULL ^ (seed * 0x9E3779B97F4A7C15ULL)) + (right ? 0x5851F42D4C957F2DULL : 0) +
              (uint64_t)view * 0xA24BAED4963EE407ULL);
    for (int y = 0; y < height; y++) memset(dst + (size_t)y * stride, 128, (size_t)width);
    const int nrect = low ? 20 : 420, ntri = low ? 7 : 140;
    const int total = nrect + ntri;
    // interleave rectangles and triangles so neither kind is systematically on top
    int ri = 0, ti = 0;
    for (int s = 0; s < total; s++) {
        bool is_tri = (ti < ntri) && ((ri >= nrect) || (shapes.below((uint32_t)total) < (uint32_t)ntri));
        int disparity = 2 + (int)shapes.below(59);   // per-shape disparity in [2,60]
        int shift = right ? -disparity : 0;
        uint8_t g = (uint8_t)shapes.below(256);
        if (!is_tri) {
            int w = 4 + (int)shapes.below(90), h = 4 + (int)shapes.below(90);
            int x0 = (int)shapes.below((uint32_t)width) - w / 2, y0 = (int)shapes.below((uint32_t)height) - h / 2;
            fill_rect(dst, width, height, stride, x0 + shift + dx, y0 + dy, w, h, g);
            ri++;
        } else {
            int cx = (int)shapes.below((uint32_t)width), cy = (int)shapes.below((uint32_t)height);
            int vx[3], vy[3];
            for (int i = 0; i < 3; i++) {
                vx[i] = cx + (int)shapes.below(101) - 50 + shift + dx;
                vy[i] = cy + (int)shapes.below(101) - 50 + dy;
            }
            fill_tri(dst, width, height, stride, vx, vy, g);
            ti++;
        }
    }
    const int amp = low ? 2 : 6;
    for (int y = 0; y < height; y++) {
        uint8_t *row = dst + (size_t)y * stride;
        for (int x = 0; x < width; x++) {
            int v = row[x] + (int)noise.below((uint32_t)(2 * amp + 1)) - amp;
            row[x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    return 0;
}
