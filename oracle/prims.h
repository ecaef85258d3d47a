/* oracle/prims.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Plain C restatement of the five OpenCV primitives the reference extractor
 * calls.  OpenCV is NOT vendored under /root/reference and its version is not
 * pinned by the reference (CMakeLists.txt:31-40), so what is restated here is
 * the published OpenCV 4.x generic (non-IPP, non-OpenCL) C++ behaviour of:
 *
 *   cv::resize INTER_LINEAR, CV_8UC1      call site src/ORBextractor.cc:1696
 *   cv::FAST  TYPE_9_16 + 3x3 NMS         call sites src/ORBextractor.cc:1126,1135
 *   cv::GaussianBlur 7x7 s=2 REFLECT_101  call site src/ORBextractor.cc:1629
 *   cv::fastAtan2                         call site src/ORBextractor.cc:160
 *   cvRound / cvFloor / cvCeil            throughout
 *
 * PARITY UNPINNED: the reference holds no test or golden vector for any of
 * these; the semantics below are the project's pinned
 * choice and everything downstream (cvshim build of the
 * unmodified reference, the independent restatement, the HIP kernels) is
 * checked against THIS file.
 *
 * Build with -ffp-contract=off: every float product/sum below is rounded
 * separately, exactly as the un-fused scalar OpenCV code does.
 */
#ifndef ORB_ORACLE_PRIMS_H
#define ORB_ORACLE_PRIMS_H

#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- rounding helpers: cvRound = round-half-to-even (lrint in the default
 * rounding mode, as OpenCV's SSE2 cvtsd2si path), cvFloor / cvCeil. ---- */
static inline int op_round_d(double v) { return (int)lrint(v); }
static inline int op_round_f(float v) { return (int)lrintf(v); }
static inline int op_floor_d(double v) { int i = (int)v; return i - (i > v); }
static inline int op_ceil_d(double v) { int i = (int)v; return i + (i < v); }
static inline short op_sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

/* ---- cv::fastAtan2 (degrees, [0,360)), scalar path of OpenCV >= 3.x
 * (modules/core/src/mathfuncs_core.simd.hpp atan_f32).  All float. ---- */
static inline float op_fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ---- cv::resize(src,dst,dsize,0,0,INTER_LINEAR) for CV_8UC1.
 * resizeGeneric_<HResizeLinear<uchar,int,short,2048>, VResizeLinear<uchar,int,short,
 * FixedPtCast<int,uchar,22>>>: 11-bit fixed-point coefficients, the classic
 * ((b0*(H0>>4))>>16 + (b1*(H1>>4))>>16 + 2)>>2 vertical blend. ---- */
static inline void op_resize_linear_u8(const uint8_t *src, int sw, int sh, size_t sstep,
                                       uint8_t *dst, int dw, int dh, size_t dstep)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * (size_t)dw);
    int *rows0 = (int *)malloc(sizeof(int) * (size_t)dw);
    int *rows1 = (int *)malloc(sizeof(int) * (size_t)dw);
    int dx, dy;
    for (dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = op_floor_d(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = op_sat_short(op_round_f((1.f - fx) * 2048));
        ialpha[2 * dx + 1] = op_sat_short(op_round_f(fx * 2048));
    }
    for (dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = op_floor_d(fy);
        fy -= sy;
        short b0 = op_sat_short(op_round_f((1.f - fy) * 2048));
        short b1 = op_sat_short(op_round_f(fy * 2048));
        int y0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
        const uint8_t *S0 = src + (size_t)y0 * sstep, *S1 = src + (size_t)y1 * sstep;
        uint8_t *D = dst + (size_t)dy * dstep;
        for (dx = 0; dx < dw; dx++) {
            int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sx;
            int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
            rows0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            rows1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        for (dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (rows0[dx] >> 4)) >> 16) + ((b1 * (rows1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(rows0); free(rows1);
}

/* ---- cv::FAST TYPE_9_16.  Circle offsets in OpenCV's order (fast.cpp makeOffsets). ---- */
static const int8_t OP_FAST_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int8_t OP_FAST_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* cornerScore<16>(ptr, pixel, threshold) of OpenCV fast_score.cpp, restated with
 * its early-outs.  For a pixel that IS a corner at `threshold` the result is the
 * largest t at which it is still a corner (independent of `threshold`). */
static inline int op_fast_corner_score(const uint8_t *p, size_t step, int threshold)
{
    int d[25], k, v = p[0];
    for (k = 0; k < 25; k++) {
        int kk = k & 15;
        d[k] = v - p[(ptrdiff_t)OP_FAST_DY[kk] * (ptrdiff_t)step + OP_FAST_DX[kk]];
    }
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        a = a < d[k + 4] ? a : d[k + 4];
        a = a < d[k + 5] ? a : d[k + 5];
        a = a < d[k + 6] ? a : d[k + 6];
        a = a < d[k + 7] ? a : d[k + 7];
        a = a < d[k + 8] ? a : d[k + 8];
        { int m = a < d[k] ? a : d[k]; if (m > a0) a0 = m; }
        { int m = a < d[k + 9] ? a : d[k + 9]; if (m > a0) a0 = m; }
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        b = b > d[k + 3] ? b : d[k + 3];
        b = b > d[k + 4] ? b : d[k + 4];
        b = b > d[k + 5] ? b : d[k + 5];
        if (b >= b0) continue;
        b = b > d[k + 6] ? b : d[k + 6];
        b = b > d[k + 7] ? b : d[k + 7];
        b = b > d[k + 8] ? b : d[k + 8];
        { int m = b > d[k] ? b : d[k]; if (m < b0) b0 = m; }
        { int m = b > d[k + 9] ? b : d[k + 9]; if (m < b0) b0 = m; }
    }
    return -b0 - 1;
}

/* Segment test: >= 9 contiguous circle pixels all < v-t or all > v+t (strict). */
static inline int op_fast_is_corner(const uint8_t *p, size_t step, int t)
{
    int v = p[0], k, cd = 0, cb = 0;
    for (k = 0; k < 25; k++) {
        int kk = k & 15;
        int x = p[(ptrdiff_t)OP_FAST_DY[kk] * (ptrdiff_t)step + OP_FAST_DX[kk]];
        if (x < v - t) { if (++cd > 8) return 1; } else cd = 0;
        if (x > v + t) { if (++cb > 8) return 1; } else cb = 0;
    }
    return 0;
}

typedef struct { int x, y, score; } op_fast_kp;

/* FAST_t<16>(img, keypoints, threshold, nonmax=true): detect rows 3..rows-4 x cols
 * 3..cols-4 of the image passed in, score every corner, keep strict 3x3 local maxima
 * (anything that is not a detected corner of THIS image counts as 0), raster order.
 * Returns the number of keypoints written (<= cap). */
static inline int op_fast9_nms(const uint8_t *img, int rows, int cols, size_t step, int threshold,
                               op_fast_kp *out, int cap)
{
    int n = 0, i, j;
    if (rows < 7 || cols < 7) return 0;
    threshold = threshold < 0 ? 0 : (threshold > 255 ? 255 : threshold);
    uint8_t *sc = (uint8_t *)calloc((size_t)rows * cols, 1);
    for (i = 3; i < rows - 3; i++)
        for (j = 3; j < cols - 3; j++) {
            const uint8_t *p = img + (size_t)i * step + j;
            if (op_fast_is_corner(p, step, threshold))
                sc[(size_t)i * cols + j] = (uint8_t)op_fast_corner_score(p, step, threshold);
        }
    for (i = 3; i < rows - 3; i++)
        for (j = 3; j < cols - 3; j++) {
            const uint8_t *s = sc + (size_t)i * cols + j;
            int v = s[0];
            /* a detected corner always enters OpenCV's cornerpos list, even with score 0
             * (threshold 0); such a point can never be a strict maximum, so v>0 is exact. */
            if (v == 0) continue;
            if (v > s[-1] && v > s[1] && v > s[-cols - 1] && v > s[-cols] && v > s[-cols + 1] &&
                v > s[cols - 1] && v > s[cols] && v > s[cols + 1]) {
                if (n < cap) { out[n].x = j; out[n].y = i; out[n].score = v; }
                n++;
            }
        }
    free(sc);
    return n;
}

/* ---- cv::GaussianBlur(src,dst,Size(7,7),2,2,BORDER_REFLECT_101), CV_8UC1, OpenCV 4.x
 * bit-exact fixed-point path (smooth.simd.hpp, ufixedpoint16 kernel from
 * getGaussianKernelFixedPoint_ED): taps 18,34,48,56,48,34,18 (/256, sum exactly 256),
 * horizontal pass exact in u16 (8.8), vertical in u32, out = (sum + 2^15) >> 16.
 * `taps` may be overridden (e.g. 18,34,49,55,49,34,18 of OpenCV 3.4.1-4.5.0); the
 * result is then saturated to 255 like ufixedpoint32 -> uint8_t. ---- */
static const uint16_t OP_GAUSS7_TAPS[7] = {18, 34, 48, 56, 48, 34, 18};

static inline int op_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

static inline void op_gauss7_u8(const uint8_t *src, int w, int h, size_t sstep,
                                uint8_t *dst, size_t dstep, const uint16_t *taps)
{
    if (!taps) taps = OP_GAUSS7_TAPS;
    uint32_t *hb = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)w * h);
    int x, y, k;
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) {
            uint32_t s = 0;
            for (k = -3; k <= 3; k++) s += (uint32_t)taps[k + 3] * src[(size_t)y * sstep + op_reflect101(x + k, w)];
            hb[(size_t)y * w + x] = s > 65535u ? 65535u : s;
        }
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) {
            uint32_t s = 0;
            for (k = -3; k <= 3; k++) s += (uint32_t)taps[k + 3] * hb[(size_t)op_reflect101(y + k, h) * w + x];
            s = (s + 32768u) >> 16;
            dst[(size_t)y * dstep + x] = (uint8_t)(s > 255u ? 255u : s);
        }
    free(hb);
}

/* ---- cv::undistortPoints(src, dst, K, distCoeffs, R = noArray(), P) for CV_32FC2 points
 * (calib3d/imgproc undistort.cpp, cvUndistortPoints of OpenCV 2.4.x .. 3.3: FIVE fixed
 * iterations of the inverse distortion model, everything in double, result stored as
 * float; later versions add a convergence criterion -- un-vendored and unpinned like the
 * rest of this file).  K: 3x3 camera matrix, k[8] = k1 k2 p1 p2 k3 k4 k5 k6 (zeros when
 * absent), RR = P * R (3x3).  iters = 5 when distortion coefficients are given, else 0. */
static inline void op_undistort_points(const float *src, float *dst, int n, const double *K, const double *k,
                                       const double *RR, int iters)
{
    const double fx = K[0], fy = K[4], ifx = 1. / fx, ify = 1. / fy, cx = K[2], cy = K[5];
    for (int i = 0; i < n; i++) {
        double x = src[2 * i], y = src[2 * i + 1], x0, y0;
        x0 = x = (x - cx) * ifx;
        y0 = y = (y - cy) * ify;
        for (int j = 0; j < iters; j++) {
            double r2 = x * x + y * y;
            double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        double xx = RR[0] * x + RR[1] * y + RR[2];
        double yy = RR[3] * x + RR[4] * y + RR[5];
        double ww = 1. / (RR[6] * x + RR[7] * y + RR[8]);
        dst[2 * i] = (float)(xx * ww);
        dst[2 * i + 1] = (float)(yy * ww);
    }
}

/* ---- glibc 2.35 sinf/cosf (sysdeps/ieee754/flt-32/s_sincosf.h, s_sinf.c, s_cosf.c;
 * the Arm optimized-routines algorithm) restated for |x| < 120: double polynomial,
 * one multiply-subtract range reduction, result rounded to float.  The reference
 * calls libm through `cos(angle)`/`sin(angle)` on a float (src/ORBextractor.cc:181);
 * tests/test_sincos.py checks this restatement against the libm of the box over
 * every float in [0, 6.5] so that the device copy of the same arithmetic is pinned
 * to what the compiled reference computes. ---- */
typedef struct { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; } op_sincos_t;
static const op_sincos_t OP_SINCOS_TAB[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
     0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16,
     -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
     -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16,
     -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};

static inline uint32_t op_abstop12(float x) { uint32_t u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ff; }

static inline float op_sinf_poly(double x, double x2, const op_sincos_t *p, int n)
{
    if ((n & 1) == 0) {
        double x3 = x * x2, s1 = p->s2 + x2 * p->s3, x7 = x3 * x2, s = x + x3 * p->s1;
        return (float)(s + x7 * s1);
    } else {
        double x4 = x2 * x2, c2 = p->c3 + x2 * p->c4, c1 = p->c0 + x2 * p->c1, x6 = x4 * x2, c = c1 + x4 * p->c2;
        return (float)(c + x6 * c2);
    }
}

static inline void op_sincosf(float y, float *sn, float *cs)
{
    double x = y;
    const op_sincos_t *p = &OP_SINCOS_TAB[0];
    if (op_abstop12(y) < op_abstop12(0x1.921FB6p-1f)) {
        double x2 = x * x;
        if (op_abstop12(y) < op_abstop12(0x1p-12f)) { *sn = y; *cs = 1.0f; return; }
        *sn = op_sinf_poly(x, x2, p, 0);
        *cs = op_sinf_poly(x, x2, p, 1);
        return;
    }
    double r = x * p->hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    x = x - n * p->hpi;
    double s = p->sign[n & 3];
    if (n & 2) p = &OP_SINCOS_TAB[1];
    *sn = op_sinf_poly(x * s, x * x, p, n);
    *cs = op_sinf_poly(x * s, x * x, p, n ^ 1);
}

#ifdef __cplusplus
}
#endif
#endif /* ORB_ORACLE_PRIMS_H */
