// orbx_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ORB extractor.
//
// The stages of ORB_SLAM2::ORBextractor::operator() (reference src/ORBextractor.cc:1544-1668) as
// k_resize (x7), k_fast_cells, k_octree, k_blur, k_orient_describe; every launch covers the whole
// batch (blockIdx.y = frame) and, where the stage has no level-to-level dependency, all pyramid
// levels at once (blockIdx.x -> (level, tile) through OrbxGeom).  All launches go through emit():
// onto a stream, or as nodes of the single-frame hipGraph.
// Integer / byte work bounded by HBM and VALU issue; no MFMA.
// Compiled with -ffp-contract=off: the few float expressions must round exactly like
// the un-fused scalar reference.
#include <algorithm>
#include "orbx_internal.h"

namespace {

// rBRIEF test pairs (reference src/ORBextractor.cc:231-489) and the disc of IC_Angle (umax, :579-608 for HALF_PATCH_SIZE 15) as the
// tables k_orient_describe copies into LDS: built at compile time from the generated pattern / the half-widths.
struct OdTables {
    float pat[256][4];          // test pair t: x0, x1, y0, y1 (the rotation is packed FP32 arithmetic on both points of a pair)
    uint32_t disc[64][4];       // lane = (disc row r = lane / 2, half): byte mask of its 16-byte row segment, |u| <= umax[|r - 15|]
};
constexpr int8_t kPattern[1024] = {
#include "orb_pattern.inc"
};
constexpr int kUmax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};      // checked against the host table at handle creation
constexpr OdTables make_od_tables()
{
    OdTables t{};
    for (int i = 0; i < 256; i++) {
        t.pat[i][0] = (float)kPattern[4 * i]; t.pat[i][2] = (float)kPattern[4 * i + 1];
        t.pat[i][1] = (float)kPattern[4 * i + 2]; t.pat[i][3] = (float)kPattern[4 * i + 3];
    }
    for (int ml = 0; ml < 64; ml++)
        for (int j = 0; j < 4; j++) {
            const int mr = ml >> 1, mh = ml & 1;
            uint32_t m = 0u;
            if (mr < 31) {
                const int v = mr - 15, d = kUmax[v < 0 ? -v : v];
                const int lo = mh ? 0 : 15 - d, hi = mh ? d - 1 : 15;      // valid bytes c of the segment: u = c - 15 (left half) / c + 1 (right half)
                for (int k = 0; k < 4; k++)
                    if (4 * j + k >= lo && 4 * j + k <= hi) m |= 0xffu << (8 * k);
            }
            t.disc[ml][j] = m;
        }
    return t;
}
__constant__ OdTables c_od = make_od_tables();

// FAST-16 circle, OpenCV order (reference call sites src/ORBextractor.cc:1126,1135)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "orbx_kernels.hip uses gfx9 DPP row_bcast controls and v_min3_u32 / v_max3_u32 inline assembly: build for gfx950 (--offload-arch=gfx950)"
#endif
#define FAST_DX(k) ((k) == 0 ? 0 : (k) == 1 ? 1 : (k) == 2 ? 2 : (k) <= 5 ? 3 : (k) == 6 ? 2 : (k) == 7 ? 1 : (k) == 8 ? 0 : (k) == 9 ? -1 : (k) == 10 ? -2 : (k) <= 13 ? -3 : (k) == 14 ? -2 : -1)
#define FAST_DY(k) ((k) <= 1 ? 3 : (k) == 2 ? 2 : (k) == 3 ? 1 : (k) == 4 ? 0 : (k) == 5 ? -1 : (k) == 6 ? -2 : (k) <= 9 ? -3 : (k) == 10 ? -2 : (k) == 11 ? -1 : (k) == 12 ? 0 : (k) == 13 ? 1 : (k) == 14 ? 2 : 3)

__device__ __forceinline__ const uint8_t *level_ptr(const OrbxGeom *g, int l, int f, const uint8_t *img0, int img0Stride,
                                                    size_t img0FramePitch, const uint8_t *pyr, int &pitch)
{
    if (l == 0) { pitch = img0Stride; return img0 + (size_t)f * img0FramePitch; }
    pitch = g->lv[l].pitch;
    return pyr + (size_t)f * g->pyrBytes + g->lv[l].off;
}

__device__ __forceinline__ int find_level(const int *base, int nlevels, int idx)
{
    int l = 0;
    for (int i = 1; i < nlevels; i++) if (idx >= base[i]) l = i;
    return l;
}

__device__ __forceinline__ int wave_incl_scan_i(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}

// ------------------------------------------------------------------------------------
// Pyramid: level L = cv::resize(level L-1, INTER_LINEAR) (src/ORBextractor.cc:1696-1701),
// 11-bit fixed point; the column/row tables are built on the host with the exact
// float/double arithmetic of OpenCV's resize (orbx_extractor.hip: build_geometry).
// One thread -> 4 consecutive dst pixels x 4 rows; per source row ONE unaligned 8-byte load
// covers all eight taps of the four pixels (scale 1.2: span <= 6 bytes), one dword store per row.
// ------------------------------------------------------------------------------------
#define RS_ROWS 8   /* dst rows per thread */
#define ORBX_OCTREE_COPY_BLOCKS 48
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c)
{
    union { uint32_t u; u16x2 v; } x, y;
    x.u = a; y.u = b;
    return __builtin_amdgcn_udot2(x.v, y.v, c, false);   // v_dot2_u32_u16: a.lo*b.lo + a.hi*b.hi + c
}

// cv::resize(INTER_LINEAR, 8-bit) from level l-1.  OpenCV's index / coefficient arithmetic (double
// product, float fraction, cvRound to 11 bits, border rules) is evaluated once per level on the host
// (orbx_extractor.hip: build_resize_tables); a thread fetches its column group's 12 words and one
// word pair per row instead of ~25 VALU instructions each.
// Horizontal taps of one pixel = ONE v_perm_b32 (two bytes of the 8-byte source window -> a u16
// pair, selector from the table) + ONE v_dot2_u32_u16 against (a0, a1); the vertical blend keeps
// OpenCV's two truncating >>16 terms.  Thread = 4 dst columns x 8 dst rows, 16 independent 8-byte
// loads in flight.
// ---------------------------------------------------------------------------------------------
// XCD-aware block mapping.  The dispatcher places linear block b on XCD b % 8 (observed, MI355X guide; a pure speed assumption,
// never a correctness one), and every XCD has its own 4 MiB L2.  With the natural (item, frame) order, horizontally adjacent
// tiles / cells of an image - which share the 128-byte lines that their 80-byte window rows straddle, and their halos - run on
// eight different XCDs, each of which fetches the shared lines from the fabric again (rocprofv3 FETCH_SIZE: 4.4x the level
// pixels for k_blur, 4.7x for k_fast_cells, 5.2x for the descriptor kernel's gathers).  This bijection hands every XCD one CONTIGUOUS range
// of the (frame, item) space instead, so neighbours meet in one L2 and a frame's images are fetched by one XCD.
//   linear = by * gx + bx;  XCD x owns [x*q + min(x, r), ...) with q = total / 8, r = total % 8;  slot = linear / 8.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void xcd_remap(int gx, int total, int linear, int &bx, int &by)
{
    const int xcd = linear & 7, slot = linear >> 3, q = total >> 3, r = total & 7;
    const int t = xcd * q + min(xcd, r) + slot;
    by = t / gx;
    bx = t - by * gx;
}
#define XCD_REMAP_XY(BX, BY) int BX, BY; xcd_remap((int)gridDim.x, (int)(gridDim.x * gridDim.y), (int)(blockIdx.y * gridDim.x + blockIdx.x), BX, BY)

template <bool PADDED>   // source rows readable 11 bytes past the last pixel: every thread takes the aligned path
__global__ __launch_bounds__(256) void k_resize(const OrbxGeom *__restrict__ g, int level, const uint8_t *__restrict__ img0, int img0Stride,
                                                size_t img0FramePitch, uint8_t *__restrict__ pyr, const uint32_t *__restrict__ rsTab)
{
    const OrbxLevel &lv = g->lv[level];
    int bx, f;
    xcd_remap((int)gridDim.x, (int)(gridDim.x * gridDim.z), (int)(blockIdx.z * gridDim.x + blockIdx.x), bx, f);
    // flat item = (row group, column group of 4 px): no lane idles on a ragged right edge
    const int G = (lv.w + 3) >> 2, RG = (lv.h + RS_ROWS - 1) / RS_ROWS;
    const int item = bx * 256 + threadIdx.x;
    if (item >= G * RG) return;
    const int rg = item / G, cg = item - rg * G;
    const int dx0 = cg * 4, dyBase = rg * RS_ROWS;
    int sp;
    const uint8_t *src = level_ptr(g, level - 1, f, img0, img0Stride, img0FramePitch, pyr, sp);
    const int sw = g->lv[level - 1].w;
    const uint32_t *ct = rsTab + lv.rsColOff + 12 * cg;
    const uint4 c0 = *(const uint4 *)ct, c1 = *(const uint4 *)(ct + 4), c2 = *(const uint4 *)(ct + 8);
    const int sx0 = (int)c0.x, span = (int)c0.y;
    const uint32_t sel[4] = {c0.z, c0.w, c1.x, c1.y}, coef[4] = {c1.z, c1.w, c2.x, c2.y};   // 0 <= a0, a1 <= 2048
    const uint2 *rt = (const uint2 *)(rsTab + lv.rsRowOff);
    // all taps inside 8 bytes from sx0, and the 12 aligned bytes that contain them inside the row pitch
    const int sxa = sx0 & ~3;
    const uint32_t mis = (uint32_t)(sx0 & 3);
    const bool wide = span <= 7 && (PADDED || sxa + 12 <= sw);
    uint8_t *dstBase = pyr + (size_t)f * g->pyrBytes + lv.off;
    if (wide) {
        uint32_t v0[RS_ROWS][2], v1[RS_ROWS][2], bb[RS_ROWS];
#pragma unroll
        for (int r = 0; r < RS_ROWS; r++) {
            const uint2 t = rt[min(dyBase + r, lv.h - 1)];       // rows: no coefficient reset at the far edge, only index clamping
            bb[r] = t.y;
            // 8 source bytes from byte sx0: three ALIGNED dwords + two v_alignbyte_b32
            const uint32_t *p0 = (const uint32_t *)(src + (size_t)(t.x & 0xffffu) * sp + sxa), *p1 = (const uint32_t *)(src + (size_t)(t.x >> 16) * sp + sxa);
            const uint32_t a0_ = p0[0], a1_ = p0[1], a2_ = p0[2], c0_ = p1[0], c1_ = p1[1], c2_ = p1[2];
            v0[r][0] = __builtin_amdgcn_alignbyte(a1_, a0_, mis); v0[r][1] = __builtin_amdgcn_alignbyte(a2_, a1_, mis);
            v1[r][0] = __builtin_amdgcn_alignbyte(c1_, c0_, mis); v1[r][1] = __builtin_amdgcn_alignbyte(c2_, c1_, mis);
        }
#pragma unroll
        for (int r = 0; r < RS_ROWS; r++) {
            const int dy = dyBase + r;
            if (dy >= lv.h) break;
            const uint32_t b0 = bb[r] & 0xffffu, b1 = bb[r] >> 16;
            uint32_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t r0 = udot2(__builtin_amdgcn_perm(v0[r][1], v0[r][0], sel[k]), coef[k], 0u);
                const uint32_t r1 = udot2(__builtin_amdgcn_perm(v1[r][1], v1[r][0], sel[k]), coef[k], 0u);
                const uint32_t v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2u) >> 2;
                out |= v << (8 * k);      // v <= 255: a0 + a1 and b0 + b1 are 2048 (+-1 from cvRound), so the blend of two bytes cannot reach 256
            }
            *(uint32_t *)(dstBase + (size_t)dy * lv.pitch + dx0) = out;   // pitch is a multiple of 64 >= round_up(w,4)
        }
    } else {
        for (int r = 0; r < RS_ROWS; r++) {
            const int dy = dyBase + r;
            if (dy >= lv.h) break;
            const uint2 t = rt[dy];
            const uint32_t b0 = t.y & 0xffffu, b1 = t.y >> 16;
            const uint8_t *S0 = src + (size_t)(t.x & 0xffffu) * sp + sx0, *S1 = src + (size_t)(t.x >> 16) * sp + sx0;
            uint32_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int off = (int)((c2.z >> (8 * k)) & 0xffu), off1 = (int)((c2.w >> (8 * k)) & 0xffu);
                const uint32_t r0 = udot2((uint32_t)S0[off] | ((uint32_t)S0[off1] << 16), coef[k], 0u);
                const uint32_t r1 = udot2((uint32_t)S1[off] | ((uint32_t)S1[off1] << 16), coef[k], 0u);
                const uint32_t v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2u) >> 2;
                out |= v << (8 * k);      // v <= 255: a0 + a1 and b0 + b1 are 2048 (+-1 from cvRound), so the blend of two bytes cannot reach 256
            }
            *(uint32_t *)(dstBase + (size_t)dy * lv.pitch + dx0) = out;
        }
    }
}

// ------------------------------------------------------------------------------------
// The whole pyramid of ONE frame in one launch (single-frame call).  Seven dependent k_resize launches of a few
// workgroups each are ~4.3 us apiece on an otherwise idle device - launch, one memory round trip, store - whatever the
// level's size.  Here a workgroup takes one tile of the image through ALL levels inside LDS: it stages the window of
// level 0 that its share of the deepest level depends on, and computes level after level the rectangle the next level
// reads (host-planned per tile and level from the same resize tables, OrbxPyrTile; neighbouring workgroups recompute
// each other's halos - the device is idle, the redundancy is free), storing the part it owns.  Same tables, same
// arithmetic as k_resize's 8-byte-window path, source = LDS.
// ------------------------------------------------------------------------------------
#define PT_PITCH(cw) (((cw) + 12 + 7) & ~7)      /* LDS row pitch of a computed rectangle: 12 readable bytes behind every group window */
#define PT_WIN_ITEMS 8          /* 8-byte units of the level-0 window per thread (all requested before the first is stored); the plan keeps windows below 16 KB */
__global__ __launch_bounds__(256) void k_pyramid_tiles(const OrbxGeom *__restrict__ g, const uint8_t *__restrict__ img0, int img0Stride, size_t img0FramePitch,
                                                       uint8_t *__restrict__ pyr, const uint32_t *__restrict__ rsTab, const OrbxPyrTile *__restrict__ tiles, int bufBytes,
                                                       int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];      // two image buffers of bufBytes (level parity), then the table slices of all levels
    const int nl = g->nlevels, tid = threadIdx.x;
    const OrbxPyrTile *tt = tiles + (size_t)blockIdx.x * nl;
    uint8_t *tab = lds + 2 * bufBytes;
    // blockIdx.y = frame of a (small) batch: the single-frame call, or the frames of concurrent callers combined into one launch set
    img0 += (size_t)blockIdx.y * img0FramePitch;
    pyr += (size_t)blockIdx.y * g->pyrBytes;
    // (measured and not kept: a second store of every level straight into the caller's pinned pyramid copy - 4-byte stores in ~44-byte runs
    // cross PCIe badly: the call took 126 us instead of 115 with a 16-byte-per-lane copy at the end of the chain; the copy now rides in the
    // quadtree's launch, k_octree)
    // the frames' capacity words and the batch word (first kernel of the chain: no memset node)
    if (blockIdx.x == 0 && threadIdx.x == 0) { status[blockIdx.y] = 0; if (blockIdx.y == 0) status[gridDim.y] = 0; }
    // rectangles and level parameters of the level loop out of LDS too: a scalar load from global memory at the top of every level is a
    // dependent ~0.7 us each, seven times
    __shared__ OrbxPyrTile sT[ORBX_MAX_LEVELS];
    __shared__ int sLv[ORBX_MAX_LEVELS][2];
    if (tid < nl) { sT[tid] = tt[tid]; sLv[tid][0] = g->lv[tid].off; sLv[tid][1] = g->lv[tid].pitch; }
    // ---- ONE memory round trip for everything the tile needs: the table slices of all levels (column groups of the computed rectangle: 48 bytes
    // each; its rows: 8 bytes each; at most one item per thread and level, requested for all levels before the first one is stored - a loop over
    // the levels would wait for each level's data before asking for the next) and the window of level 0
    {
        uint4 cv[ORBX_MAX_LEVELS];
        int dstOff[ORBX_MAX_LEVELS];
        bool colItem[ORBX_MAX_LEVELS];
        int run = 0;
#pragma unroll
        for (int L = 1; L < ORBX_MAX_LEVELS; L++) {
            const int Lc = min(L, nl - 1);       // (levels past the last: the last one again, not stored - every load unconditional)
            const OrbxPyrTile t = tt[Lc];
            const OrbxLevel &lv = g->lv[Lc];
            const int ng = (t.cx1 - t.cx0) >> 2, ch = t.cy1 - t.cy0, nc = 3 * ng;
            const bool isCol = tid < nc;
            const int i = isCol ? tid : min(tid - nc, max(ch - 1, 0));
            const uint32_t *srcw = isCol ? rsTab + lv.rsColOff + 12 * (t.cx0 >> 2) + 4 * i : rsTab + lv.rsRowOff + 2 * (t.cy0 + i);
            const uint2 lo = *(const uint2 *)srcw, hi = *(const uint2 *)(isCol ? srcw + 2 : srcw);      // (row items are 8 bytes)
            cv[L] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            colItem[L] = isCol;
            dstOff[L] = (L < nl && tid < nc + ch) ? (isCol ? run + 16 * i : run + 16 * nc + 8 * i) : -1;
            if (L < nl) run += 16 * nc + ((8 * ch + 15) & ~15);
        }
        const OrbxPyrTile t0 = tt[0];
        const int ch0 = t0.cy1 - t0.cy0, P0 = PT_PITCH(t0.cx1 - t0.cx0), nu = P0 >> 3, nitem = ch0 * nu;      // (planned: nitem <= 256 * PT_WIN_ITEMS)
        uint2 wv[PT_WIN_ITEMS];
#pragma unroll
        for (int k = 0; k < PT_WIN_ITEMS; k++) {      // (the staging rows are readable past the window: pitch >= W + 16, 256 spare bytes behind the frame)
            const int idx = min(tid + 256 * k, nitem - 1), r = idx / nu, c = idx - r * nu;
            __builtin_memcpy(&wv[k], img0 + (size_t)(t0.cy0 + r) * img0Stride + t0.cx0 + 8 * c, 8);
        }
#pragma unroll
        for (int k = 0; k < PT_WIN_ITEMS; k++) {
            const int idx = tid + 256 * k, r = idx / nu, c = idx - r * nu;
            if (idx < nitem) *(uint2 *)(lds + r * P0 + 8 * c) = wv[k];
        }
#pragma unroll
        for (int L = 1; L < ORBX_MAX_LEVELS; L++)
            if (dstOff[L] >= 0) {
                if (colItem[L]) *(uint4 *)(tab + dstOff[L]) = cv[L];
                else *(uint2 *)(tab + dstOff[L]) = make_uint2(cv[L].x, cv[L].y);
            }
    }
    __syncthreads();
    int run = 0;
    for (int L = 1; L < nl; L++) {
        const OrbxPyrTile s = sT[L - 1], t = sT[L];
        const int lvOff = sLv[L][0], lvPitch = sLv[L][1];
        const int sP = PT_PITCH(s.cx1 - s.cx0), dP = PT_PITCH(t.cx1 - t.cx0);
        const uint8_t *src = lds + ((L - 1) & 1) * bufBytes;
        uint8_t *dst = lds + (L & 1) * bufBytes;
        const int ng = (t.cx1 - t.cx0) >> 2, ch = t.cy1 - t.cy0;
        const uint4 *ctab = (const uint4 *)(tab + run);
        const uint2 *rtab = (const uint2 *)(tab + run + 48 * ng);
        run += 48 * ng + ((8 * ch + 15) & ~15);
        if (ng > 0 && ch > 0) {
            // thread = (column group, row phase): the group's twelve table words once per level, one row-table pair per row, both out of LDS
            // tid / ng and 256 / ng without the integer-division sequences (three of them were a third of a small level's time with one wave
            // per SIMD): float quotient, corrected by one step either way (exact for operands below 2^12)
            const float rng = __builtin_amdgcn_rcpf((float)ng);
            int rph = (int)((float)tid * rng), rpp = (int)(256.f * rng);
            rph += (rph + 1) * ng <= tid ? 1 : 0; rph -= rph * ng > tid ? 1 : 0;
            rpp += (rpp + 1) * ng <= 256 ? 1 : 0; rpp -= rpp * ng > 256 ? 1 : 0;
            const int gi = tid - rph * ng;
            if (rph < rpp) {
                const int X = t.cx0 + 4 * gi;
                const uint4 c0 = ctab[3 * gi], c1 = ctab[3 * gi + 1], c2 = ctab[3 * gi + 2];
                const int sx0 = (int)c0.x, xoff = (sx0 & ~3) - s.cx0;      // (s.cx0 is a multiple of 4: the window stays dword aligned)
                const uint32_t mis = (uint32_t)(sx0 & 3);
                const uint32_t sel[4] = {c0.z, c0.w, c1.x, c1.y}, coef[4] = {c1.z, c1.w, c2.x, c2.y};
                const bool ownX = X >= t.ox0 && X < t.ox1;
                uint8_t *gdst = pyr + lvOff;
                for (int r = rph; r < ch; r += rpp) {
                    const int Y = t.cy0 + r;
                    const uint2 tr = rtab[r];
                    const uint32_t b0 = tr.y & 0xffffu, b1 = tr.y >> 16;
                    const uint32_t *p0 = (const uint32_t *)(src + ((int)(tr.x & 0xffffu) - s.cy0) * sP + xoff), *p1 = (const uint32_t *)(src + ((int)(tr.x >> 16) - s.cy0) * sP + xoff);
                    const uint32_t a0_ = p0[0], a1_ = p0[1], a2_ = p0[2], d0_ = p1[0], d1_ = p1[1], d2_ = p1[2];
                    const uint32_t v00 = __builtin_amdgcn_alignbyte(a1_, a0_, mis), v01 = __builtin_amdgcn_alignbyte(a2_, a1_, mis);
                    const uint32_t v10 = __builtin_amdgcn_alignbyte(d1_, d0_, mis), v11 = __builtin_amdgcn_alignbyte(d2_, d1_, mis);
                    uint32_t out = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t r0 = udot2(__builtin_amdgcn_perm(v01, v00, sel[k]), coef[k], 0u);
                        const uint32_t r1 = udot2(__builtin_amdgcn_perm(v11, v10, sel[k]), coef[k], 0u);
                        const uint32_t v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2u) >> 2;
                        out |= v << (8 * k);      // v <= 255: a0 + a1 and b0 + b1 are 2048 (+-1 from cvRound), so the blend of two bytes cannot reach 256
                    }
                    *(uint32_t *)(dst + r * dP + 4 * gi) = out;
                    if (ownX && Y >= t.oy0 && Y < t.oy1) *(uint32_t *)(gdst + (size_t)Y * lvPitch + X) = out;
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// Cell detector: FAST-9/16 score + per-cell 3x3 NMS + threshold fallback + raster-ordered emission
// (ComputeKeyPointsOctTree cell loop, src/ORBextractor.cc:1089-1157, with cv::FAST at :1126,1135).
// ONE WAVE PER 30-px CELL; the cell's input window (detectable area + 3 px ring) and its score
// tile live in LDS, scores never reach HBM.  `corner at t  <=>  score >= t`, so one score serves
// both the iniThFAST and the minThFAST pass of the reference.
//   A  pre-test, all pixels: lane = (row, 16-px segment), three 24-byte window rows in registers, two
//      horizontally adjacent centres per packed-u16 op.  A 9-arc of the 16-circle contains k or
//      k+8 for every k, so  v - max_k min(x_k, x_k+8) > t  (dark arc)  or
//      min_k max(x_k, x_k+8) - v > t  (bright arc)  is necessary - for any subset of the k.  Only the
//      compass pairs k = 0, 4 are tested: at iniThFAST they let 5.7 % of the pixels of a textured
//      frame through (all eight pairs: 3.2 %, true corners: 2.0 %), at 10 instead of 25 VALU
//      instructions per pixel, and phase B's exact score costs ~100 per SURVIVOR - 16 instead of 29
//      per pixel in total.  Survivors are appended to an LDS list - in raster order, because the lanes are.
//   B  full score of the survivors on dense waves: 9-arc min / max as min3(min3) chains.
//   C  strict 3x3 maximum inside the cell (zero ring = "not a corner of this sub-image"),
//      iniThFAST survivors or - when the cell has none - all of them (:1132), ordered __ballot
//      compaction into the cell's slot.
// VALU bound: ~570 wave-instructions per cell (119.0 M per launch of 208,640 cells: profiles/r05_fast_phases.txt, r05_valu_issue.txt; 1.05k in round 3).
// ------------------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    union { uint32_t u; u16x2 v; } x, y, z;
    x.u = a; y.u = b; z.v = __builtin_elementwise_min(x.v, y.v);
    return z.u;
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)
{
    union { uint32_t u; u16x2 v; } x, y, z;
    x.u = a; y.u = b; z.v = __builtin_elementwise_max(x.v, y.v);
    return z.u;
}
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b)
{
    union { uint32_t u; s16x2 v; } x, y, z;
    x.u = a; y.u = b; z.v = x.v - y.v;
    return z.u;
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b)
{
    union { uint32_t u; s16x2 v; } x, y, z;
    x.u = a; y.u = b; z.v = __builtin_elementwise_max(x.v, y.v);
    return z.u;
}
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(a, min(b, c)); }   // v_min3_i32
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(a, max(b, c)); }   // v_max3_i32
// The three-operand forms spelt out: left to itself the compiler shares min(x[k], x[k+1]) between neighbouring arcs and ends up with 62 two-operand + 40
// three-operand instructions for the 9-arc extrema of a survivor where 80 three-operand ones do (profiles/r05_fast_phases.txt: phase B is a fifth of the kernel).
__device__ __forceinline__ int min3u_asm(int a, int b, int c) { int r; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int max3u_asm(int a, int b, int c) { int r; asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// bytes (b, b+1) of the 24-byte window row W[0..5] as a u16 pair; b is a compile-time constant <= 20
#define FC_PAIR(W, b) __builtin_amdgcn_perm((W)[((b) >> 2) + 1 > 5 ? 5 : ((b) >> 2) + 1], (W)[(b) >> 2], \
                                            0x0c000c00u | (uint32_t)((b) & 3) | ((uint32_t)(((b) & 3) + 1) << 16))


// PROF (tools/fast_phases.py, never in a product launch): s_memtime stamps at the phase boundaries, per cell (= per pass of the wave's loop) into
// one 16-word record per cell behind a 32-word header: 0 prologue  1 staging + zeroing  2 A: windows + pre-test  3 A: scan + list append  4 B  5 C
//   6..9 = 2..5 of the minThFAST retry  10 emission  11 stamp cost  12 retried  13 survivors of the first pre-test  14 of the retry's  15 candidates emitted
#define FC_STAMP(i) do { if (PROF) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[i] += t_ - tPrev; tPrev = __builtin_amdgcn_s_memtime(); pacc[11] += tPrev - t_; } } while (0)

__device__ __forceinline__ uint4 load16u(const uint8_t *p)      // unaligned 16-byte load (global_load_dwordx4)
{
    typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
    const u32x4_u v = *(const u32x4_u *)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// inclusive prefix sum over the 64 lanes with data-parallel-primitive adds: four shifts inside the rows of 16, then the two row broadcasts of gfx9
// (six VALU instructions; __shfl_up is six dependent LDS-crossbar round trips)
__device__ __forceinline__ int wave_incl_scan_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return v;
}

// P = LDS row pitch of both tiles, an ODD multiple of 16 bytes (48 for cells up to 32 px wide, 80 up to 64): the rows of a window land on all 32 banks.
// NS = 16-byte units of a cell's window per lane (window rows x units per row <= 64 NS), K (an argument) = cells per wave.
// A wave takes K consecutive cells of a frame through the phases one after the other; everything it needs to know about a cell is ONE 32-byte table
// entry (OrbxFcCell, built with the geometry: the v1 prologue - a chain of ~12 dependent scalar loads through OrbxGeom and three integer divisions - was
// 28 % of a wave's lifetime, profiles/r05_fast_phases.txt), and the window of cell k+1 is requested before the phases of cell k run and stored into LDS
// after them (v1: three dependent load -> store round trips per cell, 18 % of the lifetime).
template <int P, int SP, int NS, bool PROF = false, bool DBG = false>      // DBG: the parity tap (score map of a single minThFAST pass); the product instantiation carries none of it
__global__ __launch_bounds__(64) void k_fast_cells(const OrbxFcCell *__restrict__ cells, int K, int cellsPerFrame, int slotsPerFrame, size_t pyrBytes, int iniTh, int minTh,
                                                   int inBytes, int scBytes, const uint8_t *__restrict__ img0, int img0Stride, size_t img0FramePitch,
                                                   const uint8_t *__restrict__ pyr, uint8_t *__restrict__ scoreDbg, int *__restrict__ cellCount,
                                                   uint32_t *__restrict__ cellSlots, unsigned long long *__restrict__ prof)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *inT = lds;                                   // (ah+6) x P: input window, tile (0,0) = pixel (x0-3, y0-3)
    uint8_t *scT = lds + inBytes;                         // (ah+2) x SP: scores, area pixel (c, r) at byte (r+1)*SP + c+4
    unsigned short *cand = (unsigned short *)(scT + scBytes);
    unsigned long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tPrev = 0;
    if (PROF) tPrev = __builtin_amdgcn_s_memtime();
    XCD_REMAP_XY(bx, f);
    const int lane = threadIdx.x;
    const int cFirst = bx * K, cEnd = min(cFirst + K, cellsPerFrame);
    const uint8_t *frame0 = img0 + (size_t)f * img0FramePitch, *frameP = pyr + (size_t)f * pyrBytes;
    // window of a cell: rows y0-3 .. y1+2, bytes x0-3 .. in units of 16 (the last unit spills up to 15 bytes past x1+2: x1+18 <= w-1 stays inside the
    // row); unit u = (row u / nu, column u % nu), all of a lane's units requested back to back.  (Skipped cells carry a 1 x 1 area: their loads are
    // valid and unused - unconditional loads keep the units in registers; behind a branch the compiler parks them in scratch memory and waits.)
    uint4 wv[NS];
    int wo[NS], ur[NS], uc[NS];      // per unit of the lane: LDS offset in the window (-1: none), window row, 16-byte column: recomputed only when a cell's window shape differs from the previous one's
    uint32_t shape = 0xffffffffu;
#define FC_REQUEST(E) do { \
        const int lvl_ = (int)(((E).dim >> 16) & 0xffu), sp_ = lvl_ ? (E).pitch : img0Stride; \
        const uint8_t *base_ = (lvl_ ? frameP + (E).off : frame0) + (size_t)((int)((E).xy >> 16) - 3) * sp_ + ((int)((E).xy & 0xffffu) - 3); \
        if ((E).units != shape) { \
            shape = (E).units; \
            const int nu_ = (int)((E).units & 0xffu), ntot_ = (int)((E).units >> 8), inv_ = (int)(E).inv; \
            _Pragma("unroll") for (int k_ = 0; k_ < NS; k_++) { \
                const int u_ = min(lane + 64 * k_, ntot_ - 1); \
                ur[k_] = __mul24(u_, inv_) >> 16; uc[k_] = 16 * (u_ - __mul24(ur[k_], nu_)); \
                wo[k_] = lane + 64 * k_ < ntot_ ? ur[k_] * P + uc[k_] : -1; \
            } \
        } \
        _Pragma("unroll") for (int k_ = 0; k_ < NS; k_++) wv[k_] = load16u(base_ + (size_t)ur[k_] * sp_ + uc[k_]); \
    } while (0)
    OrbxFcCell e = cells[cFirst], en = cells[min(cFirst + 1, cEnd - 1)];
    FC_REQUEST(e);
    FC_STAMP(0);
    for (int ci = cFirst; ci < cEnd; ci++) {
    int pNc1 = 0, pNc2 = 0, pRetry = 0;
    const int aw = (int)(e.dim & 0xffu), ah = (int)((e.dim >> 8) & 0xffu);
    const bool valid = (e.dim >> 24) != 0;
    const int x0 = (int)(e.xy & 0xffffu), y0 = (int)(e.xy >> 16);
    const int lpitch = e.pitch, cellCap = (int)e.cap;
    const uint32_t loff = e.off, slotOff = e.slot;
    int *cnt = cellCount + (size_t)f * cellsPerFrame + ci;
    {
#pragma unroll
        for (int k = 0; k < NS; k++)
            if (wo[k] >= 0) *(uint4 *)(inT + wo[k]) = wv[k];
        uint4 *z = (uint4 *)scT;
        for (int i = lane; i < (ah + 2) * (SP / 16); i += 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    // the next cell's window and the entry of the cell after it: in flight while this cell is computed (past the wave's last cell: that cell's
    // window once more, unused)
    e = en;
    FC_REQUEST(e);
    en = cells[min(ci + 2, cEnd - 1)];
    if (!valid) {             // skipped / degenerate cell (:1101, :1112)
        if (lane == 0) *cnt = 0;
        continue;
    }
    __syncthreads();
    FC_STAMP(1);
    uint8_t *dbg = DBG && scoreDbg ? scoreDbg + (size_t)f * pyrBytes + loff : nullptr;
    if (dbg)   // parity tap: pixels that fail the pre-test have score 0
        for (int p = lane; p < aw * ah; p += 64) dbg[(size_t)(y0 + p / aw) * lpitch + (x0 + p % aw)] = 0;

    // ---- phase A ----
    const int S = (aw + 15) >> 4;                                           // segments per row (1..4)
    const int rpp = S == 1 ? 64 : S == 2 ? 32 : S == 3 ? 21 : 16;           // rows per pass
    const int rl = S == 1 ? lane : S == 2 ? (lane >> 1) : S == 3 ? ((lane * 43) >> 7) : (lane >> 2);
    const int seg = lane - rl * S, c0 = 16 * seg;
    const uint32_t colMask = (aw - c0 >= 16) ? 0xffffu : (aw - c0 <= 0 ? 0u : ((1u << (aw - c0)) - 1u));
    // The reference runs FAST at iniThFAST and only a cell without any keypoint again at minThFAST (:1126-1136).  Here too: the first
    // pass pre-tests against iniThFAST - a third of the survivors of a minThFAST pre-test on textured images, and the exact score of
    // the survivors (phase B) is ~40 % of the kernel - and scores below iniThFAST count as "not a corner", exactly what the
    // reference's first cv::FAST call sees; only when no iniThFAST keypoint survives the NMS does the cell run again at minThFAST.
    // (With the parity taps on, the single minThFAST pass writes the full score map.)
    int nc = 0, th = dbg ? minTh : iniTh, base = 0;
    bool anyIni = false;
    uint32_t *slot = cellSlots + (size_t)f * slotsPerFrame + slotOff;
    for (;;) {
    nc = 0;
    const uint32_t thrPk = (uint32_t)(th & 0xffff) * 0x00010001u;
    for (int rowBase = 0; rowBase < ah; rowBase += rpp) {
        const int r = rowBase + rl;
        const bool live = rl < rpp && r < ah;
        const uint8_t *wp = inT + min(r, ah - 1) * P + c0;
        uint32_t W[7][6];
#pragma unroll
        for (int dy = 0; dy < 7; dy += 3) {      // rows -3, 0, +3: the four compass points of the circle and the centre
            const uint4 a = *(const uint4 *)(wp + dy * P);
            const uint2 b = *(const uint2 *)(wp + dy * P + 16);
            W[dy][0] = a.x; W[dy][1] = a.y; W[dy][2] = a.z; W[dy][3] = a.w; W[dy][4] = b.x; W[dy][5] = b.y;
        }
        uint32_t mask = 0;
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            uint32_t A = 0u, B = 0x00ff00ffu;
#pragma unroll
            for (int k = 0; k < 8; k += 4) {     // the opposite pairs (0, 8) and (4, 12) only - see the header
                const uint32_t xa = FC_PAIR(W[3 + FAST_DY(k)], j + 3 + FAST_DX(k));
                const uint32_t xb = FC_PAIR(W[3 + FAST_DY(k + 8)], j + 3 + FAST_DX(k + 8));
                A = pk_max_u16(A, pk_min_u16(xa, xb));
                B = pk_min_u16(B, pk_max_u16(xa, xb));
            }
            const uint32_t vv = FC_PAIR(W[3], j + 3);
            const uint32_t m = pk_max_i16(pk_sub_i16(vv, A), pk_sub_i16(B, vv));
            // m > th  <=>  th - m < 0: the two sign bits go to bit j (pixel j) and bit 16 + j (pixel j + 1)
            mask |= ((pk_sub_i16(thrPk, m) >> 15) & 0x00010001u) << j;
        }
        mask = (mask & 0x5555u) | ((mask >> 15) & 0xaaaau);
        mask = live ? (mask & colMask) : 0u;
        if (PROF) { asm volatile("" :: "v"(mask)); FC_STAMP(th == iniTh || dbg ? 2 : 6); }
        const int cntL = __popc(mask);
        const int incl = wave_incl_scan_dpp(cntL);
        unsigned short *o = cand + (nc + incl - cntL);
        const int code0 = (r << 6) | c0;
        while (mask) {
            const int bpos = __ffs(mask) - 1;
            mask &= mask - 1;
            *o++ = (unsigned short)(code0 + bpos);
        }
        nc += __builtin_amdgcn_readlane(incl, 63);
        FC_STAMP(th == iniTh || dbg ? 3 : 7);
    }
    __syncthreads();
    if (PROF) { if (th == iniTh || dbg) pNc1 = nc; else { pNc2 = nc; pRetry = 1; } }

    // ---- phase B: full FAST score of the surviving pixels ----
    for (int i = lane; i < nc; i += 64) {
        const int code = cand[i], r = code >> 6, c = code & 63;
        const uint8_t *q = inT + r * P + c;          // window byte of circle offset (dx, dy): q[(dy+3)*P + dx+3]
        const int v = q[3 * P + 3];
        int x[16];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = q[(3 + FAST_DY(k)) * P + 3 + FAST_DX(k)];
        int a3[16], b3[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            a3[k] = min3u_asm(x[k], x[(k + 1) & 15], x[(k + 2) & 15]);
            b3[k] = max3u_asm(x[k], x[(k + 1) & 15], x[(k + 2) & 15]);
        }
        int maxA = 0, minB = 255;   // max over the 16 arcs of min(x), min over the arcs of max(x)
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            const int a9 = min3u_asm(a3[k], a3[(k + 3) & 15], a3[(k + 6) & 15]), a9n = min3u_asm(a3[k + 1], a3[(k + 4) & 15], a3[(k + 7) & 15]);
            const int b9 = max3u_asm(b3[k], b3[(k + 3) & 15], b3[(k + 6) & 15]), b9n = max3u_asm(b3[k + 1], b3[(k + 4) & 15], b3[(k + 7) & 15]);
            maxA = k == 0 ? max(a9, a9n) : max3u_asm(maxA, a9, a9n);
            minB = k == 0 ? min(b9, b9n) : min3u_asm(minB, b9, b9n);
        }
        // dark arc: all x < v - t  <=>  t < v - max(x);  bright arc: all x > v + t  <=>  t < min(x) - v;  score = largest such t
        int sco = max(v - minB, maxA - v) - 1;
        sco = sco >= th ? sco : 0;
        scT[(r + 1) * SP + c + 4] = (uint8_t)sco;
        if (dbg) dbg[(size_t)(y0 + r) * lpitch + (x0 + c)] = (uint8_t)sco;
    }
    __syncthreads();
    FC_STAMP(th == iniTh || dbg ? 4 : 8);

    // ---- phase C: strict 3x3 maxima, threshold choice, ordered emission ----
    // A pass at iniThFAST sees scores below iniThFAST as zero, so every maximum it finds is an iniThFAST keypoint and all of them are kept; the retry at
    // minThFAST runs only when there was none (a maximum of the first pass stays one: its neighbours only gained smaller scores... and a non-maximum
    // stays none), so it keeps every maximum too: the maxima go to the cell's slot as they are found, in list (= raster) order by ballot.  (The parity
    // tap's single minThFAST pass has both kinds in one pass: it flags them and emits afterwards.)
    base = 0;
    anyIni = false;
    for (int i0 = 0; i0 < nc; i0 += 64) {
        const int i = i0 + lane;
        bool nms = false, ini = false;
        int code = 0, v = 0;
        if (i < nc) {
            code = cand[i];
            const int r = code >> 6, c = code & 63;
            const uint8_t *sp = scT + (r + 1) * SP + c + 4;
            v = sp[0];
            const int nb = max3i(max3i(sp[-1], sp[1], sp[-SP - 1]), max3i(sp[-SP], sp[-SP + 1], sp[SP - 1]), max(sp[SP], sp[SP + 1]));
            nms = v > nb;                             // (v > nb >= 0: a zero score is never a maximum)
            ini = nms && v >= iniTh;
        }
        if (dbg) {
            if (i < nc) cand[i] = (unsigned short)(code | (nms ? 0x1000 : 0) | (ini ? 0x2000 : 0));   // own entry: no cross-lane hazard
            anyIni = anyIni || ini;
        } else {
            const unsigned long long m = __ballot(nms);
            if (nms) {
                const int r = code >> 6, c = code & 63;
                const int idx = base + __popcll(m & ((1ull << lane) - 1ull));
                if (idx < cellCap) slot[idx] = (uint32_t)(x0 + c - ORBX_BORDER) | ((uint32_t)(y0 + r - ORBX_BORDER) << 12) | ((uint32_t)v << 24);
            }
            base += __popcll(m);
        }
    }
    if (dbg) anyIni = __any(anyIni); else anyIni = base > 0;
    FC_STAMP(th == iniTh || dbg ? 5 : 9);
    if (anyIni || th == minTh) break;
    th = minTh;           // :1132-1136: nothing at iniThFAST -> the whole cell again at minThFAST (its phase B rewrites every score of the first pass)
    __syncthreads();      // (single wave: orders the LDS list / score tile reuse)
    }
    if (dbg) {
        const int keepBit = anyIni ? 0x2000 : 0x1000;   // iniThFAST keypoints exist -> the minThFAST retry is skipped (:1132)
        base = 0;
        for (int i0 = 0; i0 < nc; i0 += 64) {
            const int i = i0 + lane;
            bool keep = false;
            int code = 0;
            if (i < nc) { code = cand[i]; keep = (code & keepBit) != 0; }
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int r = (code >> 6) & 63, c = code & 63;
                const uint32_t v = scT[(r + 1) * SP + c + 4];
                const int idx = base + __popcll(m & ((1ull << lane) - 1ull));
                if (idx < cellCap) slot[idx] = (uint32_t)(x0 + c - ORBX_BORDER) | ((uint32_t)(y0 + r - ORBX_BORDER) << 12) | (v << 24);
            }
            base += __popcll(m);
        }
    }
    if (lane == 0) *cnt = min(base, cellCap);
    __syncthreads();      // (the next cell's window and zeros overwrite the tiles)
    FC_STAMP(10);
    if (PROF && lane == 0) {      // one 16-word record per cell, no atomics (208k waves adding to the same words stall the loads of the waves behind them)
        unsigned long long *rec = prof + 32 + ((size_t)f * cellsPerFrame + ci) * 16;
        for (int i = 0; i < 12; i++) { rec[i] = pacc[i]; pacc[i] = 0; }
        rec[12] = (unsigned long long)pRetry; rec[13] = (unsigned long long)pNc1; rec[14] = (unsigned long long)pNc2; rec[15] = (unsigned long long)min(base, cellCap);
    }
    }
}
#undef FC_STAMP
#undef FC_REQUEST

// ------------------------------------------------------------------------------------
// Quadtree distribution (DistributeOctTree + DivideNode, src/ORBextractor.cc:635-703,
// 706-1049).  One workgroup per (frame, level).
//
// Points never move.  Every point carries the index of the node that holds it (its LABEL); a
// node is a box + a point count in an LDS array that IS the reference's std::list (list
// order, creation order and the "later-created first" tie rule of the careful rounds follow
// DESIGN.md section 5 / oracle/orb_oracle.cc octree()).  A split round is
//   node phase   per candidate (count > 1) the four quadrant counts are already known (qc, filled by
//                the previous point pass): children per candidate, creation indices by ONE scan,
//                [careful rounds: processing order = rank by (count desc, list position asc), the
//                first candidates that bring the list to >= N nodes], new list = reversed children
//                ++ untouched nodes, and a table  map[old node][quadrant] -> new node;
//   point pass   label = map[label][quadrant of the point in its old box]; if the new node has more
//                than one point, its quadrant counter for the next round gets the point (LDS atomic).
// The split is a stable partition in the reference, so the points of a node are always in the order
// of the candidate list: "first maximum wins" (:1029-1041) = largest (response, -original index),
// one LDS atomicMax per point at the end.  No sorting, no scatter, no per-chunk bookkeeping:
// 5 workgroup barriers per full pass and 9 per careful round (the previous partition-based kernel
// needed ~25), and no limit on the number of candidates (the point arrays are sized for the
// worst case, every second pixel in both directions a maximum).
// ------------------------------------------------------------------------------------
struct OtBox { short x0, y0, x1, y1; };

template <typename T> __device__ __forceinline__ T wave_incl_scan(T v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        T u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}

// exclusive scan of arr[0..n) in place (LDS); returns the total.  All BT threads of the workgroup call.
template <int BT = 256, typename T> __device__ T block_exscan(T *arr, int n, T *wsum /* >= BT / 64 entries */)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (n + BT - 1) / BT;
    const int b = tid * per;
    T s = 0;
    for (int k = 0; k < per; k++) if (b + k < n) s += arr[b + k];
    T inc = wave_incl_scan(s, lane);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    T off = 0, total = 0;
#pragma unroll
    for (int i = 0; i < BT / 64; i++) { const T v = wsum[i]; off += i < w ? v : (T)0; total += v; }
    T run = off + inc - s;
    for (int k = 0; k < per; k++)
        if (b + k < n) { T v = arr[b + k]; arr[b + k] = run; run += v; }
    __syncthreads();
    return total;
}

// Two exclusive scans at once over entries that the calling thread FILLS itself: thread t owns the contiguous entries
// [t*per, (t+1)*per), fill(i, a, b) produces entry i of both arrays, so no barrier is needed between filling and scanning.
template <int BT = 256, typename F> __device__ __forceinline__ void block_fill_exscan2(int *A, int *B, int n, int *wsA, int *wsB, int &totA, int &totB, F fill)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (n + BT - 1) / BT;
    const int b = tid * per;
    int sa = 0, sb = 0;
    for (int k = 0; k < per; k++)
        if (b + k < n) { int a, c; fill(b + k, a, c); A[b + k] = a; B[b + k] = c; sa += a; sb += c; }
    const int incA = wave_incl_scan(sa, lane), incB = wave_incl_scan(sb, lane);
    if (lane == 63) { wsA[w] = incA; wsB[w] = incB; }
    __syncthreads();
    int offA = 0, offB = 0;
    totA = 0; totB = 0;
#pragma unroll
    for (int i = 0; i < BT / 64; i++) { const int va = wsA[i], vb = wsB[i]; offA += i < w ? va : 0; offB += i < w ? vb : 0; totA += va; totB += vb; }
    int runA = offA + incA - sa, runB = offB + incB - sb;
    for (int k = 0; k < per; k++)
        if (b + k < n) { const int a = A[b + k], c = B[b + k]; A[b + k] = runA; B[b + k] = runB; runA += a; runB += c; }
    __syncthreads();
}

__device__ __forceinline__ int ot_quadrant(uint32_t p, const OtBox b)
{
    const int mx = b.x0 + ((b.x1 - b.x0 + 1) >> 1), my = b.y0 + ((b.y1 - b.y0 + 1) >> 1);   // x0 + ceil(w / 2), :640-641
    const int x = (int)(p & 0xfffu), y = (int)((p >> 12) & 0xfffu);
    return (x < mx ? 0 : 1) + (y < my ? 0 : 2);
}

// The quadtree of ONE (level, frame): the body shared by k_octree (batches: one workgroup per (level, frame) through the XCD bijection) and by
// k_octree_blur (combined single-frame calls: the blur's tiles and the host pyramid copy ride in the same launch).
template <int NODECAP, int BT>      // BT = threads of the workgroup (256 in batches: 2048 quadtrees per launch; 1024 in the single-frame launch sets: eight)
__device__ __forceinline__ void octree_body(const OrbxGeom *__restrict__ g, const int *__restrict__ cellCount, const uint32_t *__restrict__ cellSlots,
                                            const uint8_t *__restrict__ binTab, uint32_t *__restrict__ ptBuf, uint32_t *__restrict__ labBuf,
                                            OrbxLevelKp *__restrict__ lvlKp, int *__restrict__ lvlCnt, int *__restrict__ status, const int l, const int f, const int nframes)
{
    __shared__ OtBox box[2][NODECAP];
    __shared__ int cnt[2][NODECAP];
    __shared__ uint32_t qc[2][NODECAP][4];          // quadrant counts of the nodes with more than one point, per list
    __shared__ unsigned short nmap[NODECAP][4];     // old node, quadrant -> new node
    __shared__ int scanA[NODECAP];                  // children created before slot k (processing order)
    __shared__ int scanB[NODECAP];                  // untouched nodes before node i (list order)
    __shared__ __attribute__((aligned(16))) uint32_t key[NODECAP];     // careful rounds: (count << log2 NODECAP) | (NODECAP - 1 - list position) of a candidate, 0 otherwise
    __shared__ unsigned short byProc[NODECAP];      // careful rounds: processing rank -> node
    __shared__ unsigned char sel[NODECAP];          // node is split in this round
    __shared__ int wsA[16], wsB[16];
    __shared__ int sh_misc[4 + ORBX_MAX_INI];

    const int tid = threadIdx.x, lane = tid & 63;
    const OrbxLevel &lv = g->lv[l];
    const int N = lv.quota;
    uint32_t *pts = ptBuf + (size_t)f * g->slotsPerFrame + lv.slotBase;    // capacity = cells x slots per cell: every candidate the detector can emit
    uint32_t *lab = labBuf + (size_t)f * g->slotsPerFrame + lv.slotBase;
    int *stat = status + f;          // per-frame word; status[nframes] collects the whole batch (device-visible for resident consumers)
    int *statAll = status + nframes;

    // ---- gather the level's candidates in vToDistributeKeys order (cells row-major) ----
    const int ncell = lv.nCols * lv.nRows;
    const int *cc = cellCount + (size_t)f * g->cellsPerFrame + lv.cellBase;
    int *cellOff = (int *)lab;        // scratch: labels are written after the gather
    for (int i = tid; i < ncell; i += BT) cellOff[i] = cc[i];
    if (tid < 4 + ORBX_MAX_INI) sh_misc[tid] = 0;
    __syncthreads();
    const int M = block_exscan<BT>(cellOff, ncell, wsA);
    {
        const uint32_t *slots = cellSlots + (size_t)f * g->slotsPerFrame + lv.slotBase;
        // one THREAD per cell: a cell holds a handful of candidates, and a wave walking the cells one after the other pays a memory
        // round trip per cell
        for (int c = tid; c < ncell; c += BT) {
            const int n = cc[c], o = cellOff[c];
            const uint32_t *sc = slots + (size_t)c * lv.cellCap;
            for (int k = 0; k < n; k++) pts[o + k] = sc[k];
        }
    }
    __syncthreads();
    if (M == 0) { if (tid == 0) lvlCnt[f * g->nlevels + l] = 0; return; }

    // ---- initial nodes (src/ORBextractor.cc:719-788): the root split by x; empty ones are erased ----
    int cur = 0, nn = 0, ncand = 0;
    {
        const uint8_t *bin = binTab + lv.binOff;
        const int nIni = lv.nIni;      // 1 .. ORBX_MAX_INI initial nodes (round(width / height) of the level's detection window, :719)
        int c8[ORBX_MAX_INI] = {};
        for (int j0 = 0; j0 < M; j0 += BT) {
            const int j = j0 + tid;
            const int q = j < M ? (int)bin[pts[j] & 0xfff] : -1;
#pragma unroll
            for (int Q = 0; Q < ORBX_MAX_INI; Q++) if (Q < nIni) c8[Q] += __popcll(__ballot(q == Q));
        }
        if (lane == 0)
#pragma unroll
            for (int Q = 0; Q < ORBX_MAX_INI; Q++) if (c8[Q]) atomicAdd(&sh_misc[4 + Q], c8[Q]);
        __syncthreads();
        int idx[ORBX_MAX_INI], k = 0;
#pragma unroll
        for (int Q = 0; Q < ORBX_MAX_INI; Q++) { idx[Q] = -1; if (Q < nIni && sh_misc[4 + Q] > 0) idx[Q] = k++; }
        nn = k;
        {
            int myIdx = -1;
#pragma unroll
            for (int Q = 0; Q < ORBX_MAX_INI; Q++) myIdx = tid == Q ? idx[Q] : myIdx;
            if (tid < ORBX_MAX_INI && myIdx >= 0) {
                OtBox b;
                b.x0 = (short)lv.iniX[tid]; b.x1 = (short)lv.iniX[tid + 1]; b.y0 = 0; b.y1 = (short)(lv.h - 2 * ORBX_BORDER);
                box[0][myIdx] = b; cnt[0][myIdx] = sh_misc[4 + tid];
                qc[0][myIdx][0] = qc[0][myIdx][1] = qc[0][myIdx][2] = qc[0][myIdx][3] = 0u;
            }
            if (tid < ORBX_MAX_INI) wsB[4 + tid] = myIdx;      // bin -> node (a runtime-indexed register array would live in scratch memory)
        }
#pragma unroll
        for (int Q = 0; Q < ORBX_MAX_INI; Q++) if (idx[Q] >= 0 && sh_misc[4 + Q] > 1) ncand++;
        __syncthreads();
        for (int j = tid; j < M; j += BT) {
            const uint32_t p = pts[j];
            const int nd = wsB[4 + bin[p & 0xfff]];
            lab[j] = (uint32_t)nd;
            if (cnt[0][nd] > 1) atomicAdd(&qc[0][nd][ot_quadrant(p, box[0][nd])], 1u);
        }
        __syncthreads();
    }

    // ---- split rounds ----
    bool careful = false;
    while (ncand > 0) {            // no candidate: the pass changes nothing, "size unchanged" -> bFinish (:907-913)
        const int nxt = cur ^ 1;
        const int prev = nn;
        int nsel, totalCreated, nUnsel;
        if (!careful) {
            // full pass: every candidate is split, in list order (slot k = node k)
            nsel = nn;
            block_fill_exscan2<BT>(scanA, scanB, nn, wsA, wsB, totalCreated, nUnsel, [&](int i, int &a, int &b) {
                const bool isc = cnt[cur][i] > 1;
                sel[i] = isc ? 1 : 0;
                a = isc ? (qc[cur][i][0] != 0) + (qc[cur][i][1] != 0) + (qc[cur][i][2] != 0) + (qc[cur][i][3] != 0) : 0;
                b = isc ? 0 : 1;
            });
        } else {
            // careful round (:934-1011): candidates by (count desc, list position asc) - the list position encodes the creation order, so
            // the reference's pointer tie-break is an integer compare -, split until the list reaches N nodes
            // One key per candidate, (count << log2 NODECAP) | (NODECAP - 1 - position): larger key = earlier; the rank is the number of larger
            // keys, counted four keys per LDS read (a scalar loop over the counts took 4 us of a 9 us round at 217 nodes, 15 of 20 us at 434).
            // The count field is 32 - log2 NODECAP bits wide (2^21 candidates in one node at NODECAP 2048, 2^24 at 256): build_geometry
            // (orbx_extractor.hip) refuses a level whose worst-case candidate count does not fit.
            constexpr int KEY_SHIFT = NODECAP == 256 ? 8 : NODECAP == 512 ? 9 : NODECAP == 1024 ? 10 : 11;
            for (int i = tid; i < ((nn + 3) & ~3); i += BT) {
                const int c = i < nn ? cnt[cur][i] : 0;
                key[i] = c > 1 ? ((uint32_t)c << KEY_SHIFT) | (uint32_t)(NODECAP - 1 - i) : 0u;
                if (i < nn) sel[i] = 0;
            }
            if (tid == 0) sh_misc[1] = ncand;
            __syncthreads();
            for (int i = tid; i < nn; i += BT) {
                const uint32_t ki = key[i];
                if (ki) {
                    int rank = 0;
                    for (int s4 = 0; s4 < nn; s4 += 4) {
                        const uint4 k4 = *(const uint4 *)&key[s4];
                        rank += (k4.x > ki) + (k4.y > ki) + (k4.z > ki) + (k4.w > ki);
                    }
                    byProc[rank] = (unsigned short)i;
                }
            }
            __syncthreads();
            int totAll, dummy;
            block_fill_exscan2<BT>(scanA, scanB, ncand, wsA, wsB, totAll, dummy, [&](int k, int &a, int &b) {
                const int i = byProc[k];
                a = (qc[cur][i][0] != 0) + (qc[cur][i][1] != 0) + (qc[cur][i][2] != 0) + (qc[cur][i][3] != 0);
                b = a;          // children of slot k, kept next to its exclusive prefix
            });
            // smallest k with  nn + sum_{j<=k}(children_j - 1) >= N  (:1003); scanB[k] is exclusive too: children_k = next prefix - this one
            for (int k = tid; k < ncand; k += BT) {
                const int ck = (k + 1 < ncand ? scanA[k + 1] : totAll) - scanA[k];
                if (nn + scanA[k] + ck - (k + 1) >= N) atomicMin(&sh_misc[1], k + 1);
            }
            __syncthreads();
            nsel = sh_misc[1];
            totalCreated = nsel < ncand ? scanA[nsel] : totAll;
            for (int k = tid; k < nsel; k += BT) sel[byProc[k]] = 1;
            __syncthreads();
            int dummy2;
            block_fill_exscan2<BT>(scanB, scanB, nn, wsA, wsB, dummy2, nUnsel, [&](int i, int &a, int &b) { a = b = sel[i] ? 0 : 1; });
        }
        const int newN = totalCreated + nUnsel;
        if (newN > NODECAP) { if (tid == 0) { atomicOr(stat, ORBX_DEV_ERR_NODECAP); atomicOr(statAll, ORBX_DEV_ERR_NODECAP); } break; }
        // ---- the new list: reversed children (push_front of each child, :676-699 / :964-1000) ++ the untouched nodes in their old order ----
        int myCand = 0, myExpand = 0;
        for (int i = tid; i < nn; i += BT) {
            if (sel[i]) continue;
            const int pos = totalCreated + scanB[i];
            box[nxt][pos] = box[cur][i];
            const int c = cnt[cur][i];
            cnt[nxt][pos] = c;
            if (c > 1) {     // an unsplit candidate keeps its points, hence its quadrant counts
                myCand++;
                qc[nxt][pos][0] = qc[cur][i][0]; qc[nxt][pos][1] = qc[cur][i][1]; qc[nxt][pos][2] = qc[cur][i][2]; qc[nxt][pos][3] = qc[cur][i][3];
            }
            nmap[i][0] = nmap[i][1] = nmap[i][2] = nmap[i][3] = (unsigned short)pos;
        }
        for (int k = tid; k < nsel; k += BT) {
            const int i = careful ? (int)byProc[k] : k;
            if (!sel[i]) continue;
            const OtBox nd = box[cur][i];
            const short mx = (short)(nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1)), my = (short)(nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1));
            int ci = scanA[k];
#pragma unroll
            for (int Q = 0; Q < 4; Q++) {
                const int c = (int)qc[cur][i][Q];
                if (c > 0) {
                    const int pos = totalCreated - 1 - ci;     // later children sit further front
                    ci++;
                    OtBox ch;
                    ch.x0 = (Q & 1) ? mx : nd.x0; ch.x1 = (Q & 1) ? nd.x1 : mx;
                    ch.y0 = (Q & 2) ? my : nd.y0; ch.y1 = (Q & 2) ? nd.y1 : my;
                    box[nxt][pos] = ch; cnt[nxt][pos] = c;
                    qc[nxt][pos][0] = qc[nxt][pos][1] = qc[nxt][pos][2] = qc[nxt][pos][3] = 0u;
                    nmap[i][Q] = (unsigned short)pos;
                    if (c > 1) myExpand++;
                }
            }
        }
        if (myExpand) atomicAdd(&sh_misc[2], myExpand);
        if (myCand) atomicAdd(&sh_misc[3], myCand);
        __syncthreads();
        const int nToExpand = sh_misc[2];
        ncand = nToExpand + sh_misc[3];
        const bool finish = newN >= N || newN == prev;                   // :907-913, :1007
        // ---- point pass: new labels; points of freshly created nodes with more than one point feed the next round's counts ----
        for (int j = tid; j < M; j += BT) {
            const int nd = (int)lab[j];
            if (!sel[nd]) { lab[j] = nmap[nd][0]; continue; }
            const uint32_t p = pts[j];
            const int nl = nmap[nd][ot_quadrant(p, box[cur][nd])];
            lab[j] = (uint32_t)nl;
            if (!finish && cnt[nxt][nl] > 1) atomicAdd(&qc[nxt][nl][ot_quadrant(p, box[nxt][nl])], 1u);
        }
        __syncthreads();
        if (tid == 0) { sh_misc[2] = 0; sh_misc[3] = 0; }
        nn = newN;
        cur = nxt;
        if (finish) break;
        if (!careful && nn + nToExpand * 3 > N) careful = true;         // :931
    }

    // ---- best response per node, first maximum wins (:1018-1048): largest (response, -candidate index) ----
    {
        uint32_t *best = &qc[cur ^ 1][0][0];     // free now: one word per node
        OrbxLevelKp *out = lvlKp + (size_t)f * g->kpPerFrame + lv.kpBase;
        if (nn > lv.kpCap) { if (tid == 0) { atomicOr(stat, ORBX_DEV_ERR_KPCAP); atomicOr(statAll, ORBX_DEV_ERR_KPCAP); } nn = lv.kpCap; }
        __syncthreads();
        for (int i = tid; i < NODECAP; i += BT) best[i] = 0u;
        __syncthreads();
        for (int j = tid; j < M; j += BT) {
            const uint32_t nd = lab[j];
            if (nd < (uint32_t)NODECAP) atomicMax(&best[nd], (pts[j] & 0xff000000u) | (0x00ffffffu - (uint32_t)j));
        }
        __syncthreads();
        for (int i = tid; i < nn; i += BT) {
            const uint32_t b = pts[0x00ffffffu - (best[i] & 0x00ffffffu)];
            OrbxLevelKp kp;
            kp.x = (uint16_t)((b & 0xfff) + ORBX_BORDER); kp.y = (uint16_t)(((b >> 12) & 0xfff) + ORBX_BORDER);
            kp.score = (uint8_t)(b >> 24); kp.pad[0] = kp.pad[1] = kp.pad[2] = 0; kp.angle = 0.f; kp.ca = 1.f; kp.sb = 0.f;
            out[i] = kp;
        }
        if (tid == 0) lvlCnt[f * g->nlevels + l] = nn;
    }
}

// the caller's HOST pyramid copy of frame f (the public mvImagePyramid; levels >= 1 in the device layout, 16 bytes per lane across PCIe), chunk `chunk` of `nchunks`
__device__ __forceinline__ void host_pyramid_copy(const OrbxGeom *__restrict__ g, const OrbxCombMember *__restrict__ comb, const uint8_t *__restrict__ engPyr, int f, int chunk, int nchunks)
{
    uint4 *dst = (uint4 *)comb[f].hostPyr;
    if (!dst) return;
    const uint4 *src = (const uint4 *)(engPyr + (size_t)f * g->pyrBytes);
    const size_t n = g->pyrBytes >> 4, bt = blockDim.x, stride = (size_t)nchunks * bt;
    for (size_t i = (size_t)chunk * bt + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

template <int NODECAP>
__global__ __launch_bounds__(256) void k_octree(const OrbxGeom *__restrict__ g, const int *__restrict__ cellCount, const uint32_t *__restrict__ cellSlots,
                                                const uint8_t *__restrict__ binTab, uint32_t *__restrict__ ptBuf, uint32_t *__restrict__ labBuf,
                                                OrbxLevelKp *__restrict__ lvlKp, int *__restrict__ lvlCnt, int *__restrict__ status,
                                                const OrbxCombMember *__restrict__ comb, const uint8_t *__restrict__ engPyr)
{
    // (level, frame) through the same XCD bijection as the detector before and the descriptor kernel after this stage: a frame's
    // candidates were written, and its keypoints will be read, by the XCD that owns the frame's range - in the natural order
    // block (l, f) lands on XCD l, and both hand-overs cross the fabric
    XCD_REMAP_XY(l, f);
    if (l >= g->nlevels) {
        // Combined single-frame calls (comb != NULL, gridDim.x = nlevels + copy blocks): the quadtree is eight latency-bound workgroups per
        // frame on an otherwise idle device, so the caller's host pyramid copy travels in the same launch, next to it, instead of as ~19 us
        // at the end of the chain.
        host_pyramid_copy(g, comb, engPyr, f, l - g->nlevels, (int)gridDim.x - g->nlevels);
        return;
    }
    octree_body<NODECAP, 256>(g, cellCount, cellSlots, binTab, ptBuf, labBuf, lvlKp, lvlCnt, status, l, f, (int)gridDim.y);
}

// sin/cos of the keypoint angle: glibc's sinf/cosf algorithm in double, restated so the device
// rounds like the libm the reference calls (oracle/prims.h op_sincosf, tests/test_sincos.py).
__device__ __forceinline__ float sinf_poly_d(double x, double x2, int n, bool neg)
{
    // coefficients of glibc 2.35 __sincosf_table[0]; table[1] is its negation (neg)
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        double x3 = x * x2, t1 = s2 + x2 * s3, x7 = x3 * x2, s = x + x3 * s1;
        return (float)(s + x7 * t1);
    } else {
        double k0 = neg ? -c0 : c0, k1 = neg ? -c1 : c1, k2 = neg ? -c2 : c2, k3 = neg ? -c3 : c3, k4 = neg ? -c4 : c4;
        double x4 = x2 * x2, u2 = k3 + x2 * k4, u1 = k0 + x2 * k1, x6 = x4 * x2, c = u1 + x4 * k2;
        return (float)(c + x6 * u2);
    }
}

__device__ __forceinline__ void sincosf_glibc(float y, float &sn, float &cs)
{
    double x = y;
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ff;
    if (top < ((0x3f490fdbu >> 20) & 0x7ff)) {   // |y| < pi/4 (abstop12 compare)
        double x2 = x * x;
        if (top < ((0x39800000u >> 20) & 0x7ff)) { sn = y; cs = 1.0f; return; }   // |y| < 2^-12
        sn = sinf_poly_d(x, x2, 0, false);
        cs = sinf_poly_d(x, x2, 1, false);
        return;
    }
    double r = x * 0x1.45F306DC9C883p+23;
    int n = ((int)r + 0x800000) >> 24;
    x = x - n * 0x1.921FB54442D18p0;
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    const bool neg = (n & 2) != 0;
    sn = sinf_poly_d(x * sgn, x * x, n, neg);
    cs = sinf_poly_d(x * sgn, x * x, n ^ 1, neg);
}

// ------------------------------------------------------------------------------------
// Orientation: IC_Angle (src/ORBextractor.cc:108-161) = integer moments over the
// 749-pixel disc of radius 15 on the UNBLURRED level, then cv::fastAtan2.  Half a wave per
// keypoint, lanes own the disc rows; integer sums are exact in any order.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale, p5 = 0.1555786518463281f * scale,
                p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + 2.220446049250313e-16f);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + 2.220446049250313e-16f);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ------------------------------------------------------------------------------------
// 7x7 Gaussian, sigma 2, BORDER_REFLECT_101, u8 fixed point (cv::GaussianBlur on the
// cloned level, src/ORBextractor.cc:1626-1634): horizontal pass exact in 16 bits,
// vertical in 32, (sum + 2^15) >> 16.
// ONE WAVE per 64x32 output tile.  The 80-byte x 38-row input window is staged in LDS with
// 16-byte loads; lane = (4-column group, 8-row group) then runs BOTH passes in registers:
//   horizontal, 14 rows x 4 px: two or three v_dot4_u32_u8 per pixel on the window words as loaded (the TAPS are
//     shifted to the pixel's byte position, not the data), the sums of rows (2m, 2m+1) of a column packed into one word;
//   vertical, 8 rows x 4 px: four v_dot2_u32_u16 per output on those row pairs (even and odd output rows differ in the
//     tap pairs, not in the data), the +2^15 rounding rides in the accumulator.
// Taps are <= 255 and sum to <= 257 (checked at handle creation): the 16-bit sums cannot saturate.
// ~14 VALU lane-instructions per pixel; algorithmic traffic 2 bytes/pixel, the halo makes the
// read side 1.48x.
// ------------------------------------------------------------------------------------
#define BT_W 64
#define BT_H 32
#define BT_P 80                 /* LDS row pitch in bytes: x = X0-8 .. X0+71 */
#define BT_IH (BT_H + 6)

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (p < 0) p = -p;
    if (p >= len) p = 2 * len - 2 - p;
    return p;
}

// One 64 x 32 tile of the blurred pyramid by ONE wave: the body shared by k_blur (batches: one single-wave workgroup per tile) and k_octree_blur
// (combined single-frame calls: four tiles per 256-thread workgroup, next to the quadtree's workgroups).  `in` = the wave's own LDS window,
// `store` = false for the padding waves of the last workgroup (they run through the same barriers and store nothing).
template <bool CLAMP>      // taps that add up to more than 256 (the configuration allows 257): the output needs its clamp
__device__ __forceinline__ void blur_body(const OrbxGeom *__restrict__ g, const uint8_t *__restrict__ img0, int img0Stride, size_t img0FramePitch,
                                          const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur, const int bx, const int f, uint32_t *__restrict__ in, const int lane,
                                          const bool store)
{
    int bases[ORBX_MAX_LEVELS];
    const int nl = g->nlevels;
    for (int i = 0; i < nl; i++) bases[i] = g->lv[i].blurTileBase;
    const int l = find_level(bases, nl, bx);
    const OrbxLevel &lv = g->lv[l];
    const int t = bx - lv.blurTileBase;
    const int X0 = (t % lv.blurTilesX) * BT_W, Y0 = (t / lv.blurTilesX) * BT_H;
    const int w = lv.w, h = lv.h;
    int pitch;
    const uint8_t *src = level_ptr(g, l, f, img0, img0Stride, img0FramePitch, pyr, pitch);
    // ---- stage: item = (row r, 16-byte chunk c); interior chunks are one 16-byte load ----
#pragma unroll
    for (int it = 0; it < 3; it++) {
        const int item = lane + 64 * it;
        if (item < BT_IH * 5) {
            const int r = (item * 205) >> 10, c = item - 5 * r;   // item / 5 (exact for item < 1024)
            int y = reflect101(Y0 - 3 + r, h);
            y = min(max(y, 0), h - 1);
            const int x = X0 - 8 + 16 * c;
            const uint8_t *row = src + (size_t)y * pitch;
            uint4 v;
            if (x >= 0 && x + 15 < w) __builtin_memcpy(&v, row + x, 16);
            else {
                // a chunk that sticks out of the row: ONE load at the nearest position inside it, the bytes moved to where they belong.  What lies
                // outside the image is written afterwards (the three mirrored pixels on either side, below); the rest of the window's margin is
                // never read by a tap.  (The former byte-by-byte BORDER_REFLECT_101 was 170 instructions per chunk for the whole wave - a third of
                // the kernel's instructions, since almost half of all tiles touch an edge.)
                const int xs = min(max(x, 0), w - 16), d = x - xs;      // d = -8 at the left edge, 1 .. at the right one
                uint4 sv;
                __builtin_memcpy(&sv, row + xs, 16);
                if (d < 0) v = make_uint4(0u, 0u, sv.x, sv.y);
                else {
                    const uint32_t sw[8] = {sv.x, sv.y, sv.z, sv.w, 0u, 0u, 0u, 0u};
                    const int q = min(d >> 2, 4), bsh = d & 3;
                    uint32_t o[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        uint32_t lo = 0u, hi = 0u;
#pragma unroll
                        for (int t = 0; t < 8; t++) { lo = (k + q == t) ? sw[t] : lo; hi = (k + q + 1 == t) ? sw[t] : hi; }
                        o[k] = bsh == 0 ? lo : bsh == 1 ? __builtin_amdgcn_alignbyte(hi, lo, 1) : bsh == 2 ? __builtin_amdgcn_alignbyte(hi, lo, 2) : __builtin_amdgcn_alignbyte(hi, lo, 3);
                    }
                    v = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
            *(uint4 *)(in + r * (BT_P / 4) + 4 * c) = v;
        }
    }
    __syncthreads();
    if (X0 == 0 || X0 + BT_W + 3 > w) {
        // BORDER_REFLECT_101 in x for the three pixels a 7-tap filter reads beyond either edge: x = -1, -2, -3 <- 1, 2, 3 and x = w, w + 1, w + 2 <- w - 2,
        // w - 3, w - 4 (window byte of pixel x: x - X0 + 8).  One lane per window row; the rows themselves were mirrored when they were loaded.
        uint8_t *inb = (uint8_t *)in;
        if (lane < BT_IH) {
            uint8_t *rowb = inb + lane * BT_P;
            if (X0 == 0) { rowb[7] = rowb[9]; rowb[6] = rowb[10]; rowb[5] = rowb[11]; }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int xo = w + k - X0 + 8, xi = w - 2 - k - X0 + 8;
                if (xo >= 0 && xo < BT_P && xi >= 0 && xi < BT_P) rowb[xo] = rowb[xi];
            }
        }
    }
    __syncthreads();      // (unconditional: in k_octree_blur four waves with different tiles share the workgroup's barriers)
    const uint32_t k0 = g->taps[0], k1 = g->taps[1], k2 = g->taps[2], k3 = g->taps[3], k4 = g->taps[4], k5 = g->taps[5], k6 = g->taps[6];
    const int gq = lane & 15, rg = lane >> 4;
    const int x = X0 + 4 * gq;
    if (!store || x >= w || Y0 + 8 * rg >= h) return;
    // ---- horizontal: rows 8rg .. 8rg+13 of the window; pixel j of the group = bytes 1+j .. 7+j of (w0,w1,w2) ----
    // The TAPS are shifted, not the data: pixel j's seven taps sit at bytes 1+j .. 7+j of a 12-byte tap vector (uniform: scalar registers), so
    // a pixel is two or three v_dot4_u32_u8 on the window words as loaded - 10 per row of four pixels, where byte-aligning the window for every
    // pixel first took 6 v_alignbyte_b32 + 8 v_dot4.
    const uint32_t A0 = (k0 << 8) | (k1 << 16) | (k2 << 24), A1 = k3 | (k4 << 8) | (k5 << 16) | (k6 << 24);                      // j = 0: w0, w1
    const uint32_t B0 = (k0 << 16) | (k1 << 24), B1 = k2 | (k3 << 8) | (k4 << 16) | (k5 << 24), B2 = k6;                           // j = 1: w0, w1, w2
    const uint32_t C0 = k0 << 24, C1 = k1 | (k2 << 8) | (k3 << 16) | (k4 << 24), C2 = k5 | (k6 << 8);                              // j = 2: w0, w1, w2
    const uint32_t D1 = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24), D2 = k4 | (k5 << 8) | (k6 << 16);                                // j = 3: w1, w2
    // the 16-bit sums of two vertically adjacent rows (2m, 2m + 1) of a column share a word: exactly the operand pairs of the vertical pass below
    uint32_t hp[4][7];
#pragma unroll
    for (int m = 0; m < 7; m++) {
        uint32_t o[2][4];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const uint32_t *pw = in + (8 * rg + 2 * m + s) * (BT_P / 4) + gq + 1;
            const uint32_t w0 = pw[0], w1 = pw[1], w2 = pw[2];
            o[s][0] = __builtin_amdgcn_udot4(w1, A1, __builtin_amdgcn_udot4(w0, A0, 0u, false), false);
            o[s][1] = __builtin_amdgcn_udot4(w2, B2, __builtin_amdgcn_udot4(w1, B1, __builtin_amdgcn_udot4(w0, B0, 0u, false), false), false);
            o[s][2] = __builtin_amdgcn_udot4(w2, C2, __builtin_amdgcn_udot4(w1, C1, __builtin_amdgcn_udot4(w0, C0, 0u, false), false), false);
            o[s][3] = __builtin_amdgcn_udot4(w2, D2, __builtin_amdgcn_udot4(w1, D1, 0u, false), false);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) hp[j][m] = o[0][j] | (o[1][j] << 16);
    }
    // ---- vertical: output row q = 2t takes rows 2t .. 2t+6 = pairs t .. t+3 with the taps (k0,k1) (k2,k3) (k4,k5) (k6,0); row q = 2t+1 takes rows
    // 2t+1 .. 2t+7 = the SAME pairs with the taps (0,k0) (k1,k2) (k3,k4) (k5,k6): the taps move (scalar registers), not the data.  (Until round 5 every
    // output row had its own (h[q+2i], h[q+2i+1]) pairs, cut out of column-paired words by 14 v_perm_b32 per column: 56 of the kernel's ~500 instructions.)
    const uint32_t E0 = k0 | (k1 << 16), E1 = k2 | (k3 << 16), E2 = k4 | (k5 << 16), E3 = k6;
    const uint32_t O0 = k0 << 16, O1 = k1 | (k2 << 16), O2 = k3 | (k4 << 16), O3 = k5 | (k6 << 16);
    uint32_t outw[8];
#pragma unroll
    for (int q = 0; q < 8; q++) outw[q] = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        // (sum + 32768) >> 16 is byte 2 of the sum, and with taps that add up to at most 256 it cannot exceed 255 (65280 * 256 + 32768 <
        // 2^24): ONE v_perm_b32 drops that byte into byte j of the output word (shift, clamp and shift-or before).  Taps summing to 257
        // (the configuration allows them) keep the clamp.
        const uint32_t selo = j == 0 ? 0x03020106u : j == 1 ? 0x03020600u : j == 2 ? 0x03060100u : 0x06020100u;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int t = q >> 1;
            const uint32_t sum = (q & 1) ? udot2(hp[j][t + 3], O3, udot2(hp[j][t + 2], O2, udot2(hp[j][t + 1], O1, udot2(hp[j][t], O0, 32768u))))
                                         : udot2(hp[j][t + 3], E3, udot2(hp[j][t + 2], E2, udot2(hp[j][t + 1], E1, udot2(hp[j][t], E0, 32768u))));
            if (CLAMP) outw[q] |= min(sum >> 16, 255u) << (8 * j);
            else outw[q] = __builtin_amdgcn_perm(sum, outw[q], selo);
        }
    }
    uint8_t *dst = blur + (size_t)f * g->pyrBytes + lv.off;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int y = Y0 + 8 * rg + q;
        if (y >= h) break;
        *(uint32_t *)(dst + (size_t)y * lv.pitch + x) = outw[q];   // pitch >= round_up(w,4): bytes beyond w are padding
    }
}

template <bool CLAMP>
__global__ __launch_bounds__(64) void k_blur(const OrbxGeom *__restrict__ g, const uint8_t *__restrict__ img0, int img0Stride, size_t img0FramePitch,
                                             const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur, const int *__restrict__ lvlCnt, int *__restrict__ outBase)
{
    __shared__ __attribute__((aligned(16))) uint32_t in[BT_IH * (BT_P / 4)];
    XCD_REMAP_XY(bx, f);
    if (bx == 0 && threadIdx.x == 0 && outBase) {
        // rides along in the batch path (the quadtree precedes this kernel, the descriptor kernel follows it): where each level's keypoints start in the
        // frame's output list (level 0..n-1 in order, src/ORBextractor.cc:1577-1668), so that the 270k per-keypoint waves of k_orient_describe need two scalar
        // loads instead of a prefix sum each (its own per-workgroup prefix, used by the combined single-frame calls, costs a batch 3.5 % of that kernel)
        int run = 0;
        for (int i = 0; i < g->nlevels; i++) { outBase[f * g->nlevels + i] = run; run += lvlCnt[f * g->nlevels + i]; }
    }
    blur_body<CLAMP>(g, img0, img0Stride, img0FramePitch, pyr, blur, bx, f, in, (int)threadIdx.x, true);
}

// Combined single-frame calls: quadtree, host pyramid copy AND blur in one launch.  The quadtree is eight workgroups of dependent latency
// (~30 us for a 640x480 level 0) on a device that has nothing else to do; the blur needs nothing from it (the descriptor kernel takes the
// level prefix of the output list from the counts itself), so its tiles - four per workgroup, one wave each - run beside it instead of
// behind it: one kernel boundary and the blur's own ~7-13 us off the chain of every launch set.  Batches keep the separate launches
// (k_octree's 20 KB of LDS per workgroup would cap the blur's occupancy there).  blockIdx.x: [0, nlevels) quadtree, then the copy
// workgroups, then the blur workgroups; blockIdx.y = frame.
#define OB_THREADS 1024      /* sixteen waves: the quadtree's passes over a level's candidates (3000 at level 0 of a 1241x376 frame) take a quarter of the steps */
template <int NODECAP, bool CLAMP>
__global__ __launch_bounds__(OB_THREADS) void k_octree_blur(const OrbxGeom *__restrict__ g, const int *__restrict__ cellCount, const uint32_t *__restrict__ cellSlots,
                                                     const uint8_t *__restrict__ binTab, uint32_t *__restrict__ ptBuf, uint32_t *__restrict__ labBuf,
                                                     OrbxLevelKp *__restrict__ lvlKp, int *__restrict__ lvlCnt, int *__restrict__ status,
                                                     const OrbxCombMember *__restrict__ comb, const uint8_t *__restrict__ img0, int img0Stride, size_t img0FramePitch,
                                                     uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t blurLds[];      // OB_THREADS / 64 windows of BT_IH x BT_P bytes (the blur role)
    const int bx = (int)blockIdx.x, f = (int)blockIdx.y, nl = g->nlevels;
    if (bx < nl) { octree_body<NODECAP, OB_THREADS>(g, cellCount, cellSlots, binTab, ptBuf, labBuf, lvlKp, lvlCnt, status, bx, f, (int)gridDim.y); return; }
    if (bx < nl + ORBX_OCTREE_COPY_BLOCKS) { host_pyramid_copy(g, comb, pyr, f, bx - nl, ORBX_OCTREE_COPY_BLOCKS); return; }
    const int wv = (int)(threadIdx.x >> 6), tile = __builtin_amdgcn_readfirstlane((bx - nl - ORBX_OCTREE_COPY_BLOCKS) * (OB_THREADS / 64) + wv);      // (uniform in the wave: scalar bookkeeping)
    blur_body<CLAMP>(g, img0, img0Stride, img0FramePitch, pyr, blur, min(tile, g->blurTiles - 1), f, blurLds + wv * (BT_IH * (BT_P / 4)), (int)(threadIdx.x & 63), tile < g->blurTiles);
}

// ------------------------------------------------------------------------------------
// Orientation + descriptor in one pass over the keypoints (IC_Angle, src/ORBextractor.cc:108-161, then
// computeOrbDescriptor, :173-227, and the final KeyPoint, :1175-1190 / :1651-1660).  One wave per
// keypoint, four per workgroup.  Everything a keypoint needs from HBM is requested up front with
// row-wise dword loads whose addresses depend on the keypoint alone:
//   the 31 x 31 disc of the UNBLURRED level   lane = (disc row, half row): 16 bytes each, integer moments in registers;
//   the 37 x 37 patch of the BLURRED level    (rotated test points stay within 18.39 px: |coordinate| <= 18) as
//                                              37 rows x 10 aligned dwords -> LDS.
// The steered 512 samples are then LDS byte reads: the former kernel issued them as 8 scattered byte gathers per lane
// AFTER the angle was known (three dependent memory levels, up to 64 cache lines per instruction); now one memory
// round trip covers both stages and a load instruction touches ~7 lines.  fastAtan2 and the libm-exact sin / cos run
// once per workgroup on four lanes (one per keypoint) between two barriers instead of on all 64 lanes of every wave.
// ------------------------------------------------------------------------------------
// sum over the 64 lanes with data-parallel-primitive adds (one VALU instruction per step, no LDS crossbar): quad, half row, row, then the
// two row broadcasts of gfx9; the total lands in lane 63 and is returned as a scalar
__device__ __forceinline__ int wave_sum_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);     // row_mirror: every lane holds its row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);     // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);     // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// per-level constants of the geometry as a kernel ARGUMENT: the level lookup and the addresses come from a few wide scalar loads of the
// kernarg segment instead of a chain of dependent loads through OrbxGeom (the prologue is most of a wave's latency here)
struct OdLevels {
    int nlevels, kpPerFrame, outCap, pad;
    unsigned long long pyrBytes;
    int kpBase[ORBX_MAX_LEVELS], off[ORBX_MAX_LEVELS], pitch[ORBX_MAX_LEVELS], patch[ORBX_MAX_LEVELS];
    float scale[ORBX_MAX_LEVELS];
};

// Round 5, measured and NOT kept (profiles/r05_orient_describe_ablation.txt): (1) one wave per FOUR keypoints, no workgroup barriers, all forty pixel loads
// of the four in flight together, the lane's test pairs and disc mask in registers: 0.204 vs 0.207 ms alone, 275k vs 279k frames/s in the pipeline;
// (2) the level's keypoints visited in (128-byte column strip, row) order so that a wave's four patches share cache lines: no change.  The kernel is
// bound by its two pixel loads, not by arithmetic or its dependent chain: without the sample loop 0.200 ms, without the moments / angle 0.204, without
// the blurred-patch loads 0.114, without the disc load 0.122.  Its L2 misses (TCC_EA0_RDREQ 4.14 M x 128 B = 530 MB per 256 frames) are the two
// pyramids fetched ONCE (2 x 243 MB: the patches of ~1000 keypoints cover every 128-byte line of a frame), at the ~2.5 TB/s that scattered 128-byte
// lines reach here (TCC hit rate 81 %, L1 63 %); the byte-granular "algorithmic" 331 MB of SURVEY 8d is not reachable by any visiting order.
#define OD_WPB 4
#define OD_R 18
#define OD_DW 10
#define OD_ROWS (2 * OD_R + 1)
#define OD_PATCH_DW (OD_ROWS * OD_DW)

__global__ __launch_bounds__(64 * OD_WPB) void k_orient_describe(const OdLevels A, const uint8_t *__restrict__ img0, int img0Stride, size_t img0FramePitch,
                                                                 const uint8_t *__restrict__ pyr, const uint8_t *__restrict__ blur, OrbxLevelKp *__restrict__ lvlKp,
                                                                 const int *__restrict__ lvlCnt, const int *__restrict__ outBase, orbx_keypoint *__restrict__ outKp,
                                                                 uint8_t *__restrict__ outDesc, int *__restrict__ outCnt, const int *__restrict__ status,
                                                                 int *__restrict__ outStatus)
{
    __shared__ uint32_t sPatch[OD_WPB][OD_PATCH_DW + 2];
    __shared__ __attribute__((aligned(16))) float sPat[256][4];                  // test pair t as floats: x0, x1, y0, y1
    __shared__ __attribute__((aligned(16))) uint32_t sDisc[64][4];               // byte masks of the disc: lane (row, half) x 4 dwords
    __shared__ int sMom[OD_WPB][3];
    __shared__ int sBase[ORBX_MAX_LEVELS];
    __shared__ float sTrig[OD_WPB][3];
    XCD_REMAP_XY(bx, f);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // uniform: the keypoint bookkeeping below is scalar work
    if (threadIdx.x < 256) {     // once per workgroup: both constant tables into LDS
        *(float4 *)sPat[threadIdx.x] = *(const float4 *)c_od.pat[threadIdx.x];
        ((uint32_t *)sDisc)[threadIdx.x] = ((const uint32_t *)c_od.disc)[threadIdx.x];
    }
    const int *cnts = lvlCnt + f * A.nlevels;
    // where each level's keypoints start in the frame's output list (level 0 .. n-1 in order, src/ORBextractor.cc:1577-1668): once per workgroup,
    // lane = level, into LDS (k_blur's first workgroup used to leave it in global memory; the blur no longer has to run behind the quadtree for it)
    if (!outBase && threadIdx.x < (unsigned)A.nlevels) {      // (batches: k_blur's first workgroup left the prefix in outBase)
        int run = 0;
        for (int i = 0; i < (int)threadIdx.x; i++) run += cnts[i];
        sBase[threadIdx.x] = run;
    }
    if (bx == 0 && threadIdx.x == 0) {
        int tot = 0;
        for (int i = 0; i < A.nlevels; i++) tot += cnts[i];
        outCnt[f] = tot;
        // the capacity words travel with the results: the running batch's words are cleared by the NEXT batch, a consumer of this
        // result buffer reads the snapshot (which the producer only overwrites behind the consumer's event, like the results)
        outStatus[f] = status[f];
        if (f == 0) outStatus[gridDim.y] = status[gridDim.y];
    }
    // (Measured and not kept: a workgroup taking TWO groups of four slots in a row, the table copies and the prologue paid once per eight keypoints: 63.2 M
    // instead of 67.9 M vector instructions, but 0.230 instead of 0.207 ms alone and 271k instead of 282k frames/s - the kernel lives on the number of
    // workgroups that have their pixel loads in flight.)
    const int slot = bx * OD_WPB + wv;
    // Is the slot in use?  (Measured: requesting the keypoint record and the pixels BEFORE the counts are known - clamped coordinates for
    // the ~7 % unused slots - shortens the dependent chain by one level but is 7 % slower: 0.242 vs 0.226 ms per 256 frames.)
    bool inRange = slot < A.kpPerFrame;
    int l = 0, kb = 0;
#pragma unroll
    for (int i = 1; i < ORBX_MAX_LEVELS; i++) {      // (entries past the last level hold INT_MAX)
        const bool ge = slot >= A.kpBase[i];
        l += ge ? 1 : 0; kb = ge ? A.kpBase[i] : kb;
    }
    const int idx = slot - kb;
    inRange = inRange && idx < cnts[l];
    const bool live = inRange;
    const uint2 kraw = *(const uint2 *)&lvlKp[(size_t)f * A.kpPerFrame + (inRange ? slot : 0)];     // x | y << 16, score | pad: the first 8 bytes of OrbxLevelKp
    const int kx = (int)(kraw.x & 0xffffu), ky = (int)(kraw.x >> 16);
    const int ksc = (int)(kraw.y & 0xffu);
    const int xa = (kx - OD_R) & ~3;
    int m10 = 0, m01 = 0;
    uint2 pw[3] = {make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u)};
    uint4 wd = make_uint4(0u, 0u, 0u, 0u);
    const int r = lane >> 1, hf = lane & 1;      // disc row r, bytes x - 15 + 16 * half .. + 15 (19 <= x < w - 19: inside the level)
    if (inRange) {
        const int bp = A.pitch[l], up = l ? bp : img0Stride;
        const uint8_t *unb = l ? pyr + (size_t)f * A.pyrBytes + A.off[l] : img0 + (size_t)f * img0FramePitch;
        const uint8_t *bl = blur + (size_t)f * A.pyrBytes + A.off[l];
        if (r < 31) __builtin_memcpy(&wd, unb + (size_t)(ky + r - 15) * up + (kx - 15 + 16 * hf), 16);     // ONE 16-byte load per lane
        // patch: 37 rows x 5 aligned 8-byte units
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int item = lane + 64 * t;
            if (item < OD_ROWS * (OD_DW / 2)) {
                const int pr = (item * 205) >> 10, pc = item - (OD_DW / 2) * pr;      // item / 5
                __builtin_memcpy(&pw[t], bl + (size_t)(ky - OD_R + pr) * bp + xa + 8 * pc, 8);
            }
        }
    }
    __syncthreads();           // the tables are in LDS (the pixel loads are in flight meanwhile)
    if (inRange && r < 31) {
        // integer moments of the row segment: sum I and sum (u + 15) I as v_dot4_u32_u8 over the masked bytes, u + 15 = c (left half) / c + 16 (right half)
        const uint4 mk = *(const uint4 *)sDisc[lane];
        const uint32_t wb = hf ? 0x10101010u : 0u;
        uint32_t s1 = 0u, sw = 0u;
        s1 = __builtin_amdgcn_udot4(wd.x & mk.x, 0x01010101u, s1, false); sw = __builtin_amdgcn_udot4(wd.x & mk.x, 0x03020100u + wb, sw, false);
        s1 = __builtin_amdgcn_udot4(wd.y & mk.y, 0x01010101u, s1, false); sw = __builtin_amdgcn_udot4(wd.y & mk.y, 0x07060504u + wb, sw, false);
        s1 = __builtin_amdgcn_udot4(wd.z & mk.z, 0x01010101u, s1, false); sw = __builtin_amdgcn_udot4(wd.z & mk.z, 0x0b0a0908u + wb, sw, false);
        s1 = __builtin_amdgcn_udot4(wd.w & mk.w, 0x01010101u, s1, false); sw = __builtin_amdgcn_udot4(wd.w & mk.w, 0x0f0e0d0cu + wb, sw, false);
        m10 = (int)sw - 15 * (int)s1;
        m01 = (r - 15) * (int)s1;
    }
    m10 = wave_sum_dpp(m10); m01 = wave_sum_dpp(m01);        // totals as wave-uniform scalars
    if (lane == 0) { sMom[wv][0] = m01; sMom[wv][1] = m10; sMom[wv][2] = live ? slot : -1; }
    if (live) {
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int item = lane + 64 * t;
            if (item < OD_ROWS * (OD_DW / 2)) *(uint2 *)&sPatch[wv][2 * item] = pw[t];
        }
    }
    __syncthreads();
    if (threadIdx.x < OD_WPB && sMom[threadIdx.x][2] >= 0) {
        const float ang = fast_atan2_deg((float)sMom[threadIdx.x][0], (float)sMom[threadIdx.x][1]);
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        float sn, cs;
        sincosf_glibc(ang * factorPI, sn, cs);
        sTrig[threadIdx.x][0] = cs; sTrig[threadIdx.x][1] = sn; sTrig[threadIdx.x][2] = ang;
        OrbxLevelKp *o = lvlKp + (size_t)f * A.kpPerFrame + sMom[threadIdx.x][2];
        o->angle = ang; o->ca = cs; o->sb = sn;        // (read back by the stage taps and by nothing else)
    }
    __syncthreads();
    if (!live) return;
    const float a = sTrig[wv][0], b = sTrig[wv][1];
    const uint8_t *pb = (const uint8_t *)sPatch[wv] + OD_R * (4 * OD_DW) + (kx - xa);     // the keypoint's own pixel
    unsigned long long bits[4];
    typedef float f2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int rd = 0; rd < 4; rd++) {
        // both points of test pair 64 rd + lane at once (v_pk_mul_f32 / v_pk_add_f32: every product and sum is rounded on its own, exactly
        // like the scalar expressions of src/ORBextractor.cc:192-193 compiled without contraction)
        const float4 pt = *(const float4 *)sPat[64 * rd + lane];
        const f2_t xs = {pt.x, pt.y}, ys = {pt.z, pt.w};
        const f2_t rr = xs * b + ys * a, cc = xs * a - ys * b;
        // cvRound of both coordinates, then row * 40 + column: small integers, exact in float
        const int o0 = (int)__builtin_fmaf(__builtin_rintf(rr.x), (float)(4 * OD_DW), __builtin_rintf(cc.x));
        const int o1 = (int)__builtin_fmaf(__builtin_rintf(rr.y), (float)(4 * OD_DW), __builtin_rintf(cc.y));
        const int t0 = pb[o0], t1 = pb[o1];
        bits[rd] = __ballot(t0 < t1);
    }
    const int outIdx = (outBase ? outBase[f * A.nlevels + l] : sBase[l]) + idx;      // (< outCap: the output capacity is the sum of the levels' capacities)
    if (outIdx >= A.outCap) return;
    unsigned long long *d64 = (unsigned long long *)(outDesc + ((size_t)f * A.outCap + outIdx) * 32);
    if (lane < 4) d64[lane] = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
    if (lane == 0) {
        orbx_keypoint o;
        const float sc = A.scale[l];
        o.x = l ? (float)kx * sc : (float)kx;
        o.y = l ? (float)ky * sc : (float)ky;
        o.size = (float)A.patch[l]; o.angle = sTrig[wv][2]; o.response = (float)ksc; o.octave = l; o.class_id = -1;
        outKp[(size_t)f * A.outCap + outIdx] = o;
    }
}

// ------------------------------------------------------------------------------------
// Combined single-frame batches (orbx_extractor.hip: the combiner).  ORBextractor::operator() is a one-frame call (reference
// include/ORBextractor.h:110); when several callers - the two extractor threads of the stereo Frame constructor, src/Frame.cc:159-167,
// or the tracking threads of several sequences - are inside it at the same moment, their frames run as ONE launch set on a shared
// engine.  These two kernels are the set's first and last node: they move the frames in from, and the results out to, the members'
// own buffers named by a table in pinned memory (the engine's graph is the same whoever the members are).
// k_comb_upload: blockIdx.y = member; its frame (rows already at the device pitch) -> frame slot y of the engine's staging area, 16 bytes
// per lane, four loads in flight per lane.  The source is the member's pinned staging buffer - the reads cross PCIe: 14 us for one
// 640x480 frame, but 66 us for eight, with the whole chain waiting behind them - or, when the member had to wait for the engine anyway
// and uploaded its frame meanwhile on its own stream (DMA, overlapping the previous launch set), the member's device copy.
// ------------------------------------------------------------------------------------
// It also leaves a copy of the member's table entry in device memory for k_comb_finish (whose first act was a PCIe round trip for it).
__global__ __launch_bounds__(256) void k_comb_upload(const OrbxCombMember *__restrict__ tab, OrbxCombMember *__restrict__ tabCopy, uint8_t *__restrict__ staging, size_t framePitch)
{
    static_assert(sizeof(OrbxCombMember) % 8 == 0, "the table entry is copied in 8-byte words");
    if (blockIdx.x == 0 && threadIdx.x < sizeof(OrbxCombMember) / 8)
        ((unsigned long long *)&tabCopy[blockIdx.y])[threadIdx.x] = ((const unsigned long long *)&tab[blockIdx.y])[threadIdx.x];
    const uint4 *src = (const uint4 *)tab[blockIdx.y].srcImg;
    uint4 *dst = (uint4 *)(staging + (size_t)blockIdx.y * framePitch);
    const size_t n = framePitch >> 4, stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// k_comb_finish: blockIdx.y = member.  Frame y of the engine's result arena -> the member's arena in its one-frame layout, on the device
// (what SearchByBoW / ComputeStereoMatches chained behind the extractor read) and in pinned memory (what operator() converts to
// cv::KeyPoint / cv::Mat); the engine's pyramid and frame slot -> the member's device pyramid / level 0 (and, if asked for, the pinned
// pyramid copy behind the public mvImagePyramid).  Only the frame's real keypoints are moved, not the arena's capacity.
__global__ __launch_bounds__(256) void k_comb_finish(const OrbxCombMember *__restrict__ tab, const int *__restrict__ engCnt, const int *__restrict__ engSt,
                                                     const orbx_keypoint *__restrict__ engKp, const uint8_t *__restrict__ engDesc, int cap, const uint8_t *__restrict__ engPyr,
                                                     size_t pyrBytes, const uint8_t *__restrict__ engImg, size_t framePitch, size_t kpOff, size_t descOff, int hostPyrHere,
                                                     unsigned *sync, unsigned long long *hostFlag)
{
    const int f = blockIdx.y;
    const OrbxCombMember m = tab[f];
    const int cnt = min(engCnt[f], cap);
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    if (tid == 0) {
        const int st = engSt[f];      // the member's "batch" is its own frame: another member's capacity bits are not its business
        int *d = (int *)m.devArena, *h = (int *)m.hostOut;
        d[0] = cnt; d[1] = st; d[2] = st;
        h[0] = cnt; h[1] = st; h[2] = st;
    }
    {   // keypoints: 28-byte records, frame f starts at a multiple of 4 bytes only -> dwords
        const uint32_t *s = (const uint32_t *)(engKp + (size_t)f * cap);
        uint32_t *d = (uint32_t *)(m.devArena + kpOff), *h = (uint32_t *)(m.hostOut + kpOff);
        for (size_t i = tid; i < (size_t)cnt * 7; i += stride) { const uint32_t v = s[i]; d[i] = v; h[i] = v; }
    }
    {   // descriptors: 32 bytes each
        const uint4 *s = (const uint4 *)(engDesc + (size_t)f * cap * 32);
        uint4 *d = (uint4 *)(m.devArena + descOff), *h = (uint4 *)(m.hostOut + descOff);
        for (size_t i = tid; i < (size_t)cnt * 2; i += stride) { const uint4 v = s[i]; d[i] = v; h[i] = v; }
    }
    {   // pyramid levels >= 1 (device layout, padding included: one flat copy)
        const uint4 *s = (const uint4 *)(engPyr + (size_t)f * pyrBytes);
        uint4 *d = (uint4 *)m.devPyr, *h = hostPyrHere ? (uint4 *)m.hostPyr : nullptr;      // (k_pyramid_tiles stores the host copy itself where it runs)
        const size_t n = pyrBytes >> 4;
        if (h) for (size_t i = tid; i < n; i += stride) { const uint4 v = s[i]; d[i] = v; h[i] = v; }
        else for (size_t i = tid; i < n; i += stride) d[i] = s[i];
    }
    {   // level 0 = the frame itself
        const uint4 *s = (const uint4 *)(engImg + (size_t)f * framePitch);
        uint4 *d = (uint4 *)m.devImg;
        if (d) for (size_t i = tid; i < (framePitch >> 4); i += stride) d[i] = s[i];
    }
    if (hostFlag) {
        // completion of the launch set, for the leader that polls pinned memory instead of synchronising the stream (the graph's arguments are fixed:
        // the count of completed sets lives on the device): the last workgroup to arrive raises it behind everybody's stores
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (__hip_atomic_fetch_add(sync, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.y - 1) {
                __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long *cnt = (unsigned long long *)(sync + 2);
                const unsigned long long done = *cnt + 1;
                *cnt = done;
                __hip_atomic_store(hostFlag, done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Pipelined host batches: frames that live in pinned / registered host memory are read where they are (PCIe reads issued by the kernel, many rows
// in flight) and written at the device input layout; `tab` (pinned) holds their addresses.  T = uint4 when every frame and both strides are 16-byte
// aligned, uint32_t otherwise (the caller checked 4-byte alignment).
template <typename T>
__global__ __launch_bounds__(256) void k_gather_frames(const uint8_t *const *__restrict__ tab, int W, int H, int srcStride, uint8_t *__restrict__ dst, int dstStride, size_t framePitch)
{
    const uint8_t *src = tab[blockIdx.y];
    uint8_t *d = dst + (size_t)blockIdx.y * framePitch;
    const int upr = (W + (int)sizeof(T) - 1) / (int)sizeof(T), n = upr * H;      // units per row (the last one may run into the row padding: stride >= width rounded up to 4)
    for (int i = (int)(blockIdx.x * 256 + threadIdx.x); i < n; i += (int)(gridDim.x * 256)) {
        const int r = i / upr, c = i - r * upr;
        if (r + 1 < H || (c + 1) * (int)sizeof(T) <= W) {      // (a unit that runs past the pixels of the LAST row would leave the caller's buffer)
            const T v = *(const T *)(src + (size_t)r * srcStride + (size_t)c * sizeof(T));
            if ((c + 1) * (int)sizeof(T) <= dstStride) *(T *)(d + (size_t)r * dstStride + (size_t)c * sizeof(T)) = v;
            else for (int k = 0; k < dstStride - c * (int)sizeof(T); k++) d[(size_t)r * dstStride + (size_t)c * sizeof(T) + k] = ((const uint8_t *)&v)[k];
        } else {      // the last unit of the last row would read past the caller's buffer: bytes
            for (int k = 0; c * (int)sizeof(T) + k < W; k++) d[(size_t)r * dstStride + (size_t)c * sizeof(T) + k] = src[(size_t)r * srcStride + (size_t)c * sizeof(T) + k];
        }
    }
}

}  // namespace

int orbx_launch_gather_frames(hipStream_t stream, const uint8_t *const *tab, int batch, int W, int H, int srcStride, uint8_t *dst, int dstStride, size_t framePitch, bool aligned16)
{
    const int unit = aligned16 ? 16 : 4, n = (W + unit - 1) / unit * H;
    const dim3 grid((unsigned)std::min(std::max((n + 1023) / 1024, 1), 64), (unsigned)batch);      // four units per thread in flight
    if (aligned16) hipLaunchKernelGGL(k_gather_frames<uint4>, grid, dim3(256), 0, stream, tab, W, H, srcStride, dst, dstStride, framePitch);
    else hipLaunchKernelGGL(k_gather_frames<uint32_t>, grid, dim3(256), 0, stream, tab, W, H, srcStride, dst, dstStride, framePitch);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { orbx_set_error("k_gather_frames launch failed: %s", hipGetErrorString(e)); return ORBX_ERR_HIP; }
    return ORBX_OK;
}

// Every kernel of the extractor goes out through emit(): onto the stream (batches), or as a kernel node of a hipGraph under
// construction (the single-frame graph of orbx_extractor.hip: L.graph set; the node depends on L.deps[0..ndeps) and is returned
// in L.node).  The graph is built with the explicit node API, not by stream capture: a capture is invalidated by what OTHER
// threads do on the device meanwhile (two extractors on two threads is the reference's stereo constructor, src/Frame.cc:159-167).
#include <tuple>
#include <utility>

template <typename... KArgs, size_t... I>
static int emit_impl(const OrbxLaunch &L, void (*kern)(KArgs...), dim3 grid, dim3 block, size_t lds, std::tuple<KArgs...> &t, std::index_sequence<I...>)
{
    void *args[] = {(void *)&std::get<I>(t)...};
    if (L.graph) {
        hipKernelNodeParams p;
        memset(&p, 0, sizeof(p));
        p.func = (void *)kern; p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = (unsigned)lds; p.kernelParams = args; p.extra = nullptr;
        hipGraphNode_t node = nullptr;
        const hipError_t e = hipGraphAddKernelNode(&node, L.graph, L.ndeps ? L.deps : nullptr, (size_t)L.ndeps, &p);
        if (e != hipSuccess) { orbx_set_error("hipGraphAddKernelNode failed: %s", hipGetErrorString(e)); return ORBX_ERR_HIP; }
        *L.node = node;
        return ORBX_OK;
    }
    const hipError_t e = hipLaunchKernel((const void *)kern, grid, block, args, lds, L.stream);
    if (e != hipSuccess) { orbx_set_error("kernel launch failed: %s", hipGetErrorString(e)); return ORBX_ERR_HIP; }
    return ORBX_OK;
}

template <typename... KArgs, typename... Args>
static int emit(const OrbxLaunch &L, void (*kern)(KArgs...), dim3 grid, dim3 block, size_t lds, Args... args)
{
    std::tuple<KArgs...> t(static_cast<KArgs>(args)...);      // the kernel's exact parameter types, addressable
    return emit_impl(L, kern, grid, block, lds, t, std::index_sequence_for<KArgs...>());
}

int orbx_launch_resize(const OrbxLaunch &L, int level)
{
    const OrbxLevel &lv = L.geom->lv[level];
    const int items = ((lv.w + 3) >> 2) * ((lv.h + RS_ROWS - 1) / RS_ROWS);
    dim3 grid((unsigned)((items + 255) / 256), 1u, (unsigned)L.batch);
    // pyramid levels are allocated with >= 16 spare bytes per row; the caller's level-0 rows only when the stride says so
    const bool padded = level > 1 || (L.img0Stride >= ((L.geom->lv[0].w + 3) & ~3) + 12 && L.img0FramePitch >= (size_t)L.img0Stride * (size_t)L.geom->lv[0].h);
    if (padded) return emit(L, k_resize<true>, grid, dim3(256), 0, L.geomDev, level, L.img0, L.img0Stride, L.img0FramePitch, L.pyr, L.rsTab);
    return emit(L, k_resize<false>, grid, dim3(256), 0, L.geomDev, level, L.img0, L.img0Stride, L.img0FramePitch, L.pyr, L.rsTab);
}

int orbx_launch_pyramid_tiles(const OrbxLaunch &L)
{
    const size_t ldsBytes = 2 * (size_t)L.pyrTileBuf + (size_t)L.pyrTileTab;
    // the attribute belongs to the function, not to the handle: only ever raised (graphs of other handles keep their larger nodes valid);
    // callers build one graph at a time (build_single_graph's mutex)
    static size_t ldsGranted[64];      // per device (0 = nothing beyond the default 48 KB yet)
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t &granted = ldsGranted[dev & 63];
    if (ldsBytes > std::max<size_t>(granted, 48 * 1024)) {
        const hipError_t e = hipFuncSetAttribute((const void *)k_pyramid_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e != hipSuccess) { orbx_set_error("hipFuncSetAttribute(k_pyramid_tiles) failed: %s", hipGetErrorString(e)); return ORBX_ERR_HIP; }
        granted = ldsBytes;
    }
    return emit(L, k_pyramid_tiles, dim3((unsigned)L.pyrTileCount, (unsigned)L.batch), dim3(256), ldsBytes, L.geomDev, L.img0, L.img0Stride, L.img0FramePitch, L.pyr, L.rsTab,
                L.pyrTiles, L.pyrTileBuf, L.status);
}

int orbx_launch_comb_upload(const OrbxLaunch &L, uint8_t *stagingDev)
{
    const size_t units = L.img0FramePitch >> 4;
    // one 16-byte unit per thread (four until round 6: 19 workgroups per 640x480 frame kept too few reads in flight across PCIe - 17 us for 307 KB)
    const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>((units + 255) / 256, 1), 256);
    return emit(L, k_comb_upload, dim3(blocks, (unsigned)L.batch), dim3(256), 0, L.combTab, L.combTabCopy, stagingDev, L.img0FramePitch);
}

int orbx_launch_comb_finish(const OrbxLaunch &L)
{
    const size_t units = (L.geom->pyrBytes + L.img0FramePitch) >> 4;
    // (every workgroup ends with an agent-scope release - a write-back of its XCD's L2 - before it counts itself in: few, fat workgroups)
    const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>((units + 2047) / 2048, 1), 64);
    return emit(L, k_comb_finish, dim3(blocks, (unsigned)L.batch), dim3(256), 0, (const OrbxCombMember *)L.combTabCopy, L.outCnt, L.outStatus, L.outKp, L.outDesc, L.geom->outCap, L.pyr, L.geom->pyrBytes, L.img0,
                L.img0FramePitch, L.combKpOff, L.combDescOff, 0, L.combSync, L.combFlag);      // (the host pyramid copy rides in the quadtree's launch)
}

// developer tap (tools/fast_phases.py; not part of include/orbx.h): a device array of 32 u64 that the PROF instantiation of k_fast_cells adds
// its per-phase shader cycles to; NULL = the product kernel
static unsigned long long *g_fcProf = nullptr;
extern "C" void orbx_debug_fast_cells_profile(unsigned long long *dev32) { g_fcProf = dev32; }

int orbx_launch_fast_cells(const OrbxLaunch &L)
{
    const OrbxGeom &g = *L.geom;
    static const int kEnv = getenv("ORBX_FC_CELLS_PER_WAVE") ? atoi(getenv("ORBX_FC_CELLS_PER_WAVE")) : 0;      // (developer knob)
    // batches: four cells per wave (the next cell's window is fetched under the current cell's phases); single frames and the combiner's small launch
    // sets: one - 815 waves per 640x480 frame do not fill the device, and four cells in a row are four times a cell's latency (111 vs 100 us per call)
    const int K = kEnv > 0 ? kEnv : (L.batch >= 32 ? ORBX_FC_CELLS_PER_WAVE : 1);
    dim3 grid((unsigned)((g.cellsPerFrame + K - 1) / K), (unsigned)L.batch);
    const size_t ldsBytes = (size_t)g.fcLdsBytes;
#define FC_ARGS L.fcCells, K, g.cellsPerFrame, g.slotsPerFrame, g.pyrBytes, g.iniTh, g.minTh, g.fcInBytes, g.fcScBytes, L.img0, L.img0Stride, L.img0FramePitch, L.pyr, L.score, L.cellCount, L.cellSlots
#define FC_LAUNCH(PP, SS, NN) do { if (L.score) return emit(L, k_fast_cells<PP, SS, NN, false, true>, grid, dim3(64), ldsBytes, FC_ARGS, (unsigned long long *)nullptr); \
    if (g_fcProf) return emit(L, k_fast_cells<PP, SS, NN, true, false>, grid, dim3(64), ldsBytes, FC_ARGS, g_fcProf); \
    return emit(L, k_fast_cells<PP, SS, NN, false, false>, grid, dim3(64), ldsBytes, FC_ARGS, (unsigned long long *)nullptr); } while (0)
#define FC_LAUNCH_P(PP, SS) switch (g.fcNS) { case 2: FC_LAUNCH(PP, SS, 2); case 3: FC_LAUNCH(PP, SS, 3); case 4: FC_LAUNCH(PP, SS, 4); default: FC_LAUNCH(PP, SS, 6); }
    // (build_geometry guarantees: score pitch 80 only with window pitch 80, and score pitch 48 only when every area is at most 46 wide)
    if (g.fcPitch == 48 && g.fcScPitch == 48) FC_LAUNCH_P(48, 48)
    if (g.fcPitch == 64 && g.fcScPitch == 48) FC_LAUNCH_P(64, 48)
    if (g.fcPitch == 80 && g.fcScPitch == 48) FC_LAUNCH_P(80, 48)
    if (g.fcPitch != 80 || g.fcScPitch != 80) { orbx_set_error("detector LDS pitches %d / %d have no kernel instantiation", g.fcPitch, g.fcScPitch); return ORBX_ERR_STATE; }
    FC_LAUNCH_P(80, 80)
#undef FC_LAUNCH_P
#undef FC_LAUNCH
#undef FC_ARGS
}

int orbx_launch_octree(const OrbxLaunch &L)
{
    // (combined single-frame calls: 48 more workgroups per frame carry the host pyramid copy, see the kernel)
    dim3 grid((unsigned)(L.geom->nlevels + (L.combTab ? ORBX_OCTREE_COPY_BLOCKS : 0)), (unsigned)L.batch);
#define OT_LAUNCH(NC) return emit(L, k_octree<NC>, grid, dim3(256), 0, L.geomDev, L.cellCount, L.cellSlots, L.binTab, L.ptBuf, L.labBuf, L.lvlKp, L.lvlCnt, L.status, L.combTab, \
                                  (const uint8_t *)L.pyr)
    if (L.nodeCap <= 256) OT_LAUNCH(256);   // 1000 features at 640x480: 224 nodes at most; a fraction of the LDS, more resident quadtrees per CU
    if (L.nodeCap <= 512) OT_LAUNCH(512);
    if (L.nodeCap <= 1024) OT_LAUNCH(1024);
    OT_LAUNCH(2048);
#undef OT_LAUNCH
}

// combined single-frame calls: quadtree + host pyramid copy + blur as ONE launch (k_octree_blur); *fused = false where that kernel does not
// apply (large per-level quotas: the quadtree's LDS leaves no room for the blur windows) and the caller launches the two stages separately
int orbx_launch_octree_blur(const OrbxLaunch &L, bool *fused)
{
    *fused = false;
    if (!L.combTab || L.nodeCap > 512) return ORBX_OK;
    unsigned tapSum = 0;
    for (int i = 0; i < 7; i++) tapSum += L.geom->taps[i];
    const int tpb = OB_THREADS / 64;      // blur tiles per workgroup
    const dim3 grid((unsigned)(L.geom->nlevels + ORBX_OCTREE_COPY_BLOCKS + (L.geom->blurTiles + tpb - 1) / tpb), (unsigned)L.batch);
    const size_t lds = (size_t)tpb * BT_IH * BT_P;
    {   // (static quadtree arrays + the blur windows: above the default 64 KB of dynamic + static LDS per workgroup)
        static bool granted[2][2];
        const int a = L.nodeCap <= 256 ? 0 : 1;
        unsigned ts = 0;
        for (int i = 0; i < 7; i++) ts += L.geom->taps[i];
        const int b = ts > 256u ? 1 : 0;
        if (!granted[a][b]) {
            const void *fn = a == 0 ? (b ? (const void *)k_octree_blur<256, true> : (const void *)k_octree_blur<256, false>) : (b ? (const void *)k_octree_blur<512, true> : (const void *)k_octree_blur<512, false>);
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { orbx_set_error("hipFuncSetAttribute(k_octree_blur) failed: %s", hipGetErrorString(e)); return ORBX_ERR_HIP; }
            granted[a][b] = true;
        }
    }
    *fused = true;
#define OB_LAUNCH(NC, CL) return emit(L, k_octree_blur<NC, CL>, grid, dim3(OB_THREADS), lds, L.geomDev, L.cellCount, L.cellSlots, L.binTab, L.ptBuf, L.labBuf, L.lvlKp, L.lvlCnt, L.status, L.combTab, \
                                      L.img0, L.img0Stride, L.img0FramePitch, L.pyr, L.blur)
    if (L.nodeCap <= 256) { if (tapSum > 256u) OB_LAUNCH(256, true); OB_LAUNCH(256, false); }
    if (tapSum > 256u) OB_LAUNCH(512, true);
    OB_LAUNCH(512, false);
#undef OB_LAUNCH
}

int orbx_launch_blur(const OrbxLaunch &L)
{
    dim3 grid((unsigned)L.geom->blurTiles, (unsigned)L.batch);
    unsigned tapSum = 0;
    for (int i = 0; i < 7; i++) tapSum += L.geom->taps[i];
    if (tapSum > 256u) return emit(L, k_blur<true>, grid, dim3(64), 0, L.geomDev, L.img0, L.img0Stride, L.img0FramePitch, L.pyr, L.blur, L.lvlCnt, L.outBase);
    return emit(L, k_blur<false>, grid, dim3(64), 0, L.geomDev, L.img0, L.img0Stride, L.img0FramePitch, L.pyr, L.blur, L.lvlCnt, L.outBase);
}

int orbx_launch_orient_describe(const OrbxLaunch &L)
{
    dim3 grid((unsigned)((L.geom->kpPerFrame + OD_WPB - 1) / OD_WPB), (unsigned)L.batch);
    const OrbxGeom &g = *L.geom;
    OdLevels A;
    memset(&A, 0, sizeof(A));
    A.nlevels = g.nlevels; A.kpPerFrame = g.kpPerFrame; A.outCap = g.outCap; A.pyrBytes = g.pyrBytes;
    for (int l = 0; l < ORBX_MAX_LEVELS; l++) A.kpBase[l] = 0x7fffffff;
    for (int l = 0; l < g.nlevels; l++) { A.kpBase[l] = g.lv[l].kpBase; A.off[l] = g.lv[l].off; A.pitch[l] = g.lv[l].pitch; A.patch[l] = g.lv[l].patchSize; A.scale[l] = g.lv[l].scale; }
    for (int i = 0; i < 16; i++)
        if (g.umax[i] != kUmax[i]) { orbx_set_error("disc half-widths differ from the compiled table"); return ORBX_ERR_STATE; }
    return emit(L, k_orient_describe, grid, dim3(64 * OD_WPB), 0, A, L.img0, L.img0Stride, L.img0FramePitch, L.pyr, L.blur, L.lvlKp, L.lvlCnt, L.combTab ? (const int *)nullptr : (const int *)L.outBase, L.outKp, L.outDesc,
                L.outCnt, L.status, L.outStatus);
}
