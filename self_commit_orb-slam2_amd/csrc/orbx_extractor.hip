// orbx_extractor.hip -- host side of the extractor C ABI (include/orbx.h).
//
// Mirrors ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:92-161,
// src/ORBextractor.cc:492-609, 1544-1734): the constructor tables are built on the
// host with the reference's float/double arithmetic, the per-frame work runs as the
// HIP kernels of orbx_kernels.hip on the handle's stream.  No CPU fallback.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "orbx_internal.h"

static thread_local char g_err[512] = "";

void orbx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *orbx_last_error(void) { return g_err; }
extern "C" int orbx_version(void) { return 100; }

static const char *kStageNames[] = {"pyramid", "fast_cells", "octree", "orient", "blur", "describe"};
#define ORBX_PROF_RING 64
enum { ST_PYR = 0, ST_FAST, ST_OCTREE, ST_ORIENT, ST_BLUR, ST_DESC, ST_COUNT };
extern "C" const char *orbx_stage_name(int s) { return (s >= 0 && s < ST_COUNT) ? kStageNames[s] : ""; }

namespace {

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int ensure(size_t count)
    {
        if (count <= n) return ORBX_OK;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        ORBX_HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
        n = count;
        return ORBX_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

inline int round_f(float v) { return (int)lrintf(v); }
inline int floor_d(double v) { int i = (int)v; return i - (i > v); }
inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct orbx_extractor {
    orbx_extractor_config cfg;
    // ORBextractor tables (src/ORBextractor.cc:499-554)
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> quota;
    int umax[16];
    uint32_t taps[7];
    // current geometry
    OrbxGeom geom;
    bool geomValid = false;
    std::vector<uint8_t> binHost;
    std::vector<uint32_t> rsHost;   // cv::resize tables of all levels (build_resize_tables)
    std::vector<OrbxFcCell> fcHost; // k_fast_cells: one entry per cell of a frame
    // k_pyramid_tiles (single-frame call): per (tile, level) rectangles, planned from rsHost when the single-frame graph is built
    OrbxDevBuf<OrbxPyrTile> ptDev;
    int ptBatchW = 0, ptBatchH = 0;              // geometry the plan below was made for by run_batch (ORBX_BATCH_PYR_TILES)
    int ptTiles = 0, ptLds = 0, ptTab = 0;       // 0 tiles: no plan (one level, taps wider than 8 bytes, LDS budget) -> the per-level launches
    int nodeCap = 512;
    // device state
    hipStream_t stream = nullptr;
    // profiling: a ring of event sets so that every batch call of a timed region keeps its
    // own events and nothing has to be read (= synchronised) inside the region
    hipEvent_t ev[ORBX_PROF_RING][ST_COUNT + 1] = {};
    bool profiling = false;
    int profCount = 0;
    bool debugTaps = false;   // keep the FAST score map for orbx_debug_download_scores
    DevBuf<OrbxGeom> geomDev;
    DevBuf<uint8_t> binDev, pyr, blur, score, staging;
    DevBuf<uint32_t> rsDev;
    DevBuf<OrbxFcCell> fcDev;
    DevBuf<int> cellCount, lvlCnt, lvlBase, status;
    // results are double buffered: a consumer (matcher) may still read batch i while batch i+1 is
    // extracted; consumerEv[b] = event after which buffer b may be overwritten again
    // counts | capacity words | keypoints | descriptors of a buffer live in ONE allocation, so that a whole-batch read-back is one copy
    DevBuf<uint8_t> outArena[2];
    int *outCntP[2] = {nullptr, nullptr};
    int *outStP[2] = {nullptr, nullptr};      // [allocBatch + 1]: per frame, then (at [batch of the call]) the OR over the batch
    orbx_keypoint *outKpP[2] = {nullptr, nullptr};
    uint8_t *outDescP[2] = {nullptr, nullptr};
    size_t arenaKpOff = 0, arenaDescOff = 0, arenaBytes = 0;
    hipEvent_t consumerEv[2] = {nullptr, nullptr};
    hipEvent_t pyrConsumerEv = nullptr;   // a consumer still reads the (single buffered) pyramid of the last batch
    int cur = 0;
    DevBuf<uint32_t> cellSlots, ptBuf, labBuf;
    DevBuf<OrbxLevelKp> lvlKp;
    int allocBatch = 0;
    // last run
    int lastBatch = 0;
    bool lastChunked = false;         // the last call was a host batch pipelined in chunks: the device holds its LAST chunk only, lastBatch = 0 (no device-side view)
    const uint8_t *lastImg0 = nullptr;
    int lastStride = 0;
    size_t lastFramePitch = 0;
    uint8_t *hostStaging = nullptr;   // pinned: rows packed at the device layout when hipMemcpy2D would crawl
    size_t hostStagingBytes = 0;
    int stagingStride = 0;
    size_t stagingFramePitch = 0;
    uint8_t *hostOut = nullptr;       // pinned: results / pyramid on their way to the caller's arrays
    size_t hostOutBytes = 0;
    // The single-frame host call (ORBextractor::operator()) as ONE hipGraph per result buffer: upload from the pinned staging
    // buffer, the 11 kernels, the read-back of the result arena into pinned memory.  A frame is then one hipGraphLaunch + one
    // hipStreamSynchronize instead of ~16 runtime calls (each 5-10 us of host time under the runtime's lock: they, not the
    // kernels, bounded the call - and serialised extractors on different threads).
    hipGraph_t sgGraph[2] = {nullptr, nullptr};
    hipGraphExec_t sgExec[2] = {nullptr, nullptr};
    bool sgValid = false;
    bool sgDisabled = false;          // ORBX_NO_GRAPH=1, or graph construction failed once: plain stream launches
    // combined single-frame calls (the combiner below): the shared engine set of this handle's (device, configuration, image size)
    struct Combiner *comb = nullptr;
    bool combDisabled = false;        // ORBX_COMBINE=0, or the engines could not be built: the handle's own graph
    orbx_extractor *expectPartner = nullptr;   // one-shot hint: the next call's batch should wait (briefly) for this handle's call
    bool isEngine = false;
    size_t hostPyrOff = 0;            // pinned copy of the pyramid (levels >= 1) inside hostOut, behind the result arena
    bool hostPyrValid = false;        // ... and whether the last call filled it
    bool hostSynced = false;          // the last call was a synchronous single-frame call: complete when it returned, capacity word in hostOut[1]
    bool lastCombined = false;        // the last call ran on a shared engine: this handle's `blur` buffer (a parity tap) was not written
    // Pipelined host batches (orbx_extract_batch_begin / _end, and orbx_extract_batch itself in chunks): two slots, each with its own pinned
    // input, device input and pinned result buffer; uploads on upStream, read-backs on downStream, kernels on `stream`.
    struct PipeSlot {
        uint8_t *hostIn = nullptr; size_t hostInBytes = 0;
        DevBuf<uint8_t> devIn;
        uint8_t *hostRes = nullptr; size_t hostResBytes = 0;
        const uint8_t **ptrTab = nullptr; size_t ptrTabBytes = 0;      // pinned: addresses of a batch's frames for k_gather_frames
        hipEvent_t evUp = nullptr, evKern = nullptr, evDown = nullptr;
        int batch = 0, cap = 0;          // frames and per-frame capacity of the batch the slot holds (the handle's geometry may change before its _end)
        size_t offKp = 0, offDesc = 0, offSt = 0;
    } pipe[2];
    int pipeHead = 0, pipeCount = 0;  // oldest begun slot, number of begun and not yet ended batches (<= 2)
    double pipeUs[4] = {0, 0, 0, 0};  // ORBX_PIPE_STATS=1: host microseconds in staging + enqueue, the launch set's enqueue, the wait for the read-back, the copy-out
    long pipeCalls = 0;
    hipStream_t upStream = nullptr, downStream = nullptr;
};

static bool plan_pyramid_tiles(orbx_extractor *h, std::vector<OrbxPyrTile> &out, int &tiles, int &bufBytes, int &tabBytes);      // (below; run_batch plans too)
namespace {

void invalidate_single_graph(orbx_extractor *h);

// ORBextractor::ORBextractor, src/ORBextractor.cc:492-609
void build_tables(orbx_extractor *h)
{
    const int nl = h->cfg.nlevels;
    const double scaleFactor = h->cfg.scale_factor;   // double member initialised from a float (ORBextractor.h:206)
    h->scale.assign((size_t)nl, 1.f); h->sigma2.assign((size_t)nl, 1.f);
    h->invScale.assign((size_t)nl, 1.f); h->invSigma2.assign((size_t)nl, 1.f);
    h->quota.assign((size_t)nl, 0);
    for (int i = 1; i < nl; i++) {
        h->scale[(size_t)i] = (float)(h->scale[(size_t)i - 1] * scaleFactor);
        h->sigma2[(size_t)i] = h->scale[(size_t)i] * h->scale[(size_t)i];
    }
    for (int i = 0; i < nl; i++) {
        h->invScale[(size_t)i] = 1.0f / h->scale[(size_t)i];
        h->invSigma2[(size_t)i] = 1.0f / h->sigma2[(size_t)i];
    }
    float factor = (float)(1.0f / scaleFactor);
    float nDesired = h->cfg.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        h->quota[(size_t)l] = round_f(nDesired);
        sum += h->quota[(size_t)l];
        nDesired *= factor;
    }
    h->quota[(size_t)nl - 1] = std::max(h->cfg.nfeatures - sum, 0);
    // umax: quarter circle of radius HALF_PATCH_SIZE made symmetric (:579-608)
    int v, v0;
    const int vmax = floor_d(ORBX_HALF_PATCH * sqrt(2.f) / 2 + 1);
    const int vmin = (int)ceil(ORBX_HALF_PATCH * sqrt(2.f) / 2);
    const double hp2 = ORBX_HALF_PATCH * ORBX_HALF_PATCH;
    for (v = 0; v <= vmax; ++v) h->umax[v] = (int)lrint(sqrt(hp2 - v * v));
    for (v = ORBX_HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (h->umax[v0] == h->umax[v0 + 1]) ++v0;
        h->umax[v] = v0;
        ++v0;
    }
    static const uint16_t kDefaultTaps[7] = {18, 34, 48, 56, 48, 34, 18};
    bool zero = true;
    for (int i = 0; i < 7; i++) zero = zero && h->cfg.gauss_taps[i] == 0;
    for (int i = 0; i < 7; i++) h->taps[i] = zero ? kDefaultTaps[i] : h->cfg.gauss_taps[i];
}

// Level sizes (ComputePyramid :1680-1682), cell grid (:1064-1086), initial quadtree nodes
// (:719-748), cv::resize coefficient tables, tile tables and buffer offsets for W x H.
// OpenCV resize.cpp (INTER_LINEAR, 8-bit): source index and the two 11-bit coefficients of destination
// index d.  fx = (float)((d+0.5)*scale - 0.5) with the product and the difference in double, floor,
// fraction in float, cvRound((1-fx)*2048) / cvRound(fx*2048), saturate_cast<short>.  Columns reset the
// coefficients at both borders; rows only clamp the indices.  (IEEE arithmetic, no contraction.)
static void resize_coef_host(int d, double scale, int slen, bool isX, int &s0, int &c0, int &c1)
{
    float fx = (float)(((double)d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (isX) {
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= slen - 1) { fx = 0.f; sx = slen - 1; }
    }
    s0 = sx;
    c0 = std::min(std::max((int)lrintf((1.f - fx) * 2048.f), -32768), 32767);
    c1 = std::min(std::max((int)lrintf(fx * 2048.f), -32768), 32767);
}

// Tables k_resize reads instead of redoing that arithmetic in every thread.
//   columns: 12 u32 per group of 4 destination columns:
//     [0] first source column sx0   [1] largest tap offset from sx0 (span)
//     [2..5] v_perm selectors of the 4 columns (tap bytes off, off1 inside the 8-byte window; valid when span <= 7)
//     [6..9] coefficient pairs a0 | a1 << 16      [10] the four `off` as bytes   [11] the four `off1` as bytes
//   rows: 2 u32 per destination row: clamped source rows y0 | y1 << 16, coefficients b0 | b1 << 16
static void build_resize_tables(std::vector<uint32_t> &tab, OrbxLevel &lv, int sw, int sh)
{
    while (tab.size() % 4) tab.push_back(0u);     // 16-byte aligned groups
    lv.rsColOff = (int)tab.size();
    const int G = (lv.w + 3) >> 2;
    for (int gi = 0; gi < G; gi++) {
        uint32_t e[12] = {0};
        int sx0 = 0, span = 0;
        for (int k = 0; k < 4; k++) {
            const int dx = std::min(4 * gi + k, lv.w - 1);
            int sx, a0, a1;
            resize_coef_host(dx, lv.rsScaleX, sw, true, sx, a0, a1);
            const int sx1 = sx + 1 < sw ? sx + 1 : sx;
            if (k == 0) sx0 = sx;
            const int off = sx - sx0, off1 = sx1 - sx0;
            span = std::max(span, off1);
            e[2 + k] = 0x0c000c00u | (uint32_t)(off & 7) | ((uint32_t)(off1 & 7) << 16);
            e[6 + k] = (uint32_t)(uint16_t)a0 | ((uint32_t)(uint16_t)a1 << 16);
            e[10] |= (uint32_t)(off & 0xff) << (8 * k);
            e[11] |= (uint32_t)(off1 & 0xff) << (8 * k);
        }
        e[0] = (uint32_t)sx0; e[1] = (uint32_t)span;
        tab.insert(tab.end(), e, e + 12);
    }
    lv.rsRowOff = (int)tab.size();
    for (int dy = 0; dy < lv.h; dy++) {
        int sy, b0, b1;
        resize_coef_host(dy, lv.rsScaleY, sh, false, sy, b0, b1);
        const int y0 = std::min(std::max(sy, 0), sh - 1), y1 = std::min(std::max(sy + 1, 0), sh - 1);
        tab.push_back((uint32_t)y0 | ((uint32_t)y1 << 16));
        tab.push_back((uint32_t)(uint16_t)b0 | ((uint32_t)(uint16_t)b1 << 16));
    }
}

int build_geometry(orbx_extractor *h, int W, int H)
{
    OrbxGeom &g = h->geom;
    memset(&g, 0, sizeof(g));
    const int nl = h->cfg.nlevels;
    g.nlevels = nl; g.W = W; g.H = H; g.iniTh = h->cfg.ini_th_fast; g.minTh = h->cfg.min_th_fast;
    for (int i = 0; i < 7; i++) g.taps[i] = h->taps[i];
    for (int i = 0; i < 16; i++) g.umax[i] = h->umax[i];
    h->binHost.clear();
    h->rsHost.clear();
    size_t off = 0;
    int cells = 0, slots = 0, kps = 0, btiles = 0, maxNodes = 0, maxWCell = 0, maxHCell = 0;
    for (int l = 0; l < nl; l++) {
        OrbxLevel &lv = g.lv[l];
        lv.w = round_f((float)W * h->invScale[(size_t)l]);
        lv.h = round_f((float)H * h->invScale[(size_t)l]);
        if (lv.w < 2 * ORBX_BORDER + ORBX_CELL_W || lv.h < 2 * ORBX_BORDER + ORBX_CELL_W) {
            orbx_set_error("image %dx%d too small for pyramid level %d (%dx%d)", W, H, l, lv.w, lv.h);
            return ORBX_ERR_ARG;
        }
        if (lv.w > 4095 + ORBX_BORDER || lv.h > 4095 + ORBX_BORDER) { orbx_set_error("image too large (max 4111 px per side)"); return ORBX_ERR_ARG; }
        lv.pitch = (int)align_up((size_t)lv.w + 16, 64);   // >= 16 readable bytes behind every row (k_resize reads aligned 12-byte windows)
        lv.off = (int)off;
        off += align_up((size_t)lv.pitch * lv.h, 256);
        const int minB = ORBX_BORDER, maxBX = lv.w - ORBX_BORDER, maxBY = lv.h - ORBX_BORDER;
        const float width = (float)(maxBX - minB), height = (float)(maxBY - minB);
        lv.nCols = (int)(width / (float)ORBX_CELL_W);
        lv.nRows = (int)(height / (float)ORBX_CELL_W);
        lv.wCell = (int)ceil(width / lv.nCols);
        lv.hCell = (int)ceil(height / lv.nRows);
        lv.cellBase = cells;
        cells += lv.nCols * lv.nRows;
        lv.cellCap = ((lv.wCell + 1) / 2) * ((lv.hCell + 1) / 2);
        lv.slotBase = slots;
        slots += lv.nCols * lv.nRows * lv.cellCap;
        lv.quota = h->quota[(size_t)l];
        lv.nIni = (int)roundf((float)(maxBX - minB) / (maxBY - minB));
        if (lv.nIni < 1 || lv.nIni > ORBX_MAX_INI) {      // (0: a window more than twice as high as wide - the reference indexes an empty vector there, :766)
            orbx_set_error("aspect ratio of level %d (%dx%d) gives %d initial quadtree nodes; supported: 1..%d", l, lv.w, lv.h, lv.nIni, ORBX_MAX_INI);
            return ORBX_ERR_ARG;
        }
        const float hX = (float)(maxBX - minB) / lv.nIni;
        for (int i = 0; i <= lv.nIni; i++) lv.iniX[i] = (int)(hX * (float)i);
        lv.binOff = (int)h->binHost.size();
        for (int x = 0; x < maxBX - minB; x++) {
            int b = (int)((float)x / hX);   // vpIniNodes[kp.pt.x/hX], :766
            h->binHost.push_back((uint8_t)std::min(b, lv.nIni - 1));
        }
        lv.kpCap = lv.quota + 3 + 4 * lv.nIni;
        lv.kpBase = kps;
        kps += lv.kpCap;
        maxNodes = std::max(maxNodes, lv.kpCap);
        maxWCell = std::max(maxWCell, lv.wCell);
        maxHCell = std::max(maxHCell, lv.hCell);
        lv.blurTilesX = (lv.w + 63) / 64;
        lv.blurTilesY = (lv.h + 31) / 32;
        lv.blurTileBase = btiles;
        btiles += lv.blurTilesX * lv.blurTilesY;
        lv.scale = h->scale[(size_t)l];
        lv.patchSize = (int)(ORBX_PATCH * h->scale[(size_t)l]);
        // cv::resize(INTER_LINEAR) from level l-1: OpenCV's index / coefficient arithmetic is tabulated here
        lv.rsScaleX = lv.rsScaleY = 1.0;
        lv.rsColOff = lv.rsRowOff = 0;
        if (l > 0) {
            lv.rsScaleX = 1. / ((double)lv.w / g.lv[l - 1].w);
            lv.rsScaleY = 1. / ((double)lv.h / g.lv[l - 1].h);
            build_resize_tables(h->rsHost, lv, g.lv[l - 1].w, g.lv[l - 1].h);
        }
    }
    g.cellsPerFrame = cells; g.slotsPerFrame = slots; g.kpPerFrame = kps; g.outCap = kps;
    g.blurTiles = btiles;
    // LDS carve-up of k_fast_cells (one wave per cell): row pitch 16*segments+16 bytes for both tiles
    if (maxWCell > 64 || maxHCell > 63) { orbx_set_error("cell %dx%d larger than the detector supports", maxWCell, maxHCell); return ORBX_ERR_ARG; }
    // window rows: the widest window (aw + 6 rounded up to 16) and the widest pre-test read (16 * segments + 8); score rows: aw + 6
    g.fcPitch = maxWCell <= 32 ? 48 : maxWCell <= 48 ? 64 : 80;
    g.fcScPitch = maxWCell <= 42 ? 48 : 80;
    {
        // The kernel is instantiated on (window pitch, score pitch) = (48, 48), (64, 48), (80, 48) or (80, 80).  A score row of pitch SP holds the area's
        // columns -1 .. aw (a zero ring), so SP = 48 needs aw + 2 <= 48 for EVERY cell - not "wCell <= 42": wide cells exist whose area is narrower (a level of
        // one column of cells).  The widest area of this geometry decides (round-5 advice: (64, 80) used to run the SP = 48 instantiation on the strength
        // of that unstated fact).
        int maxAw = 1;
        for (int l = 0; l < nl; l++) {
            const OrbxLevel &lv = g.lv[l];
            const int maxBX = lv.w - ORBX_BORDER;
            for (int cj = 0; cj < lv.nCols; cj++) {
                const int iniX = ORBX_BORDER + cj * lv.wCell, maxX = std::min(iniX + lv.wCell + 6, maxBX);
                maxAw = std::max(maxAw, (maxX - 3) - (iniX + 3));
            }
        }
        g.fcScPitch = maxAw + 2 <= 48 ? 48 : 80;
        if (g.fcScPitch == 80) g.fcPitch = 80;      // (there is no (48 | 64, 80) instantiation)
    }
    { const char *pe = getenv("ORBX_FC_PITCH"); if (pe && atoi(pe) == 80 && g.fcPitch == 64) g.fcPitch = 80; }      // (developer knob)
    {
        const int units = (maxHCell + 6) * ((maxWCell + 6 + 15) / 16), ns = (units + 63) / 64;      // 16-byte window units per lane
        g.fcNS = ns <= 2 ? 2 : ns <= 3 ? 3 : ns <= 4 ? 4 : 6;
    }
    g.fcInBytes = g.fcPitch * (maxHCell + 6);
    g.fcScBytes = g.fcScPitch * (maxHCell + 2);
    g.fcLdsBytes = g.fcInBytes + g.fcScBytes + (int)align_up((size_t)maxWCell * maxHCell * 2, 16);
    // the detector's cell table (ComputeKeyPointsOctTree's cell loop, src/ORBextractor.cc:1089-1123)
    h->fcHost.assign((size_t)cells, OrbxFcCell{});
    for (int l = 0; l < nl; l++) {
        const OrbxLevel &lv = g.lv[l];
        const int maxBX = lv.w - ORBX_BORDER, maxBY = lv.h - ORBX_BORDER;
        for (int ci = 0; ci < lv.nRows; ci++)
            for (int cj = 0; cj < lv.nCols; cj++) {
                const int cell = ci * lv.nCols + cj;
                OrbxFcCell &e = h->fcHost[(size_t)lv.cellBase + cell];
                const int iniX = ORBX_BORDER + cj * lv.wCell, iniY = ORBX_BORDER + ci * lv.hCell;
                const int maxX = std::min(iniX + lv.wCell + 6, maxBX), maxY = std::min(iniY + lv.hCell + 6, maxBY);
                const int x0 = iniX + 3, x1 = maxX - 3, y0 = iniY + 3, y1 = maxY - 3, aw = x1 - x0, ah = y1 - y0;
                const bool valid = !(iniY >= maxBY - 3 || iniX >= maxBX - 6 || aw <= 0 || ah <= 0);      // :1101, :1112
                // (a skipped cell stands in as the 1 x 1 area at the level's first detectable pixel: the kernel's unconditional window loads stay valid)
                e.xy = valid ? ((uint32_t)x0 | ((uint32_t)y0 << 16)) : ((uint32_t)ORBX_EDGE | ((uint32_t)ORBX_EDGE << 16));
                const int ew = valid ? aw : 1, eh = valid ? ah : 1, nu = (ew + 6 + 15) / 16;
                e.dim = (uint32_t)ew | ((uint32_t)eh << 8) | ((uint32_t)l << 16) | (valid ? 1u << 24 : 0u);
                e.inv = (uint32_t)((65536 + nu - 1) / nu); e.units = (uint32_t)nu | ((uint32_t)((eh + 6) * nu) << 8);
                e.pitch = lv.pitch; e.off = (uint32_t)lv.off;
                e.slot = (uint32_t)(lv.slotBase + cell * lv.cellCap); e.cap = (uint32_t)lv.cellCap;
            }
    }
    g.pyrBytes = align_up(off + 256, 256);
    if (maxNodes > 2048) { orbx_set_error("per-level feature quota %d exceeds the quadtree node capacity 2048", maxNodes); return ORBX_ERR_ARG; }
    h->nodeCap = maxNodes <= 256 ? 256 : (maxNodes <= 512 ? 512 : (maxNodes <= 1024 ? 1024 : 2048));
    // k_octree's careful rounds sort nodes by a 32-bit key (candidate count << log2 nodeCap | position): the count of one node - at most
    // every candidate slot of its level - has to fit the remaining bits (2^21 at 2048 nodes: ~8.4 Mpixel levels; 2^24 at 256)
    const long long keyLimit = 1ll << (32 - (h->nodeCap == 256 ? 8 : h->nodeCap == 512 ? 9 : h->nodeCap == 1024 ? 10 : 11));
    for (int l = 0; l < nl; l++)
        if ((long long)g.lv[l].nCols * g.lv[l].nRows * g.lv[l].cellCap >= keyLimit) {
            orbx_set_error("level %d (%dx%d) can hold %lld FAST candidates, the quadtree's sort key %lld at %d nodes per level", l, g.lv[l].w, g.lv[l].h,
                           (long long)g.lv[l].nCols * g.lv[l].nRows * g.lv[l].cellCap, keyLimit - 1, h->nodeCap);
            return ORBX_ERR_ARG;
        }
    return ORBX_OK;
}

int ensure_geometry(orbx_extractor *h, int W, int H, int batch)
{
    if (W <= 0 || H <= 0 || batch <= 0) { orbx_set_error("bad image size / batch"); return ORBX_ERR_ARG; }
    if (W > h->cfg.max_width || H > h->cfg.max_height || batch > h->cfg.max_batch) {
        orbx_set_error("request %dx%d x%d exceeds the handle's configured maximum %dx%d x%d", W, H, batch, h->cfg.max_width,
                       h->cfg.max_height, h->cfg.max_batch);
        return ORBX_ERR_CAPACITY;
    }
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    if (!h->geomValid || h->geom.W != W || h->geom.H != H) {
        h->geomValid = false;
        invalidate_single_graph(h);
        int rc = build_geometry(h, W, H);
        if (rc != ORBX_OK) return rc;
        if ((rc = h->geomDev.ensure(1)) != ORBX_OK) return rc;
        if ((rc = h->binDev.ensure(std::max<size_t>(h->binHost.size(), 1))) != ORBX_OK) return rc;
        if ((rc = h->rsDev.ensure(std::max<size_t>(h->rsHost.size(), 4))) != ORBX_OK) return rc;
        if ((rc = h->fcDev.ensure(std::max<size_t>(h->fcHost.size(), 1))) != ORBX_OK) return rc;
        ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
        ORBX_HIP_CHECK(hipMemcpy(h->geomDev.p, &h->geom, sizeof(OrbxGeom), hipMemcpyHostToDevice));
        ORBX_HIP_CHECK(hipMemcpy(h->binDev.p, h->binHost.data(), h->binHost.size(), hipMemcpyHostToDevice));
        ORBX_HIP_CHECK(hipMemcpy(h->rsDev.p, h->rsHost.data(), h->rsHost.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        ORBX_HIP_CHECK(hipMemcpy(h->fcDev.p, h->fcHost.data(), h->fcHost.size() * sizeof(OrbxFcCell), hipMemcpyHostToDevice));
        h->geomValid = true;
        h->allocBatch = 0;   // per-frame sizes changed: re-check every buffer
    }
    if (batch > h->allocBatch) {
        invalidate_single_graph(h);       // buffers move
        const OrbxGeom &g = h->geom;
        const size_t B = (size_t)batch;
        int rc;
        if ((rc = h->pyr.ensure(B * g.pyrBytes)) != ORBX_OK) return rc;
        if ((rc = h->blur.ensure(B * g.pyrBytes)) != ORBX_OK) return rc;
        if (h->debugTaps && (rc = h->score.ensure(B * g.pyrBytes)) != ORBX_OK) return rc;
        if ((rc = h->cellCount.ensure(B * g.cellsPerFrame)) != ORBX_OK) return rc;
        if ((rc = h->cellSlots.ensure(B * g.slotsPerFrame)) != ORBX_OK) return rc;
        if ((rc = h->ptBuf.ensure(B * g.slotsPerFrame)) != ORBX_OK) return rc;     // every candidate the detector can emit: no capacity error
        if ((rc = h->labBuf.ensure(B * g.slotsPerFrame)) != ORBX_OK) return rc;
        if ((rc = h->lvlKp.ensure(B * g.kpPerFrame)) != ORBX_OK) return rc;
        if ((rc = h->lvlCnt.ensure(B * g.nlevels)) != ORBX_OK) return rc;
        if ((rc = h->lvlBase.ensure(B * g.nlevels)) != ORBX_OK) return rc;
        h->arenaKpOff = align_up((2 * B + 1) * sizeof(int), 256);
        h->arenaDescOff = h->arenaKpOff + align_up(B * g.outCap * sizeof(orbx_keypoint), 256);
        h->arenaBytes = h->arenaDescOff + B * g.outCap * 32;
        for (int b = 0; b < 2; b++) {
            if ((rc = h->outArena[b].ensure(h->arenaBytes)) != ORBX_OK) return rc;
            h->outCntP[b] = (int *)h->outArena[b].p;
            h->outStP[b] = h->outCntP[b] + B;
            h->outKpP[b] = (orbx_keypoint *)(h->outArena[b].p + h->arenaKpOff);
            h->outDescP[b] = h->outArena[b].p + h->arenaDescOff;
        }
        if ((rc = h->status.ensure(B + 1)) != ORBX_OK) return rc;   // per frame + one word for the whole batch
        // the score map is only written inside the detectable window; clear it once so the
        // parity taps (orbx_debug_download_scores) see zeros elsewhere
        if (h->debugTaps) ORBX_HIP_CHECK(hipMemsetAsync(h->score.p, 0, B * g.pyrBytes, h->stream));
        h->allocBatch = batch;
    }
    return ORBX_OK;
}

void fill_launch(orbx_extractor *h, OrbxLaunch &L, const uint8_t *img0Dev, int batch, int stride, size_t framePitch, int cb)
{
    L.stream = h->stream; L.geomDev = h->geomDev.p; L.geom = &h->geom; L.batch = batch;
    L.img0 = img0Dev; L.img0Stride = stride; L.img0FramePitch = framePitch;
    L.pyr = h->pyr.p; L.blur = h->blur.p; L.score = h->debugTaps ? h->score.p : nullptr; L.blurBytes = h->geom.pyrBytes;
    L.binTab = h->binDev.p;
    L.rsTab = h->rsDev.p;
    L.fcCells = h->fcDev.p;
    L.cellCount = h->cellCount.p; L.cellSlots = h->cellSlots.p; L.ptBuf = h->ptBuf.p; L.labBuf = h->labBuf.p;
    L.lvlKp = h->lvlKp.p; L.lvlCnt = h->lvlCnt.p; L.outBase = h->lvlBase.p; L.outKp = h->outKpP[cb]; L.outDesc = h->outDescP[cb]; L.outCnt = h->outCntP[cb];
    L.status = h->status.p; L.outStatus = h->outStP[cb]; L.nodeCap = h->nodeCap;
}

void invalidate_single_graph(orbx_extractor *h)
{
    for (int b = 0; b < 2; b++) {
        if (h->sgExec[b]) { (void)hipGraphExecDestroy(h->sgExec[b]); h->sgExec[b] = nullptr; }
        if (h->sgGraph[b]) { (void)hipGraphDestroy(h->sgGraph[b]); h->sgGraph[b] = nullptr; }
    }
    h->sgValid = false;
}

int run_batch(orbx_extractor *h, const uint8_t *img0Dev, int batch, int W, int H, int stride, size_t framePitch)
{
    int rc = ensure_geometry(h, W, H, batch);
    if (rc != ORBX_OK) return rc;
    OrbxLaunch L;
    h->cur ^= 1;
    const int cb = h->cur;
    fill_launch(h, L, img0Dev, batch, stride, framePitch, cb);
    const bool prof = h->profiling;
    hipEvent_t *ev = h->ev[h->profCount % ORBX_PROF_RING];
    if (h->pyrConsumerEv) { ORBX_HIP_CHECK(hipStreamWaitEvent(h->stream, h->pyrConsumerEv, 0)); h->pyrConsumerEv = nullptr; }
    ORBX_HIP_CHECK(hipMemsetAsync(h->status.p, 0, (size_t)(batch + 1) * sizeof(int), h->stream));
    if (prof) ORBX_HIP_CHECK(hipEventRecord(ev[0], h->stream));
    static const bool batchTiles = getenv("ORBX_BATCH_PYR_TILES") && getenv("ORBX_BATCH_PYR_TILES")[0] == '1';
    bool tiled = false;
    if (batchTiles && h->allocBatch > 1) {      // (measurement switch: the one-launch pyramid of the single-frame paths for a whole batch)
        if (h->ptBatchW != W || h->ptBatchH != H) {
            std::vector<OrbxPyrTile> plan;
            h->ptTiles = 0;
            if (plan_pyramid_tiles(h, plan, h->ptTiles, h->ptLds, h->ptTab)) {
                if ((rc = h->ptDev.ensure(plan.size())) != ORBX_OK) return rc;
                ORBX_HIP_CHECK(hipMemcpy(h->ptDev.p, plan.data(), plan.size() * sizeof(OrbxPyrTile), hipMemcpyHostToDevice));
            } else h->ptTiles = 0;
            h->ptBatchW = W; h->ptBatchH = H;
        }
        if (h->ptTiles > 0) {
            L.pyrTiles = h->ptDev.p; L.pyrTileCount = h->ptTiles; L.pyrTileBuf = h->ptLds; L.pyrTileTab = h->ptTab;
            if ((rc = orbx_launch_pyramid_tiles(L)) != ORBX_OK) return rc;
            tiled = true;
        }
    }
    if (!tiled)
        for (int l = 1; l < h->geom.nlevels; l++)
            if ((rc = orbx_launch_resize(L, l)) != ORBX_OK) return rc;
    if (prof) ORBX_HIP_CHECK(hipEventRecord(ev[ST_PYR + 1], h->stream));
    if ((rc = orbx_launch_fast_cells(L)) != ORBX_OK) return rc;   // FAST score + cell NMS fused; score map only for the parity taps
    if (prof) ORBX_HIP_CHECK(hipEventRecord(ev[ST_FAST + 1], h->stream));
    if ((rc = orbx_launch_octree(L)) != ORBX_OK) return rc;
    if (prof) ORBX_HIP_CHECK(hipEventRecord(ev[ST_OCTREE + 1], h->stream));
    if (prof) ORBX_HIP_CHECK(hipEventRecord(ev[ST_ORIENT + 1], h->stream));      // (orientation is part of the descriptor kernel: this span is empty)
    if ((rc = orbx_launch_blur(L)) != ORBX_OK) return rc;
    if (prof) ORBX_HIP_CHECK(hipEventRecord(ev[ST_BLUR + 1], h->stream));
    if (h->consumerEv[cb]) { ORBX_HIP_CHECK(hipStreamWaitEvent(h->stream, h->consumerEv[cb], 0)); h->consumerEv[cb] = nullptr; }
    if ((rc = orbx_launch_orient_describe(L)) != ORBX_OK) return rc;
    if (prof) { ORBX_HIP_CHECK(hipEventRecord(ev[ST_DESC + 1], h->stream)); h->profCount++; }
    h->lastBatch = batch; h->lastImg0 = img0Dev; h->lastStride = stride; h->lastFramePitch = framePitch;
    h->lastCombined = false; h->hostSynced = false; h->lastChunked = false;
    return ORBX_OK;
}

// host-side copies of a batch entry point (frames into pinned staging, results out of the pinned arena) on a few threads
static int host_copy_threads()
{
    static const int n = [] {
        const char *e = getenv("ORBX_HOST_COPY_THREADS");
        int v = e && *e ? atoi(e) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 2));
        return v < 1 ? 1 : v > 64 ? 64 : v;
    }();
    return n;
}
// fn(begin, end) over [0, n) in nthreads contiguous slices on the process's copy pool (the caller takes the first slice).  The workers are created
// once: a chunked host batch stages and copies out eight times per call, and eight thread creations per slice set cost more than the copies.
struct CopyPool {
    std::mutex m;                       // one job at a time (callers of different handles queue here; a job is a fraction of a millisecond)
    std::mutex jm;
    std::condition_variable cvWork, cvDone;
    std::vector<std::thread> workers;
    std::function<void(int, int)> job;
    int n = 0, parts = 0, gen = 0, pending = 0;
    bool stop = false;
    void worker(int id)
    {
        int seen = 0;
        for (;;) {
            std::function<void(int, int)> fn;
            int n_, parts_;
            {
                std::unique_lock<std::mutex> lk(jm);
                cvWork.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen; fn = job; n_ = n; parts_ = parts;
            }
            if (id < parts_) fn((int)((long long)n_ * id / parts_), (int)((long long)n_ * (id + 1) / parts_));
            {
                std::lock_guard<std::mutex> lk(jm);
                if (--pending == 0) cvDone.notify_all();
            }
        }
    }
    void run(int n_, int nthreads, const std::function<void(int, int)> &fn)
    {
        nthreads = std::max(1, std::min(nthreads, n_));
        if (nthreads == 1) { fn(0, n_); return; }
        std::lock_guard<std::mutex> one(m);
        {
            std::lock_guard<std::mutex> lk(jm);
            while ((int)workers.size() < nthreads - 1) { const int id = (int)workers.size() + 1; workers.emplace_back([this, id] { worker(id); }); }
            job = fn; n = n_; parts = nthreads; pending = (int)workers.size(); gen++;
        }
        cvWork.notify_all();
        fn(0, (int)((long long)n_ / nthreads));
        std::unique_lock<std::mutex> lk(jm);
        cvDone.wait(lk, [&] { return pending == 0; });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(jm); stop = true; }
        cvWork.notify_all();
        for (auto &t : workers) t.join();
    }
};
static CopyPool &copy_pool() { static CopyPool *p = new CopyPool; return *p; }      // (never destroyed: handles may outlive static destruction order)
template <typename F> static void parallel_slices(int n, int nthreads, F fn) { copy_pool().run(n, nthreads, std::function<void(int, int)>(fn)); }

int upload(orbx_extractor *h, const uint8_t *const *images, int batch, int W, int H, int stride)
{
    if (!images || stride < W) { orbx_set_error("bad image pointer / stride"); return ORBX_ERR_ARG; }
    const int dstStride = (int)align_up((size_t)W + 16, 64);   // same row padding as the pyramid levels
    const size_t fp = align_up((size_t)dstStride * H + 256, 256);
    int rc = h->staging.ensure(fp * (size_t)batch);
    if (rc != ORBX_OK) return rc;
    for (int f = 0; f < batch; f++)
        if (!images[f]) { orbx_set_error("image %d is NULL", f); return ORBX_ERR_ARG; }
    const int nth = batch >= 16 ? host_copy_threads() : 1;
    if ((W & 3) || (stride & 3) || nth > 1) {
        // hipMemcpy2D from pageable memory takes a row-by-row path for widths that are not a multiple of 4
        // (measured: 2.8 ms for one 1241x376 frame against 0.03 ms for 640x480): lay the rows out at the
        // device pitch in a pinned buffer and move them from there.  Batches of 16 frames and more do that on several host threads
        // (ORBX_HOST_COPY_THREADS, default min(8, cores)), every thread sending its slice as soon as it is staged: 256 pageable
        // hipMemcpy2DAsync calls move 75 MB at 17 GB/s (the runtime stages them one by one), one thread's memcpy is no faster.
        const size_t bytes = fp * (size_t)batch;
        ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));   // the previous batch's copy has left the pinned buffer
        if (bytes > h->hostStagingBytes) {
            if (h->hostStaging) (void)hipHostFree(h->hostStaging);
            h->hostStaging = nullptr; h->hostStagingBytes = 0;
            ORBX_HIP_CHECK(hipHostMalloc((void **)&h->hostStaging, bytes, hipHostMallocDefault));
            h->hostStagingBytes = bytes;
        }
        std::atomic<int> failed{0};
        const int dev = h->cfg.device;
        parallel_slices(batch, nth, [&](int f0, int f1) {
            for (int f = f0; f < f1; f++) {
                uint8_t *dst = h->hostStaging + fp * (size_t)f;
                if (stride == dstStride) memcpy(dst, images[f], (size_t)dstStride * (size_t)(H - 1) + (size_t)W);
                else for (int y = 0; y < H; y++) memcpy(dst + (size_t)y * dstStride, images[f] + (size_t)y * stride, (size_t)W);
            }
            if (f1 > f0 && (hipSetDevice(dev) != hipSuccess ||
                            hipMemcpyAsync(h->staging.p + fp * (size_t)f0, h->hostStaging + fp * (size_t)f0, fp * (size_t)(f1 - f0), hipMemcpyHostToDevice, h->stream) != hipSuccess))
                failed.store(1);
        });
        if (failed.load()) { orbx_set_error("upload of the batch failed: %s", hipGetErrorString(hipGetLastError())); return ORBX_ERR_HIP; }
    } else {
        for (int f = 0; f < batch; f++)
            ORBX_HIP_CHECK(hipMemcpy2DAsync(h->staging.p + fp * (size_t)f, (size_t)dstStride, images[f], (size_t)stride, (size_t)W, (size_t)H,
                                            hipMemcpyHostToDevice, h->stream));
    }
    h->stagingStride = dstStride; h->stagingFramePitch = fp;
    return ORBX_OK;
}

}  // namespace

static int no_batch_error(const orbx_extractor *h)
{
    if (h->lastChunked)
        orbx_set_error("the last call was a host batch pipelined in chunks (orbx_extract_batch with batch >= 2 x ORBX_HOST_BATCH_CHUNK): its results went to the caller's "
                       "arrays and the device keeps the last chunk only - for device-side results use orbx_upload_frames + orbx_extract_batch_device, or ORBX_HOST_BATCH_CHUNK=0");
    else orbx_set_error("no batch has been extracted yet");
    return ORBX_ERR_STATE;
}

hipStream_t orbx_extractor_stream_internal(orbx_extractor *h) { return h ? h->stream : nullptr; }
/* true: the handle's last call was a synchronous single-frame call - its results are complete (no event to wait for) and *status is its
 * capacity word, read from the pinned result arena */
bool orbx_extractor_host_complete_internal(orbx_extractor *h, int *status)
{
    if (!h || !h->hostSynced || !h->lastBatch || !h->hostOut) return false;
    if (status) *status = ((const int *)h->hostOut)[1];
    return true;
}
int orbx_extractor_host_count_internal(orbx_extractor *h) { return h && h->hostOut ? ((const int *)h->hostOut)[0] : 0; }
void orbx_extractor_set_consumer_event_internal(orbx_extractor *h, hipEvent_t ev) { if (h) h->consumerEv[h->cur] = ev; }
void orbx_extractor_set_pyramid_consumer_event_internal(orbx_extractor *h, hipEvent_t ev) { if (h) h->pyrConsumerEv = ev; }
int orbx_extractor_last_batch_view_internal(orbx_extractor *h, OrbxLastBatchView *v)
{
    if (!h || !v) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!h->lastBatch) return no_batch_error(h);
    v->batch = h->lastBatch; v->nlevels = h->geom.nlevels; v->cap = h->geom.outCap;
    v->kp = h->outKpP[h->cur]; v->desc = h->outDescP[h->cur]; v->counts = h->outCntP[h->cur];
    v->img0 = h->lastImg0; v->img0Stride = h->lastStride; v->img0FramePitch = h->lastFramePitch;
    v->pyr = h->pyr.p; v->pyrBytes = h->geom.pyrBytes; v->geomDev = h->geomDev.p; v->geom = &h->geom;
    v->scale = h->scale.data(); v->invScale = h->invScale.data();
    return ORBX_OK;
}

// "<uuid>@<pci domain:bus:device.function>" of HIP device `device` (what tells two GPUs apart; bench.py's ranks all-gather it), and the NUMA
// node its PCI function hangs on (-1: unknown).  Touches the HIP runtime only through hipGetDeviceProperties: no stream, no context of another
// library is created (a torch.cuda query before the timed region moved bench.py's own streams to other hardware queues: -13 % throughput).
extern "C" int orbx_device_identity(int device, char *identity, int capacity, int *numa_node)
{
    if (!identity || capacity < 64) { orbx_set_error("identity buffer too small (need 64 bytes)"); return ORBX_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { orbx_set_error("no HIP device available"); return ORBX_ERR_NODEVICE; }
    if (device < 0 || device >= ndev) { orbx_set_error("device %d out of range (have %d)", device, ndev); return ORBX_ERR_ARG; }
    hipDeviceProp_t p;
    ORBX_HIP_CHECK(hipGetDeviceProperties(&p, device));
    char uuid[40];
    static const char *hex = "0123456789abcdef";
    for (int i = 0; i < 16; i++) { uuid[2 * i] = hex[((unsigned char)p.uuid.bytes[i]) >> 4]; uuid[2 * i + 1] = hex[((unsigned char)p.uuid.bytes[i]) & 15]; }
    uuid[32] = 0;
    char bus[32];
    snprintf(bus, sizeof(bus), "%04x:%02x:%02x.0", p.pciDomainID, p.pciBusID, p.pciDeviceID);
    snprintf(identity, (size_t)capacity, "%s@%s", uuid, bus);
    if (numa_node) {
        *numa_node = -1;
        char path[96];
        snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
        if (FILE *f = fopen(path, "r")) { int v = -1; if (fscanf(f, "%d", &v) == 1) *numa_node = v; fclose(f); }
    }
    return ORBX_OK;
}

extern "C" int orbx_extractor_create(const orbx_extractor_config *cfg, orbx_extractor **out)
{
    if (!cfg || !out) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    *out = nullptr;
    if (cfg->nlevels < 1 || cfg->nlevels > ORBX_MAX_LEVELS || cfg->nfeatures < 1 || !(cfg->scale_factor > 1.0f) || cfg->min_th_fast < 1 ||
        cfg->ini_th_fast < cfg->min_th_fast || cfg->ini_th_fast > 255 || cfg->max_width < 1 || cfg->max_height < 1 || cfg->max_batch < 1) {
        orbx_set_error("bad extractor configuration (need 1<=nlevels<=%d, nfeatures>=1, scale_factor>1, 1<=minTh<=iniTh<=255)", ORBX_MAX_LEVELS);
        return ORBX_ERR_ARG;
    }
    {   // 8.8 fixed-point Gaussian taps: bytes, and a sum <= 257 keeps the 16-bit horizontal pass from saturating
        unsigned sum = 0;
        bool ok = true;
        for (int i = 0; i < 7; i++) { sum += cfg->gauss_taps[i]; ok = ok && cfg->gauss_taps[i] <= 255; }
        if (sum != 0 && (!ok || sum > 257)) { orbx_set_error("gauss_taps must be <= 255 each and sum to <= 257 (got sum %u)", sum); return ORBX_ERR_ARG; }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        orbx_set_error("no HIP device available: liborbx has no CPU fallback");
        return ORBX_ERR_NODEVICE;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { orbx_set_error("device %d out of range (have %d)", cfg->device, ndev); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(cfg->device));
    orbx_extractor *h = new orbx_extractor();
    h->cfg = *cfg;
    { const char *ng = getenv("ORBX_NO_GRAPH"); h->sgDisabled = ng && ng[0] == '1'; }
    { const char *nc = getenv("ORBX_COMBINE"); h->combDisabled = nc && nc[0] == '0'; }
    build_tables(h);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        orbx_set_error("hipStreamCreate failed");
        delete h;
        return ORBX_ERR_HIP;
    }
    for (int r = 0; r < ORBX_PROF_RING; r++)
        for (int i = 0; i <= ST_COUNT; i++) (void)hipEventCreate(&h->ev[r][i]);
    // validate the largest geometry up front so a bad size fails at construction
    int rc = build_geometry(h, cfg->max_width, cfg->max_height);
    if (rc != ORBX_OK) { orbx_extractor_destroy(h); return rc; }
    h->geomValid = false;
    *out = h;
    return ORBX_OK;
}

static void comb_detach(orbx_extractor *h);      // (the combiner, below)

extern "C" void orbx_extractor_destroy(orbx_extractor *h)
{
    if (!h) return;
    if (h->comb && !h->isEngine) comb_detach(h);      // the last handle of an engine set takes the set with it
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    invalidate_single_graph(h);
    h->geomDev.release(); h->binDev.release(); h->rsDev.release(); h->fcDev.release(); h->ptDev.release(); h->pyr.release(); h->blur.release();
    if (h->hostStaging) { (void)hipHostFree(h->hostStaging); h->hostStaging = nullptr; h->hostStagingBytes = 0; }
    if (h->hostOut) { (void)hipHostFree(h->hostOut); h->hostOut = nullptr; h->hostOutBytes = 0; }
    if (h->pipeCalls && getenv("ORBX_PIPE_STATS"))
        fprintf(stderr, "[orbx] host batch pipeline, %ld batches: staging + upload enqueue %.0f us, launch set + read-back enqueue %.0f us, wait for the read-back %.0f us, copy-out %.0f us (means)\n",
                h->pipeCalls, h->pipeUs[0] / h->pipeCalls, h->pipeUs[1] / h->pipeCalls, h->pipeUs[2] / h->pipeCalls, h->pipeUs[3] / h->pipeCalls);
    if (h->upStream) { (void)hipStreamSynchronize(h->upStream); (void)hipStreamDestroy(h->upStream); }
    if (h->downStream) { (void)hipStreamSynchronize(h->downStream); (void)hipStreamDestroy(h->downStream); }
    for (int i = 0; i < 2; i++) {
        orbx_extractor::PipeSlot &S = h->pipe[i];
        if (S.hostIn) (void)hipHostFree(S.hostIn);
        if (S.hostRes) (void)hipHostFree(S.hostRes);
        if (S.ptrTab) (void)hipHostFree(S.ptrTab);
        S.devIn.release();
        if (S.evUp) (void)hipEventDestroy(S.evUp);
        if (S.evKern) (void)hipEventDestroy(S.evKern);
        if (S.evDown) (void)hipEventDestroy(S.evDown);
    }
    h->score.release(); h->staging.release(); h->cellCount.release(); h->lvlCnt.release(); h->lvlBase.release();
    for (int b = 0; b < 2; b++) h->outArena[b].release();
    h->status.release(); h->cellSlots.release(); h->ptBuf.release(); h->labBuf.release(); h->lvlKp.release();
    for (int r = 0; r < ORBX_PROF_RING; r++)
        for (int i = 0; i <= ST_COUNT; i++) if (h->ev[r][i]) (void)hipEventDestroy(h->ev[r][i]);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int orbx_extractor_tables(const orbx_extractor *h, int *nlevels, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                                     int *features_per_level)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    const int nl = h->cfg.nlevels;
    if (nlevels) *nlevels = nl;
    for (int i = 0; i < nl; i++) {
        if (scale) scale[i] = h->scale[(size_t)i];
        if (inv_scale) inv_scale[i] = h->invScale[(size_t)i];
        if (sigma2) sigma2[i] = h->sigma2[(size_t)i];
        if (inv_sigma2) inv_sigma2[i] = h->invSigma2[(size_t)i];
        if (features_per_level) features_per_level[i] = h->quota[(size_t)i];
    }
    return ORBX_OK;
}

extern "C" int orbx_extractor_tables_for(const orbx_extractor_config *cfg, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2, int *features_per_level)
{
    if (!cfg || cfg->nlevels < 1 || cfg->nlevels > ORBX_MAX_LEVELS || cfg->nfeatures < 1 || !(cfg->scale_factor > 1.0f)) { orbx_set_error("bad extractor configuration"); return ORBX_ERR_ARG; }
    orbx_extractor tmp;           // host tables only: no device, no stream
    tmp.cfg = *cfg;
    build_tables(&tmp);
    for (int i = 0; i < cfg->nlevels; i++) {
        if (scale) scale[i] = tmp.scale[(size_t)i];
        if (inv_scale) inv_scale[i] = tmp.invScale[(size_t)i];
        if (sigma2) sigma2[i] = tmp.sigma2[(size_t)i];
        if (inv_sigma2) inv_sigma2[i] = tmp.invSigma2[(size_t)i];
        if (features_per_level) features_per_level[i] = tmp.quota[(size_t)i];
    }
    return ORBX_OK;
}

extern "C" int orbx_extractor_capacity(const orbx_extractor *h)
{
    if (!h) return ORBX_ERR_ARG;
    int cap = 0;
    for (int l = 0; l < h->cfg.nlevels; l++) cap += h->quota[(size_t)l] + 3 + 16;
    return cap;
}

extern "C" int orbx_upload_frames(orbx_extractor *h, const uint8_t *const *images, int batch, int width, int height, int stride,
                                  const void **images_dev, int *dev_stride, size_t *dev_frame_pitch)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    int rc = upload(h, images, batch, width, height, stride);
    if (rc != ORBX_OK) return rc;
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (images_dev) *images_dev = h->staging.p;
    if (dev_stride) *dev_stride = h->stagingStride;
    if (dev_frame_pitch) *dev_frame_pitch = h->stagingFramePitch;
    return ORBX_OK;
}

extern "C" int orbx_extract_batch_device(orbx_extractor *h, const void *images_dev, int batch, int width, int height, int stride, size_t frame_pitch)
{
    if (!h || !images_dev) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (stride < width || frame_pitch < (size_t)stride * (size_t)(height - 1) + (size_t)width) { orbx_set_error("bad stride / frame pitch"); return ORBX_ERR_ARG; }
    return run_batch(h, (const uint8_t *)images_dev, batch, width, height, stride, frame_pitch);
}

extern "C" int orbx_batch_results_device(orbx_extractor *h, const orbx_keypoint **keypoints_dev, const uint8_t **descriptors_dev,
                                         const int32_t **counts_dev, int *capacity)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (!h->lastBatch) return no_batch_error(h);
    if (keypoints_dev) *keypoints_dev = h->outKpP[h->cur];
    if (descriptors_dev) *descriptors_dev = h->outDescP[h->cur];
    if (counts_dev) *counts_dev = h->outCntP[h->cur];
    if (capacity) *capacity = h->geom.outCap;
    return ORBX_OK;
}

extern "C" int orbx_extractor_status(orbx_extractor *h, int32_t *bits)
{
    if (!h || !bits) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!h->lastBatch) return no_batch_error(h);
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    int v = 0;
    ORBX_HIP_CHECK(hipMemcpy(&v, h->outStP[h->cur] + h->lastBatch, sizeof(int), hipMemcpyDeviceToHost));
    *bits = v;
    return ORBX_OK;
}

extern "C" int orbx_batch_status_device(orbx_extractor *h, const int32_t **status_dev, int *batch)
{
    if (!h || !status_dev) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!h->lastBatch) return no_batch_error(h);
    *status_dev = h->outStP[h->cur];
    if (batch) *batch = h->lastBatch;
    return ORBX_OK;
}

/* device word of the last batch (OR of its frames' capacity bits) for consumers ordered behind this handle's stream: it lives in the
 * batch's result buffer, so the consumer event that guards the results guards it too */
const int *orbx_extractor_status_word_internal(orbx_extractor *h) { return h && h->lastBatch ? h->outStP[h->cur] + h->lastBatch : nullptr; }

extern "C" int orbx_extractor_sync(orbx_extractor *h)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    return ORBX_OK;
}

static int ensure_host_out(orbx_extractor *h, size_t bytes)
{
    if (bytes <= h->hostOutBytes) return ORBX_OK;
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (h->hostOut) (void)hipHostFree(h->hostOut);
    h->hostOut = nullptr; h->hostOutBytes = 0;
    invalidate_single_graph(h);           // the read-back node targets this buffer
    ORBX_HIP_CHECK(hipHostMalloc((void **)&h->hostOut, bytes, hipHostMallocDefault));
    h->hostOutBytes = bytes;
    return ORBX_OK;
}

// Counts, capacity words and the result arrays come back whole through ONE pinned buffer, enqueued behind the kernels and
// followed by ONE synchronisation (a single-frame call is latency bound: every extra sync / blocking copy costs 10-20 us).
// -> offsets of the keypoints / descriptors inside h->hostOut (counts at 0); checks the capacity words.
static int fetch_results(orbx_extractor *h, int batch, bool wantKp, bool wantDesc, size_t *offKpOut, size_t *offDescOut)
{
    const int cap = h->geom.outCap;
    const size_t B = (size_t)batch;
    const bool whole = batch == h->allocBatch && wantKp && wantDesc;      // the buffer's own layout: one copy brings counts, capacity words, keypoints and descriptors
    const size_t offKp = whole ? h->arenaKpOff : align_up(B * sizeof(int), 256), offDesc = whole ? h->arenaDescOff : offKp + align_up(B * cap * sizeof(orbx_keypoint), 256),
                 offSt = whole ? B * sizeof(int) : offDesc + align_up(B * cap * 32, 256), bytes = whole ? h->arenaBytes : offSt + align_up((B + 1) * sizeof(int), 256);
    int rc = ensure_host_out(h, bytes);
    if (rc != ORBX_OK) return rc;
    uint8_t *hp = h->hostOut;
    if (whole) ORBX_HIP_CHECK(hipMemcpyAsync(hp, h->outArena[h->cur].p, h->arenaBytes, hipMemcpyDeviceToHost, h->stream));
    else {
        ORBX_HIP_CHECK(hipMemcpyAsync(hp, h->outCntP[h->cur], B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        if (wantKp) ORBX_HIP_CHECK(hipMemcpyAsync(hp + offKp, h->outKpP[h->cur], B * cap * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, h->stream));
        if (wantDesc) ORBX_HIP_CHECK(hipMemcpyAsync(hp + offDesc, h->outDescP[h->cur], B * cap * 32, hipMemcpyDeviceToHost, h->stream));
        ORBX_HIP_CHECK(hipMemcpyAsync(hp + offSt, h->outStP[h->cur], B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    }
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    const int *st = (const int *)(hp + offSt);
    for (int f = 0; f < batch; f++)
        if (st[f]) {
            orbx_set_error("frame %d: device capacity error bits 0x%x (1: FAST candidates of a level, 2: quadtree node list, 4: level keypoints)", f, st[f]);
            return ORBX_ERR_CAPACITY;
        }
    *offKpOut = offKp; *offDescOut = offDesc;
    return ORBX_OK;
}

// Plan of k_pyramid_tiles for the current geometry: a gx x gy grid of tiles (~12 px at the deepest level), for every tile and level
// the rectangle it owns (grid boundaries scaled to the level, x rounded down to multiples of 4) and the rectangle it computes =
// own + what its computed rectangle of the NEXT level reads according to the resize tables (columns: first source column and span of
// every group of 4; rows: the two source rows).  False: no plan (the caller uses the per-level launches).
static bool plan_pyramid_tiles(orbx_extractor *h, std::vector<OrbxPyrTile> &out, int &tiles, int &bufBytes, int &tabBytes)
{
    const OrbxGeom &g = h->geom;
    const int nl = g.nlevels;
    if (nl < 2) return false;
    int gx = std::max(1, g.lv[nl - 1].w / 12), gy = std::max(1, g.lv[nl - 1].h / 12);
    while (gx * gy > 1024) { if (gx >= gy) gx = (gx + 1) / 2; else gy = (gy + 1) / 2; }
    auto bx = [&](int L, int i) { return i >= gx ? ((g.lv[L].w + 3) & ~3) : (int)((long long)i * g.lv[L].w / gx) & ~3; };
    auto by = [&](int L, int j) { return j >= gy ? g.lv[L].h : (int)((long long)j * g.lv[L].h / gy); };
    out.assign((size_t)gx * gy * nl, OrbxPyrTile{0, 0, 0, 0, 0, 0, 0, 0});
    size_t maxBuf = 0, maxTab = 0;
    for (int j = 0; j < gy; j++)
        for (int i = 0; i < gx; i++) {
            OrbxPyrTile *tt = out.data() + ((size_t)j * gx + i) * nl;
            size_t tabSum = 0;
            for (int L = nl - 1; L >= 0; L--) {
                int x0 = 0, x1 = 0, y0 = 0, y1 = 0;          // computed rectangle, empty so far
                if (L > 0) { x0 = bx(L, i); x1 = bx(L, i + 1); y0 = by(L, j); y1 = by(L, j + 1); }
                OrbxPyrTile &t = tt[L];
                t.ox0 = (short)x0; t.ox1 = (short)x1; t.oy0 = (short)y0; t.oy1 = (short)y1;
                if (x1 <= x0 || y1 <= y0) { x0 = x1 = y0 = y1 = 0; t.ox0 = t.ox1 = t.oy0 = t.oy1 = 0; }
                if (L < nl - 1) {
                    const OrbxPyrTile &n = tt[L + 1];
                    if (n.cx1 > n.cx0 && n.cy1 > n.cy0) {
                        const OrbxLevel &ln = g.lv[L + 1];
                        int nx0 = 1 << 30, nx1 = 0, ny0 = 1 << 30, ny1 = 0;
                        for (int gq = n.cx0 >> 2; gq < n.cx1 >> 2; gq++) {
                            const uint32_t *ct = h->rsHost.data() + ln.rsColOff + 12 * gq;
                            if ((int)ct[1] > 7) return false;                      // taps of a group wider than the 8-byte window
                            nx0 = std::min(nx0, (int)ct[0]);
                            nx1 = std::max(nx1, (int)ct[0] + (int)ct[1] + 1);
                        }
                        for (int y = n.cy0; y < n.cy1; y++) {
                            const uint32_t ty = h->rsHost[(size_t)ln.rsRowOff + 2 * y];
                            ny0 = std::min(ny0, (int)(ty & 0xffffu));
                            ny1 = std::max(ny1, (int)(ty >> 16) + 1);
                        }
                        if (x1 <= x0) { x0 = nx0; x1 = nx1; y0 = ny0; y1 = ny1; }
                        else { x0 = std::min(x0, nx0); x1 = std::max(x1, nx1); y0 = std::min(y0, ny0); y1 = std::max(y1, ny1); }
                    }
                }
                x0 &= ~3; x1 = (x1 + 3) & ~3;
                t.cx0 = (short)x0; t.cx1 = (short)x1; t.cy0 = (short)y0; t.cy1 = (short)y1;
                if (y1 - y0 > 255 || x1 - x0 > 1020) return false;
                const size_t pitch = (size_t)((x1 - x0 + 12 + 7) & ~7);               // PT_PITCH of orbx_kernels.hip
                if (L == 0 && pitch * (size_t)(y1 - y0) > 2048 * 8) return false;      // level-0 window: at most 8 eight-byte units per thread (PT_WIN_ITEMS)
                if (L > 0) {      // table slices staged in LDS: 48 bytes per column group, 8 per row, one item per thread and level
                    const int ng = (x1 - x0) >> 2, ch = y1 - y0;
                    if (3 * ng + ch > 256) return false;
                    tabSum += (size_t)48 * ng + (((size_t)8 * ch + 15) & ~(size_t)15);
                }
                maxBuf = std::max(maxBuf, pitch * (size_t)(y1 - y0) + 16);
            }
            maxTab = std::max(maxTab, tabSum);
        }
    maxBuf = align_up(maxBuf, 16);
    if (2 * maxBuf + maxTab > 96 * 1024) return false;
    tiles = gx * gy; bufBytes = (int)maxBuf; tabBytes = (int)align_up(maxTab + 16, 16);
    return true;
}

// ---------------------------------------------------------------------------------------------
// One host frame in, results in pinned memory out: the body of ORBextractor::operator() for a handle created with max_batch = 1
// (shim/ORBextractor.cc).  The frame's rows are laid out at the device pitch in the pinned staging buffer, then ONE graph launch
// replays  upload -> k_pyramid_tiles (all levels, one launch) -> k_fast_cells -> k_octree -> k_blur -> k_orient_describe -> read-back
// with the pointers of result buffer `cb` baked in (one graph per buffer), followed by one synchronisation.
// ---------------------------------------------------------------------------------------------
static std::mutex g_graphBuildMutex;

static int build_single_graph(orbx_extractor *h)
{
    // explicit node API (no stream capture - see emit() in orbx_kernels.hip), ONE chain:
    //   upload -> k_pyramid_tiles (clears the status words; without a tile plan: status clear -> k_resize per level) -> k_fast_cells -> k_octree -> k_blur
    //   -> k_orient_describe -> read-back
    // (measured: a second branch for k_blur next to the detector chain makes hipGraphLaunch use internal side streams - 166 us per frame
    // instead of 127 for this chain, and concurrent launches of such graphs from several threads crashed inside the runtime)
    const size_t fp = h->stagingFramePitch;
    std::lock_guard<std::mutex> lock(g_graphBuildMutex);      // graphs of different handles are built one at a time (first call of every handle)
    {   // the pyramid as ONE node (k_pyramid_tiles) where the geometry has a plan; ORBX_PYR_LEVELS=1: one node per level (measurement switch)
        std::vector<OrbxPyrTile> plan;
        const char *e = getenv("ORBX_PYR_LEVELS");
        h->ptTiles = 0;
        if (!(e && e[0] == '1') && plan_pyramid_tiles(h, plan, h->ptTiles, h->ptLds, h->ptTab)) {
            int rc = h->ptDev.ensure(plan.size());
            if (rc != ORBX_OK) return rc;
            ORBX_HIP_CHECK(hipMemcpy(h->ptDev.p, plan.data(), plan.size() * sizeof(OrbxPyrTile), hipMemcpyHostToDevice));
        } else h->ptTiles = 0;
    }
    for (int cb = 0; cb < 2; cb++) {
        OrbxLaunch L;
        fill_launch(h, L, h->staging.p, 1, h->stagingStride, fp, cb);
        hipGraph_t g = nullptr;
        ORBX_HIP_CHECK(hipGraphCreate(&g, 0));
        h->sgGraph[cb] = g;
        hipGraphNode_t nUp = nullptr, nClr = nullptr, nPyr = nullptr, nChain = nullptr, nBlur = nullptr, nDesc = nullptr, nDown = nullptr;
        ORBX_HIP_CHECK(hipGraphAddMemcpyNode1D(&nUp, g, nullptr, 0, h->staging.p, h->hostStaging, fp, hipMemcpyHostToDevice));
        hipMemsetParams mp;
        memset(&mp, 0, sizeof(mp));
        mp.dst = h->status.p; mp.elementSize = sizeof(int); mp.width = 2; mp.height = 1; mp.pitch = 2 * sizeof(int); mp.value = 0;
        if (h->ptTiles > 0) nClr = nUp;      // k_pyramid_tiles clears the two status words itself
        else ORBX_HIP_CHECK(hipGraphAddMemsetNode(&nClr, g, &nUp, 1, &mp));
        L.graph = g;
        int rc;
        nPyr = nClr;
        if (h->ptTiles > 0) {
            L.pyrTiles = h->ptDev.p; L.pyrTileCount = h->ptTiles; L.pyrTileBuf = h->ptLds; L.pyrTileTab = h->ptTab;
            L.deps[0] = nPyr; L.ndeps = 1; L.node = &nPyr;
            if ((rc = orbx_launch_pyramid_tiles(L)) != ORBX_OK) return rc;
        } else
            for (int l = 1; l < h->geom.nlevels; l++) {
                L.deps[0] = nPyr; L.ndeps = 1; L.node = &nPyr;
                if ((rc = orbx_launch_resize(L, l)) != ORBX_OK) return rc;
            }
        L.deps[0] = nPyr; L.ndeps = 1; L.node = &nChain;
        if ((rc = orbx_launch_fast_cells(L)) != ORBX_OK) return rc;
        L.deps[0] = nChain;
        if ((rc = orbx_launch_octree(L)) != ORBX_OK) return rc;
        L.deps[0] = nChain; L.node = &nBlur;
        if ((rc = orbx_launch_blur(L)) != ORBX_OK) return rc;
        L.deps[0] = nBlur; L.node = &nDesc;
        if ((rc = orbx_launch_orient_describe(L)) != ORBX_OK) return rc;
        ORBX_HIP_CHECK(hipGraphAddMemcpyNode1D(&nDown, g, &nDesc, 1, h->hostOut, h->outArena[cb].p, h->arenaBytes, hipMemcpyDeviceToHost));
        ORBX_HIP_CHECK(hipGraphInstantiate(&h->sgExec[cb], g, nullptr, nullptr, 0));
    }
    h->sgValid = true;
    return ORBX_OK;
}

// ---------------------------------------------------------------------------------------------
// THE COMBINER.  ORBextractor::operator() is a one-frame, synchronous call (reference include/ORBextractor.h:110; callers
// src/Frame.cc:159-167, 394, 503), and one frame occupies a fraction of the device for ~0.1 ms of mostly dependent latency.  Calls that
// are inside the library AT THE SAME MOMENT - the left / right extractor threads of the stereo Frame constructor, the tracking threads
// of several sequences - are therefore merged into ONE launch set of n frames on a shared ENGINE (an extractor handle with
// max_batch = ORBX_COMBINE_MAX, its own stream, one graph per n):
//     k_comb_upload (members' pinned frames -> engine) -> k_pyramid_tiles -> k_fast_cells -> k_octree -> k_blur -> k_orient_describe
//     -> k_comb_finish (engine -> every member's device buffers + pinned result arena [+ pinned pyramid])
// Protocol ("group commit"): a caller stages its frame in its own pinned buffer (in parallel with the others), then joins the OPEN batch
// of its (device, configuration, image size); the first to join is the batch's leader.  The leader launches as soon as an engine is free
// and nobody else is on the way in (callers announce themselves before they copy their frame) - a lone caller never waits -, bounded by
// COMB_WAIT_US; while all engines are busy, arrivals keep joining, so the batch size follows the load.  A handle whose call was given a
// partner hint (orbx_extractor_expect_partner: the stereo constructor's other extractor) is waited for up to COMB_PARTNER_US.  The
// leader synchronises the engine's stream and releases the followers.  Every member ends up with exactly the state a call of its own
// would have left: results of the frame in its double-buffered arena on the device and in pinned memory, its pyramid and level 0 on the
// device (ComputeStereoMatches, SearchByBoW chained behind it), so nothing downstream can tell the difference.
// ---------------------------------------------------------------------------------------------
#define COMB_MAX_LIMIT 64
#define COMB_WAIT_US 40.0
// How long a set waits for a partner that a member announced (orbx_extractor_expect_partner).  Default 0 = not at all.  Measured on the
// reference's stereo constructor (two std::threads started one after the other, src/Frame.cc:159-167; 1241x376, 2000 features): the right
// extractor's call arrives 30-40 us after the left one's; waiting for it and running ONE set of two frames makes ExtractORB 275 us per call
// (constructor 410 us), not waiting - the second call takes the other engine, whose stream sits on a hardware queue of its own - 255 us
// (358 us), every handle on its own graph (ORBX_COMBINE=0) 214 us (347 us).  The skew, not the launch count, is what the pair pays for.
// ORBX_COMBINE_PARTNER_US=<microseconds> turns the wait on (a caller whose two calls arrive together, e.g. from a thread pool).
static double comb_partner_us()
{
    static const double us = [] { const char *e = getenv("ORBX_COMBINE_PARTNER_US"); return e && *e ? atof(e) : 0.0; }();      // (read once: the leader polls this in its wait loop)
    return us;
}

struct CombEngine {
    orbx_extractor *eng = nullptr;
    OrbxCombMember *tab = nullptr;           // pinned; read by the first and the last kernel of a batch
    const OrbxCombMember *tabDev = nullptr;
    hipGraph_t graph[COMB_MAX_LIMIT + 1] = {};
    hipGraphExec_t exec[COMB_MAX_LIMIT + 1] = {};
    bool planned = false;
    unsigned *syncDev = nullptr;             // device: arrivals of k_comb_finish + the count of completed launch sets
    unsigned long long *flag = nullptr, *flagDev = nullptr;   // mapped host: that count as the host sees it (behind the set's results)
    unsigned long long launched = 0;         // launch sets enqueued on this engine so far (what the leader waits for)
    bool pooledStream = false, highPrio = false;   // eng->stream came from (and goes back to) the process's engine-stream pool
    std::atomic<int> busy{0};                // taken (under Combiner::mu) by a leader, released by it without the lock
};

struct CombBatch {
    orbx_extractor *m[COMB_MAX_LIMIT];
    bool wantPyr[COMB_MAX_LIMIT];
    int n = 0;                               // under Combiner::mu
    std::vector<orbx_extractor *> waitFor;   // partners announced by members and not here yet (under Combiner::mu)
    std::atomic<int> nNow{0}, waiting{0};    // copies of n and waitFor.size() for the leader, which polls without the lock
    std::atomic<int> uploaded[COMB_MAX_LIMIT] = {};   // member's own upload of its frame: 0 none, 1 in flight, 2 complete
    std::atomic<int> done{0};
    int rc = ORBX_OK;
    char err[256] = "";
};

struct Combiner {
    orbx_extractor_config cfg;               // of the members (max_batch / max sizes aside)
    int W = 0, H = 0, maxB = 16;
    std::atomic<int> maxEngines{2};          // lowered by a leader that could not build another engine, read by everybody
    int users = 0;                           // handles attached to this engine set (under g_combMu): the set is released with the last one
    int dstStride = 0;
    size_t fp = 0, kpOff = 0, descOff = 0;
    std::mutex mu;
    std::atomic<int> entering{0};
    std::atomic<int> active{0};              // calls inside the combiner right now (staging, waiting, in flight): the concurrency the set size follows
    std::shared_ptr<CombBatch> open;
    CombEngine *engines[8] = {};             // only ever appended to (by the one waiting leader); read without the lock up to nEngines
    std::atomic<int> nEngines{0};
    std::atomic<long> batches{0}, frames{0};
    // where the calls' time goes (microseconds, summed over calls / launch sets): staging copy, leader's wait, graph launch call, device + sync
    std::atomic<long> usStage{0}, usWait{0}, usLaunch{0}, usSync{0};
    std::atomic<long> setsByN[COMB_MAX_LIMIT + 1] = {}, usByN[COMB_MAX_LIMIT + 1] = {};      // launch sets of n frames, and their device + sync time
};

static std::mutex g_combMu;
static std::vector<Combiner *> g_combs;

// engine streams, per device and priority class: created on demand, never destroyed (see comb_new_engine)
static std::mutex g_streamPoolMu;
static std::vector<hipStream_t> g_streamPool[64][2];
static hipStream_t comb_stream_take(int device, bool high)
{
    std::lock_guard<std::mutex> lock(g_streamPoolMu);
    std::vector<hipStream_t> &pool = g_streamPool[device & 63][high ? 1 : 0];
    if (!pool.empty()) { hipStream_t s = pool.back(); pool.pop_back(); return s; }
    hipStream_t s = nullptr;
    if (high) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return s;
    }
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return s;
}
static void comb_stream_give(int device, bool high, hipStream_t s)
{
    std::lock_guard<std::mutex> lock(g_streamPoolMu);
    g_streamPool[device & 63][high ? 1 : 0].push_back(s);
}

static inline void cpu_relax() { __builtin_ia32_pause(); }
static inline double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int env_int(const char *name, int dflt, int lo, int hi)
{
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    const int v = atoi(e);
    return v < lo ? lo : v > hi ? hi : v;
}

static bool comb_cfg_equal(const orbx_extractor_config &a, const orbx_extractor_config &b)
{
    return a.device == b.device && a.nfeatures == b.nfeatures && a.scale_factor == b.scale_factor && a.nlevels == b.nlevels && a.ini_th_fast == b.ini_th_fast &&
           a.min_th_fast == b.min_th_fast && memcmp(a.gauss_taps, b.gauss_taps, sizeof(a.gauss_taps)) == 0;
}

// a handle leaves its engine set (it moved to another image size, or is being destroyed); the LAST one takes the set with it: engines, graphs,
// pinned tables - every distinct (size, configuration) a process ever saw used to keep up to two sixteen-frame engines until exit.
// g_combMu held.  No call of a detached handle can be inside the set: a handle is inside only during its own synchronous call.
static void comb_detach_locked(orbx_extractor *h)
{
    Combiner *c = h->comb;
    h->comb = nullptr;
    if (!c || --c->users > 0) return;
    static const bool keep = getenv("ORBX_COMBINE_KEEP") && getenv("ORBX_COMBINE_KEEP")[0] == '1';      // (measurement switch: the round-4 behaviour, engine sets live until exit)
    if (keep) return;
    g_combs.erase(std::remove(g_combs.begin(), g_combs.end(), c), g_combs.end());
    const int ne = c->nEngines.load();
    for (int i = 0; i < ne; i++) {
        CombEngine *E = c->engines[i];
        if (!E) continue;
        if (E->eng && E->eng->stream) { (void)hipSetDevice(E->eng->cfg.device); (void)hipStreamSynchronize(E->eng->stream); }
        for (int n = 0; n <= COMB_MAX_LIMIT; n++) {
            if (E->exec[n]) (void)hipGraphExecDestroy(E->exec[n]);
            if (E->graph[n]) (void)hipGraphDestroy(E->graph[n]);
        }
        if (E->eng && E->pooledStream && E->eng->stream) { comb_stream_give(E->eng->cfg.device, E->highPrio, E->eng->stream); E->eng->stream = nullptr; }
        if (E->eng) orbx_extractor_destroy(E->eng);
        if (E->tab) (void)hipHostFree(E->tab);
        if (E->flag) (void)hipHostFree(E->flag);
        if (E->syncDev) (void)hipFree(E->syncDev);
        delete E;
    }
    delete c;
}

static void comb_detach(orbx_extractor *h)
{
    std::lock_guard<std::mutex> lock(g_combMu);
    comb_detach_locked(h);
}

// the shared engine set for h's configuration at W x H (created on first use; released with the last handle attached to it)
static Combiner *combiner_for(orbx_extractor *h, int W, int H)
{
    if (h->comb && h->comb->W == W && h->comb->H == H) return h->comb;
    std::lock_guard<std::mutex> lock(g_combMu);
    if (h->comb) comb_detach_locked(h);
    for (Combiner *c : g_combs)
        if (c->W == W && c->H == H && comb_cfg_equal(c->cfg, h->cfg)) { c->users++; return h->comb = c; }
    Combiner *c = new Combiner();
    c->cfg = h->cfg; c->W = W; c->H = H;
    c->maxB = env_int("ORBX_COMBINE_MAX", 16, 1, COMB_MAX_LIMIT);
    c->maxEngines.store(env_int("ORBX_COMBINE_ENGINES", 2, 1, 8));
    c->dstStride = (int)align_up((size_t)W + 16, 64);
    c->fp = align_up((size_t)c->dstStride * H + 256, 256);
    g_combs.push_back(c);
    c->users = 1;
    return h->comb = c;
}

// one more engine (the leader that found every existing one busy calls this WITHOUT Combiner::mu; engines are created one at a time)
static int comb_new_engine(Combiner *C, CombEngine **out)
{
    static std::mutex createMu;
    std::lock_guard<std::mutex> lock(createMu);
    orbx_extractor_config cfg = C->cfg;
    cfg.max_width = C->W; cfg.max_height = C->H; cfg.max_batch = C->maxB;
    orbx_extractor *e = nullptr;
    int rc = orbx_extractor_create(&cfg, &e);
    if (rc != ORBX_OK) return rc;
    e->isEngine = true; e->combDisabled = true;
    CombEngine *E = new CombEngine();
    E->eng = e;
    {   // The runtime spreads streams over a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in creation order, and every
        // handle of this library owns a stream: whether two engines' streams land on different queues - i.e. whether their launch sets
        // overlap at all - was a matter of luck (measured: sets of one frame on two engines took 167-172 us each instead of 96 when they
        // shared a queue).  Streams of another PRIORITY come from a queue pool of their own: the second engine takes the high one.
        // ORBX_COMBINE_PRIO=0: plain streams for all.
        // The engines' streams are POOLED per device and outlive their engine sets (an engine set is released with its last handle): which
        // hardware queue a stream lands on depends on the streams alive when it is created, and a set rebuilt later in a process's life used to
        // land differently - the stereo constructor took 370 instead of 245 us after an unrelated engine set had come and gone.
        const char *pe = getenv("ORBX_COMBINE_PRIO");
        const bool high = !(pe && pe[0] == '0') && C->nEngines.load() == 1;
        hipStream_t ps = comb_stream_take(C->cfg.device, high);
        if (ps) {
            (void)hipStreamDestroy(e->stream);
            e->stream = ps;
            E->pooledStream = true; E->highPrio = high;
        }
    }
    auto fail = [&](int code) {
        if (E->pooledStream && e->stream) { (void)hipStreamSynchronize(e->stream); comb_stream_give(C->cfg.device, E->highPrio, e->stream); e->stream = nullptr; }
        orbx_extractor_destroy(e); if (E->tab) (void)hipHostFree(E->tab); if (E->flag) (void)hipHostFree(E->flag); if (E->syncDev) (void)hipFree(E->syncDev); delete E; return code;
    };
    if ((rc = ensure_geometry(e, C->W, C->H, C->maxB)) != ORBX_OK) return fail(rc);
    if ((rc = e->staging.ensure(C->fp * (size_t)C->maxB)) != ORBX_OK) return fail(rc);
    if (hipHostMalloc((void **)&E->tab, sizeof(OrbxCombMember) * (size_t)C->maxB, hipHostMallocDefault) != hipSuccess) { orbx_set_error("hipHostMalloc (member table) failed"); return fail(ORBX_ERR_HIP); }
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, E->tab, 0) != hipSuccess) { orbx_set_error("hipHostGetDevicePointer (member table) failed"); return fail(ORBX_ERR_HIP); }
    E->tabDev = (const OrbxCombMember *)dp;
    // (behind the 64 bytes of counters: the device's copy of the member table, made by the set's first kernel for its last one)
    const size_t syncBytes = 64 + sizeof(OrbxCombMember) * (size_t)C->maxB;
    if (hipMalloc((void **)&E->syncDev, syncBytes) != hipSuccess || hipMemset(E->syncDev, 0, syncBytes) != hipSuccess || hipHostMalloc((void **)&E->flag, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&E->flagDev, E->flag, 0) != hipSuccess) { orbx_set_error("completion word of the engine could not be allocated"); return fail(ORBX_ERR_HIP); }
    *E->flag = 0;
    // the members' arenas have the one-frame layout (ensure_geometry with batch 1): same capacity, same offsets for all of them
    C->kpOff = align_up(3 * sizeof(int), 256);
    C->descOff = C->kpOff + align_up((size_t)e->geom.outCap * sizeof(orbx_keypoint), 256);
    *out = E;
    return ORBX_OK;
}

// the launch set of n combined frames as one graph on engine E (built the first time a batch of n members meets on E)
static int comb_build_graph(Combiner *C, CombEngine *E, int n)
{
    orbx_extractor *e = E->eng;
    std::lock_guard<std::mutex> lock(g_graphBuildMutex);
    if (!E->planned) {
        std::vector<OrbxPyrTile> plan;
        const char *pl = getenv("ORBX_PYR_LEVELS");
        e->ptTiles = 0;
        if (!(pl && pl[0] == '1') && plan_pyramid_tiles(e, plan, e->ptTiles, e->ptLds, e->ptTab)) {
            int rc = e->ptDev.ensure(plan.size());
            if (rc != ORBX_OK) return rc;
            ORBX_HIP_CHECK(hipMemcpy(e->ptDev.p, plan.data(), plan.size() * sizeof(OrbxPyrTile), hipMemcpyHostToDevice));
        } else e->ptTiles = 0;
        E->planned = true;
    }
    OrbxLaunch L;
    fill_launch(e, L, e->staging.p, n, C->dstStride, C->fp, 0);
    L.combTab = E->tabDev; L.combTabCopy = (OrbxCombMember *)((uint8_t *)E->syncDev + 64); L.combKpOff = C->kpOff; L.combDescOff = C->descOff; L.combSync = E->syncDev; L.combFlag = E->flagDev;
    hipGraph_t g = nullptr;
    ORBX_HIP_CHECK(hipGraphCreate(&g, 0));
    // (a failure below must not leave a half-built graph behind - it would be found "built" without an executable and rebuilt on every set of this size)
    struct GraphGuard { hipGraph_t g; bool keep; ~GraphGuard() { if (!keep && g) (void)hipGraphDestroy(g); } } guard = {g, false};
    L.graph = g;
    hipGraphNode_t cur = nullptr, nxt = nullptr;
    int rc;
    L.ndeps = 0; L.node = &nxt;
    if ((rc = orbx_launch_comb_upload(L, e->staging.p)) != ORBX_OK) return rc;
    cur = nxt;
    auto chain = [&](int r) { if (r == ORBX_OK) cur = nxt; return r; };
    L.ndeps = 1;
    if (e->ptTiles > 0) {
        L.pyrTiles = e->ptDev.p; L.pyrTileCount = e->ptTiles; L.pyrTileBuf = e->ptLds; L.pyrTileTab = e->ptTab;
        L.deps[0] = cur;
        if ((rc = chain(orbx_launch_pyramid_tiles(L))) != ORBX_OK) return rc;
    } else {
        hipMemsetParams mp;
        memset(&mp, 0, sizeof(mp));
        mp.dst = e->status.p; mp.elementSize = sizeof(int); mp.width = (size_t)n + 1; mp.height = 1; mp.pitch = ((size_t)n + 1) * sizeof(int); mp.value = 0;
        ORBX_HIP_CHECK(hipGraphAddMemsetNode(&nxt, g, &cur, 1, &mp));
        cur = nxt;
        for (int l = 1; l < e->geom.nlevels; l++) {
            L.deps[0] = cur;
            if ((rc = chain(orbx_launch_resize(L, l))) != ORBX_OK) return rc;
        }
    }
    L.deps[0] = cur; if ((rc = chain(orbx_launch_fast_cells(L))) != ORBX_OK) return rc;
    bool fused = false;
    {   // quadtree, host pyramid copy and blur as one node where k_octree_blur applies (ORBX_COMBINE_FUSE=0: separate nodes, measurement switch)
        const char *fe = getenv("ORBX_COMBINE_FUSE");
        L.deps[0] = cur; L.node = &nxt;
        nxt = nullptr;
        if (!(fe && fe[0] == '0') && (rc = orbx_launch_octree_blur(L, &fused)) != ORBX_OK) return rc;
        if (fused) cur = nxt;
    }
    if (!fused) {
        L.deps[0] = cur; if ((rc = chain(orbx_launch_octree(L))) != ORBX_OK) return rc;
        L.deps[0] = cur; if ((rc = chain(orbx_launch_blur(L))) != ORBX_OK) return rc;
    }
    L.deps[0] = cur; if ((rc = chain(orbx_launch_orient_describe(L))) != ORBX_OK) return rc;
    L.deps[0] = cur; if ((rc = chain(orbx_launch_comb_finish(L))) != ORBX_OK) return rc;
    ORBX_HIP_CHECK(hipGraphInstantiate(&E->exec[n], g, nullptr, nullptr, 0));
    E->graph[n] = g;
    guard.keep = true;
    return ORBX_OK;
}

// the frame's rows at the device pitch in the handle's pinned staging buffer (what every single-frame path uploads from, and what
// level 0 of the host pyramid points at)
static int stage_rows(orbx_extractor *h, const uint8_t *image, int W, int H, int stride, int dstStride, size_t fp)
{
    if (fp > h->hostStagingBytes) {
        ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
        invalidate_single_graph(h);           // its upload node reads this buffer
        if (h->hostStaging) (void)hipHostFree(h->hostStaging);
        h->hostStaging = nullptr; h->hostStagingBytes = 0;
        ORBX_HIP_CHECK(hipHostMalloc((void **)&h->hostStaging, fp, hipHostMallocDefault));
        h->hostStagingBytes = fp;
    }
    if (stride == dstStride) memcpy(h->hostStaging, image, (size_t)dstStride * (size_t)(H - 1) + (size_t)W);
    else for (int y = 0; y < H; y++) memcpy(h->hostStaging + (size_t)y * dstStride, image + (size_t)y * stride, (size_t)W);
    return ORBX_OK;
}

static int check_single_status(orbx_extractor *h, size_t *offKpOut, size_t *offDescOut)
{
    const int st = ((const int *)h->hostOut)[1];      // arena of a one-frame handle: count | frame word | batch word | ...
    if (st) {
        orbx_set_error("frame 0: device capacity error bits 0x%x (1: FAST candidates of a level, 2: quadtree node list, 4: level keypoints)", st);
        return ORBX_ERR_CAPACITY;
    }
    *offKpOut = h->arenaKpOff; *offDescOut = h->arenaDescOff;
    return ORBX_OK;
}

// The leader of batch B (Combiner::mu held through `lk` on entry, released on return): wait for an engine and for whoever is on the
// way in, close the batch, run it, release the followers.
static void comb_lead(Combiner *C, const std::shared_ptr<CombBatch> &B, std::unique_lock<std::mutex> &lk)
{
    const double t0 = now_us();
    CombEngine *E = nullptr;
    // The leader polls WITHOUT the lock - a thread that re-takes a mutex in a loop starves the ones sleeping on it (the joiners) -: engines
    // are taken by compare-and-swap, the batch's size and pending partners are mirrored in atomics; the lock is taken once, to close the batch.
    lk.unlock();
    for (int idle = 0;;) {
        const int ne = C->nEngines.load(std::memory_order_acquire);
        CombEngine *cand = nullptr;
        for (int i = 0; i < ne && !cand; i++) if (!C->engines[i]->busy.load(std::memory_order_acquire)) cand = C->engines[i];
        if (cand) {
            const int waiting = B->waiting.load(std::memory_order_acquire);
            // set size: the callers present share the engines - with E engines a set takes 1/E of them and leaves at once, so that the sets
            // of the others overlap it (16 threads on 2 engines: sets of 5-8 in flight side by side, 65k frames/s; one set of 16 at a time:
            // 45k; one engine: 42-52k).  A lone caller's share is itself: it never waits.
            const int maxE = C->maxEngines.load(std::memory_order_relaxed);
            const int share = (C->active.load(std::memory_order_acquire) + maxE - 1) / maxE;
            const int nNow = B->nNow.load(std::memory_order_acquire);
            const bool full = nNow >= C->maxB || (waiting == 0 && nNow >= std::max(1, share));
            const bool quiet = C->entering.load(std::memory_order_acquire) == 0 && waiting == 0;
            const bool timeUp = now_us() - t0 > (waiting ? comb_partner_us() : COMB_WAIT_US);
            if (full || quiet || timeUp) {
                int expected = 0;
                if (cand->busy.compare_exchange_strong(expected, 1, std::memory_order_acq_rel)) { E = cand; break; }
                continue;
            }
        } else if (ne >= C->maxEngines.load(std::memory_order_relaxed)) {
            // every engine is busy and no more may be built: the leader's own frame goes up meanwhile, like the followers' (see the caller)
            orbx_extractor *me = B->m[0];
            const int up = B->uploaded[0].load(std::memory_order_relaxed);
            if (up == 0) {
                B->uploaded[0].store(1, std::memory_order_release);
                if (hipMemcpyAsync(me->staging.p, me->hostStaging, C->fp, hipMemcpyHostToDevice, me->stream) != hipSuccess) B->uploaded[0].store(3, std::memory_order_release);
            } else if (up == 1 && hipStreamQuery(me->stream) == hipSuccess) B->uploaded[0].store(2, std::memory_order_release);
        } else {
            // every engine is busy (or none exists yet) and one more is allowed: build it while the batch keeps collecting members
            CombEngine *fresh = nullptr;
            const int rcE = comb_new_engine(C, &fresh);
            if (rcE == ORBX_OK) {
                C->engines[ne] = fresh;
                C->nEngines.store(ne + 1, std::memory_order_release);
            } else if (ne == 0) {      // no engine at all: the batch's members fall back to their own graphs
                lk.lock();
                C->open.reset();
                lk.unlock();
                B->rc = ORBX_ERR_STATE;
                snprintf(B->err, sizeof(B->err), "%s", orbx_last_error());
                B->done.store(1, std::memory_order_release);
                return;
            } else C->maxEngines.store(ne, std::memory_order_relaxed);      // (no memory for another one: live with what exists)
            continue;
        }
        // a set in flight takes 0.1 - 0.4 ms: spin that long (a sleeping thread wakes 50+ us late, and this wait IS the call's latency), then yield, and
        // only when the engines are gone for milliseconds - an oversubscribed host - stop burning the core
        if ((++idle & 63) != 0) cpu_relax();
        else { const double w = now_us() - t0; if (w > 5000.0) std::this_thread::sleep_for(std::chrono::microseconds(50)); else if (w > 600.0) std::this_thread::yield(); }
    }
    lk.lock();
    C->open.reset();                    // later arrivals start the next batch (and elect its leader)
    const int n = B->n;
    lk.unlock();
    C->usWait.fetch_add((long)(now_us() - t0), std::memory_order_relaxed);
    int lrc = ORBX_OK;
    for (int i = 0; i < n; i++) {
        orbx_extractor *mh = B->m[i];
        OrbxCombMember &t = E->tab[i];
        const int up = B->uploaded[i].load(std::memory_order_acquire);
        t.srcImg = up == 2 ? mh->staging.p : mh->hostStaging;
        t.devImg = (up == 0 || up == 3) ? mh->staging.p : nullptr;
        t.devPyr = mh->pyr.p;
        t.devArena = mh->outArena[mh->cur].p; t.hostOut = mh->hostOut;
        t.hostPyr = B->wantPyr[i] ? mh->hostOut + mh->hostPyrOff : nullptr;
    }
    if (!E->exec[n]) lrc = comb_build_graph(C, E, n);
    if (lrc == ORBX_OK) {
        const double tL = now_us();
        hipError_t he = hipGraphLaunch(E->exec[n], E->eng->stream);
        const double tY = now_us();
        if (he == hipSuccess) {
            // the set's last kernel raises the engine's completion count in pinned memory behind its last result: polled (a stream synchronisation
            // returns ~10 us after the device is done); the stream is asked only when that takes implausibly long
            const unsigned long long want = ++E->launched;
            static const bool poll = !(getenv("ORBX_COMBINE_POLL") && getenv("ORBX_COMBINE_POLL")[0] == '0');
            bool arrived = false;
            // (sets of several members = many calling threads, the throughput case: measured with 16 threads, polling leaders cost 40 % of the frames/s -
            // the runtime retires a stream's finished commands inside its synchronisation call, and sixteen spinning hosts do not leave it the cores)
            if (poll && n <= 2)
                for (unsigned spins = 1; !(arrived = __atomic_load_n(E->flag, __ATOMIC_ACQUIRE) >= want); spins++) {
                    cpu_relax();
                    if ((spins & 0xfff) == 0 && now_us() - tY > 20000.0) break;
                }
            if (!arrived) he = hipStreamSynchronize(E->eng->stream);
        }
        const long usS = (long)(now_us() - tY);
        C->usLaunch.fetch_add((long)(tY - tL), std::memory_order_relaxed); C->usSync.fetch_add(usS, std::memory_order_relaxed);
        C->setsByN[n].fetch_add(1, std::memory_order_relaxed); C->usByN[n].fetch_add(usS, std::memory_order_relaxed);
        if (he != hipSuccess) { orbx_set_error("combined batch of %d frames failed: %s", n, hipGetErrorString(he)); lrc = ORBX_ERR_HIP; }
    }
    B->rc = lrc;
    if (lrc != ORBX_OK) snprintf(B->err, sizeof(B->err), "%s", orbx_last_error());
    C->batches.fetch_add(1, std::memory_order_relaxed); C->frames.fetch_add(n, std::memory_order_relaxed);
    E->busy.store(0, std::memory_order_release);
    if (B->uploaded[0].load() == 1 && hipStreamSynchronize(B->m[0]->stream) != hipSuccess && B->rc == ORBX_OK) {
        // (the leader's own upload was still in flight when the set was launched: it has to be complete before its buffers are reused)
        B->rc = ORBX_ERR_HIP; snprintf(B->err, sizeof(B->err), "upload of the leader's frame failed");
    }
    B->done.store(1, std::memory_order_release);
}

static bool comb_engine_free(const Combiner *C)
{
    const int ne = C->nEngines.load(std::memory_order_acquire);
    for (int i = 0; i < ne; i++) if (!C->engines[i]->busy.load(std::memory_order_acquire)) return true;
    return ne < C->maxEngines.load(std::memory_order_relaxed) && ne == 0;      // (no engine yet: the first one is about to be built, nobody waits for a busy one)
}

// One call through the combiner.  ORBX_ERR_STATE + combDisabled: the engines cannot be built here, the caller takes the handle's own path.
static int extract_single_combined(orbx_extractor *h, const uint8_t *image, int W, int H, int stride, bool wantPyr, size_t *offKpOut, size_t *offDescOut)
{
    Combiner *C = combiner_for(h, W, H);
    const int dstStride = C->dstStride;
    const size_t fp = C->fp;
    int rc;
    // the member's own buffers (nothing of an earlier call is in flight: calls on a handle are synchronous)
    if ((rc = h->staging.ensure(fp)) != ORBX_OK) return rc;
    h->hostPyrOff = align_up(h->arenaBytes, 256);
    if ((rc = ensure_host_out(h, h->hostPyrOff + h->geom.pyrBytes)) != ORBX_OK) return rc;
    struct ActiveScope {
        Combiner *c;
        explicit ActiveScope(Combiner *cc) : c(cc) { c->active.fetch_add(1, std::memory_order_acq_rel); }
        ~ActiveScope() { c->active.fetch_sub(1, std::memory_order_acq_rel); }
    } activeScope(C);
    C->entering.fetch_add(1, std::memory_order_acq_rel);        // "on my way in": a leader about to launch waits for the copy below
    const double tS = now_us();
    if ((rc = stage_rows(h, image, W, H, stride, dstStride, fp)) != ORBX_OK) { C->entering.fetch_sub(1); return rc; }
    C->usStage.fetch_add((long)(now_us() - tS), std::memory_order_relaxed);
    h->stagingStride = dstStride; h->stagingFramePitch = fp;
    h->cur ^= 1;
    const int cb = h->cur;
    // consumers of this handle's buffers on other streams (a matcher chained behind the previous frames): the engine's stream knows nothing of
    // them, the host waits (they finished long ago in a synchronous caller)
    if (h->pyrConsumerEv) { (void)hipEventSynchronize(h->pyrConsumerEv); h->pyrConsumerEv = nullptr; }
    if (h->consumerEv[cb]) { (void)hipEventSynchronize(h->consumerEv[cb]); h->consumerEv[cb] = nullptr; }

    std::unique_lock<std::mutex> lk(C->mu);
    while (C->open && C->open->n >= C->maxB) { lk.unlock(); cpu_relax(); lk.lock(); }      // a full batch waiting for an engine: the next one
    C->entering.fetch_sub(1, std::memory_order_acq_rel);
    const bool leader = !C->open;
    if (leader) C->open = std::make_shared<CombBatch>();
    std::shared_ptr<CombBatch> B = C->open;
    const int slot = B->n++;
    B->m[slot] = h; B->wantPyr[slot] = wantPyr;
    B->waitFor.erase(std::remove(B->waitFor.begin(), B->waitFor.end(), h), B->waitFor.end());
    if (h->expectPartner) {
        orbx_extractor *p = h->expectPartner;
        h->expectPartner = nullptr;
        bool here = false;
        for (int i = 0; i < B->n; i++) here = here || B->m[i] == p;
        if (!here && p != h) B->waitFor.push_back(p);
    }
    B->waiting.store((int)B->waitFor.size(), std::memory_order_release);
    B->nNow.store(B->n, std::memory_order_release);
    if (!leader) {
        lk.unlock();
        if (!comb_engine_free(C)) {
            // every engine is busy: this call waits for one anyway, so its frame goes up NOW, on the handle's own stream (a DMA that overlaps
            // the launch set in flight), and the set's first kernel gathers it on the device instead of reading it across PCIe
            B->uploaded[slot].store(1, std::memory_order_release);
            if (hipMemcpyAsync(h->staging.p, h->hostStaging, fp, hipMemcpyHostToDevice, h->stream) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess)
                B->uploaded[slot].store(2, std::memory_order_release);
            // (on an error the state stays 1: the set reads the pinned copy, and this member's device copy is rewritten by nobody - flagged below)
        }
        const double tw0 = now_us();
        for (int spins = 1; !B->done.load(std::memory_order_acquire); spins++) {      // (as the leader: spin for a set's duration, yield beyond it, sleep only after milliseconds)
            if ((spins & 63) != 0) cpu_relax();
            else { const double w = now_us() - tw0; if (w > 5000.0) std::this_thread::sleep_for(std::chrono::microseconds(50)); else if (w > 600.0) std::this_thread::yield(); }
        }
        if (B->uploaded[slot].load() == 1) { orbx_set_error("upload of the frame failed"); h->cur ^= 1; return ORBX_ERR_HIP; }
    } else comb_lead(C, B, lk);
    if (B->rc != ORBX_OK) {
        if (B->rc == ORBX_ERR_STATE) h->combDisabled = true;
        orbx_set_error("%s", B->err);
        h->cur ^= 1;                        // nothing was written: the previous results stay current
        return B->rc;
    }
    h->lastBatch = 1; h->lastImg0 = h->staging.p; h->lastStride = dstStride; h->lastFramePitch = fp;
    h->hostPyrValid = wantPyr; h->lastCombined = true;
    return check_single_status(h, offKpOut, offDescOut);
}

static int extract_single_host_impl(orbx_extractor *h, const uint8_t *image, int W, int H, int stride, bool wantPyr, size_t *offKpOut, size_t *offDescOut);
static int extract_single_host(orbx_extractor *h, const uint8_t *image, int W, int H, int stride, bool wantPyr, size_t *offKpOut, size_t *offDescOut)
{
    const int rc = extract_single_host_impl(h, image, W, H, stride, wantPyr, offKpOut, offDescOut);
    if (rc == ORBX_OK && h->allocBatch == 1) h->hostSynced = true;      // every single-frame path ends with a synchronisation: nothing of this call is in flight any more
    return rc;
}

static int extract_single_host_impl(orbx_extractor *h, const uint8_t *image, int W, int H, int stride, bool wantPyr, size_t *offKpOut, size_t *offDescOut)
{
    if (stride < W) { orbx_set_error("bad image pointer / stride"); return ORBX_ERR_ARG; }
    int rc = ensure_geometry(h, W, H, 1);
    if (rc != ORBX_OK) return rc;
    const int dstStride = (int)align_up((size_t)W + 16, 64);
    const size_t fp = align_up((size_t)dstStride * H + 256, 256);
    h->hostPyrValid = false;
    const bool graphOk = !h->sgDisabled && !h->profiling && !h->debugTaps && h->allocBatch == 1;
    if (graphOk && !h->combDisabled) {
        rc = extract_single_combined(h, image, W, H, stride, wantPyr, offKpOut, offDescOut);
        if (!(rc == ORBX_ERR_STATE && h->combDisabled)) return rc;      // (engines unavailable: the handle's own graph below, from now on)
    }
    h->hostPyrOff = align_up(h->arenaBytes, 256);
    if (!graphOk) {
        if ((rc = h->staging.ensure(fp)) != ORBX_OK) return rc;
        if ((rc = stage_rows(h, image, W, H, stride, dstStride, fp)) != ORBX_OK) return rc;
        h->stagingStride = dstStride; h->stagingFramePitch = fp;
        if (wantPyr && (rc = ensure_host_out(h, h->hostPyrOff + h->geom.pyrBytes)) != ORBX_OK) return rc;      // (before the results land in it)
        ORBX_HIP_CHECK(hipMemcpyAsync(h->staging.p, h->hostStaging, fp, hipMemcpyHostToDevice, h->stream));
        if ((rc = run_batch(h, h->staging.p, 1, W, H, h->stagingStride, h->stagingFramePitch)) != ORBX_OK) return rc;
        if ((rc = fetch_results(h, 1, true, true, offKpOut, offDescOut)) != ORBX_OK) return rc;
    } else {
        if (!h->sgValid || h->stagingStride != dstStride || h->stagingFramePitch != fp || fp > h->hostStagingBytes || h->hostPyrOff + h->geom.pyrBytes > h->hostOutBytes) {
            // (re)build: every buffer the graph names must exist first, outside the capture
            ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
            invalidate_single_graph(h);
            if ((rc = h->staging.ensure(fp)) != ORBX_OK) return rc;
            if (fp > h->hostStagingBytes) {
                if (h->hostStaging) (void)hipHostFree(h->hostStaging);
                h->hostStaging = nullptr; h->hostStagingBytes = 0;
                ORBX_HIP_CHECK(hipHostMalloc((void **)&h->hostStaging, fp, hipHostMallocDefault));
                h->hostStagingBytes = fp;
            }
            if ((rc = ensure_host_out(h, h->hostPyrOff + h->geom.pyrBytes)) != ORBX_OK) return rc;
            h->stagingStride = dstStride; h->stagingFramePitch = fp;
            if (build_single_graph(h) != ORBX_OK) {       // no graph support for this sequence: fall back to stream launches for good
                invalidate_single_graph(h);
                h->sgDisabled = true;
                return extract_single_host_impl(h, image, W, H, stride, wantPyr, offKpOut, offDescOut);
            }
        }
        if ((rc = stage_rows(h, image, W, H, stride, dstStride, fp)) != ORBX_OK) return rc;
        h->cur ^= 1;
        const int cb = h->cur;
        if (h->pyrConsumerEv) { ORBX_HIP_CHECK(hipStreamWaitEvent(h->stream, h->pyrConsumerEv, 0)); h->pyrConsumerEv = nullptr; }
        if (h->consumerEv[cb]) { ORBX_HIP_CHECK(hipStreamWaitEvent(h->stream, h->consumerEv[cb], 0)); h->consumerEv[cb] = nullptr; }
        ORBX_HIP_CHECK(hipGraphLaunch(h->sgExec[cb], h->stream));
        h->lastCombined = false;
        h->lastBatch = 1; h->lastImg0 = h->staging.p; h->lastStride = dstStride; h->lastFramePitch = fp;
        if (wantPyr) ORBX_HIP_CHECK(hipMemcpyAsync(h->hostOut + h->hostPyrOff, h->pyr.p, h->geom.pyrBytes, hipMemcpyDeviceToHost, h->stream));
        ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
        if ((rc = check_single_status(h, offKpOut, offDescOut)) != ORBX_OK) return rc;
        h->hostPyrValid = wantPyr;
        return ORBX_OK;
    }
    if (wantPyr) {      // the plain-launch path (profiling / parity taps): one more copy and wait
        ORBX_HIP_CHECK(hipMemcpyAsync(h->hostOut + h->hostPyrOff, h->pyr.p, h->geom.pyrBytes, hipMemcpyDeviceToHost, h->stream));
        ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
        h->hostPyrValid = true;
    }
    return ORBX_OK;
}

extern "C" int orbx_batch_download(orbx_extractor *h, int batch, orbx_keypoint *keypoints, uint8_t *descriptors, int capacity, int *counts)
{
    if (!h || !counts) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (batch <= 0 || batch > h->lastBatch) { orbx_set_error("batch %d not available (last run had %d frames)", batch, h->lastBatch); return ORBX_ERR_STATE; }
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    const int cap = h->geom.outCap;
    const size_t B = (size_t)batch;
    size_t offKp = 0, offDesc = 0;
    int rc = fetch_results(h, batch, keypoints != nullptr, descriptors != nullptr, &offKp, &offDesc);
    if (rc != ORBX_OK) return rc;
    const uint8_t *hp = h->hostOut;
    memcpy(counts, hp, B * sizeof(int));
    for (int f = 0; f < batch; f++)
        if (counts[f] > capacity) { orbx_set_error("frame %d has %d keypoints but the caller's capacity is %d", f, counts[f], capacity); return ORBX_ERR_CAPACITY; }
    parallel_slices(batch, batch >= 16 ? host_copy_threads() : 1, [&](int f0, int f1) {
        for (int f = f0; f < f1; f++) {
            const int n = counts[f];
            if (n == 0) continue;
            if (keypoints) memcpy(keypoints + (size_t)f * capacity, hp + offKp + (size_t)f * cap * sizeof(orbx_keypoint), (size_t)n * sizeof(orbx_keypoint));
            if (descriptors) memcpy(descriptors + (size_t)f * capacity * 32, hp + offDesc + (size_t)f * cap * 32, (size_t)n * 32);
        }
    });
    return ORBX_OK;
}

// All pyramid levels of one frame of the last call in ONE device->host copy (levels 1.. are contiguous in the handle's pyramid
// buffer; level 0 is the input image) through pinned memory: what shim/ORBextractor.cc needs to refill the public mvImagePyramid
// (eight pageable hipMemcpy2D calls cost ~9 ms per frame, this one ~0.1 ms).
extern "C" int orbx_download_pyramid_all(orbx_extractor *h, int frame, uint8_t *const *dst, const int *dst_strides, int nlevels)
{
    if (!h || !dst || !dst_strides) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!h->lastBatch || frame < 0 || frame >= h->lastBatch || nlevels != h->geom.nlevels) { orbx_set_error("frame / level count not available"); return ORBX_ERR_STATE; }
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    const OrbxGeom &g = h->geom;
    const size_t img0Bytes = (size_t)h->lastStride * (size_t)(g.lv[0].h - 1) + (size_t)g.lv[0].w, off0 = align_up(g.pyrBytes, 256);
    int rc = ensure_host_out(h, off0 + img0Bytes);
    if (rc != ORBX_OK) return rc;
    ORBX_HIP_CHECK(hipMemcpyAsync(h->hostOut, h->pyr.p + (size_t)frame * g.pyrBytes, g.pyrBytes, hipMemcpyDeviceToHost, h->stream));
    ORBX_HIP_CHECK(hipMemcpyAsync(h->hostOut + off0, h->lastImg0 + (size_t)frame * h->lastFramePitch, img0Bytes, hipMemcpyDeviceToHost, h->stream));
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    for (int l = 0; l < nlevels; l++) {
        const OrbxLevel &lv = g.lv[l];
        if (!dst[l] || dst_strides[l] < lv.w) { orbx_set_error("bad destination for level %d", l); return ORBX_ERR_ARG; }
        const uint8_t *src = l == 0 ? h->hostOut + off0 : h->hostOut + lv.off;
        const size_t pitch = l == 0 ? (size_t)h->lastStride : (size_t)lv.pitch;
        for (int y = 0; y < lv.h; y++) memcpy(dst[l] + (size_t)y * dst_strides[l], src + (size_t)y * pitch, (size_t)lv.w);
    }
    return ORBX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Host batches as a pipeline (orbx_extract_batch(handle, const uint8_t* const* imgs, int B, ...) with host pointers in and host arrays
// out).  A synchronous call that stages 75 MB, uploads them, runs the launch set and reads 16 MB back one after the other spends 4.3 ms per 256
// frames of 640x480 where PCIe needs 1.5.  _begin stages the frames into the slot's pinned buffer on the copy pool (every thread sends its slice as
// soon as it is staged; frames that already live in pinned / registered memory go straight from there), uploads on its own stream, enqueues the
// launch set behind the upload and the read-back of counts x (28 + 32) bytes behind the launch set on a third stream, and returns; _end waits
// for the OLDEST begun batch's read-back and fills the caller's arrays.  With two batches begun, the staging and upload of batch i+1 and the
// read-back of batch i-1 overlap the kernels of batch i.  orbx_extract_batch runs the same pipeline over chunks of its batch.
// ---------------------------------------------------------------------------------------------------------------------------
static int pipe_streams(orbx_extractor *h)
{
    if (!h->upStream) ORBX_HIP_CHECK(hipStreamCreateWithFlags(&h->upStream, hipStreamNonBlocking));
    if (!h->downStream) ORBX_HIP_CHECK(hipStreamCreateWithFlags(&h->downStream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
        orbx_extractor::PipeSlot &S = h->pipe[i];
        if (!S.evUp) ORBX_HIP_CHECK(hipEventCreateWithFlags(&S.evUp, hipEventDisableTiming));
        if (!S.evKern) ORBX_HIP_CHECK(hipEventCreateWithFlags(&S.evKern, hipEventDisableTiming));
        if (!S.evDown) ORBX_HIP_CHECK(hipEventCreateWithFlags(&S.evDown, hipEventDisableTiming));
    }
    return ORBX_OK;
}

static bool host_pointer_is_pinned(const void *p)      // hipHostMalloc'ed or hipHostRegister'ed memory: the DMA engines read it in place
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }      // (pageable memory: "invalid value")
    return a.type == hipMemoryTypeHost;
}

// Whether EVERY byte of every frame of a batch lies in pinned / registered host memory (k_gather_frames and the DMA engines dereference the
// frames where they are: one pageable frame in the middle of a pinned batch is a GPU memory fault, not an error code).  The runtime is asked
// for the allocation range around a frame's first byte once per distinct allocation (a pool of 256 frames in one hipHostMalloc costs one
// query); a frame has to lie inside ONE such range from its first to its last byte.  When the runtime does not report ranges, the first and the
// last byte of every frame are queried.  The answer is only good for this call (memory may be unregistered between calls): nothing is kept.
static bool host_frames_all_pinned(const uint8_t *const *images, int batch, size_t frameBytes)
{
    struct Range { uintptr_t lo, hi; };
    Range known[8];
    int nk = 0;
    for (int f = 0; f < batch; f++) {
        const uintptr_t a = (uintptr_t)images[f], b = a + frameBytes;      // [a, b)
        bool inside = false;
        for (int k = 0; k < nk && !inside; k++) inside = a >= known[k].lo && b <= known[k].hi;
        if (inside) continue;
        if (!host_pointer_is_pinned(images[f])) return false;
        void *base = nullptr;
        size_t size = 0;
        if (hipPointerGetAttribute(&base, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, (hipDeviceptr_t)images[f]) == hipSuccess &&
            hipPointerGetAttribute(&size, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, (hipDeviceptr_t)images[f]) == hipSuccess && base && size &&
            (uintptr_t)base <= a && a < (uintptr_t)base + size) {
            if (b > (uintptr_t)base + size) {
                // the frame runs past the end of this allocation: the rest may be another registration right behind it, or pageable memory
                if (!host_pointer_is_pinned((const void *)(b - 1))) return false;
                void *base2 = nullptr;
                if (hipPointerGetAttribute(&base2, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, (hipDeviceptr_t)(b - 1)) != hipSuccess || (uintptr_t)base2 != (uintptr_t)base + size) {
                    (void)hipGetLastError();
                    return false;      // (a gap between two registrations cannot be ruled out: stage the batch)
                }
                continue;
            }
            if (nk < 8) known[nk++] = Range{(uintptr_t)base, (uintptr_t)base + size};
            else known[f & 7] = Range{(uintptr_t)base, (uintptr_t)base + size};
        } else {
            (void)hipGetLastError();
            if (!host_pointer_is_pinned((const void *)(b - 1))) return false;
        }
    }
    return true;
}

extern "C" int orbx_extract_batch_begin(orbx_extractor *h, const uint8_t *const *images, int batch, int width, int height, int stride)
{
    if (!h || !images) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (batch <= 0 || stride < width) { orbx_set_error("bad batch / stride"); return ORBX_ERR_ARG; }
    if (h->pipeCount >= 2) { orbx_set_error("two batches are already begun: call orbx_extract_batch_end first"); return ORBX_ERR_STATE; }
    for (int f = 0; f < batch; f++)
        if (!images[f]) { orbx_set_error("image %d is NULL", f); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    int rc = pipe_streams(h);
    if (rc != ORBX_OK) return rc;
    if (h->pipeCount > 0 && (!h->geomValid || h->geom.W != width || h->geom.H != height || batch > h->allocBatch)) {
        // the handle's buffers are about to be rebuilt under a batch in flight: let it finish (its results wait in the slot's pinned buffer)
        ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
        ORBX_HIP_CHECK(hipStreamSynchronize(h->downStream));
    }
    if ((rc = ensure_geometry(h, width, height, batch)) != ORBX_OK) return rc;
    const int W = width, H = height;
    // rows travel PACKED (stride = width rounded up to 4): the padded device pitch of the resident path would put 10 % more bytes on PCIe, and a
    // frame whose rows are already packed is staged with one memcpy; the kernels take any stride (256 spare bytes behind every frame)
    const int dstStride = (int)align_up((size_t)W, 4);
    size_t fp = align_up((size_t)dstStride * H + 256, 256);
    orbx_extractor::PipeSlot &S = h->pipe[(h->pipeHead + h->pipeCount) & 1];      // free: whatever used it last has been ended
    const auto tB0 = std::chrono::steady_clock::now();
    // decided per frame: a batch with ONE pageable frame between pinned ones is staged as a whole (the gather kernel and the DMA engines would fault on it)
    static const bool forceStage = getenv("ORBX_HOST_FORCE_STAGE") && getenv("ORBX_HOST_FORCE_STAGE")[0] == '1';
    const bool pinned = !forceStage && host_frames_all_pinned(images, batch, (size_t)stride * (size_t)(H - 1) + (size_t)W);
    // an error return below, once copies are enqueued, first lets them finish: the slot's pinned input (or the caller's frames) and devIn are reused / freed by the next call
    auto drain = [&](int code) { (void)hipStreamSynchronize(h->upStream); (void)hipStreamSynchronize(h->stream); return code; };
    // packed frames back to back in pinned memory (one (B, H, W) array): the device layout takes the same frame pitch and the batch is ONE copy
    bool oneCopy = pinned && stride == dstStride && ((size_t)stride * H) % 16 == 0;
    for (int f = 1; f < batch && oneCopy; f++) oneCopy = images[f] == images[0] + (size_t)f * stride * H;
    if (oneCopy) fp = (size_t)stride * H;
    const size_t bytes = fp * (size_t)batch;
    if ((rc = S.devIn.ensure(bytes + 256)) != ORBX_OK) return rc;      // (256 readable bytes behind the last frame)
    bool gathered = false;
    if (oneCopy) {
        ORBX_HIP_CHECK(hipMemcpyAsync(S.devIn.p, images[0], bytes, hipMemcpyHostToDevice, h->upStream));
        gathered = true;
    } else if (pinned && (stride & 3) == 0) {
        // one kernel reads the frames where they are (a table of their addresses in pinned memory): 256 two-dimensional copies cost 15 us of host time each
        bool al4 = true, al16 = (stride & 15) == 0 && (dstStride & 15) == 0;
        for (int f = 0; f < batch; f++) { al4 = al4 && ((uintptr_t)images[f] & 3) == 0; al16 = al16 && ((uintptr_t)images[f] & 15) == 0; }
        if (al4) {
            if ((size_t)batch * sizeof(void *) > S.ptrTabBytes) {
                if (S.ptrTab) (void)hipHostFree(S.ptrTab);
                S.ptrTab = nullptr; S.ptrTabBytes = 0;
                const size_t nb = align_up((size_t)std::max(batch, h->cfg.max_batch) * sizeof(void *), 256);
                ORBX_HIP_CHECK(hipHostMalloc((void **)&S.ptrTab, nb, hipHostMallocDefault));
                S.ptrTabBytes = nb;
            }
            for (int f = 0; f < batch; f++) S.ptrTab[f] = images[f];
            if ((rc = orbx_launch_gather_frames(h->upStream, (const uint8_t *const *)S.ptrTab, batch, W, H, stride, S.devIn.p, dstStride, fp, al16)) != ORBX_OK) return drain(rc);
            gathered = true;
        }
    }
    if (gathered) {
    } else if (pinned) {
        for (int f = 0; f < batch; f++)
            if (hipMemcpy2DAsync(S.devIn.p + fp * (size_t)f, (size_t)dstStride, images[f], (size_t)stride, (size_t)W, (size_t)H, hipMemcpyHostToDevice, h->upStream) != hipSuccess) {
                orbx_set_error("upload of frame %d failed: %s", f, hipGetErrorString(hipGetLastError()));
                return drain(ORBX_ERR_HIP);
            }
    } else {
        if (bytes > S.hostInBytes) {
            if (S.hostIn) (void)hipHostFree(S.hostIn);
            S.hostIn = nullptr; S.hostInBytes = 0;
            ORBX_HIP_CHECK(hipHostMalloc((void **)&S.hostIn, bytes, hipHostMallocDefault));
            S.hostInBytes = bytes;
        }
        std::atomic<int> failed{0};
        const int dev = h->cfg.device;
        parallel_slices(batch, batch >= 16 ? host_copy_threads() : 1, [&](int f0, int f1) {
            // (measured and not kept: sending every four staged frames at once - 64 copies per batch from 16 threads queue on the runtime's lock: 98k instead of 145k frames/s)
            for (int f = f0; f < f1; f++) {
                uint8_t *dst = S.hostIn + fp * (size_t)f;
                if (stride == dstStride) memcpy(dst, images[f], (size_t)dstStride * (size_t)(H - 1) + (size_t)W);
                else for (int y = 0; y < H; y++) memcpy(dst + (size_t)y * dstStride, images[f] + (size_t)y * stride, (size_t)W);
            }
            if (f1 > f0 && (hipSetDevice(dev) != hipSuccess ||
                            hipMemcpyAsync(S.devIn.p + fp * (size_t)f0, S.hostIn + fp * (size_t)f0, fp * (size_t)(f1 - f0), hipMemcpyHostToDevice, h->upStream) != hipSuccess))
                failed.store(1);
        });
        if (failed.load()) { orbx_set_error("upload of the batch failed: %s", hipGetErrorString(hipGetLastError())); return drain(ORBX_ERR_HIP); }
    }
    const auto tB1 = std::chrono::steady_clock::now();
    if (hipEventRecord(S.evUp, h->upStream) != hipSuccess || hipStreamWaitEvent(h->stream, S.evUp, 0) != hipSuccess) {
        orbx_set_error("ordering the launch set behind the upload failed: %s", hipGetErrorString(hipGetLastError()));
        return drain(ORBX_ERR_HIP);
    }
    if ((rc = run_batch(h, S.devIn.p, batch, W, H, dstStride, fp)) != ORBX_OK) return drain(rc);
    if (hipEventRecord(S.evKern, h->stream) != hipSuccess) { orbx_set_error("hipEventRecord: %s", hipGetErrorString(hipGetLastError())); return drain(ORBX_ERR_HIP); }
    // read-back: counts and capacity words, then the keypoint / descriptor arrays of the batch's frames (the arena is laid out for allocBatch frames)
    const int cap = h->geom.outCap, cb = h->cur;
    const size_t B = (size_t)batch;
    S.offSt = align_up(B * sizeof(int), 256); S.offKp = S.offSt + align_up((B + 1) * sizeof(int), 256);
    S.offDesc = S.offKp + align_up(B * cap * sizeof(orbx_keypoint), 256);
    const size_t resBytes = S.offDesc + B * cap * 32;
    if (resBytes > S.hostResBytes) {
        if (S.hostRes) (void)hipHostFree(S.hostRes);
        S.hostRes = nullptr; S.hostResBytes = 0;
        if (hipHostMalloc((void **)&S.hostRes, resBytes, hipHostMallocDefault) != hipSuccess) {
            S.hostRes = nullptr;
            orbx_set_error("hipHostMalloc of %zu result bytes failed: %s", resBytes, hipGetErrorString(hipGetLastError()));
            return drain(ORBX_ERR_HIP);
        }
        S.hostResBytes = resBytes;
    }
    if (hipStreamWaitEvent(h->downStream, S.evKern, 0) != hipSuccess ||
        hipMemcpyAsync(S.hostRes, h->outCntP[cb], B * sizeof(int), hipMemcpyDeviceToHost, h->downStream) != hipSuccess ||
        hipMemcpyAsync(S.hostRes + S.offSt, h->outStP[cb], B * sizeof(int), hipMemcpyDeviceToHost, h->downStream) != hipSuccess ||
        hipMemcpyAsync(S.hostRes + S.offKp, h->outKpP[cb], B * cap * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, h->downStream) != hipSuccess ||
        hipMemcpyAsync(S.hostRes + S.offDesc, h->outDescP[cb], B * cap * 32, hipMemcpyDeviceToHost, h->downStream) != hipSuccess ||
        hipEventRecord(S.evDown, h->downStream) != hipSuccess) {
        orbx_set_error("read-back of the batch could not be enqueued: %s", hipGetErrorString(hipGetLastError()));
        (void)hipStreamSynchronize(h->downStream);
        return drain(ORBX_ERR_HIP);
    }
    h->consumerEv[cb] = S.evDown;      // the batch after the next one overwrites this result buffer: only behind the read-back
    S.batch = batch; S.cap = cap;
    h->pipeCount++;
    const auto tB2 = std::chrono::steady_clock::now();
    h->pipeUs[0] += std::chrono::duration<double, std::micro>(tB1 - tB0).count(); h->pipeUs[1] += std::chrono::duration<double, std::micro>(tB2 - tB1).count();
    h->pipeCalls++;
    return ORBX_OK;
}

extern "C" int orbx_extract_batch_end(orbx_extractor *h, orbx_keypoint *keypoints, uint8_t *descriptors, int capacity, int *counts)
{
    if (!h || !counts) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (h->pipeCount <= 0) { orbx_set_error("no batch has been begun"); return ORBX_ERR_STATE; }
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    orbx_extractor::PipeSlot &S = h->pipe[h->pipeHead];
    h->pipeHead ^= 1; h->pipeCount--;          // the slot is handed back whatever happens below
    const auto tE0 = std::chrono::steady_clock::now();
    ORBX_HIP_CHECK(hipEventSynchronize(S.evDown));
    const auto tE1 = std::chrono::steady_clock::now();
    const int batch = S.batch, cap = S.cap;
    const uint8_t *hp = S.hostRes;
    const int *st = (const int *)(hp + S.offSt);
    for (int f = 0; f < batch; f++)
        if (st[f]) {
            orbx_set_error("frame %d: device capacity error bits 0x%x (1: FAST candidates of a level, 2: quadtree node list, 4: level keypoints)", f, st[f]);
            return ORBX_ERR_CAPACITY;
        }
    memcpy(counts, hp, (size_t)batch * sizeof(int));
    for (int f = 0; f < batch; f++)
        if (counts[f] > capacity) { orbx_set_error("frame %d has %d keypoints but the caller's capacity is %d", f, counts[f], capacity); return ORBX_ERR_CAPACITY; }
    const size_t offKp = S.offKp, offDesc = S.offDesc;
    parallel_slices(batch, batch >= 16 ? host_copy_threads() : 1, [&](int f0, int f1) {
        for (int f = f0; f < f1; f++) {
            const int n = counts[f];
            if (n == 0) continue;
            if (keypoints) memcpy(keypoints + (size_t)f * capacity, hp + offKp + (size_t)f * cap * sizeof(orbx_keypoint), (size_t)n * sizeof(orbx_keypoint));
            if (descriptors) memcpy(descriptors + (size_t)f * capacity * 32, hp + offDesc + (size_t)f * cap * 32, (size_t)n * 32);
        }
    });
    h->pipeUs[2] += std::chrono::duration<double, std::micro>(tE1 - tE0).count();
    h->pipeUs[3] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tE1).count();
    return ORBX_OK;
}

extern "C" int orbx_extract_batch(orbx_extractor *h, const uint8_t *const *images, int batch, int width, int height, int stride,
                                  orbx_keypoint *keypoints, uint8_t *descriptors, int capacity, int *counts)
{
    if (!h || !counts) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    if (batch == 1 && h->cfg.max_batch == 1 && images && images[0] && keypoints && descriptors) {
        // the single-frame call of a one-frame handle: one graph launch (extract_single_host), then the copy into the caller's arrays
        size_t offKp = 0, offDesc = 0;
        int rc1 = extract_single_host(h, images[0], width, height, stride, false, &offKp, &offDesc);
        if (rc1 != ORBX_OK) return rc1;
        const int n = *(const int *)h->hostOut;
        if (n > capacity) { orbx_set_error("frame 0 has %d keypoints but the caller's capacity is %d", n, capacity); return ORBX_ERR_CAPACITY; }
        counts[0] = n;
        memcpy(keypoints, h->hostOut + offKp, (size_t)n * sizeof(orbx_keypoint));
        memcpy(descriptors, h->hostOut + offDesc, (size_t)n * 32);
        return ORBX_OK;
    }
    static const int chunk = env_int("ORBX_HOST_BATCH_CHUNK", 64, 0, 1 << 20);      // frames per pipeline stage (0: the round-4 path, one stage)
    if (batch > h->cfg.max_batch) { orbx_set_error("batch %d exceeds the handle's configured maximum %d", batch, h->cfg.max_batch); return ORBX_ERR_CAPACITY; }
    if (chunk > 0 && batch >= 2 * chunk && h->pipeCount == 0 && images) {
        // the synchronous call as a pipeline over its own chunks: staging + upload of chunk c+1 and the read-back of chunk c-1 under the kernels of chunk c
        int rc = ORBX_OK, begun = 0, ended = 0;
        const int nch = (batch + chunk - 1) / chunk;
        auto first = [&](int c) { return c * chunk; };
        auto count = [&](int c) { return std::min(chunk, batch - c * chunk); };
        while (ended < nch) {
            while (rc == ORBX_OK && begun < nch && begun - ended < 2) {
                rc = orbx_extract_batch_begin(h, images + first(begun), count(begun), width, height, stride);
                if (rc == ORBX_OK) begun++;
            }
            if (begun == ended) break;      // (a begin failed with nothing in flight)
            const int f0 = first(ended);
            const int rce = orbx_extract_batch_end(h, keypoints ? keypoints + (size_t)f0 * capacity : nullptr, descriptors ? descriptors + (size_t)f0 * capacity * 32 : nullptr,
                                                   capacity, counts + f0);
            ended++;
            if (rce != ORBX_OK && rc == ORBX_OK) rc = rce;      // (after an error nothing more is begun: drain what was, report the first error)
            if (rc != ORBX_OK && begun == ended) break;
        }
        // the handle's device buffers (results, pyramid, status) now hold the last chunk: no device-side view of this call (include/orbx.h, orbx_extract_batch)
        h->lastBatch = 0; h->lastChunked = true;
        return rc;
    }
    if (h->pipeCount > 0) { orbx_set_error("orbx_extract_batch while batches begun with orbx_extract_batch_begin are in flight"); return ORBX_ERR_STATE; }
    int rc = upload(h, images, batch, width, height, stride);
    if (rc != ORBX_OK) return rc;
    rc = run_batch(h, h->staging.p, batch, width, height, h->stagingStride, h->stagingFramePitch);
    if (rc != ORBX_OK) return rc;
    return orbx_batch_download(h, batch, keypoints, descriptors, capacity, counts);
}

extern "C" int orbx_extract(orbx_extractor *h, const uint8_t *image, int width, int height, int stride, orbx_keypoint *keypoints,
                            uint8_t *descriptors, int capacity, int *count)
{
    if (!h || !count) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    *count = 0;
    if (!image || width <= 0 || height <= 0) return ORBX_OK;   // reference: empty image -> silent return (:1553-1554)
    const uint8_t *imgs[1] = {image};
    return orbx_extract_batch(h, imgs, 1, width, height, stride, keypoints, descriptors, capacity, count);
}

extern "C" int orbx_extract_view_pyramid(orbx_extractor *h, const uint8_t *image, int width, int height, int stride, const orbx_keypoint **keypoints,
                                         const uint8_t **descriptors, int *count, orbx_host_pyramid *pyramid)
{
    if (!h || !count || !keypoints || !descriptors) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    *count = 0; *keypoints = nullptr; *descriptors = nullptr;
    if (pyramid) memset(pyramid, 0, sizeof(*pyramid));
    if (!image || width <= 0 || height <= 0) return ORBX_OK;   // reference: empty image -> silent return (:1553-1554)
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    size_t offKp = 0, offDesc = 0;
    int rc = extract_single_host(h, image, width, height, stride, pyramid != nullptr, &offKp, &offDesc);
    if (rc != ORBX_OK) return rc;
    *count = *(const int *)h->hostOut;
    *keypoints = (const orbx_keypoint *)(h->hostOut + offKp);
    *descriptors = h->hostOut + offDesc;
    if (pyramid) {
        // level 0 = the handle's pinned copy of the caller's image (rows at the staging pitch), levels >= 1 = the pinned copy of the device
        // pyramid, in its layout: views, nothing is copied again
        const OrbxGeom &g = h->geom;
        pyramid->nlevels = g.nlevels;
        for (int l = 0; l < g.nlevels && l < ORBX_PYRAMID_MAX_LEVELS; l++) {
            pyramid->width[l] = g.lv[l].w; pyramid->height[l] = g.lv[l].h;
            pyramid->stride[l] = l ? g.lv[l].pitch : h->stagingStride;
            pyramid->level[l] = l ? h->hostOut + h->hostPyrOff + g.lv[l].off : h->hostStaging;
        }
    }
    return ORBX_OK;
}

extern "C" int orbx_extract_view(orbx_extractor *h, const uint8_t *image, int width, int height, int stride, const orbx_keypoint **keypoints,
                                 const uint8_t **descriptors, int *count)
{
    return orbx_extract_view_pyramid(h, image, width, height, stride, keypoints, descriptors, count, nullptr);
}

extern "C" int orbx_extractor_expect_partner(orbx_extractor *h, orbx_extractor *partner)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    h->expectPartner = partner;
    return ORBX_OK;
}

extern "C" int orbx_combiner_profile(const orbx_extractor *h, double *us4)
{
    if (!h || !us4) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    const Combiner *c = h->comb;
    us4[0] = c ? (double)c->usStage.load() : 0; us4[1] = c ? (double)c->usWait.load() : 0;
    us4[2] = c ? (double)c->usLaunch.load() : 0; us4[3] = c ? (double)c->usSync.load() : 0;
    return ORBX_OK;
}

/* launch sets of n frames (n = 0 .. maxn) and the mean device + synchronisation time of one, in microseconds */
extern "C" int orbx_combiner_histogram(const orbx_extractor *h, int maxn, int64_t *sets, double *mean_us)
{
    if (!h || !sets || !mean_us || maxn < 0) { orbx_set_error("bad argument"); return ORBX_ERR_ARG; }
    const Combiner *c = h->comb;
    for (int n = 0; n <= maxn; n++) {
        const long k = (c && n <= COMB_MAX_LIMIT) ? c->setsByN[n].load() : 0;
        sets[n] = k;
        mean_us[n] = k ? (double)c->usByN[n].load() / (double)k : 0.0;
    }
    return ORBX_OK;
}

extern "C" int orbx_combiner_reset_stats(orbx_extractor *h)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    Combiner *c = h->comb;
    if (!c) return ORBX_OK;
    c->batches = 0; c->frames = 0; c->usStage = 0; c->usWait = 0; c->usLaunch = 0; c->usSync = 0;
    for (int n = 0; n <= COMB_MAX_LIMIT; n++) { c->setsByN[n] = 0; c->usByN[n] = 0; }
    return ORBX_OK;
}

extern "C" int orbx_combiner_stats(const orbx_extractor *h, int64_t *batches, int64_t *frames, int *engines)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    const Combiner *c = h->comb;
    if (batches) *batches = c ? (int64_t)c->batches.load() : 0;
    if (frames) *frames = c ? (int64_t)c->frames.load() : 0;
    if (engines) *engines = c ? c->nEngines.load() : 0;
    return ORBX_OK;
}

extern "C" int orbx_pyramid_level_size(const orbx_extractor *h, int width, int height, int level, int *w, int *hgt)
{
    if (!h || level < 0 || level >= h->cfg.nlevels) { orbx_set_error("bad handle / level"); return ORBX_ERR_ARG; }
    if (w) *w = round_f((float)width * h->invScale[(size_t)level]);
    if (hgt) *hgt = round_f((float)height * h->invScale[(size_t)level]);
    return ORBX_OK;
}

static int download_plane(orbx_extractor *h, const uint8_t *base, int pitch, int w, int hh, uint8_t *dst, int dst_stride)
{
    if (!dst || dst_stride < w) { orbx_set_error("bad destination"); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    ORBX_HIP_CHECK(hipMemcpy2D(dst, (size_t)dst_stride, base, (size_t)pitch, (size_t)w, (size_t)hh, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

extern "C" int orbx_download_pyramid(orbx_extractor *h, int frame, int level, int blurred, uint8_t *dst, int dst_stride)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (!h->lastBatch || frame < 0 || frame >= h->lastBatch || level < 0 || level >= h->geom.nlevels) { orbx_set_error("frame/level not available"); return ORBX_ERR_STATE; }
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    const OrbxLevel &lv = h->geom.lv[level];
    if (blurred && h->lastCombined) {
        orbx_set_error("the blurred pyramid of a combined single-frame call stays on the shared engine: enable orbx_extractor_set_debug_taps (or ORBX_COMBINE=0) to keep it");
        return ORBX_ERR_STATE;
    }
    if (blurred) return download_plane(h, h->blur.p + (size_t)frame * h->geom.pyrBytes + lv.off, lv.pitch, lv.w, lv.h, dst, dst_stride);
    if (level == 0) return download_plane(h, h->lastImg0 + (size_t)frame * h->lastFramePitch, h->lastStride, lv.w, lv.h, dst, dst_stride);
    return download_plane(h, h->pyr.p + (size_t)frame * h->geom.pyrBytes + lv.off, lv.pitch, lv.w, lv.h, dst, dst_stride);
}

extern "C" int orbx_extractor_set_debug_taps(orbx_extractor *h, int enable)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if ((enable != 0) != h->debugTaps) { h->debugTaps = enable != 0; h->allocBatch = 0; invalidate_single_graph(h); }
    return ORBX_OK;
}

extern "C" int orbx_debug_download_scores(orbx_extractor *h, int frame, int level, uint8_t *dst, int dst_stride)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (!h->debugTaps) { orbx_set_error("score map not kept: call orbx_extractor_set_debug_taps(h, 1) before extracting"); return ORBX_ERR_STATE; }
    if (!h->lastBatch || frame < 0 || frame >= h->lastBatch || level < 0 || level >= h->geom.nlevels) { orbx_set_error("frame/level not available"); return ORBX_ERR_STATE; }
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    const OrbxLevel &lv = h->geom.lv[level];
    return download_plane(h, h->score.p + (size_t)frame * h->geom.pyrBytes + lv.off, lv.pitch, lv.w, lv.h, dst, dst_stride);
}

extern "C" int orbx_debug_download_candidates(orbx_extractor *h, int frame, int level, uint32_t *packed, int cap, int *count)
{
    if (!h || !count) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!h->lastBatch || frame < 0 || frame >= h->lastBatch || level < 0 || level >= h->geom.nlevels) { orbx_set_error("frame/level not available"); return ORBX_ERR_STATE; }
    if (h->lastCombined) { orbx_set_error("stage taps of a combined single-frame call stay on the shared engine: enable orbx_extractor_set_debug_taps first"); return ORBX_ERR_STATE; }
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    const OrbxGeom &g = h->geom;
    const OrbxLevel &lv = g.lv[level];
    const int ncell = lv.nCols * lv.nRows;
    std::vector<int> cc((size_t)ncell);
    ORBX_HIP_CHECK(hipMemcpy(cc.data(), h->cellCount.p + (size_t)frame * g.cellsPerFrame + lv.cellBase, (size_t)ncell * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<uint32_t> slots((size_t)ncell * lv.cellCap);
    ORBX_HIP_CHECK(hipMemcpy(slots.data(), h->cellSlots.p + (size_t)frame * g.slotsPerFrame + lv.slotBase, slots.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    int n = 0;
    for (int c = 0; c < ncell; c++)
        for (int k = 0; k < cc[(size_t)c]; k++, n++)
            if (packed && n < cap) packed[n] = slots[(size_t)c * lv.cellCap + k];
    *count = n;
    return ORBX_OK;
}

extern "C" int orbx_debug_download_level_keypoints(orbx_extractor *h, int frame, int level, orbx_keypoint *kps, int cap, int *count)
{
    if (!h || !count) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!h->lastBatch || frame < 0 || frame >= h->lastBatch || level < 0 || level >= h->geom.nlevels) { orbx_set_error("frame/level not available"); return ORBX_ERR_STATE; }
    if (h->lastCombined) { orbx_set_error("stage taps of a combined single-frame call stay on the shared engine: enable orbx_extractor_set_debug_taps first"); return ORBX_ERR_STATE; }
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    const OrbxGeom &g = h->geom;
    const OrbxLevel &lv = g.lv[level];
    int n = 0;
    ORBX_HIP_CHECK(hipMemcpy(&n, h->lvlCnt.p + (size_t)frame * g.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
    std::vector<OrbxLevelKp> tmp((size_t)std::max(n, 1));
    if (n > 0) ORBX_HIP_CHECK(hipMemcpy(tmp.data(), h->lvlKp.p + (size_t)frame * g.kpPerFrame + lv.kpBase, (size_t)n * sizeof(OrbxLevelKp), hipMemcpyDeviceToHost));
    for (int i = 0; i < n && i < cap && kps; i++) {
        kps[i].x = tmp[(size_t)i].x; kps[i].y = tmp[(size_t)i].y; kps[i].size = (float)lv.patchSize; kps[i].angle = tmp[(size_t)i].angle;
        kps[i].response = tmp[(size_t)i].score; kps[i].octave = level; kps[i].class_id = -1;
    }
    *count = n;
    return ORBX_OK;
}

extern "C" int orbx_extractor_set_profiling(orbx_extractor *h, int enable)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    h->profiling = enable != 0;
    if (enable) h->profCount = 0;   // averages restart
    return ORBX_OK;
}

extern "C" int orbx_extractor_last_timing(orbx_extractor *h, float *total_ms, float *stage_ms, int *nstages)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (h->profCount == 0) { orbx_set_error("no timing: enable profiling before the batch calls"); return ORBX_ERR_STATE; }
    ORBX_HIP_CHECK(hipSetDevice(h->cfg.device));
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    const int n = std::min(h->profCount, ORBX_PROF_RING);
    float acc[ST_COUNT] = {0};
    for (int r = 0; r < n; r++)
        for (int s = 0; s < ST_COUNT; s++) {
            float ms = 0.f;
            ORBX_HIP_CHECK(hipEventElapsedTime(&ms, h->ev[r][s], h->ev[r][s + 1]));
            acc[s] += ms;
        }
    float tot = 0.f;
    for (int s = 0; s < ST_COUNT; s++) {
        acc[s] /= (float)n;
        if (stage_ms) stage_ms[s] = acc[s];
        tot += acc[s];
    }
    if (total_ms) *total_ms = tot;
    if (nstages) *nstages = ST_COUNT;
    return ORBX_OK;
}
