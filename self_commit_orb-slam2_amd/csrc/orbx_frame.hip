// orbx_frame.hip -- the rest of the Frame constructor on gfx950.
//
//   orbx_frame_undistort     ==  Frame::UndistortKeyPoints    (reference src/Frame.cc:899-947)
//   orbx_frame_image_bounds  ==  Frame::ComputeImageBounds    (src/Frame.cc:950-1004)
//   orbx_frame_assign_grid   ==  Frame::AssignFeaturesToGrid  (src/Frame.cc:460-491, PosInGrid :868-878)
//   orbx_frame_finish_device ==  the first and the last fused, on an extractor's device-resident batch
//
// cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) is OpenCV's fixed five-iteration
// inverse of the radial/tangential model, evaluated in double from float inputs and stored as
// float (cvUndistortPoints; un-vendored, see DESIGN.md section 3 "parity unpinned").  The
// operation order below is that function's; the library is built with -ffp-contract=off and
// double division is IEEE, so the device result equals the host result bit for bit.
//
// One workgroup per frame: undistort + cell id per keypoint, then the 64x48 grid as CSR
// (cell = x*48 + y as in mGrid[x][y]; indices of a cell ascending = push_back order):
// LDS histogram, scan, unordered scatter, per-cell insertion sort (cells hold < 1 feature on
// average).  A few hundred bytes per keypoint and ~200 FP64 flops: negligible next to the
// extraction it follows on the same stream.
#include <string.h>

#include <atomic>
#include <chrono>

#include "orbx_internal.h"

namespace {

const int GC = ORBX_FRAME_GRID_COLS, GR = ORBX_FRAME_GRID_ROWS, NCELL = GC * GR;

struct CamDev {
    double fx, fy, cx, cy, k[8];
    int distorted;
};

__device__ __forceinline__ void undistort_point(const CamDev &c, float u, float v, float *ou, float *ov)
{
    const double ifx = 1. / c.fx, ify = 1. / c.fy;
    double x = u, y = v, x0, y0;
    x0 = x = (x - c.cx) * ifx;
    y0 = y = (y - c.cy) * ify;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((c.k[7] * r2 + c.k[6]) * r2 + c.k[5]) * r2) / (1 + ((c.k[4] * r2 + c.k[1]) * r2 + c.k[0]) * r2);
        const double deltaX = 2 * c.k[2] * x * y + c.k[3] * (r2 + 2 * x * x);
        const double deltaY = c.k[2] * (r2 + 2 * y * y) + 2 * c.k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    // P = K, R = I: RR = K
    const double xx = c.fx * x + 0.0 * y + c.cx;
    const double yy = 0.0 * x + c.fy * y + c.cy;
    const double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    *ou = (float)(xx * ww);
    *ov = (float)(yy * ww);
}

__global__ void k_undistort_corners(CamDev c, float cols, float rows, float *out)
{
    const int t = threadIdx.x;
    if (t >= 4) return;
    const float u = (t & 1) ? cols : 0.0f, v = (t & 2) ? rows : 0.0f;   // (0,0) (cols,0) (0,rows) (cols,rows), src/Frame.cc:961-968
    undistort_point(c, u, v, out + 2 * t, out + 2 * t + 1);
}

// undistort: kp -> kpUn (else kp is already mvKeysUn and kpUn may be NULL); grid: build the CSR
// FT threads per frame (1024: the kernel is one workgroup's chain of dependent steps - histogram, scan, scatter, sort, write-out -, sixteen
// waves shorten every one of them; 24 -> 11 us for 2000 keypoints).  doneFlag != NULL (latency form, one frame): doneSeq is written there
// - pinned host memory - after the last result, the host polls it instead of synchronising the stream.
#define FT 1024
static_assert(NCELL % FT == 0, "cells per thread");
__global__ __launch_bounds__(FT) void k_frame_finish(CamDev c, orbx_frame_grid g, int undistort, int grid, const orbx_keypoint *__restrict__ kp,
                                                     const int32_t *__restrict__ counts, int cap, orbx_keypoint *__restrict__ kpUn,
                                                     int32_t *__restrict__ gridOff, int32_t *__restrict__ gridIdx, int *__restrict__ doneFlag, int doneSeq, int wide)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int sWave[FT / 64];
    int *cnt = (int *)smem;                                   // [NCELL] counts, then cursors
    int *off = cnt + NCELL;                                   // [NCELL + 1]
    unsigned short *cell = (unsigned short *)(off + NCELL + 4);       // [cap]  (off padded to a multiple of four entries: 16-byte rows for the write-out)
    int *sorted = (int *)(cell + ((cap + 7) & ~7));           // [cap rounded up to 4]
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = counts ? min(counts[f], cap) : cap;
    const orbx_keypoint *in = kp + (size_t)f * cap;
    for (int t = tid; t < NCELL; t += FT) cnt[t] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += FT) {
        orbx_keypoint k = in[i];
        if (undistort) {
            if (c.distorted) undistort_point(c, k.x, k.y, &k.x, &k.y);   // else mvKeysUn = mvKeys, src/Frame.cc:901-905
            kpUn[(size_t)f * cap + i] = k;
        }
        if (grid) {
            // PosInGrid, src/Frame.cc:868-878 (round() of a float: half away from zero)
            const int px = (int)roundf((k.x - g.min_x) * g.width_inv), py = (int)roundf((k.y - g.min_y) * g.height_inv);
            int ce = 0xffff;
            if (!(px < 0 || px >= GC || py < 0 || py >= GR)) { ce = px * GR + py; atomicAdd(&cnt[ce], 1); }
            cell[i] = (unsigned short)ce;
        }
    }
    if (!grid) {
        if (doneFlag) {
            __threadfence_system();
            __syncthreads();
            if (tid == 0) __hip_atomic_store(doneFlag, doneSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    __syncthreads();
    // exclusive scan of the NCELL counts: 12 consecutive cells per thread
    const int PER = NCELL / FT;
    int local = 0;
    for (int t = 0; t < PER; t++) local += cnt[tid * PER + t];
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) sWave[wv] = incl;
    __syncthreads();
    int base = incl - local;
    for (int w = 0; w < wv; w++) base += sWave[w];
    for (int t = 0; t < PER; t++) { const int cval = cnt[tid * PER + t]; off[tid * PER + t] = base; base += cval; }
    if (tid == FT - 1) off[NCELL] = base;
    __syncthreads();
    for (int t = tid; t < NCELL; t += FT) cnt[t] = off[t];   // cursors
    __syncthreads();
    for (int i = tid; i < n; i += FT) {
        const int ce = cell[i];
        if (ce != 0xffff) sorted[atomicAdd(&cnt[ce], 1)] = i;
    }
    __syncthreads();
    for (int ce = tid; ce < NCELL; ce += FT) {
        const int b = off[ce], e = off[ce + 1];
        for (int a = b + 1; a < e; a++) {
            const int v = sorted[a];
            int q = a - 1;
            while (q >= b && sorted[q] > v) { sorted[q + 1] = sorted[q]; q--; }
            sorted[q + 1] = v;
        }
    }
    __syncthreads();
    int32_t *go = gridOff + (size_t)f * (NCELL + 1), *gi = gridIdx + (size_t)f * cap;
    // 16 bytes per lane: in the latency form the destination is pinned host memory, and four-byte stores across PCIe made the release below
    // wait 6-8 us for their completion (s_memrealtime: 7 of the kernel's 16 us); the rows and the buffers are padded to multiples of four entries
    // (wide: one frame, 16-byte aligned destinations padded to multiples of four entries - the latency forms; the batched form packs the frames)
    const int total = off[NCELL];
    if (wide) {
        for (int t = tid; t < (NCELL + 4) / 4; t += FT) ((uint4 *)go)[t] = ((const uint4 *)off)[t];
        for (int t = tid; t < (total + 3) / 4; t += FT) ((uint4 *)gi)[t] = ((const uint4 *)sorted)[t];
    } else {
        for (int t = tid; t <= NCELL; t += FT) go[t] = off[t];
        for (int t = tid; t < total; t += FT) gi[t] = sorted[t];
    }
    if (doneFlag) {
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(doneFlag, doneSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace

struct orbx_frame_ops {
    int device = 0;
    hipStream_t stream = nullptr;   // host-array forms and the corner pass; the device form runs on the extractor's stream
    CamDev cam;
    // results, double buffered in lockstep with the extractor's result buffers
    OrbxDevBuf<orbx_keypoint> kpUn[2];
    OrbxDevBuf<int32_t> gridOff[2], gridIdx[2];
    int cur = 0, lastBatch = 0, lastCap = 0;
    // capacity word of the extractor batch behind result buffer b (device form only): COPIED into this handle's own memory on the
    // extractor's stream when the frame is finished - the extractor may grow, reuse or free its result buffers before the download
    OrbxDevBuf<int> producerCopy;
    bool producerValid[2] = {false, false};
    uint8_t *hostIO = nullptr, *hostIODev = nullptr;   // pinned: inputs and outputs of the host-array forms, read / written by the kernel itself
    size_t hostIOBytes = 0;
    // orbx_frame_finish_begin .. _end: offsets of mvKeysUn / grid offsets / grid indices inside hostIO, the frame's keypoint count
    int pending = 0, pendingN = 0;
    bool pendingUn = false, pendingGrid = false;
    size_t pendUn = 0, pendOff = 0, pendIdx = 0, pendFlag = 0;
    int pendSeq = 0;
    OrbxDevBuf<orbx_keypoint> hostKp;
    OrbxDevBuf<int32_t> hostCount;
    OrbxDevBuf<float> corners;
};

extern "C" int orbx_frame_ops_create(int device, const orbx_camera *cam, orbx_frame_ops **out)
{
    if (!out || !cam) { orbx_set_error("bad frame-ops arguments"); return ORBX_ERR_ARG; }
    *out = nullptr;
    if (cam->ndist != 4 && cam->ndist != 5) { orbx_set_error("ndist must be 4 or 5 (k1 k2 p1 p2 [k3]), got %d", cam->ndist); return ORBX_ERR_ARG; }
    if (!(cam->fx != 0.0f) || !(cam->fy != 0.0f)) { orbx_set_error("focal length must be non-zero"); return ORBX_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { orbx_set_error("no HIP device available: liborbx has no CPU fallback"); return ORBX_ERR_NODEVICE; }
    if (device < 0 || device >= ndev) { orbx_set_error("device %d out of range", device); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(device));
    orbx_frame_ops *h = new orbx_frame_ops();
    h->device = device;
    // (a plain stream: on a high-priority one - tried for the latency forms - the kernel shared a hardware queue with the combiner's second engines,
    // which are the other high-priority streams of the process: the stereo constructor went from 265 to 390 us once a second image size had been used)
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; orbx_set_error("hipStreamCreate failed"); return ORBX_ERR_HIP; }
    CamDev &c = h->cam;
    c.fx = cam->fx; c.fy = cam->fy; c.cx = cam->cx; c.cy = cam->cy;
    for (int i = 0; i < 8; i++) c.k[i] = i < cam->ndist ? (double)cam->dist[i] : 0.0;
    c.distorted = cam->dist[0] != 0.0f;   // the reference's test, src/Frame.cc:901, 953
    int rc = h->corners.ensure(8);
    if (rc == ORBX_OK) rc = h->hostCount.ensure(1);
    if (rc != ORBX_OK) { orbx_frame_ops_destroy(h); return rc; }
    *out = h;
    return ORBX_OK;
}

extern "C" void orbx_frame_ops_destroy(orbx_frame_ops *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    for (int b = 0; b < 2; b++) { h->kpUn[b].release(); h->gridOff[b].release(); h->gridIdx[b].release(); }
    h->hostKp.release(); h->hostCount.release(); h->corners.release(); h->producerCopy.release();
    if (h->hostIO) (void)hipHostFree(h->hostIO);
    delete h;
}

// Frame::ComputeImageBounds, src/Frame.cc:950-1004
extern "C" int orbx_frame_image_bounds(orbx_frame_ops *h, int cols, int rows, float *bounds)
{
    if (!h || !bounds || cols < 1 || rows < 1) { orbx_set_error("bad argument"); return ORBX_ERR_ARG; }
    if (!h->cam.distorted) {
        bounds[0] = 0.0f; bounds[1] = (float)cols; bounds[2] = 0.0f; bounds[3] = (float)rows;
        return ORBX_OK;
    }
    ORBX_HIP_CHECK(hipSetDevice(h->device));
    float hc[8];
    hipLaunchKernelGGL(k_undistort_corners, dim3(1), dim3(64), 0, h->stream, h->cam, (float)cols, (float)rows, h->corners.p);
    ORBX_HIP_CHECK(hipMemcpyAsync(hc, h->corners.p, sizeof(hc), hipMemcpyDeviceToHost, h->stream));
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    bounds[0] = hc[0] < hc[4] ? hc[0] : hc[4];   // mnMinX = min(top-left.x, bottom-left.x)
    bounds[1] = hc[2] > hc[6] ? hc[2] : hc[6];   // mnMaxX = max(top-right.x, bottom-right.x)
    bounds[2] = hc[1] < hc[3] ? hc[1] : hc[3];   // mnMinY = min(top-left.y, top-right.y)
    bounds[3] = hc[5] > hc[7] ? hc[5] : hc[7];   // mnMaxY = max(bottom-left.y, bottom-right.y)
    return ORBX_OK;
}

static int launch_finish(orbx_frame_ops *h, hipStream_t stream, const orbx_frame_grid *grid, bool undistort, const orbx_keypoint *kp, const int32_t *counts,
                         int batch, int cap)
{
    int rc;
    if (cap < 1 || cap > 0xfff0) { orbx_set_error("feature capacity %d out of range", cap); return ORBX_ERR_CAPACITY; }
    h->cur ^= 1;
    const int b = h->cur;
    if (undistort && (rc = h->kpUn[b].ensure((size_t)batch * cap))) return rc;
    if (grid && ((rc = h->gridOff[b].ensure((size_t)batch * (NCELL + 1))) || (rc = h->gridIdx[b].ensure((size_t)batch * cap)))) return rc;
    const size_t lds = (size_t)(2 * NCELL + 4) * 4 + (size_t)((cap + 7) & ~7) * 2 + (size_t)((cap + 3) & ~3) * 4;
    if (lds > 160 * 1024) { orbx_set_error("feature capacity %d too large for the LDS tile", cap); return ORBX_ERR_CAPACITY; }
    if (lds > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_frame_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    orbx_frame_grid g = {0.0f, 0.0f, 0.0f, 0.0f};
    if (grid) g = *grid;
    hipLaunchKernelGGL(k_frame_finish, dim3((unsigned)batch), dim3(FT), lds, stream, h->cam, g, undistort ? 1 : 0, grid ? 1 : 0, kp, counts, cap,
                       undistort ? h->kpUn[b].p : nullptr, grid ? h->gridOff[b].p : nullptr, grid ? h->gridIdx[b].p : nullptr, (int *)nullptr, 0, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { orbx_set_error("kernel launch failed: %s", hipGetErrorString(e)); return ORBX_ERR_HIP; }
    h->lastBatch = batch; h->lastCap = cap;
    return ORBX_OK;
}

extern "C" int orbx_frame_finish_device(orbx_frame_ops *h, orbx_extractor *ext, const orbx_frame_grid *grid)
{
    if (!h || !ext || !grid) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    OrbxLastBatchView view;
    int rc = orbx_extractor_last_batch_view_internal(ext, &view);
    if (rc != ORBX_OK) return rc;
    ORBX_HIP_CHECK(hipSetDevice(h->device));
    rc = launch_finish(h, orbx_extractor_stream_internal(ext), grid, true, view.kp, view.counts, view.batch, view.cap);
    // the capacity word of the batch these results come from (it lives in the extractor's result buffer): orbx_frame_download reports it
    h->producerValid[h->cur] = false;
    const int *word = rc == ORBX_OK ? orbx_extractor_status_word_internal(ext) : nullptr;
    if (word) {
        if ((rc = h->producerCopy.ensure(2)) != ORBX_OK) return rc;
        ORBX_HIP_CHECK(hipMemcpyAsync(h->producerCopy.p + h->cur, word, sizeof(int), hipMemcpyDeviceToDevice, orbx_extractor_stream_internal(ext)));
        h->producerValid[h->cur] = true;
    }
    return rc;
}

extern "C" int orbx_frame_results_device(orbx_frame_ops *h, const orbx_keypoint **kp_un_dev, const int32_t **grid_offsets_dev, const int32_t **grid_indices_dev,
                                         int *capacity)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (!h->lastBatch) { orbx_set_error("no frame has been finished yet"); return ORBX_ERR_STATE; }
    if (kp_un_dev) *kp_un_dev = h->kpUn[h->cur].p;
    if (grid_offsets_dev) *grid_offsets_dev = h->gridOff[h->cur].p;
    if (grid_indices_dev) *grid_indices_dev = h->gridIdx[h->cur].p;
    if (capacity) *capacity = h->lastCap;
    return ORBX_OK;
}

extern "C" int orbx_frame_download(orbx_frame_ops *h, orbx_extractor *ext, int batch, orbx_keypoint *kp_un, int32_t *grid_offsets, int32_t *grid_indices)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (batch < 1 || batch > h->lastBatch) { orbx_set_error("batch not available"); return ORBX_ERR_STATE; }
    ORBX_HIP_CHECK(hipSetDevice(h->device));
    ORBX_HIP_CHECK(hipStreamSynchronize(ext ? orbx_extractor_stream_internal(ext) : h->stream));
    const int b = h->cur;
    if (ext && h->producerValid[b]) {
        int v = 0;
        ORBX_HIP_CHECK(hipMemcpy(&v, h->producerCopy.p + b, sizeof(int), hipMemcpyDeviceToHost));
        if (v) {
            orbx_set_error("the extractor batch these results were computed from overflowed a device capacity (bits 0x%x): results are not the reference's", v);
            return ORBX_ERR_CAPACITY;
        }
    }
    if (kp_un) ORBX_HIP_CHECK(hipMemcpy(kp_un, h->kpUn[b].p, (size_t)batch * h->lastCap * sizeof(orbx_keypoint), hipMemcpyDeviceToHost));
    if (grid_offsets) ORBX_HIP_CHECK(hipMemcpy(grid_offsets, h->gridOff[b].p, (size_t)batch * (NCELL + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (grid_indices) ORBX_HIP_CHECK(hipMemcpy(grid_indices, h->gridIdx[b].p, (size_t)batch * h->lastCap * sizeof(int32_t), hipMemcpyDeviceToHost));
    return ORBX_OK;
}

// Host-array forms (UndistortKeyPoints / AssignFeaturesToGrid of ONE frame, called from the tracking thread: latency is what counts).
// Inputs and outputs live in ONE pinned buffer that the kernel reads and writes directly across PCIe (56 KB in, <= 80 KB out for 2000
// keypoints): memcpy in, one launch, one synchronisation, memcpy out - no copy engine, no device staging (the former form issued two uploads,
// a stream synchronisation and three blocking downloads: 0.07-0.10 ms per call).
static int host_form(orbx_frame_ops *h, const orbx_frame_grid *grid, bool undistort, const orbx_keypoint *keypoints, int n, orbx_keypoint *kp_un,
                     int32_t *grid_offsets, int32_t *grid_indices)
{
    ORBX_HIP_CHECK(hipSetDevice(h->device));
    if (h->pending == 2) ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));      // a frame begun and never ended: its kernel may still be storing into the pinned buffer this call reuses
    h->pending = 0;      // (... and that frame is dropped)
    const int cap = n > 0 ? n : 1;
    if (cap > 0xfff0) { orbx_set_error("feature capacity %d out of range", cap); return ORBX_ERR_CAPACITY; }
    const size_t A = 256, szKp = ((size_t)cap * sizeof(orbx_keypoint) + A - 1) / A * A, szIdx = ((size_t)cap * 4 + A - 1) / A * A, szOff = ((size_t)(NCELL + 1) * 4 + A - 1) / A * A;
    const size_t oIn = 0, oCnt = oIn + szKp, oUn = oCnt + A, oOff = oUn + szKp, oIdx = oOff + szOff, total = oIdx + szIdx;
    if (total > h->hostIOBytes) {
        ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->hostIO) (void)hipHostFree(h->hostIO);
        h->hostIO = nullptr; h->hostIOBytes = 0;
        ORBX_HIP_CHECK(hipHostMalloc((void **)&h->hostIO, total, hipHostMallocDefault));
        void *dp = nullptr;
        ORBX_HIP_CHECK(hipHostGetDevicePointer(&dp, h->hostIO, 0));
        h->hostIODev = (uint8_t *)dp;
        h->hostIOBytes = total;
    }
    if (n > 0) memcpy(h->hostIO + oIn, keypoints, (size_t)n * sizeof(orbx_keypoint));
    *(int32_t *)(h->hostIO + oCnt) = n;
    const size_t lds = (size_t)(2 * NCELL + 4) * 4 + (size_t)((cap + 7) & ~7) * 2 + (size_t)((cap + 3) & ~3) * 4;
    if (lds > 160 * 1024) { orbx_set_error("feature capacity %d too large for the LDS tile", cap); return ORBX_ERR_CAPACITY; }
    if (lds > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_frame_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    orbx_frame_grid g = {0.0f, 0.0f, 0.0f, 0.0f};
    if (grid) g = *grid;
    uint8_t *d = h->hostIODev;
    hipLaunchKernelGGL(k_frame_finish, dim3(1), dim3(FT), lds, h->stream, h->cam, g, undistort ? 1 : 0, grid ? 1 : 0, (const orbx_keypoint *)(d + oIn), (const int32_t *)(d + oCnt), cap,
                       undistort ? (orbx_keypoint *)(d + oUn) : nullptr, grid ? (int32_t *)(d + oOff) : nullptr, grid ? (int32_t *)(d + oIdx) : nullptr, (int *)nullptr, 0, 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { orbx_set_error("kernel launch failed: %s", hipGetErrorString(e)); return ORBX_ERR_HIP; }
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (undistort && n > 0 && kp_un) memcpy(kp_un, h->hostIO + oUn, (size_t)n * sizeof(orbx_keypoint));
    if (grid && grid_offsets) memcpy(grid_offsets, h->hostIO + oOff, (size_t)(NCELL + 1) * 4);
    if (grid && n > 0 && grid_indices) memcpy(grid_indices, h->hostIO + oIdx, (size_t)n * 4);
    return ORBX_OK;
}

// The latency form behind the Frame constructors (include/orbx.h): input = the extractor's device-resident keypoints of its last single-frame
// call, output = this handle's pinned memory, written by the kernel itself; nothing is waited for here.
extern "C" int orbx_frame_finish_begin(orbx_frame_ops *h, orbx_extractor *ext, const orbx_frame_grid *grid)
{
    if (!h || !ext) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (h->pending == 2) {      // a frame begun and never ended: its kernel may still be storing into the pinned buffer (and its completion word) this call reuses
        ORBX_HIP_CHECK(hipSetDevice(h->device));
        ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    }
    h->pending = 0;
    int st = 0;
    if (!orbx_extractor_host_complete_internal(ext, &st)) { orbx_set_error("the extractor's last call was not a completed single-frame call"); return ORBX_ERR_STATE; }
    if (st) { orbx_set_error("the extractor call these features come from overflowed a device capacity (bits 0x%x): results are not the reference's", st); return ORBX_ERR_CAPACITY; }
    OrbxLastBatchView view;
    int rc = orbx_extractor_last_batch_view_internal(ext, &view);
    if (rc != ORBX_OK) return rc;
    ORBX_HIP_CHECK(hipSetDevice(h->device));
    const int cap = view.cap, n = orbx_extractor_host_count_internal(ext);
    if (cap < 1 || cap > 0xfff0 || n > cap) { orbx_set_error("feature capacity %d out of range", cap); return ORBX_ERR_CAPACITY; }
    const bool undist = h->cam.distorted != 0;
    if (!undist && !grid) { h->pending = 1; h->pendingN = n; h->pendingUn = h->pendingGrid = false; return ORBX_OK; }      // (nothing to compute)
    const size_t A = 256, szKp = ((size_t)cap * sizeof(orbx_keypoint) + A - 1) / A * A, szIdx = ((size_t)cap * 4 + A - 1) / A * A, szOff = ((size_t)(NCELL + 1) * 4 + A - 1) / A * A;
    const size_t oUn = 0, oOff = oUn + szKp, oIdx = oOff + szOff, oFlag = oIdx + szIdx, total = oFlag + A + szKp + A;      // (+ the host-array forms' input area: one buffer serves both)
    if (total > h->hostIOBytes) {
        ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->hostIO) (void)hipHostFree(h->hostIO);
        h->hostIO = nullptr; h->hostIOBytes = 0;
        ORBX_HIP_CHECK(hipHostMalloc((void **)&h->hostIO, total, hipHostMallocDefault));
        void *dp = nullptr;
        ORBX_HIP_CHECK(hipHostGetDevicePointer(&dp, h->hostIO, 0));
        h->hostIODev = (uint8_t *)dp;
        h->hostIOBytes = total;
    }
    const size_t lds = (size_t)(2 * NCELL + 4) * 4 + (size_t)((cap + 7) & ~7) * 2 + (size_t)((cap + 3) & ~3) * 4;
    if (lds > 160 * 1024) { orbx_set_error("feature capacity %d too large for the LDS tile", cap); return ORBX_ERR_CAPACITY; }
    if (lds > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_frame_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    orbx_frame_grid g = {0.0f, 0.0f, 0.0f, 0.0f};
    if (grid) g = *grid;
    uint8_t *d = h->hostIODev;
    const int seq = ++h->pendSeq;
    *(volatile int *)(h->hostIO + oFlag) = 0;
    hipLaunchKernelGGL(k_frame_finish, dim3(1), dim3(FT), lds, h->stream, h->cam, g, undist ? 1 : 0, grid ? 1 : 0, view.kp, view.counts, cap,
                       undist ? (orbx_keypoint *)(d + oUn) : nullptr, grid ? (int32_t *)(d + oOff) : nullptr, grid ? (int32_t *)(d + oIdx) : nullptr, (int *)(d + oFlag), seq, 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { orbx_set_error("kernel launch failed: %s", hipGetErrorString(e)); return ORBX_ERR_HIP; }
    h->pending = 2; h->pendingN = n; h->pendingUn = undist; h->pendingGrid = grid != nullptr;
    h->pendUn = oUn; h->pendOff = oOff; h->pendIdx = oIdx; h->pendFlag = oFlag;
    return ORBX_OK;
}

extern "C" int orbx_frame_finish_end(orbx_frame_ops *h, const orbx_keypoint **kp_un, const int32_t **grid_offsets, const int32_t **grid_indices, int *n)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    const int pending = h->pending;
    h->pending = 0;
    if (!pending) { orbx_set_error("no frame has been begun"); return ORBX_ERR_STATE; }
    if (pending == 2) {      // the kernel's completion word in pinned memory (written behind its last result); the stream only if that takes implausibly long
        const volatile int *flag = (const volatile int *)(h->hostIO + h->pendFlag);
        const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        bool arrived = false;
        for (int spin = 0; !(arrived = *flag == h->pendSeq); spin++) {
            __builtin_ia32_pause();
            if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (!arrived) {
            ORBX_HIP_CHECK(hipSetDevice(h->device));
            ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
        }
    }
    if (kp_un) *kp_un = h->pendingUn ? (const orbx_keypoint *)(h->hostIO + h->pendUn) : nullptr;
    if (grid_offsets) *grid_offsets = h->pendingGrid ? (const int32_t *)(h->hostIO + h->pendOff) : nullptr;
    if (grid_indices) *grid_indices = h->pendingGrid ? (const int32_t *)(h->hostIO + h->pendIdx) : nullptr;
    if (n) *n = h->pendingN;
    return ORBX_OK;
}

extern "C" int orbx_frame_undistort(orbx_frame_ops *h, const orbx_keypoint *keypoints, int n, orbx_keypoint *kp_un)
{
    if (!h || n < 0 || (n > 0 && (!keypoints || !kp_un))) { orbx_set_error("bad argument"); return ORBX_ERR_ARG; }
    if (n == 0) return ORBX_OK;
    return host_form(h, nullptr, true, keypoints, n, kp_un, nullptr, nullptr);
}

extern "C" int orbx_frame_assign_grid(orbx_frame_ops *h, const orbx_frame_grid *grid, const orbx_keypoint *kp_un, int n, int32_t *grid_offsets,
                                      int32_t *grid_indices)
{
    if (!h || !grid || !grid_offsets || n < 0 || (n > 0 && (!kp_un || !grid_indices))) { orbx_set_error("bad argument"); return ORBX_ERR_ARG; }
    return host_form(h, grid, false, kp_un, n, nullptr, grid_offsets, grid_indices);
}
