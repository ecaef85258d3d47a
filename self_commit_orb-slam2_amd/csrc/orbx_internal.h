// orbx_internal.h -- shared host/device definitions of liborbx (gfx950 only).
#ifndef ORBX_INTERNAL_H
#define ORBX_INTERNAL_H

#include <hip/hip_runtime.h>
#include <cstring>
#include <stdint.h>

#include "../../include/orbx.h"

#define ORBX_MAX_LEVELS 12
#define ORBX_EDGE 19          /* EDGE_THRESHOLD, reference src/ORBextractor.cc:93 */
#define ORBX_BORDER 16        /* EDGE_THRESHOLD-3 = minBorderX/Y, :1067 */
#define ORBX_HALF_PATCH 15    /* HALF_PATCH_SIZE, :92 */
#define ORBX_PATCH 31         /* PATCH_SIZE, :91 */
#define ORBX_CELL_W 30        /* W, :1060 */
#define ORBX_MAX_INI 8        /* initial quadtree nodes = round(window width / height), :719: levels up to 8.5 times as wide as high */

/* error bits written by kernels into the per-frame status word */
#define ORBX_DEV_ERR_PTCAP 1   /* (unused since the quadtree's point arrays are sized for the worst case) */
#define ORBX_DEV_ERR_NODECAP 2 /* quadtree node list overflow (internal)      */
#define ORBX_DEV_ERR_KPCAP 4   /* level keypoint buffer overflow (internal)   */

/* Geometry of one pyramid level for the current image size (host-built, device-read). */
struct OrbxLevel {
    int w, h, pitch;        /* level size, row pitch in bytes (levels >= 1; level 0 uses the input stride) */
    int off;                /* byte offset of the level inside a frame's pyramid block (level 0: unused)   */
    int nCols, nRows, wCell, hCell;
    int cellBase;           /* first cell of the level inside a frame's cell array        */
    int slotBase;           /* first u32 of the level inside a frame's candidate slots    */
    int cellCap;            /* u32 entries per cell slot                                  */
    int quota;              /* mnFeaturesPerLevel[level]                                  */
    int nIni;               /* initial quadtree nodes (1..ORBX_MAX_INI)                   */
    int iniX[ORBX_MAX_INI + 1]; /* their x bounds                                         */
    int binOff;             /* offset of this level's x -> initial-node table (u8)        */
    int kpBase, kpCap;      /* slice of the per-frame level-keypoint array                */
    int blurTileBase, blurTilesX, blurTilesY;
    double rsScaleX, rsScaleY; /* cv::resize inverse scale from level l-1: 1.0 / ((double)w / w_prev)  */
    int rsColOff, rsRowOff; /* u32 offsets of this level's cv::resize tables (levels >= 1), see build_resize_tables */
    int patchSize;          /* (int)(PATCH_SIZE*scale), src/ORBextractor.cc:1175          */
    float scale;            /* mvScaleFactor[level]                                       */
};

struct OrbxGeom {
    int nlevels, W, H;
    int iniTh, minTh;
    int cellsPerFrame, slotsPerFrame, kpPerFrame, outCap;
    int blurTiles;
    int fcPitch, fcScPitch, fcNS;   /* k_fast_cells template arguments: LDS row pitch of the window (48 / 64 / 80) and of the score tile (48 / 80), 16-byte window units per lane (2 / 3 / 4 / 6) */
    int fcInBytes, fcScBytes, fcLdsBytes;   /* LDS carve-up of k_fast_cells: input window, score tile, total */
    size_t pyrBytes;        /* bytes of levels 1.. of one frame */
    uint32_t taps[7];
    int umax[16];
    OrbxLevel lv[ORBX_MAX_LEVELS];
};

/* Everything k_fast_cells needs to know about one 30-px cell (src/ORBextractor.cc:1089-1123), built with the geometry: one 32-byte scalar load per
 * cell instead of a chain of dependent loads through OrbxGeom and three integer divisions. */
struct OrbxFcCell {
    uint32_t xy;            /* x0 | y0 << 16: first pixel of the cell's detection area (= iniX + 3, iniY + 3), level coordinates          */
    uint32_t dim;           /* aw | ah << 8 | level << 16 | valid << 24: size of the detection area; valid = 0: the reference skips the cell (1 x 1 stand-in) */
    int32_t pitch;          /* row pitch of the level (the pyramid block's; level 0 pixels are read at the caller's stride)                 */
    uint32_t off;           /* byte offset of the level inside a frame's pyramid block                                                     */
    uint32_t slot;          /* first u32 of the cell inside a frame's candidate slots                                                      */
    uint32_t cap;           /* u32 entries of the cell's slot                                                                              */
    uint32_t inv;           /* ceil(2^16 / nu), nu = 16-byte units per window row = (aw + 6 + 15) / 16: u * inv >> 16 == u / nu for u < 2^12 */
    uint32_t units;         /* nu | (window rows * nu) << 8                                                                                */
};
#define ORBX_FC_CELLS_PER_WAVE 4

/* One workgroup's share of one pyramid level in k_pyramid_tiles (single-frame call): the rectangle it COMPUTES (what its part of
 * the next level reads, united with what it owns; x bounds multiples of 4) and the rectangle it OWNS (stores to the pyramid buffer).
 * Level 0: only the computed rectangle, = the window of the input image it stages. */
struct OrbxPyrTile { short cx0, cy0, cx1, cy1, ox0, oy0, ox1, oy1; };

/* One member of a COMBINED single-frame batch (orbx_extractor.hip: the combiner): where the frame comes from and where its results go.
 * The table lives in pinned host memory; k_comb_upload / k_comb_finish read it, so the graph that replays a batch of n frames
 * names no member and serves whichever n callers happen to arrive together. */
struct OrbxCombMember {
    const uint8_t *srcImg;    /* the frame's rows at the device pitch: the member's pinned staging buffer (read across PCIe), or its
                                 device copy when the member's own upload - started while it waited for the engine - is complete      */
    uint8_t *devImg;          /* member's device copy of the frame (level 0 for the device-resident consumers) to fill, NULL = the
                                 member uploads (has uploaded) it itself                                                             */
    uint8_t *devPyr;          /* member's device pyramid, levels >= 1                                                */
    uint8_t *devArena;        /* member's result arena in the one-frame layout: count | 2 status words | kps | desc  */
    uint8_t *hostOut;         /* the same arena in the member's pinned memory                                        */
    uint8_t *hostPyr;         /* pinned copy of the pyramid (levels >= 1, device layout), NULL = not wanted          */
};

/* level-coordinates keypoint produced by the quadtree + orientation stages */
/* ca / sb = cos / sin of the angle as the reference's libm rounds them; k_orient_describe fills them with the angle (read back by the stage taps only) */
struct OrbxLevelKp { uint16_t x, y; uint8_t score, pad[3]; float angle, ca, sb; };

void orbx_set_error(const char *fmt, ...);
#define ORBX_HIP_CHECK(expr)                                                                           \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            orbx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return ORBX_ERR_HIP;                                                                       \
        }                                                                                              \
    } while (0)

/* kernel launchers (orbx_kernels.hip) */
struct OrbxLaunch {
    hipStream_t stream;
    const OrbxGeom *geomDev;
    const OrbxGeom *geom;         /* host copy */
    int batch;
    const uint8_t *img0;          /* device: level 0 of frame f at img0 + f*img0FramePitch */
    int img0Stride;
    size_t img0FramePitch;
    uint8_t *pyr, *blur, *score;  /* device: per frame pyrBytes / blurBytes / scoreBytes    */
    size_t blurBytes;             /* blurred copy of ALL levels (level 0 included)          */
    const uint8_t *binTab;
    const uint32_t *rsTab;        /* cv::resize index / coefficient tables of all levels */
    const OrbxFcCell *fcCells;    /* k_fast_cells: one entry per cell of a frame */
    int *cellCount;
    uint32_t *cellSlots;
    uint32_t *ptBuf, *labBuf;     /* quadtree: candidates of a level in list order and their node labels; slotsPerFrame u32 per frame each */
    OrbxLevelKp *lvlKp;
    int *lvlCnt;                  /* nlevels per frame */
    int *outBase;                 /* nlevels per frame: first output index of each level's keypoints (k_blur -> k_orient_describe) */
    orbx_keypoint *outKp;
    uint8_t *outDesc;
    int *outCnt;
    int *status;                  /* per frame error bits (scratch of the running batch; [batch] = OR over the batch) */
    int *outStatus;               /* snapshot of `status` in the result buffer (written by k_orient_describe, guarded like the results) */
    int nodeCap;                  /* 256 / 512 / 1024 / 2048 */
    /* graph construction (single-frame call): when `graph` is set, a launcher adds a kernel node that depends on deps[0..ndeps)
     * and returns it in *node instead of launching on `stream` */
    const OrbxCombMember *combTab = nullptr;      /* combined single-frame batches: member table (pinned host memory) */
    OrbxCombMember *combTabCopy = nullptr;        /* ... its copy in device memory: written by k_comb_upload, read by k_comb_finish */
    unsigned *combSync = nullptr;                 /* ... device: [0] arrivals of k_comb_finish's workgroups, [2..3] launch sets completed so far (u64) */
    unsigned long long *combFlag = nullptr;       /* ... mapped host: the same count, stored behind a set's last result - what the leader polls instead of synchronising the stream */
    size_t combKpOff = 0, combDescOff = 0;        /* one-frame arena layout of the members */
    const OrbxPyrTile *pyrTiles = nullptr; int pyrTileCount = 0, pyrTileBuf = 0, pyrTileTab = 0;   /* k_pyramid_tiles plan (single-frame call): bytes of one of its two LDS image buffers, bytes of its LDS table slices */
    hipGraph_t graph = nullptr;
    hipGraphNode_t deps[2] = {nullptr, nullptr};
    int ndeps = 0;
    hipGraphNode_t *node = nullptr;
};

int orbx_launch_resize(const OrbxLaunch &L, int level);
int orbx_launch_pyramid_tiles(const OrbxLaunch &L);   /* all levels of a frame in one launch (L.pyrTiles); single frames and small combined batches */
int orbx_launch_comb_upload(const OrbxLaunch &L, uint8_t *stagingDev);   /* members' pinned frames -> the engine's staging area (L.combTab) */
int orbx_launch_comb_finish(const OrbxLaunch &L);     /* the engine's results / pyramid / frames -> every member's device and pinned buffers */
int orbx_launch_fast_cells(const OrbxLaunch &L);   /* FAST score + cell NMS + emission; L.score (parity tap) may be NULL */
int orbx_launch_octree(const OrbxLaunch &L);
int orbx_launch_blur(const OrbxLaunch &L);
int orbx_launch_octree_blur(const OrbxLaunch &L, bool *fused);   /* combined single-frame calls: quadtree + host pyramid copy + blur in one launch */
/* frames in pinned / registered host memory -> the device input layout, read in place by the kernel (pipelined host batches) */
int orbx_launch_gather_frames(hipStream_t stream, const uint8_t *const *framesPinnedTab, int batch, int W, int H, int srcStride, uint8_t *dst, int dstStride, size_t framePitch, bool aligned16);
int orbx_launch_orient_describe(const OrbxLaunch &L);   /* IC_Angle + rBRIEF + final KeyPoint in one pass (after the blur) */

/* stream of an extractor handle (orbx_extractor.hip), so other handles can order work after it */
hipStream_t orbx_extractor_stream_internal(orbx_extractor *h);
bool orbx_extractor_host_complete_internal(orbx_extractor *h, int *status);
int orbx_extractor_host_count_internal(orbx_extractor *h);      // keypoints of frame 0 of a host-complete call (its pinned result arena)
/* `ev` (recorded by a consumer on its own stream) guards the result buffer of the LAST batch: the
 * extractor waits for it before that buffer is overwritten two batches later */
void orbx_extractor_set_consumer_event_internal(orbx_extractor *h, hipEvent_t ev);
/* device word holding the OR of the capacity bits of the producer's last batch (nullptr before the first batch) */
const int *orbx_extractor_status_word_internal(orbx_extractor *h);


/* Grow-only device array owned by a handle (hipMalloc on demand, never shrinks, released by the handle's destroy). */
template <typename T> struct OrbxDevBuf {
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

/* Host arrays of one call packed into ONE pinned buffer and moved with ONE copy: a dozen small
 * hipMemcpyAsync calls from pageable memory cost more than the kernels of a single-frame call.
 * begin(total) sizes both sides, put() copies a host array in and returns where it will live on the
 * device, flush() issues the copy.  The caller synchronises the stream before the next begin()
 * (every host-array entry point ends with a synchronous download). */
struct OrbxHostStage {
    uint8_t *host = nullptr;
    size_t hostBytes = 0, used = 0;
    OrbxDevBuf<uint8_t> dev;
    static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    int begin(size_t total)
    {
        used = 0;
        int rc = dev.ensure(total ? total : 256);
        if (rc != ORBX_OK) return rc;
        if (total > hostBytes) {
            if (host) (void)hipHostFree(host);
            host = nullptr; hostBytes = 0;
            ORBX_HIP_CHECK(hipHostMalloc((void **)&host, total, hipHostMallocDefault));
            hostBytes = total;
        }
        return ORBX_OK;
    }
    template <typename T> T *put(const T *src, size_t count)
    {
        T *d = (T *)(dev.p + used);
        if (src && count) memcpy(host + used, src, count * sizeof(T));
        used += padded(count * sizeof(T));
        return d;
    }
    int flush(hipStream_t st)
    {
        if (used) ORBX_HIP_CHECK(hipMemcpyAsync(dev.p, host, used, hipMemcpyHostToDevice, st));
        return ORBX_OK;
    }
    void release() { if (host) (void)hipHostFree(host); host = nullptr; hostBytes = 0; used = 0; dev.release(); }
};

/* One small synchronous host call WITHOUT a copy engine and WITHOUT a stream synchronisation.  Measured around an 11 us kernel
 * (tools/ubench_call.hip, profiles/r06_call_floor.txt): pageable upload + hipStreamSynchronize + two hipMemcpy downloads = 63 us; inputs in MAPPED pinned
 * memory read in place by the kernel, results written by the kernel into mapped pinned memory, a sequence word raised behind them by the last workgroup
 * and polled by the host = 22 us.  (hipMemcpyAsync from pinned memory alone costs 17 us of stream time before the kernel starts.)
 *   begin(in, out) sizes the two mapped buffers (grow-only) and waits for a call that was abandoned half way;
 *   put(src, n)    copies a host array into the input buffer and returns the address the DEVICE reads it at (each byte should be read once: inputs that
 *                  several workgroups read again and again are copied to device memory by the first kernel of the chain);
 *   outDev / outHost(off) address the result buffer from both sides;
 *   arm() hands out the sequence number the chain's last kernel publishes with orbx_publish(); wait() spins until it shows up. */
struct OrbxCallBox {
    uint8_t *in = nullptr, *inDev = nullptr;
    size_t inBytes = 0, used = 0;
    uint8_t *out = nullptr, *outDevP = nullptr;
    size_t outBytes = 0;
    unsigned long long *flag = nullptr, *flagDev = nullptr;      /* mapped: the last published sequence number */
    unsigned *counter = nullptr;                                 /* device: arrivals of the publishing kernel's workgroups (0 between calls) */
    unsigned long long seq = 0;
    bool pending = false;
    static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    int grow(uint8_t **host, uint8_t **dev, size_t *have, size_t want)
    {
        if (want <= *have) return ORBX_OK;
        if (*host) (void)hipHostFree(*host);
        *host = nullptr; *dev = nullptr; *have = 0;
        want = padded(want + want / 2);
        ORBX_HIP_CHECK(hipHostMalloc((void **)host, want, hipHostMallocMapped));
        ORBX_HIP_CHECK(hipHostGetDevicePointer((void **)dev, *host, 0));
        *have = want;
        return ORBX_OK;
    }
    int begin(size_t inTotal, size_t outTotal, hipStream_t st)
    {
        if (pending) { ORBX_HIP_CHECK(hipStreamSynchronize(st)); pending = false; }      /* (a call that returned an error before its wait()) */
        used = 0;
        int rc;
        if ((rc = grow(&in, &inDev, &inBytes, inTotal ? inTotal : 256)) != ORBX_OK) return rc;
        if ((rc = grow(&out, &outDevP, &outBytes, outTotal ? outTotal : 256)) != ORBX_OK) return rc;
        if (!flag) {
            ORBX_HIP_CHECK(hipHostMalloc((void **)&flag, 64, hipHostMallocMapped));
            ORBX_HIP_CHECK(hipHostGetDevicePointer((void **)&flagDev, flag, 0));
            *flag = 0;
            ORBX_HIP_CHECK(hipMalloc((void **)&counter, 64));
            ORBX_HIP_CHECK(hipMemset(counter, 0, 64));
        }
        return ORBX_OK;
    }
    template <typename T> const T *put(const T *src, size_t count)
    {
        const T *d = (const T *)(inDev + used);
        if (src && count) memcpy(in + used, src, count * sizeof(T));
        used += padded(count * sizeof(T));
        return d;
    }
    template <typename T> T *hostIn(const T *devAddr) { return (T *)(in + ((const uint8_t *)devAddr - inDev)); }      /* where put() placed an array, host side */
    template <typename T> T *outDev(size_t off) { return (T *)(outDevP + off); }
    template <typename T> const T *outHost(size_t off) const { return (const T *)(out + off); }
    unsigned long long arm() { pending = true; return ++seq; }
    /* spins on the sequence word; the stream is queried now and then so that a failed launch becomes an error instead of a hang */
    int wait(hipStream_t st)
    {
        const unsigned long long want = seq;
        for (unsigned spins = 1;; spins++) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) >= want) break;
            if ((spins & 0x3fff) == 0) {
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess) {
                    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) >= want) break;
                    pending = false;
                    orbx_set_error("the stream drained without the call's results");
                    return ORBX_ERR_HIP;
                }
                if (q != hipErrorNotReady) { pending = false; orbx_set_error("%s", hipGetErrorString(q)); return ORBX_ERR_HIP; }
            }
            __builtin_ia32_pause();
        }
        pending = false;
        return ORBX_OK;
    }
    void release()
    {
        if (in) (void)hipHostFree(in);
        if (out) (void)hipHostFree(out);
        if (flag) (void)hipHostFree(flag);
        if (counter) (void)hipFree(counter);
        in = out = nullptr; flag = nullptr; counter = nullptr; inBytes = outBytes = 0;
    }
};

#ifdef __HIPCC__
/* End of the LAST kernel of an OrbxCallBox chain, called by every thread of every workgroup after its last store of results (to mapped host memory):
 * the workgroup that arrives last raises the sequence word with system scope behind everybody's stores (agent-scope release per workgroup, acquire +
 * system-scope release by the last one; a per-thread __threadfence_system costs 3 us more per call, profiles/r06_call_floor.txt). */
__device__ __forceinline__ void orbx_publish(unsigned *counter, unsigned long long *flag, unsigned long long seq, unsigned nblocks)
{
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (nblocks <= 1 || __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            if (nblocks > 1) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
/* The inputs of a single host call that the kernels behind read MANY times: copied ONCE from mapped pinned memory into the handle's device arena
 * (uint4 units; ~25 GB/s across PCIe: 64 KB in ~4 us - a hipMemcpyAsync from the same pinned buffer costs 17 us of stream time). */
static __global__ __launch_bounds__(256) void k_stage_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
#endif

/* Device view of the LAST batch of an extractor: results + the unblurred pyramid (what the
 * reference keeps in ORBextractor::mvImagePyramid, read by Frame::ComputeStereoMatches). */
struct OrbxLastBatchView {
    int batch, nlevels, cap;
    const orbx_keypoint *kp;     /* [batch*cap]            */
    const uint8_t *desc;         /* [batch*cap*32]         */
    const int32_t *counts;       /* [batch]                */
    const uint8_t *img0;         /* level 0 = caller's input frames */
    int img0Stride;
    size_t img0FramePitch;
    const uint8_t *pyr;          /* levels >= 1, frame f at pyr + f*pyrBytes + lv[l].off, pitch lv[l].pitch */
    size_t pyrBytes;
    const OrbxGeom *geomDev;
    const OrbxGeom *geom;        /* host copy */
    const float *scale, *invScale; /* host tables, nlevels entries */
};
int orbx_extractor_last_batch_view_internal(orbx_extractor *h, OrbxLastBatchView *v);
/* `ev` guards the PYRAMID of the last batch (single buffered): the next batch waits for it before
 * its first kernel */
void orbx_extractor_set_pyramid_consumer_event_internal(orbx_extractor *h, hipEvent_t ev);

#endif
