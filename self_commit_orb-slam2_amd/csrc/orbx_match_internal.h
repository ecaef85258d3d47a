// orbx_match_internal.h -- shared by orbx_match.hip (SearchByBoW, SearchForTriangulation, stereo) and orbx_match_proj.hip
// (the projection / area searches): constants, device-side views and helpers, and the matcher handle.
#ifndef ORBX_MATCH_INTERNAL_H
#define ORBX_MATCH_INTERNAL_H

#include "orbx_internal.h"

#define TH_HIGH 100       /* src/ORBmatcher.cc:49 */
#define TH_LOW 50         /* :50 */
#define HISTO_LENGTH 30   /* :51 */
#define KEY_EMPTY 0xffffffffu
#define TOPK 8            /* candidates kept per KeyFrame feature; the exact rescan handles overflow */
#define MATCH_PROF_RING 64

#define KEY64_EMPTY 0xffffffffffffffffull
#define GRID_COLS 64   /* include/Frame.h:60 */
#define GRID_ROWS 48   /* include/Frame.h:55 */

struct FeatDev {   // orbx_feature_set with device pointers
    const orbx_keypoint *kp;
    const uint8_t *desc;
    const int32_t *counts;
    const int32_t *groups;
    const uint8_t *valid;
    int cap;
};

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t u = __shfl_xor(v, o); v = u < v ? u : v; }
    return v;
}

__device__ __forceinline__ int hamming256(const unsigned long long a[4], unsigned long long b0, unsigned long long b1, unsigned long long b2,
                                          unsigned long long b3)
{
    return __popcll(a[0] ^ b0) + __popcll(a[1] ^ b1) + __popcll(a[2] ^ b2) + __popcll(a[3] ^ b3);
}

// ComputeThreeMaxima (src/ORBmatcher.cc:1866-1908) of a rotation histogram in LDS, by one wave with a bin per lane: the scan's strict comparisons in
// ascending bin order pick the three largest counts, equal counts in bin order, an empty bin never - i.e. the three largest keys count << 5 | (31 - bin)
// among the non-empty bins.  (Every thread walking the thirty bins itself is ~360 instructions; with sixteen waves on one CU that was 3.5 us of the
// replay kernels.)  Every wave of a workgroup may call it: same result.
__device__ __forceinline__ void three_maxima_wave(const int *hist, int lane, int &ind1, int &ind2, int &ind3)
{
    const int c = lane < HISTO_LENGTH ? hist[lane] : 0;
    uint32_t key = c > 0 ? ((uint32_t)c << 5) | (uint32_t)(31 - lane) : 0u;
    uint32_t top[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t v = key;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t u = __shfl_xor(v, o); v = u > v ? u : v; }
        top[k] = v;
        if (key == v) key = 0u;
    }
    const int max1 = (int)(top[0] >> 5), max2 = (int)(top[1] >> 5), max3 = (int)(top[2] >> 5);
    ind1 = top[0] ? 31 - (int)(top[0] & 31u) : -1; ind2 = top[1] ? 31 - (int)(top[1] & 31u) : -1; ind3 = top[2] ? 31 - (int)(top[2] & 31u) : -1;
    if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
}

struct ProjFrameDev { const orbx_keypoint *kp; const uint8_t *desc; const float *uRight; const uint8_t *occupied; const int32_t *counts; int cap;
                      float minX, minY, gwInv, ghInv; };

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long u = __shfl_xor(v, o); v = u < v ? u : v; }
    return v;
}

struct orbx_matcher {
    int device = 0, maxFeatures = 0, maxPairs = 0;
    hipStream_t stream = nullptr;
    hipEvent_t evDep = nullptr, evDone[2] = {nullptr, nullptr};
    int doneSlot = 0;
    hipEvent_t ev0[MATCH_PROF_RING] = {}, ev1[MATCH_PROF_RING] = {};   // one pair per *_device call (ring)
    hipEvent_t evMid[MATCH_PROF_RING] = {};                             // SearchByBoW: between the distance kernel and the greedy replay
    bool midValid[MATCH_PROF_RING] = {};
    float lastDistanceMs = 0.f, lastReplayMs = 0.f;
    int profCount = 0;
    OrbxDevBuf<int32_t> pairsA, pairsB, order, matches, dists, nmatches;
    OrbxDevBuf<uint32_t> topk;
    OrbxDevBuf<float> scales, uright, depth, pf[2];          // pf: projection staging (floats)
    OrbxDevBuf<unsigned long long> topk64;
    OrbxDevBuf<uint8_t> pb[2];
    OrbxDevBuf<int32_t> pi32[2];
    OrbxDevBuf<orbx_keypoint> pkp;
    OrbxDevBuf<uint32_t> projDec, projQueue;   // k_proj_greedy: chosen feature per map point, rescan queue
    OrbxHostStage hostStage;     // host-array entry points: all inputs of a call in one pinned buffer, one copy
    OrbxCallBox box;             // single-call host entry points of the tracking thread: mapped pinned inputs / results + sequence word (no copy engine, no stream sync)
    OrbxDevBuf<uint8_t> arena;   // ... their inputs on the device, copied there once by k_stage_copy (same offsets as in box.in)
    OrbxDevBuf<int32_t> sad;
    OrbxDevBuf<int32_t> stRowStart, stRowList;   // ComputeStereoMatches: vRowIndices of the right frames (k_stereo_rows)
    hipEvent_t evDep2 = nullptr, evPyr[2] = {nullptr, nullptr};
    int lastStereoPairs = 0;
    // orbx_stereo_frame (one stereo frame, host-synchronous): scale tables as last uploaded, a device zero for the pair arrays, pinned results
    float sfScales[128] = {};
    int sfLevels = 0;
    OrbxDevBuf<int32_t> sfZero;
    OrbxDevBuf<float> sfScalesDev;               // its own copy of the tables: the other calls overwrite `scales`
    float *sfHost = nullptr, *sfHostDev = nullptr;   // pinned: uright | depth, written by the kernels across PCIe
    size_t sfHostFloats = 0;
    int sfSeq = 0, sfFlagSeq = 0;                 // k_stereo_one: call counter; the value the kernel of the pending call writes into the pinned completion word (0: none)
    int sfPending = 0;                            // orbx_stereo_frame_begin .. _end: 0 none, 1 launched (results land in sfHost), 2 the general path ran (download at the end)
    // staging for the host-array convenience calls
    OrbxDevBuf<orbx_keypoint> hk[2];
    OrbxDevBuf<uint8_t> hd[2], hv[2];
    OrbxDevBuf<int32_t> hc[2], hg[2];
    OrbxDevBuf<uint8_t> hs[2];                               // SearchForTriangulation: stereo flags (host form)
    OrbxDevBuf<float> triGeom;                               // F12 + epipole per pair
    // Frame::isInFrustum results (proj_x | proj_y | proj_xr | view_cos, level, in_view), orbx_projection_points layout
    OrbxDevBuf<float> frProj;
    OrbxDevBuf<int32_t> frLevel;
    OrbxDevBuf<uint8_t> frInView;
    size_t frCount = 0;
    int lastPairs = 0, lastStride = 0;
    OrbxDevBuf<int> producerStatus;   // OR of the capacity words of the extractors this handle's calls were chained behind
};

#define MLAUNCH_CHECK()                                                                                              \
    do {                                                                                                             \
        hipError_t e_ = hipGetLastError();                                                                           \
        if (e_ != hipSuccess) { orbx_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); return ORBX_ERR_HIP; } \
    } while (0)

// host helpers defined in orbx_match.hip
namespace orbx_match {
int prep_pairs(orbx_matcher *m, const orbx_feature_set *a, const orbx_feature_set *b, const int32_t *pa, const int32_t *pb, int npairs, orbx_extractor *after);
int chain_back(orbx_matcher *m, orbx_extractor *after);
int inherit_status(orbx_matcher *m, orbx_extractor *after, bool first = true);
int check_producer_status(orbx_matcher *m);
FeatDev to_dev(const orbx_feature_set *s);
/* stage one frame of host features into the matcher's staging buffers (side 0 / 1) */
int stage_host(orbx_matcher *m, int side, const orbx_feature_set *h, orbx_feature_set *d);
}  // namespace orbx_match

#endif
