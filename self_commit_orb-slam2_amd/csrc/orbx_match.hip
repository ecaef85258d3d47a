// orbx_match.hip -- 256-bit Hamming matchers on gfx950 (wave64, __ballot/__popcll).
//
//   orbx_search_by_bow_device  == ORBmatcher::SearchByBoW, both overloads
//                                 (reference src/ORBmatcher.cc:230-382, 656-799)
//   orbx_stereo_match_device   == Hamming stage of Frame::ComputeStereoMatches
//                                 (reference src/Frame.cc:1041-1216)
//
// SearchByBoW is a greedy, order dependent assignment: KeyFrame features are visited in
// (node id, feature index) order and a Frame feature that has been taken is skipped by
// every later KeyFrame feature (:288, :717).  The O(N1*N2) part - all descriptor
// distances - does not depend on that order, so it runs fully parallel and leaves, per
// KeyFrame feature, its 8 best candidates by (distance, index).  One wave per pair then
// replays the greedy pass in the reference order over those short lists; whenever fewer
// than two of the eight are still free (and the list was full) it falls back to an exact
// wave-parallel rescan.  The result is index-exact, including first-minimum-wins ties.
// VALU/LDS bound (XOR + v_bcnt_u32), not HBM: 144 KB of traffic per 2000x2000 pair.
#include <math.h>
#include <string.h>

#include <atomic>
#include <chrono>

#include <type_traits>
#include <vector>

#include "orbx_match_internal.h"

using namespace orbx_match;

namespace {

// processing order of the KeyFrame features: ascending (node id, feature index) = std::map
// iteration over the FeatureVector, features ascending inside a node.
__global__ __launch_bounds__(256) void k_bow_order(FeatDev A, const int32_t *__restrict__ pairsA, int32_t *__restrict__ order, int stride)
{
    const int p = blockIdx.y, fa = pairsA[p];
    const int nA = min(A.counts[fa], A.cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nA) return;
    int32_t *ord = order + (size_t)p * stride;
    if (!A.groups) { ord[i] = i; return; }
    const int32_t *g = A.groups + (size_t)fa * A.cap;
    const int gi = g[i];
    int rank = 0;
    for (int k = 0; k < nA; k++) { int gk = g[k]; rank += (gk < gi) || (gk == gi && k < i); }
    ord[rank] = i;
}

// top-TOPK candidates of every A feature by key = dist<<16 | j over the B features of the same
// node (mode 1: that also carry a valid MapPoint).  LANE = TWO A features (descriptors in 16
// VGPRs); the B descriptors are staged through LDS in tiles of 1024 and read with wave-uniform
// (broadcast) ds_read_b128, so one LDS fetch feeds 128 distances and a distance is
// 8 x (v_xor_b32 + v_bcnt_u32_b32) + key + min.
// Distance cut-off: a candidate at distance d >= dcut can never be accepted as best
// (d > TH_LOW) and, as second best, can never fail the ratio test of an acceptable best
// (nnratio * d > TH_LOW >= best, orbx_search_by_bow_device computes dcut in the float arithmetic
// of the test); dropping it leaves the greedy replay bit-identical and makes list updates rare.
// The per-lane lists start filled with the sentinel dcut<<16, so "key < kk[TOPK-1]" is the whole
// test, evaluated once per 4 B features on the minimum of the keys.
#define TOPK_NROW 1      /* A features per lane in k_bow_topk (1: 4096 waves per 256 pairs x 1000 features instead of 2048) */
#define TOPK_ROWS (256 * TOPK_NROW)   /* A features per block */
#define TOPK_TILE 1024  /* B features per LDS tile */

__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:810-1017): the per-candidate geometry.
struct TriDev {
    const float *f12;      // [9] per pair, row-major
    const float *epipole;  // [2] per pair: ex, ey (:824-826)
    const uint8_t *stereoA, *stereoB;   // mvuRight >= 0, laid out like the feature sets; NULL = monocular
    float epiTh[ORBX_MAX_LEVELS];       // 100*pKF2->mvScaleFactors[octave] (:893)
    double chiTh[ORBX_MAX_LEVELS];      // 3.84*pKF2->mvLevelSigma2[octave] (:224), a double product in the reference
};

// the epipole gate (:888-895) and CheckDistEpipolarLine (:188-227), float operation order of the reference
__device__ __forceinline__ bool tri_geom_ok(const TriDev &T, int p, const orbx_keypoint &k1, bool st1, const orbx_keypoint &k2, bool st2)
{
    if (!st1 && !st2) {
        const float distex = T.epipole[2 * p] - k2.x, distey = T.epipole[2 * p + 1] - k2.y;
        if (distex * distex + distey * distey < T.epiTh[k2.octave]) return false;
    }
    const float *F = T.f12 + 9 * p;
    const float a = k1.x * F[0] + k1.y * F[3] + F[6];
    const float b = k1.x * F[1] + k1.y * F[4] + F[7];
    const float c = k1.x * F[2] + k1.y * F[5] + F[8];
    const float num = a * k2.x + b * k2.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return (double)dsqr < T.chiTh[k2.octave];
}

// insert into an ascending list, dropping its largest element: slot q of the new list is the median of its old
// neighbours kk[q-1], kk[q] and the key (one v_med3_u32 per slot, all independent, no compares or selects).
// A key >= kk[TOPK-1] leaves the list unchanged, so no predication is needed.
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ void topk_insert(uint32_t (&kk)[TOPK], uint32_t key)
{
    // in place, last slot first: slot q only needs the OLD kk[q - 1], which is updated after it
#pragma unroll
    for (int q = TOPK - 1; q > 0; q--) kk[q] = med3_u32(kk[q - 1], kk[q], key);
    kk[0] = min(kk[0], key);
}

// TRI (SearchForTriangulation): among equal distances the LAST candidate in scan order wins
// (`dist>bestDist` skips, :880), so the key carries 0xffff - j; a candidate below the list threshold
// is inserted only if it passes the epipole gate and the epipolar-line test.
template <bool FILTER, bool TRI, int NROW>   // FILTER: B side has node ids and/or a validity mask; NROW: A features per lane (2: fewer LDS reads per distance, 1: twice the waves)
__global__ __launch_bounds__(256) void k_bow_topk(FeatDev A, FeatDev B, const int32_t *__restrict__ pairsA, const int32_t *__restrict__ pairsB, int mode,
                                                  uint32_t dcut, uint32_t *__restrict__ topk, int stride, TriDev T)
{
    __shared__ uint4 sB[TOPK_TILE * 2];     // 32-byte descriptors
    __shared__ int32_t sG[TOPK_TILE];       // node id, 0x80000000 = excluded (FILTER only)
    const int p = blockIdx.y, fa = pairsA[p], fb = pairsB[p];
    const int nA = min(A.counts[fa], A.cap), nB = min(B.counts[fb], B.cap);
    const int capB = B.cap;
    const int row0 = blockIdx.x * (256 * NROW);
    if (row0 >= nA) return;
    const int tid = threadIdx.x;
    int ir[NROW];
    bool live[NROW], act[NROW];
    uint32_t a[NROW][8], kk[NROW][TOPK], th[NROW];
    int gA[NROW];
    const uint32_t sentinel = dcut << 16;
#pragma unroll
    for (int r = 0; r < NROW; r++) {
        ir[r] = row0 + 256 * r + tid;
        live[r] = ir[r] < nA;
        const size_t ia = (size_t)fa * A.cap + (live[r] ? ir[r] : nA - 1);
        const uint32_t *da = (const uint32_t *)(A.desc + ia * 32);
#pragma unroll
        for (int w = 0; w < 8; w++) a[r][w] = da[w];
        gA[r] = (FILTER && A.groups) ? A.groups[ia] : 0;
        act[r] = live[r] && !(FILTER && gA[r] < 0);      // unfiled A features keep empty lists
#pragma unroll
        for (int q = 0; q < TOPK; q++) kk[r][q] = act[r] ? sentinel : 0u;   // inactive rows never insert
        th[r] = kk[r][TOPK - 1] >> 16;                   // distance part of the list's threshold
    }
    const uint4 *gD = (const uint4 *)(B.desc + (size_t)fb * capB * 32);
    for (int t0base = 0; t0base < nB; t0base += TOPK_TILE) {
        const int nt = min(TOPK_TILE, nB - t0base), ntPad = (nt + 3) & ~3;
        __syncthreads();
        for (int t = tid; t < 2 * ntPad; t += 256) sB[t] = t < 2 * nt ? gD[2 * (size_t)t0base + t] : make_uint4(0u, 0u, 0u, 0u);
        if (FILTER)
            for (int j = tid; j < ntPad; j += 256) {
                int gq = (int)0x80000000;
                if (j < nt) {
                    gq = B.groups ? B.groups[(size_t)fb * capB + t0base + j] : 0;
                    if (gq < 0) gq = (int)0x80000000;   // negative node id = not filed in the FeatureVector: never matched
                    if (mode >= 1 && B.valid && !B.valid[(size_t)fb * capB + t0base + j]) gq = (int)0x80000000;
                }
                sG[j] = gq;
            }
        __syncthreads();
        // four B features per step; only the last step of a tile can contain padding (TAIL): its check stays out of the main loop
        auto group = [&](const int j0, auto tailTag) {
            constexpr bool TAIL = decltype(tailTag)::value;
            // the event test runs on the bare distances (the list threshold's distance th[r]): the key
            // dist << 16 | j is only formed for the few candidates that reach the list.  An excluded
            // candidate carries distance 0xffff, above every threshold (dcut <= 257).
            uint32_t dd[NROW][4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint4 lo = sB[2 * (j0 + u)], hi = sB[2 * (j0 + u) + 1];   // wave-uniform address: LDS broadcast
                const uint32_t bw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                uint32_t d[NROW];
                // v_bcnt_u32_b32 adds its second operand: one instruction per word instead of bcnt + add tree
#pragma unroll
                for (int r = 0; r < NROW; r++) d[r] = bcnt_acc(a[r][0] ^ bw[0], 0u);
#pragma unroll
                for (int w = 1; w < 8; w++)
#pragma unroll
                    for (int r = 0; r < NROW; r++) d[r] = bcnt_acc(a[r][w] ^ bw[w], d[r]);
                if (FILTER) {
                    const int gq = sG[j0 + u];
#pragma unroll
                    for (int r = 0; r < NROW; r++) d[r] = gq == gA[r] ? d[r] : 0xffffu;
                } else if (TAIL && j0 + u >= nt) {   // wave-uniform (zero padding of the tile)
#pragma unroll
                    for (int r = 0; r < NROW; r++) d[r] = 0xffffu;
                }
#pragma unroll
                for (int r = 0; r < NROW; r++) dd[r][u] = d[r];
            }
            // plain scan order: an equal distance with a later j has the larger key, so only d < t can enter;
            // TRI keys carry 0xffff - j, an equal distance with a later j enters: d <= t (conservative at the sentinel)
            auto reaches = [](uint32_t d, uint32_t t) { return TRI ? d <= t : d < t; };
            bool any4 = false;
#pragma unroll
            for (int r = 0; r < NROW; r++) any4 = any4 || reaches(min(min(dd[r][0], min(dd[r][1], dd[r][2])), dd[r][3]), th[r]);
            if (__any(any4)) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    bool anyU = false;
#pragma unroll
                    for (int r = 0; r < NROW; r++) anyU = anyU || reaches(dd[r][u], th[r]);
                    if (__any(anyU)) {
                        const uint32_t j = TRI ? 0xffffu - (uint32_t)(t0base + j0 + u) : (uint32_t)(t0base + j0 + u);
                        uint32_t ins[NROW];
#pragma unroll
                        for (int r = 0; r < NROW; r++) ins[r] = (dd[r][u] << 16) | j;
                        if (TRI) {
                            const int jb = t0base + j0 + u;
                            const orbx_keypoint k2 = B.kp[(size_t)fb * capB + jb];
                            const bool st2 = T.stereoB && T.stereoB[(size_t)fb * capB + jb];
#pragma unroll
                            for (int r = 0; r < NROW; r++) {
                                bool pass = false;
                                if (ins[r] < kk[r][TOPK - 1]) {
                                    const size_t ia = (size_t)fa * A.cap + ir[r];
                                    pass = tri_geom_ok(T, p, A.kp[ia], T.stereoA && T.stereoA[ia], k2, st2);
                                }
                                ins[r] = pass ? ins[r] : 0xffffffffu;
                            }
                        }
                        // events are sparse (a few lanes per wave): usually only one of the rows has one
#pragma unroll
                        for (int r = 0; r < NROW; r++)
                            if (__any(ins[r] < kk[r][TOPK - 1])) { topk_insert(kk[r], ins[r]); th[r] = kk[r][TOPK - 1] >> 16; }
                    }
                }
            }
        };
        int j0 = 0;
        for (; j0 + 4 <= nt; j0 += 4) group(j0, std::false_type());
        if (j0 < ntPad) group(j0, std::true_type());
    }
#pragma unroll
    for (int r = 0; r < NROW; r++)
        if (live[r]) {
            uint32_t *out = topk + ((size_t)p * stride + ir[r]) * TOPK;
#pragma unroll
            for (int k = 0; k < TOPK; k++) out[k] = (!act[r] || kk[r][k] >= sentinel) ? KEY_EMPTY : kk[r][k];
        }
}

// ---------------------------------------------------------------------------------------------
// The same lists for the unfiltered case (one node = brute force: BASELINE's headline match) on the MATRIX CORES.
// All-pairs Hamming distance is an inner product: with a = 1 - 2 A in {+1, -1} and b = B in {0, 1} per bit,
//     sum_k b_jk a_ik = |B_j| - 2 A_i . B_j,      d(A_i, B_j) = |A_i| + sum_k b_jk a_ik          (exact in i8 x i8 -> i32)
// so a 32 x 32 tile of distances is eight v_mfma_i32_32x32x32_i8 (K = 256 bits) instead of 1024 x 8 x (v_xor + v_bcnt): 16.4 k
// multiply-adds per 32-cycle matrix instruction against 64 lanes x one 32-bit word per 4-cycle vector instruction.
// One WAVE = 64 A features (two stripes of 32, kept as +-1 bytes in 64 registers: the MFMA's B operand, column = lane & 31 = the A
// feature) against all B features in tiles of 32 (the MFMA's A operand, row = lane & 31 = the B feature); waves do not talk to each
// other - no staging through LDS, no barrier in the loop (a first version that expanded the tile once per workgroup into LDS spent 52 us
// per batch in barriers and LDS round trips alone).  Which bit is which k is free as long as both operands agree: lane half g = lane >> 5
// takes bytes 16 g .. 16 g + 15 of the descriptor, step t of them bytes 2 t, 2 t + 1 - so a lane loads ONE dwordx4 per tile and turns
// each of its nibbles into four 0 / 1 bytes with a table in LDS that has one column per lane (two vector instructions and a conflict-free
// ds_read_b32 per nibble instead of three vector instructions).
// The accumulator of a lane holds ONE A feature (its column) x 16 of the tile's B features (rows (v & 3) + 8 (v >> 2) + 4 g): the list
// logic of k_bow_topk stays per lane - `acc < th - |A_i|`, first on group minima, the key only for the rare candidate that reaches the
// list, ascending j per lane -, and the two lanes that share an A feature merge their lists at the end.
// Bit-identical lists to k_bow_topk<false, false> (tests/test_matcher.py).  NOT the default: BASELINE's north_star wants the Hamming match on
// popcount-class wavefront primitives; ORBX_MATCH_MFMA=1 selects this kernel (bench.py reports its rate beside the headline).
// ---------------------------------------------------------------------------------------------
typedef int mfma_v4i __attribute__((ext_vector_type(4)));
typedef int mfma_v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t nib01(uint32_t n) { return __umul24(n, 0x00204081u) & 0x01010101u; }      // 4 bits -> 4 bytes of 0 / 1

__global__ __launch_bounds__(256, 3) void k_bow_topk_mfma(FeatDev A, FeatDev B, const int32_t *__restrict__ pairsA, const int32_t *__restrict__ pairsB, uint32_t dcut,
                                                       uint32_t *__restrict__ topk, int stride, int dbg)
{
    __shared__ uint32_t sLut[16 * 64];    // [nibble][lane]: the nibble's four bits as four bytes, one copy per lane - every read is conflict free (a 256-entry
                                          // byte table read with ds_read_b64 at data-dependent addresses kept the LDS busy 4.6 k cycles per tile and CU)
    const int p = blockIdx.y, fa = pairsA[p], fb = pairsB[p];
    const int nA = min(A.counts[fa], A.cap), nB = min(B.counts[fb], B.cap);
    const int capB = B.cap;
    const int row0 = blockIdx.x * 256;
    if (row0 >= nA) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 5, c = lane & 31;
    for (int e = tid; e < 16 * 64; e += 256) sLut[e] = nib01((uint32_t)e >> 6);
    __syncthreads();
    if (row0 + 64 * wv >= nA) return;     // (no barrier below)
    const uint32_t sentinel = dcut << 16;
    const char *lut = (const char *)sLut;
    const uint32_t lane4 = (uint32_t)lane << 2;
    mfma_v4i a[2][8];
    uint32_t kk[2][TOPK];
    int iA[2], nAc[2], T[2];
    bool live[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        iA[s] = row0 + 64 * wv + 32 * s + c;
        live[s] = iA[s] < nA;
        const uint4 *da = (const uint4 *)(A.desc + ((size_t)fa * A.cap + (live[s] ? iA[s] : nA - 1)) * 32);
        const uint4 lo = da[0], hi = da[1];
        nAc[s] = __popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w) + __popc(hi.x) + __popc(hi.y) + __popc(hi.z) + __popc(hi.w);
        const uint4 mine = g ? hi : lo;
        const uint32_t w[4] = {mine.x, mine.y, mine.z, mine.w};
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const uint32_t two = (w[t >> 1] >> (16 * (t & 1))) & 0xffffu;
#pragma unroll
            for (int q = 0; q < 4; q++) a[s][t][q] = (int)__builtin_amdgcn_perm(0u, 0x0000ff01u, nib01((two >> (4 * q)) & 15u));      // bit 0 -> +1, bit 1 -> -1
        }
#pragma unroll
        for (int q = 0; q < TOPK; q++) kk[s][q] = live[s] ? sentinel : 0u;      // rows beyond nA never insert (d >= 0 > their threshold)
        T[s] = (int)(kk[s][TOPK - 1] >> 16) - nAc[s];                          // d < th  <=>  acc < th - |A_i|
    }
    const uint4 *gD = (const uint4 *)(B.desc + (size_t)fb * capB * 32);
    const int ntiles = (nB + 31) >> 5;
    auto load_raw = [&](int tile) { const int j = tile * 32 + c; return j < nB ? gD[(size_t)j * 2 + g] : make_uint4(0u, 0u, 0u, 0u); };
    // Candidates that reach a list (d below the list's current 8th distance) are frequent - consecutive frames of one scene produce ~13 per
    // stripe and tile, 18 k pairs below dcut per 1000 x 1000 pair - but sparse per lane (0.2 per stripe and tile).  Finding them register by
    // register with wave-wide tests costs a scalar branch per register (two versions of that ran at 154-157 us per batch: the branches, not
    // the arithmetic).  So the tile is handled WITHOUT branches: every lane runs its sixteen keys through a five-slot sorted buffer that starts
    // filled with the list's threshold key (a key at or above the threshold never enters: no compare, no select), 2 + 5 instructions per
    // register; afterwards the occupied slots are inserted into the lists with wave-wide insertions - slot 0 about once per stripe and tile,
    // slot 1 for 2 % of the lanes, ... - and a fifth occupied slot (a lane with five candidates in one tile, which may have lost a sixth)
    // sends the stripe through the exact path: all sixteen keys, one insertion each.
    uint4 raw = load_raw(0);
    auto one_tile = [&](const int tile, auto lastTag) {
        constexpr bool LAST = decltype(lastTag)::value;      // the last tile may hold rows beyond nB (all-zero descriptors): excluded explicitly
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
        if (!LAST) raw = load_raw(tile + 1);                  // in flight during this tile
        mfma_v4i b[8];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const uint32_t ww = w[t >> 1];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int sh = 16 * (t & 1) + 4 * q - 8;       // nibble value to bits 8..11: the table row, 256 bytes each
                const uint32_t off = ((sh >= 0 ? ww >> sh : ww << -sh) & 0xf00u) | lane4;
                b[t][q] = (int)*(const uint32_t *)(lut + off);
            }
        }
        mfma_v16i acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
        if (!(dbg & 2)) {
#pragma unroll
            for (int t = 0; t < 8; t++) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[t], a[0][t], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[t], a[1][t], acc1, 0, 0, 0);
            }
        } else { acc0[0] = b[0][0] + b[1][1] + b[2][2] + b[3][3] + b[4][0] + b[5][1] + b[6][2] + b[7][3]; acc1[0] = acc0[0] + 1000; }
        if (dbg & 1) { if (acc0[3] + acc1[5] == 0x12345678) kk[0][0] = 0; return; }
        const uint32_t jb = (uint32_t)(tile * 32 + 4 * g);      // row of accumulator register v: jb | (v & 3) | 8 (v >> 2)  (disjoint bits)
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const mfma_v16i &acc = s ? acc1 : acc0;
            const uint32_t Tkey = (uint32_t)(T[s] + nAc[s]) << 16;      // the list's threshold distance as a key: d < th  <=>  key < Tkey
            auto key_of = [&](int v) {
                const uint32_t cv = (uint32_t)((v & 3) | (8 * (v >> 2)));
                const uint32_t k = ((uint32_t)(nAc[s] + acc[v]) << 16) | jb | cv;
                return (LAST && !((int)(jb | cv) < nB)) ? 0xffffffffu : k;
            };
            uint32_t q0 = Tkey, q1 = Tkey, q2 = Tkey, q3 = Tkey, q4 = Tkey;
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const uint32_t k = key_of(v);
                q4 = med3_u32(q3, q4, k); q3 = med3_u32(q2, q3, k); q2 = med3_u32(q1, q2, k); q1 = med3_u32(q0, q1, k);
                q0 = min(q0, k);
            }
            if (__any(q0 < Tkey)) {
                if (__any(q4 < Tkey)) {
#pragma unroll
                    for (int v = 0; v < 16; v++) { const uint32_t k = key_of(v); topk_insert(kk[s], k < Tkey ? k : 0xffffffffu); }
                } else {
                    topk_insert(kk[s], q0 < Tkey ? q0 : 0xffffffffu);
                    if (__any(q1 < Tkey)) {
                        topk_insert(kk[s], q1 < Tkey ? q1 : 0xffffffffu);
                        if (__any(q2 < Tkey)) {
                            topk_insert(kk[s], q2 < Tkey ? q2 : 0xffffffffu);
                            if (__any(q3 < Tkey)) topk_insert(kk[s], q3 < Tkey ? q3 : 0xffffffffu);
                        }
                    }
                }
                T[s] = (int)(kk[s][TOPK - 1] >> 16) - nAc[s];
            }
        }
    };
    for (int tile = 0; tile + 1 < ntiles; tile++) one_tile(tile, std::false_type());
    if (ntiles > 0) one_tile(ntiles - 1, std::true_type());
    // the two lanes of an A feature (rows 4 g .. of every tile each): merge the lists, lane half 0 writes
#pragma unroll
    for (int s = 0; s < 2; s++) {
        uint32_t other[TOPK];
#pragma unroll
        for (int q = 0; q < TOPK; q++) other[q] = (uint32_t)__shfl((int)kk[s][q], lane ^ 32);
#pragma unroll
        for (int q = 0; q < TOPK; q++) topk_insert(kk[s], other[q]);
        if (g == 0 && live[s]) {
            uint32_t *out = topk + ((size_t)p * stride + iA[s]) * TOPK;
#pragma unroll
            for (int k = 0; k < TOPK; k++) out[k] = kk[s][k] >= sentinel ? KEY_EMPTY : kk[s][k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The SINGLE-PAIR geometry (ORBmatcher::SearchByBoW as src/Tracking.cc:1195, 2073 call it: one KeyFrame against one Frame, ~1000 x 1000).
// k_bow_topk above gives a pair ceil(nA / 256) workgroups that each walk ALL of B: four workgroups on 256 CUs, 100 us for one pair where 256 pairs
// take 216.  Here a WAVE is one A feature: lane s scans slice s of B - 64 slices, ~16 features each for a 1000-feature frame -, a workgroup is four A
// features (252 workgroups for 1005: the chip).  All of B sits in the workgroup's LDS (descriptors + node ids, 36 bytes per feature); a lane scans its slice
// with its feature's descriptor in 8 VGPRs and keeps its top-TOPK, then the lanes of a feature merge their lists with six rounds of butterfly exchanges -
// no partial lists in memory, no atomics, no second pass.  Keys carry the B index (dist << 16 | j), so the smallest TOPK of the union of the slices'
// lists ARE the lists k_bow_topk produces: the replay below sees identical input.  A slice's descriptors start 16 bytes further into the 256-byte bank
// window than its neighbour's.  (Measured on the tracked-frame loop: 16 slices x 16 features per workgroup 15.3 us, 32 x 8 13.0 us, 64 x 4 9.6 us - the
// scan is one wave per SIMD waiting on its own LDS reads and inserts, so its length is what counts.)
// The processing order of the A features (k_bow_order: rank by (node id, index), 46 us for one pair with its 1000-step loop over global memory) is
// computed by further workgroups of the SAME launch: 32 features x 8 lanes per workgroup, node ids in LDS, four per read.
// ---------------------------------------------------------------------------------------------
#define PAIR_SLICES 64                     /* B slices = lanes per A feature */
#define PAIR_ROWS (4 * (64 / PAIR_SLICES))   /* A features per workgroup (four waves) */
template <bool FILTER>
__global__ __launch_bounds__(256) void k_bow_topk_pair(FeatDev A, FeatDev B, int mode, uint32_t dcut, int nRowBlocks, int per, uint32_t *__restrict__ topk,
                                                       int32_t *__restrict__ order)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smemP[];
    const int nA = min(A.counts[0], A.cap), nB = min(B.counts[0], B.cap);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if ((int)blockIdx.x >= nRowBlocks) {
        // ---- processing order: rank of feature i = #{k : (g[k], k) < (g[i], i)}; 32 features per workgroup, 8 lanes share a feature
        int32_t *sGr = (int32_t *)smemP;      // [4096]
        const int ob = (int)blockIdx.x - nRowBlocks, i = ob * 32 + (tid >> 3), seg = tid & 7;
        if (!A.groups) { if (seg == 0 && i < nA) order[i] = i; return; }
        const int gi = i < nA ? A.groups[i] : 0;
        int rank = 0;
        for (int t0 = 0; t0 < nA; t0 += 4096) {
            const int tn = min(4096, nA - t0), tn4 = (tn + 3) & ~3;
            __syncthreads();
            for (int k = tid; k < tn4; k += 256) sGr[k] = k < tn ? A.groups[t0 + k] : 0x7fffffff;      // (padding ranks behind everything)
            __syncthreads();
#pragma unroll 4
            for (int k = 4 * seg; k < tn4; k += 32) {
                const int4 g4 = *(const int4 *)&sGr[k];
                const int kk = t0 + k;
                rank += (g4.x < gi) || (g4.x == gi && kk < i);
                rank += (g4.y < gi) || (g4.y == gi && kk + 1 < i);
                rank += (g4.z < gi) || (g4.z == gi && kk + 2 < i);
                rank += (g4.w < gi) || (g4.w == gi && kk + 3 < i);
            }
        }
        rank += __shfl_xor(rank, 1); rank += __shfl_xor(rank, 2); rank += __shfl_xor(rank, 4);
        if (seg == 0 && i < nA) order[rank] = i;
        return;
    }
    // LDS: slice s of B = `per` descriptors at sliceBase(s) = s * (per * 32 + 16) bytes; then the node ids [PAIR_SLICES * per] (0x80000000 = excluded)
    const int slicePitch = per * 32 + 16;
    unsigned char *sDesc = smemP;
    int32_t *sG = (int32_t *)(smemP + (size_t)PAIR_SLICES * slicePitch);
    // this lane's KeyFrame feature first: its loads are in flight while B is staged
    const int r = lane / PAIR_SLICES, sl = lane % PAIR_SLICES;
    const int row = (int)blockIdx.x * PAIR_ROWS + wv * (64 / PAIR_SLICES) + r;
    const bool live = row < nA;
    uint4 aLo, aHi;
    {
        const uint4 *da = (const uint4 *)(A.desc + (size_t)(live ? row : nA - 1) * 32);
        aLo = da[0]; aHi = da[1];
    }
    const int gA = (FILTER && A.groups) ? A.groups[live ? row : nA - 1] : 0;
    {
        // B -> LDS, four 16-byte units per thread per step with all four loads requested before the first store (a load-store loop is one L2 round trip
        // per unit: eight in a row for 1000 features, most of this kernel's 16 us); unconditional loads from clamped addresses, masked afterwards
        const uint4 *gD = (const uint4 *)B.desc;
        const int total = 2 * PAIR_SLICES * per, last = 2 * nB - 1;
        for (int t0 = tid; t0 < total; t0 += 4 * 256) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = gD[min(t0 + 256 * u, last)];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = t0 + 256 * u;
                if (t < total) {
                    const int j = t >> 1, sl2 = j / per, jj = j - sl2 * per;
                    *(uint4 *)(sDesc + (size_t)sl2 * slicePitch + (size_t)jj * 32 + (t & 1) * 16) = j < nB ? v[u] : make_uint4(0u, 0u, 0u, 0u);
                }
            }
        }
        if (FILTER) {
            const int totalG = PAIR_SLICES * per;
            for (int j0 = tid; j0 < totalG; j0 += 4 * 256) {
                int g[4], vv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = min(j0 + 256 * u, nB - 1);
                    g[u] = B.groups ? B.groups[j] : 0;
                    vv[u] = (mode >= 1 && B.valid) ? (int)B.valid[j] : 1;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = j0 + 256 * u;
                    if (j < totalG) {
                        int gq = g[u];
                        if (j >= nB || gq < 0 || !vv[u]) gq = (int)0x80000000;      // padding; not filed in the FeatureVector; no valid MapPoint (mode 1): never matched
                        sG[j] = gq;
                    }
                }
            }
        }
    }
    const uint32_t sentinel = dcut << 16;
    uint32_t a[8], kk[TOPK];
    a[0] = aLo.x; a[1] = aLo.y; a[2] = aLo.z; a[3] = aLo.w; a[4] = aHi.x; a[5] = aHi.y; a[6] = aHi.z; a[7] = aHi.w;
    const bool act = live && !(FILTER && gA < 0);      // unfiled A features keep empty lists
#pragma unroll
    for (int q = 0; q < TOPK; q++) kk[q] = act ? sentinel : 0u;      // inactive rows never insert
    uint32_t th = kk[TOPK - 1] >> 16;
    __syncthreads();
    const unsigned char *myB = sDesc + (size_t)sl * slicePitch;
    const int j0 = sl * per, jn = max(0, min(per, nB - j0));      // this lane's slice: B features j0 .. j0 + jn
    for (int j = 0; j < per; j++) {                               // (uniform trip count; a step beyond the slice's end compares against zero padding and is masked)
        const uint4 lo = *(const uint4 *)(myB + (size_t)j * 32), hi = *(const uint4 *)(myB + (size_t)j * 32 + 16);
        uint32_t d = bcnt_acc(a[0] ^ lo.x, 0u);
        d = bcnt_acc(a[1] ^ lo.y, d); d = bcnt_acc(a[2] ^ lo.z, d); d = bcnt_acc(a[3] ^ lo.w, d);
        d = bcnt_acc(a[4] ^ hi.x, d); d = bcnt_acc(a[5] ^ hi.y, d); d = bcnt_acc(a[6] ^ hi.z, d); d = bcnt_acc(a[7] ^ hi.w, d);
        if (FILTER) d = sG[j0 + j] == gA ? d : 0xffffu;
        // scan order inside a slice is ascending j: an equal distance with a later j has the larger key, only d < th can enter (as in k_bow_topk)
        const uint32_t key = (j < jn && d < th) ? (d << 16) | (uint32_t)(j0 + j) : 0xffffffffu;
        if (__any(key < kk[TOPK - 1])) { topk_insert(kk, key); th = kk[TOPK - 1] >> 16; }
    }
    // the slices of a feature: butterfly over the lane bits 1, 2, 4, ... - after round m every lane holds the TOPK of its group of 2^(m+1) slices
#pragma unroll
    for (int m = 1; m < PAIR_SLICES; m <<= 1) {
        uint32_t other[TOPK];
#pragma unroll
        for (int q = 0; q < TOPK; q++) other[q] = (uint32_t)__shfl_xor((int)kk[q], m);
#pragma unroll
        for (int q = 0; q < TOPK; q++) if (__any(other[q] < kk[TOPK - 1])) topk_insert(kk, other[q]);
    }
    if (sl == 0 && live) {
        uint4 *out = (uint4 *)(topk + (size_t)row * TOPK);
        uint32_t o[TOPK];
#pragma unroll
        for (int q = 0; q < TOPK; q++) o[q] = (!act || kk[q] >= sentinel) ? KEY_EMPTY : kk[q];
        out[0] = make_uint4(o[0], o[1], o[2], o[3]); out[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
}

// greedy replay + rotation histogram + three-maxima pruning; one workgroup per pair.
//
// The reference's pass over the A features is sequential only through the "already matched" flag
// of the B features (src/ORBmatcher.cc:276-279, 722-724).  Number the A features by their position
// r in the processing order: the decision of rank r is a function of the decisions of the ranks
// below r only, so the sequential result is the UNIQUE fixed point of
//     dec[r] = decide(candidates of r that no accepted rank r' < r has chosen)
// (induction on r).  The kernel iterates that map for all ranks in parallel: owner[b] = lowest
// accepted rank currently choosing b (LDS atomicMin), then every rank re-decides with "b is free
// iff owner[b] >= r", until a round changes nothing.  After round t the ranks < t are final, so it
// terminates; in practice conflicts are sparse and a handful of rounds suffice, instead of one
// dependent step per feature.  A rank whose (full) candidate list has fewer than two free entries
// left cannot be decided from the list: those ranks are queued and one wave each rescans all of B
// exactly, as before.
#define GREEDY_THREADS 1024   /* a row per thread for up to 1024 features: every round of the fixed point is one parallel sweep */
__global__ __launch_bounds__(GREEDY_THREADS) void k_bow_greedy(FeatDev A, FeatDev B, const int32_t *__restrict__ pairsA, const int32_t *__restrict__ pairsB, int mode,
                                                    float nnratio, int checkOri, const uint32_t *__restrict__ topk, const int32_t *__restrict__ order,
                                                    int32_t *__restrict__ matches, int32_t *__restrict__ dists, int32_t *__restrict__ nmatches, int stride, TriDev T,
                                                    int32_t *__restrict__ pubMatches, unsigned long long *pubFlag, unsigned long long pubSeq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int sChangedP[2], sQueued, sTotal, sRemoved;
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, fa = pairsA[p], fb = pairsB[p];
    const int nA = min(A.counts[fa], A.cap), nB = min(B.counts[fb], B.cap);
    // LDS: owner[B.cap] | dec[A.cap] (KEY_EMPTY = no match, else dist << 16 | b, later | rotation bin << 26) | order (u16) | rescan queue (u16)
    uint32_t *owner = (uint32_t *)smem;
    uint32_t *dec = owner + B.cap;
    unsigned short *sOrd = (unsigned short *)(dec + A.cap);
    unsigned short *queue = sOrd + ((A.cap + 7) & ~7);
    // the single-pair host call (pubFlag: the latency of this ONE workgroup is the call's): the B angles and the match list live in LDS as well - angB[B.cap] |
    // res[max(A.cap, B.cap)] behind the queue (the host sizes the allocation) -, so that the epilogue waits for no global load and reads nothing back that it
    // has just stored; the A angle of a thread's own rank is requested with the rank
    float *sAngB = pubFlag ? (float *)(queue + ((A.cap + 7) & ~7)) : nullptr;
    int32_t *sRes = pubFlag ? (int32_t *)(sAngB + B.cap) : nullptr;
    int32_t *mout = matches + (size_t)p * stride, *dout = dists + (size_t)p * stride;
    const uint4 *tk = (const uint4 *)(topk + (size_t)p * stride * TOPK);
    const orbx_keypoint *kA = A.kp + (size_t)fa * A.cap, *kB = B.kp + (size_t)fb * B.cap;
    if (tid < HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) { sTotal = 0; sRemoved = 0; }
    float angA0 = 0.0f;
    {
        const int32_t *ord = order + (size_t)p * stride;
        for (int r = tid; r < nA; r += GREEDY_THREADS) {
            int i = ord[r];
            if (pubFlag && r == tid) angA0 = kA[i].angle;
            if (A.valid && !A.valid[(size_t)fa * A.cap + i]) i = 0xffff;
            sOrd[r] = (unsigned short)i;
            dec[r] = KEY_EMPTY;
        }
    }
    if (pubFlag) {
        for (int j = tid; j < nB; j += GREEDY_THREADS) sAngB[j] = kB[j].angle;
        const int nOut0 = mode == 0 ? nB : nA;
        for (int s2 = tid; s2 < nOut0; s2 += GREEDY_THREADS) sRes[s2] = -1;
    }
    for (int s = tid; s < stride; s += GREEDY_THREADS) { mout[s] = -1; dout[s] = 256; }
    uint4 myK0 = make_uint4(0u, 0u, 0u, 0u), myK1 = myK0, othK0 = myK0, othK1 = myK0;
    bool haveKeys = false;
    // A round = claims (atomicMin), decisions, rarely rescans: three barriers (a fourth behind rescans).  The "changed" flag alternates between two words so
    // that the next round may clear its own while this round's is still being read; owner[] and the queue counter are reset for the NEXT round behind the
    // barrier that ends this round's reads of them.
    for (int j = tid; j < nB; j += GREEDY_THREADS) owner[j] = 0xffffffffu;
    if (tid == 0) { sChangedP[0] = 0; sChangedP[1] = 0; sQueued = 0; }
    for (int par = 0;; par ^= 1) {
        __syncthreads();
        for (int r = tid; r < nA; r += GREEDY_THREADS) {
            const uint32_t d = dec[r];
            if (d != KEY_EMPTY) atomicMin(&owner[d & 0xffff], (uint32_t)r);
        }
        __syncthreads();
        bool changed = false;
        for (int r = tid; r < nA; r += GREEDY_THREADS) {
            const int i = sOrd[r];
            if (i == 0xffff) continue;
            // (rank tid's list stays in registers across the rounds: a round was two dependent L2 round trips per rank)
            if (r != tid || !haveKeys) { const uint4 q0 = tk[i * 2], q1 = tk[i * 2 + 1]; if (r == tid) { myK0 = q0; myK1 = q1; haveKeys = true; } else { othK0 = q0; othK1 = q1; } }
            const uint4 q0 = r == tid ? myK0 : othK0, q1 = r == tid ? myK1 : othK1;
            const uint32_t keys[TOPK] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            // The first two free candidates in list order, without a branch per candidate (a loop of "test, then read owner[]" is eight dependent LDS round
            // trips behind eight branches): four owner words are requested together (word 0 for an empty slot), then the four slots are walked BACKWARDS
            // with two selects each - what remains in (k1, k2) are the first two free.  The second half of the list is looked at only by a rank that found
            // fewer than two free candidates in the first (sixteen waves share one CU's LDS and issue slots: the rounds are this kernel's duration).
            auto half = [&](const int k0, uint32_t &h1, uint32_t &h2) -> int {
                uint32_t bi[4], ow[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t low = keys[k0 + k] & 0xffffu;
                    bi[k] = keys[k0 + k] == KEY_EMPTY ? 0u : (mode == 2 ? 0xffffu - low : low);   // mode 2 keys carry 0xffff - j
                }
#pragma unroll
                for (int k = 0; k < 4; k++) ow[k] = owner[bi[k]];
                int nf = 0;
                h1 = KEY_EMPTY; h2 = KEY_EMPTY;
#pragma unroll
                for (int k = 3; k >= 0; k--) {
                    const bool fr = keys[k0 + k] != KEY_EMPTY && ow[k] >= (uint32_t)r;
                    const uint32_t packed = (keys[k0 + k] & 0xffff0000u) | bi[k];
                    h2 = fr ? h1 : h2;
                    h1 = fr ? packed : h1;
                    nf += fr ? 1 : 0;
                }
                return nf;
            };
            uint32_t k1, k2;
            int nfree = half(0, k1, k2);
            if (nfree < 2 && (keys[4] & keys[5] & keys[6] & keys[7]) != KEY_EMPTY) {
                uint32_t j1, j2;
                const int nb = half(4, j1, j2);
                k2 = nfree == 1 ? j1 : j2;
                k1 = nfree == 1 ? k1 : j1;
                nfree += nb;
            }
            const uint32_t bestKey = k1;
            const int best2 = nfree >= 2 ? (int)(k2 >> 16) : 256;
            // mode 2 (SearchForTriangulation) has no ratio test: the best free candidate decides alone
            if (nfree < (mode == 2 ? 1 : 2) && keys[TOPK - 1] != KEY_EMPTY) { queue[atomicAdd(&sQueued, 1)] = (unsigned short)r; continue; }
            uint32_t nd = KEY_EMPTY;
            if (bestKey != KEY_EMPTY) {
                const int best1 = (int)(bestKey >> 16);
                if (mode == 2) nd = bestKey;
                else {
                    const bool pass = mode == 0 ? (best1 <= TH_LOW) : (best1 < TH_LOW);
                    if (pass && (float)best1 < nnratio * (float)best2) nd = bestKey;
                }
            }
            if (nd != dec[r]) { dec[r] = nd; changed = true; }
        }
        if (changed) sChangedP[par] = 1;
        __syncthreads();
        // exact rescans: one wave per queued rank, lanes over the B features of its node
        const int nq = sQueued;
        for (int q = wv; q < nq; q += GREEDY_THREADS / 64) {
            const int r = queue[q], i = sOrd[r];
            const unsigned long long *da = (const unsigned long long *)(A.desc + ((size_t)fa * A.cap + i) * 32);
            unsigned long long a[4] = {da[0], da[1], da[2], da[3]};
            const int gA = A.groups ? A.groups[(size_t)fa * A.cap + i] : 0;
            uint32_t k0 = KEY_EMPTY, k1 = KEY_EMPTY;
            for (int jj = lane; jj < nB; jj += 64) {
                if (owner[jj] < (uint32_t)r) continue;
                if (B.groups && B.groups[(size_t)fb * B.cap + jj] != gA) continue;
                if (mode >= 1 && B.valid && !B.valid[(size_t)fb * B.cap + jj]) continue;
                const unsigned long long *db = (const unsigned long long *)(B.desc + ((size_t)fb * B.cap + jj) * 32);
                int d = hamming256(a, db[0], db[1], db[2], db[3]);
                uint32_t kk = ((uint32_t)d << 16) | (uint32_t)jj;
                if (mode == 2) {
                    if (d > TH_LOW) continue;
                    const size_t ia = (size_t)fa * A.cap + i;
                    if (!tri_geom_ok(T, p, A.kp[ia], T.stereoA && T.stereoA[ia], B.kp[(size_t)fb * B.cap + jj], T.stereoB && T.stereoB[(size_t)fb * B.cap + jj])) continue;
                    kk = ((uint32_t)d << 16) | (0xffffu - (uint32_t)jj);   // equal distances: the later feature wins (:880)
                }
                if (kk < k0) { k1 = k0; k0 = kk; } else if (kk < k1) k1 = kk;
            }
            uint32_t bestKey = wave_min_u32(k0);
            if (k0 == bestKey) k0 = k1;
            const uint32_t second = wave_min_u32(k0);
            uint32_t nd = KEY_EMPTY;
            if (bestKey != KEY_EMPTY) {
                const int best1 = (int)(bestKey >> 16), best2 = second != KEY_EMPTY ? (int)(second >> 16) : 256;
                if (mode == 2) nd = (bestKey & 0xffff0000u) | (0xffffu - (bestKey & 0xffffu));
                else {
                    const bool pass = mode == 0 ? (best1 <= TH_LOW) : (best1 < TH_LOW);
                    if (pass && (float)best1 < nnratio * (float)best2) nd = bestKey;
                }
            }
            if (lane == 0 && nd != dec[r]) { dec[r] = nd; sChangedP[par] = 1; }
        }
        if (nq) __syncthreads();      // (uniform: every thread read the same count behind the barrier above)
        const int again = sChangedP[par];
        if (!again) break;
        for (int j = tid; j < nB; j += GREEDY_THREADS) owner[j] = 0xffffffffu;
        if (tid == 0) { sChangedP[par ^ 1] = 0; sQueued = 0; }
    }
    // ---- results + rotation histogram (src/ORBmatcher.cc:318-332, 750-758) ----
    const float factor = HISTO_LENGTH / 360.0f;
    int total = 0;
    for (int r = tid; r < nA; r += GREEDY_THREADS) {
        const uint32_t d = dec[r];
        if (d == KEY_EMPTY) continue;
        const int i = sOrd[r], bj = (int)(d & 0xffff), slot = mode == 0 ? bj : i;
        mout[slot] = mode == 0 ? i : bj; dout[slot] = (int)(d >> 16);
        if (sRes) sRes[slot] = mode == 0 ? i : bj;
        total++;
        if (checkOri) {
            float rot = ((pubFlag && r == tid) ? angA0 : kA[i].angle) - (sAngB ? sAngB[bj] : kB[bj].angle);
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            dec[r] = d | ((uint32_t)bin << 26);
            atomicAdd(&hist[bin], 1);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    if (lane == 0 && total) atomicAdd(&sTotal, total);
    __syncthreads();
    if (checkOri) {
        // ComputeThreeMaxima, src/ORBmatcher.cc:1866-1908
        int ind1, ind2, ind3;
        three_maxima_wave(hist, lane, ind1, ind2, ind3);
        int removed = 0;
        for (int r = tid; r < nA; r += GREEDY_THREADS) {       // same rank -> thread mapping as the writes above
            const uint32_t d = dec[r];
            if (d == KEY_EMPTY) continue;
            const int b = (int)(d >> 26);
            if (b != ind1 && b != ind2 && b != ind3) {
                const int slot = mode == 0 ? (int)(d & 0xffff) : (int)sOrd[r];
                mout[slot] = -1; dout[slot] = 256; removed++;
                if (sRes) sRes[slot] = -1;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o);
        if (lane == 0 && removed) atomicAdd(&sRemoved, removed);
    }
    __syncthreads();
    if (tid == 0) nmatches[p] = sTotal - sRemoved;
    if (pubFlag) {
        // the single-pair host call: the match list and the count go straight into the caller's mapped pinned buffer, the sequence word behind them
        const int nOut = mode == 0 ? nB : nA;
        for (int s2 = tid; s2 < nOut; s2 += GREEDY_THREADS) pubMatches[s2] = sRes[s2];
        if (tid == 0) pubMatches[nOut] = sTotal - sRemoved;
        orbx_publish(nullptr, pubFlag, pubSeq, 1u);
    }
}

// Hamming stage of Frame::ComputeStereoMatches: one wave per left keypoint, lanes over the
// right keypoints; the row-band / octave / disparity gate is evaluated per candidate exactly
// as the reference builds vRowIndices (src/Frame.cc:1060-1097, 1145, 1190-1200).
__global__ __launch_bounds__(256) void k_stereo(FeatDev Lf, FeatDev Rf, const int32_t *__restrict__ pairsL, const int32_t *__restrict__ pairsR,
                                                const float *__restrict__ scaleFactors, float maxD, int32_t *__restrict__ bestIdx, int32_t *__restrict__ bestDist,
                                                int32_t *__restrict__ nmatches, int stride)
{
    const int p = blockIdx.y, fl = pairsL[p], fr = pairsR[p];
    const int nL = min(Lf.counts[fl], Lf.cap), nR = min(Rf.counts[fr], Rf.cap);
    const int lane = threadIdx.x & 63, iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (iL >= nL) return;
    const orbx_keypoint kl = Lf.kp[(size_t)fl * Lf.cap + iL];
    const int row = (int)kl.y;
    const float minU = kl.x - maxD, maxU = kl.x - 0.0f;
    const unsigned long long *da = (const unsigned long long *)(Lf.desc + ((size_t)fl * Lf.cap + iL) * 32);
    unsigned long long a[4] = {da[0], da[1], da[2], da[3]};
    uint32_t best = ((uint32_t)TH_HIGH << 16);   // bestDist = TH_HIGH, strict '<' below
    if (!(maxU < 0)) {
        const orbx_keypoint *kr = Rf.kp + (size_t)fr * Rf.cap;
        for (int iR = lane; iR < nR; iR += 64) {
            const orbx_keypoint k = kr[iR];
            const float r = 2.0f * scaleFactors[k.octave];
            const int maxr = (int)ceilf(k.y + r), minr = (int)floorf(k.y - r);
            if (row < minr || row > maxr) continue;
            if (k.octave < kl.octave - 1 || k.octave > kl.octave + 1) continue;
            if (!(k.x >= minU && k.x <= maxU)) continue;
            const unsigned long long *db = (const unsigned long long *)(Rf.desc + ((size_t)fr * Rf.cap + iR) * 32);
            const int d = hamming256(a, db[0], db[1], db[2], db[3]);
            const uint32_t key = ((uint32_t)d << 16) | (uint32_t)iR;
            if (d < TH_HIGH && key < best) best = key;
        }
    }
    best = wave_min_u32(best);
    if (lane == 0) {
        const int d = (int)(best >> 16);
        bestDist[(size_t)p * stride + iL] = d;
        bestIdx[(size_t)p * stride + iL] = d < TH_HIGH ? (int)(best & 0xffff) : 0;
        if (d < (TH_HIGH + TH_LOW) / 2) atomicAdd(&nmatches[p], 1);
    }
}


// ---------------------------------------------------------------------------------------------
// Complete Frame::ComputeStereoMatches (src/Frame.cc:1026-1420).
// k_stereo_full: one wave per left keypoint.
//   1. Hamming stage exactly as k_stereo (row band / octave / disparity gate, first minimum wins).
//   2. if bestDist < (TH_HIGH+TH_LOW)/2: 11x11 SAD of the centre-normalised left window against
//      the right window shifted by -5..+5 px on the keypoint's pyramid level (:1224-1306);
//      lane = (shift s, row group q): 55 lanes, 3/2/2/2/2 rows of 11 pixels each.
//   3. parabola fit, sub-pixel uR, disparity gate, depth (:1330-1378); float arithmetic in the
//      reference's order (-ffp-contract=off), the 0.01 clamp in double like `uL-0.01`.
// k_stereo_cut: one workgroup per pair; median of the accepted SAD values by a two-pass byte
// radix select (SAD <= 121*510 < 2^16), then the `median*1.5*1.4` cut (:1387-1416).
// ---------------------------------------------------------------------------------------------
struct PyrDev {
    const uint8_t *img0;
    int img0Stride;
    size_t img0FramePitch;
    const uint8_t *pyr;
    size_t pyrBytes;
    const OrbxGeom *geom;
};

__device__ __forceinline__ const uint8_t *level_ptr(const PyrDev &P, int f, int l, int &pitch, int &w)
{
    w = P.geom->lv[l].w;
    if (l == 0) { pitch = P.img0Stride; return P.img0 + (size_t)f * P.img0FramePitch; }
    pitch = P.geom->lv[l].pitch;
    return P.pyr + (size_t)f * P.pyrBytes + P.geom->lv[l].off;
}

// vRowIndices (src/Frame.cc:1179-1196): for every image row the right keypoints whose band [floor(y - r), ceil(y + r)], r = 2 * scale factor of
// their octave, contains it.  One workgroup per right frame: row counts in LDS, scan, fill.  The order inside a row is arbitrary - the best
// candidate is the minimum of (distance, index), which is what the reference's strict `<` over its index-ordered rows yields.  Without the rows
// every left keypoint walked ALL right keypoints (2000 x 2000 band tests per pair: 0.46 ms of the 1.5 ms of a 64-pair KITTI batch).
__global__ __launch_bounds__(256) void k_stereo_rows(FeatDev Rf, const int32_t *__restrict__ pairsR, const float *__restrict__ scaleFactors, int H, int32_t *__restrict__ rowStart,
                                                     int32_t *__restrict__ rowList, int listCap)
{
    extern __shared__ int srows[];        // [H + 1] counts -> starts, [H + 1] fill cursors
    __shared__ int wsum[4];
    int *cnt = srows, *cur = srows + H + 1;
    const int p = blockIdx.x, fr = pairsR[p], tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nR = min(Rf.counts[fr], Rf.cap);
    const orbx_keypoint *kr = Rf.kp + (size_t)fr * Rf.cap;
    for (int y = tid; y <= H; y += 256) cnt[y] = 0;
    __syncthreads();
    for (int iR = tid; iR < nR; iR += 256) {
        const orbx_keypoint k = kr[iR];
        const float r = 2.0f * scaleFactors[k.octave];
        const int maxr = min((int)ceilf(k.y + r), H - 1), minr = max((int)floorf(k.y - r), 0);
        for (int y = minr; y <= maxr; y++) atomicAdd(&cnt[y], 1);
    }
    __syncthreads();
    {   // exclusive scan of cnt[0..H]: a contiguous chunk per thread, chunk sums through the waves
        const int per = (H + 1 + 255) / 256, b = tid * per;
        int ssum = 0;
        for (int k = 0; k < per; k++) if (b + k <= H) ssum += cnt[b + k];
        int inc = ssum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int off = 0;
        for (int i = 0; i < wv; i++) off += wsum[i];
        int run = off + inc - ssum;
        for (int k = 0; k < per; k++)
            if (b + k <= H) { const int v = cnt[b + k]; cnt[b + k] = run; cur[b + k] = run; run += v; }
    }
    __syncthreads();
    int32_t *rs = rowStart + (size_t)p * (H + 1), *rl = rowList + (size_t)p * listCap;
    for (int y = tid; y <= H; y += 256) rs[y] = cnt[y];
    for (int iR = tid; iR < nR; iR += 256) {
        const orbx_keypoint k = kr[iR];
        const float r = 2.0f * scaleFactors[k.octave];
        const int maxr = min((int)ceilf(k.y + r), H - 1), minr = max((int)floorf(k.y - r), 0);
        for (int y = minr; y <= maxr; y++) {
            const int pos = atomicAdd(&cur[y], 1);
            if (pos < listCap) rl[pos] = iR;
        }
    }
}

// one wave = one left keypoint iL of pair p (left frame fl, right frame fr): candidates of its row, best descriptor, SAD refinement
// SCAN = false: the candidates of a row come from the table k_stereo_rows built (rowStart / rowList); SCAN = true (256 threads = four left
// keypoints per workgroup, pair 0): the workgroup collects them itself in `lds` ([4][listCap] indices), see k_stereo_scan.
template <bool SCAN>
__device__ __forceinline__ void stereo_full_body(const int iL, const int p, const int fl, const int fr, const FeatDev &Lf, const FeatDev &Rf, const PyrDev &PL, const PyrDev &PR,
                                                 const float *__restrict__ scaleFactors, const float *__restrict__ invScaleFactors, float mbf, float mb,
                                                 int32_t *__restrict__ bestIdx, int32_t *__restrict__ bestDist, float *__restrict__ uRight, float *__restrict__ depth,
                                                 int32_t *__restrict__ sadOut, int stride, const int32_t *__restrict__ rowStart, const int32_t *__restrict__ rowList, int H,
                                                 int listCap, int *lds)
{
    const int nL = min(Lf.counts[fl], Lf.cap);
    const int lane = threadIdx.x & 63;
    const bool active = iL < nL;                               // (wave-uniform)
    if (!SCAN && !active) return;
    const orbx_keypoint kl = Lf.kp[(size_t)fl * Lf.cap + (active ? iL : 0)];
    const int row = (int)kl.y;
    const float minZ = mb, minD = 0.0f;
    const float maxD = mbf / minZ;
    const float minU = kl.x - maxD, maxU = kl.x - minD;
    const unsigned long long *da = (const unsigned long long *)(Lf.desc + ((size_t)fl * Lf.cap + (active ? iL : 0)) * 32);
    unsigned long long a[4] = {da[0], da[1], da[2], da[3]};
    uint32_t best = ((uint32_t)TH_HIGH << 16);
    const orbx_keypoint *kr = Rf.kp + (size_t)fr * Rf.cap;
    const bool rowOk = active && !(maxU < 0) && row >= 0 && row < H;
    const int32_t *rl = nullptr;
    int c0 = 0, c1 = 0;
    if (SCAN) {
        __shared__ int sRow[4], sCnt[4];
        const int wv = threadIdx.x >> 6;
        if (lane == 0) { sRow[wv] = rowOk ? row : -1; sCnt[wv] = 0; }
        __syncthreads();
        const int r0 = sRow[0], r1 = sRow[1], r2 = sRow[2], r3 = sRow[3];
        const int nR = min(Rf.counts[fr], Rf.cap);
        for (int iR = threadIdx.x; iR < nR; iR += 256) {       // vRowIndices (src/Frame.cc:1179-1196), the four rows this workgroup needs
            const float ky = kr[iR].y, r = 2.0f * scaleFactors[kr[iR].octave];
            const int maxr = min((int)ceilf(ky + r), H - 1), minr = max((int)floorf(ky - r), 0);
            if (r0 >= minr && r0 <= maxr) { const int pos = atomicAdd(&sCnt[0], 1); if (pos < listCap) lds[pos] = iR; }
            if (r1 >= minr && r1 <= maxr) { const int pos = atomicAdd(&sCnt[1], 1); if (pos < listCap) lds[listCap + pos] = iR; }
            if (r2 >= minr && r2 <= maxr) { const int pos = atomicAdd(&sCnt[2], 1); if (pos < listCap) lds[2 * listCap + pos] = iR; }
            if (r3 >= minr && r3 <= maxr) { const int pos = atomicAdd(&sCnt[3], 1); if (pos < listCap) lds[3 * listCap + pos] = iR; }
        }
        __syncthreads();
        if (!active) return;
        rl = lds + wv * listCap; c1 = min(sCnt[wv], listCap);
    } else if (rowOk) {
        const int32_t *rs = rowStart + (size_t)p * (H + 1);
        rl = rowList + (size_t)p * listCap;
        c0 = rs[row]; c1 = min(rs[row + 1], listCap);
    }
    {
        for (int c = c0 + lane; c < c1; c += 64) {             // vCandidates = vRowIndices[vL]
            const int iR = rl[c];
            const orbx_keypoint k = kr[iR];
            if (k.octave < kl.octave - 1 || k.octave > kl.octave + 1) continue;
            if (!(k.x >= minU && k.x <= maxU)) continue;
            const unsigned long long *db = (const unsigned long long *)(Rf.desc + ((size_t)fr * Rf.cap + iR) * 32);
            const int d = hamming256(a, db[0], db[1], db[2], db[3]);
            const uint32_t key = ((uint32_t)d << 16) | (uint32_t)iR;
            if (d < TH_HIGH && key < best) best = key;
        }
    }
    best = wave_min_u32(best);
    const int bd = (int)(best >> 16);
    const size_t o = (size_t)p * stride + iL;
    float outU = -1.0f, outZ = -1.0f;
    int outSad = -1;
    if (bd < (TH_HIGH + TH_LOW) / 2) {            // wave-uniform
        const int bidx = (int)(best & 0xffff);
        const int oct = kl.octave;
        const float uR0 = kr[bidx].x;
        const float sf = invScaleFactors[oct];
        const float scaleduL = roundf(kl.x * sf), scaledvL = roundf(kl.y * sf), scaleduR0 = roundf(uR0 * sf);
        const int w = 5, L = 5;
        int pitchL, pitchR, WL, WR;
        const uint8_t *imL = level_ptr(PL, fl, oct, pitchL, WL);
        const uint8_t *imR = level_ptr(PR, fr, oct, pitchR, WR);
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        if (!(iniu < 0 || endu >= (float)WR)) {
            const int cy = (int)scaledvL, cxl = (int)scaleduL;
            int part = 0;
            if (lane < 55) {
                const int s = lane % 11, q = lane / 11;
                const int cxr = (int)(scaleduR0 + (float)(s - L));
                const int cL = imL[(size_t)cy * pitchL + cxl], cR = imR[(size_t)cy * pitchR + cxr];
                for (int rr = q; rr < 11; rr += 5) {
                    const uint8_t *pl = imL + (size_t)(cy + rr - w) * pitchL + (cxl - w);
                    const uint8_t *pr = imR + (size_t)(cy + rr - w) * pitchR + (cxr - w);
#pragma unroll
                    for (int c = 0; c < 11; c++) {
                        const int va = (int)pl[c] - cL, vb = (int)pr[c] - cR;
                        const int df = va - vb;
                        part += df < 0 ? -df : df;
                    }
                }
            }
            // total per shift: lanes s, s+11, s+22, s+33, s+44
            int tot = 0;
#pragma unroll
            for (int q = 0; q < 5; q++) tot += __shfl(part, (lane % 11) + 11 * q);
            float vD[11];
#pragma unroll
            for (int s = 0; s < 11; s++) vD[s] = (float)__shfl(tot, s);
            int bestSad = 0x7fffffff, bestinc = 0;
#pragma unroll
            for (int s = 0; s < 11; s++)
                if (vD[s] < (float)bestSad) { bestSad = (int)vD[s]; bestinc = s - L; }
            if (!(bestinc == -L || bestinc == L)) {
                float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
                for (int s = 1; s < 10; s++)
                    if (s == bestinc + L) { d1 = vD[s - 1]; d2 = vD[s]; d3 = vD[s + 1]; }
                const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = scaleFactors[oct] * ((float)scaleduR0 + (float)bestinc + deltaR);
                    float disparity = kl.x - bestuR;
                    if (disparity >= minD && disparity < maxD) {
                        if (disparity <= 0) { disparity = (float)0.01; bestuR = (float)((double)kl.x - 0.01); }
                        outZ = mbf / disparity;
                        outU = bestuR;
                        outSad = bestSad;
                    }
                }
            }
        }
    }
    if (lane == 0) {
        bestDist[o] = bd;
        bestIdx[o] = bd < TH_HIGH ? (int)(best & 0xffff) : 0;
        uRight[o] = outU;
        depth[o] = outZ;
        sadOut[o] = outSad;
    }
}

__global__ __launch_bounds__(256) void k_stereo_full(FeatDev Lf, FeatDev Rf, PyrDev PL, PyrDev PR, const int32_t *__restrict__ pairsL,
                                                     const int32_t *__restrict__ pairsR, const float *__restrict__ scaleFactors,
                                                     const float *__restrict__ invScaleFactors, float mbf, float mb, int32_t *__restrict__ bestIdx,
                                                     int32_t *__restrict__ bestDist, float *__restrict__ uRight, float *__restrict__ depth,
                                                     int32_t *__restrict__ sadOut, int stride, const int32_t *__restrict__ rowStart, const int32_t *__restrict__ rowList, int H,
                                                     int listCap)
{
    const int p = blockIdx.y;
    stereo_full_body<false>(blockIdx.x * 4 + (threadIdx.x >> 6), p, pairsL[p], pairsR[p], Lf, Rf, PL, PR, scaleFactors, invScaleFactors, mbf, mb, bestIdx, bestDist, uRight,
                            depth, sadOut, stride, rowStart, rowList, H, listCap, nullptr);
}

__global__ __launch_bounds__(256) void k_stereo_cut(FeatDev Lf, const int32_t *__restrict__ pairsL, const int32_t *__restrict__ sad,
                                                    float *__restrict__ uRight, float *__restrict__ depth, int32_t *__restrict__ nmatches, int stride)
{
    __shared__ int hist[256];
    __shared__ int sel[4];   // count, chosen high byte, rank inside it, median
    const int p = blockIdx.x, fl = pairsL[p], t = threadIdx.x;
    const int nL = min(Lf.counts[fl], Lf.cap);
    const int32_t *sd = sad + (size_t)p * stride;
    hist[t] = 0;
    __syncthreads();
    for (int i = t; i < nL; i += 256) { const int v = sd[i]; if (v >= 0) atomicAdd(&hist[(v >> 8) & 255], 1); }
    __syncthreads();
    if (t == 0) {
        int n = 0;
        for (int b = 0; b < 256; b++) n += hist[b];
        sel[0] = n;
        int rank = n / 2, acc = 0, hb = 0;   // vDistIdx[size/2] of the ascending sort (:1387-1388)
        for (int b = 0; b < 256; b++) { if (acc + hist[b] > rank) { hb = b; break; } acc += hist[b]; }
        sel[1] = hb; sel[2] = rank - acc;
    }
    __syncthreads();
    const int n = sel[0], hb = sel[1];
    if (n == 0) { if (t == 0) nmatches[p] = 0; return; }   // reference: undefined behaviour (vDistIdx[0] of an empty vector); nothing to cut
    __syncthreads();
    hist[t] = 0;
    __syncthreads();
    for (int i = t; i < nL; i += 256) { const int v = sd[i]; if (v >= 0 && ((v >> 8) & 255) == hb) atomicAdd(&hist[v & 255], 1); }
    __syncthreads();
    if (t == 0) {
        int rank = sel[2], acc = 0, lb = 0;
        for (int b = 0; b < 256; b++) { if (acc + hist[b] > rank) { lb = b; break; } acc += hist[b]; }
        sel[3] = (hb << 8) | lb;
    }
    __syncthreads();
    const float median = (float)sel[3];
    const float thDist = 1.5f * 1.4f * median;
    int kept = 0;
    for (int i = t; i < nL; i += 256) {
        const int v = sd[i];
        if (v < 0) continue;
        if ((float)v < thDist) kept++;
        else { uRight[(size_t)p * stride + i] = -1.0f; depth[(size_t)p * stride + i] = -1.0f; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
    __syncthreads();
    if (t == 0) hist[0] = 0;
    __syncthreads();
    if ((t & 63) == 0) atomicAdd(&hist[0], kept);
    __syncthreads();
    if (t == 0) nmatches[p] = hist[0];
}

// ---------------------------------------------------------------------------------------------
// ONE stereo frame (pair 0 = frame 0 of both extractors): the latency form behind orbx_stereo_frame_begin, two launches.  The three
// kernels above cost 19 + 8 + 13 us of device time plus two launch gaps for a 1241x376 / 2000-feature frame, most of it ONE workgroup's
// dependent latency (k_stereo_rows, k_stereo_cut).  Here
//   k_stereo_scan     = k_stereo_full without a row table: every workgroup walks the right keypoints once (8 per thread) and collects, in
//                       LDS, those whose band holds the row of one of its four left keypoints (vRowIndices[vL], src/Frame.cc:1179-1196,
//                       evaluated where it is needed); 500 workgroups x 2000 band tests is nothing next to a kernel and a launch gap;
//   k_stereo_cut_one  = k_stereo_cut with the values in registers, the two 256-bin rank searches as wave scans instead of one thread's
//                       loops, and `seq` written into pinned host memory behind the last result: the host polls that word.
// (One launch with grid-wide hand-offs - workgroup 0 builds the table, the last one cuts - was tried first: agent-scope release / acquire
// on this eight-XCD part writes back / invalidates L2 per workgroup; 304 us per call, and every concurrent kernel slowed down with it.)
// Same arithmetic, same outputs as the three kernels (tests/test_matcher.py::test_stereo_frame_of_two_single_frame_calls).
// ---------------------------------------------------------------------------------------------
#define STEREO_ONE_REG 16      // sad values a thread keeps in registers (features <= 256 * 16; beyond: the three kernels)

// rank search in a 256-bin histogram (LDS), wave 0: the bin b with prefix(b) <= rank < prefix(b) + hist[b]; returns (b, rank - prefix(b), total) to all of wave 0
__device__ __forceinline__ void rank_bin_wave0(const int *hist, int rank, int lane, int *bin, int *within, int *total)
{
    const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    const int s4 = h0 + h1 + h2 + h3;
    int inc = s4;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    const int exc = inc - s4;
    *total = __shfl(inc, 63);
    int b = -1, w = 0;
    if (rank >= exc && rank < inc) {
        int a = exc;
        if (rank < a + h0) { b = 4 * lane; w = rank - a; }
        else if (rank < (a += h0) + h1) { b = 4 * lane + 1; w = rank - a; }
        else if (rank < (a += h1) + h2) { b = 4 * lane + 2; w = rank - a; }
        else { a += h2; b = 4 * lane + 3; w = rank - a; }
    }
    const unsigned long long m = __ballot(b >= 0);
    const int src = m ? __ffsll((long long)m) - 1 : 0;
    *bin = __shfl(b, src); *within = __shfl(w, src);
    if (!m) { *bin = 0; *within = 0; }
}

__global__ __launch_bounds__(256) void k_stereo_scan(FeatDev Lf, FeatDev Rf, PyrDev PL, PyrDev PR, const float *__restrict__ scaleFactors,
                                                     const float *__restrict__ invScaleFactors, float mbf, float mb, int32_t *__restrict__ bestIdx,
                                                     int32_t *__restrict__ bestDist, float *__restrict__ uRight, float *__restrict__ depth, int32_t *__restrict__ sadOut,
                                                     int stride, int H, int listCap)
{
    extern __shared__ int sdyn[];         // [4][listCap] candidate lists of the four left keypoints
    stereo_full_body<true>(blockIdx.x * 4 + (threadIdx.x >> 6), 0, 0, 0, Lf, Rf, PL, PR, scaleFactors, invScaleFactors, mbf, mb, bestIdx, bestDist, uRight, depth, sadOut, stride,
                           nullptr, nullptr, H, listCap, sdyn);
}

__global__ __launch_bounds__(256) void k_stereo_cut_one(FeatDev Lf, const int32_t *__restrict__ sadIn, float *__restrict__ uRight, float *__restrict__ depth,
                                                        int32_t *__restrict__ nmatches, int *__restrict__ hostFlag, int seq)
{
    __shared__ int hist[256];
    __shared__ int sel[4];
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nL = min(Lf.counts[0], Lf.cap);
    int sv[STEREO_ONE_REG];
    hist[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < STEREO_ONE_REG; j++) {
        const int i = tid + 256 * j;
        sv[j] = i < nL ? sadIn[i] : -1;
        if (sv[j] >= 0) atomicAdd(&hist[(sv[j] >> 8) & 255], 1);
    }
    __syncthreads();
    if (tid < 64) {
        int n = 0, b0, w0, hb = 0, within = 0, tot;
        rank_bin_wave0(hist, 0, lane, &b0, &w0, &n);             // (the total; rank = n / 2 needs it)
        rank_bin_wave0(hist, n / 2, lane, &hb, &within, &tot);   // vDistIdx[size/2] of the ascending sort (:1387-1388): its high byte
        if (tid == 0) { sel[0] = n; sel[1] = hb; sel[2] = within; }
    }
    __syncthreads();
    const int n = sel[0], hb = sel[1];
    int kept = 0;
    if (n > 0) {                                                  // (n == 0: nothing to cut, as k_stereo_cut)
        __syncthreads();
        hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < STEREO_ONE_REG; j++) if (sv[j] >= 0 && ((sv[j] >> 8) & 255) == hb) atomicAdd(&hist[sv[j] & 255], 1);
        __syncthreads();
        if (tid < 64) {
            int lb, w1, tot;
            rank_bin_wave0(hist, sel[2], lane, &lb, &w1, &tot);
            if (tid == 0) sel[3] = (hb << 8) | lb;
        }
        __syncthreads();
        const float median = (float)sel[3];
        const float thDist = 1.5f * 1.4f * median;
#pragma unroll
        for (int j = 0; j < STEREO_ONE_REG; j++) {
            const int v = sv[j], i = tid + 256 * j;
            if (v < 0) continue;
            if ((float)v < thDist) kept++;
            else { uRight[i] = -1.0f; depth[i] = -1.0f; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
        if (lane == 0) wsum[wv] = kept;
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        nmatches[0] = n > 0 ? wsum[0] + wsum[1] + wsum[2] + wsum[3] : 0;
        __hip_atomic_store(hostFlag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


}  // namespace

extern "C" int orbx_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 4; i++) {
        unsigned long long x, y;
        memcpy(&x, a + 8 * i, 8);
        memcpy(&y, b + 8 * i, 8);
        dist += __builtin_popcountll(x ^ y);
    }
    return dist;
}

extern "C" int orbx_matcher_create(int device, int max_features, int max_pairs, orbx_matcher **out)
{
    if (!out || max_features < 1 || max_features > 65535 || max_pairs < 1) { orbx_set_error("bad matcher arguments (1 <= max_features <= 65535)"); return ORBX_ERR_ARG; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { orbx_set_error("no HIP device available: liborbx has no CPU fallback"); return ORBX_ERR_NODEVICE; }
    if (device < 0 || device >= ndev) { orbx_set_error("device %d out of range", device); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(device));
    orbx_matcher *m = new orbx_matcher();
    m->device = device; m->maxFeatures = max_features; m->maxPairs = max_pairs;
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { delete m; orbx_set_error("hipStreamCreate failed"); return ORBX_ERR_HIP; }
    (void)hipEventCreateWithFlags(&m->evDep, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&m->evDep2, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&m->evPyr[0], hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&m->evPyr[1], hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&m->evDone[0], hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&m->evDone[1], hipEventDisableTiming);
    for (int r = 0; r < MATCH_PROF_RING; r++) { (void)hipEventCreate(&m->ev0[r]); (void)hipEventCreate(&m->ev1[r]); (void)hipEventCreate(&m->evMid[r]); }
    const size_t S = (size_t)max_features, P = (size_t)max_pairs;
    int rc;
    if ((rc = m->pairsA.ensure(P)) || (rc = m->pairsB.ensure(P)) || (rc = m->order.ensure(P * S)) || (rc = m->matches.ensure(P * S)) ||
        (rc = m->dists.ensure(P * S)) || (rc = m->nmatches.ensure(P)) || (rc = m->topk.ensure(P * S * TOPK)) || (rc = m->scales.ensure(128)) ||
        (rc = m->uright.ensure(P * S)) || (rc = m->depth.ensure(P * S)) || (rc = m->sad.ensure(P * S))) {
        orbx_matcher_destroy(m);
        return rc;
    }
    *out = m;
    return ORBX_OK;
}

extern "C" void orbx_matcher_destroy(orbx_matcher *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    m->pairsA.release(); m->pairsB.release(); m->order.release(); m->matches.release(); m->dists.release(); m->nmatches.release();
    m->hostStage.release(); m->projDec.release(); m->projQueue.release(); m->box.release(); m->arena.release();
    m->topk.release(); m->scales.release(); m->uright.release(); m->depth.release(); m->sad.release(); m->stRowStart.release(); m->stRowList.release();
    m->topk64.release(); m->pkp.release(); m->producerStatus.release();
    for (int q = 0; q < 2; q++) { m->pf[q].release(); m->pb[q].release(); m->pi32[q].release(); }
    if (m->evDep2) (void)hipEventDestroy(m->evDep2);
    m->sfZero.release(); m->sfScalesDev.release();
    if (m->sfHost) (void)hipHostFree(m->sfHost);
    for (int i = 0; i < 2; i++) if (m->evPyr[i]) (void)hipEventDestroy(m->evPyr[i]);
    for (int s = 0; s < 2; s++) { m->hk[s].release(); m->hd[s].release(); m->hv[s].release(); m->hc[s].release(); m->hg[s].release(); m->hs[s].release(); }
    m->triGeom.release(); m->frProj.release(); m->frLevel.release(); m->frInView.release();
    if (m->evDep) (void)hipEventDestroy(m->evDep);
    for (int i = 0; i < 2; i++) if (m->evDone[i]) (void)hipEventDestroy(m->evDone[i]);
    for (int r = 0; r < MATCH_PROF_RING; r++) { if (m->ev0[r]) (void)hipEventDestroy(m->ev0[r]); if (m->ev1[r]) (void)hipEventDestroy(m->ev1[r]); if (m->evMid[r]) (void)hipEventDestroy(m->evMid[r]); }
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

// a consumer chained behind an extractor inherits its capacity word on the device (no host round trip); the consumer's download reports it
// (`first` producer of a call overwrites the word - a consumer's status describes the call it belongs to, not its history -, a second one ORs)
__global__ void k_status_or(int *dst, const int *src, int first) { if (first) *dst = *src; else if (*src) atomicOr(dst, *src); }

namespace orbx_match {

int inherit_status(orbx_matcher *m, orbx_extractor *after, bool first)
{
    if (!after) return ORBX_OK;
    const int *w = orbx_extractor_status_word_internal(after);
    if (!w) return ORBX_OK;
    if (!m->producerStatus.p) {
        int rc = m->producerStatus.ensure(1);
        if (rc != ORBX_OK) return rc;
        ORBX_HIP_CHECK(hipMemsetAsync(m->producerStatus.p, 0, sizeof(int), m->stream));
    }
    hipLaunchKernelGGL(k_status_or, dim3(1), dim3(1), 0, m->stream, m->producerStatus.p, w, first ? 1 : 0);
    return ORBX_OK;
}

// checked by the download entry points (the stream has been synchronised): clears the word
int check_producer_status(orbx_matcher *m)
{
    if (!m->producerStatus.p) return ORBX_OK;
    int v = 0;
    ORBX_HIP_CHECK(hipMemcpy(&v, m->producerStatus.p, sizeof(int), hipMemcpyDeviceToHost));
    if (!v) return ORBX_OK;
    ORBX_HIP_CHECK(hipMemset(m->producerStatus.p, 0, sizeof(int)));
    orbx_set_error("the extractor batch these results were computed from overflowed a device capacity (bits 0x%x: 1 = >%d FAST candidates in a level, "
                   "2 = quadtree node list, 4 = level keypoints): results are not the reference's", v, 32768);
    return ORBX_ERR_CAPACITY;
}

int prep_pairs(orbx_matcher *m, const orbx_feature_set *a, const orbx_feature_set *b, const int32_t *pa, const int32_t *pb, int npairs,
                      orbx_extractor *after)
{
    if (!m || !a || !b || !pa || !pb) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (npairs < 1 || npairs > m->maxPairs) { orbx_set_error("npairs %d outside 1..%d", npairs, m->maxPairs); return ORBX_ERR_CAPACITY; }
    if (a->capacity > m->maxFeatures || b->capacity > m->maxFeatures || a->capacity < 1 || b->capacity < 1) {
        orbx_set_error("feature capacity %d/%d exceeds the matcher's max_features %d", a->capacity, b->capacity, m->maxFeatures);
        return ORBX_ERR_CAPACITY;
    }
    if (!a->keypoints || !a->descriptors || !a->counts || !b->keypoints || !b->descriptors || !b->counts) { orbx_set_error("NULL feature arrays"); return ORBX_ERR_ARG; }
    for (int p = 0; p < npairs; p++)
        if (pa[p] < 0 || pa[p] >= a->nframes || pb[p] < 0 || pb[p] >= b->nframes) { orbx_set_error("pair %d references a frame out of range", p); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    if (after) {
        ORBX_HIP_CHECK(hipEventRecord(m->evDep, orbx_extractor_stream_internal(after)));
        ORBX_HIP_CHECK(hipStreamWaitEvent(m->stream, m->evDep, 0));
        int rcs = inherit_status(m, after, true);
        if (rcs != ORBX_OK) return rcs;
    } else if (m->producerStatus.p)      // not chained: the word of an earlier chained call does not belong to these results
        ORBX_HIP_CHECK(hipMemsetAsync(m->producerStatus.p, 0, sizeof(int), m->stream));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pairsA.p, pa, (size_t)npairs * sizeof(int32_t), hipMemcpyHostToDevice, m->stream));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pairsB.p, pb, (size_t)npairs * sizeof(int32_t), hipMemcpyHostToDevice, m->stream));
    return ORBX_OK;
}

// The producer double-buffers its results: the batch extracted next does not touch what this call
// reads, the one after that does.  Hand the producer an event recorded behind the match kernels; it
// waits for it right before overwriting this buffer again, so matching batch i overlaps with the
// extraction of batch i+1 (the greedy replay occupies one wave per CU and is latency bound).
int chain_back(orbx_matcher *m, orbx_extractor *after)
{
    if (!after) return ORBX_OK;
    hipEvent_t ev = m->evDone[m->doneSlot];
    m->doneSlot ^= 1;
    ORBX_HIP_CHECK(hipEventRecord(ev, m->stream));
    orbx_extractor_set_consumer_event_internal(after, ev);
    return ORBX_OK;
}

FeatDev to_dev(const orbx_feature_set *s)
{
    FeatDev f;
    f.kp = s->keypoints; f.desc = s->descriptors; f.counts = s->counts; f.groups = s->groups; f.valid = s->valid; f.cap = s->capacity;
    return f;
}

}  // namespace orbx_match

extern "C" int orbx_search_by_bow_device(orbx_matcher *m, const orbx_feature_set *a, const orbx_feature_set *b, const int32_t *pairs_a,
                                         const int32_t *pairs_b, int npairs, const orbx_bow_params *params, orbx_extractor *after)
{
    if (!params || (params->mode != 0 && params->mode != 1)) { orbx_set_error("bad bow params"); return ORBX_ERR_ARG; }
    int rc = prep_pairs(m, a, b, pairs_a, pairs_b, npairs, after);
    if (rc != ORBX_OK) return rc;
    const int stride = m->maxFeatures;
    FeatDev A = to_dev(a), B = to_dev(b);
    const int slot = m->profCount % MATCH_PROF_RING;
    ORBX_HIP_CHECK(hipEventRecord(m->ev0[slot], m->stream));
    // first distance that can neither be accepted as best nor veto an acceptable best in the ratio test
    // `(float)best < nnratio * (float)second` (src/ORBmatcher.cc:309, 741), see k_bow_topk
    uint32_t dcut = TH_LOW + 1;
    while (dcut < 257 && !(params->nn_ratio * (float)dcut > (float)TH_LOW)) dcut++;
    const bool filter = b->groups != nullptr || (params->mode == 1 && b->valid != nullptr);
    const dim3 gridTopk((unsigned)((a->capacity + TOPK_ROWS - 1) / TOPK_ROWS), (unsigned)npairs);
    if (filter)
        hipLaunchKernelGGL((k_bow_topk<true, false, TOPK_NROW>), gridTopk, dim3(256), 0, m->stream, A, B, m->pairsA.p, m->pairsB.p, params->mode, dcut, m->topk.p, stride, TriDev());
    else {
        // BASELINE's north_star asks for the Hamming match on __popcll-class wavefront primitives, not on the matrix cores: k_bow_topk is what
        // runs.  ORBX_MATCH_MFMA=1 (read per call) selects k_bow_topk_mfma, the same lists from v_mfma_i32_32x32x32_i8 - a measured alternative
        // (DESIGN.md section 7), bit-identical, not the default.
        const char *em = getenv("ORBX_MATCH_MFMA");
        const bool mfma = em && em[0] == '1';
        static const int dbg = getenv("ORBX_MFMA_DEBUG") ? atoi(getenv("ORBX_MFMA_DEBUG")) : 0;
        if (mfma) hipLaunchKernelGGL(k_bow_topk_mfma, dim3((unsigned)((a->capacity + 255) / 256), (unsigned)npairs), dim3(256), 0, m->stream, A, B, m->pairsA.p, m->pairsB.p, dcut, m->topk.p, stride, dbg);
        else hipLaunchKernelGGL((k_bow_topk<false, false, TOPK_NROW>), gridTopk, dim3(256), 0, m->stream, A, B, m->pairsA.p, m->pairsB.p, params->mode, dcut, m->topk.p, stride, TriDev());
    }
    MLAUNCH_CHECK();
    ORBX_HIP_CHECK(hipEventRecord(m->evMid[slot], m->stream));
    m->midValid[slot] = true;
    // the processing order of the A features is an input of the replay only: behind the mid event, so that the first span of last_kernel_timing is
    // k_bow_topk alone (what a rocprofv3 kernel trace calls by that name) and the second the replay with its preparation
    hipLaunchKernelGGL(k_bow_order, dim3((unsigned)((a->capacity + 255) / 256), (unsigned)npairs), dim3(256), 0, m->stream, A, m->pairsA.p, m->order.p, stride);
    MLAUNCH_CHECK();
    const size_t ldsGreedy = (size_t)b->capacity * 4 + (size_t)a->capacity * 4 + (size_t)((a->capacity + 7) & ~7) * 2 * 2;
    if (ldsGreedy > 160 * 1024) { orbx_set_error("feature capacity %d too large for the LDS tile", a->capacity); return ORBX_ERR_CAPACITY; }
    if (ldsGreedy > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_bow_greedy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsGreedy));
    hipLaunchKernelGGL(k_bow_greedy, dim3((unsigned)npairs), dim3(GREEDY_THREADS), ldsGreedy, m->stream, A, B, m->pairsA.p, m->pairsB.p, params->mode, params->nn_ratio,
                       params->check_orientation, m->topk.p, m->order.p, m->matches.p, m->dists.p, m->nmatches.p, stride, TriDev(), (int32_t *)nullptr, (unsigned long long *)nullptr, 0ull);
    MLAUNCH_CHECK();
    ORBX_HIP_CHECK(hipEventRecord(m->ev1[slot], m->stream));
    m->profCount++;
    m->lastPairs = npairs; m->lastStride = stride;
    return chain_back(m, after);
}

extern "C" int orbx_stereo_match_device(orbx_matcher *m, const orbx_feature_set *left, const orbx_feature_set *right, const int32_t *pairs_l,
                                        const int32_t *pairs_r, int npairs, const float *scale_factors, int nlevels, float max_disparity,
                                        orbx_extractor *after)
{
    if (!scale_factors || nlevels < 1 || nlevels > 64) { orbx_set_error("bad scale factor table"); return ORBX_ERR_ARG; }
    int rc = prep_pairs(m, left, right, pairs_l, pairs_r, npairs, after);
    if (rc != ORBX_OK) return rc;
    const int stride = m->maxFeatures;
    ORBX_HIP_CHECK(hipMemcpyAsync(m->scales.p, scale_factors, (size_t)nlevels * sizeof(float), hipMemcpyHostToDevice, m->stream));
    ORBX_HIP_CHECK(hipMemsetAsync(m->nmatches.p, 0, (size_t)npairs * sizeof(int32_t), m->stream));
    const int slot = m->profCount % MATCH_PROF_RING;
    ORBX_HIP_CHECK(hipEventRecord(m->ev0[slot], m->stream));
    m->midValid[slot] = false;
    hipLaunchKernelGGL(k_stereo, dim3((unsigned)((left->capacity + 3) / 4), (unsigned)npairs), dim3(256), 0, m->stream, to_dev(left), to_dev(right), m->pairsA.p,
                       m->pairsB.p, m->scales.p, max_disparity, m->matches.p, m->dists.p, m->nmatches.p, stride);
    MLAUNCH_CHECK();
    ORBX_HIP_CHECK(hipEventRecord(m->ev1[slot], m->stream));
    m->profCount++;
    m->lastPairs = npairs; m->lastStride = stride;
    return chain_back(m, after);
}

extern "C" int orbx_compute_stereo_matches_device(orbx_matcher *m, orbx_extractor *left, orbx_extractor *right, const int32_t *frames_l,
                                                  const int32_t *frames_r, int npairs, float mbf, float mb)
{
    if (!m || !left || !right || !frames_l || !frames_r) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    OrbxLastBatchView vl, vr;
    int rc;
    if ((rc = orbx_extractor_last_batch_view_internal(left, &vl)) != ORBX_OK || (rc = orbx_extractor_last_batch_view_internal(right, &vr)) != ORBX_OK) return rc;
    if (vl.nlevels != vr.nlevels || vl.geom->W != vr.geom->W || vl.geom->H != vr.geom->H) {
        orbx_set_error("left and right extractors differ in geometry (%dx%d/%d vs %dx%d/%d)", vl.geom->W, vl.geom->H, vl.nlevels, vr.geom->W, vr.geom->H, vr.nlevels);
        return ORBX_ERR_ARG;
    }
    if (vl.nlevels > 64) { orbx_set_error("too many levels"); return ORBX_ERR_ARG; }
    orbx_feature_set fl = {vl.kp, vl.desc, vl.counts, nullptr, nullptr, vl.cap, vl.batch};
    orbx_feature_set fr = {vr.kp, vr.desc, vr.counts, nullptr, nullptr, vr.cap, vr.batch};
    rc = prep_pairs(m, &fl, &fr, frames_l, frames_r, npairs, left);
    if (rc != ORBX_OK) return rc;
    if (right != left) {
        ORBX_HIP_CHECK(hipEventRecord(m->evDep2, orbx_extractor_stream_internal(right)));
        ORBX_HIP_CHECK(hipStreamWaitEvent(m->stream, m->evDep2, 0));
        if ((rc = inherit_status(m, right, false)) != ORBX_OK) return rc;
    }
    const int stride = m->maxFeatures, nl = vl.nlevels;
    ORBX_HIP_CHECK(hipMemcpyAsync(m->scales.p, vl.scale, (size_t)nl * sizeof(float), hipMemcpyHostToDevice, m->stream));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->scales.p + 64, vl.invScale, (size_t)nl * sizeof(float), hipMemcpyHostToDevice, m->stream));
    PyrDev PL = {vl.img0, vl.img0Stride, vl.img0FramePitch, vl.pyr, vl.pyrBytes, vl.geomDev};
    PyrDev PR = {vr.img0, vr.img0Stride, vr.img0FramePitch, vr.pyr, vr.pyrBytes, vr.geomDev};
    const int slot = m->profCount % MATCH_PROF_RING;
    ORBX_HIP_CHECK(hipEventRecord(m->ev0[slot], m->stream));
    m->midValid[slot] = false;
    // vRowIndices of every right frame, then the per-keypoint search over the candidates of its row
    const int H = vr.geom->H, band = 2 * (int)std::ceil(2.0f * vl.scale[nl - 1]) + 3, listCap = vr.cap * band;
    if ((rc = m->stRowStart.ensure((size_t)npairs * (size_t)(H + 1))) != ORBX_OK || (rc = m->stRowList.ensure((size_t)npairs * (size_t)listCap)) != ORBX_OK) return rc;
    {
        const size_t lds = (size_t)2 * (H + 1) * sizeof(int);
        if (lds > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_stereo_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_stereo_rows, dim3((unsigned)npairs), dim3(256), lds, m->stream, to_dev(&fr), m->pairsB.p, m->scales.p, H, m->stRowStart.p, m->stRowList.p, listCap);
        MLAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_stereo_full, dim3((unsigned)((vl.cap + 3) / 4), (unsigned)npairs), dim3(256), 0, m->stream, to_dev(&fl), to_dev(&fr), PL, PR, m->pairsA.p,
                       m->pairsB.p, m->scales.p, m->scales.p + 64, mbf, mb, m->matches.p, m->dists.p, m->uright.p, m->depth.p, m->sad.p, stride,
                       (const int32_t *)m->stRowStart.p, (const int32_t *)m->stRowList.p, H, listCap);
    MLAUNCH_CHECK();
    hipLaunchKernelGGL(k_stereo_cut, dim3((unsigned)npairs), dim3(256), 0, m->stream, to_dev(&fl), m->pairsA.p, m->sad.p, m->uright.p, m->depth.p, m->nmatches.p, stride);
    MLAUNCH_CHECK();
    ORBX_HIP_CHECK(hipEventRecord(m->ev1[slot], m->stream));
    m->profCount++;
    m->lastPairs = npairs; m->lastStride = stride; m->lastStereoPairs = npairs;
    // the pyramids are single buffered: the extractors' next batch waits for these kernels
    ORBX_HIP_CHECK(hipEventRecord(m->evPyr[0], m->stream));
    orbx_extractor_set_pyramid_consumer_event_internal(left, m->evPyr[0]);
    if (right != left) orbx_extractor_set_pyramid_consumer_event_internal(right, m->evPyr[0]);
    rc = chain_back(m, left);
    if (rc == ORBX_OK && right != left) rc = chain_back(m, right);
    return rc;
}

// Frame::ComputeStereoMatches of ONE stereo frame whose two images were extracted by synchronous single-frame calls (the stereo Frame
// constructor: two ORBextractor::operator() calls, then this, src/Frame.cc:159-168).  Same kernels as orbx_compute_stereo_matches_device,
// none of its plumbing: both producers are complete (their calls returned), so no events are recorded or waited for and their capacity
// words are read from pinned memory; the scale tables and the pair indices stay on the device from call to call; mvuRight / mvDepth are
// written by the kernels straight into pinned memory.  Three launches and one synchronisation (the general call + download: ~22 runtime
// calls, 0.16-0.19 ms per frame).  Falls back to the general path when a producer's last call was not a single-frame call.
// begin: everything up to and including the launches (returns at once); end: the wait and the copy out of pinned memory.  Between the two the
// producers must not be called (their buffers are being read).  shim/Frame_hip.cc begins from the extractor thread that finishes LAST in the
// stereo constructor - before that thread converts its keypoints for the caller - and ends in Frame::ComputeStereoMatches.
extern "C" int orbx_stereo_frame_begin(orbx_matcher *m, orbx_extractor *left, orbx_extractor *right, float mbf, float mb)
{
    if (!m || !left || !right) { orbx_set_error("bad argument"); return ORBX_ERR_ARG; }
    m->sfPending = 0;
    int stL = 0, stR = 0;
    if (!orbx_extractor_host_complete_internal(left, &stL) || !orbx_extractor_host_complete_internal(right, &stR) || left == right) {
        const int32_t zero = 0;
        int rc = orbx_compute_stereo_matches_device(m, left, right, &zero, &zero, 1, mbf, mb);
        if (rc == ORBX_OK) m->sfPending = 2;      // (the general path: orbx_stereo_frame_end downloads)
        return rc;
    }
    if (stL | stR) {
        orbx_set_error("the extractor call these features come from overflowed a device capacity (bits 0x%x): results are not the reference's", stL | stR);
        return ORBX_ERR_CAPACITY;
    }
    OrbxLastBatchView vl, vr;
    int rc;
    if ((rc = orbx_extractor_last_batch_view_internal(left, &vl)) != ORBX_OK || (rc = orbx_extractor_last_batch_view_internal(right, &vr)) != ORBX_OK) return rc;
    if (vl.nlevels != vr.nlevels || vl.geom->W != vr.geom->W || vl.geom->H != vr.geom->H) {
        orbx_set_error("left and right extractors differ in geometry (%dx%d/%d vs %dx%d/%d)", vl.geom->W, vl.geom->H, vl.nlevels, vr.geom->W, vr.geom->H, vr.nlevels);
        return ORBX_ERR_ARG;
    }
    const int nl = vl.nlevels;
    if (nl > 64) { orbx_set_error("too many levels"); return ORBX_ERR_ARG; }
    if (vl.cap > m->maxFeatures || vr.cap > m->maxFeatures) { orbx_set_error("feature capacity %d/%d exceeds the matcher's max_features %d", vl.cap, vr.cap, m->maxFeatures); return ORBX_ERR_CAPACITY; }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    // constants that survive from call to call
    float tab[128];
    for (int i = 0; i < 64; i++) { tab[i] = i < nl ? vl.scale[i] : 0.f; tab[64 + i] = i < nl ? vl.invScale[i] : 0.f; }
    if (m->sfLevels != nl || memcmp(tab, m->sfScales, sizeof(tab)) != 0) {
        if ((rc = m->sfScalesDev.ensure(128)) != ORBX_OK) return rc;
        ORBX_HIP_CHECK(hipStreamSynchronize(st));
        ORBX_HIP_CHECK(hipMemcpy(m->sfScalesDev.p, tab, sizeof(tab), hipMemcpyHostToDevice));
        memcpy(m->sfScales, tab, sizeof(tab));
        m->sfLevels = nl;
    }
    if (!m->sfZero.p) {
        if ((rc = m->sfZero.ensure(4)) != ORBX_OK) return rc;
        ORBX_HIP_CHECK(hipMemset(m->sfZero.p, 0, 4 * sizeof(int32_t)));
    }
    const size_t S = (size_t)m->maxFeatures;
    if (m->sfHostFloats < 2 * S) {
        ORBX_HIP_CHECK(hipStreamSynchronize(st));
        if (m->sfHost) (void)hipHostFree(m->sfHost);
        m->sfHost = nullptr; m->sfHostFloats = 0;
        ORBX_HIP_CHECK(hipHostMalloc((void **)&m->sfHost, (2 * S + 16) * sizeof(float), hipHostMallocDefault));      // uright | depth | completion word
        void *dp = nullptr;
        ORBX_HIP_CHECK(hipHostGetDevicePointer(&dp, m->sfHost, 0));
        m->sfHostDev = (float *)dp;
        m->sfHostFloats = 2 * S;
        *(volatile int *)(m->sfHost + 2 * S) = 0;
    }
    orbx_feature_set fl = {vl.kp, vl.desc, vl.counts, nullptr, nullptr, vl.cap, vl.batch};
    orbx_feature_set fr = {vr.kp, vr.desc, vr.counts, nullptr, nullptr, vr.cap, vr.batch};
    PyrDev PL = {vl.img0, vl.img0Stride, vl.img0FramePitch, vl.pyr, vl.pyrBytes, vl.geomDev};
    PyrDev PR = {vr.img0, vr.img0Stride, vr.img0FramePitch, vr.pyr, vr.pyrBytes, vr.geomDev};
    const int stride = m->maxFeatures;
    const int H = vr.geom->H, band = 2 * (int)std::ceil(2.0f * vl.scale[nl - 1]) + 3, listCap = vr.cap * band;
    if ((rc = m->stRowStart.ensure((size_t)(H + 1))) != ORBX_OK || (rc = m->stRowList.ensure((size_t)listCap)) != ORBX_OK) return rc;
    const size_t lds = (size_t)2 * (H + 1) * sizeof(int);
    if (lds > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_stereo_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int32_t *zero = m->sfZero.p;
    float *dU = m->sfHostDev, *dD = m->sfHostDev + S;
    const float *sc = m->sfScalesDev.p;
    m->sfFlagSeq = 0;
    {   // two launches (k_stereo_scan, k_stereo_cut_one); ORBX_STEREO_ONE=0: the three kernels of the batched path (measurement switch)
        const char *e = getenv("ORBX_STEREO_ONE");
        const int scanCap = vr.cap;                              // a row's candidate list can hold every right keypoint
        const size_t lds1 = (size_t)4 * scanCap * sizeof(int);
        if (!(e && e[0] == '0') && vl.cap <= 256 * STEREO_ONE_REG && lds1 <= 96 * 1024) {
            if (lds1 > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_stereo_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
            const int seq = ++m->sfSeq;
            hipLaunchKernelGGL(k_stereo_scan, dim3((unsigned)((vl.cap + 3) / 4)), dim3(256), lds1, st, to_dev(&fl), to_dev(&fr), PL, PR, sc, sc + 64, mbf, mb, m->matches.p, m->dists.p,
                               dU, dD, m->sad.p, stride, H, scanCap);
            MLAUNCH_CHECK();
            hipLaunchKernelGGL(k_stereo_cut_one, dim3(1), dim3(256), 0, st, to_dev(&fl), (const int32_t *)m->sad.p, dU, dD, m->nmatches.p, (int *)(m->sfHostDev + 2 * S), seq);
            MLAUNCH_CHECK();
            m->sfFlagSeq = seq;
            m->lastPairs = 1; m->lastStride = stride;
            m->sfPending = 1;
            return ORBX_OK;
        }
    }
    hipLaunchKernelGGL(k_stereo_rows, dim3(1), dim3(256), lds, st, to_dev(&fr), zero, sc, H, m->stRowStart.p, m->stRowList.p, listCap);
    MLAUNCH_CHECK();
    hipLaunchKernelGGL(k_stereo_full, dim3((unsigned)((vl.cap + 3) / 4), 1u), dim3(256), 0, st, to_dev(&fl), to_dev(&fr), PL, PR, zero, zero, sc, sc + 64, mbf, mb,
                       m->matches.p, m->dists.p, dU, dD, m->sad.p, stride, (const int32_t *)m->stRowStart.p, (const int32_t *)m->stRowList.p, H, listCap);
    MLAUNCH_CHECK();
    hipLaunchKernelGGL(k_stereo_cut, dim3(1), dim3(256), 0, st, to_dev(&fl), zero, m->sad.p, dU, dD, m->nmatches.p, stride);
    MLAUNCH_CHECK();
    m->lastPairs = 1; m->lastStride = stride;
    m->sfPending = 1;
    return ORBX_OK;
}

extern "C" int orbx_stereo_frame_end(orbx_matcher *m, float *uright, float *depth, int n)
{
    if (!m || !uright || !depth || n < 0) { orbx_set_error("bad argument"); return ORBX_ERR_ARG; }
    const int pending = m->sfPending;
    m->sfPending = 0;
    if (!pending) { orbx_set_error("no stereo frame has been begun"); return ORBX_ERR_STATE; }
    if (pending == 2) return orbx_stereo_download(m, 1, uright, depth, n < m->maxFeatures ? n : m->maxFeatures);
    const size_t S = (size_t)m->maxFeatures, cnt = (size_t)n < S ? (size_t)n : S;
    bool arrived = false;
    if (m->sfFlagSeq) {      // k_stereo_one writes its sequence number into pinned memory behind its last result: no runtime call on the way out
        const volatile int *flag = (const volatile int *)(m->sfHost + 2 * S);
        const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        for (int spin = 0; !(arrived = *flag == m->sfFlagSeq); spin++) {
            __builtin_ia32_pause();
            if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!arrived) {
        ORBX_HIP_CHECK(hipSetDevice(m->device));
        ORBX_HIP_CHECK(hipStreamSynchronize(m->stream));      // complete on return: the extractors may overwrite their buffers at will
    }
    memcpy(uright, m->sfHost, cnt * sizeof(float));
    memcpy(depth, m->sfHost + S, cnt * sizeof(float));
    return ORBX_OK;
}

extern "C" int orbx_stereo_frame(orbx_matcher *m, orbx_extractor *left, orbx_extractor *right, float mbf, float mb, float *uright, float *depth, int n)
{
    if (!m || !left || !right || !uright || !depth || n < 0) { orbx_set_error("bad argument"); return ORBX_ERR_ARG; }
    const int rc = orbx_stereo_frame_begin(m, left, right, mbf, mb);
    return rc != ORBX_OK ? rc : orbx_stereo_frame_end(m, uright, depth, n);
}

extern "C" int orbx_stereo_results_device(orbx_matcher *m, const float **uright_dev, const float **depth_dev, int *stride)
{
    if (!m) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (!m->lastStereoPairs) { orbx_set_error("no orbx_compute_stereo_matches_device call yet"); return ORBX_ERR_STATE; }
    if (uright_dev) *uright_dev = m->uright.p;
    if (depth_dev) *depth_dev = m->depth.p;
    if (stride) *stride = m->lastStride;
    return ORBX_OK;
}

extern "C" int orbx_stereo_download(orbx_matcher *m, int npairs, float *uright, float *depth, int stride)
{
    if (!m) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (npairs < 1 || npairs > m->lastStereoPairs) { orbx_set_error("npairs not available"); return ORBX_ERR_STATE; }
    if (stride < 1 || stride > m->lastStride) { orbx_set_error("bad stride"); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    ORBX_HIP_CHECK(hipStreamSynchronize(m->stream));
    { int rcs = orbx_match::check_producer_status(m); if (rcs != ORBX_OK) return rcs; }
    if (uright) ORBX_HIP_CHECK(hipMemcpy2D(uright, (size_t)stride * 4, m->uright.p, (size_t)m->lastStride * 4, (size_t)stride * 4, (size_t)npairs, hipMemcpyDeviceToHost));
    if (depth) ORBX_HIP_CHECK(hipMemcpy2D(depth, (size_t)stride * 4, m->depth.p, (size_t)m->lastStride * 4, (size_t)stride * 4, (size_t)npairs, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

extern "C" int orbx_matcher_results_device(orbx_matcher *m, const int32_t **matches_dev, const int32_t **dists_dev, const int32_t **nmatches_dev, int *stride)
{
    if (!m) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (!m->lastPairs) { orbx_set_error("no match call yet"); return ORBX_ERR_STATE; }
    if (matches_dev) *matches_dev = m->matches.p;
    if (dists_dev) *dists_dev = m->dists.p;
    if (nmatches_dev) *nmatches_dev = m->nmatches.p;
    if (stride) *stride = m->lastStride;
    return ORBX_OK;
}

extern "C" int orbx_matcher_sync(orbx_matcher *m)
{
    if (!m) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    ORBX_HIP_CHECK(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

extern "C" int orbx_matcher_download(orbx_matcher *m, int npairs, int32_t *matches, int32_t *dists, int stride, int32_t *nmatches)
{
    if (!m) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (npairs < 1 || npairs > m->lastPairs) { orbx_set_error("npairs not available"); return ORBX_ERR_STATE; }
    if (stride < 1 || stride > m->lastStride) { orbx_set_error("bad stride"); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    ORBX_HIP_CHECK(hipStreamSynchronize(m->stream));
    { int rcs = orbx_match::check_producer_status(m); if (rcs != ORBX_OK) return rcs; }
    if (matches) ORBX_HIP_CHECK(hipMemcpy2D(matches, (size_t)stride * 4, m->matches.p, (size_t)m->lastStride * 4, (size_t)stride * 4, (size_t)npairs, hipMemcpyDeviceToHost));
    if (dists) ORBX_HIP_CHECK(hipMemcpy2D(dists, (size_t)stride * 4, m->dists.p, (size_t)m->lastStride * 4, (size_t)stride * 4, (size_t)npairs, hipMemcpyDeviceToHost));
    if (nmatches) ORBX_HIP_CHECK(hipMemcpy(nmatches, m->nmatches.p, (size_t)npairs * 4, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

extern "C" int orbx_matcher_last_timing(orbx_matcher *m, float *total_ms)
{
    if (!m || !total_ms) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (m->profCount == 0) { orbx_set_error("no timing available"); return ORBX_ERR_STATE; }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    ORBX_HIP_CHECK(hipStreamSynchronize(m->stream));
    const int n = m->profCount < MATCH_PROF_RING ? m->profCount : MATCH_PROF_RING;
    float acc = 0.f, accD = 0.f, accR = 0.f;
    int nMid = 0;
    for (int r = 0; r < n; r++) {
        float ms = 0.f;
        ORBX_HIP_CHECK(hipEventElapsedTime(&ms, m->ev0[r], m->ev1[r]));
        acc += ms;
        if (m->midValid[r]) {
            float a = 0.f, b = 0.f;
            ORBX_HIP_CHECK(hipEventElapsedTime(&a, m->ev0[r], m->evMid[r]));
            ORBX_HIP_CHECK(hipEventElapsedTime(&b, m->evMid[r], m->ev1[r]));
            accD += a; accR += b; nMid++;
        }
    }
    *total_ms = acc / (float)n;
    m->lastDistanceMs = nMid ? accD / (float)nMid : 0.f;
    m->lastReplayMs = nMid ? accR / (float)nMid : 0.f;
    m->profCount = 0;   // the next call starts a new average
    return ORBX_OK;
}

extern "C" int orbx_matcher_last_kernel_timing(orbx_matcher *m, float *distance_ms, float *replay_ms)
{
    if (!m) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (distance_ms) *distance_ms = m->lastDistanceMs;
    if (replay_ms) *replay_ms = m->lastReplayMs;
    return ORBX_OK;
}

// ---- host-array convenience: stage one frame per side, run, download ----
namespace orbx_match {
int stage_host(orbx_matcher *m, int side, const orbx_feature_set *h, orbx_feature_set *d)
{
    // the five arrays of a side go through the pinned staging buffer and ONE copy (ten small copies from pageable memory
    // cost more than the kernels of a single-pair call): side 0 opens the buffer, sized for two full feature sets
    if (!h || !h->keypoints || !h->descriptors || !h->counts) { orbx_set_error("NULL feature arrays"); return ORBX_ERR_ARG; }
    const int n = h->counts[0];
    if (n < 0 || n > m->maxFeatures) { orbx_set_error("feature count %d exceeds max_features %d", n, m->maxFeatures); return ORBX_ERR_CAPACITY; }
    OrbxHostStage &hs = m->hostStage;
    const size_t cap = (size_t)m->maxFeatures;
    int rc;
    if (side == 0) {
        const size_t perSide = hs.padded(cap * sizeof(orbx_keypoint)) + hs.padded(cap * 32) + hs.padded(4) + hs.padded(cap * 4) + hs.padded(cap);
        ORBX_HIP_CHECK(hipStreamSynchronize(m->stream));   // the previous call's copy has left the pinned buffer
        if ((rc = hs.begin(2 * perSide)) != ORBX_OK) return rc;
    }
    const size_t first = hs.used;
    const int32_t cnt = n;
    d->keypoints = hs.put(h->keypoints, (size_t)n);
    d->descriptors = hs.put(h->descriptors, (size_t)n * 32);
    d->counts = hs.put(&cnt, 1);
    const int32_t *g = hs.put(h->groups, h->groups ? (size_t)n : 0);
    const uint8_t *v = hs.put(h->valid, h->valid ? (size_t)n : 0);
    d->groups = h->groups ? g : nullptr; d->valid = h->valid ? v : nullptr;
    d->capacity = m->maxFeatures; d->nframes = 1;
    ORBX_HIP_CHECK(hipMemcpyAsync(hs.dev.p + first, hs.host + first, hs.used - first, hipMemcpyHostToDevice, m->stream));
    return ORBX_OK;
}
}  // namespace orbx_match

// One pair from host arrays, the call src/Tracking.cc:1195 / 2073 make through shim/ORBmatcher_hip.cc.  No copy engine, no stream synchronisation
// (OrbxCallBox): the arrays go into mapped pinned memory, k_stage_copy reads them once into the handle's arena, k_bow_topk_pair (B split over the
// chip, the processing order in the same launch) and k_bow_greedy follow, the replay writes the match list into mapped pinned memory and raises the
// call's sequence word.  204 us -> see profiles/r06_latency_calls.txt.  ORBX_BOW_SINGLE_SPLIT=0: the round-5 path (stage, the batch kernels, download).
static int search_by_bow_single(orbx_matcher *m, const orbx_feature_set *a, const orbx_feature_set *b, const orbx_bow_params *prm, int32_t *matches, int32_t *nmatches)
{
    const int nA = a->counts[0], nB = b->counts[0];
    OrbxCallBox &bx = m->box;
    const size_t NA = (size_t)nA, NB = (size_t)nB;
    const int nOut = prm->mode == 0 ? nB : nA;
    const size_t inBytes = bx.padded(NA * sizeof(orbx_keypoint)) + bx.padded(NA * 32) + bx.padded(NA * 4) + bx.padded(NA) + bx.padded(NB * sizeof(orbx_keypoint)) + bx.padded(NB * 32) +
                           bx.padded(NB * 4) + bx.padded(NB) + bx.padded(8);
    int rc = bx.begin(inBytes, bx.padded(((size_t)nOut + 1) * 4), m->stream);
    if (rc != ORBX_OK) return rc;
    if ((rc = m->arena.ensure(inBytes)) != ORBX_OK) return rc;
    uint8_t *const ar = m->arena.p;
    auto dev = [&](const void *boxAddr) { return ar + ((const uint8_t *)boxAddr - bx.inDev); };      // the same offsets in the device arena
    const int32_t cnt[2] = {nA, nB};
    FeatDev A, B;
    A.kp = (const orbx_keypoint *)dev(bx.put(a->keypoints, NA)); A.desc = dev(bx.put(a->descriptors, NA * 32));
    const void *ga = bx.put(a->groups, a->groups ? NA : 0), *va = bx.put(a->valid, a->valid ? NA : 0);
    B.kp = (const orbx_keypoint *)dev(bx.put(b->keypoints, NB)); B.desc = dev(bx.put(b->descriptors, NB * 32));
    const void *gb = bx.put(b->groups, b->groups ? NB : 0), *vb = bx.put(b->valid, b->valid ? NB : 0);
    const int32_t *dc = (const int32_t *)dev(bx.put(cnt, 2));
    A.groups = a->groups ? (const int32_t *)dev(ga) : nullptr; A.valid = a->valid ? dev(va) : nullptr; A.counts = dc; A.cap = nA;
    B.groups = b->groups ? (const int32_t *)dev(gb) : nullptr; B.valid = b->valid ? dev(vb) : nullptr; B.counts = dc + 1; B.cap = nB;
    const size_t n16 = bx.used / 16;
    hipLaunchKernelGGL(k_stage_copy, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 512)), dim3(256), 0, m->stream, (const uint4 *)bx.inDev, (uint4 *)ar, n16);
    MLAUNCH_CHECK();
    uint32_t dcut = TH_LOW + 1;
    while (dcut < 257 && !(prm->nn_ratio * (float)dcut > (float)TH_LOW)) dcut++;
    const bool filter = b->groups != nullptr || (prm->mode == 1 && b->valid != nullptr);
    // geometry: 16 A features per workgroup (4 per wave x 16 B slices), all of B in the workgroup's LDS; the order's workgroups behind them
    const int nRowBlocks = (nA + PAIR_ROWS - 1) / PAIR_ROWS, nOrder = (nA + 31) / 32;
    const int per = (nB + PAIR_SLICES - 1) / PAIR_SLICES;
    const size_t ldsPair = std::max((size_t)PAIR_SLICES * ((size_t)per * 32 + 16) + (size_t)PAIR_SLICES * per * 4, (size_t)4096 * 4);
    if ((rc = m->topk.ensure(NA * TOPK)) != ORBX_OK || (rc = m->order.ensure(NA)) != ORBX_OK) return rc;
    const dim3 grid((unsigned)(nRowBlocks + nOrder));
    if (filter) {
        if (ldsPair > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_bow_topk_pair<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsPair));
        hipLaunchKernelGGL(k_bow_topk_pair<true>, grid, dim3(256), ldsPair, m->stream, A, B, prm->mode, dcut, nRowBlocks, per, m->topk.p, m->order.p);
    } else {
        if (ldsPair > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_bow_topk_pair<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsPair));
        hipLaunchKernelGGL(k_bow_topk_pair<false>, grid, dim3(256), ldsPair, m->stream, A, B, prm->mode, dcut, nRowBlocks, per, m->topk.p, m->order.p);
    }
    MLAUNCH_CHECK();
    const size_t ldsGreedy = NB * 4 + NA * 4 + (size_t)((nA + 7) & ~7) * 2 * 2 + NB * 4 + std::max(NA, NB) * 4;      // (+ the B angles and the match list: k_bow_greedy with pubFlag)
    if (ldsGreedy > 160 * 1024) { orbx_set_error("feature count %d too large for the LDS tile", nA); return ORBX_ERR_CAPACITY; }
    if (ldsGreedy > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_bow_greedy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsGreedy));
    if (!m->sfZero.p) {
        if ((rc = m->sfZero.ensure(4)) != ORBX_OK) return rc;
        ORBX_HIP_CHECK(hipMemsetAsync(m->sfZero.p, 0, 4 * sizeof(int32_t), m->stream));
    }
    const int stride = m->maxFeatures;
    const unsigned long long seq = bx.arm();
    hipLaunchKernelGGL(k_bow_greedy, dim3(1), dim3(GREEDY_THREADS), ldsGreedy, m->stream, A, B, (const int32_t *)m->sfZero.p, (const int32_t *)m->sfZero.p, prm->mode, prm->nn_ratio,
                       prm->check_orientation, m->topk.p, m->order.p, m->matches.p, m->dists.p, m->nmatches.p, stride, TriDev(), bx.outDev<int32_t>(0), bx.flagDev, seq);
    MLAUNCH_CHECK();
    m->lastPairs = 1; m->lastStride = stride;
    if ((rc = bx.wait(m->stream)) != ORBX_OK) return rc;
    const int32_t *res = bx.outHost<int32_t>(0);
    if (nOut > 0) memcpy(matches, res, (size_t)nOut * 4);
    *nmatches = res[nOut];
    return ORBX_OK;
}

extern "C" int orbx_search_by_bow(orbx_matcher *m, const orbx_feature_set *a_host, const orbx_feature_set *b_host, const orbx_bow_params *params,
                                  int32_t *matches, int32_t *nmatches)
{
    if (!m || !matches || !nmatches) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!params || (params->mode != 0 && params->mode != 1)) { orbx_set_error("bad bow params"); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    static const bool split = !(getenv("ORBX_BOW_SINGLE_SPLIT") && getenv("ORBX_BOW_SINGLE_SPLIT")[0] == '0');
    if (split && a_host && b_host && a_host->keypoints && a_host->descriptors && a_host->counts && b_host->keypoints && b_host->descriptors && b_host->counts &&
        a_host->counts[0] > 0 && b_host->counts[0] > 0 && a_host->counts[0] <= m->maxFeatures && b_host->counts[0] <= std::min(m->maxFeatures, 4000))      // (B in LDS: 36 bytes per feature)
        return search_by_bow_single(m, a_host, b_host, params, matches, nmatches);
    orbx_feature_set da, db;
    int rc;
    if ((rc = stage_host(m, 0, a_host, &da)) != ORBX_OK || (rc = stage_host(m, 1, b_host, &db)) != ORBX_OK) return rc;
    const int32_t zero = 0;
    if ((rc = orbx_search_by_bow_device(m, &da, &db, &zero, &zero, 1, params, nullptr)) != ORBX_OK) return rc;
    ORBX_HIP_CHECK(hipStreamSynchronize(m->stream));
    const int nOut = params->mode == 0 ? b_host->counts[0] : a_host->counts[0];
    if (nOut > 0) ORBX_HIP_CHECK(hipMemcpy(matches, m->matches.p, (size_t)nOut * 4, hipMemcpyDeviceToHost));
    ORBX_HIP_CHECK(hipMemcpy(nmatches, m->nmatches.p, 4, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

// ORBmatcher::SearchForTriangulation, src/ORBmatcher.cc:810-1017: the SearchByBoW machinery (node order,
// candidate lists, fixed-point replay, rotation histogram) with the triangulation acceptance rule.
extern "C" int orbx_search_for_triangulation_device(orbx_matcher *m, const orbx_feature_set *a, const orbx_feature_set *b, const int32_t *pairs_a,
                                                    const int32_t *pairs_b, int npairs, const orbx_triangulation_params *params, orbx_extractor *after)
{
    if (!params || !params->f12 || !params->epipole || !params->scale_factors || !params->level_sigma2) { orbx_set_error("bad triangulation params"); return ORBX_ERR_ARG; }
    if (params->nlevels < 1 || params->nlevels > ORBX_MAX_LEVELS) { orbx_set_error("nlevels %d outside 1..%d", params->nlevels, ORBX_MAX_LEVELS); return ORBX_ERR_ARG; }
    int rc = prep_pairs(m, a, b, pairs_a, pairs_b, npairs, after);
    if (rc != ORBX_OK) return rc;
    if ((rc = m->triGeom.ensure((size_t)11 * m->maxPairs)) != ORBX_OK) return rc;
    ORBX_HIP_CHECK(hipMemcpyAsync(m->triGeom.p, params->f12, (size_t)9 * npairs * sizeof(float), hipMemcpyHostToDevice, m->stream));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->triGeom.p + (size_t)9 * m->maxPairs, params->epipole, (size_t)2 * npairs * sizeof(float), hipMemcpyHostToDevice, m->stream));
    TriDev T = TriDev();
    T.f12 = m->triGeom.p; T.epipole = m->triGeom.p + (size_t)9 * m->maxPairs;
    T.stereoA = params->stereo_a; T.stereoB = params->stereo_b;
    for (int l = 0; l < params->nlevels; l++) {
        T.epiTh[l] = 100 * params->scale_factors[l];     // int * float, :893
        T.chiTh[l] = 3.84 * params->level_sigma2[l];     // double * float, :224
    }
    const int stride = m->maxFeatures;
    FeatDev A = to_dev(a), B = to_dev(b);
    const int slot = m->profCount % MATCH_PROF_RING;
    ORBX_HIP_CHECK(hipEventRecord(m->ev0[slot], m->stream));
    hipLaunchKernelGGL(k_bow_order, dim3((unsigned)((a->capacity + 255) / 256), (unsigned)npairs), dim3(256), 0, m->stream, A, m->pairsA.p, m->order.p, stride);
    MLAUNCH_CHECK();
    const dim3 gridTopk((unsigned)((a->capacity + TOPK_ROWS - 1) / TOPK_ROWS), (unsigned)npairs);
    // only dist <= TH_LOW can be accepted (:880) and there is no second-best test: cut at TH_LOW + 1
    hipLaunchKernelGGL((k_bow_topk<true, true, TOPK_NROW>), gridTopk, dim3(256), 0, m->stream, A, B, m->pairsA.p, m->pairsB.p, 2, (uint32_t)(TH_LOW + 1), m->topk.p, stride, T);
    MLAUNCH_CHECK();
    m->midValid[slot] = false;
    const size_t ldsGreedy = (size_t)b->capacity * 4 + (size_t)a->capacity * 4 + (size_t)((a->capacity + 7) & ~7) * 2 * 2;
    if (ldsGreedy > 160 * 1024) { orbx_set_error("feature capacity %d too large for the LDS tile", a->capacity); return ORBX_ERR_CAPACITY; }
    if (ldsGreedy > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_bow_greedy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsGreedy));
    hipLaunchKernelGGL(k_bow_greedy, dim3((unsigned)npairs), dim3(GREEDY_THREADS), ldsGreedy, m->stream, A, B, m->pairsA.p, m->pairsB.p, 2, 0.0f, params->check_orientation,
                       m->topk.p, m->order.p, m->matches.p, m->dists.p, m->nmatches.p, stride, T, (int32_t *)nullptr, (unsigned long long *)nullptr, 0ull);
    MLAUNCH_CHECK();
    ORBX_HIP_CHECK(hipEventRecord(m->ev1[slot], m->stream));
    m->profCount++;
    m->lastPairs = npairs; m->lastStride = stride;
    return chain_back(m, after);
}

extern "C" int orbx_search_for_triangulation(orbx_matcher *m, const orbx_feature_set *a_host, const orbx_feature_set *b_host,
                                             const orbx_triangulation_params *params_host, int32_t *matches12, int32_t *nmatches)
{
    if (!m || !params_host || !matches12 || !nmatches) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    orbx_feature_set da, db;
    int rc;
    if ((rc = stage_host(m, 0, a_host, &da)) != ORBX_OK || (rc = stage_host(m, 1, b_host, &db)) != ORBX_OK) return rc;
    orbx_triangulation_params P = *params_host;
    const uint8_t *hostFlags[2] = {params_host->stereo_a, params_host->stereo_b};
    const int cnt[2] = {a_host->counts[0], b_host->counts[0]};
    const uint8_t *devFlags[2] = {nullptr, nullptr};
    for (int sd = 0; sd < 2; sd++) {
        if (!hostFlags[sd]) continue;
        if ((rc = m->hs[sd].ensure((size_t)m->maxFeatures)) != ORBX_OK) return rc;
        if (cnt[sd] > 0) ORBX_HIP_CHECK(hipMemcpyAsync(m->hs[sd].p, hostFlags[sd], (size_t)cnt[sd], hipMemcpyHostToDevice, m->stream));
        devFlags[sd] = m->hs[sd].p;
    }
    P.stereo_a = devFlags[0]; P.stereo_b = devFlags[1];
    const int32_t zero = 0;
    if ((rc = orbx_search_for_triangulation_device(m, &da, &db, &zero, &zero, 1, &P, nullptr)) != ORBX_OK) return rc;
    ORBX_HIP_CHECK(hipStreamSynchronize(m->stream));
    if (cnt[0] > 0) ORBX_HIP_CHECK(hipMemcpy(matches12, m->matches.p, (size_t)cnt[0] * 4, hipMemcpyDeviceToHost));
    ORBX_HIP_CHECK(hipMemcpy(nmatches, m->nmatches.p, 4, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

extern "C" int orbx_stereo_match(orbx_matcher *m, const orbx_feature_set *left_host, const orbx_feature_set *right_host, const float *scale_factors,
                                 int nlevels, float max_disparity, int32_t *best_dist, int32_t *best_idx)
{
    if (!m || !best_dist || !best_idx) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    orbx_feature_set dl, dr;
    int rc;
    if ((rc = stage_host(m, 0, left_host, &dl)) != ORBX_OK || (rc = stage_host(m, 1, right_host, &dr)) != ORBX_OK) return rc;
    const int32_t zero = 0;
    if ((rc = orbx_stereo_match_device(m, &dl, &dr, &zero, &zero, 1, scale_factors, nlevels, max_disparity, nullptr)) != ORBX_OK) return rc;
    ORBX_HIP_CHECK(hipStreamSynchronize(m->stream));
    const int nL = left_host->counts[0];
    if (nL > 0) {
        ORBX_HIP_CHECK(hipMemcpy(best_idx, m->matches.p, (size_t)nL * 4, hipMemcpyDeviceToHost));
        ORBX_HIP_CHECK(hipMemcpy(best_dist, m->dists.p, (size_t)nL * 4, hipMemcpyDeviceToHost));
    }
    return ORBX_OK;
}
