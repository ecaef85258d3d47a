// orbx_match_proj.hip -- the projection-guided matchers of ORBmatcher on gfx950: SearchByProjection (all four overloads),
// Fuse (both), SearchBySim3 (through the Fuse search) and SearchForInitialization.  A feature's grid cell and the
// GetFeaturesInArea window are evaluated arithmetically per (query, feature); candidate order = 64-bit key
// distance || cell column || cell row || index.  See orbx_match.hip for SearchByBoW / SearchForTriangulation / stereo.
#include <math.h>
#include <string.h>

#include <vector>

#include "orbx_match_internal.h"

using namespace orbx_match;

namespace {

// ---------------------------------------------------------------------------------------------
// ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th)
// (src/ORBmatcher.cc:70-175), the matcher of Tracking::SearchLocalPoints, with
// Frame::GetFeaturesInArea (src/Frame.cc:741-850) and the 64x48 feature grid (:461-491, 853-877).
// The grid is never materialised: a feature's cell is round((pt - min) * inv) and "in the search
// window" is the same cell-range + |dx|,|dy| < r test the reference applies, evaluated per
// (map point, feature); the reference's candidate order (cell column, cell row, feature index)
// becomes the low half of a 64-bit key whose high half is the Hamming distance.
// k_proj_topk:   wave per map point, lanes over the frame's features -> its 8 smallest keys.
// k_proj_greedy: workgroup per frame; wave 0 replays the map points in order (a feature taken by a
//                point with observations blocks later points, :110-112), 8 points per 512-byte list
//                fetch, exact wave-parallel rescan when a full list has fewer than two free entries.
// ---------------------------------------------------------------------------------------------
struct ProjPointsDev { const float *px, *py, *pxr; const int32_t *level; const float *viewCos; const uint8_t *inView, *hasObs, *desc; const int32_t *counts; int cap; };

// gate + key of feature idx for one map point; KEY64_EMPTY when the feature is not a candidate
struct ProjQuery { float x, y, rr, xr; int minLevel, maxLevel, cx0, cx1, cy0, cy1; bool any; unsigned long long d[4]; };

__device__ __forceinline__ ProjQuery proj_query_vals(const ProjFrameDev &F, float px, float py, float pxr, int lvl, float viewCos, const uint8_t *desc32, const float *scaleFactors, float th)
{
    ProjQuery q;
    float r = (double)viewCos > 0.998 ? 2.5f : 4.0f;   // RadiusByViewingCos, :178-185
    if (th != 1.0f) r *= th;
    q.x = px; q.y = py; q.xr = pxr;
    q.rr = r * scaleFactors[lvl];
    q.minLevel = lvl - 1; q.maxLevel = lvl;
    const int nMinCellX = max(0, (int)floorf((q.x - F.minX - q.rr) * F.gwInv)), nMaxCellX = min(GRID_COLS - 1, (int)ceilf((q.x - F.minX + q.rr) * F.gwInv));
    const int nMinCellY = max(0, (int)floorf((q.y - F.minY - q.rr) * F.ghInv)), nMaxCellY = min(GRID_ROWS - 1, (int)ceilf((q.y - F.minY + q.rr) * F.ghInv));
    q.any = !(nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0);
    q.cx0 = nMinCellX; q.cx1 = nMaxCellX; q.cy0 = nMinCellY; q.cy1 = nMaxCellY;
    if (desc32) {      // (nullptr: the caller holds the descriptor already)
        const unsigned long long *dp = (const unsigned long long *)desc32;
        q.d[0] = dp[0]; q.d[1] = dp[1]; q.d[2] = dp[2]; q.d[3] = dp[3];
    } else q.d[0] = q.d[1] = q.d[2] = q.d[3] = 0ull;
    return q;
}

__device__ __forceinline__ ProjQuery proj_query(const ProjFrameDev &F, const ProjPointsDev &P, size_t pi, const float *scaleFactors, float th)
{
    return proj_query_vals(F, P.px[pi], P.py[pi], P.pxr[pi], P.level[pi], P.viewCos[pi], P.desc + pi * 32, scaleFactors, th);
}

// the window / level / stereo gates of one (query, feature) pair - everything of proj_key that does not need the descriptor; -> passes, and the feature's grid cell
__device__ __forceinline__ bool proj_gate(const ProjFrameDev &F, const ProjQuery &q, float kx, float ky, int octave, float ur, int &cx, int &cy)
{
    cx = (int)roundf((kx - F.minX) * F.gwInv); cy = (int)roundf((ky - F.minY) * F.ghInv);   // PosInGrid, :866-867
    if (cx < q.cx0 || cx > q.cx1 || cy < q.cy0 || cy > q.cy1) return false;   // (also: feature outside the grid)
    if (octave < q.minLevel || (q.maxLevel >= 0 && octave > q.maxLevel)) return false;   // GetFeaturesInArea bCheckLevels, :814-822
    const float distx = kx - q.x, disty = ky - q.y;
    if (!(fabsf(distx) < q.rr && fabsf(disty) < q.rr)) return false;
    if (ur > 0) { const float er = fabsf(q.xr - ur); if (er > q.rr) return false; }
    return true;
}

__device__ __forceinline__ unsigned long long proj_key_of(const ProjFrameDev &F, size_t fbase, int idx, int cx, int cy, const ProjQuery &q)
{
    const unsigned long long *db = (const unsigned long long *)(F.desc + (fbase + idx) * 32);
    const int dist = hamming256(q.d, db[0], db[1], db[2], db[3]);
    return ((unsigned long long)dist << 32) | ((unsigned long long)cx << 22) | ((unsigned long long)cy << 16) | (unsigned long long)idx;
}

__device__ __forceinline__ unsigned long long proj_key(const ProjFrameDev &F, size_t fbase, int idx, const ProjQuery &q)
{
    const orbx_keypoint k = F.kp[fbase + idx];
    int cx, cy;
    if (!proj_gate(F, q, k.x, k.y, k.octave, F.uRight[fbase + idx], cx, cy)) return KEY64_EMPTY;
    return proj_key_of(F, fbase, idx, cx, cy, q);
}

// the TOPK smallest keys of one map point's query over the frame's features (one wave, lanes over the features) -> out[0..TOPK)
// Round 6.  Every wave of these kernels walks ALL features of its frame (a 1000-feature frame x 1900 map points = 1.9 M gate evaluations per call), so the
// walk is kept to what rejects almost everything: (1) only the window test |x - u| < r, |y - v| < r (src/Frame.cc:832-840; the reference applies it last,
// the result is the conjunction either way) on the coordinates of PT_BATCH features per lane, requested together (clamped, unconditional loads); (2) the
// features inside the window - a handful - are queued in the wave's corner of LDS by ballot; (3) the queue is worked off one candidate per lane: the
// remaining gates (grid cell window, level, stereo), then ONE round of descriptor loads; (4) with at most 64 candidates in all - every lane holds one key or
// none - a key's place in the list is the number of smaller keys (one scalar broadcast per candidate) instead of TOPK wave-wide minima.  Keys are unique
// (the feature index is part of the key), so which lane holds which key is immaterial: the lists are those of the plain loop.
#define PT_BATCH 8
#define PT_QUEUE 256      /* candidates queued per wave before they are worked off */
__device__ __forceinline__ void proj_insert(unsigned long long (&kk)[TOPK], unsigned long long key)
{
    if (key < kk[TOPK - 1]) {
        kk[TOPK - 1] = key;
#pragma unroll
        for (int t = TOPK - 1; t > 0; t--)
            if (kk[t] < kk[t - 1]) { const unsigned long long v = kk[t - 1]; kk[t - 1] = kk[t]; kk[t] = v; }
    }
}
__device__ __forceinline__ void proj_topk_wave(const ProjFrameDev &F, size_t fbase, int n, const ProjQuery &q, int lane, unsigned long long *out, uint32_t *queue /* LDS, PT_QUEUE words of this wave */)
{
    unsigned long long kk[TOPK];
#pragma unroll
    for (int t = 0; t < TOPK; t++) kk[t] = KEY64_EMPTY;
    int drained = 0, rounds = 0;      // wave-uniform: candidates worked off, in how many rounds
    if (q.any) {
        int nq = 0;      // wave-uniform
        auto drain = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (one wave: its LDS instructions execute in order; this keeps the compiler from moving the reads up)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int c0 = 0; c0 < nq; c0 += 64) {
                if (c0 + lane < nq) proj_insert(kk, proj_key(F, fbase, (int)queue[c0 + lane], q));
                rounds++;
            }
            drained += nq;
            nq = 0;
            __builtin_amdgcn_wave_barrier();      // (the queue is rewritten from its start)
        };
        for (int base = 0; base < n; base += 64 * PT_BATCH) {
            float kx[PT_BATCH], ky[PT_BATCH];
#pragma unroll
            for (int u = 0; u < PT_BATCH; u++) {
                const int c = min(base + lane + 64 * u, n - 1);
                const orbx_keypoint *kp = F.kp + fbase + c;
                kx[u] = kp->x; ky[u] = kp->y;
            }
#pragma unroll
            for (int u = 0; u < PT_BATCH; u++) {
                const int idx = base + lane + 64 * u;
                const bool pass = idx < n && fabsf(kx[u] - q.x) < q.rr && fabsf(ky[u] - q.y) < q.rr;
                const unsigned long long mask = __ballot(pass);
                if (mask) {
                    const int cnt = __popcll(mask);
                    if (nq + cnt > PT_QUEUE) drain();
                    if (pass) queue[nq + __popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)idx;
                    nq += cnt;
                }
            }
        }
        if (nq) drain();
    }
    if (rounds <= 1) {
        // every lane holds at most one key (kk[0]): its place = the number of smaller keys among the `drained` lanes that were given a candidate
        const unsigned long long key = kk[0];
        int rank = 0;
        for (int t = 0; t < drained; t++) {
            const unsigned long long v = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(key >> 32), t) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, t);
            rank += v < key ? 1 : 0;
        }
        const int nv = __popcll(__ballot(key != KEY64_EMPTY));
        if (key != KEY64_EMPTY && rank < TOPK) out[rank] = key;
        if (lane < TOPK && lane >= nv) out[lane] = KEY64_EMPTY;
        return;
    }
#pragma unroll
    for (int k = 0; k < TOPK; k++) {
        const unsigned long long mn = wave_min_u64(kk[0]);
        if (kk[0] == mn && mn != KEY64_EMPTY) {   // keys are unique: exactly one lane pops its head
#pragma unroll
            for (int t = 0; t < TOPK - 1; t++) kk[t] = kk[t + 1];
            kk[TOPK - 1] = KEY64_EMPTY;
        }
        if (lane == 0) out[k] = mn;
    }
}

__global__ __launch_bounds__(256) void k_proj_topk(ProjFrameDev F, ProjPointsDev P, const float *__restrict__ scaleFactors, float th,
                                                   unsigned long long *__restrict__ topk)
{
    __shared__ uint32_t sQueue[4][PT_QUEUE];
    const int f = blockIdx.y, lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = min(F.counts[f], F.cap), m = min(P.counts[f], P.cap);
    if (i >= m) return;
    const size_t pi = (size_t)f * P.cap + i, fbase = (size_t)f * F.cap;
    unsigned long long *out = topk + pi * TOPK;
    if (!P.inView[pi]) { if (lane < TOPK) out[lane] = KEY64_EMPTY; return; }
    const ProjQuery q = proj_query(F, P, pi, scaleFactors, th);
    proj_topk_wave(F, fbase, n, q, lane, out, sQueue[threadIdx.x >> 6]);
}

// Replay of the reference's sequential pass over the map points (src/ORBmatcher.cc:70-175).  The pass is
// sequential only through F.mvpMapPoints: a feature that already holds a MapPoint WITH observations is
// skipped (:110-112), and an accepted point writes itself into its best feature (:169).  As in
// k_bow_greedy the result is the unique fixed point of
//     dec[r] = decide(candidates of point r that no accepted, observation-carrying point r' < r has taken)
// and the kernel iterates that map for all points in parallel: owner[b] = lowest such r' currently
// choosing feature b (LDS atomicMin), every point re-decides with "b is free iff owner[b] >= r", until
// a round changes nothing.  A point whose FULL candidate list has fewer than two free entries cannot be
// decided from the list: those points are queued and one wave each rescans all features exactly.
// An accepted point without observations does not block its feature (:110-112), later points may
// overwrite it: the feature ends up with the LAST point that chose it (atomicMax at the end).
#define PROJ_GREEDY_THREADS 1024
#define PROJ_NONE 0xffffffffu
__global__ __launch_bounds__(PROJ_GREEDY_THREADS) void k_proj_greedy(ProjFrameDev F, ProjPointsDev P, const float *__restrict__ scaleFactors, float th, float nnratio,
                                                                     const unsigned long long *__restrict__ topk, int32_t *__restrict__ assigned,
                                                                     int32_t *__restrict__ nmatches, int stride, uint32_t *__restrict__ decBuf,
                                                                     uint32_t *__restrict__ queueBuf, int32_t *__restrict__ pubAssigned, unsigned long long *pubFlag,
                                                                     unsigned long long pubSeq, int decLds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int sChangedP[2], sQueued, sTotal;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = min(F.counts[f], F.cap), m = min(P.counts[f], P.cap);
    uint32_t *owner = (uint32_t *)smem;                          // [cap]
    unsigned char *occ = smem + (size_t)F.cap * 4;               // [cap] 1 = the feature holds a MapPoint with observations on entry
    unsigned char *oct = occ + F.cap;                            // [cap] octave of the feature
    const size_t fbase = (size_t)f * F.cap, pbase = (size_t)f * P.cap;
    // chosen feature of every point / rescan queue: global (P.cap is not bounded by LDS) - except in the single-frame host call, whose ONE workgroup's latency
    // is the call's: decLds != 0 = the host found room for dec[m] and the points' observation flags in LDS (at that byte offset), and a round of the fixed
    // point touches no global memory at all (the lists of a thread's first two points are in its registers)
    uint32_t *dec = decLds ? (uint32_t *)(smem + decLds) : decBuf + pbase, *queue = queueBuf + pbase;
    unsigned char *sObs = decLds ? (unsigned char *)(dec + P.cap) : nullptr;
    int32_t *aout = assigned + (size_t)f * stride;
    for (int i = tid; i < stride; i += PROJ_GREEDY_THREADS) aout[i] = -1;
    for (int i = tid; i < n; i += PROJ_GREEDY_THREADS) { occ[i] = F.occupied ? F.occupied[fbase + i] : 0; oct[i] = (unsigned char)F.kp[fbase + i].octave; }
    for (int r = tid; r < m; r += PROJ_GREEDY_THREADS) { dec[r] = PROJ_NONE; if (sObs) sObs[r] = P.hasObs ? P.hasObs[pbase + r] : 1; }
    if (tid == 0) sTotal = 0;
    const unsigned long long *tk = topk + pbase * TOPK;
    // acceptance of the best / second best free candidate (:150-168)
    auto decide = [&](unsigned long long k1, unsigned long long k2) -> uint32_t {
        if (k1 == KEY64_EMPTY) return PROJ_NONE;
        const int bestDist = (int)(k1 >> 32), bestIdx = (int)(k1 & 0xffff);
        const int bestDist2 = k2 == KEY64_EMPTY ? 256 : (int)(k2 >> 32);
        const int bestLevel = oct[bestIdx], bestLevel2 = k2 == KEY64_EMPTY ? -1 : (int)oct[(int)(k2 & 0xffff)];
        if (bestDist > TH_HIGH) return PROJ_NONE;
        if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) return PROJ_NONE;
        return (uint32_t)bestIdx;
    };
    ulonglong2 ka0 = make_ulonglong2(0ull, 0ull), ka1 = ka0, ka2 = ka0, ka3 = ka0, kb0 = ka0, kb1 = ka0, kb2 = ka0, kb3 = ka0;
    int have = 0, inv = 0;
    // A round = claims, decisions, rarely rescans: three barriers (a fourth behind rescans); the "changed" flag alternates between two words, owner[] and the
    // queue counter are reset for the next round behind the barrier that ends this round's reads of them (as in k_bow_greedy)
    __syncthreads();
    for (int j = tid; j < n; j += PROJ_GREEDY_THREADS) owner[j] = occ[j] ? 0u : 0xffffffffu;      // owner: 0 = holds a MapPoint with observations on entry, else lowest claiming rank + 1
    if (tid == 0) { sChangedP[0] = 0; sChangedP[1] = 0; sQueued = 0; }
    for (int par = 0;; par ^= 1) {
        __syncthreads();
        for (int r = tid; r < m; r += PROJ_GREEDY_THREADS) {
            const uint32_t d = dec[r];
            if (d != PROJ_NONE && (sObs ? sObs[r] != 0 : (!P.hasObs || P.hasObs[pbase + r]))) atomicMin(&owner[d], (uint32_t)r + 1u);
        }
        __syncthreads();
        bool changed = false;
        for (int r = tid; r < m; r += PROJ_GREEDY_THREADS) {
            const int slot = (r - tid) / PROJ_GREEDY_THREADS;      // the thread's first two points keep their lists (and in-view flags) in registers across the rounds
            if (slot < 2 && !(have & (1 << slot))) {
                const bool iv = P.inView[pbase + r] != 0;
                inv = iv ? (inv | (1 << slot)) : inv;
                if (iv) {
                    const ulonglong2 *lp = (const ulonglong2 *)(tk + (size_t)r * TOPK);
                    if (slot == 0) { ka0 = lp[0]; ka1 = lp[1]; ka2 = lp[2]; ka3 = lp[3]; } else { kb0 = lp[0]; kb1 = lp[1]; kb2 = lp[2]; kb3 = lp[3]; }
                }
                have |= 1 << slot;
            }
            ulonglong2 q0, q1, q2, q3;
            if (slot == 0) { if (!(inv & 1)) continue; q0 = ka0; q1 = ka1; q2 = ka2; q3 = ka3; }
            else if (slot == 1) { if (!(inv & 2)) continue; q0 = kb0; q1 = kb1; q2 = kb2; q3 = kb3; }
            else {
                if (!P.inView[pbase + r]) continue;
                const ulonglong2 *lp = (const ulonglong2 *)(tk + (size_t)r * TOPK);
                q0 = lp[0]; q1 = lp[1]; q2 = lp[2]; q3 = lp[3];
            }
            const unsigned long long keys[TOPK] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
            // the first two free candidates in list order without a branch per candidate: the eight owner words requested together (slot 0 for an empty
            // entry), then the list walked backwards with selects - (k1, k2) end as the first two free (free for r: owner >= r + 1, see the initialisation)
            auto half = [&](const int k0, unsigned long long &h1, unsigned long long &h2) -> int {
                uint32_t ow[4];
#pragma unroll
                for (int k = 0; k < 4; k++) ow[k] = owner[keys[k0 + k] == KEY64_EMPTY ? 0u : (uint32_t)(keys[k0 + k] & 0xffffu)];
                int nf = 0;
                h1 = KEY64_EMPTY; h2 = KEY64_EMPTY;
#pragma unroll
                for (int k = 3; k >= 0; k--) {
                    const bool fr = keys[k0 + k] != KEY64_EMPTY && ow[k] > (uint32_t)r;
                    h2 = fr ? h1 : h2;
                    h1 = fr ? keys[k0 + k] : h1;
                    nf += fr ? 1 : 0;
                }
                return nf;
            };
            unsigned long long k1, k2;
            int nfree = half(0, k1, k2);
            if (nfree < 2 && (keys[4] & keys[5] & keys[6] & keys[7]) != KEY64_EMPTY) {      // the second half of the list only for a point that needs it
                unsigned long long j1, j2;
                const int nb = half(4, j1, j2);
                k2 = nfree == 1 ? j1 : j2;
                k1 = nfree == 1 ? k1 : j1;
                nfree += nb;
            }
            if (nfree < 2) k2 = KEY64_EMPTY;
            if (nfree < 2 && keys[TOPK - 1] != KEY64_EMPTY) { queue[atomicAdd(&sQueued, 1)] = (uint32_t)r; continue; }   // full list, exhausted
            const uint32_t nd = decide(k1, k2);
            if (nd != dec[r]) { dec[r] = nd; changed = true; }
        }
        if (changed) sChangedP[par] = 1;
        __syncthreads();
        // exact rescans: one wave per queued point, the two smallest keys among the features that are free for it
        const int nq = sQueued;
        for (int qi = wv; qi < nq; qi += PROJ_GREEDY_THREADS / 64) {
            const int r = (int)queue[qi];
            const ProjQuery q = proj_query(F, P, pbase + r, scaleFactors, th);
            unsigned long long a = KEY64_EMPTY, b = KEY64_EMPTY;
            if (q.any)
                for (int x = lane; x < n; x += 64) {
                    if (owner[x] <= (uint32_t)r) continue;
                    const unsigned long long kx = proj_key(F, fbase, x, q);
                    if (kx < a) { b = a; a = kx; } else if (kx < b) b = kx;
                }
            const unsigned long long k1 = wave_min_u64(a);
            if (a == k1) a = b;
            const unsigned long long k2 = wave_min_u64(a);
            const uint32_t nd = decide(k1, k2);
            if (lane == 0 && nd != dec[r]) { dec[r] = nd; sChangedP[par] = 1; }
        }
        if (nq) __syncthreads();      // (uniform)
        const int again = sChangedP[par];
        if (!again) break;
        for (int j = tid; j < n; j += PROJ_GREEDY_THREADS) owner[j] = occ[j] ? 0u : 0xffffffffu;
        if (tid == 0) { sChangedP[par ^ 1] = 0; sQueued = 0; }
    }
    // F.mvpMapPoints[bestIdx] = pMP in rank order: the last point that chose a feature keeps it
    __syncthreads();
    for (int j = tid; j < n; j += PROJ_GREEDY_THREADS) owner[j] = 0u;
    __syncthreads();
    int total = 0;
    for (int r = tid; r < m; r += PROJ_GREEDY_THREADS) {
        const uint32_t d = dec[r];
        if (d != PROJ_NONE) { atomicMax(&owner[d], (uint32_t)r + 1u); total++; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    if (lane == 0 && total) atomicAdd(&sTotal, total);
    __syncthreads();
    for (int j = tid; j < n; j += PROJ_GREEDY_THREADS) if (owner[j]) aout[j] = (int32_t)owner[j] - 1;
    if (tid == 0) nmatches[f] = sTotal;
    if (pubFlag) {
        // the single-frame host call: assigned[0..n) and the count straight into the caller's mapped pinned buffer, the sequence word behind them
        for (int j = tid; j < n; j += PROJ_GREEDY_THREADS) pubAssigned[j] = owner[j] ? (int32_t)owner[j] - 1 : -1;
        if (tid == 0) pubAssigned[n] = sTotal;
        orbx_publish(nullptr, pubFlag, pubSeq, 1u);
    }
}

// ---------------------------------------------------------------------------------------------
// ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)
// (src/ORBmatcher.cc:1569-1728), the matcher of Tracking::TrackWithMotionModel: every MapPoint of
// the last frame is projected with the current pose (float arithmetic in the reference's order),
// searched in a window whose pyramid-level range depends on forward / backward motion, and given
// to the free feature of minimum distance; rotation histogram pruning at the end.
// ---------------------------------------------------------------------------------------------
struct ProjLastDev { const uint8_t *valid; const float *pos; const uint8_t *desc, *hasObs; const int32_t *octave; const float *angle; const int32_t *counts;
                     int cap; const float *tcwCur, *tcwLast; float fx, fy, cx, cy, mbf, mb, maxX, maxY; };

__device__ __forceinline__ void proj_motion(const float *Rc, const float *Rl, float mb, int bMono, bool &bForward, bool &bBackward)
{
    float twc[3], tlcz;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        float s = (-Rc[0 * 4 + r]) * Rc[0 * 4 + 3];
        s = s + (-Rc[1 * 4 + r]) * Rc[1 * 4 + 3];
        s = s + (-Rc[2 * 4 + r]) * Rc[2 * 4 + 3];
        twc[r] = s;
    }
    {
        float s = Rl[2 * 4 + 0] * twc[0];
        s = s + Rl[2 * 4 + 1] * twc[1];
        s = s + Rl[2 * 4 + 2] * twc[2];
        tlcz = s + Rl[2 * 4 + 3];
    }
    bForward = tlcz > mb && !bMono;
    bBackward = -tlcz > mb && !bMono;
}

// query of last-frame feature li; false when the point is skipped before the window search (:1604-1633)
__device__ __forceinline__ bool proj_last_query(const ProjFrameDev &F, const ProjLastDev &L, int f, size_t li, const float *scaleFactors, float th,
                                                bool bForward, bool bBackward, ProjQuery &q)
{
    if (L.valid[li] != 1) return false;
    const float *Rc = L.tcwCur + 16 * (size_t)f, *X = L.pos + 3 * li;
    float x3Dc[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        float s = Rc[r * 4 + 0] * X[0];
        s = s + Rc[r * 4 + 1] * X[1];
        s = s + Rc[r * 4 + 2] * X[2];
        x3Dc[r] = s + Rc[r * 4 + 3];
    }
    const float invzc = (float)(1.0 / (double)x3Dc[2]);
    if (invzc < 0) return false;
    const float u = L.fx * x3Dc[0] * invzc + L.cx, v = L.fy * x3Dc[1] * invzc + L.cy;
    if (u < F.minX || u > L.maxX) return false;
    if (v < F.minY || v > L.maxY) return false;
    const int oct = L.octave[li];
    const float radius = th * scaleFactors[oct];
    q.x = u; q.y = v; q.rr = radius; q.xr = u - L.mbf * invzc;
    if (bForward) { q.minLevel = oct; q.maxLevel = -1; }
    else if (bBackward) { q.minLevel = 0; q.maxLevel = oct; }
    else { q.minLevel = oct - 1; q.maxLevel = oct + 1; }
    const int nMinCellX = max(0, (int)floorf((u - F.minX - radius) * F.gwInv)), nMaxCellX = min(GRID_COLS - 1, (int)ceilf((u - F.minX + radius) * F.gwInv));
    const int nMinCellY = max(0, (int)floorf((v - F.minY - radius) * F.ghInv)), nMaxCellY = min(GRID_ROWS - 1, (int)ceilf((v - F.minY + radius) * F.ghInv));
    q.any = !(nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0);
    q.cx0 = nMinCellX; q.cx1 = nMaxCellX; q.cy0 = nMinCellY; q.cy1 = nMaxCellY;
    const unsigned long long *dp = (const unsigned long long *)(L.desc + li * 32);
    q.d[0] = dp[0]; q.d[1] = dp[1]; q.d[2] = dp[2]; q.d[3] = dp[3];
    return q.any;
}

__global__ __launch_bounds__(256) void k_proj_last_topk(ProjFrameDev F, ProjLastDev L, const float *__restrict__ scaleFactors, float th, int bMono,
                                                        unsigned long long *__restrict__ topk)
{
    __shared__ uint32_t sQueue[4][PT_QUEUE];
    const int f = blockIdx.y, lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = min(F.counts[f], F.cap), nl = min(L.counts[f], L.cap);
    if (i >= nl) return;
    const size_t li = (size_t)f * L.cap + i, fbase = (size_t)f * F.cap;
    unsigned long long *out = topk + li * TOPK;
    bool bForward, bBackward;
    proj_motion(L.tcwCur + 16 * (size_t)f, L.tcwLast + 16 * (size_t)f, L.mb, bMono, bForward, bBackward);
    ProjQuery q;
    if (!proj_last_query(F, L, f, li, scaleFactors, th, bForward, bBackward, q)) { if (lane < TOPK) out[lane] = KEY64_EMPTY; return; }
    proj_topk_wave(F, fbase, n, q, lane, out, sQueue[threadIdx.x >> 6]);
}

// The same parallel fixed point as k_proj_greedy (the pass over the last frame's points is sequential only
// through CurrentFrame.mvpMapPoints, :1660-1662 / :1689); a point takes its best free candidate alone, there
// is no second-best test.
__global__ __launch_bounds__(PROJ_GREEDY_THREADS) void k_proj_last_greedy(ProjFrameDev F, ProjLastDev L, const float *__restrict__ scaleFactors, float th, int bMono,
                                                                          int checkOri, const unsigned long long *__restrict__ topk, int32_t *__restrict__ assigned,
                                                                          int32_t *__restrict__ nmatches, int stride, uint32_t *__restrict__ decBuf,
                                                                          uint32_t *__restrict__ queueBuf, int32_t *__restrict__ pubAssigned, unsigned long long *pubFlag,
                                                                          unsigned long long pubSeq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int sTotal, sRemoved, sChangedP[2], sQueued;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = min(F.counts[f], F.cap), nl = min(L.counts[f], L.cap);
    uint32_t *owner = (uint32_t *)smem;                     // [cap] lowest observation-carrying point choosing the feature; later: its final holder
    unsigned char *occ = smem + (size_t)F.cap * 4;          // [cap]
    const size_t fbase = (size_t)f * F.cap, lbase = (size_t)f * L.cap;
    uint32_t *dec = decBuf + lbase, *queue = queueBuf + lbase;
    int32_t *aout = assigned + (size_t)f * stride;
    for (int i = tid; i < F.cap; i += PROJ_GREEDY_THREADS) occ[i] = (i < n && F.occupied) ? F.occupied[fbase + i] : 0;
    for (int r = tid; r < nl; r += PROJ_GREEDY_THREADS) dec[r] = PROJ_NONE;
    if (tid < HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) { sTotal = 0; sRemoved = 0; }
    bool bForward, bBackward;
    proj_motion(L.tcwCur + 16 * (size_t)f, L.tcwLast + 16 * (size_t)f, L.mb, bMono, bForward, bBackward);
    const unsigned long long *tk = topk + lbase * TOPK;
    // Rounds as in k_proj_greedy: owner = 0 for a feature that holds a MapPoint with observations on entry, else (lowest claiming point + 1); the lists of a
    // thread's first two points stay in its registers; the first free candidate by four owner words requested together and a backward walk with selects
    // (the second half of the list only when the first holds none); three barriers per round, a fourth behind rescans.
    ulonglong2 ka0 = make_ulonglong2(KEY64_EMPTY, KEY64_EMPTY), ka1 = ka0, ka2 = ka0, ka3 = ka0, kb0 = ka0, kb1 = ka0, kb2 = ka0, kb3 = ka0;
    int have = 0;
    __syncthreads();
    for (int j = tid; j < n; j += PROJ_GREEDY_THREADS) owner[j] = occ[j] ? 0u : 0xffffffffu;
    if (tid == 0) { sChangedP[0] = 0; sChangedP[1] = 0; sQueued = 0; }
    for (int par = 0;; par ^= 1) {
        __syncthreads();
        for (int r = tid; r < nl; r += PROJ_GREEDY_THREADS) {
            const uint32_t d = dec[r];
            if (d != PROJ_NONE && (!L.hasObs || L.hasObs[lbase + r])) atomicMin(&owner[d], (uint32_t)r + 1u);
        }
        __syncthreads();
        bool changed = false;
        for (int r = tid; r < nl; r += PROJ_GREEDY_THREADS) {
            const int slot = (r - tid) / PROJ_GREEDY_THREADS;
            ulonglong2 q0, q1, q2, q3;
            if (slot < 2) {
                if (!(have & (1 << slot))) {
                    const ulonglong2 *lp = (const ulonglong2 *)(tk + (size_t)r * TOPK);
                    if (slot == 0) { ka0 = lp[0]; ka1 = lp[1]; ka2 = lp[2]; ka3 = lp[3]; } else { kb0 = lp[0]; kb1 = lp[1]; kb2 = lp[2]; kb3 = lp[3]; }
                    have |= 1 << slot;
                }
                if (slot == 0) { q0 = ka0; q1 = ka1; q2 = ka2; q3 = ka3; } else { q0 = kb0; q1 = kb1; q2 = kb2; q3 = kb3; }
            } else {
                const ulonglong2 *lp = (const ulonglong2 *)(tk + (size_t)r * TOPK);
                q0 = lp[0]; q1 = lp[1]; q2 = lp[2]; q3 = lp[3];
            }
            const unsigned long long keys[TOPK] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
            if (keys[0] == KEY64_EMPTY) continue;                       // skipped point or empty window
            auto half = [&](const int k0) -> unsigned long long {
                uint32_t ow[4];
#pragma unroll
                for (int k = 0; k < 4; k++) ow[k] = owner[keys[k0 + k] == KEY64_EMPTY ? 0u : (uint32_t)(keys[k0 + k] & 0xffffu)];
                unsigned long long h1 = KEY64_EMPTY;
#pragma unroll
                for (int k = 3; k >= 0; k--) h1 = (keys[k0 + k] != KEY64_EMPTY && ow[k] > (uint32_t)r) ? keys[k0 + k] : h1;      // ends on the first free entry
                return h1;
            };
            unsigned long long k1 = half(0);
            if (k1 == KEY64_EMPTY && (keys[4] & keys[5] & keys[6] & keys[7]) != KEY64_EMPTY) k1 = half(4);
            if (k1 == KEY64_EMPTY && keys[TOPK - 1] != KEY64_EMPTY) { queue[atomicAdd(&sQueued, 1)] = (uint32_t)r; continue; }   // full list, all taken
            const uint32_t nd = (k1 != KEY64_EMPTY && (int)(k1 >> 32) <= TH_HIGH) ? (uint32_t)(k1 & 0xffff) : PROJ_NONE;
            if (nd != dec[r]) { dec[r] = nd; changed = true; }
        }
        if (changed) sChangedP[par] = 1;
        __syncthreads();
        const int nq = sQueued;
        for (int qi = wv; qi < nq; qi += PROJ_GREEDY_THREADS / 64) {
            const int r = (int)queue[qi];
            ProjQuery q;
            unsigned long long a = KEY64_EMPTY;
            if (proj_last_query(F, L, f, lbase + r, scaleFactors, th, bForward, bBackward, q))
                for (int x = lane; x < n; x += 64) {
                    if (owner[x] <= (uint32_t)r) continue;
                    const unsigned long long kx = proj_key(F, fbase, x, q);
                    a = kx < a ? kx : a;
                }
            const unsigned long long k1 = wave_min_u64(a);
            const uint32_t nd = (k1 != KEY64_EMPTY && (int)(k1 >> 32) <= TH_HIGH) ? (uint32_t)(k1 & 0xffff) : PROJ_NONE;
            if (lane == 0 && nd != dec[r]) { dec[r] = nd; sChangedP[par] = 1; }
        }
        if (nq) __syncthreads();      // (uniform)
        const int again = sChangedP[par];
        if (!again) break;
        for (int j = tid; j < n; j += PROJ_GREEDY_THREADS) owner[j] = occ[j] ? 0u : 0xffffffffu;
        if (tid == 0) { sChangedP[par ^ 1] = 0; sQueued = 0; }
    }
    // CurrentFrame.mvpMapPoints[bestIdx2] = pMP in point order: the last chooser holds the feature (owner := holder + 1, 0 = none)
    __syncthreads();
    for (int j = tid; j < n; j += PROJ_GREEDY_THREADS) owner[j] = 0u;
    __syncthreads();
    // rotation histogram over the ACCEPTED assignments (:1694-1704): a feature whose first holder had no
    // observations can be re-assigned later and then sits in two bins; pruning a bin clears the feature
    // whichever holder put it there (:1712-1722).
    const float factor = HISTO_LENGTH / 360.0f;
    int total = 0;
    for (int r = tid; r < nl; r += PROJ_GREEDY_THREADS) {
        const uint32_t d = dec[r];
        if (d == PROJ_NONE) continue;
        atomicMax(&owner[d], (uint32_t)r + 1u);
        total++;
        if (checkOri) {
            float rot = L.angle[lbase + r] - F.kp[fbase + d].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            dec[r] = d | ((uint32_t)bin << 16);   // keep feature + bin (F.cap <= 65535)
            atomicAdd(&hist[bin], 1);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    if (lane == 0 && total) atomicAdd(&sTotal, total);
    __syncthreads();
    for (int j = tid; j < stride; j += PROJ_GREEDY_THREADS) aout[j] = (j < n && owner[j]) ? (int32_t)owner[j] - 1 : -1;
    __syncthreads();
    if (checkOri) {
        int ind1, ind2, ind3;
        three_maxima_wave(hist, lane, ind1, ind2, ind3);
        int removed = 0;
        for (int r = tid; r < nl; r += PROJ_GREEDY_THREADS) {      // same point -> thread mapping as the pass above
            const uint32_t d = dec[r];
            if (d == PROJ_NONE) continue;
            const int b = (int)(d >> 16);
            if (b != ind1 && b != ind2 && b != ind3) { aout[d & 0xffffu] = -2; removed++; }   // -2: assigned, then cleared (:1718)
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o);
        if (lane == 0 && removed) atomicAdd(&sRemoved, removed);
    }
    __syncthreads();
    if (tid == 0) nmatches[f] = sTotal - sRemoved;
    if (pubFlag) {
        // the single-frame host call: assigned[0..n) and the count straight into the caller's mapped pinned buffer, the sequence word behind them
        for (int j = tid; j < n; j += PROJ_GREEDY_THREADS) pubAssigned[j] = aout[j];
        if (tid == 0) pubAssigned[n] = sTotal - sRemoved;
        orbx_publish(nullptr, pubFlag, pubSeq, 1u);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// ORBmatcher::Fuse, both overloads (src/ORBmatcher.cc:1020-1177, 1179-1312): the search of steps 2-3,
// i.e. KeyFrame::GetFeaturesInArea(u, v, radius) (src/KeyFrame.cc:752-796), the level gate
// (:1107-1108, 1262-1263), for the first overload the chi-square gate on the reprojection error
// (:1111-1135) and the minimum Hamming distance with strict '<' (first minimum in the cell-major
// order of GetFeaturesInArea).  The map points are independent of each other here (the reference's
// loop only couples them through Replace / AddObservation, which stay on the host in the shim).
// One wave per map point, lanes over the KeyFrame's features.
// ---------------------------------------------------------------------------------------------
struct FusePointsDev { const float *u, *v, *ur; const int32_t *level; const float *radius; const uint8_t *active, *desc; const int32_t *counts; int cap;
                       float kfMinX, kfMinY; };
struct FuseLevels { float invSigma2[ORBX_MAX_LEVELS]; };

__global__ __launch_bounds__(256) void k_fuse_best(ProjFrameDev F, FusePointsDev P, FuseLevels LV, int chi2Gate, int32_t *__restrict__ bestIdx,
                                                   int32_t *__restrict__ bestDist, int stride)
{
    const int f = blockIdx.y, lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = min(F.counts[f], F.cap), m = min(P.counts[f], P.cap);
    if (i >= m) return;
    const size_t pi = (size_t)f * P.cap + i, fbase = (size_t)f * F.cap;
    unsigned long long best = KEY64_EMPTY;
    if (!P.active || P.active[pi]) {
        const float x = P.u[pi], y = P.v[pi], r = P.radius[pi], xr = chi2Gate ? P.ur[pi] : 0.0f;
        const int lvl = P.level[pi];
        // the window uses the KeyFrame's (int) bounds, KeyFrame.cc:760-775; the cells were filed with the Frame's float bounds
        const int cx0 = max(0, (int)floorf((x - P.kfMinX - r) * F.gwInv)), cx1 = min(GRID_COLS - 1, (int)ceilf((x - P.kfMinX + r) * F.gwInv));
        const int cy0 = max(0, (int)floorf((y - P.kfMinY - r) * F.ghInv)), cy1 = min(GRID_ROWS - 1, (int)ceilf((y - P.kfMinY + r) * F.ghInv));
        if (!(cx0 >= GRID_COLS || cx1 < 0 || cy0 >= GRID_ROWS || cy1 < 0)) {
            const unsigned long long *dp = (const unsigned long long *)(P.desc + pi * 32);
            const unsigned long long d[4] = {dp[0], dp[1], dp[2], dp[3]};
            for (int idx = lane; idx < n; idx += 64) {
                const orbx_keypoint k = F.kp[fbase + idx];
                const int cx = (int)roundf((k.x - F.minX) * F.gwInv), cy = (int)roundf((k.y - F.minY) * F.ghInv);   // the cell AssignFeaturesToGrid filed it in
                if (cx < cx0 || cx > cx1 || cy < cy0 || cy > cy1) continue;
                const float distx = k.x - x, disty = k.y - y;
                if (!(fabsf(distx) < r && fabsf(disty) < r)) continue;                                             // KeyFrame.cc:786-790
                if (k.octave < lvl - 1 || k.octave > lvl) continue;                                               // :1107-1108
                if (chi2Gate) {
                    const float kr = F.uRight[fbase + idx];
                    const float ex = x - k.x, ey = y - k.y;
                    if (kr >= 0) {
                        const float er = xr - kr;
                        const float e2 = ex * ex + ey * ey + er * er;
                        if ((double)(e2 * LV.invSigma2[k.octave]) > 7.8) continue;                                // :1123
                    } else {
                        const float e2 = ex * ex + ey * ey;
                        if ((double)(e2 * LV.invSigma2[k.octave]) > 5.99) continue;                               // :1134
                    }
                }
                const unsigned long long *db = (const unsigned long long *)(F.desc + (fbase + idx) * 32);
                const int dist = hamming256(d, db[0], db[1], db[2], db[3]);
                const unsigned long long key = ((unsigned long long)dist << 32) | ((unsigned long long)cx << 22) | ((unsigned long long)cy << 16) | (unsigned long long)idx;
                best = key < best ? key : best;
            }
        }
    }
    best = wave_min_u64(best);
    if (lane == 0) {
        const size_t o = (size_t)f * stride + i;
        bestIdx[o] = best == KEY64_EMPTY ? -1 : (int)(best & 0xffff);
        bestDist[o] = best == KEY64_EMPTY ? 256 : (int)(best >> 32);
    }
}

static int fuse_launch(orbx_matcher *m, const ProjFrameDev &F, const FusePointsDev &P, int nframes, const float *inv_level_sigma2, int nlevels, int chi2_gate)
{
    if (nframes < 1 || nframes > m->maxPairs) { orbx_set_error("nframes %d outside 1..%d", nframes, m->maxPairs); return ORBX_ERR_CAPACITY; }
    if (F.cap < 1 || F.cap > 65535 || P.cap < 1 || P.cap > m->maxFeatures) {
        orbx_set_error("bad capacities (features %d, points %d, matcher max_features %d)", F.cap, P.cap, m->maxFeatures);
        return ORBX_ERR_CAPACITY;
    }
    if (!inv_level_sigma2 || nlevels < 1 || nlevels > ORBX_MAX_LEVELS) { orbx_set_error("bad level table"); return ORBX_ERR_ARG; }
    FuseLevels LV;
    for (int l = 0; l < ORBX_MAX_LEVELS; l++) LV.invSigma2[l] = l < nlevels ? inv_level_sigma2[l] : 0.0f;
    const int stride = m->maxFeatures;
    const int slot = m->profCount % MATCH_PROF_RING;
    ORBX_HIP_CHECK(hipEventRecord(m->ev0[slot], m->stream));
    m->midValid[slot] = false;
    hipLaunchKernelGGL(k_fuse_best, dim3((unsigned)((P.cap + 3) / 4), (unsigned)nframes), dim3(256), 0, m->stream, F, P, LV, chi2_gate, m->matches.p, m->dists.p, stride);
    MLAUNCH_CHECK();
    ORBX_HIP_CHECK(hipEventRecord(m->ev1[slot], m->stream));
    m->profCount++;
    m->lastPairs = nframes; m->lastStride = stride;
    return ORBX_OK;
}

extern "C" int orbx_fuse_search_device(orbx_matcher *m, const orbx_projection_frame *kf, const orbx_fuse_points *pts, const float *inv_level_sigma2, int nlevels,
                                       int chi2_gate)
{
    if (!m || !kf || !pts) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!kf->keypoints_un || !kf->descriptors || !kf->counts || (chi2_gate && (!kf->u_right || !pts->ur)) || !pts->u || !pts->v || !pts->level || !pts->radius ||
        !pts->descriptors || !pts->counts) {
        orbx_set_error("NULL array in the fuse arguments");
        return ORBX_ERR_ARG;
    }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    ProjFrameDev F = {kf->keypoints_un, kf->descriptors, kf->u_right, nullptr, kf->counts, kf->capacity, kf->min_x, kf->min_y, kf->grid_width_inv, kf->grid_height_inv};
    FusePointsDev P = {pts->u, pts->v, pts->ur, pts->level, pts->radius, pts->active, pts->descriptors, pts->counts, pts->capacity, pts->kf_min_x, pts->kf_min_y};
    return fuse_launch(m, F, P, kf->nframes, inv_level_sigma2, nlevels, chi2_gate);
}

// host-array form for one KeyFrame: upload, run, download
extern "C" int orbx_fuse_search(orbx_matcher *m, const orbx_projection_frame *kf, const orbx_fuse_points *pt, const float *inv_level_sigma2, int nlevels,
                                int chi2_gate, int32_t *best_idx, int32_t *best_dist)
{
    if (!m || !kf || !pt || !best_idx || !best_dist) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!kf->counts || !pt->counts) { orbx_set_error("NULL counts"); return ORBX_ERR_ARG; }
    const int n = kf->counts[0], mm = pt->counts[0];
    for (int i = 0; i < mm; i++) { best_idx[i] = -1; best_dist[i] = 256; }
    if (n <= 0 || mm <= 0) return ORBX_OK;
    if (mm > m->maxFeatures) { orbx_set_error("%d map points exceed the matcher's max_features %d", mm, m->maxFeatures); return ORBX_ERR_CAPACITY; }
    if (!kf->keypoints_un || !kf->descriptors || (chi2_gate && (!kf->u_right || !pt->ur)) || !pt->u || !pt->v || !pt->level || !pt->radius || !pt->descriptors) {
        orbx_set_error("NULL array in the fuse arguments");
        return ORBX_ERR_ARG;
    }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    int rc;
    // staging: KeyFrame side in pkp / hd[0] / pf[0]; point side in pf[1] (u, v, ur, radius) / pi32[1] (level) / pb[1] (descriptors, active)
    if ((rc = m->pkp.ensure((size_t)n)) || (rc = m->hd[0].ensure((size_t)n * 32)) || (rc = m->pf[0].ensure((size_t)n)) || (rc = m->pi32[0].ensure(2)) ||
        (rc = m->pf[1].ensure((size_t)mm * 4)) || (rc = m->pi32[1].ensure((size_t)mm)) || (rc = m->pb[1].ensure((size_t)mm * 33)))
        return rc;
    hipStream_t st = m->stream;
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pkp.p, kf->keypoints_un, (size_t)n * sizeof(orbx_keypoint), hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->hd[0].p, kf->descriptors, (size_t)n * 32, hipMemcpyHostToDevice, st));
    if (kf->u_right) ORBX_HIP_CHECK(hipMemcpyAsync(m->pf[0].p, kf->u_right, (size_t)n * 4, hipMemcpyHostToDevice, st));
    const int32_t cnt[2] = {n, mm};
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pi32[0].p, cnt, sizeof(cnt), hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pf[1].p, pt->u, (size_t)mm * 4, hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pf[1].p + mm, pt->v, (size_t)mm * 4, hipMemcpyHostToDevice, st));
    if (pt->ur) ORBX_HIP_CHECK(hipMemcpyAsync(m->pf[1].p + 2 * (size_t)mm, pt->ur, (size_t)mm * 4, hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pf[1].p + 3 * (size_t)mm, pt->radius, (size_t)mm * 4, hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pi32[1].p, pt->level, (size_t)mm * 4, hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pb[1].p, pt->descriptors, (size_t)mm * 32, hipMemcpyHostToDevice, st));
    if (pt->active) ORBX_HIP_CHECK(hipMemcpyAsync(m->pb[1].p + (size_t)mm * 32, pt->active, (size_t)mm, hipMemcpyHostToDevice, st));
    ProjFrameDev F = {m->pkp.p, m->hd[0].p, m->pf[0].p, nullptr, m->pi32[0].p, n, kf->min_x, kf->min_y, kf->grid_width_inv, kf->grid_height_inv};
    FusePointsDev P = {m->pf[1].p, m->pf[1].p + mm, m->pf[1].p + 2 * (size_t)mm, m->pi32[1].p, m->pf[1].p + 3 * (size_t)mm,
                       pt->active ? m->pb[1].p + (size_t)mm * 32 : nullptr, m->pb[1].p, m->pi32[0].p + 1, mm, pt->kf_min_x, pt->kf_min_y};
    if ((rc = fuse_launch(m, F, P, 1, inv_level_sigma2, nlevels, chi2_gate)) != ORBX_OK) return rc;
    ORBX_HIP_CHECK(hipStreamSynchronize(st));
    ORBX_HIP_CHECK(hipMemcpy(best_idx, m->matches.p, (size_t)mm * 4, hipMemcpyDeviceToHost));
    ORBX_HIP_CHECK(hipMemcpy(best_dist, m->dists.p, (size_t)mm * 4, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

// ---------------------------------------------------------------------------------------------
// Greedy area search: ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (loop closing,
// src/ORBmatcher.cc:388-513) and SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist)
// (relocalisation, :1731-1864) after their per-point preparation: the queries are processed in order, each
// takes the feature of minimum distance among those in its GetFeaturesInArea window that pass the level
// gate and are not blocked (blocked at the start, or taken by an earlier query), if that distance is
// <= max_dist.  Candidate lists (top-8 by distance || cell order || index, blocking ignored) come from
// k_area_topk; k_area_greedy replays the sequence as the unique fixed point of "query r takes its best
// candidate that no accepted query < r took" (see k_bow_greedy), with an exact rescan for exhausted lists.
// ---------------------------------------------------------------------------------------------
struct AreaQueriesDev { const float *u, *v, *radius; const int32_t *minLevel, *maxLevel; const uint8_t *active, *desc; const int32_t *counts; int cap;
                        float winMinX, winMinY; };
struct AreaQuery { float x, y, r; int minLevel, maxLevel, cx0, cx1, cy0, cy1; bool any; unsigned long long d[4]; };

__device__ __forceinline__ AreaQuery area_query(const ProjFrameDev &F, const AreaQueriesDev &Q, size_t qi)
{
    AreaQuery q;
    q.x = Q.u[qi]; q.y = Q.v[qi]; q.r = Q.radius[qi];
    q.minLevel = Q.minLevel[qi]; q.maxLevel = Q.maxLevel[qi];
    const int x0 = max(0, (int)floorf((q.x - Q.winMinX - q.r) * F.gwInv)), x1 = min(GRID_COLS - 1, (int)ceilf((q.x - Q.winMinX + q.r) * F.gwInv));
    const int y0 = max(0, (int)floorf((q.y - Q.winMinY - q.r) * F.ghInv)), y1 = min(GRID_ROWS - 1, (int)ceilf((q.y - Q.winMinY + q.r) * F.ghInv));
    q.any = !(x0 >= GRID_COLS || x1 < 0 || y0 >= GRID_ROWS || y1 < 0);
    q.cx0 = x0; q.cx1 = x1; q.cy0 = y0; q.cy1 = y1;
    const unsigned long long *dp = (const unsigned long long *)(Q.desc + qi * 32);
    q.d[0] = dp[0]; q.d[1] = dp[1]; q.d[2] = dp[2]; q.d[3] = dp[3];
    return q;
}

__device__ __forceinline__ unsigned long long area_key(const ProjFrameDev &F, size_t fbase, int idx, const AreaQuery &q)
{
    const orbx_keypoint k = F.kp[fbase + idx];
    const int cx = (int)roundf((k.x - F.minX) * F.gwInv), cy = (int)roundf((k.y - F.minY) * F.ghInv);   // the cell AssignFeaturesToGrid filed it in
    if (cx < q.cx0 || cx > q.cx1 || cy < q.cy0 || cy > q.cy1) return KEY64_EMPTY;
    if (k.octave < q.minLevel || (q.maxLevel >= 0 && k.octave > q.maxLevel)) return KEY64_EMPTY;
    const float distx = k.x - q.x, disty = k.y - q.y;
    if (!(fabsf(distx) < q.r && fabsf(disty) < q.r)) return KEY64_EMPTY;
    const unsigned long long *db = (const unsigned long long *)(F.desc + (fbase + idx) * 32);
    const int dist = hamming256(q.d, db[0], db[1], db[2], db[3]);
    return ((unsigned long long)dist << 32) | ((unsigned long long)cx << 22) | ((unsigned long long)cy << 16) | (unsigned long long)idx;
}

__global__ __launch_bounds__(256) void k_area_topk(ProjFrameDev F, AreaQueriesDev Q, unsigned long long *__restrict__ topk)
{
    const int f = blockIdx.y, lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = min(F.counts[f], F.cap), m = min(Q.counts[f], Q.cap);
    if (i >= m) return;
    const size_t qi = (size_t)f * Q.cap + i, fbase = (size_t)f * F.cap;
    unsigned long long *out = topk + qi * TOPK;
    if (Q.active && !Q.active[qi]) { if (lane < TOPK) out[lane] = KEY64_EMPTY; return; }
    const AreaQuery q = area_query(F, Q, qi);
    unsigned long long kk[TOPK];
#pragma unroll
    for (int t = 0; t < TOPK; t++) kk[t] = KEY64_EMPTY;
    if (q.any)
        for (int idx = lane; idx < n; idx += 64) {
            const unsigned long long key = area_key(F, fbase, idx, q);
            if (key < kk[TOPK - 1]) {
                kk[TOPK - 1] = key;
#pragma unroll
                for (int t = TOPK - 1; t > 0; t--)
                    if (kk[t] < kk[t - 1]) { const unsigned long long v = kk[t - 1]; kk[t - 1] = kk[t]; kk[t] = v; }
            }
        }
#pragma unroll
    for (int k = 0; k < TOPK; k++) {
        const unsigned long long mn = wave_min_u64(kk[0]);
        if (kk[0] == mn && mn != KEY64_EMPTY) {   // keys are unique: exactly one lane pops its head
#pragma unroll
            for (int t = 0; t < TOPK - 1; t++) kk[t] = kk[t + 1];
            kk[TOPK - 1] = KEY64_EMPTY;
        }
        if (lane == 0) out[k] = mn;
    }
}

__global__ __launch_bounds__(256) void k_area_greedy(ProjFrameDev F, AreaQueriesDev Q, int maxDist, const unsigned long long *__restrict__ topk,
                                                     int32_t *__restrict__ assigned, int32_t *__restrict__ dists, int32_t *__restrict__ nmatches, int stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int sChanged, sQueued, sTotal;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = min(F.counts[f], F.cap), m = min(Q.counts[f], Q.cap);
    // LDS: owner[F.cap] (0 = blocked from the start, else 1 + lowest accepted query taking the feature, ~0 = free) | dec[Q.cap] | queue (u16)
    uint32_t *owner = (uint32_t *)smem;
    uint32_t *dec = owner + F.cap;                        // KEY_EMPTY or dist << 16 | feature
    unsigned short *queue = (unsigned short *)(dec + Q.cap);
    const size_t fbase = (size_t)f * F.cap, qbase = (size_t)f * Q.cap;
    const unsigned long long *tk = topk + qbase * TOPK;
    int32_t *aout = assigned + (size_t)f * stride, *dout = dists + (size_t)f * stride;
    for (int i = tid; i < m; i += 256) dec[i] = KEY_EMPTY;
    if (tid == 0) sTotal = 0;
    for (;;) {
        for (int j = tid; j < n; j += 256) owner[j] = (F.occupied && F.occupied[fbase + j]) ? 0u : 0xffffffffu;
        if (tid == 0) { sChanged = 0; sQueued = 0; }
        __syncthreads();
        for (int r = tid; r < m; r += 256) {
            const uint32_t d = dec[r];
            if (d != KEY_EMPTY) atomicMin(&owner[d & 0xffff], (uint32_t)r + 1u);
        }
        __syncthreads();
        bool changed = false;
        for (int r = tid; r < m; r += 256) {
            if (Q.active && !Q.active[qbase + r]) continue;
            uint32_t nd = KEY_EMPTY;
            bool full = true, found = false;
#pragma unroll
            for (int k = 0; k < TOPK; k++) {
                const unsigned long long key = tk[(size_t)r * TOPK + k];
                if (key == KEY64_EMPTY) { full = false; continue; }
                const uint32_t idx = (uint32_t)(key & 0xffff);
                if (!found && owner[idx] > (uint32_t)r) {   // free for this query: not blocked and not taken by an earlier one
                    found = true;
                    const uint32_t dist = (uint32_t)(key >> 32);
                    if ((int)dist <= maxDist) nd = (dist << 16) | idx;
                }
            }
            if (!found && full) { queue[atomicAdd(&sQueued, 1)] = (unsigned short)r; continue; }
            if (nd != dec[r]) { dec[r] = nd; changed = true; }
        }
        if (changed) sChanged = 1;
        __syncthreads();
        const int nq = sQueued;
        for (int qq = wv; qq < nq; qq += 4) {   // exact rescan over the free features
            const int r = queue[qq];
            const AreaQuery q = area_query(F, Q, qbase + r);
            unsigned long long best = KEY64_EMPTY;
            for (int idx = lane; idx < n; idx += 64) {
                if (!(owner[idx] > (uint32_t)r)) continue;
                const unsigned long long key = area_key(F, fbase, idx, q);
                best = key < best ? key : best;
            }
            best = wave_min_u64(best);
            uint32_t nd = KEY_EMPTY;
            if (best != KEY64_EMPTY && (int)(best >> 32) <= maxDist) nd = ((uint32_t)(best >> 32) << 16) | (uint32_t)(best & 0xffff);
            if (lane == 0 && nd != dec[r]) { dec[r] = nd; sChanged = 1; }
        }
        __syncthreads();
        const int again = sChanged;
        __syncthreads();
        if (!again) break;
    }
    int total = 0;
    for (int r = tid; r < stride; r += 256) {
        const uint32_t d = r < m ? dec[r] : KEY_EMPTY;
        aout[r] = d == KEY_EMPTY ? -1 : (int)(d & 0xffff);
        dout[r] = d == KEY_EMPTY ? 256 : (int)(d >> 16);
        total += d != KEY_EMPTY;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    if (lane == 0 && total) atomicAdd(&sTotal, total);
    __syncthreads();
    if (tid == 0) nmatches[f] = sTotal;
}

static int area_launch(orbx_matcher *m, const ProjFrameDev &F, const AreaQueriesDev &Q, int nframes, int max_dist)
{
    if (nframes < 1 || nframes > m->maxPairs) { orbx_set_error("nframes %d outside 1..%d", nframes, m->maxPairs); return ORBX_ERR_CAPACITY; }
    if (F.cap < 1 || F.cap > 65535 || Q.cap < 1 || Q.cap > m->maxFeatures || Q.cap > 65535) {
        orbx_set_error("bad capacities (features %d, queries %d, matcher max_features %d)", F.cap, Q.cap, m->maxFeatures);
        return ORBX_ERR_CAPACITY;
    }
    int rc = m->topk64.ensure((size_t)nframes * Q.cap * TOPK);
    if (rc != ORBX_OK) return rc;
    const int stride = m->maxFeatures;
    const int slot = m->profCount % MATCH_PROF_RING;
    ORBX_HIP_CHECK(hipEventRecord(m->ev0[slot], m->stream));
    m->midValid[slot] = false;
    hipLaunchKernelGGL(k_area_topk, dim3((unsigned)((Q.cap + 3) / 4), (unsigned)nframes), dim3(256), 0, m->stream, F, Q, m->topk64.p);
    MLAUNCH_CHECK();
    const size_t lds = (size_t)F.cap * 4 + (size_t)Q.cap * 4 + (size_t)Q.cap * 2 + 16;
    if (lds > 160 * 1024) { orbx_set_error("capacities too large for the LDS tile"); return ORBX_ERR_CAPACITY; }
    if (lds > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_area_greedy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_area_greedy, dim3((unsigned)nframes), dim3(256), lds, m->stream, F, Q, max_dist, m->topk64.p, m->matches.p, m->dists.p, m->nmatches.p,
                       stride);
    MLAUNCH_CHECK();
    ORBX_HIP_CHECK(hipEventRecord(m->ev1[slot], m->stream));
    m->profCount++;
    m->lastPairs = nframes; m->lastStride = stride;
    return ORBX_OK;
}

extern "C" int orbx_area_search_greedy_device(orbx_matcher *m, const orbx_projection_frame *frame, const orbx_area_queries *q, int max_dist)
{
    if (!m || !frame || !q) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!frame->keypoints_un || !frame->descriptors || !frame->counts || !q->u || !q->v || !q->radius || !q->min_level || !q->max_level || !q->descriptors ||
        !q->counts) {
        orbx_set_error("NULL array in the area-search arguments");
        return ORBX_ERR_ARG;
    }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    ProjFrameDev F = {frame->keypoints_un, frame->descriptors, frame->u_right, frame->occupied, frame->counts, frame->capacity,
                      frame->min_x, frame->min_y, frame->grid_width_inv, frame->grid_height_inv};
    AreaQueriesDev Q = {q->u, q->v, q->radius, q->min_level, q->max_level, q->active, q->descriptors, q->counts, q->capacity, q->window_min_x, q->window_min_y};
    return area_launch(m, F, Q, frame->nframes, max_dist);
}

// host-array form for one frame: upload, run, download
extern "C" int orbx_area_search_greedy(orbx_matcher *m, const orbx_projection_frame *fr, const orbx_area_queries *q, int max_dist, int32_t *assigned,
                                       int32_t *dists, int32_t *nmatches)
{
    if (!m || !fr || !q || !assigned) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!fr->counts || !q->counts) { orbx_set_error("NULL counts"); return ORBX_ERR_ARG; }
    const int n = fr->counts[0], mm = q->counts[0];
    if (nmatches) *nmatches = 0;
    for (int i = 0; i < mm; i++) { assigned[i] = -1; if (dists) dists[i] = 256; }
    if (n <= 0 || mm <= 0) return ORBX_OK;
    if (mm > m->maxFeatures) { orbx_set_error("%d queries exceed the matcher's max_features %d", mm, m->maxFeatures); return ORBX_ERR_CAPACITY; }
    if (!fr->keypoints_un || !fr->descriptors || !q->u || !q->v || !q->radius || !q->min_level || !q->max_level || !q->descriptors) {
        orbx_set_error("NULL array in the area-search arguments");
        return ORBX_ERR_ARG;
    }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    int rc;
    // staging: frame side in pkp / hd[0] / pb[0]; query side in pf[1] (u, v, radius) / pi32[1] (min, max level) / pb[1] (descriptors, active)
    if ((rc = m->pkp.ensure((size_t)n)) || (rc = m->hd[0].ensure((size_t)n * 32)) || (rc = m->pb[0].ensure((size_t)n)) || (rc = m->pi32[0].ensure(2)) ||
        (rc = m->pf[1].ensure((size_t)mm * 3)) || (rc = m->pi32[1].ensure((size_t)mm * 2)) || (rc = m->pb[1].ensure((size_t)mm * 33)))
        return rc;
    hipStream_t st = m->stream;
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pkp.p, fr->keypoints_un, (size_t)n * sizeof(orbx_keypoint), hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->hd[0].p, fr->descriptors, (size_t)n * 32, hipMemcpyHostToDevice, st));
    if (fr->occupied) ORBX_HIP_CHECK(hipMemcpyAsync(m->pb[0].p, fr->occupied, (size_t)n, hipMemcpyHostToDevice, st));
    const int32_t cnt[2] = {n, mm};
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pi32[0].p, cnt, sizeof(cnt), hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pf[1].p, q->u, (size_t)mm * 4, hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pf[1].p + mm, q->v, (size_t)mm * 4, hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pf[1].p + 2 * (size_t)mm, q->radius, (size_t)mm * 4, hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pi32[1].p, q->min_level, (size_t)mm * 4, hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pi32[1].p + mm, q->max_level, (size_t)mm * 4, hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pb[1].p, q->descriptors, (size_t)mm * 32, hipMemcpyHostToDevice, st));
    if (q->active) ORBX_HIP_CHECK(hipMemcpyAsync(m->pb[1].p + (size_t)mm * 32, q->active, (size_t)mm, hipMemcpyHostToDevice, st));
    ProjFrameDev F = {m->pkp.p, m->hd[0].p, nullptr, fr->occupied ? m->pb[0].p : nullptr, m->pi32[0].p, n, fr->min_x, fr->min_y, fr->grid_width_inv,
                      fr->grid_height_inv};
    AreaQueriesDev Q = {m->pf[1].p, m->pf[1].p + mm, m->pf[1].p + 2 * (size_t)mm, m->pi32[1].p, m->pi32[1].p + mm,
                        q->active ? m->pb[1].p + (size_t)mm * 32 : nullptr, m->pb[1].p, m->pi32[0].p + 1, mm, q->window_min_x, q->window_min_y};
    if ((rc = area_launch(m, F, Q, 1, max_dist)) != ORBX_OK) return rc;
    ORBX_HIP_CHECK(hipStreamSynchronize(st));
    ORBX_HIP_CHECK(hipMemcpy(assigned, m->matches.p, (size_t)mm * 4, hipMemcpyDeviceToHost));
    if (dists) ORBX_HIP_CHECK(hipMemcpy(dists, m->dists.p, (size_t)mm * 4, hipMemcpyDeviceToHost));
    if (nmatches) ORBX_HIP_CHECK(hipMemcpy(nmatches, m->nmatches.p, 4, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

// ---------------------------------------------------------------------------------------------
// ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:515-654, Tracking::MonocularInitialization): level-0
// features of F1 look for their best / second best F2 feature inside a window around vbPrevMatched; a candidate
// is skipped when the F2 feature is already held at a distance <= its own (vMatchedDistance, :566), an accepted
// match overrides the previous holder (:590-594).
//   k_init_topk   (parallel, one wave per F1 feature) the window / level / distance part, which does not depend on
//                 that state: the 8 smallest keys dist << 32 | cell | i2 below the distance cut-off (a candidate at
//                 distance >= dcut can neither be accepted nor veto an acceptable best in the ratio test);
//   k_search_init (one wave per frame pair) walks the F1 features in order over those lists, applies the
//                 vMatchedDistance filter and the override; a FULL list with fewer than two usable entries is
//                 re-evaluated exactly against all of F2.
// ---------------------------------------------------------------------------------------------
struct InitQuery { float x, y, r; int cx0, cx1, cy0, cy1; bool any; unsigned long long d[4]; };

__device__ __forceinline__ InitQuery init_query(const FeatDev &A, const ProjFrameDev &F2, size_t ai, const float *prevXY, float window)
{
    InitQuery q;
    q.any = false;
    const orbx_keypoint k1 = A.kp[ai];
    if (k1.octave > 0) return q;                                                               // :535-537
    q.x = prevXY[2 * ai]; q.y = prevXY[2 * ai + 1]; q.r = window;
    // Frame::GetFeaturesInArea(x, y, r, 0, 0), src/Frame.cc:741-850
    q.cx0 = max(0, (int)floorf((q.x - F2.minX - q.r) * F2.gwInv)); q.cx1 = min(GRID_COLS - 1, (int)ceilf((q.x - F2.minX + q.r) * F2.gwInv));
    q.cy0 = max(0, (int)floorf((q.y - F2.minY - q.r) * F2.ghInv)); q.cy1 = min(GRID_ROWS - 1, (int)ceilf((q.y - F2.minY + q.r) * F2.ghInv));
    if (q.cx0 >= GRID_COLS || q.cx1 < 0 || q.cy0 >= GRID_ROWS || q.cy1 < 0) return q;
    const unsigned long long *dp = (const unsigned long long *)(A.desc + ai * 32);
    q.d[0] = dp[0]; q.d[1] = dp[1]; q.d[2] = dp[2]; q.d[3] = dp[3];
    q.any = true;
    return q;
}

// key of F2 feature i2 for the query, KEY64_EMPTY when it is not in the window / not on level 0
__device__ __forceinline__ unsigned long long init_key(const ProjFrameDev &F2, size_t fbase, int i2, const InitQuery &q)
{
    const orbx_keypoint k = F2.kp[fbase + i2];
    const int cx = (int)roundf((k.x - F2.minX) * F2.gwInv), cy = (int)roundf((k.y - F2.minY) * F2.ghInv);
    if (cx < q.cx0 || cx > q.cx1 || cy < q.cy0 || cy > q.cy1) return KEY64_EMPTY;
    if (k.octave < 0 || k.octave > 0) return KEY64_EMPTY;                                      // minLevel = maxLevel = level1 = 0
    const float distx = k.x - q.x, disty = k.y - q.y;
    if (!(fabsf(distx) < q.r && fabsf(disty) < q.r)) return KEY64_EMPTY;
    const unsigned long long *db = (const unsigned long long *)(F2.desc + (fbase + i2) * 32);
    const int dist = hamming256(q.d, db[0], db[1], db[2], db[3]);
    return ((unsigned long long)dist << 32) | ((unsigned long long)cx << 22) | ((unsigned long long)cy << 16) | (unsigned long long)i2;
}

__global__ __launch_bounds__(256) void k_init_topk(FeatDev A, ProjFrameDev F2, const float *__restrict__ prevXY, float window, int dcut,
                                                   unsigned long long *__restrict__ topk)
{
    const int f = blockIdx.y, lane = threadIdx.x & 63, i1 = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n1 = min(A.counts[f], A.cap), n2 = min(F2.counts[f], F2.cap);
    if (i1 >= n1) return;
    const size_t ai = (size_t)f * A.cap + i1, fbase = (size_t)f * F2.cap;
    unsigned long long *out = topk + ai * TOPK;
    const InitQuery q = init_query(A, F2, ai, prevXY, window);
    unsigned long long kk[TOPK];
#pragma unroll
    for (int t = 0; t < TOPK; t++) kk[t] = KEY64_EMPTY;
    if (q.any)
        for (int i2 = lane; i2 < n2; i2 += 64) {
            const unsigned long long key = init_key(F2, fbase, i2, q);
            if (key < kk[TOPK - 1] && (int)(key >> 32) < dcut) {
                kk[TOPK - 1] = key;
#pragma unroll
                for (int t = TOPK - 1; t > 0; t--)
                    if (kk[t] < kk[t - 1]) { const unsigned long long v = kk[t - 1]; kk[t - 1] = kk[t]; kk[t] = v; }
            }
        }
#pragma unroll
    for (int k = 0; k < TOPK; k++) {
        const unsigned long long mn = wave_min_u64(kk[0]);
        if (kk[0] == mn && mn != KEY64_EMPTY) {   // keys are unique: exactly one lane pops its head
#pragma unroll
            for (int t = 0; t < TOPK - 1; t++) kk[t] = kk[t + 1];
            kk[TOPK - 1] = KEY64_EMPTY;
        }
        if (lane == 0) out[k] = mn;
    }
}

__global__ __launch_bounds__(256) void k_search_init(FeatDev A, ProjFrameDev F2, const float *__restrict__ prevXY, float window, float nnratio, int checkOri,
                                                     const unsigned long long *__restrict__ topk, int32_t *__restrict__ matches, int8_t *__restrict__ bins,
                                                     int32_t *__restrict__ nmatches, int stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int sTotal;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int n1 = min(A.counts[f], A.cap), n2 = min(F2.counts[f], F2.cap);
    unsigned short *holderDist = (unsigned short *)smem;   // vMatchedDistance, 0xffff = INT_MAX
    unsigned short *match21 = holderDist + F2.cap;         // vnMatches21, 0xffff = -1
    const size_t abase = (size_t)f * A.cap, fbase = (size_t)f * F2.cap;
    int32_t *m12 = matches + (size_t)f * stride;
    int8_t *bin12 = bins + (size_t)f * stride;
    for (int i = tid; i < n2; i += 256) { holderDist[i] = 0xffff; match21[i] = 0xffff; }
    for (int i = tid; i < stride; i += 256) { m12[i] = -1; bin12[i] = -1; }
    if (tid < HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) sTotal = 0;
    __syncthreads();
    const float factor = HISTO_LENGTH / 360.0f;
    if (tid < 64) {
        int total = 0;
        const unsigned long long *tk = topk + abase * TOPK;
        unsigned long long nextKeys = (0 < n1) ? tk[min((size_t)lane, (size_t)n1 * TOPK - 1)] : KEY64_EMPTY;
        for (int i0 = 0; i0 < n1; i0 += 8) {
            const unsigned long long keys = nextKeys;   // lists of features i0 .. i0+7, lane = 8*(i1-i0) + rank
            {
                const size_t nx = (size_t)(i0 + 8) * TOPK + lane;
                nextKeys = (i0 + 8 < n1) ? tk[min(nx, (size_t)n1 * TOPK - 1)] : KEY64_EMPTY;   // in flight while these 8 are replayed
            }
            for (int j = 0; j < 8 && i0 + j < n1; j++) {
                const int i1 = i0 + j;
                const unsigned long long key = __shfl(keys, 8 * j + (lane & 7));   // lanes 0..7 hold the list of feature i1
                const bool present = lane < TOPK && key != KEY64_EMPTY;
                const unsigned mAll = (1u << TOPK) - 1u;
                const unsigned mPresent = (unsigned)(__ballot(present) & mAll);
                if (!mPresent) continue;                                            // higher level, empty window or nothing below the cut-off
                const bool usable = present && !((int)holderDist[(int)(key & 0xffff)] <= (int)(key >> 32));   // :566 (0xffff stands for INT_MAX)
                const unsigned mUse = (unsigned)(__ballot(usable) & mAll);
                unsigned long long b = KEY64_EMPTY, s2 = KEY64_EMPTY;
                if (__popc(mUse) >= 2 || mPresent != mAll) {
                    if (mUse) {
                        b = __shfl(key, __ffs(mUse) - 1);
                        const unsigned rest = mUse & (mUse - 1);
                        if (rest) s2 = __shfl(key, __ffs(rest) - 1);
                    }
                } else {
                    // full list, fewer than two usable entries: the two smallest usable keys over all of F2
                    const InitQuery q = init_query(A, F2, abase + i1, prevXY, window);
                    unsigned long long k0 = KEY64_EMPTY, kk1 = KEY64_EMPTY;
                    if (q.any)
                        for (int i2 = lane; i2 < n2; i2 += 64) {
                            const unsigned long long kx = init_key(F2, fbase, i2, q);
                            if (kx == KEY64_EMPTY || (int)holderDist[i2] <= (int)(kx >> 32)) continue;
                            if (kx < k0) { kk1 = k0; k0 = kx; } else if (kx < kk1) kk1 = kx;
                        }
                    b = wave_min_u64(k0);
                    if (k0 == b) k0 = kk1;
                    s2 = wave_min_u64(k0);
                }
                if (b == KEY64_EMPTY) continue;
                const int bestDist = (int)(b >> 32), bestIdx2 = (int)(b & 0xffff);
                const float second = s2 == KEY64_EMPTY ? (float)2147483647 : (float)(int)(s2 >> 32);   // (float)INT_MAX, :576
                if (bestDist <= TH_LOW && (float)bestDist < second * nnratio) {
                    if (lane == 0) {
                        if (match21[bestIdx2] != 0xffff) { m12[match21[bestIdx2]] = -1; total--; }      // :590-594
                        m12[i1] = bestIdx2;
                        match21[bestIdx2] = (unsigned short)i1;
                        holderDist[bestIdx2] = (unsigned short)bestDist;
                        total++;
                        if (checkOri) {
                            float rot = A.kp[abase + i1].angle - F2.kp[fbase + bestIdx2].angle;
                            if (rot < 0.0f) rot += 360.0f;
                            int bin = (int)roundf(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            bin12[i1] = (int8_t)bin;
                            hist[bin]++;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (lane == 0) sTotal = total;
    }
    __syncthreads();
    if (checkOri) {
        // ComputeThreeMaxima over the counts of ALL accepted events (an overridden match stays in its bin, :606)
        int ind1, ind2, ind3;
        three_maxima_wave(hist, lane, ind1, ind2, ind3);
        __threadfence_block();
        int removed = 0;
        for (int i = tid; i < n1; i += 256) {
            const int b = bin12[i];
            if (b >= 0 && b != ind1 && b != ind2 && b != ind3 && m12[i] >= 0) { m12[i] = -1; removed++; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o);
        if (lane == 0 && removed) atomicSub(&sTotal, removed);
    }
    __syncthreads();
    if (tid == 0) nmatches[f] = sTotal;
}

extern "C" int orbx_search_for_initialization_device(orbx_matcher *m, const orbx_feature_set *f1, const orbx_projection_frame *f2, const float *prev_matched_xy,
                                                     int window_size, float nn_ratio, int check_orientation)
{
    if (!m || !f1 || !f2 || !prev_matched_xy) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!f1->keypoints || !f1->descriptors || !f1->counts || !f2->keypoints_un || !f2->descriptors || !f2->counts) { orbx_set_error("NULL feature arrays"); return ORBX_ERR_ARG; }
    const int nframes = f2->nframes;
    if (nframes < 1 || nframes > m->maxPairs || f1->nframes != nframes) { orbx_set_error("frame counts %d/%d outside 1..%d", f1->nframes, nframes, m->maxPairs); return ORBX_ERR_CAPACITY; }
    if (f1->capacity < 1 || f1->capacity > m->maxFeatures || f2->capacity < 1 || f2->capacity > 65534) { orbx_set_error("bad capacities (%d, %d)", f1->capacity, f2->capacity); return ORBX_ERR_CAPACITY; }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    int rc = m->pb[0].ensure((size_t)m->maxPairs * m->maxFeatures);
    if (rc != ORBX_OK) return rc;
    FeatDev A = to_dev(f1);
    ProjFrameDev F = {f2->keypoints_un, f2->descriptors, nullptr, nullptr, f2->counts, f2->capacity, f2->min_x, f2->min_y, f2->grid_width_inv, f2->grid_height_inv};
    const int stride = m->maxFeatures;
    const size_t lds = (size_t)f2->capacity * 4 + 16;
    if (lds > 160 * 1024) { orbx_set_error("feature capacity %d too large for the LDS tile", f2->capacity); return ORBX_ERR_CAPACITY; }
    if (lds > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_search_init, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if ((rc = m->topk64.ensure((size_t)nframes * f1->capacity * TOPK)) != ORBX_OK) return rc;
    // first distance that can neither be accepted (> TH_LOW) nor veto an acceptable best in `(float)bestDist < (float)bestDist2 * mfNNratio` (:580)
    int dcut = TH_LOW + 1;
    while (dcut < 257 && !((float)dcut * nn_ratio > (float)TH_LOW)) dcut++;
    const int slot = m->profCount % MATCH_PROF_RING;
    ORBX_HIP_CHECK(hipEventRecord(m->ev0[slot], m->stream));
    m->midValid[slot] = false;
    hipLaunchKernelGGL(k_init_topk, dim3((unsigned)((f1->capacity + 3) / 4), (unsigned)nframes), dim3(256), 0, m->stream, A, F, prev_matched_xy, (float)window_size, dcut,
                       m->topk64.p);
    MLAUNCH_CHECK();
    hipLaunchKernelGGL(k_search_init, dim3((unsigned)nframes), dim3(256), lds, m->stream, A, F, prev_matched_xy, (float)window_size, nn_ratio, check_orientation,
                       m->topk64.p, m->matches.p, (int8_t *)m->pb[0].p, m->nmatches.p, stride);
    MLAUNCH_CHECK();
    ORBX_HIP_CHECK(hipEventRecord(m->ev1[slot], m->stream));
    m->profCount++;
    m->lastPairs = nframes; m->lastStride = stride;
    return ORBX_OK;
}


// host-array form for one frame pair: upload, run, download
extern "C" int orbx_search_for_initialization(orbx_matcher *m, const orbx_feature_set *f1_host, const orbx_projection_frame *f2_host, const float *prev_matched_xy,
                                              int window_size, float nn_ratio, int check_orientation, int32_t *matches12, int32_t *nmatches)
{
    if (!m || !f1_host || !f2_host || !matches12 || !nmatches) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!f1_host->counts || !f2_host->counts) { orbx_set_error("NULL counts"); return ORBX_ERR_ARG; }
    const int n1 = f1_host->counts[0], n2 = f2_host->counts[0];
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 <= 0 || n2 <= 0) return ORBX_OK;
    if (!prev_matched_xy || !f2_host->keypoints_un || !f2_host->descriptors) { orbx_set_error("NULL array"); return ORBX_ERR_ARG; }
    if (n2 > m->maxFeatures) { orbx_set_error("%d features exceed the matcher's max_features %d", n2, m->maxFeatures); return ORBX_ERR_CAPACITY; }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    orbx_feature_set d1;
    int rc;
    if ((rc = stage_host(m, 0, f1_host, &d1)) != ORBX_OK) return rc;
    if ((rc = m->pkp.ensure((size_t)n2)) || (rc = m->hd[1].ensure((size_t)n2 * 32)) || (rc = m->pi32[0].ensure(2)) || (rc = m->pf[1].ensure((size_t)m->maxFeatures * 2)))
        return rc;
    hipStream_t st = m->stream;
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pkp.p, f2_host->keypoints_un, (size_t)n2 * sizeof(orbx_keypoint), hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->hd[1].p, f2_host->descriptors, (size_t)n2 * 32, hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pi32[0].p, &n2, sizeof(int32_t), hipMemcpyHostToDevice, st));
    ORBX_HIP_CHECK(hipMemcpyAsync(m->pf[1].p, prev_matched_xy, (size_t)n1 * 2 * sizeof(float), hipMemcpyHostToDevice, st));
    orbx_projection_frame d2 = *f2_host;
    d2.keypoints_un = m->pkp.p; d2.descriptors = m->hd[1].p; d2.u_right = nullptr; d2.occupied = nullptr; d2.counts = m->pi32[0].p; d2.capacity = n2; d2.nframes = 1;
    if ((rc = orbx_search_for_initialization_device(m, &d1, &d2, m->pf[1].p, window_size, nn_ratio, check_orientation)) != ORBX_OK) return rc;
    ORBX_HIP_CHECK(hipStreamSynchronize(st));
    ORBX_HIP_CHECK(hipMemcpy(matches12, m->matches.p, (size_t)n1 * 4, hipMemcpyDeviceToHost));
    ORBX_HIP_CHECK(hipMemcpy(nmatches, m->nmatches.p, 4, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

static int proj_launch(orbx_matcher *m, const ProjFrameDev &F, const ProjPointsDev &P, int nframes, const float *scale_factors, int nlevels, float th,
                       float nnratio)
{
    if (nframes < 1 || nframes > m->maxPairs) { orbx_set_error("nframes %d outside 1..%d", nframes, m->maxPairs); return ORBX_ERR_CAPACITY; }
    if (F.cap < 1 || F.cap > m->maxFeatures || F.cap > 65535 || P.cap < 1) { orbx_set_error("bad capacities (features %d, points %d)", F.cap, P.cap); return ORBX_ERR_CAPACITY; }
    if (!scale_factors || nlevels < 1 || nlevels > 64) { orbx_set_error("bad scale factor table"); return ORBX_ERR_ARG; }
    int rc = m->topk64.ensure((size_t)nframes * P.cap * TOPK);
    if (rc != ORBX_OK) return rc;
    const int stride = m->maxFeatures;
    ORBX_HIP_CHECK(hipMemcpyAsync(m->scales.p, scale_factors, (size_t)nlevels * sizeof(float), hipMemcpyHostToDevice, m->stream));
    const int slot = m->profCount % MATCH_PROF_RING;
    ORBX_HIP_CHECK(hipEventRecord(m->ev0[slot], m->stream));
    m->midValid[slot] = false;
    hipLaunchKernelGGL(k_proj_topk, dim3((unsigned)((P.cap + 3) / 4), (unsigned)nframes), dim3(256), 0, m->stream, F, P, m->scales.p, th, m->topk64.p);
    MLAUNCH_CHECK();
    const size_t lds = (size_t)F.cap * 6 + 16;
    if (lds > 160 * 1024) { orbx_set_error("feature capacity %d too large for the LDS tile of the projection replay", F.cap); return ORBX_ERR_CAPACITY; }
    if ((rc = m->projDec.ensure((size_t)nframes * P.cap)) != ORBX_OK || (rc = m->projQueue.ensure((size_t)nframes * P.cap)) != ORBX_OK) return rc;
    if (lds > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_proj_greedy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_proj_greedy, dim3((unsigned)nframes), dim3(PROJ_GREEDY_THREADS), lds, m->stream, F, P, m->scales.p, th, nnratio, m->topk64.p, m->matches.p,
                       m->nmatches.p, stride, m->projDec.p, m->projQueue.p, (int32_t *)nullptr, (unsigned long long *)nullptr, 0ull, 0);
    MLAUNCH_CHECK();
    ORBX_HIP_CHECK(hipEventRecord(m->ev1[slot], m->stream));
    m->profCount++;
    m->lastPairs = nframes; m->lastStride = stride;
    return ORBX_OK;
}

extern "C" int orbx_search_by_projection_device(orbx_matcher *m, const orbx_projection_frame *frame, const orbx_projection_points *points,
                                                const float *scale_factors, int nlevels, float th, float nn_ratio)
{
    if (!m || !frame || !points) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!frame->keypoints_un || !frame->descriptors || !frame->u_right || !frame->counts || !points->proj_x || !points->proj_y || !points->proj_xr ||
        !points->scale_level || !points->view_cos || !points->in_view || !points->descriptors || !points->counts) {
        orbx_set_error("NULL array in the projection arguments");
        return ORBX_ERR_ARG;
    }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    ProjFrameDev F = {frame->keypoints_un, frame->descriptors, frame->u_right, frame->occupied, frame->counts, frame->capacity,
                      frame->min_x, frame->min_y, frame->grid_width_inv, frame->grid_height_inv};
    ProjPointsDev P = {points->proj_x, points->proj_y, points->proj_xr, points->scale_level, points->view_cos, points->in_view, points->has_observations,
                       points->descriptors, points->counts, points->capacity};
    return proj_launch(m, F, P, frame->nframes, scale_factors, nlevels, th, nn_ratio);
}

// host-array form for one frame: upload, run, download
extern "C" int orbx_search_by_projection(orbx_matcher *m, const orbx_projection_frame *fr, const orbx_projection_points *pt, const float *scale_factors,
                                         int nlevels, float th, float nn_ratio, int32_t *assigned, int32_t *nmatches)
{
    if (!m || !fr || !pt || !assigned) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!fr->counts || !pt->counts) { orbx_set_error("NULL counts"); return ORBX_ERR_ARG; }
    const int n = fr->counts[0], mm = pt->counts[0];
    if (nmatches) *nmatches = 0;
    for (int i = 0; i < n; i++) assigned[i] = -1;
    if (n <= 0 || mm <= 0) return ORBX_OK;
    if (n > m->maxFeatures) { orbx_set_error("%d features exceed the matcher's max_features %d", n, m->maxFeatures); return ORBX_ERR_CAPACITY; }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    int rc;
    // no copy engine, no stream synchronisation (OrbxCallBox): all thirteen input arrays through mapped pinned memory, one pass of k_stage_copy to the
    // device arena (the frame's arrays are read by every wave, the points' by the replay's rounds), the two kernels, results + sequence word from the replay
    hipStream_t st = m->stream;
    OrbxCallBox &bx = m->box;
    const size_t N = (size_t)n, M = (size_t)mm;
    const size_t total = bx.padded(N * sizeof(orbx_keypoint)) + bx.padded(N * 32) + bx.padded(N * 4) + bx.padded(N) + bx.padded(8) + 5 * bx.padded(M * 4) +
                         bx.padded(M * 32) + 2 * bx.padded(M) + bx.padded(64 * 4);
    if (!scale_factors || nlevels < 1 || nlevels > 64) { orbx_set_error("bad scale factor table"); return ORBX_ERR_ARG; }
    if (!fr->keypoints_un || !fr->descriptors || !fr->u_right || !pt->proj_x || !pt->proj_y || !pt->proj_xr || !pt->scale_level || !pt->view_cos || !pt->in_view || !pt->descriptors) {
        orbx_set_error("NULL array in the projection arguments");
        return ORBX_ERR_ARG;
    }
    if ((rc = bx.begin(total, bx.padded((N + 1) * 4), st)) != ORBX_OK) return rc;
    const int32_t cnt[2] = {n, mm};
    const void *bKp = bx.put(fr->keypoints_un, N), *bDesc = bx.put(fr->descriptors, N * 32), *bUr = bx.put(fr->u_right, N), *bOcc = bx.put(fr->occupied, fr->occupied ? N : 0);
    const void *bCnt = bx.put(cnt, 2);
    const void *bPx = bx.put(pt->proj_x, M), *bPy = bx.put(pt->proj_y, M), *bPxr = bx.put(pt->proj_xr, M), *bCos = bx.put(pt->view_cos, M), *bLvl = bx.put(pt->scale_level, M);
    const void *bPd = bx.put(pt->descriptors, M * 32), *bIn = bx.put(pt->in_view, M), *bObs = bx.put(pt->has_observations, pt->has_observations ? M : 0);
    const void *bSc = bx.put(scale_factors, (size_t)nlevels);
    if ((rc = m->arena.ensure(bx.used)) != ORBX_OK) return rc;
    uint8_t *const ar = m->arena.p;
    auto dev = [&](const void *boxAddr) { return ar + ((const uint8_t *)boxAddr - bx.inDev); };
    const size_t n16 = bx.used / 16;
    hipLaunchKernelGGL(k_stage_copy, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 512)), dim3(256), 0, st, (const uint4 *)bx.inDev, (uint4 *)ar, n16);
    MLAUNCH_CHECK();
    const int32_t *dCnt = (const int32_t *)dev(bCnt);
    const float *dScales = (const float *)dev(bSc);
    ProjFrameDev F = {(const orbx_keypoint *)dev(bKp), dev(bDesc), (const float *)dev(bUr), fr->occupied ? dev(bOcc) : nullptr, dCnt, n, fr->min_x, fr->min_y, fr->grid_width_inv,
                      fr->grid_height_inv};
    ProjPointsDev P = {(const float *)dev(bPx), (const float *)dev(bPy), (const float *)dev(bPxr), (const int32_t *)dev(bLvl), (const float *)dev(bCos), dev(bIn),
                       pt->has_observations ? dev(bObs) : nullptr, dev(bPd), dCnt + 1, mm};
    if ((rc = m->topk64.ensure(M * TOPK)) != ORBX_OK || (rc = m->projDec.ensure(M)) != ORBX_OK || (rc = m->projQueue.ensure(M)) != ORBX_OK) return rc;
    hipLaunchKernelGGL(k_proj_topk, dim3((unsigned)((mm + 3) / 4), 1u), dim3(256), 0, st, F, P, dScales, th, m->topk64.p);
    MLAUNCH_CHECK();
    size_t lds = (size_t)n * 6 + 16;
    if (lds > 160 * 1024) { orbx_set_error("feature count %d too large for the LDS tile of the projection replay", n); return ORBX_ERR_CAPACITY; }
    int decLds = 0;      // dec[mm] + the observation flags behind the feature tile, when they fit (k_proj_greedy)
    if (((lds + 15) & ~(size_t)15) + M * 5 <= 150 * 1024) { decLds = (int)((lds + 15) & ~(size_t)15); lds = (size_t)decLds + M * 5; }
    if (lds > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_proj_greedy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int stride = m->maxFeatures;
    const unsigned long long seq = bx.arm();
    hipLaunchKernelGGL(k_proj_greedy, dim3(1), dim3(PROJ_GREEDY_THREADS), lds, st, F, P, dScales, th, nn_ratio, m->topk64.p, m->matches.p, m->nmatches.p, stride, m->projDec.p,
                       m->projQueue.p, bx.outDev<int32_t>(0), bx.flagDev, seq, decLds);
    MLAUNCH_CHECK();
    m->lastPairs = 1; m->lastStride = stride;
    if ((rc = bx.wait(st)) != ORBX_OK) return rc;
    const int32_t *as = bx.outHost<int32_t>(0);
    memcpy(assigned, as, N * 4);
    if (nmatches) *nmatches = as[n];
    return ORBX_OK;
}

static int proj_last_launch(orbx_matcher *m, const ProjFrameDev &F, const ProjLastDev &L, int nframes, const float *scale_factors, int nlevels, float th,
                            int b_mono, int check_ori)
{
    if (nframes < 1 || nframes > m->maxPairs) { orbx_set_error("nframes %d outside 1..%d", nframes, m->maxPairs); return ORBX_ERR_CAPACITY; }
    if (F.cap < 1 || F.cap > m->maxFeatures || F.cap > 65535 || L.cap < 1 || L.cap > 65535) { orbx_set_error("bad capacities (features %d, last %d)", F.cap, L.cap); return ORBX_ERR_CAPACITY; }
    if (!scale_factors || nlevels < 1 || nlevels > 64) { orbx_set_error("bad scale factor table"); return ORBX_ERR_ARG; }
    int rc = m->topk64.ensure((size_t)nframes * L.cap * TOPK);
    if (rc != ORBX_OK) return rc;
    const int stride = m->maxFeatures;
    ORBX_HIP_CHECK(hipMemcpyAsync(m->scales.p, scale_factors, (size_t)nlevels * sizeof(float), hipMemcpyHostToDevice, m->stream));
    const int slot = m->profCount % MATCH_PROF_RING;
    ORBX_HIP_CHECK(hipEventRecord(m->ev0[slot], m->stream));
    m->midValid[slot] = false;
    hipLaunchKernelGGL(k_proj_last_topk, dim3((unsigned)((L.cap + 3) / 4), (unsigned)nframes), dim3(256), 0, m->stream, F, L, m->scales.p, th, b_mono, m->topk64.p);
    MLAUNCH_CHECK();
    const size_t lds = (size_t)F.cap * 5 + 16;
    if (lds > 160 * 1024) { orbx_set_error("capacities too large for the LDS tile"); return ORBX_ERR_CAPACITY; }
    if ((rc = m->projDec.ensure((size_t)nframes * L.cap)) != ORBX_OK || (rc = m->projQueue.ensure((size_t)nframes * L.cap)) != ORBX_OK) return rc;
    if (lds > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_proj_last_greedy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_proj_last_greedy, dim3((unsigned)nframes), dim3(PROJ_GREEDY_THREADS), lds, m->stream, F, L, m->scales.p, th, b_mono, check_ori, m->topk64.p,
                       m->matches.p, m->nmatches.p, stride, m->projDec.p, m->projQueue.p, (int32_t *)nullptr, (unsigned long long *)nullptr, 0ull);
    MLAUNCH_CHECK();
    ORBX_HIP_CHECK(hipEventRecord(m->ev1[slot], m->stream));
    m->profCount++;
    m->lastPairs = nframes; m->lastStride = stride;
    return ORBX_OK;
}

extern "C" int orbx_search_by_projection_last_device(orbx_matcher *m, const orbx_projection_frame *frame, const orbx_projection_last *last,
                                                     const float *scale_factors, int nlevels, float th, int b_mono, int check_orientation)
{
    if (!m || !frame || !last) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!frame->keypoints_un || !frame->descriptors || !frame->u_right || !frame->counts || !last->valid || !last->world_pos || !last->descriptors ||
        !last->octave || !last->angle || !last->counts || !last->tcw_current || !last->tcw_last) {
        orbx_set_error("NULL array in the projection arguments");
        return ORBX_ERR_ARG;
    }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    ProjFrameDev F = {frame->keypoints_un, frame->descriptors, frame->u_right, frame->occupied, frame->counts, frame->capacity,
                      frame->min_x, frame->min_y, frame->grid_width_inv, frame->grid_height_inv};
    ProjLastDev L = {last->valid, last->world_pos, last->descriptors, last->has_observations, last->octave, last->angle, last->counts, last->capacity,
                     last->tcw_current, last->tcw_last, last->fx, last->fy, last->cx, last->cy, last->mbf, last->mb, last->max_x, last->max_y};
    return proj_last_launch(m, F, L, frame->nframes, scale_factors, nlevels, th, b_mono, check_orientation);
}

extern "C" int orbx_search_by_projection_last(orbx_matcher *m, const orbx_projection_frame *fr, const orbx_projection_last *ls, const float *scale_factors,
                                              int nlevels, float th, int b_mono, int check_orientation, int32_t *assigned, int32_t *nmatches)
{
    if (!m || !fr || !ls || !assigned) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!fr->counts || !ls->counts) { orbx_set_error("NULL counts"); return ORBX_ERR_ARG; }
    const int n = fr->counts[0], nl = ls->counts[0];
    if (nmatches) *nmatches = 0;
    for (int i = 0; i < n; i++) assigned[i] = -1;
    if (n <= 0 || nl <= 0) return ORBX_OK;
    if (n > m->maxFeatures) { orbx_set_error("%d features exceed the matcher's max_features %d", n, m->maxFeatures); return ORBX_ERR_CAPACITY; }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    // The matcher of Tracking::TrackWithMotionModel, once per tracked frame: no copy engine, no stream synchronisation (OrbxCallBox).  Both sides go into
    // mapped pinned memory, k_stage_copy brings them to the device arena in one pass (every wave of the candidate kernel reads the frame's features; the
    // replay reads the last frame's flags and angles round after round), k_proj_last_topk, then k_proj_last_greedy, which writes assigned[] + the count
    // into the mapped buffer and raises the sequence word.  (Fourteen pageable uploads, a synchronisation and two blocking downloads before.)
    int rc;
    hipStream_t st = m->stream;
    OrbxCallBox &bx = m->box;
    const size_t N = (size_t)n, NL = (size_t)nl;
    const size_t inBytes = bx.padded(N * sizeof(orbx_keypoint)) + bx.padded(N * 32) + bx.padded(N * 4) + bx.padded(N) + bx.padded(8) + bx.padded(NL * 12) + bx.padded(NL * 4) + 2 * bx.padded(64) +
                           bx.padded(NL * 4) + bx.padded(NL * 32) + 2 * bx.padded(NL) + bx.padded(64 * 4);
    if (!scale_factors || nlevels < 1 || nlevels > 64) { orbx_set_error("bad scale factor table"); return ORBX_ERR_ARG; }
    if (nl > 65535) { orbx_set_error("bad capacities (features %d, last %d)", n, nl); return ORBX_ERR_CAPACITY; }
    if ((rc = bx.begin(inBytes, bx.padded((N + 1) * 4), st)) != ORBX_OK) return rc;
    const int32_t cnt[2] = {n, nl};
    const void *bKp = bx.put(fr->keypoints_un, N), *bDesc = bx.put(fr->descriptors, N * 32), *bUr = bx.put(fr->u_right, N), *bOcc = bx.put(fr->occupied, fr->occupied ? N : 0);
    const void *bCnt = bx.put(cnt, 2), *bPos = bx.put(ls->world_pos, NL * 3), *bAng = bx.put(ls->angle, NL), *bTc = bx.put(ls->tcw_current, 16), *bTl = bx.put(ls->tcw_last, 16);
    const void *bOct = bx.put(ls->octave, NL), *bLd = bx.put(ls->descriptors, NL * 32), *bVal = bx.put(ls->valid, NL), *bObs = bx.put(ls->has_observations, ls->has_observations ? NL : 0);
    const void *bSc = bx.put(scale_factors, (size_t)nlevels);
    if ((rc = m->arena.ensure(bx.used)) != ORBX_OK) return rc;
    uint8_t *const ar = m->arena.p;
    auto dev = [&](const void *boxAddr) { return ar + ((const uint8_t *)boxAddr - bx.inDev); };
    const size_t n16 = bx.used / 16;
    hipLaunchKernelGGL(k_stage_copy, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 512)), dim3(256), 0, st, (const uint4 *)bx.inDev, (uint4 *)ar, n16);
    MLAUNCH_CHECK();
    const int32_t *dCnt = (const int32_t *)dev(bCnt);
    const float *dScales = (const float *)dev(bSc);
    ProjFrameDev F = {(const orbx_keypoint *)dev(bKp), dev(bDesc), (const float *)dev(bUr), fr->occupied ? dev(bOcc) : nullptr, dCnt, n, fr->min_x, fr->min_y, fr->grid_width_inv,
                      fr->grid_height_inv};
    ProjLastDev L = {dev(bVal), (const float *)dev(bPos), dev(bLd), ls->has_observations ? dev(bObs) : nullptr, (const int32_t *)dev(bOct), (const float *)dev(bAng), dCnt + 1, nl,
                     (const float *)dev(bTc), (const float *)dev(bTl), ls->fx, ls->fy, ls->cx, ls->cy, ls->mbf, ls->mb, ls->max_x, ls->max_y};
    if ((rc = m->topk64.ensure(NL * TOPK)) != ORBX_OK || (rc = m->projDec.ensure(NL)) != ORBX_OK || (rc = m->projQueue.ensure(NL)) != ORBX_OK) return rc;
    hipLaunchKernelGGL(k_proj_last_topk, dim3((unsigned)((nl + 3) / 4), 1u), dim3(256), 0, st, F, L, dScales, th, b_mono, m->topk64.p);
    MLAUNCH_CHECK();
    const size_t lds = (size_t)n * 5 + 16;
    if (lds > 160 * 1024) { orbx_set_error("capacities too large for the LDS tile"); return ORBX_ERR_CAPACITY; }
    if (lds > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_proj_last_greedy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int stride = m->maxFeatures;
    const unsigned long long seq = bx.arm();
    hipLaunchKernelGGL(k_proj_last_greedy, dim3(1), dim3(PROJ_GREEDY_THREADS), lds, st, F, L, dScales, th, b_mono, check_orientation, m->topk64.p, m->matches.p, m->nmatches.p, stride,
                       m->projDec.p, m->projQueue.p, bx.outDev<int32_t>(0), bx.flagDev, seq);
    MLAUNCH_CHECK();
    m->lastPairs = 1; m->lastStride = stride;
    if ((rc = bx.wait(st)) != ORBX_OK) return rc;
    const int32_t *as = bx.outHost<int32_t>(0);
    memcpy(assigned, as, N * 4);
    if (nmatches) *nmatches = as[n];
    return ORBX_OK;
}


// ---------------------------------------------------------------------------------------------
// Frame::isInFrustum (src/Frame.cc:608-742) for a list of map points per frame, the loop of
// Tracking::SearchLocalPoints (src/Tracking.cc:1580-1613) that feeds SearchByProjection(F, vpMapPoints, th):
// projection with mRcw / mtcw, image-bounds, distance-range and viewing-angle gates, MapPoint::PredictScale,
// and the mTrack* fields.  Arithmetic in the reference's order: 3x3 * 3x1 products in float left to right
// (OpenCV's small-matrix gemm path, see oracle/cvshim), cv::norm and Mat::dot accumulate in double.
// PredictScale's ceil(log(ratio)/mfLogScaleFactor) is evaluated WITHOUT a device logarithm: the host tabulates,
// with the same libm log the reference calls, the largest float ratio that still maps to each level
// (orbx_predict_scale_thresholds; exact as long as that log is monotonic), the device compares.
// ---------------------------------------------------------------------------------------------
struct FrustumDev {
    const float *tcw;        // [16] per frame
    float fx, fy, cx, cy, mbf, minX, maxX, minY, maxY, cosLimit;
    int nlevels;
    float ratioTh[ORBX_MAX_LEVELS];   // ratioTh[k] = largest ratio with PredictScale <= k, k = 0 .. nlevels-2
};
struct MapPointsDev { const float *pos, *normal, *maxDist, *minDist; const int32_t *counts; int cap; };

// Frame::isInFrustum for one point from its values: false = not in view (:615); true: the mTrack* values (:721-731)
__device__ __forceinline__ bool frustum_eval(const FrustumDev &Fr, const float *T, const float P[3], const float Pn[3], float maxD, float minD, float &u, float &v, float &ur,
                                             int &lvlOut, float &viewCosOut)
{
    float Pc[3], Ow[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        float s = T[r * 4 + 0] * P[0];
        s = s + T[r * 4 + 1] * P[1];
        s = s + T[r * 4 + 2] * P[2];
        Pc[r] = s + T[r * 4 + 3];                                                // mRcw*P + mtcw, :627
        float o = (-T[0 * 4 + r]) * T[0 * 4 + 3];                                // mOw = -mRcw.t()*mtcw, src/Frame.cc:598
        o = o + (-T[1 * 4 + r]) * T[1 * 4 + 3];
        o = o + (-T[2 * 4 + r]) * T[2 * 4 + 3];
        Ow[r] = o;
    }
    if (Pc[2] < 0.0f) return false;                                              // :635
    const float invz = 1.0f / Pc[2];
    u = Fr.fx * Pc[0] * invz + Fr.cx; v = Fr.fy * Pc[1] * invz + Fr.cy;
    if (u < Fr.minX || u > Fr.maxX) return false;                                // :653-656
    if (v < Fr.minY || v > Fr.maxY) return false;
    const float maxDistance = 1.2f * maxD, minDistance = 0.8f * minD;            // Get{Max,Min}DistanceInvariance, src/MapPoint.cc:523-533
    const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
    const float dist = (float)sqrt((double)PO[0] * (double)PO[0] + (double)PO[1] * (double)PO[1] + (double)PO[2] * (double)PO[2]);   // cv::norm, :677
    if (dist < minDistance || dist > maxDistance) return false;                  // :680
    const double dot = (double)PO[0] * (double)Pn[0] + (double)PO[1] * (double)Pn[1] + (double)PO[2] * (double)Pn[2];
    const float viewCos = (float)(dot / (double)dist);                           // :697
    if (viewCos < Fr.cosLimit) return false;
    const float ratio = maxD / dist;                                             // MapPoint::PredictScale, src/MapPoint.cc:571-586
    int lvl = Fr.nlevels - 1;
    for (int k = Fr.nlevels - 2; k >= 0; k--)
        if (!(ratio > Fr.ratioTh[k])) lvl = k;
    ur = u - Fr.mbf * invz; lvlOut = lvl; viewCosOut = viewCos;
    return true;
}

// ... for point pi of frame f, its values requested together (the early exits used to put each array's load behind the previous test)
__device__ __forceinline__ bool frustum_point(const FrustumDev &Fr, const MapPointsDev &M, int f, size_t pi, float &u, float &v, float &ur, int &lvlOut, float &viewCosOut)
{
    const float *Pp = M.pos + 3 * pi, *Np = M.normal + 3 * pi;
    const float P[3] = {Pp[0], Pp[1], Pp[2]}, Pn[3] = {Np[0], Np[1], Np[2]};
    const float maxD = M.maxDist[pi], minD = M.minDist[pi];
    return frustum_eval(Fr, Fr.tcw + 16 * (size_t)f, P, Pn, maxD, minD, u, v, ur, lvlOut, viewCosOut);
}

__global__ __launch_bounds__(256) void k_is_in_frustum(FrustumDev Fr, MapPointsDev M, float *__restrict__ projX, float *__restrict__ projY, float *__restrict__ projXR,
                                                       int32_t *__restrict__ level, float *__restrict__ viewCosOut, uint8_t *__restrict__ inView, unsigned *pubCounter,
                                                       unsigned long long *pubFlag, unsigned long long pubSeq)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int m = min(M.counts[f], M.cap);
    if (i < m) {
        const size_t pi = (size_t)f * M.cap + i;
        float u = 0, v = 0, ur = 0, vc = 0;
        int lvl = 0;
        const bool in = frustum_point(Fr, M, f, pi, u, v, ur, lvl, vc);
        inView[pi] = in ? 1 : 0;
        if (in) { projX[pi] = u; projXR[pi] = ur; projY[pi] = v; level[pi] = lvl; viewCosOut[pi] = vc; }   // :721-731
    }
    if (pubFlag) orbx_publish(pubCounter, pubFlag, pubSeq, gridDim.x * gridDim.y);      // (host-array call: the outputs are mapped host memory)
}

// Tracking::SearchLocalPoints for ONE frame from host arrays (orbx_search_local_points): the frustum test and the candidate lists in one launch.  One
// wave per map point: every lane evaluates the point's frustum test (the same few dozen operations in all lanes; the point's position / normal /
// distances are read from MAPPED host memory, once per wave), lane 0 stores the mTrack* values for the replay's rescans (device) and for the caller
// (mapped host), then the wave scans the frame's features for the point's TOPK keys as k_proj_topk does.
struct FrustumHostOut { float *px, *py, *pxr, *vc; int32_t *lvl; uint8_t *inView; };
__global__ __launch_bounds__(256) void k_frustum_topk(FrustumDev Fr, MapPointsDev M, ProjFrameDev F, const uint8_t *__restrict__ pdesc, const float *__restrict__ scaleFactors, float th,
                                                      float *__restrict__ dPx, float *__restrict__ dPy, float *__restrict__ dPxr, int32_t *__restrict__ dLvl, float *__restrict__ dVc,
                                                      uint8_t *__restrict__ dIn, FrustumHostOut H, unsigned long long *__restrict__ topk)
{
    __shared__ uint32_t sQueue[4][PT_QUEUE];
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = min(F.counts[0], F.cap), m = min(M.counts[0], M.cap);
    if (i >= m) return;
    unsigned long long *out = topk + (size_t)i * TOPK;
    // everything the wave needs of its point, requested up front (one round trip to memory instead of four: position, distances, normal, descriptor)
    const float *Pp = M.pos + 3 * (size_t)i, *Np = M.normal + 3 * (size_t)i;
    const float Pv[3] = {Pp[0], Pp[1], Pp[2]}, Nv[3] = {Np[0], Np[1], Np[2]};
    const float maxD = M.maxDist[i], minD = M.minDist[i];
    const unsigned long long *dq = (const unsigned long long *)(pdesc + (size_t)i * 32);
    const unsigned long long dq0 = dq[0], dq1 = dq[1], dq2 = dq[2], dq3 = dq[3];
    float u = 0, v = 0, ur = 0, vc = 0;
    int lvl = 0;
    const bool in = frustum_eval(Fr, Fr.tcw, Pv, Nv, maxD, minD, u, v, ur, lvl, vc);
    if (lane == 0) {
        dIn[i] = in ? 1 : 0; H.inView[i] = in ? 1 : 0;
        if (in) {
            dPx[i] = u; dPy[i] = v; dPxr[i] = ur; dLvl[i] = lvl; dVc[i] = vc;
            H.px[i] = u; H.py[i] = v; H.pxr[i] = ur; H.lvl[i] = lvl; H.vc[i] = vc;
        }
    }
    if (!in) { if (lane < TOPK) out[lane] = KEY64_EMPTY; return; }
    ProjQuery q = proj_query_vals(F, u, v, ur, lvl, vc, nullptr, scaleFactors, th);
    q.d[0] = dq0; q.d[1] = dq1; q.d[2] = dq2; q.d[3] = dq3;
    proj_topk_wave(F, 0, n, q, lane, out, sQueue[threadIdx.x >> 6]);
}

// largest float r with ceil(log((double)r) / (double)log_scale_factor) <= k, for k = 0 .. nlevels-2 (host, libm)
extern "C" int orbx_predict_scale_thresholds(float log_scale_factor, int nlevels, float *thresholds)
{
    if (!thresholds || nlevels < 1 || nlevels > ORBX_MAX_LEVELS || !(log_scale_factor > 0.0f)) { orbx_set_error("bad PredictScale arguments"); return ORBX_ERR_ARG; }
    for (int k = 0; k + 1 < nlevels; k++) {
        uint32_t lo = 0x00800000u, hi = 0x7f7fffffu;   // positive normal floats are ordered like their bit patterns
        while (lo < hi) {                               // invariant: f(lo) <= k (ratio 1.2e-38 maps far below 0)
            const uint32_t mid = lo + (hi - lo + 1) / 2;
            float r;
            memcpy(&r, &mid, 4);
            const double n = ceil(log((double)r) / (double)log_scale_factor);
            if (n <= (double)k) lo = mid; else hi = mid - 1;
        }
        memcpy(&thresholds[k], &lo, 4);
    }
    return ORBX_OK;
}

extern "C" int orbx_is_in_frustum_device(orbx_matcher *m, const orbx_frustum_frame *frame, const orbx_map_points *points, float viewing_cos_limit)
{
    if (!m || !frame || !points) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!frame->tcw || !frame->ratio_thresholds || !points->world_pos || !points->normal || !points->max_distance || !points->min_distance || !points->counts) {
        orbx_set_error("NULL array in the frustum arguments");
        return ORBX_ERR_ARG;
    }
    if (frame->nframes < 1 || frame->nframes > m->maxPairs || points->capacity < 1 || frame->nlevels < 1 || frame->nlevels > ORBX_MAX_LEVELS) {
        orbx_set_error("bad frustum sizes");
        return ORBX_ERR_CAPACITY;
    }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    const size_t n = (size_t)frame->nframes * points->capacity;
    int rc;
    if ((rc = m->frProj.ensure(4 * n)) || (rc = m->frLevel.ensure(n)) || (rc = m->frInView.ensure(n))) return rc;
    FrustumDev Fr;
    Fr.tcw = frame->tcw; Fr.fx = frame->fx; Fr.fy = frame->fy; Fr.cx = frame->cx; Fr.cy = frame->cy; Fr.mbf = frame->mbf;
    Fr.minX = frame->min_x; Fr.maxX = frame->max_x; Fr.minY = frame->min_y; Fr.maxY = frame->max_y; Fr.cosLimit = viewing_cos_limit; Fr.nlevels = frame->nlevels;
    for (int k = 0; k < ORBX_MAX_LEVELS; k++) Fr.ratioTh[k] = k + 1 < frame->nlevels ? frame->ratio_thresholds[k] : 0.0f;
    MapPointsDev M = {points->world_pos, points->normal, points->max_distance, points->min_distance, points->counts, points->capacity};
    hipLaunchKernelGGL(k_is_in_frustum, dim3((unsigned)((points->capacity + 255) / 256), (unsigned)frame->nframes), dim3(256), 0, m->stream, Fr, M, m->frProj.p,
                       m->frProj.p + n, m->frProj.p + 2 * n, m->frLevel.p, m->frProj.p + 3 * n, m->frInView.p, (unsigned *)nullptr, (unsigned long long *)nullptr, 0ull);
    MLAUNCH_CHECK();
    m->frCount = n;
    return ORBX_OK;
}

extern "C" int orbx_frustum_results_device(orbx_matcher *m, const float **proj_x, const float **proj_y, const float **proj_xr, const int32_t **scale_level,
                                           const float **view_cos, const uint8_t **in_view)
{
    if (!m) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (!m->frCount) { orbx_set_error("no frustum test has run yet"); return ORBX_ERR_STATE; }
    const size_t n = m->frCount;
    if (proj_x) *proj_x = m->frProj.p;
    if (proj_y) *proj_y = m->frProj.p + n;
    if (proj_xr) *proj_xr = m->frProj.p + 2 * n;
    if (view_cos) *view_cos = m->frProj.p + 3 * n;
    if (scale_level) *scale_level = m->frLevel.p;
    if (in_view) *in_view = m->frInView.p;
    return ORBX_OK;
}

// host-array form for one frame: upload, run, download
extern "C" int orbx_is_in_frustum(orbx_matcher *m, const orbx_frustum_frame *frame_host, const orbx_map_points *points_host, float viewing_cos_limit, float *proj_x,
                                  float *proj_y, float *proj_xr, int32_t *scale_level, float *view_cos, uint8_t *in_view)
{
    if (!m || !frame_host || !points_host || !points_host->counts || !in_view) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    const int mm = points_host->counts[0];
    if (mm <= 0) return ORBX_OK;
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    int rc;
    // no copy engine, no stream synchronisation (OrbxCallBox): every input is read once, where it lies in mapped pinned memory; the kernel writes the
    // mTrack* values into mapped pinned memory and its last workgroup raises the sequence word
    if (!frame_host->tcw || !frame_host->ratio_thresholds || !points_host->world_pos || !points_host->normal || !points_host->max_distance || !points_host->min_distance) {
        orbx_set_error("NULL array in the frustum arguments");
        return ORBX_ERR_ARG;
    }
    if (frame_host->nlevels < 1 || frame_host->nlevels > ORBX_MAX_LEVELS) { orbx_set_error("bad frustum sizes"); return ORBX_ERR_CAPACITY; }
    hipStream_t st = m->stream;
    OrbxCallBox &bx = m->box;
    const size_t n = (size_t)mm;
    const size_t inBytes = bx.padded(64) + 2 * bx.padded(n * 12) + 2 * bx.padded(n * 4) + bx.padded(4);
    const size_t oY = bx.padded(n * 4), oXr = 2 * oY, oVc = 3 * oY, oL = 4 * oY, oIn = oL + bx.padded(n * 4), outBytes = oIn + bx.padded(n);
    if ((rc = bx.begin(inBytes, outBytes, st)) != ORBX_OK) return rc;
    const int32_t cnt = mm;
    FrustumDev Fr;
    Fr.tcw = bx.put(frame_host->tcw, 16);
    Fr.fx = frame_host->fx; Fr.fy = frame_host->fy; Fr.cx = frame_host->cx; Fr.cy = frame_host->cy; Fr.mbf = frame_host->mbf;
    Fr.minX = frame_host->min_x; Fr.maxX = frame_host->max_x; Fr.minY = frame_host->min_y; Fr.maxY = frame_host->max_y; Fr.cosLimit = viewing_cos_limit; Fr.nlevels = frame_host->nlevels;
    for (int k = 0; k < ORBX_MAX_LEVELS; k++) Fr.ratioTh[k] = k + 1 < frame_host->nlevels ? frame_host->ratio_thresholds[k] : 0.0f;
    MapPointsDev M;
    M.pos = bx.put(points_host->world_pos, n * 3); M.normal = bx.put(points_host->normal, n * 3);
    M.maxDist = bx.put(points_host->max_distance, n); M.minDist = bx.put(points_host->min_distance, n);
    M.counts = bx.put(&cnt, 1); M.cap = mm;
    const unsigned long long seq = bx.arm();
    hipLaunchKernelGGL(k_is_in_frustum, dim3((unsigned)((mm + 255) / 256), 1u), dim3(256), 0, st, Fr, M, bx.outDev<float>(0), bx.outDev<float>(oY), bx.outDev<float>(oXr), bx.outDev<int32_t>(oL),
                       bx.outDev<float>(oVc), bx.outDev<uint8_t>(oIn), bx.counter, bx.flagDev, seq);
    MLAUNCH_CHECK();
    if ((rc = bx.wait(st)) != ORBX_OK) return rc;
    memcpy(in_view, bx.outHost<uint8_t>(oIn), n);
    // (the mTrack* values of a point that is not in view are not written by the kernel: the caller's arrays keep what they held, as the reference's members do)
    const float *hx = bx.outHost<float>(0), *hy = bx.outHost<float>(oY), *hxr = bx.outHost<float>(oXr), *hvc = bx.outHost<float>(oVc);
    const int32_t *hl = bx.outHost<int32_t>(oL);
    for (int k = 0; k < mm; k++)
        if (in_view[k]) {
            if (proj_x) proj_x[k] = hx[k];
            if (proj_y) proj_y[k] = hy[k];
            if (proj_xr) proj_xr[k] = hxr[k];
            if (view_cos) view_cos[k] = hvc[k];
            if (scale_level) scale_level[k] = hl[k];
        }
    return ORBX_OK;
}


// Tracking::SearchLocalPoints as one device chain (include/orbx.h): the frustum kernel's outputs are the projection search's inputs.
extern "C" int orbx_search_local_points(orbx_matcher *m, const orbx_projection_frame *fr, const orbx_frustum_frame *pose, const orbx_local_points *pt,
                                        const float *scale_factors, int nlevels, float viewing_cos_limit, float th, float nn_ratio, int32_t *assigned,
                                        int32_t *nmatches, uint8_t *in_view, float *proj_x, float *proj_y, float *proj_xr, int32_t *scale_level, float *view_cos)
{
    if (!m || !fr || !pose || !pt || !assigned || !in_view) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (!fr->counts || !pose->tcw || !pose->ratio_thresholds) { orbx_set_error("NULL array in the frame arguments"); return ORBX_ERR_ARG; }
    const int n = fr->counts[0], mm = pt->count;
    // pose->nlevels bounds the level k_is_in_frustum predicts, nlevels the scale factors k_proj_topk reads for it
    if (!scale_factors || nlevels < 1 || nlevels > ORBX_MAX_LEVELS || pose->nlevels != nlevels) {
        orbx_set_error("scale_factors has %d levels, the frustum arguments name %d (must be equal, 1..%d)", nlevels, pose->nlevels, ORBX_MAX_LEVELS);
        return ORBX_ERR_ARG;
    }
    if (n > 0 && (!fr->keypoints_un || !fr->descriptors || !fr->u_right)) { orbx_set_error("NULL array in the frame arguments"); return ORBX_ERR_ARG; }
    if (nmatches) *nmatches = 0;
    for (int i = 0; i < n; i++) assigned[i] = -1;
    if (mm <= 0) return ORBX_OK;
    if (!pt->world_pos || !pt->normal || !pt->max_distance || !pt->min_distance || !pt->descriptors) { orbx_set_error("NULL array in the point arguments"); return ORBX_ERR_ARG; }
    if (n <= 0) {      // a frame without features: the reference still runs Frame::isInFrustum over the points
        const int32_t cm = mm;
        orbx_map_points mp = {pt->world_pos, pt->normal, pt->max_distance, pt->min_distance, &cm, mm};
        return orbx_is_in_frustum(m, pose, &mp, viewing_cos_limit, proj_x, proj_y, proj_xr, scale_level, view_cos, in_view);
    }
    if (n > m->maxFeatures) { orbx_set_error("%d features exceed the matcher's max_features %d", n, m->maxFeatures); return ORBX_ERR_CAPACITY; }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    // No copy engine, no stream synchronisation (OrbxCallBox).  What EVERY wave of the chain reads - the frame's keypoints, descriptors, uRight, occupied
    // flags, the points' observation flags, the scale factors - is copied once into the device arena by k_stage_copy; what is read once per map point -
    // position, normal, distance range, descriptor - stays in mapped pinned memory and is read there.  k_frustum_topk (frustum test + candidate lists,
    // the mTrack* values to the device for the replay and to the caller's mapped buffer), then k_proj_greedy, which writes assigned[] + the count into
    // the mapped buffer and raises the sequence word.
    int rc;
    hipStream_t st = m->stream;
    OrbxCallBox &bx = m->box;
    const size_t N = (size_t)n, M = (size_t)mm;
    const size_t inBytes = bx.padded(N * sizeof(orbx_keypoint)) + bx.padded(N * 32) + bx.padded(N * 4) + bx.padded(N) + bx.padded(8) + bx.padded(64) + bx.padded(M) + bx.padded(64 * 4) +
                           2 * bx.padded(M * 12) + 2 * bx.padded(M * 4) + bx.padded(M * 32);
    const size_t offPy = bx.padded(M * 4), offPxr = 2 * offPy, offVc = 3 * offPy, offLvl = 4 * offPy, offIn = offLvl + bx.padded(M * 4), offAs = offIn + bx.padded(M),
                 outBytes = offAs + bx.padded((N + 1) * 4);
    if ((rc = bx.begin(inBytes, outBytes, st)) != ORBX_OK) return rc;
    const int32_t cnt[2] = {n, mm};
    // --- staged part (first in the buffer)
    const void *bKp = bx.put(fr->keypoints_un, N), *bDesc = bx.put(fr->descriptors, N * 32), *bUr = bx.put(fr->u_right, N), *bOcc = bx.put(fr->occupied, fr->occupied ? N : 0);
    const void *bCnt = bx.put(cnt, 2), *bTcw = bx.put(pose->tcw, 16), *bObs = bx.put(pt->has_observations, pt->has_observations ? M : 0), *bSc = bx.put(scale_factors, (size_t)nlevels);
    const float *bPos = bx.put(pt->world_pos, M * 3), *bNrm = bx.put(pt->normal, M * 3), *bMax = bx.put(pt->max_distance, M), *bMin = bx.put(pt->min_distance, M);
    const uint8_t *bPd = bx.put(pt->descriptors, M * 32);
    const size_t staged = bx.used;
    if ((rc = m->arena.ensure(staged)) != ORBX_OK) return rc;
    uint8_t *const ar = m->arena.p;
    auto dev = [&](const void *boxAddr) { return ar + ((const uint8_t *)boxAddr - bx.inDev); };
    const float *zPos = (const float *)dev(bPos), *zNrm = (const float *)dev(bNrm), *zMax = (const float *)dev(bMax), *zMin = (const float *)dev(bMin);
    const uint8_t *zDesc = dev(bPd);
    const size_t n16 = staged / 16;
    hipLaunchKernelGGL(k_stage_copy, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 512)), dim3(256), 0, st, (const uint4 *)bx.inDev, (uint4 *)ar, n16);
    MLAUNCH_CHECK();
    if ((rc = m->frProj.ensure(4 * M)) || (rc = m->frLevel.ensure(M)) || (rc = m->frInView.ensure(M)) || (rc = m->topk64.ensure(M * TOPK)) || (rc = m->projDec.ensure(M)) ||
        (rc = m->projQueue.ensure(M))) return rc;
    const int32_t *dCnt = (const int32_t *)dev(bCnt);
    FrustumDev Fr;
    Fr.tcw = (const float *)dev(bTcw); Fr.fx = pose->fx; Fr.fy = pose->fy; Fr.cx = pose->cx; Fr.cy = pose->cy; Fr.mbf = pose->mbf;
    Fr.minX = pose->min_x; Fr.maxX = pose->max_x; Fr.minY = pose->min_y; Fr.maxY = pose->max_y; Fr.cosLimit = viewing_cos_limit; Fr.nlevels = pose->nlevels;
    for (int k = 0; k < ORBX_MAX_LEVELS; k++) Fr.ratioTh[k] = k + 1 < pose->nlevels ? pose->ratio_thresholds[k] : 0.0f;
    MapPointsDev Mp = {zPos, zNrm, zMax, zMin, dCnt + 1, mm};
    ProjFrameDev F = {(const orbx_keypoint *)dev(bKp), dev(bDesc), (const float *)dev(bUr), fr->occupied ? dev(bOcc) : nullptr, dCnt, n, fr->min_x, fr->min_y, fr->grid_width_inv,
                      fr->grid_height_inv};
    const float *dScales = (const float *)dev(bSc);
    FrustumHostOut Ho = {bx.outDev<float>(0), bx.outDev<float>(offPy), bx.outDev<float>(offPxr), bx.outDev<float>(offVc), bx.outDev<int32_t>(offLvl), bx.outDev<uint8_t>(offIn)};
    hipLaunchKernelGGL(k_frustum_topk, dim3((unsigned)((mm + 3) / 4)), dim3(256), 0, st, Fr, Mp, F, zDesc, dScales, th, m->frProj.p, m->frProj.p + M, m->frProj.p + 2 * M, m->frLevel.p,
                       m->frProj.p + 3 * M, m->frInView.p, Ho, m->topk64.p);
    MLAUNCH_CHECK();
    m->frCount = M;
    ProjPointsDev P = {m->frProj.p, m->frProj.p + M, m->frProj.p + 2 * M, m->frLevel.p, m->frProj.p + 3 * M, m->frInView.p, pt->has_observations ? dev(bObs) : nullptr, zDesc, dCnt + 1, mm};
    size_t lds = (size_t)n * 6 + 16;
    if (lds > 160 * 1024) { orbx_set_error("feature count %d too large for the LDS tile of the projection replay", n); return ORBX_ERR_CAPACITY; }
    int decLds = 0;      // dec[mm] + the observation flags behind the feature tile, when they fit (k_proj_greedy)
    if (((lds + 15) & ~(size_t)15) + M * 5 <= 150 * 1024) { decLds = (int)((lds + 15) & ~(size_t)15); lds = (size_t)decLds + M * 5; }
    if (lds > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_proj_greedy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int stride = m->maxFeatures;
    const unsigned long long seq = bx.arm();
    hipLaunchKernelGGL(k_proj_greedy, dim3(1), dim3(PROJ_GREEDY_THREADS), lds, st, F, P, dScales, th, nn_ratio, m->topk64.p, m->matches.p, m->nmatches.p, stride, m->projDec.p,
                       m->projQueue.p, bx.outDev<int32_t>(offAs), bx.flagDev, seq, decLds);
    MLAUNCH_CHECK();
    m->lastPairs = 1; m->lastStride = stride;
    if ((rc = bx.wait(st)) != ORBX_OK) return rc;
    memcpy(in_view, bx.outHost<uint8_t>(offIn), M);
    // (the mTrack* values of a point that is not in view are not written by the kernel: the caller's arrays keep what they held, as with the reference's members)
    const float *hx = bx.outHost<float>(0), *hy = bx.outHost<float>(offPy), *hxr = bx.outHost<float>(offPxr), *hvc = bx.outHost<float>(offVc);
    const int32_t *hl = bx.outHost<int32_t>(offLvl);
    for (int k = 0; k < mm; k++)
        if (in_view[k]) {
            if (proj_x) proj_x[k] = hx[k];
            if (proj_y) proj_y[k] = hy[k];
            if (proj_xr) proj_xr[k] = hxr[k];
            if (view_cos) view_cos[k] = hvc[k];
            if (scale_level) scale_level[k] = hl[k];
        }
    const int32_t *as = bx.outHost<int32_t>(offAs);
    memcpy(assigned, as, N * 4);
    if (nmatches) *nmatches = as[n];
    return ORBX_OK;
}
