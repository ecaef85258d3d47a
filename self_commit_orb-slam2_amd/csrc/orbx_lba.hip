// orbx_lba.hip -- Optimizer::LocalBundleAdjustment's numerical core on gfx950, FP64.
//
// Reference: src/Optimizer.cc:698-958 driving g2o (BlockSolver_6_3, Levenberg-Marquardt,
// Schur complement; Thirdparty/g2o/g2o/core/block_solver.hpp, optimization_algorithm_
// levenberg.cpp, types/types_six_dof_expmap.cpp, types/se3quat.h).  Mapping:
//   k_unpack, k_csr_kf_* / k_csr_pt_*   one upload of the marshalled problem, distributed on the device; adjacency lists (CSR by
//                 keyframe / by landmark, edges ascending) built on the device
//   k_stage_mark / k_stage_index        initializeOptimization(level 0): active edges, free-pose / landmark numbering
//   k_errors      computeActiveErrors + chi2 + Huber rho (one thread per edge, per-workgroup partial sums of the robust chi2)
//   k_lin_sums (+ k_sum_poses_fin)      linearizeOplus + constructQuadraticForm inside the sums that consume them: Hll / b_l per landmark (16 lanes
//                 per landmark) and Hpp / b_p per free pose (four workgroups per keyframe, ordered second pass) as fixed-order sums over
//                 their edges, the Jacobian blocks computed where they are added (no atomics; only Hpl is written per edge, for the Schur
//                 kernels).  k_linearize / k_sum_points / k_sum_poses: the same as three launches with every block written out
//                 (ORBX_LBA_SPLIT=1, measurement switch)
//   k_schur_setup / k_schur_rows / k_schur_fin   S = Hpp + lambda I - sum_l B D^-1 B^T as its lower block triangle, block row per keyframe in LDS,
//                 summed in 64-bit fixed point (order independent: bit-reproducible), its right-hand side in ordered partial sums
//   k_chol_step (one launch per 32-column panel) / k_chol_backsub_reg   dense Cholesky of S + substitutions (n >= 96; k_chol_solve:
//                 one workgroup for the small systems, k_chol_backsub above n = 320)
//   k_backsub_update   x_l = D^-1 (b_l - B^T x_p), push(), oplus: T <- exp(dx) T, X <- X + dx; partial sums of the gain denominator
//   k_lm_begin / k_lm_decide   the Levenberg-Marquardt state machine of optimization_algorithm_levenberg.cpp:61-164 ON THE DEVICE: lambda, rho,
//                 accept / reject, the iteration's and the stage's termination tests; state mirrored into pinned memory, sequence number
//                 last (the host polls it once per optimize(n), not once per trial); every launch of a trial that is not needed returns at once
//   k_restore, k_classify   pop() of a rejected trial (gated by the decision); outlier flags / chi2 / estimates of the result
// The stop flag (g2o's forceStopFlag) is a pinned word the waiting host keeps current.  This path is latency bound (~45 MFLOP per iteration): the deliverable
// is parity (<= 1e-5 vs the reference's own Optimizer.cc + g2o, tests/test_optimizer_ref.py, tests/golden/lba) plus every O(E)
// stage on the device and as few dependent latencies as possible (DESIGN.md section 7 has the measured history).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <type_traits>
#include <cmath>
#include <limits>
#include <vector>

#include "orbx_internal.h"

namespace {

struct DQuat { double x, y, z, w; };
struct DPose { DQuat q; double t[3]; };

// Square roots and quotients of the pose update (SE3Quat::exp, normalizeRotation) sit on one-thread serial sections of the LM loops, where
// every dependent FP64 instruction costs ~16 cycles and an IEEE sqrt / division sequence 25-40 of them: on the device they are v_rsq_f64 /
// v_rcp_f64 + two Newton steps (<= 1 ulp; the estimates are not part of the bit-level contract, 1e-5 vs g2o).  The host keeps libm.
__host__ __device__ __forceinline__ double fast_rcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    y = __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
    return __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
#else
    return 1.0 / x;
#endif
}
// one Newton step (~2^-50: v_rcp_f64 delivers ~26 bits): the pivots of k_pose_opt's 6x6 solve and the gain ratio's denominator, on a one-lane chain where every
// FP64 instruction is ~8 cycles
__device__ __forceinline__ double fast_rcp1(double x)
{
    const double y = __builtin_amdgcn_rcp(x);
    return __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
}
__host__ __device__ __forceinline__ double fast_rsqrt(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    y = __builtin_fma(0.5 * y, __builtin_fma(-x * y, y, 1.0), y);
    return __builtin_fma(0.5 * y, __builtin_fma(-x * y, y, 1.0), y);
#else
    return 1.0 / sqrt(x);
#endif
}
#if defined(__HIP_DEVICE_COMPILE__)
#define ORBX_SQRT(x) ((x) > 0 ? (x) * fast_rsqrt(x) : 0.0)
#else
#define ORBX_SQRT(x) sqrt(x)
#endif
#define ORBX_RCP(x) fast_rcp(x)
#define ORBX_RSQRT(x) fast_rsqrt(x)

__host__ __device__ inline void quat_normalize_pos(DQuat &q)   // SE3Quat::normalizeRotation, se3quat.h:280-285
{
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    const double ni = ORBX_RSQRT(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x *= ni; q.y *= ni; q.z *= ni; q.w *= ni;
}

__host__ __device__ inline DQuat quat_from_R(const double R[9])   // Eigen::Quaterniond(Matrix3d)
{
    DQuat q;
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = ORBX_SQRT(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 * ORBX_RCP(t);
        q.x = (R[7] - R[5]) * t; q.y = (R[2] - R[6]) * t; q.z = (R[3] - R[1]) * t;
    } else {
        // Eigen's branch on the largest diagonal entry, written out per case: runtime indices into R would put the matrix into scratch
        // memory (a global-memory round trip on the one-thread chain of every update)
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > (i ? R[4] : R[0])) i = 2;
        if (i == 0) {          // j = 1, k = 2
            t = ORBX_SQRT(R[0] - R[4] - R[8] + 1.0);
            q.x = 0.5 * t;
            t = 0.5 * ORBX_RCP(t);
            q.w = (R[7] - R[5]) * t; q.y = (R[3] + R[1]) * t; q.z = (R[6] + R[2]) * t;
        } else if (i == 1) {   // j = 2, k = 0
            t = ORBX_SQRT(R[4] - R[8] - R[0] + 1.0);
            q.y = 0.5 * t;
            t = 0.5 * ORBX_RCP(t);
            q.w = (R[2] - R[6]) * t; q.z = (R[7] + R[5]) * t; q.x = (R[1] + R[3]) * t;
        } else {               // j = 0, k = 1
            t = ORBX_SQRT(R[8] - R[0] - R[4] + 1.0);
            q.z = 0.5 * t;
            t = 0.5 * ORBX_RCP(t);
            q.w = (R[3] - R[1]) * t; q.x = (R[2] + R[6]) * t; q.y = (R[5] + R[7]) * t;
        }
    }
    return q;
}

__host__ __device__ inline void quat_to_R(const DQuat &q, double R[9])
{
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

__host__ __device__ inline void quat_rot(const DQuat &q, const double v[3], double o[3])
{
    const double ux = 2 * (q.y * v[2] - q.z * v[1]), uy = 2 * (q.z * v[0] - q.x * v[2]), uz = 2 * (q.x * v[1] - q.y * v[0]);
    o[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
    o[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
    o[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}

__device__ inline DQuat quat_mul(const DQuat &a, const DQuat &b)
{
    DQuat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}

__device__ inline void pose_map(const DPose &T, const double X[3], double o[3])
{
    quat_rot(T.q, X, o);
    o[0] += T.t[0]; o[1] += T.t[1]; o[2] += T.t[2];
}

// SE3Quat::exp (se3quat.h:223-255) and exp(d)*T (types_six_dof_expmap.h:73-76)
__device__ inline void pose_oplus(DPose &T, const double d[6])
{
    const double *om = d, *up = d + 3;
    const double theta = ORBX_SQRT(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9], R[9], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) { R[i] = ((i % 4) == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        double sn, cs;
        sincos(theta, &sn, &cs);      // one argument reduction; theta^3 as two products instead of pow(theta, 3) (a ~150-instruction call on a serial section)
        const double it = ORBX_RCP(theta), it2 = it * it;
        const double a = sn * it, b = (1 - cs) * it2, c = (theta - sn) * (it2 * it);
        for (int i = 0; i < 9; i++) {
            const double I = (i % 4) == 0 ? 1.0 : 0.0;
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    DPose E;
    E.q = quat_from_R(R);
    quat_normalize_pos(E.q);
    for (int i = 0; i < 3; i++) E.t[i] = V[3 * i] * up[0] + V[3 * i + 1] * up[1] + V[3 * i + 2] * up[2];
    double rt[3];
    quat_rot(E.q, T.t, rt);
    DPose N;
    for (int i = 0; i < 3; i++) N.t[i] = E.t[i] + rt[i];
    N.q = quat_mul(E.q, T.q);
    quat_normalize_pos(N.q);
    T = N;
}

// The same update for k_pose_opt, where it sits on a ONE-THREAD section of every Levenberg trial (1460 cycles of a 7100-cycle trial: sqrt, sincos, a
// reciprocal, R, Quaternion(R) with its sqrt and reciprocal, two normalisations - a chain of ~110 dependent FP64 operations).  For |omega|^2 < 0.25 the
// five functions of theta^2 the update needs - sin(t/2)/t, cos(t/2), sin t / t, (1 - cos t) / t^2, (t - sin t) / t^3 - are their Taylor polynomials (seven
// terms: below 1e-17 relative), evaluated side by side, and the rotation's quaternion is (cos(t/2), omega sin(t/2)/t) directly: mathematically the
// Quaternion(R) of se3quat.h:223-255, rounded differently in the last bits (the estimates are not part of the bit-level contract: 1e-5 vs g2o).  Larger
// steps take pose_oplus.
__device__ inline void pose_oplus_small(DPose &T, const double d[6])
{
    const double *om = d, *up = d + 3;
    const double x = __builtin_fma(om[2], om[2], __builtin_fma(om[1], om[1], om[0] * om[0]));
    if (!(x < 0.25)) { pose_oplus(T, d); return; }
    // Horner in x = theta^2; coefficients = 1 / (2^(2k+1) (2k+1)!), 1 / (4^k (2k)!), 1 / (2k+2)!, 1 / (2k+3)! with alternating signs.  The four chains advance
    // together, one fused multiply-add each per step (written as four nested expressions the compiler evaluated them one after the other: +200 cycles); all
    // products of this function are fused multiply-adds - it is a chain of dependent FP64 operations on one lane, ~8 cycles each, that every trial waits for.
    static const double C[4][7] = {{0.5, -1.0 / 48, 1.0 / 3840, -1.0 / 645120, 1.0 / 185794560, -1.0 / 81749606400.0, 1.0 / 51011754393600.0},
                                   {1.0, -1.0 / 8, 1.0 / 384, -1.0 / 46080, 1.0 / 10321920, -1.0 / 3715891200.0, 1.0 / 1961990553600.0},
                                   {0.5, -1.0 / 24, 1.0 / 720, -1.0 / 40320, 1.0 / 3628800, -1.0 / 479001600.0, 1.0 / 87178291200.0},
                                   {1.0 / 6, -1.0 / 120, 1.0 / 5040, -1.0 / 362880, 1.0 / 39916800, -1.0 / 6227020800.0, 1.0 / 1307674368000.0}};
    double pl[4] = {C[0][6], C[1][6], C[2][6], C[3][6]};
#pragma unroll
    for (int k = 5; k >= 0; k--) {
#pragma unroll
        for (int t = 0; t < 4; t++) pl[t] = __builtin_fma(x, pl[t], C[t][k]);
    }
    const double sh = pl[0], ch = pl[1], b = pl[2], c = pl[3];
    // the rotation's quaternion (cos(t/2), omega sin(t/2)/t): a unit quaternion to 1e-16 as it stands (SE3Quat's normalisation of it would multiply by
    // 1 +- 1 ulp - a reciprocal square root on this chain for nothing; the product below is normalised), ch > 0 here: no sign flip
    DQuat e;
    e.x = om[0] * sh; e.y = om[1] * sh; e.z = om[2] * sh; e.w = ch;
    // t = V u, V = I + b Omega + c Omega^2: Omega u = om x u, Omega^2 u = om x (om x u)
    const double c1[3] = {__builtin_fma(om[1], up[2], -(om[2] * up[1])), __builtin_fma(om[2], up[0], -(om[0] * up[2])), __builtin_fma(om[0], up[1], -(om[1] * up[0]))};
    const double c2[3] = {__builtin_fma(om[1], c1[2], -(om[2] * c1[1])), __builtin_fma(om[2], c1[0], -(om[0] * c1[2])), __builtin_fma(om[0], c1[1], -(om[1] * c1[0]))};
    // exp(d) * T: rotation of T.t by e, + t
    const double *v = T.t;
    const double ux = 2 * __builtin_fma(e.y, v[2], -(e.z * v[1])), uy = 2 * __builtin_fma(e.z, v[0], -(e.x * v[2])), uz = 2 * __builtin_fma(e.x, v[1], -(e.y * v[0]));
    DPose N;
    N.t[0] = __builtin_fma(c, c2[0], __builtin_fma(b, c1[0], up[0])) + (__builtin_fma(e.w, ux, v[0]) + __builtin_fma(e.y, uz, -(e.z * uy)));
    N.t[1] = __builtin_fma(c, c2[1], __builtin_fma(b, c1[1], up[1])) + (__builtin_fma(e.w, uy, v[1]) + __builtin_fma(e.z, ux, -(e.x * uz)));
    N.t[2] = __builtin_fma(c, c2[2], __builtin_fma(b, c1[2], up[2])) + (__builtin_fma(e.w, uz, v[2]) + __builtin_fma(e.x, uy, -(e.y * ux)));
    const DQuat &g = T.q;
    N.q.w = __builtin_fma(-e.z, g.z, __builtin_fma(-e.y, g.y, __builtin_fma(-e.x, g.x, e.w * g.w)));
    N.q.x = __builtin_fma(-e.z, g.y, __builtin_fma(e.y, g.z, __builtin_fma(e.x, g.w, e.w * g.x)));
    N.q.y = __builtin_fma(-e.x, g.z, __builtin_fma(e.z, g.x, __builtin_fma(e.y, g.w, e.w * g.y)));
    N.q.z = __builtin_fma(-e.y, g.x, __builtin_fma(e.x, g.y, __builtin_fma(e.z, g.w, e.w * g.z)));
    {
        double ni = fast_rsqrt(__builtin_fma(N.q.w, N.q.w, __builtin_fma(N.q.z, N.q.z, __builtin_fma(N.q.y, N.q.y, N.q.x * N.q.x))));
        ni = N.q.w < 0 ? -ni : ni;      // SE3Quat::normalizeRotation: w >= 0
        N.q.x *= ni; N.q.y *= ni; N.q.z *= ni; N.q.w *= ni;
    }
    T = N;
}

struct LbaDev {   // device views, all sized by the handle
    int K, P, E;
    DPose *pose;
    double *pt;                 // 3P
    const double *intr;         // 5K
    const int *ep, *ek;         // edge -> point / keyframe
    const double *obs;          // 3E
    const uint8_t *stereo;      // E
    const double *info;         // E
    const uint8_t *active;      // E  (level == 0)
    const int *poseIdx;         // K: free-pose index or -1
    const int *ptIdx;           // P: active landmark index or -1
    double *err;                // 3E, _error as last computed
    double *rchi;               // E, (robust) chi2 of active edges, 0 otherwise
    double *edgeBlk;            // 72 per edge in three dense arrays: Hpl(18) | BD(18) | Hll(6) bl(3) Hpp(21) bp(6)  (eb_hpl / eb_bd / eb_rest)
};

// Layout of d.edgeBlk: three dense arrays over the edges, not one record per edge - the Schur kernels gather Hpl of partner edges all over the window
// (60k x ~6 partners per trial); out of 576-byte records that was two or three cache lines for 144 useful bytes and a 34 MB footprint, as an array of its
// own Hpl is 8.6 MB.
//   [0, 18 E)      Hpl, 18 per edge            (eb_hpl)
//   [18 E, 36 E)   B * D^-1 of the current trial (eb_bd; k_schur_setup -> k_schur_rows)
//   [36 E, 72 E)   Hll(6) bl(3) Hpp(21) bp(6), 36 per edge (eb_rest; only written by the launch-per-stage form, ORBX_LBA_SPLIT=1)
#define EB_HLL 0
#define EB_BL 6
#define EB_HPP 9
#define EB_BP 30
#define EB_SIZE 72
__device__ __forceinline__ double *eb_hpl(const LbaDev &d, int e) { return d.edgeBlk + (size_t)e * 18; }
__device__ __forceinline__ double *eb_bd(const LbaDev &d, int e) { return d.edgeBlk + (size_t)d.E * 18 + (size_t)e * 18; }
__device__ __forceinline__ double *eb_rest(const LbaDev &d, int e) { return d.edgeBlk + (size_t)d.E * 36 + (size_t)e * 36; }

// computeError (types_six_dof_expmap.h:90-95, 122-127; cam_project .cpp:141-157)
__device__ inline void edge_error(const LbaDev &d, int e, double out[3], double *depth)
{
    const int k = d.ek[e], l = d.ep[e];
    const double *in = d.intr + 5 * (size_t)k;
    double Xc[3];
    pose_map(d.pose[k], d.pt + 3 * (size_t)l, Xc);
    if (depth) *depth = Xc[2];
    if (!d.stereo[e]) {
        const double u = Xc[0] / Xc[2] * in[0] + in[2], v = Xc[1] / Xc[2] * in[1] + in[3];
        out[0] = d.obs[3 * (size_t)e] - u; out[1] = d.obs[3 * (size_t)e + 1] - v; out[2] = 0;
    } else {
        const float invz = (float)(1.0 / Xc[2]);             // float in the reference (.cpp:151)
        const double u = Xc[0] * invz * in[0] + in[2], v = Xc[1] * invz * in[1] + in[3];
        const float bfz = __fmul_rn((float)in[4], invz);     // bf passed as const float& (.cpp:150)
        out[0] = d.obs[3 * (size_t)e] - u; out[1] = d.obs[3 * (size_t)e + 1] - v; out[2] = d.obs[3 * (size_t)e + 2] - (u - (double)bfz);
    }
}

struct Huber { double dMono, dStereo, dsqrMono, dsqrStereo; };

__device__ inline void huber_rho(const Huber &h, bool stereo, double chi, double &rho0, double &rho1)   // robust_kernel_impl.cpp:78-91
{
    const double delta = stereo ? h.dStereo : h.dMono, dsqr = stereo ? h.dsqrStereo : h.dsqrMono;
    if (chi <= dsqr) { rho0 = chi; rho1 = 1.; }
    else { const double s = sqrt(chi); rho0 = 2 * s * delta - dsqr; rho1 = delta / s; }
}

// sum over the workgroup in a fixed order (lanes by butterfly, the four waves in order); the result is valid in thread 0
__device__ __forceinline__ double block_sum256(double v, double *sw /* 4 */)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
    __syncthreads();
    return sw[0] + sw[1] + sw[2] + sw[3];
}

// partChi[blockIdx.x] = this workgroup's share of activeRobustChi2 (summed in block order by k_lm_begin / k_lm_decide)
__global__ __launch_bounds__(256) void k_errors(LbaDev d, Huber h, int robust, double *partChi, const int *gate, int want)
{
    if (gate && *gate != want) return;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    __shared__ double sw[4];
    const int e = blockIdx.x * 256 + threadIdx.x;
    double r0 = 0;
    if (e < d.E) {
        if (!d.active[e]) d.rchi[e] = 0;   // _error of inactive edges stays as last computed
        else {
            double r[3];
            edge_error(d, e, r, nullptr);
            d.err[3 * (size_t)e] = r[0]; d.err[3 * (size_t)e + 1] = r[1]; d.err[3 * (size_t)e + 2] = r[2];
            const double chi = (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * d.info[e];
            double r1 = 1;
            r0 = chi;
            if (robust) huber_rho(h, d.stereo[e] != 0, chi, r0, r1);
            d.rchi[e] = r0;
        }
    }
    const double t = block_sum256(r0, sw);
    if (threadIdx.x == 0) partChi[blockIdx.x] = t;
}

// linearizeOplus (.cpp:103-139, 188-234): the Jacobians of one active edge (A: 3x3 w.r.t. the landmark, B: 3x6 w.r.t. the pose; row 2 only for a
// stereo edge), the robust weight and -Omega r rho'.  One body for k_linearize and for the fused sums (k_lin_sums), so that both produce the same bits.
__device__ __forceinline__ void edge_linearize(const LbaDev &d, const Huber &h, int robust, int e, int k, int l, bool &stOut, double (&A)[9], double (&B)[18], double &W,
                                               double (&omr)[3])
{
    const bool st = d.stereo[e] != 0;
    const double *in = d.intr + 5 * (size_t)k;
    const double fx = in[0], fy = in[1], bf = in[4];
    double Xc[3], R[9];
    pose_map(d.pose[k], d.pt + 3 * (size_t)l, Xc);
    quat_to_R(d.pose[k].q, R);
    const double x = Xc[0], y = Xc[1], z = Xc[2], z_2 = z * z;
#pragma unroll
    for (int i = 0; i < 9; i++) A[i] = 0;
#pragma unroll
    for (int i = 0; i < 18; i++) B[i] = 0;
    if (!st) {
        const double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) A[3 * i + j] = -1. / z * (tmp[3 * i] * R[j] + tmp[3 * i + 1] * R[3 + j] + tmp[3 * i + 2] * R[6 + j]);
    } else {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            A[j] = -fx * R[j] / z + fx * x * R[6 + j] / z_2;
            A[3 + j] = -fy * R[3 + j] / z + fy * y * R[6 + j] / z_2;
            A[6 + j] = A[j] - bf * R[6 + j] / z_2;
        }
    }
    B[0] = x * y / z_2 * fx; B[1] = -(1 + (x * x / z_2)) * fx; B[2] = y / z * fx; B[3] = -1. / z * fx; B[4] = 0; B[5] = x / z_2 * fx;
    B[6] = (1 + y * y / z_2) * fy; B[7] = -x * y / z_2 * fy; B[8] = -x / z * fy; B[9] = 0; B[10] = -1. / z * fy; B[11] = y / z_2 * fy;
    if (st) { B[12] = B[0] - bf * y / z_2; B[13] = B[1] + bf * x / z_2; B[14] = B[2]; B[15] = B[3]; B[16] = 0; B[17] = B[5] - bf / z_2; }
    const double w = d.info[e];
    const double *r = d.err + 3 * (size_t)e;
    double rw = 1.0;
    if (robust) { double r0; huber_rho(h, st, (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * w, r0, rw); }
    W = rw * w;
#pragma unroll
    for (int i = 0; i < 3; i++) omr[i] = -w * r[i] * rw;
    stOut = st;
}

// sums over the error dimension: rows 0, 1 and - for a stereo edge - 2, in that order from 0 like the reference's loops
// (constructQuadraticForm, base_binary_edge.hpp:55-119); everything unrolled so that A / B are registers
#define LIN_DOT(expr0, expr1, expr2) ([&] { double s_ = 0; s_ += (expr0); s_ += (expr1); if (st) s_ += (expr2); return s_; }())

// linearizeOplus + constructQuadraticForm with every block of every edge written out (the launch-per-stage form; k_lin_sums below computes the
// same blocks inside the sums that consume them)
__global__ __launch_bounds__(256) void k_linearize(LbaDev d, Huber h, int robust, const int *gate, int want)
{
    if (gate && *gate != want) return;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= d.E || !d.active[e]) return;
    const int k = d.ek[e], l = d.ep[e];
    bool st;
    double A[9], B[18], W, omr[3];
    edge_linearize(d, h, robust, e, k, l, st, A, B, W, omr);
    double *blk = eb_rest(d, e), *hpl = eb_hpl(d, e);
    {
        int o = 0;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = i; j < 3; j++) blk[EB_HLL + o++] = LIN_DOT(A[i] * W * A[j], A[3 + i] * W * A[3 + j], A[6 + i] * W * A[6 + j]);
    }
#pragma unroll
    for (int i = 0; i < 3; i++) blk[EB_BL + i] = LIN_DOT(A[i] * omr[0], A[3 + i] * omr[1], A[6 + i] * omr[2]);
    if (d.poseIdx[k] >= 0) {
        int o = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = i; j < 6; j++) blk[EB_HPP + o++] = LIN_DOT(B[i] * W * B[j], B[6 + i] * W * B[6 + j], B[12 + i] * W * B[12 + j]);
#pragma unroll
        for (int i = 0; i < 6; i++) blk[EB_BP + i] = LIN_DOT(B[i] * omr[0], B[6 + i] * omr[1], B[12 + i] * omr[2]);
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) hpl[3 * i + j] = LIN_DOT(B[i] * W * A[j], B[6 + i] * W * A[3 + j], B[12 + i] * W * A[6 + j]);
    }
}

// Hll (9, full symmetric) and b_l (3) of every active landmark: its edges in insertion order
__device__ __forceinline__ double sum16(double x)
{
    // sum over the 16 lanes of a landmark's group (xor butterfly inside the group)
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) x += __shfl_xor(x, o, 16);
    return x;
}

// 16 lanes per landmark (a landmark has ~12 observations: one thread walking them serially leaves the launch at 20 waves and a
// dozen dependent loads deep), partial sums combined by a butterfly inside the group
__global__ __launch_bounds__(256) void k_sum_points(LbaDev d, const int *ptStart, const int *ptEdges, double *Hll, double *bl, const int *gate, int want)
{
    if (gate && *gate != want) return;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    const int l = blockIdx.x * 16 + (threadIdx.x >> 4), a = threadIdx.x & 15;
    if (l >= d.P) return;
    const int li = d.ptIdx[l];
    if (li < 0) return;
    double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};     // H (6) then b (3)
    for (int s = ptStart[l] + a; s < ptStart[l + 1]; s += 16) {
        const int e = ptEdges[s];
        if (!d.active[e]) continue;
        const double *blk = eb_rest(d, e);
#pragma unroll
        for (int i = 0; i < 6; i++) v[i] += blk[EB_HLL + i];
#pragma unroll
        for (int i = 0; i < 3; i++) v[6 + i] += blk[EB_BL + i];
    }
#pragma unroll
    for (int i = 0; i < 9; i++) v[i] = sum16(v[i]);
    if (a == 0) {
        double *o = Hll + (size_t)li * 9;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[1]; o[4] = v[3]; o[5] = v[4]; o[6] = v[2]; o[7] = v[4]; o[8] = v[5];
        bl[(size_t)li * 3] = v[6]; bl[(size_t)li * 3 + 1] = v[7]; bl[(size_t)li * 3 + 2] = v[8];
    }
}

// Hpp (36) and b_p (6) of every free pose.  SP_SPLIT workgroups per keyframe sum a contiguous quarter of its edges each (the 27 values
// go through a transposed LDS tile: thread t stores at [k][t], 216 threads add 32 consecutive entries, 27 threads add the 8 parts - a
// fixed order, two barriers instead of the eight of a 256 -> 1 tree per value); k_sum_poses_fin adds the SP_SPLIT partial results in order.  One workgroup per keyframe walked ~5 edges x 27 gathered doubles per thread in series:
// 16 us with 50 workgroups on a 256-CU device.
#define SP_SPLIT 8        /* most workgroups per keyframe (size of the partial-sum array); the number used per call (spSplit) gives every thread of a
                             workgroup at most ONE edge of the longest keyframe row: a second step for a few threads costs the whole dependent chain again */
#define SP_TP 264
__global__ __launch_bounds__(256) void k_sum_poses(LbaDev d, const int *kfStart, const int *kfEdges, double *part /* K x SP_SPLIT x 27 */, int spSplit, const int *gate, int want)
{
    if (gate && *gate != want) return;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    __shared__ double redT[27 * SP_TP];
    __shared__ double part8[27][9];
    const int k = blockIdx.x, y = blockIdx.y, pi = d.poseIdx[k], tid = threadIdx.x;
    if (pi < 0) return;
    const int s0 = kfStart[k], n = kfStart[k + 1] - s0, per = (n + spSplit - 1) / spSplit;
    const int lo = s0 + min(n, y * per), hi = s0 + min(n, (y + 1) * per);
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; i++) acc[i] = 0;
    for (int s = lo + tid; s < hi; s += 256) {
        const int e = kfEdges[s];
        if (!d.active[e]) continue;
        const double *blk = eb_rest(d, e) + EB_HPP;
#pragma unroll
        for (int i = 0; i < 27; i++) acc[i] += blk[i];
    }
    const int slot = (tid >> 5) * 33 + (tid & 31);
#pragma unroll
    for (int i = 0; i < 27; i++) redT[i * SP_TP + slot] = acc[i];
    __syncthreads();
    if (tid < 216) {
        const int i = tid >> 3, p = tid & 7;
        const double *src = redT + i * SP_TP + p * 33;
        double sacc = 0;
#pragma unroll
        for (int q = 0; q < 32; q++) sacc += src[q];
        part8[i][p] = sacc;
    }
    __syncthreads();
    if (tid < 27) {
        double sacc = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) sacc += part8[tid][q];
        part[((size_t)k * SP_SPLIT + y) * 27 + tid] = sacc;
    }
}
// k_linearize + k_sum_points + k_sum_poses as ONE launch.  The three were 16 + 5.7 + 6.1 us of every LM trial's critical path (the
// speculative linearisation is what the device runs while the host decides about the trial) and moved 26 MB of per-edge blocks through HBM,
// written once and gathered back once.  Here the sums compute the blocks of their edges themselves (edge_linearize: identical expressions,
// identical bits; the pose and intrinsics arrays they need instead are a few KB), in the same order as before; only Hpl, which the Schur
// kernels read per trial, is still written (by the keyframe side, which visits every active edge of a free keyframe exactly once).
//   blocks [0, K * SP_SPLIT)     the keyframe sums (k_sum_poses' body; first: they are the longer ones)
//   the rest                     the landmark sums (k_sum_points' body), 16 lanes per landmark
__global__ __launch_bounds__(256) void k_lin_sums(LbaDev d, Huber h, int robust, const int *ptStart, const int *ptEdges, double *Hll, double *bl, const int *kfStart,
                                                  const int *kfEdges, double *part /* K x SP_SPLIT x 27 */, int spSplit, const int *gate, int want)
{
    if (gate && *gate != want) return;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    __shared__ double redT[27 * SP_TP];
    __shared__ double part8[27][9];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= d.K * spSplit) {
        const int l = ((int)blockIdx.x - d.K * spSplit) * 16 + (tid >> 4), a = tid & 15;
        if (l >= d.P) return;
        const int li = d.ptIdx[l];
        if (li < 0) return;
        double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};     // H (6) then b (3)
        for (int s = ptStart[l] + a; s < ptStart[l + 1]; s += 16) {
            const int e = ptEdges[s];
            if (!d.active[e]) continue;
            bool st;
            double A[9], B[18], W, omr[3];
            edge_linearize(d, h, robust, e, d.ek[e], l, st, A, B, W, omr);
            int o = 0;
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = i; j < 3; j++) v[o++] += LIN_DOT(A[i] * W * A[j], A[3 + i] * W * A[3 + j], A[6 + i] * W * A[6 + j]);
#pragma unroll
            for (int i = 0; i < 3; i++) v[6 + i] += LIN_DOT(A[i] * omr[0], A[3 + i] * omr[1], A[6 + i] * omr[2]);
        }
#pragma unroll
        for (int i = 0; i < 9; i++) v[i] = sum16(v[i]);
        if (a == 0) {
            double *o = Hll + (size_t)li * 9;
            o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[1]; o[4] = v[3]; o[5] = v[4]; o[6] = v[2]; o[7] = v[4]; o[8] = v[5];
            bl[(size_t)li * 3] = v[6]; bl[(size_t)li * 3 + 1] = v[7]; bl[(size_t)li * 3 + 2] = v[8];
        }
        return;
    }
    const int k = (int)blockIdx.x / spSplit, y = (int)blockIdx.x % spSplit, pi = d.poseIdx[k];
    if (pi < 0) return;
    const int s0 = kfStart[k], n = kfStart[k + 1] - s0, per = (n + spSplit - 1) / spSplit;
    const int lo = s0 + min(n, y * per), hi = s0 + min(n, (y + 1) * per);
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; i++) acc[i] = 0;
    for (int s = lo + tid; s < hi; s += 256) {
        const int e = kfEdges[s];
        if (!d.active[e]) continue;
        bool st;
        double A[9], B[18], W, omr[3];
        edge_linearize(d, h, robust, e, k, d.ep[e], st, A, B, W, omr);
        int o = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = i; j < 6; j++) acc[o++] += LIN_DOT(B[i] * W * B[j], B[6 + i] * W * B[6 + j], B[12 + i] * W * B[12 + j]);
#pragma unroll
        for (int i = 0; i < 6; i++) acc[21 + i] += LIN_DOT(B[i] * omr[0], B[6 + i] * omr[1], B[12 + i] * omr[2]);
        double *hpl = eb_hpl(d, e);
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) hpl[3 * i + j] = LIN_DOT(B[i] * W * A[j], B[6 + i] * W * A[3 + j], B[12 + i] * W * A[6 + j]);
    }
    const int slot = (tid >> 5) * 33 + (tid & 31);
#pragma unroll
    for (int i = 0; i < 27; i++) redT[i * SP_TP + slot] = acc[i];
    __syncthreads();
    if (tid < 216) {
        const int i = tid >> 3, p = tid & 7;
        const double *src = redT + i * SP_TP + p * 33;
        double sacc = 0;
#pragma unroll
        for (int q = 0; q < 32; q++) sacc += src[q];
        part8[i][p] = sacc;
    }
    __syncthreads();
    if (tid < 27) {
        double sacc = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) sacc += part8[tid][q];
        part[((size_t)k * SP_SPLIT + y) * 27 + tid] = sacc;
    }
}
#undef LIN_DOT
// ... and the SP_SPLIT partial results added in order by a second small launch (a "last workgroup adds" inside the first one needs a
// device-scope release, i.e. a write-back of the L2 - right after k_linearize has left 34 MB of dirty lines there: 28 us instead of 16)
__global__ __launch_bounds__(256) void k_sum_poses_fin(LbaDev d, const double *__restrict__ part, double *__restrict__ Hpp, double *__restrict__ bp, int spSplit, const int *gate, int want)
{
    if (gate && *gate != want) return;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    const int idx = blockIdx.x * 256 + threadIdx.x, k = idx >> 5, v = idx & 31;
    if (k >= d.K || v >= 27) return;
    const int pi = d.poseIdx[k];
    if (pi < 0) return;
    double t = 0;
    for (int q = 0; q < spSplit; q++) t += part[((size_t)k * SP_SPLIT + q) * 27 + v];
    if (v >= 21) { bp[(size_t)pi * 6 + v - 21] = t; return; }
    // entry v of the upper triangle in row-major order -> (i, j)
    int i = 0, o = v;
    while (o >= 6 - i) { o -= 6 - i; i++; }
    const int j = i + o;
    Hpp[(size_t)pi * 36 + 6 * i + j] = t;
    Hpp[(size_t)pi * 36 + 6 * j + i] = t;
}

// ---------------------------------------------------------------------------------------------
// Levenberg-Marquardt ON THE DEVICE (g2o: optimization_algorithm_levenberg.cpp:61-164, sparse_optimizer.cpp:350-424).  lambda, the
// Nielsen factor, the chi2 bookkeeping and the accept / restore decision of every trial live in LmState; the kernels of a trial take
// lambda from there, k_lm_decide ends a trial (the three sums, rho, accept or reject, the iteration's and the stage's termination tests),
// k_restore runs only behind a rejected trial, and every launch of a trial the stage turns out not to need returns at once (the `gate`
// argument of the kernels).  The host therefore enqueues a whole optimize(n) - n trials, one per iteration, which is what a bundle
// adjustment near its optimum takes - WITHOUT waiting for any of them, then reads the state once; only when a trial was rejected somewhere
// does it add trials one by one.  (Before: one host round trip per trial, ~10-20 us of idle device each, 9 per window.)
// The state is mirrored into pinned memory by every decision, the sequence number last; the stop flag (pbStopFlag) is a pinned word that
// the waiting host keeps equal to the caller's flag.
// ---------------------------------------------------------------------------------------------
struct LmState {
    double lambda, ni, currentChi, iniChi, rho, chi0, tempChi;
    int it, qmax, nBad, ok, done, lastRejected, relin, trials, iters, maxIters;
    unsigned arrive;      // workgroups of k_errors_decide that have stored their share of chi2 (0 between launches)
};

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

__device__ __forceinline__ void lm_mirror(const LmState *st, double *host, double seq, double o1, double o2, double okf)
{
    host[0] = st->tempChi; host[1] = st->currentChi; host[2] = st->lambda; host[3] = st->rho; host[4] = o1; host[5] = o2; host[6] = st->chi0;
    host[7] = (double)st->iters; host[8] = okf; host[9] = (double)st->trials; host[13] = (double)st->done; host[14] = (double)st->it;
    __threadfence_system();
    __hip_atomic_store(host + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// start of optimize(maxIters): chi2 of the start (k_errors' partial sums, in block order), computeLambdaInit (:166-180: tau = 1e-5 times
// the largest diagonal entry, from k_diag_max), counters cleared
__global__ __launch_bounds__(256) void k_lm_begin(LmState *st, const double *partChi, int nChi, const double *diagMax, int maxIters, const volatile int *stopHost, double *host,
                                                  double seq, unsigned long long *scaleBits)
{
    // (measured and not kept, round 6: the maximum of k_diag_max computed here, by these 256 threads - 21.5 us instead of 4.5 + 7.2 for the two launches)
    __shared__ double sw[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 6) scaleBits[tid] = 0ull;      // (the maxima of the Schur scale: raised by the next k_schur_setup)
    double v0 = 0;
    for (int i = tid; i < nChi; i += 256) v0 += partChi[i];
    v0 = wave_sum(v0);
    if (lane == 0) sw[wave] = v0;
    __syncthreads();
    if (tid == 0) {
        const double chi = sw[0] + sw[1] + sw[2] + sw[3];
        st->lambda = 1e-5 * diagMax[2]; st->ni = 2; st->currentChi = chi; st->iniChi = chi; st->chi0 = chi; st->tempChi = chi; st->rho = 0;
        st->it = 0; st->qmax = 0; st->nBad = 0; st->ok = 1; st->lastRejected = 0; st->relin = 0; st->trials = 0; st->iters = 0; st->maxIters = maxIters; st->arrive = 0;
        st->done = (maxIters <= 0 || (stopHost && *stopHost)) ? 1 : 0;
        lm_mirror(st, host, seq, 0.0, 0.0, 1.0);
    }
}

// The decision of one trial (one thread): rho, accept or reject, the iteration's and the stage's termination tests.  -> 1 = rejected.
__device__ __forceinline__ int lm_decide_thread0(LmState *st, double tempChi, double o1, double o2, int okf, bool stop, double *host, double seq)
{
    const double lambda = st->lambda;
    if (!okf) tempChi = 1.7976931348623157e308;      // std::numeric_limits<double>::max(): "solver failed" (:110-113)
    const double scale = o1 + o2 + 1e-3;
    const double rho = (st->currentChi - tempChi) / scale;
    st->tempChi = tempChi; st->rho = rho;
    st->trials += 1;
    int qmax = st->qmax + 1;
    if (rho > 0 && isfinite(tempChi)) {      // accept (:120-131)
        // pow(2 rho - 1, 3) as the host's libm rounds it (correctly rounded in all but knife-edge cases): the cube in double-double
        const double a3 = 2 * rho - 1;
        const double p2 = a3 * a3, e2 = __builtin_fma(a3, a3, -p2), p3 = p2 * a3, e3 = __builtin_fma(p2, a3, -p3);
        double alpha = 1. - (p3 + (e3 + e2 * a3));
        alpha = fmin(alpha, 2. / 3.);
        st->lambda = lambda * fmax(1. / 3., alpha);
        st->ni = 2;
        st->currentChi = tempChi;
        st->lastRejected = 0;
    } else {                                 // reject (:132-141): the estimates come back (below), lambda grows
        st->lambda = lambda * st->ni;
        st->ni *= 2;
        st->lastRejected = 1;
    }
    bool done = false;
    if (!(rho < 0 && qmax < 10 && !stop)) {  // the iteration ends (do-while of :98-146)
        st->iters += 1;
        bool ok = true;
        if (qmax == 10 || rho == 0) ok = false;                      // :148-151 (terminate)
        else {
            if ((st->iniChi - st->currentChi) * 1e3 < st->iniChi) st->nBad += 1; else st->nBad = 0;      // the stall test of SparseOptimizer::optimize as this fork runs it
            if (st->nBad >= 3) ok = false;
        }
        st->ok = ok ? 1 : 0;
        st->it += 1;
        qmax = 0;
        st->iniChi = st->currentChi;
        done = !(st->it < st->maxIters && !stop && ok);
    }
    st->qmax = qmax;
    st->done = done ? 1 : 0;
    st->relin = (!st->lastRejected && !done) ? 1 : 0;      // H and b at the new estimates: only behind an accepted trial that is not the last
    lm_mirror(st, host, seq, o1, o2, (double)okf);
    return st->lastRejected;
}

// end of a trial: partChi / partL = the per-workgroup shares of activeRobustChi2 and of sum xl (lambda xl + bl) (k_errors, k_backsub_update),
// added in block order; xp (lambda xp + bp) summed here; okFlag = the factorisation's.
// (the body of a whole workgroup of 256 threads; `done` stages have been dealt with by the caller)
__device__ __forceinline__ void lm_decide_block(LmState *st, const double *partChi, int nChi, const double *partL, int nL, const double *xp, const double *bp, int nP6,
                                                const int *okFlag, const volatile int *stopHost, double *host, double seq, const LbaDev &d, const DPose *poseBak, const double *ptBak,
                                                unsigned long long *scaleBits)
{
    __shared__ double sw[3][4];
    __shared__ int sRejected;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 6) scaleBits[tid] = 0ull;      // (the maxima of the Schur scale: raised again by the next k_schur_setup)
    const double lambda = st->lambda;
    double v0 = 0, v1 = 0, v2 = 0;
    for (int i = tid; i < nChi; i += 256) v0 += partChi[i];
    for (int i = tid; i < nP6; i += 256) v1 += xp[i] * (lambda * xp[i] + bp[i]);
    for (int i = tid; i < nL; i += 256) v2 += partL[i];
    v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2);
    if (lane == 0) { sw[0][wave] = v0; sw[1][wave] = v1; sw[2][wave] = v2; }
    __syncthreads();
    if (tid == 0) {
        const double tempChi = sw[0][0] + sw[0][1] + sw[0][2] + sw[0][3];
        const double o1 = nP6 > 0 ? sw[1][0] + sw[1][1] + sw[1][2] + sw[1][3] : 0.0, o2 = sw[2][0] + sw[2][1] + sw[2][2] + sw[2][3];
        const int okf = (nP6 > 0 && okFlag) ? *okFlag : 1;
        sRejected = lm_decide_thread0(st, tempChi, o1, o2, okf, stopHost && *stopHost, host, seq);
    }
    __syncthreads();
    // pop() of a rejected trial: the estimates saved by k_backsub_update come back, here (a launch of its own was ~4 us per trial, accepted
    // or not).  H, b and the stored errors are those of the state the trial started from - nothing was linearised since - so the next
    // trial, with its larger lambda, needs nothing else; _error keeps the values of the rejected trial, as in g2o.
    if (sRejected) {
        for (int g = tid; g < d.K; g += 256) d.pose[g] = poseBak[g];
        for (int g = tid; g < 3 * d.P; g += 256) d.pt[g] = ptBak[g];
    }
}

__global__ __launch_bounds__(256) void k_lm_decide(LmState *st, const double *partChi, int nChi, const double *partL, int nL, const double *xp, const double *bp, int nP6,
                                                   const int *okFlag, const volatile int *stopHost, double *host, double seq, LbaDev d, const DPose *poseBak, const double *ptBak,
                                                   unsigned long long *scaleBits)
{
    if (st->done) {      // a trial the stage did not need: only the sequence number moves (the host may be waiting for this very launch)
        if (threadIdx.x < 6) scaleBits[threadIdx.x] = 0ull;
        if (threadIdx.x == 0) { __threadfence_system(); __hip_atomic_store(host + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        return;
    }
    lm_decide_block(st, partChi, nChi, partL, nL, xp, bp, nP6, okFlag, stopHost, host, seq, d, poseBak, ptBak, scaleBits);
}

// k_errors and k_lm_decide of a trial as ONE launch: every workgroup stores its share of activeRobustChi2 (partChi[blockIdx.x], as k_errors does) and
// counts itself in; the workgroup that arrives last takes the decision - it adds the shares in BLOCK order, so the sums do not depend on who arrives
// when - and takes a rejected update back.  One launch and one kernel boundary (~6 us) less per trial.
__global__ __launch_bounds__(256) void k_errors_decide(LbaDev d, Huber h, int robust, double *partChi, LmState *st, const double *partL, int nL, const double *xp, const double *bp,
                                                       int nP6, const int *okFlag, const volatile int *stopHost, double *host, double seq, const DPose *poseBak, const double *ptBak,
                                                       unsigned long long *scaleBits)
{
    __shared__ double swE[4];
    __shared__ int sLast;
    if (st->done) {      // a trial the stage did not need: only the sequence number moves (the host may be waiting for this very launch)
        if (blockIdx.x == 0) {
            if (threadIdx.x < 6) scaleBits[threadIdx.x] = 0ull;
            if (threadIdx.x == 0) { __threadfence_system(); __hip_atomic_store(host + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
        return;
    }
    const int e = blockIdx.x * 256 + threadIdx.x;
    double r0 = 0;
    if (e < d.E) {
        if (!d.active[e]) d.rchi[e] = 0;   // _error of inactive edges stays as last computed
        else {
            double r[3];
            edge_error(d, e, r, nullptr);
            d.err[3 * (size_t)e] = r[0]; d.err[3 * (size_t)e + 1] = r[1]; d.err[3 * (size_t)e + 2] = r[2];
            const double chi = (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * d.info[e];
            double r1 = 1;
            r0 = chi;
            if (robust) huber_rho(h, d.stereo[e] != 0, chi, r0, r1);
            d.rchi[e] = r0;
        }
    }
    const double t = block_sum256(r0, swE);
    if (threadIdx.x == 0) {
        partChi[blockIdx.x] = t;
        __threadfence();      // the share (and this workgroup's _error / chi2 stores) before the arrival
        sLast = atomicAdd(&st->arrive, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!sLast) return;
    __threadfence();          // (acquire side: everybody's shares)
    if (threadIdx.x == 0) st->arrive = 0;
    lm_decide_block(st, partChi, (int)gridDim.x, partL, nL, xp, bp, nP6, okFlag, stopHost, host, seq, d, poseBak, ptBak, scaleBits);
}

// The marshalled inputs of a call arrive as ONE host-to-device copy of the pinned staging buffer; this kernel distributes the segments
// to their device arrays (a dozen hipMemcpyAsync calls and two memsets before: ~5 us of GPU and ~4 us of host time each).
#define UNPACK_MAX 16
struct UnpackSegs {
    unsigned long long src[UNPACK_MAX];   // byte offset in the arena (256-byte aligned)
    void *dst[UNPACK_MAX];                // device array (hipMalloc alignment); nullptr src offset ~0ull = fill with zero
    unsigned long long bytes[UNPACK_MAX]; // kind 0: bytes to copy / clear; kind 1, 2: number of elements
    int kind[UNPACK_MAX];                 // 0: copy / clear; 1: float -> double (the boundary's float arrays travel as floats: 1 MB less per 60k edges on the
                                          // way up, and the host does a memcpy instead of a conversion loop); 2: stereo flag of every observation triplet, !(obs[2] < 0)
    int n;
};
__global__ __launch_bounds__(256) void k_unpack(const uint8_t *arena, UnpackSegs sg)
{
    const size_t stride = (size_t)gridDim.x * 256, t0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < sg.n; i++) {
        if (sg.kind[i] == 1) {
            const float *sf = (const float *)(arena + sg.src[i]);
            double *dd = (double *)sg.dst[i];
            for (size_t w = t0; w < sg.bytes[i]; w += stride) dd[w] = (double)sf[w];
            continue;
        }
        if (sg.kind[i] == 2) {
            const float *sf = (const float *)(arena + sg.src[i]);
            uint8_t *db = (uint8_t *)sg.dst[i];
            for (size_t w = t0; w < sg.bytes[i]; w += stride) db[w] = !(sf[3 * w + 2] < 0);
            continue;
        }
        const size_t nb = sg.bytes[i], nw = nb >> 3;
        const bool zero = sg.src[i] == ~0ull;
        const unsigned long long *s8 = (const unsigned long long *)(arena + (zero ? 0 : sg.src[i]));
        unsigned long long *d8 = (unsigned long long *)sg.dst[i];
        for (size_t w = t0; w < nw; w += stride) d8[w] = zero ? 0ull : s8[w];
        if (t0 < (nb & 7)) ((uint8_t *)sg.dst[i])[8 * nw + t0] = zero ? (uint8_t)0 : (arena + sg.src[i])[8 * nw + t0];
    }
}

// ---------------------------------------------------------------------------------------------
// Adjacency lists (CSR by keyframe and by landmark, edges ascending inside a row - the order a stable host-side counting sort
// gives, so the sums that walk them keep their bits) built ON THE DEVICE from the row starts the host gets for free while it
// marshals the edges: the host-side scatter cost ~100 us per 60k edges, which the device had to wait for after its first kernels.
//   keyframe rows (long, few keys): chunks of 256 edges count their keys (k_csr_kf_count), a scan over the chunks per key gives
//     every chunk its first slot in every row (k_csr_kf_scan), and every edge adds the number of earlier edges of its chunk with
//     the same key (k_csr_kf_fill);
//   landmark rows (short): unordered fill with an atomic slot counter, then every edge counts the smaller edge ids of its row.
// ---------------------------------------------------------------------------------------------
#define CSR_CHUNK 256
__global__ __launch_bounds__(256) void k_csr_kf_count(const int *__restrict__ ek, int E, int K, int *__restrict__ chunkCnt)
{
    extern __shared__ int csrLds[];
    for (int i = threadIdx.x; i < K; i += 256) csrLds[i] = 0;
    __syncthreads();
    const int e = blockIdx.x * CSR_CHUNK + threadIdx.x;
    if (e < E) atomicAdd(&csrLds[ek[e]], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += 256) chunkCnt[(size_t)blockIdx.x * K + i] = csrLds[i];
}

// one workgroup per key: exclusive scan of its counts over the chunks, starting at the row start
__global__ __launch_bounds__(256) void k_csr_kf_scan(const int *__restrict__ kfStart, int K, int nChunk, int *chunkCnt)
{
    __shared__ int wsum[4];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int run = kfStart[k];
    for (int c0 = 0; c0 < nChunk; c0 += 256) {
        const int c = c0 + tid;
        const int v = c < nChunk ? chunkCnt[(size_t)c * K + k] : 0;
        int x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(x, o); if (lane >= o) x += u; }
        __syncthreads();
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int off = 0;
        for (int i = 0; i < w; i++) off += wsum[i];
        if (c < nChunk) chunkCnt[(size_t)c * K + k] = run + off + x - v;
        run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

// slot of edge e = first slot of its chunk in its row + the number of earlier edges of the chunk with the same keyframe (counted by
// walking the chunk's keys in LDS, the same address for every lane)
__global__ __launch_bounds__(256) void k_csr_kf_fill(const int *__restrict__ ek, int E, int K, const int *__restrict__ chunkBase, int *__restrict__ kfEdges)
{
    __shared__ int keys[CSR_CHUNK];
    const int tid = threadIdx.x, e = blockIdx.x * CSR_CHUNK + tid;
    const int key = e < E ? ek[e] : -1;
    keys[tid] = key;
    __syncthreads();
    int rank = 0;
#pragma unroll 8
    for (int j = 0; j < CSR_CHUNK; j++) rank += (j < tid && keys[j] == key) ? 1 : 0;
    if (e < E) kfEdges[chunkBase[(size_t)blockIdx.x * K + key] + rank] = e;
}

__global__ __launch_bounds__(256) void k_csr_pt_fill(const int *__restrict__ ep, int E, const int *__restrict__ ptStart, int *fill, int *__restrict__ ptTmp)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int l = ep[e];
    ptTmp[ptStart[l] + atomicAdd(&fill[l], 1)] = e;
}

__global__ __launch_bounds__(256) void k_csr_pt_rank(const int *__restrict__ ep, int E, const int *__restrict__ ptStart, const int *__restrict__ ptTmp, int *__restrict__ ptEdges)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int l = ep[e], s0 = ptStart[l], m = ptStart[l + 1] - s0;
    int rank = 0;
    for (int i = 0; i < m; i++) rank += ptTmp[s0 + i] < e;
    ptEdges[s0 + rank] = e;
}

// What k_schur_rows needs per slot of the keyframe lists without chasing five dependent indices (edge -> landmark -> row -> partner edge ->
// keyframe -> free-pose index, ~1.5 us each on a device that the launch does not fill): the landmark row of every keyframe-list slot
// (static per call) and, per stage, the free-pose index of the edge in every landmark-list slot (-1: inactive edge or fixed keyframe).
__global__ __launch_bounds__(256) void k_csr_rows(int E, const int *__restrict__ ep, const int *__restrict__ kfEdges, const int *__restrict__ ptStart, int *__restrict__ kfRowS0,
                                                  int *__restrict__ kfRowN)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= E) return;
    const int l = ep[kfEdges[s]], s0 = ptStart[l];
    kfRowS0[s] = s0;
    kfRowN[s] = ptStart[l + 1] - s0;
}
__global__ __launch_bounds__(256) void k_stage_pairs(int E, const int *__restrict__ ek, const int *__restrict__ ptEdges, const uint8_t *__restrict__ active,
                                                     const int *__restrict__ poseIdx, int *__restrict__ ptPi)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= E) return;
    const int e2 = ptEdges[j];
    ptPi[j] = active[e2] ? poseIdx[ek[e2]] : -1;
}

// initializeOptimization(level 0) on the device (sparse_optimizer.cpp:166-267): active[e] = edge not flagged as an outlier by the previous
// stage (flag == nullptr: all), a vertex takes part when one of its edges does, free keyframes and landmarks are numbered in order.
// The three counts go to pinned memory like the sums of a trial (host[10..12], sequence number last).  Before, the
// flags went to the host, through three O(E) loops and back: ~120 us of idle device between the two stages of a local BA.
__device__ __forceinline__ int block_incl_scan1024(int v, int *wsum /* 17 */, int *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(x, o); if (lane >= o) x += u; }
    __syncthreads();                       // wsum may still be read from the previous call
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int off = 0, tot = 0;
    for (int i = 0; i < 16; i++) { const int t = wsum[i]; if (i < w) off += t; tot += t; }
    *total = tot;
    return x + off;
}
// (a) wide: active flags, vertex marks (stamped with the stage number, so that nothing has to be cleared between the stages), per-workgroup counts
__global__ __launch_bounds__(256) void k_stage_mark(int E, const int *__restrict__ ep, const int *__restrict__ ek, const uint8_t *__restrict__ flag, uint8_t *__restrict__ active,
                                                    int *pAct, int *lAct, int stamp, int *__restrict__ partCnt)
{
    __shared__ int sw[4];
    const int e = blockIdx.x * 256 + threadIdx.x;
    int a = 0;
    if (e < E) {
        a = flag ? flag[e] == 0 : 1;
        active[e] = (uint8_t)a;
        if (a) { pAct[ek[e]] = stamp; lAct[ep[e]] = stamp; }      // (concurrent stores of the same value)
    }
    const int n = __popcll(__ballot(a != 0));
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) partCnt[blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}
// (b) one workgroup: numbering of the free keyframes and the active landmarks, the counts to the host
__global__ __launch_bounds__(1024) void k_stage_index(int K, int P, const uint8_t *__restrict__ fixed, const int *__restrict__ pAct, const int *__restrict__ lAct, int stamp,
                                                      int *__restrict__ poseIdx, int *__restrict__ ptIdx, const int *__restrict__ partCnt, int nPart, double *host, double seq)
{
    __shared__ int wsum[17];
    const int tid = threadIdx.x;
    int nPose = 0, nPt = 0, tot;
    for (int k0 = 0; k0 < K; k0 += 1024) {
        const int k = k0 + tid;
        const int f = (k < K && pAct[k] == stamp && !fixed[k]) ? 1 : 0;
        const int inc = block_incl_scan1024(f, wsum, &tot);
        if (k < K) poseIdx[k] = f ? nPose + inc - 1 : -1;
        nPose += tot;
    }
    for (int l0 = 0; l0 < P; l0 += 1024) {
        const int l = l0 + tid;
        const int f = (l < P && lAct[l] == stamp) ? 1 : 0;
        const int inc = block_incl_scan1024(f, wsum, &tot);
        if (l < P) ptIdx[l] = f ? nPt + inc - 1 : -1;
        nPt += tot;
    }
    int nAct = 0;
    for (int i = tid; i < nPart; i += 1024) nAct += partCnt[i];
    block_incl_scan1024(nAct, wsum, &tot);
    if (tid == 0) {
        host[10] = (double)nPose; host[11] = (double)nPt; host[12] = (double)tot;
        __threadfence_system();
        __hip_atomic_store(host + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// pop() of a rejected trial: the estimates saved by k_backsub_update come back (one launch instead of two copies)
__global__ __launch_bounds__(256) void k_restore(LbaDev d, const DPose *poseBak, const double *ptBak, const int *gate, int want)
{
    if (gate && *gate != want) return;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < d.K) d.pose[g] = poseBak[g];
    if (g < 3 * d.P) d.pt[g] = ptBak[g];
}

// computeLambdaInit (optimization_algorithm_levenberg.cpp:166-180): out[2] = max |diagonal entry| over the pose and landmark blocks
__global__ __launch_bounds__(1024) void k_diag_max(const double *Hpp, int nPose, const double *Hll, int nPt, double *out)
{
    __shared__ double m[1024];
    double b = 0;
    for (int i = threadIdx.x; i < 6 * nPose; i += 1024) b = fmax(b, fabs(Hpp[(size_t)(i / 6) * 36 + 7 * (i % 6)]));
    for (int i = threadIdx.x; i < 3 * nPt; i += 1024) b = fmax(b, fabs(Hll[(size_t)(i / 3) * 9 + 4 * (i % 3)]));
    m[threadIdx.x] = b;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) { if ((int)threadIdx.x < k) m[threadIdx.x] = fmax(m[threadIdx.x], m[threadIdx.x + k]); __syncthreads(); }
    if (threadIdx.x == 0) out[2] = m[0];
}

// S = blockdiag(Hpp) + lambda*I ; bs = bp   (setLambda + "_Hpp->add(_Hschur)", block_solver.hpp:363-365, 564-589)
// e->chi2() from the stored _error and isDepthPositive() from the CURRENT estimates (src/Optimizer.cc:880-958): flag = outlier
// poseOut / ptOut (final call): the estimates copied next to the flags, so that ONE device-to-host copy brings everything back
__global__ __launch_bounds__(256) void k_classify(LbaDev d, uint8_t *flag, double *chiOut, DPose *poseOut, double *ptOut)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (poseOut && e < d.K) poseOut[e] = d.pose[e];
    if (ptOut && e < 3 * d.P) ptOut[e] = d.pt[e];
    if (e >= d.E) return;
    const double *r = d.err + 3 * (size_t)e;
    const double chi = (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * d.info[e];
    const DPose T = d.pose[d.ek[e]];
    const double X[3] = {d.pt[3 * (size_t)d.ep[e]], d.pt[3 * (size_t)d.ep[e] + 1], d.pt[3 * (size_t)d.ep[e] + 2]};
    double Xc[3];
    quat_rot(T.q, X, Xc);
    const double z = Xc[2] + T.t[2];
    const double th = d.stereo[e] ? 7.815 : 5.991;
    flag[e] = (chi > th || !(z > 0.0)) ? 1 : 0;
    if (chiOut) chiOut[e] = chi;
}

// per landmark: D^-1, D^-1 b_l and B D^-1 of its edges
// (block_solver.hpp:381-439).  One wave per landmark, lanes over (edge, row).
__device__ __forceinline__ void schur_points_part(int block, const LbaDev &d, const int *ptStart, const int *ptEdges, const double *Hll, const double *bl, double lambda,
                                                  double *Dinv, double *Ddb)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l = block * 4 + wv;
    if (l >= d.P) return;
    const int li = d.ptIdx[l];
    if (li < 0) return;
    double M[9], I[9];
#pragma unroll
    for (int i = 0; i < 9; i++) M[i] = Hll[(size_t)li * 9 + i];
    M[0] += lambda; M[4] += lambda; M[8] += lambda;
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02, id = 1.0 / det;
    I[0] = c00 * id; I[1] = (M[2] * M[7] - M[1] * M[8]) * id; I[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    I[3] = c01 * id; I[4] = (M[0] * M[8] - M[2] * M[6]) * id; I[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    I[6] = c02 * id; I[7] = (M[1] * M[6] - M[0] * M[7]) * id; I[8] = (M[0] * M[4] - M[1] * M[3]) * id;
    double db[3];
#pragma unroll
    for (int i = 0; i < 3; i++) db[i] = I[3 * i] * bl[(size_t)li * 3] + I[3 * i + 1] * bl[(size_t)li * 3 + 1] + I[3 * i + 2] * bl[(size_t)li * 3 + 2];
    if (lane < 9) {
        double v = I[0];
#pragma unroll
        for (int i = 1; i < 9; i++) v = lane == i ? I[i] : v;
        Dinv[(size_t)li * 9 + lane] = v;
    }
    if (lane < 3) Ddb[(size_t)l * 3 + lane] = lane == 0 ? db[0] : (lane == 1 ? db[1] : db[2]);   // D^-1 b_l, applied to bs by k_schur_rows
}

// BD = B * D^-1 (6x3 per active edge of a free keyframe), one thread per edge.  The inverse of the landmark's damped 3x3 block is recomputed from
// Hll here (9 loads, ~40 operations) instead of being taken from the landmark's wave: a wave walking the ~12 edges of its landmark, six rows each,
// was two steps of four dependent loads (slot -> edge -> keyframe -> free-pose index -> block) behind the inverse; per edge everything but
// edge -> landmark -> active index -> Hll is independent.
__device__ __forceinline__ void schur_edges_part(int block, const LbaDev &d, const double *Hll, double lambda)
{
    const int e = block * 256 + threadIdx.x;
    if (e >= d.E || !d.active[e]) return;
    if (d.poseIdx[d.ek[e]] < 0) return;
    const int li = d.ptIdx[d.ep[e]];
    if (li < 0) return;
    double M[9], I[9], B[18];
    const double *hp = eb_hpl(d, e);
#pragma unroll
    for (int i = 0; i < 18; i++) B[i] = hp[i];
#pragma unroll
    for (int i = 0; i < 9; i++) M[i] = Hll[(size_t)li * 9 + i];
    M[0] += lambda; M[4] += lambda; M[8] += lambda;
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02, id = 1.0 / det;
    I[0] = c00 * id; I[1] = (M[2] * M[7] - M[1] * M[8]) * id; I[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    I[3] = c01 * id; I[4] = (M[0] * M[8] - M[2] * M[6]) * id; I[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    I[6] = c02 * id; I[7] = (M[1] * M[6] - M[0] * M[7]) * id; I[8] = (M[0] * M[4] - M[1] * M[3]) * id;
    double *bd = eb_bd(d, e);
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) bd[3 * r + c] = B[3 * r] * I[c] + B[3 * r + 1] * I[3 + c] + B[3 * r + 2] * I[6 + c];
}

// Fixed-point scale of the Schur accumulation (k_schur_rows): q[r] = smallest integer with sqrt(max_i Hpp_i[r][r]) <= 2^q[r], r = 0..5.
// Every term (B_1 D^-1 B_2^T)[r][c] of every observation pair is bounded by sqrt(Hpp_1[r][r] Hpp_2[c][c]) <= 2^(q[r] + q[c]): D^-1 is positive
// definite (Cauchy-Schwarz), and B D^-1 B^T of one edge is at most that edge's own share of Hpp (the landmark's block contains the edge's).
// The maxima are kept as the bit patterns of non-negative doubles (integer order = numeric order; a maximum does not depend on the order
// it is taken in): scaleBits[r], cleared by k_lm_begin / k_lm_decide, raised here by the workgroups that produce Hpp.
__device__ __forceinline__ int schur_q(unsigned long long bits)
{
    const double h = __longlong_as_double((long long)bits);
    if (!(h > 0) || !(h < 1.7e308)) return 0;
    const int e = ilogb(h);
    return (e + 2) >> 1;      // = ceil((e + 1) / 2): 2^(2q) >= 2^(e + 1) > h
}

// The ordered add of the keyframe partial sums (= k_sum_poses_fin, which the start of a stage still launches on its own for k_diag_max)
// as the first workgroups of k_schur_setup: Hpp / b_p of every free pose from the SP_SPLIT partial results, and the scale maxima.
__device__ __forceinline__ void schur_poses_part(int block, const LbaDev &d, const double *__restrict__ part, int spSplit, double *__restrict__ Hpp, double *__restrict__ bp,
                                                 unsigned long long *scaleBits)
{
    const int idx = block * 256 + threadIdx.x, k = idx >> 5, v = idx & 31;
    if (k >= d.K || v >= 27) return;
    const int pi = d.poseIdx[k];
    if (pi < 0) return;
    double t = 0;
    for (int q = 0; q < spSplit; q++) t += part[((size_t)k * SP_SPLIT + q) * 27 + v];
    if (v >= 21) { bp[(size_t)pi * 6 + v - 21] = t; return; }
    int i = 0, o = v;
    while (o >= 6 - i) { o -= 6 - i; i++; }
    const int j = i + o;
    Hpp[(size_t)pi * 36 + 6 * i + j] = t;
    Hpp[(size_t)pi * 36 + 6 * j + i] = t;
    if (i == j) atomicMax(&scaleBits[i], (unsigned long long)__double_as_longlong(fabs(t)));
}

__global__ __launch_bounds__(256) void k_schur_setup(LbaDev d, int nInit, const double *Hpp, const double *bp, int nPose, const int *ptStart, const int *ptEdges,
                                                     const double *Hll, const double *bl, double lambda, double *S, double *bs, double *Dinv, double *Ddb, int *okFlag, const double *lamSrc,
                                                     unsigned long long *scaleBits, const double *spPart, int spSplit, double *HppOut, double *bpOut, const int *gate, int want)
{
    if (gate && *gate != want) return;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    if (lamSrc) lambda = *lamSrc;      // the stage's lambda lives on the device (LmState, k_lm_decide)
    if (blockIdx.x == 0 && threadIdx.x == 0) *okFlag = 1;      // the factorisation clears it at a failed pivot
    const int nPtB = (d.P + 3) / 4;
    if ((int)blockIdx.x < nInit) schur_poses_part((int)blockIdx.x, d, spPart, spSplit, HppOut, bpOut, scaleBits);      // (S and bs themselves are produced by k_schur_fin)
    else if ((int)blockIdx.x < nInit + nPtB) schur_points_part((int)blockIdx.x - nInit, d, ptStart, ptEdges, Hll, bl, lambda, Dinv, Ddb);
    else schur_edges_part((int)blockIdx.x - nInit - nPtB, d, Hll, lambda);
}

// S[i1, i2] -= (B_1 D^-1) B_2^T for every pair of free-pose observations of a landmark, i2 <= i1 (the LOWER block
// triangle, which is what the Cholesky kernels read: no mirroring pass).  Block row i1 of S belongs to keyframe i1: gridDim.y workgroups per keyframe
// walk its edges (16 lanes per edge, one lane per second observation) and accumulate the row in LDS, then add it to the global accumulator once.
// BIT-REPRODUCIBLE: the sums are taken in 64-bit FIXED POINT - integer addition is associative, so neither the order in which the waves of
// a workgroup reach an LDS cell nor the order in which the workgroups reach a global one can change a bit (the former ds_add_f64 / FP64
// global atomics made S, the step and - with rho near 0 - the trial count vary from run to run).  A term t of block entry [r][c] is rounded
// to a multiple of 2^(q[r] + q[c] - 49) (k_schur_setup: |t| <= 2^(q[r] + q[c])) by ONE addition, MAGIC - t with MAGIC = 1.5 * 2^52, whose
// mantissa then holds the integer; up to 2^12 terms per cell fit the 64-bit sum.  Resolution: 2^-49 of the bound, i.e. double precision
// relative to sqrt(Hpp[r][r] Hpp[c][c]) of the best-observed keyframe.  k_schur_fin turns the sums into S = Hpp + lambda I - (...) .
// The right-hand side bs[i1] -= B (D^-1 b_l) has no such bound at hand: every lane sums its own edges in their fixed order, the 16 lane
// groups of a workgroup are added in order, and k_schur_fin adds the workgroups' partial results in order.
// GLOBAL = true: the block row does not fit into LDS (more than ~530 free keyframes, i.e. a global bundle adjustment of a large map):
// the same walk with the integer atomics going straight to the global accumulator.
#define SCHUR_MAGIC 6755399441055744.0          /* 1.5 * 2^52 */
#define SCHUR_MAGIC_BITS 0x4338000000000000ll
template <bool GLOBAL>
__global__ __launch_bounds__(256, 6) void k_schur_rows(LbaDev d, const int *kfStart, const int *kfEdges, const int *__restrict__ kfRowS0, const int *__restrict__ kfRowN,
                                                    const int *ptEdges, const int *__restrict__ ptPi, int nP6, const double *__restrict__ Ddb, unsigned long long *Sacc,
                                                    double *bsPart, const unsigned long long *__restrict__ scaleBits, const int *gate, int want)
{
    if (gate && *gate != want) return;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    extern __shared__ __attribute__((aligned(16))) unsigned long long rowLds[];   // [6][nP6] fixed-point sums
    __shared__ double bsRed[16][6];
    const int k = blockIdx.x, pi = d.poseIdx[k], tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (pi < 0) return;
    unsigned long long *row = GLOBAL ? Sacc + (size_t)(6 * pi) * nP6 : rowLds;      // GLOBAL: "row" is the block row of the accumulator itself
    if (!GLOBAL) {
        for (int i = tid; i < 6 * nP6; i += 256) rowLds[i] = 0ull;
        __syncthreads();
    }
    // 2^(24 - q[r]) for the rows of B_1 D^-1 and 2^(25 - q[c]) for the rows of B_2: the products come out in units of 2^(q[r] + q[c] - 49)
    // (applied with v_ldexp_f64 from six scalar exponents: twelve scale factors in vector registers made the kernel spill at its 80)
    int qa[6], qb[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { const int q = __builtin_amdgcn_readfirstlane(schur_q(scaleBits[i])); qa[i] = 24 - q; qb[i] = 25 - q; }
    // a wave works on four edges of the keyframe at once: 16 lanes per edge, one lane per second observation of the
    // landmark (the index lookups are done once per pair, the 6x6 block comes out of 36 registers)
    const int sub = lane >> 4, a = lane & 15, stride = 16 * gridDim.y;
    double accB = 0.0;      // lane a < 6: row a of this lane group's share of bs[i1]
    for (int s = kfStart[k] + (blockIdx.y * 4 + wv) * 4 + sub; s < kfStart[k + 1]; s += stride) {
        // three dependent round trips instead of eight: slot -> (edge | landmark row) -> (partner edge, its free-pose index) -> its block
        const int e = kfEdges[s], s0 = kfRowS0[s], nE = kfRowN[s];
        int e2v[1], i2v[1];                                   // first partner of this lane, requested before anything else is waited for
        e2v[0] = a < nE ? ptEdges[s0 + a] : 0; i2v[0] = a < nE ? ptPi[s0 + a] : -1;
        if (!d.active[e]) continue;
        const double *pBD = eb_bd(d, e);
        double BD[18];
#pragma unroll
        for (int i = 0; i < 18; i++) BD[i] = ldexp(pBD[i], qa[i / 3]);      // (a power of two: exact)
        if (a < 6) {   // bs[i1] -= B * (D^-1 b_l)
            const double *B1 = eb_hpl(d, e) + 3 * a, *db = Ddb + (size_t)d.ep[e] * 3;
            accB -= B1[0] * db[0] + B1[1] * db[1] + B1[2] * db[2];
        }
        for (int a2 = a; a2 < nE; a2 += 16) {
            const int e2 = a2 == a ? e2v[0] : ptEdges[s0 + a2];
            const int i2 = a2 == a ? i2v[0] : ptPi[s0 + a2];
            if (i2 < 0 || i2 > pi) continue;            // lower block triangle only; -1 = inactive edge or fixed keyframe
            // the partner's block one row (= one column of the 6x6 result) at a time: with all 18 values of it next to the 18 of BD the kernel needs 110
            // registers, four waves per SIMD, and its 1600 workgroups run in two rounds of ~14 us (s_memrealtime: the workgroups of the last y-splits start
            // when the first ones end); the loop itself is three dependent memory round trips per step, i.e. it wants residency, not registers
            const double *pB2 = eb_hpl(d, e2);
            unsigned long long *dst = row + 6 * i2;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const double b0 = ldexp(pB2[3 * c], qb[c]), b1 = ldexp(pB2[3 * c + 1], qb[c]), b2 = ldexp(pB2[3 * c + 2], qb[c]);
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    const double t = SCHUR_MAGIC - (BD[3 * r] * b0 + BD[3 * r + 1] * b1 + BD[3 * r + 2] * b2);      // the rounding to the fixed-point grid
                    atomicAdd(&dst[(size_t)r * nP6 + c], (unsigned long long)(__double_as_longlong(t) - SCHUR_MAGIC_BITS));   // ds_add_u64
                }
            }
        }
    }
    // the lane groups' shares of bs[i1], added in group order
    if (a < 6) bsRed[wv * 4 + sub][a] = accB;
    __syncthreads();
    if (tid < 6) {
        double t = 0.0;
#pragma unroll
        for (int gq = 0; gq < 16; gq++) t += bsRed[gq][tid];
        bsPart[((size_t)k * gridDim.y + blockIdx.y) * 6 + tid] = t;
    }
    if (GLOBAL) return;
    for (int i = tid; i < 6 * nP6; i += 256) {
        const int r = i / nP6, col = i - r * nP6;
        const unsigned long long v = rowLds[i];
        if (col < 6 * pi + 6 && v != 0ull) atomicAdd(&Sacc[(size_t)(6 * pi + r) * nP6 + col], v);
    }
}

// S = blockdiag(Hpp) + lambda I - (fixed-point sums, converted and cleared for the next trial); bs = bp + the workgroups' partial sums in order
// ("_Hpp->add(_Hschur)", block_solver.hpp:363-365, 564-589).  One launch behind k_schur_rows.
__global__ __launch_bounds__(256) void k_schur_fin(LbaDev d, const double *Hpp, const double *bp, int nPose, double lambda, const double *lamSrc, unsigned long long *Sacc,
                                                   const double *bsPart, int ySplit, const unsigned long long *__restrict__ scaleBits, double *S, double *bs, const int *gate, int want)
{
    if (gate && *gate != want) return;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    if (lamSrc) lambda = *lamSrc;
    const int n = 6 * nPose;
    // blockIdx.y = block row (6 rows of S), blockIdx.x * 256 + thread = column: no divisions, coalesced rows
    const int c = blockIdx.x * 256 + threadIdx.x, br = blockIdx.y;
    if (br < nPose && c < n) {
        const int cm = c % 6, qc = schur_q(scaleBits[cm]);
        const bool diagBlock = c / 6 == br;
#pragma unroll
        for (int rr = 0; rr < 6; rr++) {
            const size_t idx = (size_t)(6 * br + rr) * n + c;
            double v = 0;
            if (diagBlock) { v = Hpp[(size_t)br * 36 + 6 * rr + cm]; if (rr == cm) v += lambda; }
            const long long acc = (long long)Sacc[idx];
            if (acc != 0) { v += ldexp((double)acc, schur_q(scaleBits[rr]) + qc - 49); Sacc[idx] = 0ull; }
            S[idx] = v;
        }
    }
    if (blockIdx.y == gridDim.y - 1 && blockIdx.x == 0)      // (one extra block row: the right-hand side)
        for (int i = threadIdx.x; i < 6 * d.K; i += 256) {
            const int k = i / 6, r = i - 6 * k, pi = d.poseIdx[k];
            if (pi < 0) continue;
            // all partial sums requested at once (clamped addresses, unconditional loads), then added in order: a load per step of the
            // dependent chain was 30 L2 round trips = 18 us for this one workgroup
            double pv[32];
#pragma unroll
            for (int y = 0; y < 32; y++) pv[y] = bsPart[((size_t)k * ySplit + min(y, ySplit - 1)) * 6 + r];
            double t = bp[6 * pi + r];
#pragma unroll
            for (int y = 0; y < 32; y++) t += y < ySplit ? pv[y] : 0.0;
            bs[6 * pi + r] = t;
        }
}

#define CHOL_MAX_N 128          /* k_chol_solve serves n < CHOL_MULTI_MIN_N only */
#define CHOL_LDS_X 2048         /* k_chol_backsub keeps the solution vector in LDS up to this n, in global memory above */
#define CHOL_DENSE_MAX_N 24576  /* 4096 free keyframes: S and L are n x n doubles each (4.8 GB at the limit), indices stay below 2^31 */
// Dense Cholesky of the symmetric S (lower triangle read), then S x = bs.  One workgroup.
// LinearSolverEigen::solve (solvers/linear_solver_eigen.h:94-125) uses a sparse LDLT; the
// reduced system is SPD here (lambda > 0), a failed pivot reports ok = 0 like info()!=Success.
// Blocked right-looking Cholesky S = L L^T + the two triangular solves, one workgroup.  A panel of NB
// columns (all rows below the diagonal block) lives in LDS while it is factorised (two LDS barriers per
// column instead of three global round trips), is written back once, and updates the trailing matrix
// in one parallel sweep; the substitutions reuse the same panels.  n <= CHOL_MAX_N, NB*n*8 bytes of LDS.
template <int NB>
__global__ __launch_bounds__(1024) void k_chol_solve(double *S, const double *bs, int n, double *x, int *okFlag, const int *gate, int want)
{
    if (gate && *gate != want) return;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    extern __shared__ __attribute__((aligned(16))) double panel[];   // [rows][NB], rows = n - p0
    __shared__ double sx[CHOL_MAX_N];
    __shared__ int sFail;
    const int tid = threadIdx.x;
    if (tid == 0) sFail = 0;
    for (int i = tid; i < n; i += 1024) sx[i] = bs[i];
    __syncthreads();
    for (int p0 = 0; p0 < n; p0 += NB) {
        const int nb = min(NB, n - p0), rows = n - p0;
        for (int idx = tid; idx < rows * nb; idx += 1024) { const int r = idx / nb, c = idx - r * nb; panel[r * NB + c] = S[(size_t)(p0 + r) * n + p0 + c]; }
        __syncthreads();
        // (1) diagonal block L11 by ONE wave.  Full panel: lane r keeps row r in registers and the column
        //     entries travel by __shfl (no LDS round trip, no barrier); partial last panel: column by column in LDS.
        if (tid < 64) {
            if (nb == NB) {
                const int r = tid & (NB - 1);   // lanes >= NB mirror a row and write nothing
                double a[NB];
#pragma unroll
                for (int c = 0; c < NB; c++) a[c] = panel[r * NB + c];
                bool bad = false;
#pragma unroll
                for (int c = 0; c < NB; c++) {
                    const double dj = __shfl(a[c], c);
                    if (!(dj > 0) || !isfinite(dj)) bad = true;   // wave-uniform
                    const double ljj = sqrt(dj);
                    a[c] = r > c ? a[c] / ljj : (r == c ? ljj : a[c]);
#pragma unroll
                    for (int c2 = c + 1; c2 < NB; c2++) {
                        const double l2 = __shfl(a[c], c2);
                        if (r >= c2) a[c2] -= a[c] * l2;
                    }
                }
                if (bad) { if (tid == 0) sFail = 1; }
                else if (tid < NB) {
#pragma unroll
                    for (int c = 0; c < NB; c++) if (c <= r) panel[r * NB + c] = a[c];
                }
            } else {
                for (int c = 0; c < nb; c++) {
                    const double dj = panel[c * NB + c];
                    if (!(dj > 0) || !isfinite(dj)) { if (tid == 0) sFail = 1; break; }   // wave-uniform
                    const double ljj = sqrt(dj);
                    __builtin_amdgcn_wave_barrier();
                    for (int r = c + 1 + tid; r < nb; r += 64) panel[r * NB + c] /= ljj;
                    if (tid == 0) panel[c * NB + c] = ljj;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    const int ncol = nb - c - 1;
                    for (int idx = tid; idx < ncol * ncol; idx += 64) {
                        const int c2 = c + 1 + idx / ncol, r = c + 1 + idx % ncol;
                        if (r >= c2) panel[r * NB + c2] -= panel[r * NB + c] * panel[c2 * NB + c];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        __syncthreads();
        if (sFail) break;
        // (2) rows below the diagonal block, one thread per row: L21[r][:] = A21[r][:] * L11^-T (no barrier: rows are independent)
        for (int r = nb + tid; r < rows; r += 1024) {
            double *pr = panel + r * NB;
            if (nb == NB) {   // full panel: the row lives in registers, L11 is read as LDS broadcasts
                double v[NB];
#pragma unroll
                for (int c = 0; c < NB; c++) v[c] = pr[c];
#pragma unroll
                for (int c = 0; c < NB; c++) {
                    double acc = v[c];
#pragma unroll
                    for (int k = 0; k < c; k++) acc -= v[k] * panel[c * NB + k];
                    v[c] = acc / panel[c * NB + c];
                }
#pragma unroll
                for (int c = 0; c < NB; c++) pr[c] = v[c];
            } else {
                for (int c = 0; c < nb; c++) {
                    double acc = pr[c];
                    for (int k = 0; k < c; k++) acc -= pr[k] * panel[c * NB + k];
                    pr[c] = acc / panel[c * NB + c];
                }
            }
        }
        __syncthreads();
        if (sFail) break;
        // forward substitution with this panel: y[p0..p0+nb) then b[r] -= sum_c L[r][c] y[c] for the rows below
        if (tid < 64) {
            for (int c = 0; c < nb; c++) {
                const double yc = sx[p0 + c] / panel[c * NB + c];
                __builtin_amdgcn_wave_barrier();
                if (tid == 0) sx[p0 + c] = yc;
                for (int r = c + 1 + tid; r < nb; r += 64) sx[p0 + r] -= panel[r * NB + c] * yc;
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        for (int r = nb + tid; r < rows; r += 1024) {
            double acc = 0;
            for (int c = 0; c < nb; c++) acc += panel[r * NB + c] * sx[p0 + c];
            sx[p0 + r] -= acc;
        }
        // write the factorised panel back, update the trailing matrix (lower triangle)
        for (int idx = tid; idx < rows * nb; idx += 1024) { const int r = idx / nb, c = idx - r * nb; S[(size_t)(p0 + r) * n + p0 + c] = panel[r * NB + c]; }
        // trailing update of the lower triangle in 4x4 register tiles: 8 LDS reads feed 16 multiply-adds
        const int m = rows - nb, mt = (m + 3) >> 2;
        for (int tIdx = tid; tIdx < mt * mt; tIdx += 1024) {
            const int i4 = tIdx / mt, k4 = tIdx - i4 * mt;
            if (k4 > i4) continue;
            double acc[4][4];
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int w = 0; w < 4; w++) acc[u][w] = 0;
            const double *pa = panel + (nb + 4 * i4) * NB, *pb = panel + (nb + 4 * k4) * NB;
            // rows past the end of the panel are never stored; clamp their reads into the panel
            const int ra[4] = {0, min(1, m - 1 - 4 * i4) * NB, min(2, m - 1 - 4 * i4) * NB, min(3, m - 1 - 4 * i4) * NB};
            const int rb[4] = {0, min(1, m - 1 - 4 * k4) * NB, min(2, m - 1 - 4 * k4) * NB, min(3, m - 1 - 4 * k4) * NB};
            for (int c = 0; c < nb; c++) {
                double a[4], bq[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { a[u] = pa[ra[u] + c]; bq[u] = pb[rb[u] + c]; }
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int w = 0; w < 4; w++) acc[u][w] += a[u] * bq[w];
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const int i = 4 * i4 + u, k = 4 * k4 + w;
                    if (i < m && k <= i) S[(size_t)(p0 + nb + i) * n + p0 + nb + k] -= acc[u][w];
                }
        }
        __syncthreads();
    }
    if (sFail) { if (tid == 0) *okFlag = 0; return; }
    // backward substitution L^T x = y, panels from the last to the first
    const int npan = (n + NB - 1) / NB;
    for (int pi = npan - 1; pi >= 0; pi--) {
        const int p0 = pi * NB, nb = min(NB, n - p0), rows = n - p0;
        for (int idx = tid; idx < rows * nb; idx += 1024) { const int r = idx / nb, c = idx - r * nb; panel[r * NB + c] = S[(size_t)(p0 + r) * n + p0 + c]; }
        __syncthreads();
        // x[p0+c] needs y[p0+c] - sum_{r>c, r in panel} L[r][c] x[r] - sum_{rows below the panel} L[r][c] x[r]
        if (tid < nb) {
            double acc = 0;
            for (int r = nb; r < rows; r++) acc += panel[r * NB + tid] * sx[p0 + r];
            sx[p0 + tid] -= acc;
        }
        __syncthreads();
        if (tid < 64) {
            for (int c = nb - 1; c >= 0; c--) {
                const double xc = sx[p0 + c] / panel[c * NB + c];
                __builtin_amdgcn_wave_barrier();
                if (tid == 0) sx[p0 + c] = xc;
                for (int r = tid; r < c; r += 64) sx[p0 + r] -= panel[c * NB + r] * xc;
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += 1024) x[i] = sx[i];
    if (tid == 0) *okFlag = 1;
}

// ---------------------------------------------------------------------------------------------
// Multi-workgroup right-looking Cholesky for the larger reduced systems (n >= CHOL_MULTI_MIN_N): the
// single-workgroup kernel above is latency bound (one CU, ~20 barriers per panel); here every panel
// step is ONE launch (k_chol_step) in which
//   the panel workgroups    factor the 32x32 diagonal block (each redundantly), solve 64 rows of the panel below it against L11
//                           and apply the panel to the right-hand side,
//   the other workgroups    subtract the PREVIOUS panel's L21 L21^T from the trailing matrix beyond this block column, one 32x32
//                           tile of the lower triangle each,
// and the substitution L^T x = y is one last single-workgroup kernel.  L is written to its own buffer
// (workgroups read the diagonal block of S while workgroup 0 stores L11), the solved part of y likewise.
// ---------------------------------------------------------------------------------------------
#define CHOL_MULTI_MIN_N 96
#define CNB 32
#define CHOL_RPW 63          /* panel rows per workgroup of k_chol_step: one wave, lane 63 carries the right-hand side */

// value of lane `src` (compile-time constant after unrolling) as a scalar broadcast: two v_readlane_b32 instead of two ds_bpermute_b32
__device__ __forceinline__ double readlane_f64(double v, int src)
{
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffu), src), hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), src);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// The panel step: nothing goes through synchronised column steps (a version with the 32x32 block in LDS and two workgroup barriers
// per column took 31 us per panel; a wave factoring the block in registers with one v_readlane pair per multiply-add and a second wave
// eliminating the rows afterwards 17; the blocked one-wave elimination with the panel rows inside it 7.9, with the rows on a wave of their own 5).
// Where a launch's ~9 us go now (tools/chol_phases.py, s_memtime on the first panel workgroup, profiles/r06_chol_phases.txt, 21.5k cycles between
// the first and the last stamp): staging 3.6k (17 %), the previous panel's update on the matrix cores 3.0k (14 %), the diagonal block 12.4k (57 %),
// waiting for the panel rows' wave 0.1k, stores 2.4k (11 %).  Inside the diagonal block, per block of four pivots: the pivot chain ~645 cycles
// (~160 per pivot: readlanes, v_rsq_f64 + Newton, the 4 x 4 part), own entries + publishing the four columns through LDS ~570, the rank-4 update of
// the rest ~250 - the 32-pivot chain itself is 2.2 us of a 72-launch, 0.7 ms share of the window.
// 1 / sqrt(x) for the pivots: v_rsq_f64 (~26 bits) + one Newton step y += y/2 (1 - x y^2), instead of the library routine: every
// dependent FP64 operation on the pivot chain costs ~16 cycles, and the factor is not part of the bit-level contract (1e-5 vs g2o)
__device__ __forceinline__ double pivot_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-x * y, y, 1.0);
    return __builtin_fma(0.5 * y, e, y);
}

// One step of the factorisation = ONE launch with two kinds of workgroups:
//   blocks [0, nPW)   PANEL of block column p: first apply the previous panel's update to their own part of the column
//                     (S[rows, p] -= L[rows, p-1] L[p-rows, p-1]^T, a 4x4 register tile per thread out of LDS), then factor the
//                     diagonal block and solve the rows below as described above;
//   the other blocks  the REST of the previous panel's trailing update, block columns > p, one 32x32 tile each.
// The two kinds touch disjoint parts of S and only read L[:, p-1], so they need no order between them: the trailing update no
// longer sits between two panels (15 dependent launches per factorisation become 8) and runs while the panel's serial chain does.
// PROF (tools/chol_phases.py, never in a product launch): thread 0 of the first panel workgroup adds the wall cycles (s_memtime) between its phase boundaries to prof[0..7],
// the passes to prof[8..15]: 0 staging, 1 previous panel's update (MFMA), 2 diagonal block (wave 0), 3 wait for the panel rows (wave 1), 4 stores.
//   inside the diagonal block, summed over its 8 blocks of four columns: 5 the pivot chain (readlanes, 4 x rsq + Newton, the 4 x 4 part), 6 own entries + publishing the four columns
//   through LDS, 7 the rank-4 update of the rest of the block.
#define CH_STAMP(i) do { if (PROF && blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[i] += t_ - tPrev; pcnt[i] += 1; tPrev = __builtin_amdgcn_s_memtime(); } } while (0)
template <bool PROF>
__global__ __launch_bounds__(256) void k_chol_step(double *__restrict__ S, double *__restrict__ L, int n, int p0, int nPW, int T1, double *ywork, double *ysol, int *okFlag,
                                                   double *__restrict__ diagInv /* 1 / L[i][i], for the substitution kernel */, const int *gate, int want, unsigned long long *prof)
{
    if (gate && *gate != want) return;
    unsigned long long tPrev = PROF ? __builtin_amdgcn_s_memtime() : 0ull, pacc[PROF ? 8 : 1] = {}, pcnt[PROF ? 8 : 1] = {};
    (void)tPrev; (void)pacc; (void)pcnt;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    __shared__ double Ld[CNB][CNB + 1];      // diagonal block in, L11 (lower incl. diagonal) out      | update role: la
    __shared__ double Tt[64][CNB + 1];       // 63 rows of the panel below + the right-hand side (row 63) | update role: lb (first 32 rows)
    __shared__ __attribute__((aligned(16))) double LpD[CNB][CNB + 2];     // previous panel, rows of this diagonal block; afterwards the column exchange buffer of the block factorisation
    __shared__ double LpR[64][CNB + 2];      // previous panel, this workgroup's rows (pitch 34: conflict-free MFMA operand reads)
    __shared__ int sBad;
    __shared__ __attribute__((aligned(16))) double cbx[CNB / 4][CNB][4];   // per 4-column block: L[q][c0 .. c0 + 3] of the 32 rows (zero up to the block's last row)
    __shared__ __attribute__((aligned(16))) double dpub[CNB / 4][12];       // per block: the factored 4x4 diagonal part (6 sub-diagonal entries) and the 4 reciprocal pivots
    __shared__ int xReady;                                                // blocks published by the wave of the diagonal block
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nPW) {
        // ---- rest of the trailing update of the previous panel (q0 = p0 - 32): tiles (I, K), I >= K >= 1, relative to row/column p0
        const int u = (int)blockIdx.x - nPW, K = 1 + u % T1, I = 1 + u / T1;
        if (K > I) return;
        const int q0 = p0 - CNB;
        double(*la)[CNB + 1] = Ld;
        double(*lb)[CNB + 1] = Tt;
        for (int idx = tid; idx < CNB * CNB; idx += 256) {
            const int r = idx >> 5, c = idx & 31;
            const int ra = p0 + CNB * I + r, rb = p0 + CNB * K + r;
            const double va = L[(size_t)min(ra, n - 1) * n + q0 + c], vb = L[(size_t)min(rb, n - 1) * n + q0 + c];      // (unconditional loads, see the panel's staging)
            la[r][c] = ra < n ? va : 0.0;
            lb[r][c] = rb < n ? vb : 0.0;
        }
        __syncthreads();
        const int ti = (tid >> 4) * 2, tk = (tid & 15) * 2;
        double acc[2][2] = {{0, 0}, {0, 0}};
#pragma unroll 8
        for (int c = 0; c < CNB; c++) {
            const double a0 = la[ti][c], a1 = la[ti + 1][c], b0 = lb[tk][c], b1 = lb[tk + 1][c];
            acc[0][0] = __builtin_fma(a0, b0, acc[0][0]); acc[0][1] = __builtin_fma(a0, b1, acc[0][1]);
            acc[1][0] = __builtin_fma(a1, b0, acc[1][0]); acc[1][1] = __builtin_fma(a1, b1, acc[1][1]);
        }
#pragma unroll
        for (int uu = 0; uu < 2; uu++)
#pragma unroll
            for (int w = 0; w < 2; w++) {
                const int row = p0 + CNB * I + ti + uu, col = p0 + CNB * K + tk + w;
                if (row < n && col <= row) S[(size_t)row * n + col] -= acc[uu][w];
            }
        return;
    }
    // ---- panel of block column p
    const int wave = tid >> 6, lane = tid & 63, nb = min(CNB, n - p0);
    const int r0 = p0 + nb + blockIdx.x * CHOL_RPW;   // first row of this workgroup's part of the panel below
    const bool hasPrev = p0 > 0;
    if (tid == 64) { sBad = 0; xReady = 0; }
    {   // global memory is only touched by the whole workgroup, a row segment of 32 doubles per 32 threads.  Every load is UNCONDITIONAL from a
        // clamped (always valid) address and masked afterwards: predicated loads sit in their own exec regions, which the compiler does not
        // merge, so each waited for its data before the next was issued - 24 dependent round trips to rows that the previous launch's
        // workgroups wrote on other XCDs (s_memrealtime: 6.6 us of staging per launch; with all loads in flight one round trip).
        double vd[4], vt[8], pd[4], pr[8];
        const int qc = hasPrev ? p0 - CNB : 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int idx = tid + 256 * q, r = idx >> 5, c = idx & 31;
            const size_t row = (size_t)(p0 + min(r, nb - 1)) * n;
            vd[q] = S[row + p0 + min(c, nb - 1)];
            pd[q] = L[row + qc + c];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = tid + 256 * q, r = idx >> 5, c = idx & 31;
            const size_t row = (size_t)min(r0 + r, n - 1) * n;
            const double *src = r == CHOL_RPW ? ywork + p0 + min(c, nb - 1) : S + row + p0 + min(c, nb - 1);
            vt[q] = *src;
            pr[q] = L[row + qc + c];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int idx = tid + 256 * q, r = idx >> 5, c = idx & 31;
            Ld[r][c] = (r < nb && c <= r) ? vd[q] : (r == c ? 1.0 : 0.0);   // identity pad past nb
            LpD[r][c] = (hasPrev && r < nb) ? pd[q] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = tid + 256 * q, r = idx >> 5, c = idx & 31;
            const bool row = r < CHOL_RPW && r0 + r < n;
            Tt[r][c] = ((r == CHOL_RPW || row) && c < nb) ? vt[q] : 0.0;
            LpR[r][c] = (hasPrev && row) ? pr[q] : 0.0;
        }
    }
    __syncthreads();
    CH_STAMP(0);
    if (hasPrev) {
        // own part of the previous panel's update on the matrix cores: C -= A B^T as v_mfma_f64_16x16x4_f64 (layout checked by
        // tools/ubench_mfma_f64.hip: A lane l = A[l % 16][l / 16], B lane l = B[l / 16][l % 16], D lane l element v = D[4 v + l / 16][l % 16]).
        // Wave w takes the two 16x16 tiles of panel rows 16 w .. 16 w + 15 and, waves 0..2, one of the three lower tiles of the
        // diagonal block; 8 instructions (K = 32) per tile, the accumulator starts as C and A is negated.  The row pitch of 34
        // doubles makes the operand reads conflict free.  (3.5 us as 4x4 register tiles on the vector ALUs before.)
        const int li = lane & 15, lk = lane >> 4;
        typedef double d4_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const bool isD = t == 2;
            if (isD && wave == 3) break;
            const int rt = isD ? (wave + 1) >> 1 : wave, ct = isD ? wave >> 1 : t;      // diagonal tiles: (0,0), (1,0), (1,1)
            d4_t acc;
#pragma unroll
            for (int v = 0; v < 4; v++) acc[v] = isD ? Ld[16 * rt + 4 * v + lk][16 * ct + li] : Tt[16 * rt + 4 * v + lk][16 * ct + li];
#pragma unroll
            for (int k4 = 0; k4 < CNB / 4; k4++) {
                const double av = isD ? LpD[16 * rt + li][4 * k4 + lk] : LpR[16 * rt + li][4 * k4 + lk];
                const double bv = LpD[16 * ct + li][4 * k4 + lk];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int rr = 16 * rt + 4 * v + lk, cc = 16 * ct + li;
                if (isD) { if (cc <= rr && rr < nb) Ld[rr][cc] = acc[v]; }
                else if (cc < nb) Tt[rr][cc] = acc[v];      // (row 63, the right-hand side: its LpR row is zero)
            }
        }
    }
    if (hasPrev) __syncthreads();
    CH_STAMP(1);
    if (wave == 0) {
        // THE DIAGONAL BLOCK on one wave, no barrier inside, blocked by four columns.  Lane = (row r, half hf) keeps A[r][16 hf .. 16 hf + 15] in
        // registers.  Per block the 4x4 diagonal part travels by v_readlane and is factored redundantly by every lane (wave-uniform scalars), every
        // lane eliminates its own entries against it, the four finished columns go through LDS ONCE (rows up to the block's last one stored as zero,
        // which also masks the finished columns and the upper triangle; one buffer per block, cbx[cb], because the wave of the panel rows below
        // reads them too, at its own pace), and the rank-4 update of the rest of the block reads its multipliers back as 16-byte broadcasts.
        // 8 exchange rounds and 8 x 4 dependent pivots instead of 32 column steps.  History (s_memrealtime per panel): columns in LDS with two
        // barriers each 31 us; a register wave with v_readlane per multiply-add + a second wave for the rows afterwards 17; this blocked wave with
        // the panel rows inside it 7.9; with the rows on their own wave (below) 5.
        const int r = lane & 31, hf = lane >> 5;
        double(*cbDump)[4] = (double(*)[4]) & LpD[0][0];  // [32][4]: where the half that does not own a block's columns drops its (unused) values
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; q++) a[q] = Ld[r][16 * hf + q];
        bool bad = false;
        double invOwn = 1.0;
#pragma unroll
        for (int cb = 0; cb < CNB / 4; cb++) {
            const int c0 = 4 * cb, hc = c0 >> 4, j0 = c0 & 15;
            double(*cb4)[4] = cbx[cb];
            // ---- region A: the pivot chain of this block (every dependent FP64 operation ~16 cycles).  Scheduling barriers keep the compiler from
            // merging the blocks (without them it hoists the loads of all eight and spills).
            double d[4][4], inv[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int k = 0; k <= i; k++) d[i][k] = readlane_f64(a[j0 + k], c0 + i + 32 * hc);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const double piv = d[k][k];
                if (!(piv > 0) || !isfinite(piv)) bad = true;          // wave-uniform
                inv[k] = pivot_rsqrt(piv);
                d[k][k] = piv * inv[k];
#pragma unroll
                for (int i = k + 1; i < 4; i++) d[i][k] *= inv[k];
#pragma unroll
                for (int i = k + 1; i < 4; i++)
#pragma unroll
                    for (int m = k + 1; m <= i; m++) d[i][m] = __builtin_fma(-d[i][k], d[m][k], d[i][m]);
                __builtin_amdgcn_sched_barrier(0);
            }
            unsigned long long tA = 0;
            if (PROF && blockIdx.x == 0 && threadIdx.x == 0) { tA = __builtin_amdgcn_s_memtime(); pacc[5] += tA - tPrev; pcnt[5] += 1; }
#pragma unroll
            for (int k = 0; k < 4; k++) invOwn = lane == c0 + k ? inv[k] : invOwn;      // lane c collects 1 / L[c][c] (no store on the chain)
            // own entries against the block (rows of the block itself reproduce d[][] in their lower part)
            double l[4];
            l[0] = a[j0] * inv[0];
            l[1] = __builtin_fma(-l[0], d[1][0], a[j0 + 1]) * inv[1];
            l[2] = __builtin_fma(-l[1], d[2][1], __builtin_fma(-l[0], d[2][0], a[j0 + 2])) * inv[2];
            l[3] = __builtin_fma(-l[2], d[3][2], __builtin_fma(-l[1], d[3][1], __builtin_fma(-l[0], d[3][0], a[j0 + 3]))) * inv[3];
            const bool own = hf == hc;
#pragma unroll
            for (int k = 0; k < 4; k++) a[j0 + k] = own ? l[k] : a[j0 + k];
            if (lane == 0) {      // what the wave of the panel rows needs of this block besides the columns below
                *(double2 *)&dpub[cb][0] = make_double2(d[1][0], d[2][0]); *(double2 *)&dpub[cb][2] = make_double2(d[2][1], d[3][0]);
                *(double2 *)&dpub[cb][4] = make_double2(d[3][1], d[3][2]); *(double2 *)&dpub[cb][6] = make_double2(inv[0], inv[1]);
                *(double2 *)&dpub[cb][8] = make_double2(inv[2], inv[3]);
            }
            const bool below = r > c0 + 3;
            double *wr = own ? cb4[r] : cbDump[r];
            *(double2 *)wr = make_double2(below ? l[0] : 0.0, below ? l[1] : 0.0);
            *(double2 *)(wr + 2) = make_double2(below ? l[2] : 0.0, below ? l[3] : 0.0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(&xReady, cb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // block cb is published
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (PROF && blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[6] += t_ - tA; pcnt[6] += 1; tPrev = t_; }
            if (cb == CNB / 4 - 1) break;        // nothing left to update
            // ---- region B: the rest of the diagonal block (no fence needed: the LDS executes the instructions of a wave in order)
            const double2 lr01 = *(const double2 *)cb4[r], lr23 = *(const double2 *)(cb4[r] + 2);
#pragma unroll
            for (int q = (c0 < 16 ? 0 : j0 + 4); q < 16; q++) {
                const double *mp = cb4[16 * hf + q];
                const double2 m01 = *(const double2 *)mp, m23 = *(const double2 *)(mp + 2);
                a[q] = __builtin_fma(-lr23.y, m23.y, __builtin_fma(-lr23.x, m23.x, __builtin_fma(-lr01.y, m01.y, __builtin_fma(-lr01.x, m01.x, a[q]))));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (PROF && blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[7] += t_ - tPrev; pcnt[7] += 1; tPrev = t_; }
        }
#pragma unroll
        for (int q = 0; q < 16; q++) if (16 * hf + q <= r) Ld[r][16 * hf + q] = a[q];
        if (bad && lane == 0) sBad = 1;
        if (blockIdx.x == 0 && lane < CNB) diagInv[p0 + lane] = invOwn;      // (the buffer is padded to a multiple of 32)
        CH_STAMP(2);
    } else if (wave == 1) {
        // THE PANEL ROWS on a second wave (another SIMD): lane = panel row with its 32 entries in registers, lane 63 the right-hand side.  It follows
        // the diagonal block's wave one block behind: as soon as block cb is published (its four columns of L11 in cbx[cb], the factored 4x4 part and
        // the reciprocal pivots in dpub[cb]; a counter in LDS, polled) it solves its four entries and applies the rank-4 update to the rest of its
        // row.  The elimination wave is bound by instruction issue, not by latency (~1300 FP64 operations, 420 ds_read_b128, the AGPR traffic of
        // 500 live registers in 19 k cycles): the rows' share - 450 multiply-adds and 290 of the LDS reads - now issues next to it instead of inside it.
        double x[CNB];
#pragma unroll
        for (int c = 0; c < CNB; c++) x[c] = Tt[lane][c];
        const bool ywLive = lane < CHOL_RPW && r0 + lane < n;
        const double yw = ywLive ? ywork[r0 + lane] : 0.0;     // in flight during the elimination
#pragma unroll
        for (int cb = 0; cb < CNB / 4; cb++) {
            const int c0 = 4 * cb;
            while (__hip_atomic_load(&xReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= cb) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const double2 p0_ = *(const double2 *)&dpub[cb][0], p1_ = *(const double2 *)&dpub[cb][2], p2_ = *(const double2 *)&dpub[cb][4],
                          p3_ = *(const double2 *)&dpub[cb][6], p4_ = *(const double2 *)&dpub[cb][8];
            const double d10 = p0_.x, d20 = p0_.y, d21 = p1_.x, d30 = p1_.y, d31 = p2_.x, d32 = p2_.y, i0 = p3_.x, i1 = p3_.y, i2 = p4_.x, i3 = p4_.y;
            x[c0] = x[c0] * i0;
            x[c0 + 1] = __builtin_fma(-x[c0], d10, x[c0 + 1]) * i1;
            x[c0 + 2] = __builtin_fma(-x[c0 + 1], d21, __builtin_fma(-x[c0], d20, x[c0 + 2])) * i2;
            x[c0 + 3] = __builtin_fma(-x[c0 + 2], d32, __builtin_fma(-x[c0 + 1], d31, __builtin_fma(-x[c0], d30, x[c0 + 3]))) * i3;
#pragma unroll
            for (int q = c0 + 4; q < CNB; q++) {      // multipliers L[q][c0 .. c0 + 3], the same address for every lane
                const double2 m01 = *(const double2 *)cbx[cb][q], m23 = *(const double2 *)(cbx[cb][q] + 2);
                x[q] = __builtin_fma(-x[c0 + 3], m23.y, __builtin_fma(-x[c0 + 2], m23.x, __builtin_fma(-x[c0 + 1], m01.y, __builtin_fma(-x[c0], m01.x, x[q]))));
            }
        }
#pragma unroll
        for (int c = 0; c < CNB; c++) Tt[lane][c] = x[c];
        // b of the rows below -= L21 y   (y = the solved right-hand side in lane 63)
        double dot[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < CNB; c++) dot[c & 3] = __builtin_fma(x[c], readlane_f64(x[c], CHOL_RPW), dot[c & 3]);
        if (ywLive) ywork[r0 + lane] = yw - ((dot[0] + dot[1]) + (dot[2] + dot[3]));
    }
    __syncthreads();
    CH_STAMP(3);
    for (int idx = tid; idx < CHOL_RPW * CNB; idx += 256) {
        const int r = idx >> 5, c = idx & 31;
        if (r0 + r < n && c < nb) L[(size_t)(r0 + r) * n + p0 + c] = Tt[r][c];
    }
    if (blockIdx.x == 0) {      // L11 and the solved right-hand side of this panel
        for (int idx = tid; idx < CNB * CNB; idx += 256) {
            const int r = idx >> 5, c = idx & 31;
            if (r < nb && c <= r) L[(size_t)(p0 + r) * n + p0 + c] = Ld[r][c];
        }
        if (tid < nb) ysol[p0 + tid] = Tt[CHOL_RPW][tid];
        if (sBad && tid == 0) *okFlag = 0;   // not positive definite: the results are discarded by the host
    }
    if (PROF) {
        __syncthreads();
        CH_STAMP(4);
        if (blockIdx.x == 0 && threadIdx.x == 0) for (int i = 0; i < 8; i++) { atomicAdd(&prof[i], pacc[i]); atomicAdd(&prof[8 + i], pcnt[i]); }
    }
}
#undef CH_STAMP

// L^T x = y, panels from the last to the first; x in LDS
// XG = true (n > CHOL_LDS_X): the vector lives in x itself; one workgroup, so __syncthreads() orders the accesses, and they are made
// at agent scope so that no wave reads a stale line of the per-CU vector cache.
template <bool XG> struct XVec {
    double *p;
    __device__ double get(int i) const { return XG ? __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[i]; }
    __device__ void set(int i, double v) const { if (XG) __hip_atomic_store(p + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else p[i] = v; }
};
template <bool XG>
__global__ __launch_bounds__(1024) void k_chol_backsub(const double *__restrict__ L, const double *__restrict__ ysol, int n, double *x, const int *gate, int want)
{
    if (gate && *gate != want) return;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    __shared__ double sxLds[XG ? 1 : CHOL_LDS_X];
    __shared__ double part[32][CNB + 1];
    __shared__ double l11[CNB][CNB + 1];
    const int tid = threadIdx.x, c = tid & 31, pt = tid >> 5;
    XVec<XG> sx = {XG ? x : sxLds};
    for (int i = tid; i < n; i += 1024) sx.set(i, ysol[i]);
    __syncthreads();
    const int npan = (n + CNB - 1) / CNB;
    for (int pi = npan - 1; pi >= 0; pi--) {
        const int p0 = pi * CNB, nb = min(CNB, n - p0);
        double acc = 0;
        if (c < nb)
            for (int r = p0 + nb + pt; r < n; r += 32) acc += L[(size_t)r * n + p0 + c] * sx.get(r);
        part[pt][c] = acc;
        {
            const int r = tid >> 5, cc = tid & 31;
            l11[r][cc] = (r < nb && cc <= r) ? L[(size_t)(p0 + r) * n + p0 + cc] : 0.0;
        }
        __syncthreads();
        if (tid < nb) {
            double t = 0;
            for (int q = 0; q < 32; q++) t += part[q][tid];
            sx.set(p0 + tid, sx.get(p0 + tid) - t);
        }
        __syncthreads();
        if (tid < 64) {
            // L11^T x = y inside one wave, in registers: lane t keeps x_t and column t of L11; step cc broadcasts the finished x_cc
            // (v_readlane) and every lane below subtracts its product - no LDS round trip, no division on the 32-step chain (the
            // reciprocal diagonals are formed up front, all lanes at once)
            const int t = tid & (CNB - 1);
            double lcol[CNB];
#pragma unroll
            for (int cc = 0; cc < CNB; cc++) lcol[cc] = (cc > t && cc < nb) ? l11[cc][t] : 0.0;
            const double inv = t < nb ? 1.0 / l11[t][t] : 1.0;
            double xv = t < nb ? sx.get(p0 + t) : 0.0;
#pragma unroll
            for (int cc = CNB - 1; cc >= 0; cc--) {
                const double xc = readlane_f64(xv * inv, cc);
                xv = t == cc ? xc : __builtin_fma(-lcol[cc], xc, xv);
            }
            if (tid < nb) sx.set(p0 + tid, xv);
        }
        __syncthreads();
    }
    if (!XG) for (int i = tid; i < n; i += 1024) x[i] = sxLds[i];
}

// L^T x = y for n <= 32 NP (the local-BA sizes), latency version: the kernel above waits for its loads of L once per panel (~3 us
// each, 30 us for 50 keyframes); here nothing on the serial chain touches global memory.  Column oriented: after the 32 unknowns
// of panel p are solved (one wave, registers + v_readlane as above, the reciprocal diagonal folded into the columns beforehand),
// every thread adds its share of  L[p rows][col]^T x_p  to a register accumulator of ITS column (thread = (column, third of the 32
// rows)); the row block of L it needs was requested two panels earlier and sits in registers.  The three accumulators of a column
// meet in LDS only when that column's panel is next.  Two (LDS-only) barriers per panel.
// Workgroup barrier that only waits for this wave's LDS traffic: __syncthreads() is a workgroup-scope release + acquire and makes
// the compiler wait for vmcnt(0) as well, i.e. for every global load in flight - exactly the prefetches the kernel below keeps
// outstanding across its barriers.  Only LDS data is exchanged at these barriers.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NP>
__global__ __launch_bounds__(1024) void k_chol_backsub_reg(const double *__restrict__ L, const double *__restrict__ ysol, const double *__restrict__ diagInv, int n, double *x, const int *gate, int want)
{
    if (gate && *gate != want) return;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    // columns; row groups of the 1024 threads (3 above 256 columns, else 4); rows per thread and panel (3 x 11, 4 x 8 >= 32).  With four groups a
    // thread keeps 8 instead of 11 prefetched values per panel in flight: no register spill - and a spill reload inside the panel loop is followed by
    // s_waitcnt vmcnt(0), which also waits for every prefetched row block (1.8 us per panel instead of the substitution's own ~1)
    constexpr int CW = 32 * NP, RG = 1024 / CW >= 4 ? 4 : 1024 / CW, RQ = (CNB + RG - 1) / RG;
    __shared__ double y[CW], di[CW], part[4][CNB], l11s[2][CNB][CNB + 1], xs[CNB + 1];
    const int tid = threadIdx.x, col = tid % CW, rq = tid / CW, lane = tid & 63;
    const bool upd = rq < RG && tid < RG * CW;
    double lv[NP][RQ], ld[NP], accs[2] = {0, 0};
    auto request = [&](auto P) {       // row block and diagonal block of panel P (compile-time index)
        constexpr int pp = decltype(P)::value;
        const int p0 = CNB * pp;
#pragma unroll
        for (int k = 0; k < RQ; k++) {
            const int rr = RQ * rq + k;
            lv[pp][k] = (upd && rr < CNB && p0 + rr < n && col < p0) ? L[(size_t)(p0 + rr) * n + col] : 0.0;
        }
        const int r = tid >> 5, cc = tid & 31;
        ld[pp] = (p0 + r < n && cc < r) ? L[(size_t)(p0 + r) * n + p0 + cc] : 0.0;   // strictly lower part: the reciprocal diagonal comes from diagInv
    };
    request(std::integral_constant<int, NP - 1>());
    if (NP > 1) request(std::integral_constant<int, (NP > 1 ? NP - 2 : 0)>());
    for (int i = tid; i < CW; i += 1024) { y[i] = i < n ? ysol[i] : 0.0; di[i] = i < n ? diagInv[i] : 1.0; }
    if (tid == 0) xs[CNB] = 0.0;
#pragma unroll
    for (int pp = NP - 1; pp >= 0; pp--) {
        const int p0 = CNB * pp;
        l11s[pp & 1][tid >> 5][tid & 31] = ld[pp];
        if (upd && col >= p0 && col < p0 + CNB) part[rq][col - p0] = accs[0] + accs[1];
        lds_barrier();
        if (tid < 64) {
            const int t = lane & (CNB - 1);
            const double inv = di[p0 + t];
            double xv = (y[p0 + t] - ((part[0][t] + part[1][t]) + (RG > 3 ? part[2][t] + part[3][t] : part[2][t]))) * inv;      // scaled unknown: x_t once every later one is subtracted
            // column t of L11 in two halves of 16 registers (the prefetched row blocks need the rest of the 128 a thread may use)
#pragma unroll
            for (int hh = 1; hh >= 0; hh--) {
                double lc[16];
#pragma unroll
                for (int k = 0; k < 16; k++) lc[k] = l11s[pp & 1][16 * hh + k][t] * inv;      // unconditional (a guarded load becomes a branch per element): zero on and above the diagonal
#pragma unroll
                for (int k = 15; k >= 0; k--) {
                    const int cc = 16 * hh + k;
                    xv = __builtin_fma(-lc[k], readlane_f64(xv, cc), xv);      // (lc = 0 for the lanes t >= cc: they keep their value)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tid < CNB) { xs[t] = xv; y[p0 + t] = xv; }
        }
        lds_barrier();
        if (pp > 0) {
#pragma unroll
            for (int k = 0; k < RQ; k++) accs[k & 1] = __builtin_fma(lv[pp][k], xs[min(RQ * rq + k, CNB)], accs[k & 1]);   // (rows past 31: lv = 0, xs[32] = 0)
        }
        if (pp >= 2) {
            switch (pp) {   // compile-time panel index for the register arrays
#define ORBX_REQ(Q) case Q + 2: if (Q + 2 <= NP - 1) request(std::integral_constant<int, (Q <= NP - 1 ? Q : 0)>()); break;
                ORBX_REQ(0) ORBX_REQ(1) ORBX_REQ(2) ORBX_REQ(3) ORBX_REQ(4) ORBX_REQ(5) ORBX_REQ(6) ORBX_REQ(7)
#undef ORBX_REQ
            }
        }
    }
    for (int i = tid; i < n; i += 1024) x[i] = y[i];
}

// x_l = D^-1 (b_l - B^T x_p)   (block_solver.hpp:459-481)
// x_l, push() and update(x) in one launch: thread t computes the increment of landmark t (k_backsub), saves the estimates of
// keyframe t / landmark t (SparseOptimizer::push, sparse_optimizer.cpp:502-506: every vertex) and applies the increments (oplus).
__global__ __launch_bounds__(256) void k_backsub_update(LbaDev d, const int *ptStart, const int *ptEdges, const int *__restrict__ ptPi, const double *bl, const double *Dinv,
                                                        const double *xp, double *xl, DPose *poseBak, double *ptBak, double lambda, double *partL, const double *lamSrc, const int *gate, int want)
{
    if (gate && *gate != want) return;      // device-side LM: this launch belongs to a trial the stage no longer needs (or to a branch not taken)
    if (lamSrc) lambda = *lamSrc;      // the stage's lambda lives on the device (LmState, k_lm_decide)
    __shared__ double sw[4];
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < d.K) {                                   // keyframe g
        DPose T = d.pose[g];
        poseBak[g] = T;
        const int pi = d.poseIdx[g];
        if (pi >= 0) { pose_oplus(T, xp + 6 * pi); d.pose[g] = T; }
    }
    const int t = g >> 4, a = g & 15;                // landmark t, 16 lanes per landmark
    double sc = 0;                                   // this thread's share of sum xl (lambda xl + bl)
    if (t < d.P) {
        const int li = d.ptIdx[t];
        double c[3] = {0, 0, 0};
        if (li >= 0)
            for (int s = ptStart[t] + a; s < ptStart[t + 1]; s += 16) {
                const int e = ptEdges[s], pi = ptPi[s];      // the stage's per-slot free-pose index: -1 = inactive edge or fixed keyframe (two dependent loads less)
                if (pi < 0) continue;
                const double *B1 = eb_hpl(d, e);
#pragma unroll
                for (int q = 0; q < 3; q++) { double acc = 0; for (int r = 0; r < 6; r++) acc += B1[3 * r + q] * xp[6 * pi + r]; c[q] += acc; }
            }
#pragma unroll
        for (int q = 0; q < 3; q++) c[q] = sum16(c[q]);
        if (a == 0) {
            double X[3] = {d.pt[3 * (size_t)t], d.pt[3 * (size_t)t + 1], d.pt[3 * (size_t)t + 2]};
            for (int i = 0; i < 3; i++) ptBak[3 * (size_t)t + i] = X[i];
            if (li >= 0) {
                for (int q = 0; q < 3; q++) c[q] = bl[(size_t)li * 3 + q] - c[q];
                const double *I = Dinv + (size_t)li * 9;
                for (int i = 0; i < 3; i++) {
                    const double x = I[3 * i] * c[0] + I[3 * i + 1] * c[1] + I[3 * i + 2] * c[2];
                    xl[(size_t)li * 3 + i] = x;
                    d.pt[3 * (size_t)t + i] = X[i] + x;
                    sc += x * (lambda * x + bl[(size_t)li * 3 + i]);
                }
            }
        }
    }
    const double tot = block_sum256(sc, sw);
    if (threadIdx.x == 0) partL[blockIdx.x] = tot;
}


// ---------------------------------------------------------------------------------------------
// Optimizer::PoseOptimization(Frame*) (reference src/Optimizer.cc:363-605): motion-only BA.  ONE
// KERNEL, one workgroup per frame: the whole 4-round / 10-iteration / 10-trial Levenberg loop of g2o
// (OptimizationAlgorithmLevenberg::solve, optimization_algorithm_levenberg.cpp:61-164) runs on the
// device, edges strided over the 256 threads, the 6x6 system reduced through LDS and solved by one
// thread, no host round trip.  FP64 throughout; frames of a batch are independent.
// ---------------------------------------------------------------------------------------------
struct PoseOptDev {
    const float *pose0;      // [B*16] pFrame->mTcw
    const float *cam;        // [B*5] fx fy cx cy mbf
    const float *Xw;         // [B*cap*3] MapPoint::GetWorldPos of the features that have a MapPoint
    const float *obs;        // [B*cap*3] kpUn.pt.x, kpUn.pt.y, mvuRight (< 0: monocular edge)
    const float *invS2;      // [B*cap]   mvInvLevelSigma2[kpUn.octave]
    const int32_t *counts;   // [B]
    int cap;
    float *poseOut;          // [B*16]
    uint8_t *outlier;        // [B*cap]  pFrame->mvbOutlier of those features
    int32_t *ret;            // [B] nInitialCorrespondences - nBad
    double *stats;           // [B*8] per round: iterations, final (robustified) chi2
    // the host call's completion (OrbxCallBox): the outputs above are mapped pinned memory, the last frame's workgroup raises the sequence word
    unsigned *pubCounter;
    unsigned long long *pubFlag;
    unsigned long long pubSeq;
    unsigned long long *prof;   // PROF instantiation only: [32] phase cycles / counts
    // the single-frame host call: count, camera and pose travel in the kernel arguments (read from the mapped buffer they are two dependent PCIe round trips
    // - the count, then everything sized by it - at the head of a kernel the caller is waiting for)
    int one, count0;
    float cam0[5], pose00[16];
};

#define PO_NRED 28   /* 21 upper-triangle entries of H + 6 of b + chi2 */

// pose_map for k_pose_opt with fused multiply-adds (21 + 3 instead of 30 + 3 operations per point; the kernel's passes over the edges are bound by
// instruction issue).  The reference's own build contracts these expressions wherever its compiler's -march has FMA; the estimates are held to 1e-5.
__device__ __forceinline__ void po_pose_map(const DPose &T, const double v[3], double o[3])
{
    const DQuat &q = T.q;
    const double ux = 2 * __builtin_fma(q.y, v[2], -(q.z * v[1])), uy = 2 * __builtin_fma(q.z, v[0], -(q.x * v[2])), uz = 2 * __builtin_fma(q.x, v[1], -(q.y * v[0]));
    o[0] = (__builtin_fma(q.w, ux, v[0]) + __builtin_fma(q.y, uz, -(q.z * uy))) + T.t[0];
    o[1] = (__builtin_fma(q.w, uy, v[1]) + __builtin_fma(q.z, ux, -(q.x * uz))) + T.t[1];
    o[2] = (__builtin_fma(q.w, uz, v[2]) + __builtin_fma(q.x, uy, -(q.y * ux))) + T.t[2];
}

__device__ inline void po_edge_error(const DPose &T, const double in[5], const float *Xw, const float *obs, bool stereo, double out[3])
{
    const double X[3] = {(double)Xw[0], (double)Xw[1], (double)Xw[2]};
    double Xc[3];
    po_pose_map(T, X, Xc);
    if (!stereo) {   // EdgeSE3ProjectXYZOnlyPose::computeError / cam_project (types_six_dof_expmap.h:150-157, .cpp:290-296)
        const double iz = fast_rcp(Xc[2]);      // (<= 1 ulp from the reference's two divisions; 40 error passes of ~3 edges per thread in series)
        const double u = __builtin_fma(Xc[0] * iz, in[0], in[2]), v = __builtin_fma(Xc[1] * iz, in[1], in[3]);
        out[0] = (double)obs[0] - u; out[1] = (double)obs[1] - v; out[2] = 0;
    } else {         // EdgeStereoSE3ProjectXYZOnlyPose (.cpp:299-306): float invz, double bf
        const float invz = (float)fast_rcp(Xc[2]);
        const double u = __builtin_fma(Xc[0] * invz, in[0], in[2]), v = __builtin_fma(Xc[1] * invz, in[1], in[3]);
        out[0] = (double)obs[0] - u; out[1] = (double)obs[1] - v; out[2] = (double)obs[2] - __builtin_fma(-in[4], (double)invz, u);
    }
}

// The same two functions without control flow: the loops over a thread's edges are unrolled into ONE basic block, so that the scheduler interleaves the
// edges' dependent FP64 chains (one wave per SIMD: nothing else hides the ~8 cycles between dependent instructions; build 5260 -> see
// profiles/r06_pose_opt_phases.txt).  Values are those of po_edge_error / huber_rho bit for bit (the mono branch's double reciprocal, the stereo branch's
// float one; both sides of the Huber test evaluated, one selected).
__device__ __forceinline__ void po_edge_error_nb(const DPose &T, const double in[5], const float *Xw, const float *obs, bool stereo, double out[3])
{
    const double X[3] = {(double)Xw[0], (double)Xw[1], (double)Xw[2]};
    double Xc[3];
    po_pose_map(T, X, Xc);
    const double izd = fast_rcp(Xc[2]);
    const double iz = stereo ? (double)(float)izd : izd;
    const double u = __builtin_fma(Xc[0] * iz, in[0], in[2]), v = __builtin_fma(Xc[1] * iz, in[1], in[3]);
    out[0] = (double)obs[0] - u; out[1] = (double)obs[1] - v;
    const double o2 = (double)obs[2] - __builtin_fma(-in[4], iz, u);
    out[2] = stereo ? o2 : 0.0;
}
__device__ __forceinline__ void huber_rho_nb(const Huber &h, bool stereo, double chi, double &rho0, double &rho1)
{
    // sqrt(chi) and delta / sqrt(chi) from ONE v_rsq_f64 + two Newton steps (<= 1 ulp) instead of an IEEE square root and an IEEE division - ~80 of the
    // ~250 instructions an edge costs in these issue-bound passes; chi = 0: the products are NaN, and not selected
    const double delta = stereo ? h.dStereo : h.dMono, dsqr = stereo ? h.dsqrStereo : h.dsqrMono;
    const double rs = fast_rsqrt(chi), s = chi * rs;
    const bool in = chi <= dsqr;
    rho0 = in ? chi : __builtin_fma(2 * s, delta, -dsqr);
    rho1 = in ? 1. : delta * rs;
}

// a lane's value moved by a DPP control (two full-rate v_mov_b32_dpp, ~8 cycles of latency; __shfl_xor of a double is two ds_bpermute_b32 through the LDS
// crossbar, ~130 cycles per step of a dependent chain)
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u & 0xffffffffu), CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// the wave's sum in every lane, in a fixed order: pairs, quads, half rows, rows (DPP butterflies), then the four rows' sums as scalar broadcasts
__device__ __forceinline__ double wave_sum_f64(double v)
{
    v += dpp_f64<0xb1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f64<0x4e>(v);      // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);     // row_half_mirror
    v += dpp_f64<0x140>(v);     // row_mirror: every lane holds its row's sum
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return (r0 + r1) + (r2 + r3);
}

template <int N> __device__ inline void po_block_reduce(double (&v)[N], double (*red)[PO_NRED], int tid)
{
    // v[0..N) per thread -> v[0..N) of EVERY thread = the sums over the workgroup (N is a template argument: v stays in registers).  One barrier: the four
    // waves' sums meet in red[wave][k] and every thread adds them itself, in the same order.  The caller keeps a barrier between two calls (red is reused).
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const double x = wave_sum_f64(v[k]);
        if (lane == 0) red[wave][k] = x;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
}

// The 28 sums of an iteration (21 entries of H, 6 of b, chi2) through a transposed LDS tile instead of 28 x 6 double shuffles per thread
// (6 us of an 11 us iteration): thread t stores its 28 values at [k][t], 224 threads add 32 consecutive entries each (four chains) and the 8 parts
// of a sum - eight neighbouring lanes - meet by three DPP butterflies: fixed order, two barriers.  Index (k, t) -> k * 264 + (t >> 5) * 33 + (t & 31).
#define PO_TP 264
__device__ __forceinline__ void po_block_reduce28(const double (&v)[PO_NRED], double *redT, double (*red)[PO_NRED], double *sH, double *sb, int tid)
{
    const int slot = (tid >> 5) * 33 + (tid & 31);
#pragma unroll
    for (int k = 0; k < 28; k++) redT[k * PO_TP + slot] = v[k];
    __syncthreads();
    if (tid < 224) {      // (28 groups of 8 lanes: a group is entirely inside one half row of a wave, and entirely active)
        const int k = tid >> 3, p = tid & 7;
        const double *src = redT + k * PO_TP + p * 33;
        double s4[4] = {0, 0, 0, 0};      // four chains instead of one of 32 dependent additions
#pragma unroll
        for (int i = 0; i < 32; i++) s4[i & 3] += src[i];
        double sacc = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        // the eight parts of a sum sit in eight neighbouring lanes: three DPP butterflies instead of a trip through LDS and a barrier
        sacc += dpp_f64<0xb1>(sacc);
        sacc += dpp_f64<0x4e>(sacc);
        sacc += dpp_f64<0x141>(sacc);
        // the sums go where the solve reads them - H with both triangles (k = i * 6 - i (i - 1) / 2 + (j - i) of the upper one), b, and the chi2 for every thread
        if (p == 0) {
            if (k < 21) {
                const int i = (k >= 6) + (k >= 11) + (k >= 15) + (k >= 18) + (k >= 20), j = i + (k - (i * 6 - i * (i - 1) / 2));
                sH[6 * i + j] = sacc; sH[6 * j + i] = sacc;
            } else if (k < 27) sb[k - 21] = sacc;
            else red[0][27] = sacc;
        }
    }
    __syncthreads();
}

// NE = edges per thread: the correspondences of a frame and their _error live in REGISTERS for the whole call (thread t owns edges
// t, t + 256, ...): the 40 build / 40+ error passes of a call do no global memory access at all (each one waited ~1 us for its loads
// and the per-edge divisions before: 4.4 + 2.3 us of a 13 us iteration).
// PROF (tools/pose_opt_phases.py, never in a product launch): thread 0 reads s_memtime at the phase boundaries of the loop and adds the wall cycles between
// consecutive stamps (barrier waits included) to P.prof[0..15], the number of times a phase ran to P.prof[16..31].
#define PO_STAMP(i) do { if (PROF && tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[i] += t_ - tPrev; pcnt[i] += 1; tPrev = t_; } } while (0)
template <int NE, bool PROF>
__global__ __launch_bounds__(256) void k_pose_opt(PoseOptDev P, Huber hub)
{
    unsigned long long pacc[PROF ? 16 : 1] = {}, pcnt[PROF ? 16 : 1] = {}, tPrev = 0;
    (void)pacc; (void)pcnt; (void)tPrev;
    __shared__ double redT[28 * PO_TP];
    __shared__ DPose pose, savePose;
    __shared__ double red[4][PO_NRED];
    __shared__ double sH[36], sb[6], sx[6];
    __shared__ double sLambda, sNi, sCur, sRho;
    __shared__ int sOk, sIterOk, sNBad;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = min(P.one ? P.count0 : P.counts[f], P.cap);
    const size_t base = (size_t)f * P.cap;
    const float *Xw = P.Xw + base * 3, *obs = P.obs + base * 3, *invS2 = P.invS2 + base;
    uint8_t *outl = P.outlier + base;
    double in[5];
#pragma unroll
    for (int i = 0; i < 5; i++) in[i] = (double)(P.one ? P.cam0[i] : P.cam[5 * (size_t)f + i]);
    const float *p0 = P.pose0 + 16 * (size_t)f;      // (the by-value copy is read with constant indices only: a pointer into the argument block would put it on the stack)
    if (tid < 8) P.stats[8 * (size_t)f + tid] = 0;
    if (n < 3) {   // :509-510
        if (tid < n) outl[tid] = 0;
        if (tid < 16) P.poseOut[16 * (size_t)f + tid] = p0[tid];
        if (tid == 0) P.ret[f] = 0;
        if (P.pubFlag) orbx_publish(P.pubCounter, P.pubFlag, P.pubSeq, gridDim.x);
        return;
    }
    float xw[NE][3], ob[NE][3], is2[NE];
    double er[NE][3];
    unsigned outM = 0;            // bit j: edge tid + 256 j is an outlier (level 1)
    if (NE <= 8) {
        // The inputs may be MAPPED HOST memory (the single-frame host call): a thread's own edges are 12-byte records 3 KB apart, seven loads per edge
        // across PCIe (+11 us per call).  Read them as flat float arrays, coalesced, into LDS (the reduction tile is free until the first build:
        // 7 n floats <= 14784 for n <= 2048) and pick the thread's edges from there.
        float *stg = (float *)redT;
        for (int k = tid; k < 3 * n; k += 256) { stg[k] = Xw[k]; stg[3 * n + k] = obs[k]; }
        for (int k = tid; k < n; k += 256) stg[6 * n + k] = invS2[k];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NE; j++) {
            const int e = tid + 256 * j;
            const bool live = e < n;
#pragma unroll
            for (int i = 0; i < 3; i++) { xw[j][i] = live ? stg[3 * e + i] : 0.f; ob[j][i] = live ? stg[3 * n + 3 * e + i] : 0.f; er[j][i] = 0; }
            is2[j] = live ? stg[6 * n + e] : 0.f;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NE; j++) {
            const int e = tid + 256 * j;
            const bool live = e < n;
#pragma unroll
            for (int i = 0; i < 3; i++) { xw[j][i] = live ? Xw[3 * e + i] : 0.f; ob[j][i] = live ? obs[3 * e + i] : 0.f; er[j][i] = 0; }
            is2[j] = live ? invS2[e] : 0.f;
        }
    }
    __syncthreads();
    if (PROF && threadIdx.x == 0) tPrev = __builtin_amdgcn_s_memtime();
    PO_STAMP(0);      // (zero: the stamp's own cost shows up in phase 0's count)
    // vSE3->setEstimate(Converter::toSE3Quat(pFrame->mTcw)) (:520) is the same estimate in every round: converted once (a sqrt, a reciprocal and a normalisation
    // on one thread, ~1500 cycles, were repeated four times).  The number of active edges of a round is n minus the outliers the previous round's
    // classification counted (a block reduction of its own until round 6).
    __shared__ DPose pose0;
    if (tid == 0) {
        double R[9];
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int j = 0; j < 3; j++) R[3 * i + j] = (double)(P.one ? P.pose00[4 * i + j] : p0[4 * i + j]);
        }
        pose0.q = quat_from_R(R);
        quat_normalize_pos(pose0.q);
#pragma unroll
        for (int i = 0; i < 3; i++) pose0.t[i] = (double)(P.one ? P.pose00[4 * i + 3] : p0[4 * i + 3]);
    }
    int nAct = n;      // (round 0: no outliers yet)
    for (int round = 0; round < 4; round++) {
        const bool robust = round < 3;   // kernels are removed while classifying after the third round (:547-548)
        if (tid == 0) { pose = pose0; sIterOk = 1; sNBad = 0; }
        __syncthreads();
        int itersDone = 0;
        double lastChi = 0;
        PO_STAMP(1);      // round setup: estimate reset, active-edge count
        for (int it = 0; it < 10 && nAct > 0; it++) {
            if (!sIterOk) break;
            // ---- computeActiveErrors + robust chi2, buildSystem
            const DPose T = pose;
            double acc[PO_NRED];
#pragma unroll
            for (int k = 0; k < PO_NRED; k++) acc[k] = 0;
#pragma unroll
            for (int j = 0; j < NE; j++) {
                if (tid + 256 * j >= n || ((outM >> j) & 1u)) continue;      // (measured: this pass without control flow is no faster - it is bound by issue, not by latency - and pays for dead slots)
                const bool st = !(ob[j][2] < 0);
                double r[3];
                po_edge_error(T, in, xw[j], ob[j], st, r);
                er[j][0] = r[0]; er[j][1] = r[1]; er[j][2] = r[2];
                const double w = (double)is2[j];
                const double chi = __builtin_fma(r[2], r[2], __builtin_fma(r[1], r[1], r[0] * r[0])) * w;
                double r0 = chi, r1 = 1;
                if (robust) huber_rho_nb(hub, st, chi, r0, r1);
                acc[27] += r0;
                // linearizeOplus (.cpp:266-288, 335-367)
                const double X[3] = {(double)xw[j][0], (double)xw[j][1], (double)xw[j][2]};
                double Xc[3];
                po_pose_map(T, X, Xc);
                // (the Jacobian of .cpp:266-288 / 335-367 through a = x / z, b = y / z: 15 operations instead of 25 - the same quantities, products associated differently)
                const double x = Xc[0], y = Xc[1], invz = fast_rcp(Xc[2]), fx = in[0], fy = in[1], bf = in[4];
                const double a = x * invz, b = y * invz, ab = a * b, fxz = invz * fx, fyz = invz * fy, bz2 = bf * invz * invz;
                double J[18];
                J[0] = ab * fx; J[1] = -(__builtin_fma(a, a, 1.0) * fx); J[2] = b * fx; J[3] = -fxz; J[4] = 0; J[5] = a * fxz;
                J[6] = __builtin_fma(b, b, 1.0) * fy; J[7] = -(ab * fy); J[8] = -(a * fy); J[9] = 0; J[10] = -fyz; J[11] = b * fyz;
                J[12] = __builtin_fma(-bz2, y, J[0]); J[13] = __builtin_fma(bz2, x, J[1]); J[14] = J[2]; J[15] = J[3]; J[16] = 0; J[17] = J[5] - bz2;
                // rows 0, 1 and - for a stereo edge - 2, summed in that order from 0 like the reference's loop over the
                // error dimension; everything unrolled so that J / acc are registers (a runtime row count puts J into
                // scratch memory and costs ~40k cycles per edge).
                // J[4], J[9] and J[16] are exact zeros: a product with one of them is +-0.0, and adding +-0.0 to a sum that starts at +0.0 leaves its
                // bits alone (finite operands) - those 33 of the 81 product terms are not computed (known at compile time: the loops are unrolled).
                // The 48 (mono) / 75 (stereo) products go into the thread's sums as fused multiply-adds - one instruction per product instead of a multiply and
                // up to two additions: this loop is a third of the call, and it is bound by instruction issue.  (Rounded differently in the last bits than
                // g2o's Eigen expression; the estimates are held to 1e-5, not to bits.)
                const double W = r1 * w;
                double JW[12];
#pragma unroll
                for (int q = 0; q < 12; q++) JW[q] = J[q] * W;
                {
                    int k = 0;
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int jj = i; jj < 6; jj++, k++) {
                            if (i != 4 && jj != 4) acc[k] = __builtin_fma(JW[i], J[jj], acc[k]);
                            if (i != 3 && jj != 3) acc[k] = __builtin_fma(JW[6 + i], J[6 + jj], acc[k]);
                        }
                }
                const double c0 = -w * r[0] * r1, c1 = -w * r[1] * r1;
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    if (i != 4) acc[21 + i] = __builtin_fma(J[i], c0, acc[21 + i]);
                    if (i != 3) acc[21 + i] = __builtin_fma(J[6 + i], c1, acc[21 + i]);
                }
                if (st) {
                    double JW2[6];
#pragma unroll
                    for (int q = 0; q < 6; q++) JW2[q] = J[12 + q] * W;
                    int k = 0;
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int jj = i; jj < 6; jj++, k++)
                            if (i != 4 && jj != 4) acc[k] = __builtin_fma(JW2[i], J[12 + jj], acc[k]);
                    const double c2 = -w * r[2] * r1;
#pragma unroll
                    for (int i = 0; i < 6; i++)
                        if (i != 4) acc[21 + i] = __builtin_fma(J[12 + i], c2, acc[21 + i]);
                }
            }
            PO_STAMP(2);      // build: errors, Jacobians, the thread's 28 sums
            po_block_reduce28(acc, redT, red, sH, sb, tid);
            PO_STAMP(3);      // reduce28 (H, b and the chi2 arrive where the solve reads them: the stage that copied them and its barrier are gone)
            const double iniChi = red[0][27];
            int qmax = 0;
            PO_STAMP(4);
            do {
                if (tid == 0) {
                    if (qmax == 0) {
                        sCur = iniChi;
                        if (it == 0) {   // computeLambdaInit (:166-180)
                            double mx = 0;
#pragma unroll
                            for (int q = 0; q < 6; q++) mx = fmax(mx, fabs(sH[7 * q]));
                            sLambda = 1e-5 * mx; sNi = 2;
                        }
                    }
                    savePose = pose;   // push()
                    // (H + lambda I) x = b by LDL^T; "isPositive" like LinearSolverDense (linear_solver_dense.h:99-103)
                    // every loop fully unrolled, no early exit: A / Dg / xx stay in registers (with runtime indices they live in
                    // scratch memory and this one-thread solve costs ~40 us per trial).  After a failed pivot the remaining
                    // arithmetic runs on garbage, exactly the values the reference never looks at (ok == false).
                    double A[36];
#pragma unroll
                    for (int i = 0; i < 36; i++) A[i] = sH[i];
#pragma unroll
                    for (int i = 0; i < 6; i++) A[7 * i] += sLambda;
                    double Dg[6], Di[6];   // pivots and their reciprocals (6 divisions instead of 21 on this one-thread chain)
                    bool ok = true;
#pragma unroll
                    // (fused multiply-adds with the row's L D products formed once: on ONE lane every FP64 instruction of this section costs ~8 cycles, dependent or
                    // not, and every trial waits for it.  Measured against it: the same system by 3x3 blocks with cofactor inverses - two reciprocals on the chain
                    // instead of six, ~150 instead of ~160 instructions -: 1780 cycles against 1650, no gain; lanes as rows with v_readlane broadcasts: priced at
                    // ~1300, the broadcasts and selects eat what the shorter instruction stream saves.)
                    for (int j = 0; j < 6; j++) {
                        double LD[6];      // L[j][k] D[k], k < j
                        double dj = A[7 * j];
#pragma unroll
                        for (int k = 0; k < j; k++) { LD[k] = A[6 * j + k] * Dg[k]; dj = __builtin_fma(-A[6 * j + k], LD[k], dj); }
                        ok = ok && (dj > 0) && isfinite(dj);
                        Dg[j] = dj;
                        Di[j] = fast_rcp1(dj);
#pragma unroll
                        for (int i = j + 1; i < 6; i++) {
                            double lij = A[6 * i + j];
#pragma unroll
                            for (int k = 0; k < j; k++) lij = __builtin_fma(-A[6 * i + k], LD[k], lij);
                            A[6 * i + j] = lij * Di[j];
                        }
                    }
                    double xx[6];
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        double sacc = sb[i];
#pragma unroll
                        for (int k = 0; k < i; k++) sacc = __builtin_fma(-A[6 * i + k], xx[k], sacc);
                        xx[i] = sacc;
                    }
#pragma unroll
                    for (int i = 0; i < 6; i++) xx[i] *= Di[i];
#pragma unroll
                    for (int i = 5; i >= 0; i--) {
                        double sacc = xx[i];
#pragma unroll
                        for (int k = i + 1; k < 6; k++) sacc = __builtin_fma(-A[6 * k + i], xx[k], sacc);
                        xx[i] = sacc;
                    }
                    if (ok) {
#pragma unroll
                        for (int i = 0; i < 6; i++) sx[i] = xx[i];
                    }
                    sOk = ok ? 1 : 0;
                    if (PROF) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[5] += t_ - tPrev; pcnt[5] += 1; tPrev = t_; }      // 6x6 LDL^T solve (one thread)
                    pose_oplus_small(pose, sx);   // g2o applies the (possibly stale) x even when the solve failed; pop() restores
                    if (PROF) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[6] += t_ - tPrev; pcnt[6] += 1; tPrev = t_; }      // oplus (one thread)
                }
                __syncthreads();
                PO_STAMP(7);      // barrier behind the solve (the other waves wait here the whole time)
                const DPose T2 = pose;
                double cacc[1] = {0};
#pragma unroll
                for (int j = 0; j < NE; j++) {
                    const bool on = tid + 256 * j < n && !((outM >> j) & 1u);      // (no control flow per edge, as in the build)
                    const bool st = !(ob[j][2] < 0);
                    double r[3];
                    po_edge_error_nb(T2, in, xw[j], ob[j], st, r);
                    er[j][0] = on ? r[0] : er[j][0]; er[j][1] = on ? r[1] : er[j][1]; er[j][2] = on ? r[2] : er[j][2];
                    const double chi = __builtin_fma(r[2], r[2], __builtin_fma(r[1], r[1], r[0] * r[0])) * (double)is2[j];
                    double r0 = chi, r1 = 1;
                    if (robust) huber_rho_nb(hub, st, chi, r0, r1);
                    cacc[0] += on ? r0 : 0.0;
                }
                PO_STAMP(8);      // error pass at the trial pose
                po_block_reduce(cacc, red, tid);
                PO_STAMP(9);      // reduce (chi2)
                if (tid == 0) {
                    double tempChi = cacc[0];
                    if (!sOk) tempChi = 1.7976931348623157e308;
                    double rho = sCur - tempChi, scale = 0;
#pragma unroll
                    for (int j = 0; j < 6; j++) scale = __builtin_fma(sx[j], __builtin_fma(sLambda, sx[j], sb[j]), scale);
                    scale += 1e-3;
                    rho *= fast_rcp1(scale);      // (a few ulp from the division; rho's sign and 2 rho - 1 are what is used)
                    if (rho > 0 && isfinite(tempChi)) {
                        const double t2r = 2 * rho - 1;
                        double alpha = 1. - t2r * t2r * t2r;      // (pow(x, 3) in the reference; the library call costs ~0.5 us on this one-thread section)
                        alpha = fmin(alpha, 2. / 3.);
                        sLambda *= fmax(1. / 3., alpha);
                        sNi = 2;
                        sCur = tempChi;
                    } else {
                        sLambda *= sNi;
                        sNi *= 2;
                        pose = savePose;   // pop()
                    }
                    sRho = rho;
                    // the iteration's termination tests ride on the decision of its LAST trial (a section and a barrier of their own until round 6)
                    const int q1 = qmax + 1;
                    if (!(rho < 0 && q1 < 10)) {
                        if (q1 == 10 || rho == 0) sIterOk = 0;
                        else {
                            if ((iniChi - sCur) * 1e3 < iniChi) sNBad++; else sNBad = 0;
                            if (sNBad >= 3) sIterOk = 0;
                        }
                    }
                }
                __syncthreads();
                PO_STAMP(10);     // decision (one thread) + barrier
                qmax++;
            } while (sRho < 0 && qmax < 10);
            itersDone++;
            lastChi = sCur;
            PO_STAMP(11);     // end of iteration
        }
        if (tid == 0) { P.stats[8 * (size_t)f + 2 * round] = itersDone; P.stats[8 * (size_t)f + 2 * round + 1] = lastChi; }
        // ---- classification (:526-587): outliers are re-evaluated at the final pose, inliers keep their last _error
        const DPose Tf = pose;
        int nb = 0;
#pragma unroll
        for (int j = 0; j < NE; j++) {
            if (tid + 256 * j >= n) continue;
            const bool st = !(ob[j][2] < 0);
            if ((outM >> j) & 1u) {
                double r[3];
                po_edge_error(Tf, in, xw[j], ob[j], st, r);
                er[j][0] = r[0]; er[j][1] = r[1]; er[j][2] = r[2];
            }
            const float chi2 = (float)((er[j][0] * er[j][0] + er[j][1] * er[j][1] + er[j][2] * er[j][2]) * (double)is2[j]);
            const bool bad = chi2 > (st ? 7.815f : 5.991f);
            outM = bad ? (outM | (1u << j)) : (outM & ~(1u << j));
            nb += bad ? 1 : 0;
        }
        {
            double v[1] = {(double)nb};
            po_block_reduce(v, red, tid);
            nAct = n - (int)v[0];      // the next round's active edges (every thread holds the sum)
            if (tid == 0) P.ret[f] = nAct;
            __syncthreads();
        }
        PO_STAMP(12);     // classification of the round
        if (n < 10) break;   // optimizer.edges().size() < 10, :589-590
    }
#pragma unroll
    for (int j = 0; j < NE; j++)
        if (tid + 256 * j < n) outl[tid + 256 * j] = (uint8_t)((outM >> j) & 1u);      // pFrame->mvbOutlier
    if (tid == 0) {   // Converter::toCvMat(SE3Quat) + SetPose, :594-601
        double R[9];
        quat_to_R(pose.q, R);
        float *o = P.poseOut + 16 * (size_t)f;
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) o[4 * i + j] = (float)R[3 * i + j]; o[4 * i + 3] = (float)pose.t[i]; }
        o[12] = o[13] = o[14] = 0.f; o[15] = 1.f;
    }
    PO_STAMP(13);         // flags + pose out
    if (PROF && tid == 0 && P.prof) for (int i = 0; i < 16; i++) { atomicAdd(&P.prof[i], pacc[i]); atomicAdd(&P.prof[16 + i], pcnt[i]); }
    if (P.pubFlag) orbx_publish(P.pubCounter, P.pubFlag, P.pubSeq, gridDim.x);
}
#undef PO_STAMP

}  // namespace

struct orbx_lba {
    int device = 0, maxK = 0, maxP = 0, maxE = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    double flops = 0;
    OrbxDevBuf<DPose> pose, poseBak;
    OrbxDevBuf<double> pt, ptBak, intr, obs, info, err, rchi, edgeBlk, Hpp, bp, Hll, bl, Dinv, Ddb, S, bs, xp, xl, red;
    OrbxDevBuf<double> Lmat, ywork, ysol, diagInv;   // multi-workgroup Cholesky: the factor and the two halves of the right-hand side
    int numCU = 256;                                 // compute units of the device (residency of k_schur_rows)
    bool linSplit = false;                           // ORBX_LBA_SPLIT=1 (measurement switch): k_linearize, k_sum_points, k_sum_poses as separate launches
    OrbxDevBuf<int> ep, ek, ptStart, ptEdges, kfStart, kfEdges, poseIdx, ptIdx, okFlag;
    OrbxDevBuf<uint8_t> stereo, active;
    uint8_t *hostIO = nullptr;   // pinned: the marshalled inputs of a call on their way up, flags / chi2 / estimates on their way down
    size_t hostIOBytes = 0;
    OrbxDevBuf<uint8_t> flagDev, inArena, fixedDev;
    OrbxDevBuf<double> spPart;           // k_sum_poses: SP_SPLIT partial results per keyframe
    OrbxDevBuf<int> csrCnt, ptTmp, fillP, pActF, lActF, kfRowS0, kfRowN, ptPi;   // adjacency-list builder and stage preparation (device side)
    OrbxDevBuf<double> partChi, partL;   // per-workgroup partial sums of k_errors / k_backsub_update
    OrbxDevBuf<unsigned long long> Sacc; // fixed-point sums of the Schur complement (k_schur_rows -> k_schur_fin), all zero between trials
    OrbxDevBuf<double> bsPart;           // K x 32 x 6: the workgroups' shares of the reduced right-hand side
    OrbxDevBuf<unsigned long long> scaleBits;   // max |Hpp[r][r]| over the free poses as bit patterns, r = 0..5: the fixed-point scale (k_schur_setup)
    OrbxDevBuf<LmState> lm;              // Levenberg-Marquardt state of the running stage (device side)
    int *hostStop = nullptr, *hostStopDev = nullptr;   // pinned: the caller's stop flag as the device sees it (the waiting host keeps it current)
    double *hostRedDev = nullptr;        // device view of hostRed
    double seq = 0;                      // last sequence number handed out (k_stage_index, k_lm_begin, k_lm_decide)
    // test hook (ORBX_LBA_TEST_STOP_AFTER_DECISIONS=k, read per call): the CALLER's stop flag is raised - as another thread of the caller would -
    // when the host has seen the k-th trial decision of the call; from there on the library's own mirroring has to carry it to the device
    int dbgStopAfter = 0, dbgDecisions = 0;
    double *hostRed = nullptr;   // pinned: {chi, -, diag max, -, scale_p, scale_l, okFlag (as int)} of a trial, read back with ONE synchronisation
};

extern "C" int orbx_lba_create(int device, int max_keyframes, int max_points, int max_edges, orbx_lba **out)
{
    if (!out || max_keyframes < 1 || max_points < 1 || max_edges < 1) { orbx_set_error("bad LBA sizes"); return ORBX_ERR_ARG; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { orbx_set_error("no HIP device available: liborbx has no CPU fallback"); return ORBX_ERR_NODEVICE; }
    if (device < 0 || device >= ndev) { orbx_set_error("device %d out of range", device); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(device));
    orbx_lba *h = new orbx_lba();
    h->device = device; h->maxK = max_keyframes; h->maxP = max_points; h->maxE = max_edges;
    { const char *e = getenv("ORBX_LBA_SPLIT"); h->linSplit = e && e[0] == '1'; }
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) h->numCU = cu; }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; orbx_set_error("hipStreamCreate failed"); return ORBX_ERR_HIP; }
    (void)hipEventCreate(&h->ev0);
    (void)hipEventCreate(&h->ev1);
    if (hipHostMalloc((void **)&h->hostRed, 16 * sizeof(double), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&h->hostRedDev, h->hostRed, 0) != hipSuccess) { orbx_lba_destroy(h); orbx_set_error("hipHostMalloc (mapped) failed"); return ORBX_ERR_HIP; }
    for (int i = 0; i < 16; i++) h->hostRed[i] = 0;
    if (hipHostMalloc((void **)&h->hostStop, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&h->hostStopDev, h->hostStop, 0) != hipSuccess) { orbx_lba_destroy(h); orbx_set_error("hipHostMalloc (mapped) failed"); return ORBX_ERR_HIP; }
    h->hostStop[0] = 0;
    const size_t K = (size_t)max_keyframes, P = (size_t)max_points, E = (size_t)max_edges, n6 = 6 * K;
    int rc = 0;
    rc = rc ? rc : h->pose.ensure(K); rc = rc ? rc : h->poseBak.ensure(K); rc = rc ? rc : h->pt.ensure(3 * P); rc = rc ? rc : h->ptBak.ensure(3 * P);
    rc = rc ? rc : h->intr.ensure(5 * K); rc = rc ? rc : h->obs.ensure(3 * E); rc = rc ? rc : h->info.ensure(E); rc = rc ? rc : h->err.ensure(3 * E);
    rc = rc ? rc : h->rchi.ensure(E); rc = rc ? rc : h->edgeBlk.ensure(E * EB_SIZE); rc = rc ? rc : h->Hpp.ensure(36 * K); rc = rc ? rc : h->bp.ensure(n6);
    rc = rc ? rc : h->Hll.ensure(9 * P); rc = rc ? rc : h->bl.ensure(3 * P); rc = rc ? rc : h->Dinv.ensure(9 * P); rc = rc ? rc : h->Ddb.ensure(3 * P); rc = rc ? rc : h->ywork.ensure(n6); rc = rc ? rc : h->ysol.ensure(n6); rc = rc ? rc : h->diagInv.ensure(n6 + CNB);
    rc = rc ? rc : h->bs.ensure(n6); rc = rc ? rc : h->xp.ensure(n6); rc = rc ? rc : h->xl.ensure(3 * P); rc = rc ? rc : h->red.ensure(16);
    rc = rc ? rc : h->ep.ensure(E); rc = rc ? rc : h->ek.ensure(E); rc = rc ? rc : h->ptStart.ensure(P + 1); rc = rc ? rc : h->ptEdges.ensure(E);
    rc = rc ? rc : h->kfStart.ensure(K + 1); rc = rc ? rc : h->kfEdges.ensure(E); rc = rc ? rc : h->poseIdx.ensure(K); rc = rc ? rc : h->ptIdx.ensure(P);
    rc = rc ? rc : h->okFlag.ensure(1); rc = rc ? rc : h->stereo.ensure(E); rc = rc ? rc : h->active.ensure(E);
    rc = rc ? rc : h->spPart.ensure(K * SP_SPLIT * 27);
    rc = rc ? rc : h->kfRowS0.ensure(E); rc = rc ? rc : h->kfRowN.ensure(E); rc = rc ? rc : h->ptPi.ensure(E);
    rc = rc ? rc : h->fixedDev.ensure(K); rc = rc ? rc : h->ptTmp.ensure(E); rc = rc ? rc : h->fillP.ensure(P); rc = rc ? rc : h->pActF.ensure(K); rc = rc ? rc : h->lActF.ensure(P);
    rc = rc ? rc : h->lm.ensure(1);
    rc = rc ? rc : h->bsPart.ensure(K * 32 * 6); rc = rc ? rc : h->scaleBits.ensure(8);
    rc = rc ? rc : h->partChi.ensure((E + 255) / 256); rc = rc ? rc : h->partL.ensure((std::max(K, 16 * P) + 255) / 256);
    if (rc) { orbx_lba_destroy(h); return rc; }
    *out = h;
    return ORBX_OK;
}

extern "C" void orbx_lba_destroy(orbx_lba *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->pose.release(); h->poseBak.release(); h->pt.release(); h->ptBak.release(); h->intr.release(); h->obs.release(); h->info.release(); h->err.release();
    h->rchi.release(); h->edgeBlk.release(); h->Hpp.release(); h->bp.release(); h->Hll.release(); h->bl.release(); h->Dinv.release(); h->Ddb.release(); h->S.release(); h->Lmat.release(); h->ywork.release(); h->ysol.release(); h->diagInv.release();
    h->bs.release(); h->xp.release(); h->xl.release(); h->red.release(); h->ep.release(); h->ek.release(); h->ptStart.release(); h->ptEdges.release();
    h->kfStart.release(); h->kfEdges.release(); h->poseIdx.release(); h->ptIdx.release(); h->okFlag.release(); h->stereo.release(); h->active.release();
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->hostRed) (void)hipHostFree(h->hostRed);
    if (h->hostStop) (void)hipHostFree(h->hostStop);
    h->lm.release();
    if (h->hostIO) (void)hipHostFree(h->hostIO);
    h->flagDev.release(); h->partChi.release(); h->partL.release(); h->inArena.release(); h->fixedDev.release(); h->csrCnt.release(); h->ptTmp.release(); h->fillP.release(); h->pActF.release(); h->lActF.release(); h->kfRowS0.release(); h->kfRowN.release(); h->ptPi.release(); h->spPart.release();
    h->Sacc.release(); h->bsPart.release(); h->scaleBits.release();
    delete h;
}

namespace {

#define LCHECK()                                                                                                   \
    do {                                                                                                           \
        hipError_t e_ = hipGetLastError();                                                                         \
        if (e_ != hipSuccess) { orbx_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); return ORBX_ERR_HIP; } \
    } while (0)

// developer tap (tools/chol_phases.py; not part of include/orbx.h): a device array of 16 u64 the PROF instantiation of k_chol_step adds its phase cycles / passes to
static unsigned long long *g_cholProf = nullptr;

struct Ctx {
    orbx_lba *h;
    LbaDev d;
    Huber hub;
    int robust;
    int nPose, nPt;
    const volatile uint8_t *stop;
    const uint8_t *stageFlags = nullptr;   // device: outlier flags of the previous stage (level 1 edges), nullptr = every edge takes part
    int nPose0 = 0, nPt0 = 0;              // vertex counts of a stage without flags, known to the host from the row lengths
    int spSplit = 4;                       // workgroups per keyframe in k_lin_sums: ceil(longest keyframe row / 256), 1 .. SP_SPLIT
};

// Waits until the sequence number in pinned memory has reached `seq` (k_stage_index, k_lm_begin and k_lm_decide store theirs after their
// results; the numbers only grow).  The word is polled; meanwhile the caller's stop flag is copied into the pinned word the device reads,
// and the stream is queried now and then so that a failed launch surfaces as an error instead of a hang.
int wait_seq(orbx_lba *h, double seq, const volatile uint8_t *stop = nullptr)
{
    volatile double *flag = h->hostRed + 15;
    for (unsigned spins = 1;; spins++) {
        if (*flag >= seq) break;
        if (stop && *stop && !h->hostStop[0]) { h->hostStop[0] = 1; std::atomic_thread_fence(std::memory_order_release); }
        if ((spins & 0x3fff) == 0) {
            const hipError_t q = hipStreamQuery(h->stream);
            if (q == hipSuccess) {
                if (*flag >= seq) break;
                orbx_set_error("LBA: the stream drained without the trial results");
                return ORBX_ERR_HIP;
            }
            if (q != hipErrorNotReady) { orbx_set_error("LBA: %s", hipGetErrorString(q)); return ORBX_ERR_HIP; }
        }
        if (spins > 4096 && (spins & 63) == 0) std::this_thread::yield();      // a result normally arrives within ~300 us; do not starve other threads of an oversubscribed host
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return ORBX_OK;
}

// SparseOptimizer::optimize(iterations) on the edges of level 0 (see oracle/lba_oracle.cc for the CPU twin)
int optimize(Ctx &c, int iterations, double stats[4])
{
    orbx_lba *h = c.h;
    const int K = c.d.K, P = c.d.P, E = c.d.E;
    stats[0] = stats[1] = stats[2] = stats[3] = 0;
    // initializeOptimization(0): active edges / vertices and the index mapping (sparse_optimizer.cpp:166-267), on the device
    int nPose, nPt, nAct;
    {
        const double seq = (h->seq += 1.0);
        const int stamp = c.stageFlags ? 2 : 1;
        const unsigned gM = (unsigned)((E + 255) / 256);
        hipLaunchKernelGGL(k_stage_mark, dim3(gM), dim3(256), 0, h->stream, E, (const int *)h->ep.p, (const int *)h->ek.p, c.stageFlags, h->active.p, h->pActF.p, h->lActF.p, stamp,
                           h->csrCnt.p);      // (the chunk counters of the adjacency-list builder are free again: >= E / 256 entries)
        hipLaunchKernelGGL(k_stage_index, dim3(1), dim3(1024), 0, h->stream, K, P, (const uint8_t *)h->fixedDev.p, (const int *)h->pActF.p, (const int *)h->lActF.p, stamp,
                           h->poseIdx.p, h->ptIdx.p, (const int *)h->csrCnt.p, (int)gM, h->hostRedDev, seq);
        hipLaunchKernelGGL(k_stage_pairs, dim3(gM), dim3(256), 0, h->stream, E, (const int *)h->ek.p, (const int *)h->ptEdges.p, (const uint8_t *)h->active.p, (const int *)h->poseIdx.p,
                           h->ptPi.p);
        LCHECK();
        if (!c.stageFlags) { nPose = c.nPose0; nPt = c.nPt0; nAct = E; }      // nothing to wait for
        else {
            int rcw = wait_seq(h, seq, c.stop);
            if (rcw) return rcw;
            nPose = (int)h->hostRed[10]; nPt = (int)h->hostRed[11]; nAct = (int)h->hostRed[12];
        }
    }
    if (nAct == 0 || nPose + nPt == 0) return ORBX_OK;
    if (6 * nPose > CHOL_DENSE_MAX_N) { orbx_set_error("%d free keyframes exceed the dense reduced-system limit %d", nPose, CHOL_DENSE_MAX_N / 6); return ORBX_ERR_CAPACITY; }
    c.nPose = nPose; c.nPt = nPt;
    {   // the reduced system and its factor are sized by the FREE keyframes of this call (grow-only), checked before anything is allocated
        const size_t nn = (size_t)(6 * nPose) * (size_t)(6 * nPose);
        int rc = h->S.ensure(nn ? nn : 1);
        rc = rc ? rc : h->Lmat.ensure(nn ? nn : 1);
        rc = rc ? rc : h->Sacc.ensure(nn ? nn : 1);
        if (rc) return rc;
        // the fixed-point accumulator is all zero between trials (k_schur_fin clears what it reads); cleared here as well, so that a call
        // that was aborted between the two kernels leaves nothing behind
        ORBX_HIP_CHECK(hipMemsetAsync(h->Sacc.p, 0, (nn ? nn : 1) * sizeof(unsigned long long), h->stream));
    }
    if (iterations <= 0 || (c.stop && *c.stop)) return ORBX_OK;      // (the loop of sparse_optimizer.cpp:370 does not run)
    const int nP6 = 6 * nPose;
    const unsigned gE = (unsigned)((E + 255) / 256);
    LmState *st = h->lm.p;
    const double *lam = &st->lambda;
    const int *gDone = &st->done, *gLin = &st->relin;
    // What is reproducible to the bit: everything.  chi2, H and b are ordered partial sums, the Schur complement is accumulated in fixed point
    // (k_schur_rows), its right-hand side in ordered partial sums, the decisions are taken by one thread (k_lm_decide).
    auto linearize = [&]() -> int {      // H and b at the new estimates, for the next trial: only behind an accepted trial, and not when the stage is over
        if (!h->linSplit) {      // Jacobians inside the sums that consume them: one launch + the ordered add of the keyframe partials
            hipLaunchKernelGGL(k_lin_sums, dim3((unsigned)(K * c.spSplit + (P + 15) / 16)), dim3(256), 0, h->stream, c.d, c.hub, c.robust, h->ptStart.p, h->ptEdges.p, h->Hll.p,
                               h->bl.p, h->kfStart.p, h->kfEdges.p, h->spPart.p, c.spSplit, gLin, 1);
            LCHECK();      // (the ordered add of the keyframe partials is the first part of the next k_schur_setup)
            return ORBX_OK;
        }
        hipLaunchKernelGGL(k_linearize, dim3(gE), dim3(256), 0, h->stream, c.d, c.hub, c.robust, gLin, 1);
        LCHECK();
        hipLaunchKernelGGL(k_sum_points, dim3((unsigned)((P + 15) / 16)), dim3(256), 0, h->stream, c.d, h->ptStart.p, h->ptEdges.p, h->Hll.p, h->bl.p, gLin, 1);
        LCHECK();
        hipLaunchKernelGGL(k_sum_poses, dim3((unsigned)K, (unsigned)c.spSplit), dim3(256), 0, h->stream, c.d, h->kfStart.p, h->kfEdges.p, h->spPart.p, c.spSplit, gLin, 1);
        LCHECK();
        return ORBX_OK;
    };
    // ---- start of the stage: errors and chi2 of the start, H and b, computeLambdaInit, the LM state (nothing is waited for)
    h->hostStop[0] = 0;
    {
        // (the state's `done` of the previous stage is still set: the first launches are not gated)
        hipLaunchKernelGGL(k_errors, dim3(gE), dim3(256), 0, h->stream, c.d, c.hub, c.robust, h->partChi.p, (const int *)nullptr, 0);
        LCHECK();
        if (!h->linSplit) {
            hipLaunchKernelGGL(k_lin_sums, dim3((unsigned)(K * c.spSplit + (P + 15) / 16)), dim3(256), 0, h->stream, c.d, c.hub, c.robust, h->ptStart.p, h->ptEdges.p, h->Hll.p,
                               h->bl.p, h->kfStart.p, h->kfEdges.p, h->spPart.p, c.spSplit, (const int *)nullptr, 0);
        } else {
            hipLaunchKernelGGL(k_linearize, dim3(gE), dim3(256), 0, h->stream, c.d, c.hub, c.robust, (const int *)nullptr, 0);
            hipLaunchKernelGGL(k_sum_points, dim3((unsigned)((P + 15) / 16)), dim3(256), 0, h->stream, c.d, h->ptStart.p, h->ptEdges.p, h->Hll.p, h->bl.p, (const int *)nullptr, 0);
            hipLaunchKernelGGL(k_sum_poses, dim3((unsigned)K, (unsigned)c.spSplit), dim3(256), 0, h->stream, c.d, h->kfStart.p, h->kfEdges.p, h->spPart.p, c.spSplit, (const int *)nullptr, 0);
        }
        hipLaunchKernelGGL(k_sum_poses_fin, dim3((unsigned)((32 * K + 255) / 256)), dim3(256), 0, h->stream, c.d, (const double *)h->spPart.p, h->Hpp.p, h->bp.p, c.spSplit, (const int *)nullptr, 0);
        hipLaunchKernelGGL(k_diag_max, dim3(1), dim3(1024), 0, h->stream, h->Hpp.p, nPose, h->Hll.p, nPt, h->red.p);
        const double seq = (h->seq += 1.0);
        hipLaunchKernelGGL(k_lm_begin, dim3(1), dim3(256), 0, h->stream, st, (const double *)h->partChi.p, (int)gE, (const double *)h->red.p, iterations, (const volatile int *)h->hostStopDev,
                           h->hostRedDev, seq, h->scaleBits.p);
        LCHECK();
    }
    // ---- one Levenberg trial in two halves
    // the right-hand side of the reduced system is accumulated where the solver updates it in place (multi-workgroup Cholesky: ywork)
    double *const bsDev = nP6 >= CHOL_MULTI_MIN_N ? h->ywork.p : h->bs.p;
    // first half of a trial: the reduced system S, bs from H, b and lambda (three launches)
    auto trialSchur = [&]() -> int {
        {
            const int nInit = nP6 > 0 ? (32 * K + 255) / 256 : 0;      // the first workgroups: Hpp / b_p from the keyframe partial sums, and the fixed-point scale
            hipLaunchKernelGGL(k_schur_setup, dim3((unsigned)(nInit + (P + 3) / 4 + (nP6 > 0 ? (E + 255) / 256 : 0))), dim3(256), 0, h->stream, c.d, nInit, h->Hpp.p, h->bp.p, nPose, h->ptStart.p,
                               h->ptEdges.p, h->Hll.p, h->bl.p, 0.0, h->S.p, bsDev, h->Dinv.p, h->Ddb.p, h->okFlag.p, lam, h->scaleBits.p, (const double *)h->spPart.p, c.spSplit, h->Hpp.p, h->bp.p,
                               gDone, 0);
            LCHECK();
        }
        if (nP6 > 0) {
            const size_t ldsRows = (size_t)(6 * nP6) * sizeof(unsigned long long);
            // y-splits per keyframe so that ALL workgroups are resident at once (six per CU at the kernel's 80 registers, fewer where the LDS row is
            // long): a step of the edge loop is three dependent memory round trips, ~5 us, and a second round of workgroups costs a whole one
            const int perCU = std::max(1, std::min(6, (int)(150 * 1024 / std::max<size_t>(ldsRows, 1))));
            const int ySplit = std::max(4, std::min(32, perCU * h->numCU / std::max(K, 1)));
            int ys = ySplit;
            if (ldsRows > 150 * 1024) {      // > ~530 free keyframes: the block row no longer fits into LDS
                ys = 16;
                hipLaunchKernelGGL(k_schur_rows<true>, dim3((unsigned)K, 16u), dim3(256), 0, h->stream, c.d, h->kfStart.p, h->kfEdges.p, (const int *)h->kfRowS0.p, (const int *)h->kfRowN.p, h->ptEdges.p, (const int *)h->ptPi.p, nP6, h->Ddb.p, h->Sacc.p, h->bsPart.p, (const unsigned long long *)h->scaleBits.p, gDone, 0);
            } else {
                if (ldsRows > 48 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_schur_rows<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsRows));
                hipLaunchKernelGGL(k_schur_rows<false>, dim3((unsigned)K, (unsigned)ySplit), dim3(256), ldsRows, h->stream, c.d, h->kfStart.p, h->kfEdges.p, (const int *)h->kfRowS0.p, (const int *)h->kfRowN.p, h->ptEdges.p, (const int *)h->ptPi.p, nP6, h->Ddb.p, h->Sacc.p, h->bsPart.p, (const unsigned long long *)h->scaleBits.p, gDone, 0);
            }
            LCHECK();
            // S and bs from the sums (and the accumulator cleared for the next trial)
            hipLaunchKernelGGL(k_schur_fin, dim3((unsigned)((nP6 + 255) / 256), (unsigned)(nPose + 1)), dim3(256), 0, h->stream, c.d, (const double *)h->Hpp.p, (const double *)h->bp.p, nPose, 0.0,
                               lam, h->Sacc.p, (const double *)h->bsPart.p, ys, (const unsigned long long *)h->scaleBits.p, h->S.p, bsDev, gDone, 0);
            LCHECK();
        }
        return ORBX_OK;
    };
    // second half: factorisation and solve, update, errors, the decision, and - for the next trial - the restore of a rejected update and H, b
    // at the estimates the decision left.  *seqOut = the decision's sequence number.
    auto trialChol = [&]() -> int {
        if (nP6 > 0) {
            if (nP6 >= CHOL_MULTI_MIN_N) {
                const int n = nP6;
                for (int p0 = 0; p0 < n; p0 += CNB) {
                    const int nb = std::min(CNB, n - p0), below = n - p0 - nb;
                    const int nPW = std::max(1, (below + CHOL_RPW - 1) / CHOL_RPW);
                    const int T1 = p0 > 0 ? (n - p0 + CNB - 1) / CNB - 1 : 0;        // tile rows / columns beyond block column p that still await the previous panel's update
                    if (g_cholProf) hipLaunchKernelGGL(k_chol_step<true>, dim3((unsigned)(nPW + T1 * T1)), dim3(256), 0, h->stream, h->S.p, h->Lmat.p, n, p0, nPW, std::max(T1, 1), h->ywork.p,
                                                       h->ysol.p, h->okFlag.p, h->diagInv.p, gDone, 0, g_cholProf);
                    else hipLaunchKernelGGL(k_chol_step<false>, dim3((unsigned)(nPW + T1 * T1)), dim3(256), 0, h->stream, h->S.p, h->Lmat.p, n, p0, nPW, std::max(T1, 1), h->ywork.p,
                                            h->ysol.p, h->okFlag.p, h->diagInv.p, gDone, 0, (unsigned long long *)nullptr);
                }
                if (n <= 128) hipLaunchKernelGGL(k_chol_backsub_reg<4>, dim3(1), dim3(1024), 0, h->stream, h->Lmat.p, h->ysol.p, h->diagInv.p, n, h->xp.p, gDone, 0);
                else if (n <= 224) hipLaunchKernelGGL(k_chol_backsub_reg<7>, dim3(1), dim3(1024), 0, h->stream, h->Lmat.p, h->ysol.p, h->diagInv.p, n, h->xp.p, gDone, 0);
                else if (n <= 256) hipLaunchKernelGGL(k_chol_backsub_reg<8>, dim3(1), dim3(1024), 0, h->stream, h->Lmat.p, h->ysol.p, h->diagInv.p, n, h->xp.p, gDone, 0);
                else if (n <= 320) hipLaunchKernelGGL(k_chol_backsub_reg<10>, dim3(1), dim3(1024), 0, h->stream, h->Lmat.p, h->ysol.p, h->diagInv.p, n, h->xp.p, gDone, 0);
                else if (n <= CHOL_LDS_X) hipLaunchKernelGGL(k_chol_backsub<false>, dim3(1), dim3(1024), 0, h->stream, h->Lmat.p, h->ysol.p, n, h->xp.p, gDone, 0);
                else hipLaunchKernelGGL(k_chol_backsub<true>, dim3(1), dim3(1024), 0, h->stream, h->Lmat.p, h->ysol.p, n, h->xp.p, gDone, 0);
            } else {   // widest panel whose n x NB doubles fit next to the solution vector in LDS
                const size_t budget = 120 * 1024;
                if ((size_t)nP6 * 32 * 8 <= budget) {
                    const size_t lds = (size_t)nP6 * 32 * 8;
                    if (lds > 32 * 1024) ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_chol_solve<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(k_chol_solve<32>, dim3(1), dim3(1024), lds, h->stream, h->S.p, h->bs.p, nP6, h->xp.p, h->okFlag.p, gDone, 0);
                } else if ((size_t)nP6 * 16 * 8 <= budget) {
                    const size_t lds = (size_t)nP6 * 16 * 8;
                    ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_chol_solve<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(k_chol_solve<16>, dim3(1), dim3(1024), lds, h->stream, h->S.p, h->bs.p, nP6, h->xp.p, h->okFlag.p, gDone, 0);
                } else {
                    const size_t lds = (size_t)nP6 * 4 * 8;
                    ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_chol_solve<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(k_chol_solve<4>, dim3(1), dim3(1024), lds, h->stream, h->S.p, h->bs.p, nP6, h->xp.p, h->okFlag.p, gDone, 0);
                }
            }
            LCHECK();
        }
        return ORBX_OK;
    };
    auto trialRest = [&](double *seqOut) -> int {
        // (push() happens inside k_backsub_update, right before the estimates are changed)
        const unsigned gU = (unsigned)((std::max(K, 16 * P) + 255) / 256);
        hipLaunchKernelGGL(k_backsub_update, dim3(gU), dim3(256), 0, h->stream, c.d, h->ptStart.p, h->ptEdges.p, (const int *)h->ptPi.p, h->bl.p,
                           h->Dinv.p, h->xp.p, h->xl.p, h->poseBak.p, h->ptBak.p, 0.0, h->partL.p, lam, gDone, 0);
        LCHECK();
        const double seq = (h->seq += 1.0);
        // ORBX_LBA_MERGE_DECIDE=1 (measured, round 6: profiles/r06_lba_merge_decide.txt): one launch instead of two, nine launches less per window - and 14.7 us per
        // trial instead of 4.9 + 6.4 + a kernel boundary: the last arriver's fences cost what the boundary did.  Not the default.
        static const bool mergeDecide = getenv("ORBX_LBA_MERGE_DECIDE") && getenv("ORBX_LBA_MERGE_DECIDE")[0] == '1';
        if (mergeDecide) {
            hipLaunchKernelGGL(k_errors_decide, dim3(gE), dim3(256), 0, h->stream, c.d, c.hub, c.robust, h->partChi.p, st, (const double *)h->partL.p, (int)gU, (const double *)h->xp.p,
                               (const double *)h->bp.p, nP6, nP6 > 0 ? (const int *)h->okFlag.p : (const int *)nullptr, (const volatile int *)h->hostStopDev, h->hostRedDev, seq,
                               (const DPose *)h->poseBak.p, (const double *)h->ptBak.p, h->scaleBits.p);
            LCHECK();
        } else {
            hipLaunchKernelGGL(k_errors, dim3(gE), dim3(256), 0, h->stream, c.d, c.hub, c.robust, h->partChi.p, gDone, 0);
            LCHECK();
            hipLaunchKernelGGL(k_lm_decide, dim3(1), dim3(256), 0, h->stream, st, (const double *)h->partChi.p, (int)gE, (const double *)h->partL.p, (int)gU, (const double *)h->xp.p,
                               (const double *)h->bp.p, nP6, nP6 > 0 ? (const int *)h->okFlag.p : (const int *)nullptr, (const volatile int *)h->hostStopDev, h->hostRedDev, seq, c.d,
                               (const DPose *)h->poseBak.p, (const double *)h->ptBak.p, h->scaleBits.p);
            LCHECK();
        }
        // (a rejected update is taken back inside k_lm_decide); H and b at the new estimates behind an accepted trial
        int rcl = linearize();
        if (rcl) return rcl;
        h->flops += 400.0 * nAct + 250.0 * nAct + (double)nP6 * nP6 * nP6 / 3.0;
        *seqOut = seq;
        return ORBX_OK;
    };
    // The host never lets the device wait for it: behind the decision of trial t it has already queued the restore / errors / linearisation and
    // the Schur half of trial t + 1 (~40 us of device work) when it starts to wait for that decision, and queues the solve half of t + 1 while
    // those run.  A stage that ends leaves those seven launches to fall through (~4 us each); a whole trial enqueued in advance would cost
    // twenty (measured: every trial of both stages up front, 15 per window where 9 are needed: 2.08 ms per window instead of 1.87).
    // ORBX_LBA_SPEC_CHOL=1 (measured): the factorisation of trial t + 1 is queued speculatively too, so that the device never waits for
    // the host's nine panel launches behind a decision; a stage that ends then lets nine more gated launches fall through.
    static const bool specChol = getenv("ORBX_LBA_SPEC_CHOL") && getenv("ORBX_LBA_SPEC_CHOL")[0] == '1';
    int rct = trialSchur();
    if (rct) return rct;
    if (specChol && (rct = trialChol()) != ORBX_OK) return rct;
    for (int t = 0;; t++) {
        double seqT = 0;
        if (!specChol && (rct = trialChol()) != ORBX_OK) return rct;
        if ((rct = trialRest(&seqT)) != ORBX_OK) return rct;
        if ((rct = trialSchur()) != ORBX_OK) return rct;      // of trial t + 1, speculatively
        if (specChol && (rct = trialChol()) != ORBX_OK) return rct;
        int rcw = wait_seq(h, seqT, c.stop);
        if (rcw) return rcw;
        if (h->dbgStopAfter > 0 && c.stop && ++h->dbgDecisions == h->dbgStopAfter) *const_cast<volatile uint8_t *>(c.stop) = 1;      // (test hook, see the handle)
        if (h->hostRed[13] != 0.0) break;      // done
        if (t >= 10 * iterations + 10) { orbx_set_error("LBA: the Levenberg loop did not terminate"); return ORBX_ERR_STATE; }
    }
    stats[0] = h->hostRed[7]; stats[1] = h->hostRed[9]; stats[2] = h->hostRed[6]; stats[3] = h->hostRed[1];
    return ORBX_OK;
}

}  // namespace

// iters1 LM iterations (Huber kernels when robust1); with secondStage the LocalBundleAdjustment continuation (outlier
// classification + 10 iterations without kernels on the inliers)
static int lba_run(orbx_lba *h, const orbx_lba_problem *p, const volatile uint8_t *stop, orbx_lba_result *res, int iters1, bool robust1, bool secondStage)
{
    if (!h || !p || !res || !res->poses || !res->points || !res->edge_outlier) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    const int K = p->num_keyframes, P = p->num_points, E = p->num_edges;
    if (K < 1 || P < 1 || E < 1 || K > h->maxK || P > h->maxP || E > h->maxE) { orbx_set_error("problem size %d/%d/%d outside the handle's capacity %d/%d/%d", K, P, E, h->maxK, h->maxP, h->maxE); return ORBX_ERR_CAPACITY; }
    if (!p->poses || !p->fixed || !p->intrinsics || !p->points || !p->edge_point || !p->edge_keyframe || !p->edge_obs || !p->edge_inv_sigma2) { orbx_set_error("NULL problem array"); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(h->device));
    for (int i = 0; i < 8; i++) res->stats[i] = 0;
    h->flops = 0;
    { const char *e = getenv("ORBX_LBA_TEST_STOP_AFTER_DECISIONS"); h->dbgStopAfter = e && *e ? atoi(e) : 0; h->dbgDecisions = 0; }
    // ---- host marshalling: float boundary -> double state (Converter::toSE3Quat / toVector3d), written straight into ONE pinned
    // buffer (copies from pageable vectors are staged and synchronous: a dozen of them cost 0.3 ms of a 5 ms call) ----
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t oPose = 0, oIntr = oPose + pad((size_t)K * sizeof(DPose)), oPt = oIntr + pad((size_t)5 * K * 8), oObs = oPt + pad((size_t)3 * P * 8),
                 oInfo = oObs + pad((size_t)3 * E * 4), oEp = oInfo + pad((size_t)E * 4), oEk = oEp + pad((size_t)E * 4),      // observations / information as floats
                 oPs = oEk + pad((size_t)E * 4), oKs = oPs + pad(((size_t)P + 1) * 4), oFx = oKs + pad(((size_t)K + 1) * 4), inBytes = oFx + pad((size_t)K);
    const size_t dFlag = 0, dChi = dFlag + pad((size_t)E), dPose = dChi + pad((size_t)E * 8), dPt = dPose + pad((size_t)K * sizeof(DPose)),
                 outBytes = dPt + pad((size_t)3 * P * 8);
    const int nChunk = (E + CSR_CHUNK - 1) / CSR_CHUNK;
    const size_t csrLds = (size_t)K * sizeof(int);
    if (csrLds > 150 * 1024) { orbx_set_error("%d keyframes exceed the adjacency-list builder's limit %d", K, 150 * 1024 / 4); return ORBX_ERR_CAPACITY; }
    {
        const size_t need = std::max(inBytes, outBytes);
        if (need > h->hostIOBytes) {
            ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
            if (h->hostIO) (void)hipHostFree(h->hostIO);
            h->hostIO = nullptr; h->hostIOBytes = 0;
            ORBX_HIP_CHECK(hipHostMalloc((void **)&h->hostIO, need, hipHostMallocDefault));
            h->hostIOBytes = need;
        }
        int rcb = h->flagDev.ensure(outBytes);
        rcb = rcb ? rcb : h->inArena.ensure(inBytes);
        rcb = rcb ? rcb : h->csrCnt.ensure((size_t)nChunk * (size_t)K);
        if (rcb) return rcb;
    }
    // The host only converts (float boundary -> double state) and counts; everything goes up in ONE copy of the pinned buffer, one
    // kernel distributes it, and the adjacency lists and the index mapping of the stages are made on the device.  Nothing waits for
    // the copy: the results come back into the front of the same buffer long after the device has consumed it.
    uint8_t *io = h->hostIO;
    DPose *pose = (DPose *)(io + oPose);
    double *intr = (double *)(io + oIntr), *pt = (double *)(io + oPt);
    uint8_t *fixedH = io + oFx;
    int *epH = (int *)(io + oEp), *ekH = (int *)(io + oEk), *ptStart = (int *)(io + oPs), *kfStart = (int *)(io + oKs);
    for (int k = 0; k < K; k++) {
        double R[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = p->poses[16 * (size_t)k + 4 * i + j];
        pose[k].q = quat_from_R(R);
        quat_normalize_pos(pose[k].q);
        for (int i = 0; i < 3; i++) pose[k].t[i] = p->poses[16 * (size_t)k + 4 * i + 3];
        for (int i = 0; i < 5; i++) intr[5 * (size_t)k + i] = p->intrinsics[5 * (size_t)k + i];
        fixedH[k] = p->fixed[k] ? 1 : 0;
    }
    for (int i = 0; i < 3 * P; i++) pt[i] = p->points[i];
    for (int l = 0; l <= P; l++) ptStart[l] = 0;
    for (int k = 0; k <= K; k++) kfStart[k] = 0;
    // the edge arrays travel as they are (float observations / information, int ids): four memcpy; the device converts and derives the stereo flags
    // (k_unpack).  The host only checks the ids and counts the row lengths of the adjacency lists.
    memcpy(io + oObs, p->edge_obs, (size_t)3 * E * sizeof(float));
    memcpy(io + oInfo, p->edge_inv_sigma2, (size_t)E * sizeof(float));
    memcpy(epH, p->edge_point, (size_t)E * sizeof(int));
    memcpy(ekH, p->edge_keyframe, (size_t)E * sizeof(int));
    for (int e = 0; e < E; e++) {
        const int l = epH[e], k = ekH[e];
        if (l < 0 || l >= P || k < 0 || k >= K) { orbx_set_error("edge %d references a vertex out of range", e); return ORBX_ERR_ARG; }
        ptStart[l + 1]++; kfStart[k + 1]++;
    }
    Ctx c;
    c.h = h; c.stop = stop;
    for (int l = 0; l < P; l++) { c.nPt0 += ptStart[l + 1] > 0; ptStart[l + 1] += ptStart[l]; }                       // row starts; vertices with an edge
    {
        int longest = 0;
        for (int k = 0; k < K; k++) longest = std::max(longest, kfStart[k + 1]);
        c.spSplit = std::max(1, std::min(SP_SPLIT, (longest + 255) / 256));
    }
    for (int k = 0; k < K; k++) { c.nPose0 += kfStart[k + 1] > 0 && !fixedH[k]; kfStart[k + 1] += kfStart[k]; }
    hipStream_t s = h->stream;
    ORBX_HIP_CHECK(hipEventRecord(h->ev0, s));
    {
        ORBX_HIP_CHECK(hipMemcpyAsync(h->inArena.p, io, inBytes, hipMemcpyHostToDevice, s));
        UnpackSegs sg;
        int ns = 0;
        auto seg = [&](size_t off, void *dst, size_t bytes, int kind = 0) { sg.src[ns] = off; sg.dst[ns] = dst; sg.bytes[ns] = bytes; sg.kind[ns] = kind; ns++; };
        seg(oPose, h->pose.p, (size_t)K * sizeof(DPose)); seg(oPt, h->pt.p, (size_t)3 * P * 8); seg(oIntr, h->intr.p, (size_t)5 * K * 8);
        seg(oObs, h->obs.p, (size_t)3 * E, 1); seg(oInfo, h->info.p, (size_t)E, 1); seg(oObs, h->stereo.p, (size_t)E, 2);
        seg(oEp, h->ep.p, (size_t)E * 4); seg(oEk, h->ek.p, (size_t)E * 4);
        seg(oPs, h->ptStart.p, ((size_t)P + 1) * 4); seg(oKs, h->kfStart.p, ((size_t)K + 1) * 4); seg(oFx, h->fixedDev.p, (size_t)K);
        seg(~(size_t)0, h->err.p, (size_t)E * 3 * 8);
        seg(~(size_t)0, h->fillP.p, (size_t)P * 4); seg(~(size_t)0, h->pActF.p, (size_t)K * 4); seg(~(size_t)0, h->lActF.p, (size_t)P * 4);
        sg.n = ns;
        hipLaunchKernelGGL(k_unpack, dim3(256), dim3(256), 0, s, (const uint8_t *)h->inArena.p, sg);
        LCHECK();
        // adjacency lists
        if (csrLds > 48 * 1024) {
            ORBX_HIP_CHECK(hipFuncSetAttribute((const void *)k_csr_kf_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)csrLds));
        }
        const unsigned gE = (unsigned)((E + 255) / 256);
        hipLaunchKernelGGL(k_csr_kf_count, dim3((unsigned)nChunk), dim3(256), csrLds, s, (const int *)h->ek.p, E, K, h->csrCnt.p);
        hipLaunchKernelGGL(k_csr_kf_scan, dim3((unsigned)K), dim3(256), 0, s, (const int *)h->kfStart.p, K, nChunk, h->csrCnt.p);
        hipLaunchKernelGGL(k_csr_kf_fill, dim3((unsigned)nChunk), dim3(256), 0, s, (const int *)h->ek.p, E, K, (const int *)h->csrCnt.p, h->kfEdges.p);
        hipLaunchKernelGGL(k_csr_pt_fill, dim3(gE), dim3(256), 0, s, (const int *)h->ep.p, E, (const int *)h->ptStart.p, h->fillP.p, h->ptTmp.p);
        hipLaunchKernelGGL(k_csr_pt_rank, dim3(gE), dim3(256), 0, s, (const int *)h->ep.p, E, (const int *)h->ptStart.p, (const int *)h->ptTmp.p, h->ptEdges.p);
        hipLaunchKernelGGL(k_csr_rows, dim3(gE), dim3(256), 0, s, E, (const int *)h->ep.p, (const int *)h->kfEdges.p, (const int *)h->ptStart.p, h->kfRowS0.p, h->kfRowN.p);
        LCHECK();
    }
    LbaDev &d = c.d;
    d.K = K; d.P = P; d.E = E; d.pose = h->pose.p; d.pt = h->pt.p; d.intr = h->intr.p; d.ep = h->ep.p; d.ek = h->ek.p; d.obs = h->obs.p;
    d.stereo = h->stereo.p; d.info = h->info.p; d.active = h->active.p; d.poseIdx = h->poseIdx.p; d.ptIdx = h->ptIdx.p; d.err = h->err.p;
    d.rchi = h->rchi.p; d.edgeBlk = h->edgeBlk.p;
    // float deltas; LocalBundleAdjustment uses sqrt(5.991) (src/Optimizer.cc:764-765), BundleAdjustment sqrt(5.99) (:141-142)
    const float thMono = (float)sqrt(secondStage ? 5.991 : 5.99), thStereo = (float)sqrt(7.815);
    c.hub.dMono = thMono; c.hub.dStereo = thStereo;
    c.hub.dsqrMono = (double)(float)((double)thMono * (double)thMono);        // `float dsqr` member (robust_kernel_impl.h:84)
    c.hub.dsqrStereo = (double)(float)((double)thStereo * (double)thStereo);
    // classification on the device (k_classify); only the flags come back between the stages, the rest with the final results
    const unsigned gEc = (unsigned)((std::max(E, std::max(K, 3 * P)) + 255) / 256);
    uint8_t *oa = h->flagDev.p;     // device image of the result layout dFlag | dChi | dPose | dPt
    int rc = ORBX_OK;
    if (!(stop && *stop)) {
        c.robust = robust1 ? 1 : 0;
        if ((rc = optimize(c, iters1, res->stats)) != ORBX_OK) return rc;       // :863-864 (LBA), :247 (BundleAdjustment)
        if (secondStage && !(stop && *stop)) {
            // :880-912: the outlier flags stay on the device, the second stage starts from them (k_stage_prep)
            hipLaunchKernelGGL(k_classify, dim3(gEc), dim3(256), 0, h->stream, c.d, oa + dFlag, (double *)nullptr, (DPose *)nullptr, (double *)nullptr);
            LCHECK();
            c.stageFlags = oa + dFlag;
            c.robust = 0;
            if ((rc = optimize(c, 10, res->stats + 4)) != ORBX_OK) return rc;   // :916-917
        }
    }
    ORBX_HIP_CHECK(hipEventRecord(h->ev1, s));
    h->timed = true;
    // :921-958: final classification, chi2 and estimates in one kernel and one copy
    hipLaunchKernelGGL(k_classify, dim3(gEc), dim3(256), 0, h->stream, c.d, oa + dFlag, res->edge_chi2 ? (double *)(oa + dChi) : (double *)nullptr, (DPose *)(oa + dPose),
                       (double *)(oa + dPt));
    LCHECK();
    ORBX_HIP_CHECK(hipMemcpyAsync(io, oa, outBytes, hipMemcpyDeviceToHost, h->stream));
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    memcpy(res->edge_outlier, io + dFlag, (size_t)E);
    if (res->edge_chi2) memcpy(res->edge_chi2, io + dChi, (size_t)E * 8);
    pose = (DPose *)(io + dPose); pt = (double *)(io + dPt);                  // final estimates (pinned read-back)
    for (int k = 0; k < K; k++) {                                             // Converter::toCvMat(SE3Quat)
        double R[9];
        quat_to_R(pose[k].q, R);
        float *o = res->poses + 16 * (size_t)k;
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) o[4 * i + j] = (float)R[3 * i + j]; o[4 * i + 3] = (float)pose[k].t[i]; }
        o[12] = o[13] = o[14] = 0.f; o[15] = 1.f;
    }
    for (int i = 0; i < 3 * P; i++) res->points[i] = (float)pt[i];
    return ORBX_OK;
}

extern "C" void orbx_debug_chol_step_profile(unsigned long long *dev16) { g_cholProf = dev16; }

extern "C" int orbx_lba_solve(orbx_lba *h, const orbx_lba_problem *p, const volatile uint8_t *stop, orbx_lba_result *res)
{
    return lba_run(h, p, stop, res, 5, true, true);
}

extern "C" int orbx_bundle_adjustment(orbx_lba *h, const orbx_lba_problem *p, int iterations, int robust, const volatile uint8_t *stop, orbx_lba_result *res)
{
    if (iterations < 0) { orbx_set_error("negative iteration count"); return ORBX_ERR_ARG; }
    return lba_run(h, p, stop, res, iterations, robust != 0, false);
}

// ---- PoseOptimization: its own small handle (stream + staging), batch of independent frames ----
struct orbx_pose_optimizer {
    int device = 0, maxFrames = 0, maxFeatures = 0;
    hipStream_t stream = nullptr;
    OrbxCallBox box;   // inputs of a call in mapped pinned memory (read in place), results written by the kernel into mapped pinned memory, sequence word
};

extern "C" int orbx_pose_optimizer_create(int device, int max_frames, int max_features, orbx_pose_optimizer **out)
{
    if (!out || max_frames < 1 || max_features < 1) { orbx_set_error("bad pose optimizer arguments"); return ORBX_ERR_ARG; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { orbx_set_error("no HIP device available: liborbx has no CPU fallback"); return ORBX_ERR_NODEVICE; }
    if (device < 0 || device >= ndev) { orbx_set_error("device %d out of range", device); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(device));
    orbx_pose_optimizer *h = new orbx_pose_optimizer();
    h->device = device; h->maxFrames = max_frames; h->maxFeatures = max_features;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; orbx_set_error("hipStreamCreate failed"); return ORBX_ERR_HIP; }
    *out = h;
    return ORBX_OK;
}

extern "C" void orbx_pose_optimizer_destroy(orbx_pose_optimizer *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    h->box.release();
    delete h;
}

// developer tap (tools/pose_opt_phases.py; not part of include/orbx.h): a device array of 32 u64 the PROF instantiation of k_pose_opt adds its phase cycles / counts to
static unsigned long long *g_poProf = nullptr;
extern "C" void orbx_debug_pose_opt_profile(unsigned long long *dev32) { g_poProf = dev32; }

extern "C" int orbx_pose_optimization(orbx_pose_optimizer *h, const orbx_pose_problem *p, float *poses_out, uint8_t *outlier, int32_t *inliers, double *stats)
{
    if (!h || !p || !p->poses || !p->cameras || !p->counts || !p->world_points || !p->observations || !p->inv_sigma2) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    const int B = p->num_frames, cap = p->capacity;
    if (B < 1 || B > h->maxFrames || cap < 1 || cap > h->maxFeatures) { orbx_set_error("problem (%d frames x %d features) exceeds the handle (%d x %d)", B, cap, h->maxFrames, h->maxFeatures); return ORBX_ERR_CAPACITY; }
    ORBX_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const size_t N = (size_t)B * cap;
    // No copy engine, no stream synchronisation (OrbxCallBox): every input is read exactly once, into the registers of the thread that owns the edge, straight
    // from mapped pinned memory; pose / flags / inlier count / statistics are written into mapped pinned memory and the last workgroup raises the sequence word.
    OrbxCallBox &bx = h->box;
    const size_t inBytes = bx.padded((size_t)B * 64) + bx.padded((size_t)B * 20) + bx.padded((size_t)B * 4) + 2 * bx.padded(N * 12) + bx.padded(N * 4);
    const size_t q1 = bx.padded((size_t)B * 64), q2 = q1 + bx.padded(N), q3 = q2 + bx.padded((size_t)B * 4), outBytes = q3 + bx.padded((size_t)B * 64);
    int rcs = bx.begin(inBytes, outBytes, st);
    if (rcs != ORBX_OK) return rcs;
    const float *dPose = bx.put(p->poses, (size_t)B * 16), *dCam = bx.put(p->cameras, (size_t)B * 5);
    const int32_t *dCnt = bx.put(p->counts, (size_t)B);
    const float *dXw = bx.put(p->world_points, N * 3), *dObs = bx.put(p->observations, N * 3), *dInv = bx.put(p->inv_sigma2, N);
    const uint8_t *o0 = bx.outHost<uint8_t>(0), *o1 = bx.outHost<uint8_t>(q1), *o2 = bx.outHost<uint8_t>(q2), *o3 = bx.outHost<uint8_t>(q3);
    PoseOptDev D = {dPose, dCam, dXw, dObs, dInv, dCnt, cap, bx.outDev<float>(0), bx.outDev<uint8_t>(q1), bx.outDev<int32_t>(q2), bx.outDev<double>(q3), bx.counter, bx.flagDev, bx.arm(), g_poProf,
                    B == 1 ? 1 : 0, p->counts[0], {}, {}};
    if (B == 1) { memcpy(D.cam0, p->cameras, sizeof D.cam0); memcpy(D.pose00, p->poses, sizeof D.pose00); }
    const float thMono = (float)sqrt(5.991), thStereo = (float)sqrt(7.815);   // deltaMono / deltaStereo are floats (:389-390)
    Huber hub;
    hub.dMono = thMono; hub.dStereo = thStereo;
    hub.dsqrMono = (float)(hub.dMono * hub.dMono); hub.dsqrStereo = (float)(hub.dStereo * hub.dStereo);   // dsqr is a float member in this fork
    int maxCount = 0;
    for (int i = 0; i < B; i++) maxCount = std::max(maxCount, std::min((int)p->counts[i], cap));
    if (maxCount > 256 * 32) { orbx_set_error("%d correspondences in a frame exceed the pose optimizer's limit %d", maxCount, 256 * 32); return ORBX_ERR_CAPACITY; }
    // edges per thread = ceil(count / 256) up to 4 (a thread's dead slots cost what live ones do in the error pass), then 8 / 16 / 32
    if (g_poProf && maxCount <= 256 * 2) hipLaunchKernelGGL((k_pose_opt<2, true>), dim3((unsigned)B), dim3(256), 0, st, D, hub);
    else if (g_poProf && maxCount <= 256 * 3) hipLaunchKernelGGL((k_pose_opt<3, true>), dim3((unsigned)B), dim3(256), 0, st, D, hub);
    else if (g_poProf && maxCount <= 256 * 4) hipLaunchKernelGGL((k_pose_opt<4, true>), dim3((unsigned)B), dim3(256), 0, st, D, hub);
    else if (maxCount <= 256) hipLaunchKernelGGL((k_pose_opt<1, false>), dim3((unsigned)B), dim3(256), 0, st, D, hub);
    else if (maxCount <= 256 * 2) hipLaunchKernelGGL((k_pose_opt<2, false>), dim3((unsigned)B), dim3(256), 0, st, D, hub);
    else if (maxCount <= 256 * 3) hipLaunchKernelGGL((k_pose_opt<3, false>), dim3((unsigned)B), dim3(256), 0, st, D, hub);
    else if (maxCount <= 256 * 4) hipLaunchKernelGGL((k_pose_opt<4, false>), dim3((unsigned)B), dim3(256), 0, st, D, hub);
    else if (maxCount <= 256 * 8) hipLaunchKernelGGL((k_pose_opt<8, false>), dim3((unsigned)B), dim3(256), 0, st, D, hub);
    else if (maxCount <= 256 * 16) hipLaunchKernelGGL((k_pose_opt<16, false>), dim3((unsigned)B), dim3(256), 0, st, D, hub);
    else hipLaunchKernelGGL((k_pose_opt<32, false>), dim3((unsigned)B), dim3(256), 0, st, D, hub);      // up to 8192 correspondences: 32 per thread, one wave per SIMD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { orbx_set_error("kernel launch failed: %s", hipGetErrorString(e)); return ORBX_ERR_HIP; }
    if ((rcs = bx.wait(st)) != ORBX_OK) return rcs;
    if (poses_out) memcpy(poses_out, o0, (size_t)B * 64);
    if (outlier)      // entries past a frame's count are not written by the kernel: they read 0
        for (int f = 0; f < B; f++) {
            const size_t c = (size_t)std::min(std::max((int)p->counts[f], 0), cap);
            memcpy(outlier + (size_t)f * cap, o1 + (size_t)f * cap, c);
            memset(outlier + (size_t)f * cap + c, 0, (size_t)cap - c);
        }
    if (inliers) memcpy(inliers, o2, (size_t)B * 4);
    if (stats) memcpy(stats, o3, (size_t)B * 64);
    return ORBX_OK;
}

extern "C" int orbx_lba_last_timing(orbx_lba *h, float *device_ms, double *flops)
{
    if (!h) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (!h->timed) { orbx_set_error("no solve yet"); return ORBX_ERR_STATE; }
    ORBX_HIP_CHECK(hipSetDevice(h->device));
    ORBX_HIP_CHECK(hipEventSynchronize(h->ev1));
    if (device_ms) ORBX_HIP_CHECK(hipEventElapsedTime(device_ms, h->ev0, h->ev1));
    if (flops) *flops = h->flops;
    return ORBX_OK;
}
