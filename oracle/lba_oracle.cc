// oracle/lba_oracle.cc -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement (double precision, flat arrays) of Optimizer::LocalBundleAdjustment's
// numerical core: the two-stage g2o Levenberg-Marquardt run of reference
// src/Optimizer.cc:698-958 on a BlockSolver_6_3 with Schur complement.  Restated from
//   Thirdparty/g2o/g2o/types/types_six_dof_expmap.{h,cpp}   edges, Jacobians, cam_project
//   Thirdparty/g2o/g2o/types/se3quat.h, se3_ops.hpp         SE3Quat exp / map / product
//   Thirdparty/g2o/g2o/core/base_binary_edge.hpp:55-119     constructQuadraticForm
//   Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:65-91    Huber (dsqr stored as float)
//   Thirdparty/g2o/g2o/core/block_solver.hpp:354-604        Schur solve, lambda, restore
//   Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-189
//   Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:199-267, 354-435
// g2o needs Eigen, which is neither vendored by the reference nor present here, so the
// reference's own BA cannot be compiled: PARITY UNPINNED (no golden vector exists in the
// reference either).  Pinning available: analytic Jacobians vs central differences and an
// independent scipy solve of the same window (tests/test_lba.py).  Eigen-specific pieces
// restated: Quaterniond(Matrix3d), toRotationMatrix, quaternion*vector, and the sparse LDLT
// (replaced by a dense LDLT of the reduced pose system: same solution up to rounding).
// The contract with the HIP path is |delta| <= 1e-5 on poses, points and residuals, not
// bit equality (accumulation order in g2o follows pointer-ordered maps).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <limits>
#include <vector>

#define LO_API extern "C" __attribute__((visibility("default")))

namespace {

struct Quat { double x, y, z, w; };
struct Pose { Quat q; double t[3]; };

void quat_normalize_pos(Quat &q)   // SE3Quat::normalizeRotation, se3quat.h:280-285
{
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}

Quat quat_from_R(const double R[9])   // Eigen::Quaterniond(Matrix3d)
{
    Quat q;
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (R[7] - R[5]) * t; q.y = (R[2] - R[6]) * t; q.z = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        double v[3];
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (R[3 * k + j] - R[3 * j + k]) * t;
        v[j] = (R[3 * j + i] + R[3 * i + j]) * t;
        v[k] = (R[3 * k + i] + R[3 * i + k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}

void quat_to_R(const Quat &q, double R[9])   // Eigen toRotationMatrix
{
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

void quat_rot(const Quat &q, const double v[3], double o[3])   // Eigen quaternion * vector
{
    const double ux = 2 * (q.y * v[2] - q.z * v[1]), uy = 2 * (q.z * v[0] - q.x * v[2]), uz = 2 * (q.x * v[1] - q.y * v[0]);
    o[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
    o[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
    o[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}

Quat quat_mul(const Quat &a, const Quat &b)
{
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}

void pose_map(const Pose &T, const double X[3], double o[3])   // SE3Quat::map, se3quat.h:217-220
{
    quat_rot(T.q, X, o);
    o[0] += T.t[0]; o[1] += T.t[1]; o[2] += T.t[2];
}

void mat3_mul(const double A[9], const double B[9], double C[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// SE3Quat::exp (se3quat.h:223-255) followed by the left product exp(d)*T
// (VertexSE3Expmap::oplusImpl, types_six_dof_expmap.h:73-76); d = [omega(3), upsilon(3)].
void pose_oplus(Pose &T, const double d[6])
{
    const double *om = d, *up = d + 3;
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9], R[9], V[9];
    mat3_mul(O, O, O2);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) { R[i] = I[i] + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 9; i++) { R[i] = I[i] + a * O[i] + b * O2[i]; V[i] = I[i] + b * O[i] + c * O2[i]; }
    }
    Pose E;
    E.q = quat_from_R(R);
    quat_normalize_pos(E.q);
    for (int i = 0; i < 3; i++) E.t[i] = V[3 * i] * up[0] + V[3 * i + 1] * up[1] + V[3 * i + 2] * up[2];
    // operator*: t = t1 + r1*t2 ; r = r1*r2 ; normalizeRotation (se3quat.h:106-112)
    double rt[3];
    quat_rot(E.q, T.t, rt);
    Pose N;
    for (int i = 0; i < 3; i++) N.t[i] = E.t[i] + rt[i];
    N.q = quat_mul(E.q, T.q);
    quat_normalize_pos(N.q);
    T = N;
}

struct Problem {
    int K, P, E;
    std::vector<Pose> pose;
    std::vector<uint8_t> fixed;
    std::vector<double> intr;      // fx fy cx cy bf
    std::vector<double> pt;        // 3P
    std::vector<int> ep, ek;       // edge -> point, keyframe
    std::vector<double> obs;       // 3E
    std::vector<uint8_t> stereo;
    std::vector<double> info;      // invSigma2
    std::vector<int> level;        // 0 active, 1 excluded
    std::vector<double> err;       // 3E, last computed _error
    bool robust;
    bool onlyPose = false;         // unary EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose edges (PoseOptimization): points are constants
    // Huber deltas: thHuberMono/Stereo are FLOATS (src/Optimizer.cc:781-782); dsqr is a float
    // member of this fork's RobustKernelHuber (robust_kernel_impl.h:84)
    double deltaMono, deltaStereo;
    float dsqrMono, dsqrStereo;
};

// computeError: mono types_six_dof_expmap.h:90-95 / stereo :122-127 with cam_project .cpp:141-157
void edge_error(const Problem &p, int e, double out[3])
{
    const int k = p.ek[e], l = p.ep[e];
    const double *in = &p.intr[5 * (size_t)k];
    double Xc[3];
    pose_map(p.pose[(size_t)k], &p.pt[3 * (size_t)l], Xc);
    if (!p.stereo[e]) {
        const double u = Xc[0] / Xc[2] * in[0] + in[2], v = Xc[1] / Xc[2] * in[1] + in[3];   // project2d then *f + c
        out[0] = p.obs[3 * (size_t)e] - u; out[1] = p.obs[3 * (size_t)e + 1] - v; out[2] = 0;
    } else {
        const float invz = (float)(1.0 / Xc[2]);   // `const float invz = 1.0f/trans_xyz[2]` (.cpp:151, :300): double divide, rounded to float
        const double u = Xc[0] * invz * in[0] + in[2], v = Xc[1] * invz * in[1] + in[3];
        // binary edge: bf arrives as `const float&` (.cpp:150): float*float product; OnlyPose edge: `double bf` member (.h:201, .cpp:304)
        const double bfz = p.onlyPose ? in[4] * (double)invz : (double)((float)in[4] * invz);
        const double ur = u - bfz;
        out[0] = p.obs[3 * (size_t)e] - u; out[1] = p.obs[3 * (size_t)e + 1] - v; out[2] = p.obs[3 * (size_t)e + 2] - ur;
    }
}

double edge_chi2(const Problem &p, int e)   // base_edge.h:58-61 with information = invSigma2 * I
{
    const double *r = &p.err[3 * (size_t)e];
    return (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * p.info[e];
}

void huber(const Problem &p, int e, double chi, double rho[2])   // robust_kernel_impl.cpp:78-91
{
    const double delta = p.stereo[e] ? p.deltaStereo : p.deltaMono;
    const double dsqr = p.stereo[e] ? (double)p.dsqrStereo : (double)p.dsqrMono;
    if (chi <= dsqr) { rho[0] = chi; rho[1] = 1.; }
    else { const double s = sqrt(chi); rho[0] = 2 * s * delta - dsqr; rho[1] = delta / s; }
}

// linearizeOplus: mono .cpp:103-139, stereo .cpp:188-234.  A = d e/d point (D x 3), B = d e/d pose (D x 6)
void edge_jacobians(const Problem &p, int e, double A[9], double B[18])
{
    const int k = p.ek[e], l = p.ep[e];
    const double *in = &p.intr[5 * (size_t)k];
    const double fx = in[0], fy = in[1], bf = in[4];
    double Xc[3], R[9];
    pose_map(p.pose[(size_t)k], &p.pt[3 * (size_t)l], Xc);
    quat_to_R(p.pose[(size_t)k].q, R);
    const double x = Xc[0], y = Xc[1], z = Xc[2], z_2 = z * z;
    memset(A, 0, 9 * sizeof(double));
    memset(B, 0, 18 * sizeof(double));
    if (p.onlyPose) {   // EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose::linearizeOplus (.cpp:266-288, 335-367): reciprocal forms
        const double invz = 1.0 / z, invz_2 = invz * invz;
        B[0] = x * y * invz_2 * fx; B[1] = -(1 + (x * x * invz_2)) * fx; B[2] = y * invz * fx; B[3] = -invz * fx; B[4] = 0; B[5] = x * invz_2 * fx;
        B[6] = (1 + y * y * invz_2) * fy; B[7] = -x * y * invz_2 * fy; B[8] = -x * invz * fy; B[9] = 0; B[10] = -invz * fy; B[11] = y * invz_2 * fy;
        if (p.stereo[e]) {
            B[12] = B[0] - bf * y * invz_2; B[13] = B[1] + bf * x * invz_2; B[14] = B[2]; B[15] = B[3]; B[16] = 0; B[17] = B[5] - bf * invz_2;
        }
        return;
    }
    if (!p.stereo[e]) {
        const double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) A[3 * i + j] = -1. / z * (tmp[3 * i] * R[j] + tmp[3 * i + 1] * R[3 + j] + tmp[3 * i + 2] * R[6 + j]);
    } else {
        for (int j = 0; j < 3; j++) {
            A[j] = -fx * R[j] / z + fx * x * R[6 + j] / z_2;
            A[3 + j] = -fy * R[3 + j] / z + fy * y * R[6 + j] / z_2;
            A[6 + j] = A[j] - bf * R[6 + j] / z_2;
        }
    }
    B[0] = x * y / z_2 * fx; B[1] = -(1 + (x * x / z_2)) * fx; B[2] = y / z * fx; B[3] = -1. / z * fx; B[4] = 0; B[5] = x / z_2 * fx;
    B[6] = (1 + y * y / z_2) * fy; B[7] = -x * y / z_2 * fy; B[8] = -x / z * fy; B[9] = 0; B[10] = -1. / z * fy; B[11] = y / z_2 * fy;
    if (p.stereo[e]) {
        B[12] = B[0] - bf * y / z_2; B[13] = B[1] + bf * x / z_2; B[14] = B[2]; B[15] = B[3]; B[16] = 0; B[17] = B[5] - bf / z_2;
    }
}

bool depth_positive(const Problem &p, int e)   // isDepthPositive, types_six_dof_expmap.h:97-101
{
    double Xc[3];
    pose_map(p.pose[(size_t)p.ek[e]], &p.pt[3 * (size_t)p.ep[e]], Xc);
    return Xc[2] > 0.0;
}

// dense LDL^T solve of the symmetric n x n system (upper triangle of S is authoritative)
bool ldlt_solve(std::vector<double> &S, int n, const double *b, double *x)
{
    std::vector<double> L((size_t)n * n, 0.0), D((size_t)n, 0.0);
    for (int j = 0; j < n; j++) {
        double d = S[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k] * D[k];
        if (!(fabs(d) > 0) || !std::isfinite(d)) return false;
        D[j] = d;
        L[(size_t)j * n + j] = 1;
        for (int i = j + 1; i < n; i++) {
            double v = S[(size_t)j * n + i];   // upper triangle: S(j,i)
            for (int k = 0; k < j; k++) v -= L[(size_t)i * n + k] * L[(size_t)j * n + k] * D[k];
            L[(size_t)i * n + j] = v / d;
        }
    }
    std::vector<double> y((size_t)n);
    for (int i = 0; i < n; i++) { double v = b[i]; for (int k = 0; k < i; k++) v -= L[(size_t)i * n + k] * y[k]; y[i] = v; }
    for (int i = 0; i < n; i++) y[i] /= D[i];
    for (int i = n - 1; i >= 0; i--) { double v = y[i]; for (int k = i + 1; k < n; k++) v -= L[(size_t)k * n + i] * x[k]; x[i] = v; }
    return true;
}

struct Stats { int iters, trials; double chi0, chi1, lambda; };

// Test input "the caller's stop flag is raised while trial k of the call is being decided" (lo_set_stop_after_trials): g2o reads the force-stop
// flag at the end of every Levenberg trial (optimization_algorithm_levenberg.cpp:146, `!_optimizer->terminate()`), before every iteration
// (sparse_optimizer.cpp:370) and LocalBundleAdjustment reads it between its stages (src/Optimizer.cc:869-871).  With k > 0 the run owns a flag
// that becomes 1 right after the k-th trial of the call (counted across both stages) has been accepted or rejected, i.e. before the first of those
// three reads that follows it.
static thread_local int g_stop_after = 0, g_trials_total = 0;
static thread_local volatile uint8_t g_own_stop = 0;

// SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg on the edges of level 0
int optimize(Problem &p, int iterations, const volatile uint8_t *stop, Stats *st)
{
    // ---- initializeOptimization(0): active edges / vertices, index mapping (sparse_optimizer.cpp:199-267,166-190)
    std::vector<int> act;
    for (int e = 0; e < p.E; e++) if (p.level[e] == 0) act.push_back(e);
    std::vector<int> poseIdx((size_t)p.K, -1), ptIdx((size_t)p.P, -1);
    std::vector<char> poseAct((size_t)p.K, 0), ptAct((size_t)p.P, 0);
    for (size_t a = 0; a < act.size(); a++) { poseAct[(size_t)p.ek[act[a]]] = 1; ptAct[(size_t)p.ep[act[a]]] = 1; }
    int nPose = 0, nPt = 0;
    for (int k = 0; k < p.K; k++) if (poseAct[(size_t)k] && !p.fixed[(size_t)k]) poseIdx[(size_t)k] = nPose++;
    for (int l = 0; l < p.P; l++) if (ptAct[(size_t)l] && !p.onlyPose) ptIdx[(size_t)l] = nPt++;
    if (st) { st->iters = 0; st->trials = 0; st->chi0 = st->chi1 = 0; st->lambda = 0; }
    if (act.empty() || nPose + nPt == 0) return 0;
    const int nP6 = 6 * nPose;

    auto compute_errors = [&]() { for (size_t a = 0; a < act.size(); a++) edge_error(p, act[a], &p.err[3 * (size_t)act[a]]); };
    auto robust_chi2 = [&]() {
        double chi = 0;
        for (size_t a = 0; a < act.size(); a++) {
            const double c = edge_chi2(p, act[a]);
            if (p.robust) { double r[2]; huber(p, act[a], c, r); chi += r[0]; } else chi += c;
        }
        return chi;
    };

    std::vector<double> Hpp((size_t)nPose * 36), Hll((size_t)nPt * 9), bp((size_t)nP6), bl((size_t)nPt * 3);
    std::vector<double> Hpl(act.size() * 18);   // per active edge (6x3), only meaningful when the pose is free
    double lambda = 0, ni = 2;
    int nBad = 0, done = 0;
    bool ok = true;
    for (int it = 0; it < iterations && !(stop && *stop) && ok; it++) {
        // ---- OptimizationAlgorithmLevenberg::solve (optimization_algorithm_levenberg.cpp:61-164)
        compute_errors();
        double currentChi = robust_chi2();
        const double iniChi = currentChi;
        if (st && it == 0) st->chi0 = currentChi;
        // buildSystem: linearizeOplus + constructQuadraticForm per active edge (block_solver.hpp:502-560)
        std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(Hll.begin(), Hll.end(), 0.0);
        std::fill(bp.begin(), bp.end(), 0.0); std::fill(bl.begin(), bl.end(), 0.0);
        for (size_t a = 0; a < act.size(); a++) {
            const int e = act[a], D = p.stereo[e] ? 3 : 2;
            double A[9], B[18];
            edge_jacobians(p, e, A, B);
            double w = p.info[e], rw = 1.0;
            if (p.robust) { double r[2]; huber(p, e, edge_chi2(p, e), r); rw = r[1]; }
            const double W = rw * w;                       // weightedOmega = rho' * Omega (second-order term is commented out)
            double omr[3];
            for (int d = 0; d < 3; d++) omr[d] = -w * p.err[3 * (size_t)e + d] * rw;   // omega_r = -Omega*e, then *= rho[1]
            const int li = ptIdx[(size_t)p.ep[e]], pi = poseIdx[(size_t)p.ek[e]];
            for (int i = 0; i < 3 && li >= 0; i++) {
                for (int j = 0; j < 3; j++) { double s = 0; for (int d = 0; d < D; d++) s += A[3 * d + i] * W * A[3 * d + j]; Hll[(size_t)li * 9 + 3 * i + j] += s; }
                double s = 0; for (int d = 0; d < D; d++) s += A[3 * d + i] * omr[d]; bl[(size_t)li * 3 + i] += s;
            }
            if (pi >= 0) {
                for (int i = 0; i < 6; i++) {
                    for (int j = 0; j < 6; j++) { double s = 0; for (int d = 0; d < D; d++) s += B[6 * d + i] * W * B[6 * d + j]; Hpp[(size_t)pi * 36 + 6 * i + j] += s; }
                    double s = 0; for (int d = 0; d < D; d++) s += B[6 * d + i] * omr[d]; bp[(size_t)pi * 6 + i] += s;
                    for (int j = 0; j < 3; j++) { double s2 = 0; for (int d = 0; d < D; d++) s2 += B[6 * d + i] * W * A[3 * d + j]; Hpl[a * 18 + 3 * i + j] = s2; }
                }
            }
        }
        if (it == 0) {   // computeLambdaInit (:166-180): tau * max |diag H| over every free vertex
            double mx = 0;
            for (int i = 0; i < nPose; i++) for (int j = 0; j < 6; j++) mx = std::max(mx, fabs(Hpp[(size_t)i * 36 + 7 * j]));
            for (int i = 0; i < nPt; i++) for (int j = 0; j < 3; j++) mx = std::max(mx, fabs(Hll[(size_t)i * 9 + 4 * j]));
            lambda = 1e-5 * mx; ni = 2; nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        std::vector<double> xp((size_t)nP6), xl((size_t)nPt * 3);
        do {
            const std::vector<Pose> savePose = p.pose;      // push()
            const std::vector<double> savePt = p.pt;
            // ---- setLambda + Schur solve (block_solver.hpp:564-589, 354-486)
            std::vector<double> S((size_t)nP6 * nP6, 0.0), bs(bp), Dinv((size_t)nPt * 9), db((size_t)nPt * 3);
            for (int i = 0; i < nPose; i++)
                for (int r = 0; r < 6; r++)
                    for (int c = 0; c < 6; c++) S[(size_t)(6 * i + r) * nP6 + 6 * i + c] = Hpp[(size_t)i * 36 + 6 * r + c] + (r == c ? lambda : 0.0);
            for (int l = 0; l < nPt; l++) {
                double M[9];
                for (int i = 0; i < 9; i++) M[i] = Hll[(size_t)l * 9 + i];
                M[0] += lambda; M[4] += lambda; M[8] += lambda;
                const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
                const double det = M[0] * c00 + M[1] * c01 + M[2] * c02, id = 1.0 / det;
                double *I = &Dinv[(size_t)l * 9];
                I[0] = c00 * id; I[1] = (M[2] * M[7] - M[1] * M[8]) * id; I[2] = (M[1] * M[5] - M[2] * M[4]) * id;
                I[3] = c01 * id; I[4] = (M[0] * M[8] - M[2] * M[6]) * id; I[5] = (M[2] * M[3] - M[0] * M[5]) * id;
                I[6] = c02 * id; I[7] = (M[1] * M[6] - M[0] * M[7]) * id; I[8] = (M[0] * M[4] - M[1] * M[3]) * id;
                for (int i = 0; i < 3; i++) db[(size_t)l * 3 + i] = I[3 * i] * bl[(size_t)l * 3] + I[3 * i + 1] * bl[(size_t)l * 3 + 1] + I[3 * i + 2] * bl[(size_t)l * 3 + 2];
            }
            // per landmark: its free-pose edges
            std::vector<std::vector<int> > byPt((size_t)nPt);
            for (size_t a = 0; a < act.size(); a++)
                if (poseIdx[(size_t)p.ek[act[a]]] >= 0 && ptIdx[(size_t)p.ep[act[a]]] >= 0) byPt[(size_t)ptIdx[(size_t)p.ep[act[a]]]].push_back((int)a);
            for (int l = 0; l < nPt; l++) {
                const double *I = &Dinv[(size_t)l * 9];
                for (size_t u = 0; u < byPt[(size_t)l].size(); u++) {
                    const int a1 = byPt[(size_t)l][u], i1 = poseIdx[(size_t)p.ek[act[(size_t)a1]]];
                    const double *B1 = &Hpl[(size_t)a1 * 18];
                    double BD[18];
                    for (int r = 0; r < 6; r++)
                        for (int c = 0; c < 3; c++) BD[3 * r + c] = B1[3 * r] * I[c] + B1[3 * r + 1] * I[3 + c] + B1[3 * r + 2] * I[6 + c];
                    for (int r = 0; r < 6; r++) bs[(size_t)6 * i1 + r] -= B1[3 * r] * db[(size_t)l * 3] + B1[3 * r + 1] * db[(size_t)l * 3 + 1] + B1[3 * r + 2] * db[(size_t)l * 3 + 2];
                    for (size_t v = 0; v < byPt[(size_t)l].size(); v++) {
                        const int a2 = byPt[(size_t)l][v], i2 = poseIdx[(size_t)p.ek[act[(size_t)a2]]];
                        if (i2 < i1) continue;   // upper triangular block pairs only (:419-430)
                        const double *B2 = &Hpl[(size_t)a2 * 18];
                        for (int r = 0; r < 6; r++)
                            for (int c = 0; c < 6; c++)
                                S[(size_t)(6 * i1 + r) * nP6 + 6 * i2 + c] -= BD[3 * r] * B2[3 * c] + BD[3 * r + 1] * B2[3 * c + 1] + BD[3 * r + 2] * B2[3 * c + 2];
                    }
                }
            }
            bool ok2 = true;
            if (nP6 > 0) ok2 = ldlt_solve(S, nP6, bs.data(), xp.data());
            if (ok2) {   // landmark back-substitution (:459-481)
                std::vector<double> cl(bl);
                for (size_t a = 0; a < act.size(); a++) {
                    const int i1 = poseIdx[(size_t)p.ek[act[a]]];
                    if (i1 < 0) continue;
                    const int l = ptIdx[(size_t)p.ep[act[a]]];
                    if (l < 0) continue;
                    const double *B1 = &Hpl[a * 18];
                    for (int c = 0; c < 3; c++) { double s = 0; for (int r = 0; r < 6; r++) s += B1[3 * r + c] * xp[(size_t)6 * i1 + r]; cl[(size_t)l * 3 + c] -= s; }
                }
                for (int l = 0; l < nPt; l++)
                    for (int i = 0; i < 3; i++) xl[(size_t)l * 3 + i] = Dinv[(size_t)l * 9 + 3 * i] * cl[(size_t)l * 3] + Dinv[(size_t)l * 9 + 3 * i + 1] * cl[(size_t)l * 3 + 1] + Dinv[(size_t)l * 9 + 3 * i + 2] * cl[(size_t)l * 3 + 2];
            }
            // update(x): oplus on every free vertex.  NOTE: g2o applies the (stale) x even when the
            // linear solve failed; the state is restored by pop() below in that case.
            for (int k = 0; k < p.K; k++) if (poseIdx[(size_t)k] >= 0) pose_oplus(p.pose[(size_t)k], &xp[(size_t)6 * poseIdx[(size_t)k]]);
            for (int l = 0; l < p.P; l++) if (ptIdx[(size_t)l] >= 0) for (int i = 0; i < 3; i++) p.pt[3 * (size_t)l + i] += xl[(size_t)3 * ptIdx[(size_t)l] + i];
            compute_errors();
            double tempChi = robust_chi2();
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            double scale = 0;   // computeScale (:182-189): sum x_j (lambda x_j + b_j)
            for (int j = 0; j < nP6; j++) scale += xp[(size_t)j] * (lambda * xp[(size_t)j] + bp[(size_t)j]);
            for (int j = 0; j < 3 * nPt; j++) scale += xl[(size_t)j] * (lambda * xl[(size_t)j] + bl[(size_t)j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                p.pose = savePose; p.pt = savePt;   // pop(): estimates restored, _error stays as last computed
            }
            qmax++;
            if (st) st->trials++;
            if (g_stop_after > 0 && ++g_trials_total >= g_stop_after) g_own_stop = 1;
        } while (rho < 0 && qmax < 10 && !(stop && *stop));
        done++;
        if (st) { st->iters = done; st->chi1 = currentChi; st->lambda = lambda; }
        if (qmax == 10 || rho == 0) { ok = false; continue; }   // Terminate
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) ok = false;
    }
    return done;
}

}  // namespace

// poses: K x 16 float (Tcw row-major 4x4, as KeyFrame::GetPose / Converter::toSE3Quat src/Converter.cc:57-70)
// intr:  K x 5 float {fx,fy,cx,cy,bf};  points: P x 3 float;  edge_obs: E x 3 float {u,v,uR (<0: monocular)}
// Outputs: poses_out K x 16 float, points_out P x 3 float, edge_chi2 E double (e->chi2() as LocalBundleAdjustment
// reads it at :921-958), edge_outlier E (chi2 > 5.991/7.815 or depth <= 0), stats[8] doubles.
// iters1 LM iterations (Huber kernels when robust1); with secondStage the LocalBundleAdjustment continuation: outlier
// classification, then 10 iterations without kernels on the inliers.
// optional FP64 taps of the final state (set by lo_ba_f64 around a run): poses K x 12 (R row-major, t), points P x 3
static thread_local double *g_poses_d = nullptr, *g_points_d = nullptr;

/* k > 0: the following runs on this thread see their stop flag raised right after their k-th Levenberg trial; 0: back to the caller's flag */
LO_API void lo_set_stop_after_trials(int k) { g_stop_after = k > 0 ? k : 0; }

static int run_ba(int K, const float *poses, const uint8_t *fixed, const float *intr, int P, const float *points, int E, const int32_t *edge_point,
                  const int32_t *edge_kf, const float *edge_obs, const float *edge_inv_sigma2, const volatile uint8_t *stop, int iters1, bool robust1,
                  bool secondStage, float *poses_out, float *points_out, double *edge_chi2_out, uint8_t *edge_outlier, double *stats)
{
    Problem p;
    p.K = K; p.P = P; p.E = E;
    p.pose.resize((size_t)K); p.fixed.assign(fixed, fixed + K); p.intr.resize((size_t)5 * K); p.pt.resize((size_t)3 * P);
    for (int k = 0; k < K; k++) {
        double R[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = poses[16 * (size_t)k + 4 * i + j];
        p.pose[(size_t)k].q = quat_from_R(R);
        quat_normalize_pos(p.pose[(size_t)k].q);
        for (int i = 0; i < 3; i++) p.pose[(size_t)k].t[i] = poses[16 * (size_t)k + 4 * i + 3];
        for (int i = 0; i < 5; i++) p.intr[5 * (size_t)k + i] = intr[5 * (size_t)k + i];
    }
    for (int i = 0; i < 3 * P; i++) p.pt[(size_t)i] = points[i];
    p.ep.assign(edge_point, edge_point + E); p.ek.assign(edge_kf, edge_kf + E);
    p.obs.resize((size_t)3 * E); p.stereo.resize((size_t)E); p.info.resize((size_t)E); p.level.assign((size_t)E, 0); p.err.assign((size_t)3 * E, 0.0);
    for (int e = 0; e < E; e++) {
        for (int i = 0; i < 3; i++) p.obs[3 * (size_t)e + i] = edge_obs[3 * (size_t)e + i];
        p.stereo[(size_t)e] = !(edge_obs[3 * (size_t)e + 2] < 0);   // mvuRight < 0 -> monocular (src/Optimizer.cc:797)
        p.info[(size_t)e] = edge_inv_sigma2[e];
    }
    // LocalBundleAdjustment: sqrt(5.991) (src/Optimizer.cc:764-765); BundleAdjustment: sqrt(5.99) (:141-142)
    const float thMono = (float)sqrt(secondStage ? 5.991 : 5.99), thStereo = (float)sqrt(7.815);
    p.deltaMono = thMono; p.deltaStereo = thStereo;
    p.dsqrMono = (float)(p.deltaMono * p.deltaMono); p.dsqrStereo = (float)(p.deltaStereo * p.deltaStereo);
    Stats s1 = {0, 0, 0, 0, 0}, s2 = {0, 0, 0, 0, 0};
    if (g_stop_after > 0) { g_own_stop = 0; g_trials_total = 0; stop = &g_own_stop; }      // (lo_set_stop_after_trials: the run's own flag)
    if (!(stop && *stop)) {
        p.robust = robust1;
        optimize(p, iters1, stop, &s1);                              // :863-864 (LBA), :247 (BundleAdjustment)
        if (secondStage && !(stop && *stop)) {
            for (int e = 0; e < E; e++) {                            // :880-912
                const double th = p.stereo[(size_t)e] ? 7.815 : 5.991;
                if (edge_chi2(p, e) > th || !depth_positive(p, e)) p.level[(size_t)e] = 1;
            }
            p.robust = false;
            optimize(p, 10, stop, &s2);                              // :916-917
        }
    }
    for (int e = 0; e < E; e++) {                                    // :921-958
        const double th = p.stereo[(size_t)e] ? 7.815 : 5.991, c = edge_chi2(p, e);
        edge_chi2_out[e] = c;
        edge_outlier[e] = (c > th || !depth_positive(p, e)) ? 1 : 0;
    }
    for (int k = 0; k < K; k++) {                                    // Converter::toCvMat(SE3Quat), :981-989
        double R[9];
        quat_to_R(p.pose[(size_t)k].q, R);
        float *o = poses_out + 16 * (size_t)k;
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) o[4 * i + j] = (float)R[3 * i + j]; o[4 * i + 3] = (float)p.pose[(size_t)k].t[i]; }
        o[12] = o[13] = o[14] = 0.f; o[15] = 1.f;
    }
    for (int i = 0; i < 3 * P; i++) points_out[i] = (float)p.pt[(size_t)i];
    if (g_poses_d)
        for (int k = 0; k < K; k++) {
            double R[9];
            quat_to_R(p.pose[(size_t)k].q, R);
            for (int i = 0; i < 9; i++) g_poses_d[12 * (size_t)k + i] = R[i];
            for (int i = 0; i < 3; i++) g_poses_d[12 * (size_t)k + 9 + i] = p.pose[(size_t)k].t[i];
        }
    if (g_points_d) for (int i = 0; i < 3 * P; i++) g_points_d[i] = p.pt[(size_t)i];
    if (stats) {
        stats[0] = s1.iters; stats[1] = s1.trials; stats[2] = s1.chi0; stats[3] = s1.chi1;
        stats[4] = s2.iters; stats[5] = s2.trials; stats[6] = s2.chi0; stats[7] = s2.chi1;
    }
    return 0;
}

LO_API int lo_local_bundle_adjustment(int K, const float *poses, const uint8_t *fixed, const float *intr, int P, const float *points, int E,
                                      const int32_t *edge_point, const int32_t *edge_kf, const float *edge_obs, const float *edge_inv_sigma2,
                                      const volatile uint8_t *stop, float *poses_out, float *points_out, double *edge_chi2_out,
                                      uint8_t *edge_outlier, double *stats)
{
    return run_ba(K, poses, fixed, intr, P, points, E, edge_point, edge_kf, edge_obs, edge_inv_sigma2, stop, 5, true, true, poses_out, points_out, edge_chi2_out,
                  edge_outlier, stats);
}

// Optimizer::BundleAdjustment / GlobalBundleAdjustemnt (src/Optimizer.cc:55-84, 86-360): the same graph, ONE optimize(nIterations)
// with Huber kernels iff bRobust (:175-180, 201-206), no outlier pass.  Same array conventions as above.
LO_API int lo_bundle_adjustment(int K, const float *poses, const uint8_t *fixed, const float *intr, int P, const float *points, int E,
                                const int32_t *edge_point, const int32_t *edge_kf, const float *edge_obs, const float *edge_inv_sigma2,
                                const volatile uint8_t *stop, int iterations, int robust, float *poses_out, float *points_out, double *edge_chi2_out,
                                uint8_t *edge_outlier, double *stats)
{
    return run_ba(K, poses, fixed, intr, P, points, E, edge_point, edge_kf, edge_obs, edge_inv_sigma2, stop, iterations, robust != 0, false, poses_out, points_out,
                  edge_chi2_out, edge_outlier, stats);
}

// The same two functions with the FP64 state exposed (poses_d K x 12: R row-major then t; points_d P x 3), for the comparison
// with the reference's g2o below float32 resolution (oracle/refslam_wrap.cc: orbslam_g2o_ba).  secondStage selects
// LocalBundleAdjustment (iters1 = 5, robust) or BundleAdjustment.
LO_API int lo_ba_f64(int K, const float *poses, const uint8_t *fixed, const float *intr, int P, const float *points, int E, const int32_t *edge_point,
                     const int32_t *edge_kf, const float *edge_obs, const float *edge_inv_sigma2, int iters1, int robust1, int secondStage, double *poses_d,
                     double *points_d, double *edge_chi2_out, uint8_t *edge_outlier, double *stats)
{
    std::vector<float> po((size_t)16 * (K > 0 ? K : 1)), xo((size_t)3 * (P > 0 ? P : 1));
    g_poses_d = poses_d; g_points_d = points_d;
    const int r = run_ba(K, poses, fixed, intr, P, points, E, edge_point, edge_kf, edge_obs, edge_inv_sigma2, nullptr, iters1, robust1 != 0, secondStage != 0,
                         po.data(), xo.data(), edge_chi2_out, edge_outlier, stats);
    g_poses_d = g_points_d = nullptr;
    return r;
}

// test hooks: error and analytic Jacobians of one edge at a given state (for the central-difference check)
LO_API void lo_edge_eval(const float *pose16, const float *intr5, const double *X, const float *obs3, double *err3, double *A9, double *B18)
{
    Problem p;
    p.K = 1; p.P = 1; p.E = 1;
    p.pose.resize(1);
    double R[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = pose16[4 * i + j];
    p.pose[0].q = quat_from_R(R);
    quat_normalize_pos(p.pose[0].q);
    for (int i = 0; i < 3; i++) p.pose[0].t[i] = pose16[4 * i + 3];
    p.intr.assign(intr5, intr5 + 5);
    p.pt.assign(X, X + 3);
    p.ep.assign(1, 0); p.ek.assign(1, 0);
    p.obs.assign(obs3, obs3 + 3);
    p.stereo.assign(1, !(obs3[2] < 0));
    p.info.assign(1, 1.0);
    edge_error(p, 0, err3);
    edge_jacobians(p, 0, A9, B18);
}

// error after applying the 6-dof update d to the pose and the 3-vector dx to the point
LO_API void lo_edge_error_perturbed(const float *pose16, const float *intr5, const double *X, const float *obs3, const double *d6, const double *dx3, double *err3)
{
    Problem p;
    p.K = 1; p.P = 1; p.E = 1;
    p.pose.resize(1);
    double R[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = pose16[4 * i + j];
    p.pose[0].q = quat_from_R(R);
    quat_normalize_pos(p.pose[0].q);
    for (int i = 0; i < 3; i++) p.pose[0].t[i] = pose16[4 * i + 3];
    pose_oplus(p.pose[0], d6);
    p.intr.assign(intr5, intr5 + 5);
    p.pt.resize(3);
    for (int i = 0; i < 3; i++) p.pt[(size_t)i] = X[i] + dx3[i];
    p.ep.assign(1, 0); p.ek.assign(1, 0);
    p.obs.assign(obs3, obs3 + 3);
    p.stereo.assign(1, !(obs3[2] < 0));
    p.info.assign(1, 1.0);
    edge_error(p, 0, err3);
}


// Optimizer::PoseOptimization(Frame *pFrame) (src/Optimizer.cc:363-605): motion-only BA of one frame.
// One VertexSE3Expmap, unary EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose per feature that
// has a MapPoint (obs[3i+2] < 0: monocular), Huber kernels (deltas sqrt(5.991) / sqrt(7.815) as floats),
// BlockSolver_6_3 + LinearSolverDense + Levenberg.  Four rounds of optimize(10), EVERY round restarting from
// the frame's initial pose (:520) on the edges currently classified as inliers; after each round the
// edges are re-classified with chi2 > 5.991 / 7.815 read as FLOAT (:533, :563; inliers keep the error of
// the last LM evaluation, outliers are re-evaluated), the robust kernels are dropped after the third.
// Returns nInitialCorrespondences - nBad; pose_out = pFrame->mTcw after SetPose(:598-601).
// PARITY UNPINNED (g2o needs Eigen); cross-checked against scipy in tests/test_pose_optimization.py.
LO_API int lo_pose_optimization(const float *pose16, const float *cam5, int n, const float *Xw, const float *obs, const float *inv_sigma2,
                                float *pose_out16, uint8_t *outlier, double *stats)
{
    Problem p;
    p.K = 1; p.P = n; p.E = n; p.onlyPose = true;
    p.pose.resize(1); p.fixed.assign(1, 0); p.intr.resize(5); p.pt.resize((size_t)3 * n);
    auto load_pose = [&]() {
        double R[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = pose16[4 * i + j];
        p.pose[0].q = quat_from_R(R);
        quat_normalize_pos(p.pose[0].q);
        for (int i = 0; i < 3; i++) p.pose[0].t[i] = pose16[4 * i + 3];
    };
    load_pose();
    for (int i = 0; i < 5; i++) p.intr[(size_t)i] = cam5[i];
    for (int i = 0; i < 3 * n; i++) p.pt[(size_t)i] = Xw[i];
    p.ep.resize((size_t)n); p.ek.assign((size_t)n, 0);
    p.obs.resize((size_t)3 * n); p.stereo.resize((size_t)n); p.info.resize((size_t)n); p.level.assign((size_t)n, 0); p.err.assign((size_t)3 * n, 0.0);
    for (int e = 0; e < n; e++) {
        p.ep[(size_t)e] = e;
        for (int i = 0; i < 3; i++) p.obs[3 * (size_t)e + i] = obs[3 * (size_t)e + i];
        p.stereo[(size_t)e] = !(obs[3 * (size_t)e + 2] < 0);      // mvuRight < 0 -> monocular (:404)
        p.info[(size_t)e] = inv_sigma2[e];
        outlier[e] = 0;
    }
    const float thMono = (float)sqrt(5.991), thStereo = (float)sqrt(7.815);
    p.deltaMono = thMono; p.deltaStereo = thStereo;
    p.dsqrMono = (float)(p.deltaMono * p.deltaMono); p.dsqrStereo = (float)(p.deltaStereo * p.deltaStereo);
    if (stats) for (int i = 0; i < 8; i++) stats[i] = 0;
    if (n < 3) { memcpy(pose_out16, pose16, 64); return 0; }      // :509-510
    const float chi2Mono = 5.991f, chi2Stereo = 7.815f;
    int nBad = 0;
    p.robust = true;
    for (int it = 0; it < 4; it++) {
        load_pose();                                                // :520
        Stats st = {0, 0, 0, 0, 0};
        optimize(p, 10, nullptr, &st);                              // :521-522
        if (stats) { stats[2 * it] = st.iters; stats[2 * it + 1] = st.chi1; }
        nBad = 0;
        for (int e = 0; e < n; e++) {
            if (outlier[e]) edge_error(p, e, &p.err[3 * (size_t)e]);   // :529-532
            const float chi2 = (float)edge_chi2(p, e);
            if (chi2 > (p.stereo[(size_t)e] ? chi2Stereo : chi2Mono)) { outlier[e] = 1; p.level[(size_t)e] = 1; nBad++; }
            else { outlier[e] = 0; p.level[(size_t)e] = 0; }
        }
        if (it == 2) p.robust = false;                               // e->setRobustKernel(0), :547-548
        if (n < 10) break;                                           // optimizer.edges().size()<10, :589-590
    }
    double R[9];
    quat_to_R(p.pose[0].q, R);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) pose_out16[4 * i + j] = (float)R[3 * i + j]; pose_out16[4 * i + 3] = (float)p.pose[0].t[i]; }
    pose_out16[12] = pose_out16[13] = pose_out16[14] = 0.f; pose_out16[15] = 1.f;
    return n - nBad;
}
