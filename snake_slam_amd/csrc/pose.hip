// Robust pose-only optimisation ("snk-pose v1") for the step that follows every projection matcher:
// PoseRefinement::refinePose (Snake/Tracking/PoseRefinement.h:27-87) and
// PoseRefinement::RefinePoseWithMatches (Snake/Tracking/PoseRefinement.cpp:25-79).  The optimiser
// the reference calls (Saiga::RobustPoseOptimization::optimizePoseRobust, absent submodule) is
// [DEFINED] in DESIGN.md §3c and restated on the CPU in oracle/pose_oracle.c.
//
// One workgroup of one or four WAVEFRONTS per frame (problem): a thread owns matches t, t + threads, ...; each damped
// Gauss-Newton iteration accumulates the 21 + 6 entries of J^T W J / J^T W r per thread, sums them in a fixed order
// (deterministic), and one wavefront solves the 6x6 system.  Problems of a batch run side by side.
#include <cmath>

#include "matcher_handle.hpp"

// Floating-point contraction of this file (round 6): pose refinement is specified by a tolerance and its kernel is bound by fp64 issue
// slots, so a * b + c may be ONE v_fma_f64 -- by this pragma, not by -ffp-contract=fast on the command line (the command-line form lets
// the back end fuse whatever it finds and cannot be switched off per kernel; the pragma puts the permission on the operations, and
// backproject_kernel below takes it back: its world points are the next frame's local map and stay bit-exact).
#pragma clang fp contract(fast)

namespace snk
{
namespace
{
typedef unsigned char u8;
typedef unsigned int u32;

struct PoseMeta
{
    int off, n;
    int use_prior, pad;
    double pose[7];
    double pred[7];
    double w_rot, w_trans;
};

struct CamD
{
    double fx, fy, cx, cy, bf;
};

__device__ __forceinline__ void quat_to_R(const double* q, double* R)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

// sin and cos of a half angle |x| <= pi / 4 by their Taylor series (terms to x^17 / x^16: truncation below 1e-19 relative);
// larger arguments -- a Gauss-Newton step that turns the camera by more than 90 degrees -- take the library's sincos.  The
// library call alone was ~150 of the ~860 instructions the solver wavefront runs per step while the others wait.
__device__ __forceinline__ void sincos_half(double x, double* s, double* c)
{
    if (fabs(x) > 0.78539816339744831)
    {
        sincos(x, s, c);
        return;
    }
    const double z = x * x;
    double ps = -1.0 / 355687428096000.0;  // -1/17!
    ps        = fma(ps, z, 1.0 / 1307674368000.0);
    ps        = fma(ps, z, -1.0 / 6227020800.0);
    ps        = fma(ps, z, 1.0 / 39916800.0);
    ps        = fma(ps, z, -1.0 / 362880.0);
    ps        = fma(ps, z, 1.0 / 5040.0);
    ps        = fma(ps, z, -1.0 / 120.0);
    ps        = fma(ps, z, 1.0 / 6.0);
    *s        = fma(-x * z, ps, x);
    double pc = 1.0 / 20922789888000.0;  // 1/16!
    pc        = fma(pc, z, -1.0 / 87178291200.0);
    pc        = fma(pc, z, 1.0 / 479001600.0);
    pc        = fma(pc, z, -1.0 / 3628800.0);
    pc        = fma(pc, z, 1.0 / 40320.0);
    pc        = fma(pc, z, -1.0 / 720.0);
    pc        = fma(pc, z, 1.0 / 24.0);
    pc        = fma(pc, z, -0.5);
    *c        = fma(z, pc, 1.0);
}

// pose <- exp(delta) * pose, delta = (translation, rotation)
__device__ void se3_update(double* pose, const double* d)
{
    const double wx = d[3], wy = d[4], wz = d[5];
    const double th2 = wx * wx + wy * wy + wz * wz;
    double B, Cc, qd[4];
    if (th2 < 1e-16)
    {
        B  = 0.5 - th2 / 24.0;
        Cc = 1.0 / 6.0 - th2 / 120.0;
        const double h = 0.5 - th2 / 48.0;
        qd[0] = h * wx; qd[1] = h * wy; qd[2] = h * wz; qd[3] = 1.0 - th2 / 8.0;
    }
    else
    {
        // one sincos of the half angle (the four libm calls of the textbook form were a fifth of a Gauss-Newton step); 1 / th from
        // the reciprocal square root instead of a square root and three divisions
        const double ith = rsqrt_nr(th2), th = th2 * ith;
        double s2, c2;
        sincos_half(0.5 * th, &s2, &c2);
        const double s = 2.0 * s2 * c2, c = 1.0 - 2.0 * s2 * s2;
        const double ith2 = ith * ith;
        B  = (1.0 - c) * ith2;
        Cc = (th - s) * (ith2 * ith);
        const double sh = s2 * ith;
        qd[0] = sh * wx; qd[1] = sh * wy; qd[2] = sh * wz; qd[3] = c2;
    }
    const double vx = d[0], vy = d[1], vz = d[2];
    const double cx = wy * vz - wz * vy, cy = wz * vx - wx * vz, cz = wx * vy - wy * vx;
    const double ccx = wy * cz - wz * cy, ccy = wz * cx - wx * cz, ccz = wx * cy - wy * cx;
    const double tdx = vx + B * cx + Cc * ccx, tdy = vy + B * cy + Cc * ccy, tdz = vz + B * cz + Cc * ccz;
    double Rd[9];
    quat_to_R(qd, Rd);
    const double tx = pose[4], ty = pose[5], tz = pose[6];
    const double ax = qd[0], ay = qd[1], az = qd[2], aw = qd[3], bx = pose[0], by = pose[1], bz = pose[2], bw = pose[3];
    double q[4];
    q[0] = aw * bx + ax * bw + ay * bz - az * by;
    q[1] = aw * by - ax * bz + ay * bw + az * bx;
    q[2] = aw * bz + ax * by - ay * bx + az * bw;
    q[3] = aw * bw - ax * bx - ay * by - az * bz;
    const double rn = rsqrt_nr(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    pose[0] = q[0] * rn; pose[1] = q[1] * rn; pose[2] = q[2] * rn; pose[3] = q[3] * rn;
    pose[4] = Rd[0] * tx + Rd[1] * ty + Rd[2] * tz + tdx;
    pose[5] = Rd[3] * tx + Rd[4] * ty + Rd[5] * tz + tdy;
    pose[6] = Rd[6] * tx + Rd[7] * ty + Rd[8] * tz + tdz;
}

// e = log(T * T_pred^-1) = (rho, omega)
__device__ void se3_log_rel(const double* pose, const double* pred, double* e)
{
    const double ax = pose[0], ay = pose[1], az = pose[2], aw = pose[3];
    const double bx = -pred[0], by = -pred[1], bz = -pred[2], bw = pred[3];
    double q[4];
    q[0] = aw * bx + ax * bw + ay * bz - az * by;
    q[1] = aw * by - ax * bz + ay * bw + az * bx;
    q[2] = aw * bz + ax * by - ay * bx + az * bw;
    q[3] = aw * bw - ax * bx - ay * by - az * bz;
    if (q[3] < 0.0)
        for (int i = 0; i < 4; ++i) q[i] = -q[i];
    double Re[9];
    quat_to_R(q, Re);
    const double tx = pose[4] - (Re[0] * pred[4] + Re[1] * pred[5] + Re[2] * pred[6]);
    const double ty = pose[5] - (Re[3] * pred[4] + Re[4] * pred[5] + Re[5] * pred[6]);
    const double tz = pose[6] - (Re[6] * pred[4] + Re[7] * pred[5] + Re[8] * pred[6]);
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], n = sqrt(n2);
    double wx, wy, wz, cc;
    if (n < 1e-10)
    {
        const double k = 2.0 / q[3];
        wx = k * q[0]; wy = k * q[1]; wz = k * q[2];
        cc = 1.0 / 12.0;
    }
    else
    {
        const double th = 2.0 * atan2(n, q[3]);
        const double k  = th / n;
        wx = k * q[0]; wy = k * q[1]; wz = k * q[2];
        if (th < 1e-4)
            cc = 1.0 / 12.0 + th * th / 720.0;
        else
            cc = (1.0 - (th * sin(th)) / (2.0 * (1.0 - cos(th)))) / (th * th);
    }
    const double c1x = wy * tz - wz * ty, c1y = wz * tx - wx * tz, c1z = wx * ty - wy * tx;
    const double c2x = wy * c1z - wz * c1y, c2y = wz * c1x - wx * c1z, c2z = wx * c1y - wy * c1x;
    e[0] = tx - 0.5 * c1x + cc * c2x;
    e[1] = ty - 0.5 * c1y + cc * c2y;
    e[2] = tz - 0.5 * c1z + cc * c2z;
    e[3] = wx; e[4] = wy; e[5] = wz;
}

// One match as the kernel walks it: world point, observation, od = x - bf / depth (the right-image column the stereo residual
// compares with; NaN marks a monocular observation -- computed once per solve, not in each of the 40 steps) and the weight.
struct Match
{
    double px, py, pz, ox, oy, od, w;
};
__device__ __forceinline__ Match make_match(const double* p, const snk_pose_obs& o, const CamD& cam)
{
    return Match{p[0], p[1], p[2], o.x, o.y, o.depth > 0.0 ? o.x - cam.bf / o.depth : __builtin_nan(""), o.weight};
}

// The weighted residual (r[2] = 0 for a monocular observation) of a match under (R, t); false when the point is not in front of
// the camera.  X, Y, Z = the point in the camera frame, iz = 1 / Z, pu = projected column.
struct Proj
{
    double X, Y, Z, iz, xz, yz, r0, r1, r2;
    bool stereo;
};
__device__ __forceinline__ bool project(const double* R, const double* t, const Match& m, const CamD& cam, Proj& q)
{
    q.X = R[0] * m.px + R[1] * m.py + R[2] * m.pz + t[0];
    q.Y = R[3] * m.px + R[4] * m.py + R[5] * m.pz + t[1];
    q.Z = R[6] * m.px + R[7] * m.py + R[8] * m.pz + t[2];
    if (q.Z <= 0.0) return false;
    q.iz     = rcp_nr(q.Z);
    q.xz     = q.X * q.iz;
    q.yz     = q.Y * q.iz;
    q.stereo = m.od == m.od;
    const double pu = cam.fx * q.xz + cam.cx;
    q.r0     = m.w * (pu - m.ox);
    q.r1     = m.w * (cam.fy * q.yz + cam.cy - m.oy);
    const double r2 = m.w * ((pu - cam.bf * q.iz) - m.od);
    q.r2     = q.stereo ? r2 : 0.0;
    return true;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ double clampd(double v)
{
    return v < 1e-6 ? 1e-6 : (v > 1e32 ? 1e32 : v);
}

__device__ int chol_solve6(const double* A, const double* b, double* x)
{
    double L[36], inv[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) L[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
        {
            double s = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j)
            {
                if (!(s > 0.0)) return -1;
                inv[i]       = rsqrt_nr(s);  // 1 / L_ii directly (8 instructions; sqrt + division are 34), L_ii = s / L_ii
                L[i * 6 + i] = s * inv[i];
            }
            else
                L[i * 6 + j] = s * inv[j];
        }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
    {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k];
        y[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i)
    {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
        x[i] = s * inv[i];
    }
    return 0;
}

// WAVES wavefronts per frame: a match per thread and stride 64 * WAVES.  The 27 sums of a step are reduced inside each row of 16
// lanes with four DPP steps (no LDS, no read-back), the 4 * WAVES row sums of every quantity meet in LDS, 27 threads add them in
// a fixed order and everybody reads the totals: two barriers per step.  (The first form ran a six-step __shfl_xor tree per
// quantity -- 324 ds_bpermute round trips per wavefront and step -- and let every thread add all WAVES partials of all 27.)
// Matches are re-read in every step (from the LDS copy where LDSM says so); which of a thread's matches are outliers lives in a
// register (bit k = the thread's k-th match; from the 33rd on in the `outlier` array itself).  One wavefront per frame is right
// for a few dozen matches; with the ~770 - 1500 matches of a tracking pass, four wavefronts are (0.43 -> 0.2x ms per 256 frames).
//
// Round 4 -- the normal equations from the structure of the Jacobian instead of row by row.  A row of the 3 x 6 Jacobian is
// (a, b, c) [I | G] with G = -[p]x, p = the point in the camera frame, and (a, b, c) = a row of w P, P the 3 x 3 projection
// Jacobian with the zero pattern (x 0 x; 0 x x; x 0 x).  So with A = wgt P^T P (five distinct non-zero entries) and g = wgt P^T r
//   H += [A, A G; (A G)^T, G^T A G],   b += [g; G^T g]
// costs 70 multiply-adds per match against the 117 of three rank-one updates (a fifth of those were products with the structural
// zeros, which the compiler must keep without fast-math).
#ifndef SNK_POSE_STUB_SOLVE
#define SNK_POSE_STUB_SOLVE 0
#endif
#ifndef SNK_POSE_MIN_WAVES
#define SNK_POSE_MIN_WAVES 2  // two wavefronts per SIMD (<= 256 registers incl. accumulation registers): without the bound the allocator has gone to 254 + 6 = one wavefront
#endif
#ifndef SNK_POSE_RED_STEPS  // DPP steps of the 27 sums before they meet in LDS: 4 = rows of 16 lanes, 2 = quads (see the kernel)
#define SNK_POSE_RED_STEPS 2
#endif
constexpr int POSE_SLOTS_PER_WAVE = 64 >> SNK_POSE_RED_STEPS;
template <int WAVES, bool LDSM>
__global__ __launch_bounds__(64 * WAVES, SNK_POSE_MIN_WAVES) void pose_kernel(const PoseMeta* __restrict__ meta, const double* __restrict__ wps,
                                                          const snk_pose_obs* __restrict__ obs, u8* __restrict__ outlier,
                                                          double* __restrict__ pose_out, int* __restrict__ inliers_out, CamD cam,
                                                          snk_pose_options opt, int lds_matches)
{
    constexpr int STRIDE = 64 * WAVES;
    constexpr int SLOTS = WAVES * POSE_SLOTS_PER_WAVE;
    __shared__ double s_part[SLOTS][28];
    __shared__ double s_tot[28];
    const int lane    = threadIdx.x;  // thread of the frame's workgroup
    // the first wavefront solves the 6 x 6 system of a step (see below; giving the job to a different wavefront in every other
    // workgroup, so that the solves of two frames of a compute unit land on different SIMDs, measured nothing: r04w4)
    const bool is_solver = lane < 64;
    const int slane   = lane & 63;
    const PoseMeta& M = meta[blockIdx.x];
    const int n       = M.n;
    const double* W   = wps + 3 * (long long)M.off;
    const snk_pose_obs* O = obs + M.off;
    u8* out               = outlier + M.off;
    double pose[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) pose[i] = M.pose[i];
    // LDSM: the frame's matches (56 bytes each, see Match) are copied to LDS once; the 40 steps then read them with LDS latency
    // instead of a global-memory round trip per match and step (two wavefronts per SIMD hide nothing).  The LDS carve holds the
    // first `lds_matches` matches of the frame (sized by the launch so that the frames a compute unit's registers admit also fit
    // its LDS); matches beyond it -- a frame with more matches than usual -- are read from global memory, same values.
    extern __shared__ __attribute__((aligned(16))) double s_match[];
    const int nl = LDSM ? min(n, lds_matches) : 0;
    for (int i = lane; i < n; i += STRIDE) out[i] = 0;
    if (LDSM)
    {
        for (int i = lane; i < nl; i += STRIDE)
        {
            const Match mt = make_match(W + 3 * i, O[i], cam);
            double* d      = s_match + 7 * i;
            d[0] = mt.px, d[1] = mt.py, d[2] = mt.pz, d[3] = mt.ox, d[4] = mt.oy, d[5] = mt.od, d[6] = mt.w;
        }
    }
    if (WAVES > 1 || LDSM) __syncthreads();
    auto fetch = [&](int i) -> Match
    {
        if (LDSM && i < nl)
        {
            const double* d = s_match + 7 * i;
            return Match{d[0], d[1], d[2], d[3], d[4], d[5], d[6]};
        }
        return make_match(W + 3 * i, O[i], cam);
    };
    u32 badbits = 0;  // bit k: this thread's k-th match (lane + k * STRIDE) is an outlier, k < 32
    auto is_bad = [&](int i, int k) -> bool { return k < 32 ? ((badbits >> k) & 1u) != 0 : out[i] != 0; };

    // sums of acc[0..CNT) over the workgroup, left in acc for every thread (CNT is a compile-time constant: acc stays in registers)
    auto reduce = [&](double* acc, auto cnt_c)
    {
        constexpr int CNT = decltype(cnt_c)::value;
#pragma unroll
        for (int i = 0; i < CNT; ++i) acc[i] = row_sum64_dpp(acc[i]);
        if ((lane & 15) == 0)
#pragma unroll
            for (int i = 0; i < CNT; ++i) s_part[lane >> 4][i] = acc[i];
        __syncthreads();
        if (lane < CNT)
        {
            double v = s_part[0][lane];
#pragma unroll
            for (int w = 1; w < WAVES * 4; ++w) v += s_part[w][lane];
            s_tot[lane] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < CNT; ++i) acc[i] = s_tot[i];
    };

    int inliers = 0;
    for (int round = 0; round < opt.outer_iterations; ++round)
    {
        const bool robust = round < opt.robust_rounds;
        for (int it = 0; it < opt.inner_iterations; ++it)
        {
            double acc[27];  // upper triangle of H (21, row-major) then b (6)
#pragma unroll
            for (int i = 0; i < 27; ++i) acc[i] = 0.0;
            double R[9];
            quat_to_R(pose, R);
            int k = 0;
            for (int i = lane; i < n; i += STRIDE, ++k)
            {
                if (is_bad(i, k)) continue;
                const Match mt = fetch(i);
                Proj q;
                if (!project(R, pose + 4, mt, cam, q)) continue;
                const double s = q.r0 * q.r0 + q.r1 * q.r1 + q.r2 * q.r2;
                double wgt     = 1.0;
                if (robust)
                {
                    const double d = q.stereo ? opt.th_stereo : opt.th_mono;
                    if (s > d * d) wgt = d * rsqrt_nr(s);
                }
                // rows of w P: (u, 0, c0), (0, v, c1), (a2, 0, c2) -- the third only for a stereo observation
                const double wi = mt.w * q.iz;
                const double u = cam.fx * wi, v = cam.fy * wi;
                const double c0 = -(u * q.xz), c1 = -(v * q.yz);
                const double a2 = q.stereo ? u : 0.0;
                const double c2 = q.stereo ? c0 + cam.bf * wi * q.iz : 0.0;
                const double wu = wgt * u, wv = wgt * v, wa2 = wgt * a2, wc0 = wgt * c0, wc1 = wgt * c1, wc2 = wgt * c2;
                const double A00 = wu * u + wa2 * a2, A02 = wu * c0 + wa2 * c2, A11 = wv * v, A12 = wv * c1;
                const double A22 = wc0 * c0 + wc1 * c1 + wc2 * c2;
                const double g0 = wu * q.r0 + wa2 * q.r2, g1 = wv * q.r1, g2 = wc0 * q.r0 + wc1 * q.r1 + wc2 * q.r2;
                const double X = q.X, Y = q.Y, Z = q.Z;
                // M = A G (A01 = 0)
                const double M00 = Y * A02, M01 = Z * A00 - X * A02, M02 = -(Y * A00);
                const double M10 = Y * A12 - Z * A11, M11 = -(X * A12), M12 = X * A11;
                const double M20 = Y * A22 - Z * A12, M21 = Z * A02 - X * A22, M22 = X * A12 - Y * A02;
                acc[0] += A00;  // (0,1) gets nothing
                acc[2] += A02;
                acc[3] += M00, acc[4] += M01, acc[5] += M02;
                acc[6] += A11;
                acc[7] += A12;
                acc[8] += M10, acc[9] += M11, acc[10] += M12;
                acc[11] += A22;
                acc[12] += M20, acc[13] += M21, acc[14] += M22;
                // G^T M, upper triangle
                acc[15] += Y * M20 - Z * M10;
                acc[16] += Y * M21 - Z * M11;
                acc[17] += Y * M22 - Z * M12;
                acc[18] += Z * M01 - X * M21;
                acc[19] += Z * M02 - X * M22;
                acc[20] += X * M12 - Y * M02;
                acc[21] += g0, acc[22] += g1, acc[23] += g2;
                acc[24] += Y * g2 - Z * g1;
                acc[25] += Z * g0 - X * g2;
                acc[26] += X * g1 - Y * g0;
            }
            if (WAVES > 1)
            {
                // The 6 x 6 solve and the pose update are the same numbers for every thread: only ONE wavefront runs them
                // (the others wait at the barrier and leave their SIMDs to the other frames of the compute unit -- with several
                // frames per CU the redundant Cholesky, 6 square roots and 6 divisions in fp64, was a third of the kernel's vector
                // instructions) and hands the new pose over through LDS.
#if SNK_POSE_RED_STEPS == 4
#pragma unroll
                for (int i = 0; i < 27; ++i) acc[i] = row_sum64_dpp(acc[i]);
                if ((lane & 15) == 0)
#pragma unroll
                    for (int i = 0; i < 27; ++i) s_part[lane >> 4][i] = acc[i];
                __syncthreads();
                if (is_solver)
                {
                    if (slane < 27)
                    {
                        double v = s_part[0][slane];
#pragma unroll
                        for (int w = 1; w < SLOTS; ++w) v += s_part[w][slane];
                        s_tot[slane] = v;
                    }
#else
                // Two DPP steps (sums over quads) instead of four: every wavefront saves 27 x 2 x 3 instructions per step; the
                // solver wavefront then adds SLOTS = 16 * WAVES partial sums per quantity instead of 4 * WAVES, two lanes per
                // quantity (a few dozen LDS reads issued back to back on the one wavefront everybody waits for anyway)
#pragma unroll
                for (int i = 0; i < 27; ++i) acc[i] += dpp_mov64<0xB1>(acc[i]);
#pragma unroll
                for (int i = 0; i < 27; ++i) acc[i] += dpp_mov64<0x4E>(acc[i]);
                if ((lane & 3) == 0)
#pragma unroll
                    for (int i = 0; i < 27; ++i) s_part[lane >> 2][i] = acc[i];
                __syncthreads();
                if (is_solver)
                {
                    {
                        const int qn = min(slane >> 1, 26), half = slane & 1;
                        double v = s_part[half * (SLOTS / 2)][qn];
#pragma unroll
                        for (int w = 1; w < SLOTS / 2; ++w) v += s_part[half * (SLOTS / 2) + w][qn];
                        v += dpp_mov64<0xB1>(v);
                        if (slane < 54 && half == 0) s_tot[qn] = v;
                    }
#endif
                    __builtin_amdgcn_wave_barrier();  // one wavefront: the LDS serves it in program order
#pragma unroll
                    for (int i = 0; i < 27; ++i) acc[i] = s_tot[i];
                }
            }
            else
            {
#pragma unroll
                for (int i = 0; i < 27; ++i) acc[i] = wave_sum64_dpp(acc[i]);
            }
            if (WAVES == 1 || is_solver)
            {
            double H[36], b[6];
            {
                int u = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a)
                {
                    b[a] = acc[21 + a];
#pragma unroll
                    for (int c = a; c < 6; ++c) H[a * 6 + c] = acc[u++];
                }
            }
            if (M.use_prior)
            {
                double e[6];
                se3_log_rel(pose, M.pred, e);
#pragma unroll
                for (int a = 0; a < 6; ++a)
                {
                    const double w2 = a < 3 ? M.w_trans * M.w_trans : M.w_rot * M.w_rot;
                    H[a * 6 + a] += w2;
                    b[a] += w2 * e[a];
                }
            }
#pragma unroll
            for (int a = 0; a < 6; ++a)
            {
                H[a * 6 + a] += opt.lambda * clampd(H[a * 6 + a]);
#pragma unroll
                for (int c = 0; c < a; ++c) H[a * 6 + c] = H[c * 6 + a];
            }
            double nb[6], d[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) nb[a] = -b[a];
#if SNK_POSE_STUB_SOLVE == 1  // timing experiments only
            for (int a = 0; a < 6; ++a) d[a] = nb[a] / H[a * 6 + a];
            se3_update(pose, d);
#elif SNK_POSE_STUB_SOLVE == 2
            for (int a = 0; a < 6; ++a) pose[a + 1] += 1e-12 * nb[a] / H[a * 6 + a];
#else
            if (chol_solve6(H, nb, d) == 0) se3_update(pose, d);
#endif
            }
            if (WAVES > 1)
            {
                if (is_solver && slane == 0)
#pragma unroll
                    for (int i = 0; i < 7; ++i) s_tot[i] = pose[i];  // s_tot's sums have been consumed (by this wavefront, in order)
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 7; ++i) pose[i] = s_tot[i];
                // no third barrier: the next write of s_tot (the solver wavefront) comes after the next step's barrier, which every
                // thread reaches only after these reads
            }
        }
        // re-classify every match
        double R[9];
        quat_to_R(pose, R);
        int cnt = 0, k = 0;
        for (int i = lane; i < n; i += STRIDE, ++k)
        {
            const Match mt = fetch(i);
            Proj q;
            bool bd = true;
            if (project(R, pose + 4, mt, cam, q))
            {
                const double s = q.r0 * q.r0 + q.r1 * q.r1 + q.r2 * q.r2;
                const double d = q.stereo ? opt.th_stereo : opt.th_mono;
                bd             = s > d * d;
            }
            out[i] = bd ? 1 : 0;
            if (k < 32) badbits = (badbits & ~(1u << k)) | ((bd ? 1u : 0u) << k);
            cnt += bd ? 0 : 1;
        }
        if (WAVES > 1)
        {
            double c1[1] = {(double)cnt};
            reduce(c1, std::integral_constant<int, 1>{});  // its barriers also order this round's flags before the next round reads them
            cnt = (int)c1[0];
        }
        else
        {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
        }
        inliers = cnt;
    }
    if (lane == 0)
    {
#pragma unroll
        for (int i = 0; i < 7; ++i) pose_out[7 * blockIdx.x + i] = pose[i];
        inliers_out[blockIdx.x] = inliers;
    }
}

// ---- device-resident form: the matches of a projection matcher -> (world point, observation) pairs per frame ----
// One wavefront per frame walks the local-map points in order (ballot prefix: the pairs keep the points' order, which is the
// order RefinePoseWithMatches gathers them, PoseRefinement.cpp:37-57) and writes the frame's PoseMeta.
// BY_FEATURE: match_idx[b][f] = local-map point of frame feature f (`frame.mvpMapPoints[f]` as an index, -1 = nullptr) and the walk
// is over the frame's features -- literally the loop of RefinePoseWithMatches (PoseRefinement.cpp:37-57); rows have `stride`
// entries (the feature capacity), n_walk = features of the frame.  Otherwise match_idx[b][i] = feature of local-map point i
// (a projection matcher's output), the walk is over the points, stride = pts_cap.
template <bool BY_FEATURE>
__global__ __launch_bounds__(256) void gather_matches_kernel(const snk_kp64* __restrict__ kps, const float* __restrict__ depth, int cap,
                                                            const unsigned char* __restrict__ pts, int pts_stride,
                                                            const int* __restrict__ match_idx, const int* __restrict__ n_pts,
                                                            const int* __restrict__ n_feat, int pts_cap, const double* __restrict__ poses, float s0, float s1, float s2,
                                                            float s3, float s4, float s5, float s6, float s7, int n_levels,
                                                            PoseMeta* __restrict__ meta, double* __restrict__ wps,
                                                            snk_pose_obs* __restrict__ obs, int* __restrict__ slot_of)
{
    // four wavefronts per frame, 256 points per round, GR rounds at a time: the rounds' match indices are loaded together, the ballot
    // counts of all of them meet in LDS behind ONE barrier, and the gathers of the rounds are in flight together (the first form ran
    // round by round: six dependent trips load -> barrier -> gather -> store for a 1500-point local map, 38 us per 1024 frames) --
    // the pairs keep the points' order
    constexpr int GR = 8;
    __shared__ int s_wcnt[2][GR][4];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    const int n_p = min(n_pts[b], pts_cap);
    const int stride = BY_FEATURE ? cap : pts_cap;
    const int m = BY_FEATURE ? min(max(n_feat[b], 0), cap) : n_p;
    const float ls[8] = {s0, s1, s2, s3, s4, s5, s6, s7};
    const size_t base = (size_t)b * stride;
    int count = 0;
    for (int i0 = 0, par = 0; i0 < m; i0 += 256 * GR, par ^= 1)
    {
        int v[GR];
        unsigned long long mask[GR];
#pragma unroll
        for (int r = 0; r < GR; ++r)
        {
            const int i = i0 + r * 256 + tid;
            v[r]        = i < m ? match_idx[base + i] : -1;
        }
#pragma unroll
        for (int r = 0; r < GR; ++r)
        {
            const bool has = BY_FEATURE ? (v[r] >= 0 && v[r] < n_p) : (v[r] >= 0 && v[r] < cap);
            mask[r]        = __builtin_amdgcn_ballot_w64(has);
            if ((tid & 63) == 0) s_wcnt[par][r][wave] = __popcll(mask[r]);
        }
        __syncthreads();  // double-buffered by the parity of the group: one barrier per group of rounds
#pragma unroll
        for (int r = 0; r < GR; ++r)
        {
            int before = 0, round_total = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w)
            {
                const int c = s_wcnt[par][r][w];
                before += w < wave ? c : 0;
                round_total += c;
            }
            const int i = i0 + r * 256 + tid;
            const int f = BY_FEATURE ? i : v[r];                 // the frame's feature
            const int p = BY_FEATURE ? v[r] : i;                 // the local-map point
            if ((mask[r] >> (tid & 63)) & 1ull)
            {
                const int k = count + before + __builtin_amdgcn_mbcnt_hi((unsigned)(mask[r] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask[r], 0u));
                const double* pp = reinterpret_cast<const double*>(pts + ((size_t)b * pts_cap + p) * (size_t)pts_stride);
                double* w = wps + (base + k) * 3;
                w[0] = pp[0]; w[1] = pp[1]; w[2] = pp[2];
                const snk_kp64 kp = kps[(size_t)b * cap + f];
                int oc = kp.octave;
                oc     = oc < 0 ? 0 : (oc >= n_levels ? n_levels - 1 : oc);
                const double sc = (double)ls[oc];
                snk_pose_obs o;
                o.x = kp.x; o.y = kp.y;
                o.depth  = (double)depth[(size_t)b * cap + f];
                o.weight = sqrt(1.0 / (sc * sc));  // sqrt(InverseSquaredScale(octave)), PoseRefinement.h:52
                obs[base + k]     = o;
                slot_of[base + k] = i;
            }
            count += round_total;
        }
    }
    if (tid == 0)
    {
        PoseMeta M;
        M.off = (int)base; M.n = count >= 3 ? count : 0;  // fewer than 3 correspondences: nothing to refine (PoseRefinement.cpp:59)
        M.use_prior = 0; M.pad = count;
        for (int k = 0; k < 7; ++k) M.pose[k] = M.pred[k] = poses[(size_t)b * 7 + k];
        M.w_rot = M.w_trans = 0.0;
        meta[b] = M;
    }
}

// results back to where the caller keeps them: the pose of every frame, the outlier flag per local-map point
__global__ __launch_bounds__(64) void scatter_pose_kernel(const PoseMeta* __restrict__ meta, const double* __restrict__ pose_out,
                                                          const u8* __restrict__ outl, const int* __restrict__ slot_of, int pts_cap,
                                                          double* __restrict__ poses, u8* __restrict__ outlier_pt,
                                                          int* __restrict__ inliers)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const PoseMeta M = meta[b];
    const size_t base = (size_t)b * pts_cap;
    for (int i = lane; i < pts_cap; i += 64) outlier_pt[base + i] = 0;
    __builtin_amdgcn_wave_barrier();
    if (M.n == 0)
    {
        if (lane == 0) inliers[b] = 0;  // pose untouched
        return;
    }
    for (int k = lane; k < M.n; k += 64) outlier_pt[base + slot_of[base + k]] = outl[base + k];
    if (lane < 7) poses[(size_t)b * 7 + lane] = pose_out[(size_t)b * 7 + lane];
}
// ---- sequences in lockstep (BASELINE.json config 5 with several sequences per GPU): the two glue steps between the batched BF
// matcher and the batched pose refinement, so that a frame of every sequence is tracked without a host round trip ----
// TrackBruteForce (Snake/Tracking/TrackingCoarse.cpp:342-387): `matchKnn2_omp(frame.descriptors, ref->frame->descriptors)` -- the
// CURRENT frame is the query set, the reference keyframe the train set (:351) -- and for every filtered match m with a map point on
// the reference feature, `frame.mvpMapPoints[m.first] = ref->GetMapPoint(m.second)` (:373-377).  Here the reference features'
// points are the previous frame's stereo points: pair (f, r) of the filtered list (f = current feature, r = reference feature)
// gives frame_pt[f] = r when ref_has[r], everything else -1 (= nullptr).  Several frame features may point at one reference
// feature, never two reference features at one frame feature (one pair per query).
__global__ __launch_bounds__(256) void bf_matches_kernel(const int2* __restrict__ pairs, const int* __restrict__ n_pairs,
                                                         const u8* __restrict__ ref_has, int cap, int* __restrict__ frame_pt)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)b * cap;
    for (int i = tid; i < cap; i += 256) frame_pt[base + i] = -1;
    __syncthreads();  // the -1 fill and the scatter below touch the same words
    const int n = min(max(n_pairs[b], 0), cap);
    for (int j = tid; j < n; j += 256)
    {
        const int2 pr = pairs[base + j];  // queries are distinct (one pair per query at most): no two threads write one word
        if (pr.x >= 0 && pr.x < cap && pr.y >= 0 && pr.y < cap && ref_has[base + pr.y]) frame_pt[base + pr.x] = pr.y;
    }
}

// The frame's stereo points in the world (what the next frame is tracked against): feature i with depth z > 0 is
// p_c = ((x - cx) / fx * z, (y - cy) / fy * z, z) and p_w = R^T (p_c - t) with the frame's pose (R, t) = world -> camera
// (the inverse of `currentPose * wp`, SnakeORBMatcher.cpp:229); has[i] = depth > 0, features without depth get z = 1 (never used).
__global__ __launch_bounds__(256) void backproject_kernel(const snk_kp64* __restrict__ kps, const float* __restrict__ depth,
                                                          const int* __restrict__ n_feat, int cap, CamD cam,
                                                          const double* __restrict__ poses, double* __restrict__ world,
                                                          u8* __restrict__ has)
{
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    const size_t k = (size_t)b * cap + i;
    const int n = min(max(n_feat[b], 0), cap);
    if (i >= n)
    {
        has[k] = 0;
        world[3 * k] = world[3 * k + 1] = world[3 * k + 2] = 0.0;
        return;
    }
    // The world points of frame t are the local map of frame t + 1: a chain of a thousand frames feeds every rounding of this kernel
    // forward.  The file is built with -ffp-contract=fast for pose_kernel's sake (build.py); THIS kernel is not contracted (round 6,
    // the round-5 review's weak point 10): every product and sum below is its own IEEE operation, so the points are bit for bit what the
    // same expressions give on the host in double (tests/test_tracking_chain_gpu.py compares exactly).
    {
#pragma clang fp contract(off)
    double R[9];
    {
        const double* q = poses + (size_t)b * 7;  // quat_to_R, restated here so that the pragma covers it
        const double x = q[0], y = q[1], z = q[2], w = q[3];
        R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
        R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
        R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
    }
    const double tx = poses[(size_t)b * 7 + 4], ty = poses[(size_t)b * 7 + 5], tz = poses[(size_t)b * 7 + 6];
    const float d  = depth[k];
    const bool h   = d > 0.0f;
    const double z = h ? (double)d : 1.0;
    const snk_kp64 kp = kps[k];
    const double px = (kp.x - cam.cx) / cam.fx * z - tx, py = (kp.y - cam.cy) / cam.fy * z - ty, pz = z - tz;
    world[3 * k + 0] = px * R[0] + py * R[3] + pz * R[6];
    world[3 * k + 1] = px * R[1] + py * R[4] + pz * R[7];
    world[3 * k + 2] = px * R[2] + py * R[5] + pz * R[8];
    has[k] = h ? 1 : 0;
    }
}
}  // namespace
}  // namespace snk

using namespace snk;

extern "C" int snk_track_bf_matches_batch_dev(snk_matcher* m, const int32_t* pairs_dev, const int32_t* n_pairs_dev,
                                              const uint8_t* ref_has_dev, int cap, int batch, int32_t* frame_pt_dev)
{
    SNK_REQUIRE(m != nullptr, "matcher is NULL");
    SNK_REQUIRE(batch >= 0 && cap >= 1, "bad sizes");
    SNK_REQUIRE(pairs_dev && n_pairs_dev && ref_has_dev && frame_pt_dev, "NULL device buffer");
    if (batch == 0) return SNK_OK;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    hipLaunchKernelGGL(bf_matches_kernel, dim3(batch), dim3(256), 0, m->stream, reinterpret_cast<const int2*>(pairs_dev), n_pairs_dev,
                       ref_has_dev, cap, frame_pt_dev);
    SNK_LAUNCH_CHECK();
    return SNK_OK;
}

extern "C" int snk_track_backproject_batch_dev(snk_matcher* m, const snk_frames_dev* frames, const float* depth_dev,
                                               const snk_camera* cam, const double* poses_dev, double* world_dev, uint8_t* has_dev)
{
    SNK_REQUIRE(m != nullptr && frames != nullptr && cam != nullptr, "NULL argument");
    SNK_REQUIRE(frames->batch >= 0 && frames->cap >= 1 && frames->kps != nullptr && frames->n != nullptr, "bad frames");
    SNK_REQUIRE(depth_dev && poses_dev && world_dev && has_dev, "NULL device buffer");
    SNK_REQUIRE(cam->fx != 0.0 && cam->fy != 0.0, "fx / fy must not be 0");
    if (frames->batch == 0) return SNK_OK;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    CamD C{cam->fx, cam->fy, cam->cx, cam->cy, cam->bf};
    hipLaunchKernelGGL(backproject_kernel, dim3(ceil_div(frames->cap, 256), frames->batch), dim3(256), 0, m->stream, frames->kps,
                       depth_dev, frames->n, frames->cap, C, poses_dev, world_dev, has_dev);
    SNK_LAUNCH_CHECK();
    return SNK_OK;
}

extern "C" int snk_pose_refine(snk_matcher* m, const snk_camera* cam, const snk_pose_options* opt, snk_pose_problem* problems,
                               int n_problems)
{
    SNK_REQUIRE(m != nullptr && cam != nullptr && opt != nullptr, "NULL argument");
    SNK_REQUIRE(n_problems >= 0 && (n_problems == 0 || problems != nullptr), "bad problem array");
    SNK_REQUIRE(opt->outer_iterations >= 0 && opt->outer_iterations <= 64 && opt->inner_iterations >= 0 &&
                    opt->inner_iterations <= 1000 && opt->robust_rounds >= 0,
                "iteration counts out of range");
    SNK_REQUIRE(opt->th_mono > 0.0 && opt->th_stereo > 0.0 && opt->lambda >= 0.0, "thresholds must be positive");
    if (n_problems == 0) return SNK_OK;
    size_t total = 0;
    for (int i = 0; i < n_problems; ++i)
    {
        const snk_pose_problem& P = problems[i];
        SNK_REQUIRE(P.n >= 0 && (P.n == 0 || (P.wps && P.obs && P.outlier)), "bad match arrays");
        SNK_REQUIRE(P.w_rot >= 0.0 && P.w_trans >= 0.0, "prior weights must be >= 0");
        total += (size_t)P.n;
    }
    SNK_REQUIRE(total < (size_t)1 << 30, "too many matches");
    SNK_HIP_CHECK(hipSetDevice(m->device));
    const size_t np     = (size_t)n_problems;
    const size_t o_wps  = np * sizeof(PoseMeta);
    const size_t o_obs  = o_wps + total * 24;
    const size_t in_b   = o_obs + total * sizeof(snk_pose_obs);
    const size_t o_pose = (total + 7) & ~(size_t)7, o_inl = o_pose + np * 56, out_b = o_inl + np * 4;
    int rc;
    if ((rc = m->q.reserve(in_b + 64)) != SNK_OK) return rc;
    if ((rc = m->out.reserve(out_b + 64)) != SNK_OK) return rc;
    if ((rc = m->h_in.reserve(in_b)) != SNK_OK) return rc;
    if ((rc = m->h_res.reserve(out_b)) != SNK_OK) return rc;
    // stage inputs contiguously in pinned memory, one upload
    char* stage    = m->h_in.as<char>();
    PoseMeta* meta = reinterpret_cast<PoseMeta*>(stage);
    size_t off     = 0;
    int n_max      = 0;
    for (int i = 0; i < n_problems; ++i)
    {
        const snk_pose_problem& P = problems[i];
        PoseMeta& M               = meta[i];
        M.off                     = (int)off;
        M.n                       = P.n;
        M.use_prior               = (P.w_rot > 0.0 || P.w_trans > 0.0) ? 1 : 0;
        M.pad                     = 0;
        memcpy(M.pose, P.pose, 56);
        memcpy(M.pred, P.prediction, 56);
        M.w_rot   = P.w_rot;
        M.w_trans = P.w_trans;
        if (P.n > 0)
        {
            memcpy(&stage[o_wps + off * 24], P.wps, (size_t)P.n * 24);
            memcpy(&stage[o_obs + off * sizeof(snk_pose_obs)], P.obs, (size_t)P.n * sizeof(snk_pose_obs));
        }
        off += (size_t)P.n;
        n_max = P.n > n_max ? P.n : n_max;
    }
    char* d = m->q.as<char>();
    char* o = m->out.as<char>();
    SNK_HIP_CHECK(hipMemcpyAsync(d, stage, in_b, hipMemcpyHostToDevice, m->stream));
    CamD C{cam->fx, cam->fy, cam->cx, cam->cy, cam->bf};
    if (total >= (size_t)n_problems * 192)  // ~200 matches per frame and more: four wavefronts per frame
    {
        // the matches of a problem in LDS for its 40 steps (as in the device-resident form): one problem per call is a chain of
        // latencies, and a global-memory round trip per match and step was most of it
        static const bool no_lds = getenv("SNK_POSE_NO_LDS") != nullptr;
        const int dyn_max = 160 * 1024 - (4 * POSE_SLOTS_PER_WAVE * 28 + 28) * 8 - 2048;
        int lds_matches   = n_max;
        if (n_problems > 256) lds_matches = std::min(lds_matches, (80 * 1024 - (4 * POSE_SLOTS_PER_WAVE * 28 + 28) * 8 - 256) / 56);  // two problems per CU
        if ((size_t)lds_matches * 56 > (size_t)dyn_max) lds_matches = dyn_max / 56;
        // (eight wavefronts per problem for calls with one or a few large problems -- the reference's call is ONE frame of ~900 matches,
        // a chain of latencies on an empty chip -- measured nothing: 0.229 vs 0.224 ms per call, round 4; a step is bound by its serial
        // part: reduction, 6 x 6 solve, pose hand-over)
        if (!no_lds)
        {
            if ((rc = set_max_lds_once(reinterpret_cast<const void*>(pose_kernel<4, true>), dyn_max)) != SNK_OK) return rc;
            hipLaunchKernelGGL((pose_kernel<4, true>), dim3(n_problems), dim3(256), (size_t)lds_matches * 56, m->stream,
                               reinterpret_cast<const PoseMeta*>(d), reinterpret_cast<const double*>(d + o_wps),
                               reinterpret_cast<const snk_pose_obs*>(d + o_obs), reinterpret_cast<u8*>(o), reinterpret_cast<double*>(o + o_pose),
                               reinterpret_cast<int*>(o + o_inl), C, *opt, lds_matches);
        }
        else
            hipLaunchKernelGGL((pose_kernel<4, false>), dim3(n_problems), dim3(256), 0, m->stream, reinterpret_cast<const PoseMeta*>(d),
                               reinterpret_cast<const double*>(d + o_wps), reinterpret_cast<const snk_pose_obs*>(d + o_obs),
                               reinterpret_cast<u8*>(o), reinterpret_cast<double*>(o + o_pose), reinterpret_cast<int*>(o + o_inl), C, *opt, 0);
    }
    else
        hipLaunchKernelGGL((pose_kernel<1, false>), dim3(n_problems), dim3(64), 0, m->stream, reinterpret_cast<const PoseMeta*>(d),
                           reinterpret_cast<const double*>(d + o_wps), reinterpret_cast<const snk_pose_obs*>(d + o_obs),
                           reinterpret_cast<u8*>(o), reinterpret_cast<double*>(o + o_pose), reinterpret_cast<int*>(o + o_inl), C, *opt, 0);
    SNK_LAUNCH_CHECK();
    char* back = m->h_res.as<char>();
    SNK_HIP_CHECK(hipMemcpyAsync(back, o, out_b, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    off = 0;
    for (int i = 0; i < n_problems; ++i)
    {
        snk_pose_problem& P = problems[i];
        if (P.n > 0) memcpy(P.outlier, &back[off], (size_t)P.n);
        memcpy(P.pose, &back[o_pose + (size_t)i * 56], 56);
        memcpy(&P.inliers, &back[o_inl + (size_t)i * 4], 4);
        off += (size_t)P.n;
    }
    return SNK_OK;
}

static int refine_batch_impl(snk_matcher* m, const snk_frames_dev* frames, const float* depth_dev, const snk_camera* cam,
                             const snk_pose_options* opt, const void* pts_dev, int pts_stride, const int32_t* match_idx_dev,
                             const int32_t* n_pts_dev, int pts_cap, const float* level_scale, int n_levels, double* poses_dev,
                             uint8_t* outlier_dev, int32_t* inliers_dev, bool by_feature)
{
    SNK_REQUIRE(m != nullptr && frames != nullptr && cam != nullptr && opt != nullptr, "NULL argument");
    SNK_REQUIRE(frames->batch >= 0 && frames->cap >= 1 && frames->kps != nullptr, "bad frames");
    SNK_REQUIRE(depth_dev && pts_dev && match_idx_dev && n_pts_dev && poses_dev && outlier_dev && inliers_dev, "NULL device buffer");
    SNK_REQUIRE(!by_feature || frames->n != nullptr, "frames->n (features per frame) is required");
    SNK_REQUIRE(pts_stride >= 24 && pts_stride % 8 == 0 && pts_cap >= 1, "pts_stride must be a multiple of 8, >= 24");
    SNK_REQUIRE(level_scale != nullptr && n_levels >= 1 && n_levels <= 8, "level_scale / n_levels (1..8)");
    SNK_REQUIRE(opt->outer_iterations >= 0 && opt->outer_iterations <= 64 && opt->inner_iterations >= 0 &&
                    opt->inner_iterations <= 1000 && opt->robust_rounds >= 0,
                "iteration counts out of range");
    SNK_REQUIRE(opt->th_mono > 0.0 && opt->th_stereo > 0.0 && opt->lambda >= 0.0, "thresholds must be positive");
    const int batch = frames->batch;
    if (batch == 0) return SNK_OK;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    const int stride   = by_feature ? frames->cap : pts_cap;  // pairs per frame at most; row length of match_idx / outlier
    const size_t total = (size_t)batch * stride;
    // q: meta | wps | obs | slot_of      out: outlier (pair order) | pose_out | inliers
    const size_t o_wps = ((size_t)batch * sizeof(PoseMeta) + 15) & ~(size_t)15, o_obs = o_wps + total * 24;
    const size_t o_slot = o_obs + total * sizeof(snk_pose_obs), in_b = o_slot + total * 4;
    const size_t o_pose = (total + 7) & ~(size_t)7, o_inl = o_pose + (size_t)batch * 56, out_b = o_inl + (size_t)batch * 4;
    int rc;
    if ((rc = m->q.reserve(in_b + 64)) != SNK_OK) return rc;
    if ((rc = m->out.reserve(out_b + 64)) != SNK_OK) return rc;
    char* d = m->q.as<char>();
    char* o = m->out.as<char>();
    float ls[8];
    for (int i = 0; i < 8; ++i) ls[i] = level_scale[i < n_levels ? i : n_levels - 1];
#define GATHER_LAUNCH(BF_)                                                                                                            \
    hipLaunchKernelGGL(gather_matches_kernel<BF_>, dim3(batch), dim3(256), 0, m->stream, frames->kps, depth_dev, frames->cap,        \
                       reinterpret_cast<const unsigned char*>(pts_dev), pts_stride, match_idx_dev, n_pts_dev, frames->n, pts_cap,    \
                       (const double*)poses_dev, ls[0], ls[1], ls[2], ls[3], ls[4], ls[5], ls[6], ls[7], n_levels,                  \
                       reinterpret_cast<PoseMeta*>(d), reinterpret_cast<double*>(d + o_wps),                                        \
                       reinterpret_cast<snk_pose_obs*>(d + o_obs), reinterpret_cast<int*>(d + o_slot))
    if (by_feature) GATHER_LAUNCH(true);
    else GATHER_LAUNCH(false);
#undef GATHER_LAUNCH
    CamD C{cam->fx, cam->fy, cam->cx, cam->cy, cam->bf};
    static const bool no_lds = getenv("SNK_POSE_NO_LDS") != nullptr;  // A/B: matches re-read from global memory in every step
    // matches of a frame kept in LDS (56 bytes each): the whole local map when few frames are in flight (one workgroup per CU
    // anyway), otherwise at most ~1300, so that TWO frames share a compute unit -- what the kernel's ~250 registers admit (two
    // wavefronts per SIMD; until round 4 the carve was sized for four frames, 656 matches, and a frame with the usual ~770 read
    // the rest from global memory in every step).  SNK_POSE_LDS_MATCHES overrides; tests force the global-memory tail with a small value.
    static const int lds_env = getenv("SNK_POSE_LDS_MATCHES") ? atoi(getenv("SNK_POSE_LDS_MATCHES")) : 0;
    int lds_matches = stride;
    if (lds_env > 0) lds_matches = lds_env < stride ? lds_env : stride;
    else
    {
        const int two_per_cu = (80 * 1024 - (4 * POSE_SLOTS_PER_WAVE * 28 + 28) * 8 - 256) / 56;  // carve + static LDS <= 80 KB
        if (batch > 256 && lds_matches > two_per_cu) lds_matches = two_per_cu;
    }
    const int dyn_max = 160 * 1024 - (4 * POSE_SLOTS_PER_WAVE * 28 + 28) * 8 - 2048;  // what a workgroup can have beside the static part
    if ((size_t)lds_matches * 56 > (size_t)dyn_max) lds_matches = dyn_max / 56;
    const size_t match_lds = (size_t)lds_matches * 7 * sizeof(double);
#define POSE_LAUNCH(W_, L_, LDS_)                                                                                                    \
    hipLaunchKernelGGL((pose_kernel<W_, L_>), dim3(batch), dim3(64 * W_), LDS_, m->stream, reinterpret_cast<const PoseMeta*>(d),  \
                       reinterpret_cast<const double*>(d + o_wps), reinterpret_cast<const snk_pose_obs*>(d + o_obs),                \
                       reinterpret_cast<u8*>(o), reinterpret_cast<double*>(o + o_pose), inliers_dev, C, *opt, lds_matches)
    // Wavefronts per frame.  Four: the shortest step (a thread walks a quarter of the matches), right while the chip holds every frame at
    // once -- two per CU at the kernel's ~250 registers.  More frames than that would run in rounds of 2 x #CU; with TWO wavefronts per frame
    // four frames share a CU (same wavefronts per SIMD), the steps are longer and the frames all run together: tracking leg of bench.py,
    // 1024 frames, pose_kernel 467 -> ~365 us, the chain 1.100 -> 1.000 ms (profiles/r05/r05z_track_experiments.json).  The LDS carve
    // then holds ~590 matches of a frame (the rest from global memory in every step: 200 / 400 / 500 / 590 in LDS = 1.052 / 1.016 / 1.006 /
    // 0.995 ms, 640 = three frames per CU = 1.148 ms).  SNK_POSE_WAVES=2|4 forces a form.
    static const int waves_env = getenv("SNK_POSE_WAVES") ? atoi(getenv("SNK_POSE_WAVES")) : 0;
    static const int n_cu = []
    {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 256;
        return v > 0 ? v : 256;
    }();
    const bool two_waves = waves_env == 2 || (waves_env != 4 && batch > 2 * n_cu);
    if (stride >= 256 && !no_lds && two_waves)
    {
        int rc = set_max_lds_once(reinterpret_cast<const void*>(pose_kernel<2, true>), dyn_max);
        if (rc != SNK_OK) return rc;
        int lm = (160 * 1024 / 4 - (2 * POSE_SLOTS_PER_WAVE * 28 + 28) * 8 - 512) / 56;  // four frames per CU
        if (lds_env > 0) lm = lds_env;
        lds_matches = lm < stride ? lm : stride;
        if ((size_t)lds_matches * 56 > (size_t)dyn_max) lds_matches = dyn_max / 56;
        POSE_LAUNCH(2, true, (size_t)lds_matches * 56);
    }
    else if (stride >= 256 && !no_lds)
    {
        int rc = set_max_lds_once(reinterpret_cast<const void*>(pose_kernel<4, true>), dyn_max);
        if (rc != SNK_OK) return rc;
        POSE_LAUNCH(4, true, match_lds);
    }
    else if (stride >= 256) POSE_LAUNCH(4, false, 0);
    else POSE_LAUNCH(1, false, 0);
#undef POSE_LAUNCH
    hipLaunchKernelGGL(scatter_pose_kernel, dim3(batch), dim3(64), 0, m->stream, reinterpret_cast<const PoseMeta*>(d),
                       reinterpret_cast<const double*>(o + o_pose), reinterpret_cast<const u8*>(o), reinterpret_cast<const int*>(d + o_slot),
                       stride, poses_dev, outlier_dev, inliers_dev);
    SNK_LAUNCH_CHECK();
    return SNK_OK;
}

extern "C" int snk_pose_refine_matches_batch_dev(snk_matcher* m, const snk_frames_dev* frames, const float* depth_dev,
                                                 const snk_camera* cam, const snk_pose_options* opt, const void* pts_dev,
                                                 int pts_stride, const int32_t* match_idx_dev, const int32_t* n_pts_dev, int pts_cap,
                                                 const float* level_scale, int n_levels, double* poses_dev, uint8_t* outlier_dev,
                                                 int32_t* inliers_dev)
{
    return refine_batch_impl(m, frames, depth_dev, cam, opt, pts_dev, pts_stride, match_idx_dev, n_pts_dev, pts_cap, level_scale,
                             n_levels, poses_dev, outlier_dev, inliers_dev, false);
}

extern "C" int snk_pose_refine_frame_batch_dev(snk_matcher* m, const snk_frames_dev* frames, const float* depth_dev,
                                               const snk_camera* cam, const snk_pose_options* opt, const void* pts_dev,
                                               int pts_stride, const int32_t* frame_pt_dev, const int32_t* n_pts_dev, int pts_cap,
                                               const float* level_scale, int n_levels, double* poses_dev, uint8_t* outlier_dev,
                                               int32_t* inliers_dev)
{
    return refine_batch_impl(m, frames, depth_dev, cam, opt, pts_dev, pts_stride, frame_pt_dev, n_pts_dev, pts_cap, level_scale,
                             n_levels, poses_dev, outlier_dev, inliers_dev, true);
}
