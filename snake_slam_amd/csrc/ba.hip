// Local bundle adjustment for gfx950 — replaces Saiga::BARecRel::create / initAndSolve / solve and
// Scene::residual2/3 as Snake uses them (reference Snake/Optimizer/LocalBundleAdjustment.cpp:
// options :47-64, scene :187-293, solve :353-413, chi-square passes :372-395,:423-457).
//
// Semantics: "snk-ba v1" (DESIGN.md §BA): Levenberg-Marquardt, Huber IRLS, explicit Schur
// complement on the points, block-Jacobi PCG on the reduced camera system, all fp64.
//
// Mapping to the hardware.  A local window is small (<= 36 free cameras, a few thousand points,
// ~10^4 observations), so the design goals are (1) no host round trip inside the LM loop — every
// decision (accept / reject, lambda) is taken on the device, the host only enqueues a fixed
// kernel sequence per iteration; (2) bit-reproducible sums — every reduction has a fixed order:
// observations are stored sorted by point, per-camera and per-block sums walk precomputed index
// lists, no floating-point atomics; (3) many windows per launch (blockIdx.y = problem) so that
// disjoint keyframe windows / sequences fill the 256 CUs.
//
//   point_wave    wavefront per <= 64 observations of whole points : lane = observation (residual,
//                 Jacobians, W; rows leave through an LDS transpose as full lines), lane = point (V, b_p,
//                 damping, V^-1, cost).  point_pass<0> (thread per point) is the fallback for points with
//                 more than 64 observations
//   rpc_pass      thread per relative pose constraint (IMU scenes) : cost, -J^T r, J1^T J1, J1^T W
//   cam_pass      workgroup per free camera : U, b_c, rhs = b_c - sum Y b_p (+ constraint terms), fixed-order sums;
//                 J_c, r, Y b_p rebuilt from static camera-ordered records + one 48-byte gather per observation
//   schur_set     point-major Schur pass (batches, global BA): wavefront per <= 50 points of one camera set, the point's
//   + schur_sum   rows staged once in LDS, (pair, row) work units with their sums in registers; then wavefront per upper
//                 block adds the partial sums in fixed order: S = U - sum (W V^-1) W^T (+ constraint cross blocks)
//   schur_pass    block-major form of the same (wavefront per upper camera-pair block over its co-observation list):
//                 single windows and problems outside schur_set's limits; XCD-per-window launch for batches
//   pcg_solve     one workgroup per problem : block-Jacobi PCG, vectors (and S when it fits) in LDS
//   pcgl_*        multi-workgroup PCG for reduced systems beyond the LDS (global BA): vectors in HBM,
//                 (row chunk x column part) matvec, fixed-order partial sums, 5 launches per iteration
//   update_wave / update_pass : back-substitution of the points / trial poses (SE3 exp)
//   cost_wave     robust cost at the trial state (point_pass<1> as fallback); rpc_pass(trial) for constraints
//   accept_pass   one workgroup per problem : fixed-order cost sums, accept / reject, lambda schedule
//   point_pass<2> chi-square per caller observation (snk_ba_residuals)
#include "common.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <map>
#include <unordered_map>
#include <tuple>
#include <utility>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// This file is compiled with -ffp-contract=fast (snake_slam_amd/build.py): BA is specified by a tolerance, its kernels
// are bound by fp64 issue slots, and a * b + c as one v_fma_f64 halves them.  Results stay deterministic.

// Occupancy experiments (build-time A/B): minimum wavefronts per SIMD the register allocator must leave room
// for in cam_pass / update_cost (1 = no constraint: the allocator's own choice, two wavefronts per SIMD at ~195 registers).
#ifndef SNK_BA_CAM_WAVES
#define SNK_BA_CAM_WAVES 1
#endif
#ifndef SNK_SF_SKIP
#define SNK_SF_SKIP 0  // timing experiments only (results are then meaningless): schur_fused without 1 = its matrix products, 2 = phase 2b, 4 = the linearisation, 8 = phase 2a
#endif
#ifndef SNK_BA_CAM_BUTTERFLY
#define SNK_BA_CAM_BUTTERFLY 0
#endif
#ifndef SNK_BA_UC_WAVES
#define SNK_BA_UC_WAVES 1
#endif

namespace snk
{
namespace
{
struct Prob
{
    int ni, np, no;     // images, points, valid observations (sorted by point)
    int nfc, n6;        // free cameras, 6 * nfc
    int img_off, pt_off, obs_off, cam_off;
    int ptstart_off, camstart_off, citem_off, blkstart_off, ent_off;
    int vec_off;        // n6-vectors (rhs, x)
    long long s_off;    // S (n6 * n6 doubles)
    int orig_off;       // first caller-order observation of this problem
    int n_wv;           // wavefront work items of point_wave (whole points, <= 64 observations each); 0: not available
    int wv_off;
    int n_rpc;          // valid relative pose constraints (IMU scenes)
    int rpc_off;        // into rpc_meta / rpc_out
    int camrpc_off;     // into cam_rpc_start (nfc + 1 entries per problem)
    int n_set;          // work items of schur_set (0: not available for this problem)
    int set_off;        // into set_items
    int cblk_off;       // into cblk_start (nfc * nfc + 1 entries per problem)
    int ccam_off;       // into cc_start (nfc + 1 entries per problem): the per-camera lists of cam_part partial sums
    int be_nch;         // device-built block entries: 64-item chunks of the longest camera list
    int becnt_off;      // ... and this problem's [nfc][be_nch][nfc] counters
    double K[4];
    double bf;
};

struct Opt
{
    int max_pcg;
    int pcg_general;  // SNK_BA_PCG_GENERAL=1: the 256-thread PCG loop for every size (A/B against the replicated one)
    double pcg_tol, huber_mono, huber_stereo, lambda_init;
    double chi2_mono, chi2_stereo;  // point_pass<3> (the chi-square pass of SolveLocalScene): thresholds of the squared residual
};

struct State  // per problem, device resident
{
    double cost, cost_new, lambda, vfac, cost_initial;
    int accepted, iter, pcg_iters;
    int marked;  // observations the chi-square pass after this solve marked (point_pass<3>); begin_solve resets it
    double first_cost_initial, first_cost;  // the costs at the time of that pass (a conditional extra iteration overwrites the others)
};

struct RpcMeta
{
    int img1, img2;  // image index inside the problem
    int c1, c2;      // free-camera index or -1
    double rel[7];
    double w_rot, w_trans;
};
constexpr int RPC_STRIDE = 72;

// What cam_pass needs of an observation that never changes during a solve, stored in the order of the camera lists
// so that a camera's workgroup streams it: 40 bytes (round 5; 48 before: cam_pass and update_cost run at the speed their records
// stream at, so the index fields are packed -- the point index and "the point is an unknown" share a word, the image is the
// camera's and is looked up once per camera).
struct CamObs
{
    double u, v, depth, weight;
    int ptw;   // point index | "the point is an unknown" << 31
    int orig;  // caller-order index (global)
};

// One wavefront's share of the point-major Schur pass: <= SET_CHUNK points that are all observed by the same cameras
// in the same order (same "camera set"), so that lane q owns pair slot q = rows (ra, rb) of a point's run for all of
// them and the 6 x 6 sum of block (camera(ra), camera(rb)) stays in its registers.
struct SetItem
{
    int pts_off, n_pts;    // into set_pts; n_pts <= SET_CHUNK <= 64 (one list entry per lane)
    int pair_off, npairs;  // into set_pairs: ra | rb << 8, camera(ra) < camera(rb) or ra == rb
    int part_off;          // first of its npairs partial sums in s_part (36 doubles each)
    int run;               // observations per point of this set
    int nfree;             // free-camera observations per point (k)
    int aux_off;           // into set_pairs: k run positions ordered by camera index, then the k x k table "pair slot of (i, j)", i <= j
    int rec_off;           // into set_obs: n_pts x run static observation records in (point of the item, observation) order
    int cpart_off;         // first of its nfree per-camera partial sums in cam_part (schur_fused<3, true>; 33 doubles each)
};

// camera sums of schur_fused<3, true>: 33 terms per (work item, free camera), in passes of eight: 0..7, 8..15, 16..23,
// 24..26 (from J_c and r) and 27..32 (W V^-1 b_p)
constexpr int CS_TERMS = 33, CS_PASSES = 5;

// Static part of an observation in the order schur_fused walks it (item, point of the item, observation of the point): a lane's
// record is at rec_off + group * run + lane, so the first round of loads of a group is three coalesced 16-byte loads.
// In memory 40 bytes (round 5; 48 before): image, free-camera index and the "unknown" flag share a word.  SET_MAX_IMG bounds the images of a
// problem for which the packed form exists (snk_ba_set_problems refuses more).
constexpr int SET_MAX_IMG = 32767;
struct SetObs
{
    double u, v, depth, weight;
    int orig;  // caller-order index (global)
    int pk;    // image inside the problem (15 bits) | point is an unknown << 15 | (free-camera index + 1) << 16
};
struct SetRec  // the same in a lane's registers (the index word stays packed: two registers fewer per record set than four ints)
{
    double u, v, depth, weight;
    int orig, pk;
    __device__ __forceinline__ int img() const { return pk & 0x7FFF; }           // image inside the problem
    __device__ __forceinline__ int cam() const { return (int)((unsigned)pk >> 16) - 1; }  // free-camera index or -1
    __device__ __forceinline__ bool ptfree() const { return (pk & 0x8000) != 0; }  // the point is an unknown
};
__host__ __device__ inline int set_pack(int img, int cam, int ptfree) { return img | (ptfree ? 1 << 15 : 0) | ((cam + 1) << 16); }

struct Arrays
{
    const Prob* prob;
    State* state;
    double* pose;  // [img][7]
    double* pose_new;
    double* pt;  // [pt][3]
    double* pt_new;
    const unsigned char* pt_const;
    const int* cam_idx;  // [img] free-camera index or -1
    const int* pt_start;
    // observations sorted by point
    const int* o_img;
    const int* o_cam;               // free-camera index of the observation's image, or -1
    const unsigned char* o_ptfree;  // 1 when the observation's point is an unknown
    const double2* o_uv;
    const double* o_depth;
    const double* o_weight;
    const int* o_orig;             // caller-order index (global over problems)
    const int* o_pt;               // point index (inside the problem) of the observation
    const int* wv_pt;              // [n_wv + 1] per problem: first point of every point_wave work item
    // relative pose constraints
    const RpcMeta* rpc_meta;       // [rpc]
    double* rpc_out;               // [rpc][RPC_STRIDE]: cost, trial cost, g1[6], g2[6], H11 upper[21], H12[36]
    const int* cam_rpc_start;      // [nfc + 1] per problem
    const int* cam_rpc_items;      // rpc index (inside the problem) * 2 + side (0: the camera is img1, 1: img2)
    const int* blk_rpc;            // [nfc * nfc] per problem (at blkstart_off - problem index): 0 or 1 + (rpc * 2 + transposed)
    const int* rpc_next;           // [rpc] chain of further constraints on the same camera pair, same encoding
    const unsigned char* outlier;  // caller order
    double* o_r;   // [obs][4]  scaled residual, [3] = dim (0: inactive in this iteration)
    double* o_W;   // [obs][18]
    double* ptv;   // [pt][6]   point position of the linearisation | V^-1 b_p (cam_pass rebuilds J_c, r, Y b_p from them)
    const CamObs* cs_obs;  // static observation records in camera order (indexed like cam_items)
    double* Vinv;  // [pt][6]
    double* bp;    // [pt][3]
    double* cost_pt;
    double* cost_pt_new;
    double* U;  // [cam][36] damped
    const int* cam_start;
    const int* cam_items;
    const int* blk_start;
    const int4* blk_ent;  // (observation of c1, observation of c2, point, 0), observation indices relative to the problem
    const SetItem* set_items;
    const int2* set_pts;  // (point, first observation of the point) inside the problem
    const int* set_pairs;
    const SetObs* set_obs;
    const int* cblk_start;  // per problem, per block: its partial sums in cblk_items (fixed order)
    const int* cblk_items;  // index into s_part
    double* s_part;         // [partial][36]
    const int* cc_start;    // per problem, per free camera: its per-item partial sums in cc_items (fixed order)
    const int* cc_items;    // index into cam_part
    double* cam_part;       // [partial][33]: b_c (6) | U upper (21) | Y b_p (6) of one camera over one work item's points
    double* S;
    double* rhs;
    double* x;
    double* chi2;  // caller order
};

#ifndef SNK_BA_IEEE_DIV
#define SNK_BA_IEEE_DIV 0
#endif
__device__ __forceinline__ void quat_to_R(const double* q, double* R)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

// residual (dim 2|3) and Jacobians; returns dim or 0 (not in front of the camera)
template <bool JAC>
__device__ __forceinline__ int obs_linearize(const double* pose, const double* R, const double* pt, const double* K, double bf,
                                             double u, double v, double depth, double w, double* r, double* Jc, double* Jp)
{
    const double X = R[0] * pt[0] + R[1] * pt[1] + R[2] * pt[2] + pose[4];
    const double Y = R[3] * pt[0] + R[4] * pt[1] + R[5] * pt[2] + pose[5];
    const double Z = R[6] * pt[0] + R[7] * pt[1] + R[8] * pt[2] + pose[6];
    if (Z <= 0.0) return 0;
    // round 4: reciprocals / reciprocal square roots by rcp_nr / rsqrt_nr (common.hpp: hardware approximation + two Newton steps, within
    // an ulp or two) instead of the IEEE division / square root the compiler expands to 14 - 18 dependent instructions each -- an
    // observation needs two divisions here and a division and two roots in huber_rho, in each of the three kernels that linearise it.
    // "snk-ba v1" is specified by tolerances.  SNK_BA_IEEE_DIV=1 (build-time A/B) keeps the divisions.
    const double iz = SNK_BA_IEEE_DIV ? 1.0 / Z : rcp_nr(Z), iz2 = iz * iz;
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const int dim = depth > 0.0 ? 3 : 2;
    r[0] = w * (fx * X * iz + cx - u);
    r[1] = w * (fy * Y * iz + cy - v);
    r[2] = dim == 3 ? w * ((fx * X * iz + cx - bf * iz) - (u - (SNK_BA_IEEE_DIV ? bf / depth : bf * rcp_nr(depth)))) : 0.0;
    if (JAC)
    {
        double P[9];
        P[0] = fx * iz; P[1] = 0.0;     P[2] = -fx * X * iz2;
        P[3] = 0.0;     P[4] = fy * iz; P[5] = -fy * Y * iz2;
        P[6] = fx * iz; P[7] = 0.0;     P[8] = -fx * X * iz2 + bf * iz2;
#pragma unroll
        for (int k = 0; k < 3; ++k)
        {
            const double m = k < dim ? w : 0.0;
            const double a = m * P[3 * k], b = m * P[3 * k + 1], c = m * P[3 * k + 2];
            Jc[6 * k + 0] = a;
            Jc[6 * k + 1] = b;
            Jc[6 * k + 2] = c;
            Jc[6 * k + 3] = -b * Z + c * Y;
            Jc[6 * k + 4] = a * Z - c * X;
            Jc[6 * k + 5] = -a * Y + b * X;
            Jp[3 * k + 0] = a * R[0] + b * R[3] + c * R[6];
            Jp[3 * k + 1] = a * R[1] + b * R[4] + c * R[7];
            Jp[3 * k + 2] = a * R[2] + b * R[5] + c * R[8];
        }
    }
    return dim;
}

// ---- the observation Jacobians by their structure (round 5; the batch kernels schur_fused, cam_pass, update_cost) ----
// With Xc = R p + t and the rows m_k = w * d proj_k / d Xc (k < dim; obs_linearize's P scaled by the weight)
//     J_c,k = [ m_k | Xc x m_k ]   (translation | rotation part of obs_linearize's Jc),      J_p,k = m_k^T R,
// and m_0 = (a0, 0, c0), m_1 = (0, b1, c1), m_2 = (a0, 0, c2) (stereo only).  Every product the solver takes of them is a function
// of the symmetric M = sum_k m_k m_k^T (M01 = 0), h = sum_k m_k r_k, Xc and R:
//     W = J_c^T J_p = [ N ; Xc x N ] with N = M R (cross product column by column),     V = J_p^T J_p = R^T N,      b_p = -R^T h,
//     U = J_c^T J_c = [ M, T ; T^T, Xc x T ] with T[i] = Xc x M[:, i],                   b_c = -[ h ; Xc x h ],
//     J_c^T J_p v = [ f ; Xc x f ] with f = M (R v),                                      J_p^T (J_c dc) = R^T M (dt + dr x Xc).
// ~80 instead of ~150 fp64 operations per observation in schur_fused, ~125 instead of ~210 in cam_pass, ~35 instead of ~110 in the
// back-substitution half of update_cost, and 5 + 3 + 3 live values instead of J_c (18) and J_p (9).  Algebraically identical; the
// rounding differs in the last bits (bundle adjustment is specified by a tolerance).  The residual -- and with
// it every cost -- is computed by the same expressions as obs_linearize.
struct ObsCore
{
    double X, Y, Z;             // the point in the camera frame
    double a0, c0, b1, c1, c2;  // the rows of w * d proj / d Xc (c2 = 0 for a monocular observation)
    int dim;
};

__device__ __forceinline__ int obs_core(const double* pose, const double* R, const double* pt, const double* K, double bf, double u,
                                        double v, double depth, double w, double* r, ObsCore& o)
{
    const double X = R[0] * pt[0] + R[1] * pt[1] + R[2] * pt[2] + pose[4];
    const double Y = R[3] * pt[0] + R[4] * pt[1] + R[5] * pt[2] + pose[5];
    const double Z = R[6] * pt[0] + R[7] * pt[1] + R[8] * pt[2] + pose[6];
    if (Z <= 0.0) return 0;
    const double iz = SNK_BA_IEEE_DIV ? 1.0 / Z : rcp_nr(Z), iz2 = iz * iz;
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const int dim = depth > 0.0 ? 3 : 2;
    r[0] = w * (fx * X * iz + cx - u);
    r[1] = w * (fy * Y * iz + cy - v);
    r[2] = dim == 3 ? w * ((fx * X * iz + cx - bf * iz) - (u - (SNK_BA_IEEE_DIV ? bf / depth : bf * rcp_nr(depth)))) : 0.0;
    o.X = X; o.Y = Y; o.Z = Z;
    o.a0 = w * (fx * iz);
    o.c0 = w * (-fx * X * iz2);
    o.b1 = w * (fy * iz);
    o.c1 = w * (-fy * Y * iz2);
    o.c2 = dim == 3 ? w * (-fx * X * iz2 + bf * iz2) : 0.0;
    o.dim = dim;
    return dim;
}

struct ObsMoments
{
    double M00, M02, M11, M12, M22;  // s2 * sum_k m_k m_k^T (M01 = 0)
    double h0, h1, h2;               // s2 * sum_k m_k r_k
};

// s2 = the squared IRLS scale (J_c, J_p and r each carry one factor of it)
__device__ __forceinline__ void obs_moments(const ObsCore& o, const double* r, double s2, ObsMoments& m)
{
    const double a2 = o.dim == 3 ? o.a0 : 0.0;
    m.M00 = s2 * (o.a0 * o.a0 + a2 * a2);
    m.M02 = s2 * (o.a0 * o.c0 + a2 * o.c2);
    m.M11 = s2 * (o.b1 * o.b1);
    m.M12 = s2 * (o.b1 * o.c1);
    m.M22 = s2 * (o.c0 * o.c0 + o.c1 * o.c1 + o.c2 * o.c2);
    m.h0  = s2 * (o.a0 * r[0] + a2 * r[2]);
    m.h1  = s2 * (o.b1 * r[1]);
    m.h2  = s2 * (o.c0 * r[0] + o.c1 * r[1] + o.c2 * r[2]);
}

// obs_linearize's J_c scaled by sw, from the core (only the camera-sum variant of schur_fused still wants the matrix itself)
__device__ __forceinline__ void core_to_Jc(const ObsCore& o, double sw, double* Jc)
{
    const double ma[3] = {sw * o.a0, 0.0, o.dim == 3 ? sw * o.a0 : 0.0};
    const double mb[3] = {0.0, sw * o.b1, 0.0};
    const double mc[3] = {sw * o.c0, sw * o.c1, sw * o.c2};
#pragma unroll
    for (int k = 0; k < 3; ++k)
    {
        const double a = ma[k], b = mb[k], c = mc[k];
        Jc[6 * k + 0] = a;
        Jc[6 * k + 1] = b;
        Jc[6 * k + 2] = c;
        Jc[6 * k + 3] = -b * o.Z + c * o.Y;
        Jc[6 * k + 4] = a * o.Z - c * o.X;
        Jc[6 * k + 5] = -a * o.Y + b * o.X;
    }
}

__device__ __forceinline__ double huber_rho(double s, double d, double& sqrt_w)
{
    const double d2 = d * d;
    if (s <= d2)
    {
        sqrt_w = 1.0;
        return s;
    }
#if SNK_BA_IEEE_DIV
    const double rt = sqrt(s);
    sqrt_w = sqrt(d / rt);
#else
    const double irt = rsqrt_nr(s), rt = s * irt, q = d * irt;  // s > d^2 > 0
    sqrt_w = q * rsqrt_nr(q);                                    // sqrt(d / sqrt(s))
#endif
    return 2.0 * d * rt - d2;
}

__device__ __forceinline__ double clampd(double v)
{
    return v < 1e-6 ? 1e-6 : (v > 1e32 ? 1e32 : v);
}

// MODE 0: linearise at the current state; MODE 1: robust cost at the trial state;
// MODE 2: chi-square (squared weighted residual norm) per caller observation at the current state
template <int MODE>
__global__ __launch_bounds__(128) void point_pass(Arrays A, Opt O)
{
    const int pb  = blockIdx.y;
    const Prob pr = A.prob[pb];
    const int p   = blockIdx.x * 128 + threadIdx.x;
    if (MODE == 3 && blockIdx.x == 0 && threadIdx.x == 0)  // the costs at the time of the chi-square pass (see State)
    {
        A.state[pb].first_cost_initial = A.state[pb].cost_initial;
        A.state[pb].first_cost         = A.state[pb].cost;
    }
    if (p >= pr.np) return;
    int marked          = 0;
    const int gp        = pr.pt_off + p;
    const double* poses = (MODE == 1 ? A.pose_new : A.pose) + (size_t)pr.img_off * 7;
    const double* ptp   = (MODE == 1 ? A.pt_new : A.pt) + (size_t)gp * 3;
    const double pt[3]  = {ptp[0], ptp[1], ptp[2]};
    const bool pfree    = !A.pt_const[gp];
    const int s0 = A.pt_start[pr.ptstart_off + p], s1 = A.pt_start[pr.ptstart_off + p + 1];
    double V[6] = {0, 0, 0, 0, 0, 0}, bp[3] = {0, 0, 0};
    double cost = 0.0;

    for (int s = s0; s < s1; ++s)
    {
        const int go = pr.obs_off + s;
        const int oo = A.o_orig[go];
        if (MODE == 0) A.o_r[(size_t)go * 4 + 3] = 0.0;
        if (A.outlier[oo])
        {
            if (MODE == 2) A.chi2[oo] = 0.0;
            continue;
        }
        const double* pose = poses + (size_t)A.o_img[go] * 7;
        double R[9];
        quat_to_R(pose, R);
        double r[3], Jc[18], Jp[9];
        const double2 uv = A.o_uv[go];
        const int dim = obs_linearize<MODE == 0>(pose, R, pt, pr.K, pr.bf, uv.x, uv.y, A.o_depth[go], A.o_weight[go], r, Jc, Jp);
        const double sq = dim ? r[0] * r[0] + r[1] * r[1] + r[2] * r[2] : 0.0;
        if (MODE == 2)
        {
            A.chi2[oo] = sq;
            continue;
        }
        if (MODE == 3)
        {
            // SolveLocalScene's chi-square pass (LocalBundleAdjustment.cpp:368-397): a valid observation that is not an outlier
            // yet (those were skipped above) becomes one when its squared residual exceeds the threshold of its kind
            if (sq > (A.o_depth[go] > 0.0 ? O.chi2_stereo : O.chi2_mono))
            {
                const_cast<unsigned char*>(A.outlier)[oo] = 1;
                ++marked;
            }
            continue;
        }
        if (!dim) continue;
        double sw;
        cost += huber_rho(sq, dim == 3 ? O.huber_stereo : O.huber_mono, sw);
        if (MODE == 0)
        {
#pragma unroll
            for (int k = 0; k < 3; ++k) r[k] *= sw;
#pragma unroll
            for (int k = 0; k < 18; ++k) Jc[k] *= sw;
#pragma unroll
            for (int k = 0; k < 9; ++k) Jp[k] *= sw;
            const int c = A.o_cam[go];
            double* rr  = A.o_r + (size_t)go * 4;
            rr[0] = r[0];
            rr[1] = r[1];
            rr[2] = r[2];
            rr[3] = (double)dim;
            if (pfree)
            {
                // V (upper: 00 01 02 11 12 22), b_p = -Jp^T r
                V[0] += Jp[0] * Jp[0] + Jp[3] * Jp[3] + Jp[6] * Jp[6];
                V[1] += Jp[0] * Jp[1] + Jp[3] * Jp[4] + Jp[6] * Jp[7];
                V[2] += Jp[0] * Jp[2] + Jp[3] * Jp[5] + Jp[6] * Jp[8];
                V[3] += Jp[1] * Jp[1] + Jp[4] * Jp[4] + Jp[7] * Jp[7];
                V[4] += Jp[1] * Jp[2] + Jp[4] * Jp[5] + Jp[7] * Jp[8];
                V[5] += Jp[2] * Jp[2] + Jp[5] * Jp[5] + Jp[8] * Jp[8];
#pragma unroll
                for (int a = 0; a < 3; ++a) bp[a] -= Jp[a] * r[0] + Jp[3 + a] * r[1] + Jp[6 + a] * r[2];
                if (c >= 0)
                {
                    double* Wp = A.o_W + (size_t)go * 18;
#pragma unroll
                    for (int a = 0; a < 6; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b) Wp[a * 3 + b] = Jc[a] * Jp[b] + Jc[6 + a] * Jp[3 + b] + Jc[12 + a] * Jp[6 + b];
                }
            }
        }
    }
    if (MODE == 2) return;
    if (MODE == 3)
    {
        if (marked) atomicAdd(&A.state[pb].marked, marked);  // integer: order independent
        return;
    }
    if (MODE == 1)
    {
        A.cost_pt_new[gp] = cost;
        return;
    }
    A.cost_pt[gp] = cost;
    {
        double* pv = A.ptv + (size_t)gp * 6;
        pv[0] = pt[0]; pv[1] = pt[1]; pv[2] = pt[2];
        pv[3] = pv[4] = pv[5] = 0.0;
    }
    if (!pfree) return;
    const double lambda = A.state[pb].lambda;
    V[0] += lambda * clampd(V[0]);
    V[3] += lambda * clampd(V[3]);
    V[5] += lambda * clampd(V[5]);
    double Vi[6];
    {
        const double a = V[0], b = V[1], c = V[2], d = V[3], e = V[4], f = V[5];
        const double Aa = d * f - e * e, Bb = c * e - b * f, Cc = b * e - c * d;
        const double det = a * Aa + b * Bb + c * Cc;
        const double id  = det == 0.0 ? 0.0 : 1.0 / det;
        Vi[0] = Aa * id;
        Vi[1] = Bb * id;
        Vi[2] = Cc * id;
        Vi[3] = (a * f - c * c) * id;
        Vi[4] = (b * c - a * e) * id;
        Vi[5] = (a * d - b * b) * id;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) A.Vinv[(size_t)gp * 6 + k] = Vi[k];
    A.bp[(size_t)gp * 3 + 0] = bp[0];
    A.bp[(size_t)gp * 3 + 1] = bp[1];
    A.bp[(size_t)gp * 3 + 2] = bp[2];
    // V^-1 b_p: cam_pass forms the camera right-hand sides' Y b_p = J_c^T (J_p V^-1 b_p) from it
    {
        double* pv = A.ptv + (size_t)gp * 6;
        pv[3] = Vi[0] * bp[0] + Vi[1] * bp[1] + Vi[2] * bp[2];
        pv[4] = Vi[1] * bp[0] + Vi[3] * bp[1] + Vi[4] * bp[2];
        pv[5] = Vi[2] * bp[0] + Vi[4] * bp[1] + Vi[5] * bp[2];
    }
}

// Linearisation (MODE 0 of point_pass) with one WAVEFRONT per <= 64 consecutive observations that
// belong to whole points.  point_pass gives a thread one point and lets it walk its observations: every
// lane then writes 8-byte words 144 bytes apart and the L2 sees ~1100 partial-line writes per 64
// observations.  Here lane = observation for the per-observation math (coalesced reads), lane = point
// for the small per-point part (V, b_p, V^-1; same summation order as point_pass, so the results are
// bit-identical), and the 18-double rows of W leave through an LDS transpose as full 128-byte
// lines.  LDS per wavefront: 14 + 18 doubles per lane = 16 KB.
constexpr int PW_JP = 14;  // Jp[9], r[3], cost, dim
__global__ __launch_bounds__(64) void point_wave(Arrays A, Opt O)
{
    // one buffer: the per-observation scratch of phases 1 / 2 (14 doubles per lane) is dead when phase 3 transposes the W
    // rows through it (18 per lane); 9 KB per wavefront instead of 16 lets 17 instead of 10 wavefronts share a CU
    __shared__ __attribute__((aligned(16))) double s_st[64 * 18];
    double* s_jp = s_st;
    const int lane = threadIdx.x;
    const int pb   = blockIdx.y;
    const Prob pr  = A.prob[pb];
    const int w    = blockIdx.x;
    if (w >= pr.n_wv) return;
    const int p0 = A.wv_pt[pr.wv_off + w], p1 = A.wv_pt[pr.wv_off + w + 1];
    const int sb = A.pt_start[pr.ptstart_off + p0], se = A.pt_start[pr.ptstart_off + p1];
    const int nob = se - sb, npt = p1 - p0;
    const double* poses = A.pose + (size_t)pr.img_off * 7;
    const size_t gbase  = (size_t)pr.obs_off + sb;  // first observation of the work item

    // ---- phase 1: lane = observation ----
    const bool act = lane < nob;
    const size_t go = gbase + (act ? lane : 0);
    double r[3] = {0, 0, 0}, Jc[18], Jp[9];
#pragma unroll
    for (int k = 0; k < 18; ++k) Jc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) Jp[k] = 0.0;
    int dim = 0, c = -1, lp = 0;
    double cost = 0.0;
    if (act)
    {
        // two rounds of loads instead of four dependent ones: everything indexed by the observation first, then the
        // gathers through those indices (point, pose, outlier flag); the flag only masks the result
        const int opt = A.o_pt[go], oimg = A.o_img[go], oorig = A.o_orig[go];
        c = A.o_cam[go];
        const double2 uv = A.o_uv[go];
        const double odepth = A.o_depth[go], oweight = A.o_weight[go];
        lp = opt - p0;
        const double* ptp  = A.pt + (size_t)(pr.pt_off + opt) * 3;
        const double pt[3] = {ptp[0], ptp[1], ptp[2]};
        const double* posep = poses + (size_t)oimg * 7;
        double pose[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) pose[k] = posep[k];
        const bool is_out = A.outlier[oorig] != 0;
        if (!is_out)
        {
            double R[9];
            quat_to_R(pose, R);
            dim = obs_linearize<true>(pose, R, pt, pr.K, pr.bf, uv.x, uv.y, odepth, oweight, r, Jc, Jp);
            if (dim)
            {
                const double sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
                double sw;
                cost = huber_rho(sq, dim == 3 ? O.huber_stereo : O.huber_mono, sw);
#pragma unroll
                for (int k = 0; k < 3; ++k) r[k] *= sw;
#pragma unroll
                for (int k = 0; k < 18; ++k) Jc[k] *= sw;
#pragma unroll
                for (int k = 0; k < 9; ++k) Jp[k] *= sw;
            }
            else
            {
                r[0] = r[1] = r[2] = 0.0;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) s_jp[lane * PW_JP + k] = Jp[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) s_jp[lane * PW_JP + 9 + k] = r[k];
    s_jp[lane * PW_JP + 12] = cost;
    s_jp[lane * PW_JP + 13] = (double)dim;
    // o_r (scaled residual | activity flag) is not written here: its readers are the fallback passes that run with
    // point_pass<0> (schur_pass with activity lookups, the point part of update_pass); with point_wave inactive rows
    // of W are zero and cam_pass / update_wave rebuild what they need
    __builtin_amdgcn_wave_barrier();

    // ---- phase 2: lane = point ----
    if (lane < npt)
    {
        const int p  = p0 + lane;
        const int gp = pr.pt_off + p;
        const bool pfree = !A.pt_const[gp];
        const int a0 = A.pt_start[pr.ptstart_off + p] - sb, a1 = A.pt_start[pr.ptstart_off + p + 1] - sb;
        double V[6] = {0, 0, 0, 0, 0, 0}, bp[3] = {0, 0, 0};
        double cst = 0.0;
        for (int a = a0; a < a1; ++a)
        {
            const double* q = s_jp + a * PW_JP;
            if (q[13] == 0.0) continue;
            cst += q[12];
            if (pfree)
            {
                V[0] += q[0] * q[0] + q[3] * q[3] + q[6] * q[6];
                V[1] += q[0] * q[1] + q[3] * q[4] + q[6] * q[7];
                V[2] += q[0] * q[2] + q[3] * q[5] + q[6] * q[8];
                V[3] += q[1] * q[1] + q[4] * q[4] + q[7] * q[7];
                V[4] += q[1] * q[2] + q[4] * q[5] + q[7] * q[8];
                V[5] += q[2] * q[2] + q[5] * q[5] + q[8] * q[8];
#pragma unroll
                for (int b = 0; b < 3; ++b) bp[b] -= q[b] * q[9] + q[3 + b] * q[10] + q[6 + b] * q[11];
            }
        }
        A.cost_pt[gp] = cst;
        double Vi[6] = {0, 0, 0, 0, 0, 0};
        double vb[3] = {0, 0, 0};
        if (pfree)
        {
            const double lambda = A.state[pb].lambda;
            V[0] += lambda * clampd(V[0]);
            V[3] += lambda * clampd(V[3]);
            V[5] += lambda * clampd(V[5]);
            const double a = V[0], b = V[1], cc = V[2], d = V[3], e = V[4], f = V[5];
            const double Aa = d * f - e * e, Bb = cc * e - b * f, Cc = b * e - cc * d;
            const double det = a * Aa + b * Bb + cc * Cc;
            const double id  = det == 0.0 ? 0.0 : 1.0 / det;
            Vi[0] = Aa * id;
            Vi[1] = Bb * id;
            Vi[2] = Cc * id;
            Vi[3] = (a * f - cc * cc) * id;
            Vi[4] = (b * cc - a * e) * id;
            Vi[5] = (a * d - b * b) * id;
#pragma unroll
            for (int k = 0; k < 6; ++k) A.Vinv[(size_t)gp * 6 + k] = Vi[k];
            A.bp[(size_t)gp * 3 + 0] = bp[0];
            A.bp[(size_t)gp * 3 + 1] = bp[1];
            A.bp[(size_t)gp * 3 + 2] = bp[2];
            vb[0] = Vi[0] * bp[0] + Vi[1] * bp[1] + Vi[2] * bp[2];
            vb[1] = Vi[1] * bp[0] + Vi[3] * bp[1] + Vi[4] * bp[2];
            vb[2] = Vi[2] * bp[0] + Vi[4] * bp[1] + Vi[5] * bp[2];
        }
        {
            // position of this linearisation | V^-1 b_p: what cam_pass gathers per observation instead of J_c, r, Y b_p
            const double* ptp = A.pt + (size_t)gp * 3;
            const double px = ptp[0], py = ptp[1], pz = ptp[2];  // loads first: pt and ptv may alias for the compiler
            double* pv        = A.ptv + (size_t)gp * 6;
            pv[0] = px; pv[1] = py; pv[2] = pz;
            pv[3] = vb[0]; pv[4] = vb[1]; pv[5] = vb[2];
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- phase 3: lane = observation: W = Jc^T Jp (zero rows for inactive couplings) ----
    const bool cpl = act && dim && c >= 0 && !A.pt_const[pr.pt_off + p0 + lp];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            s_st[lane * 18 + a * 3 + b] = cpl ? Jc[a] * Jp[b] + Jc[6 + a] * Jp[3 + b] + Jc[12 + a] * Jp[6 + b] : 0.0;
    __builtin_amdgcn_wave_barrier();
    {
        // 16 bytes per lane and store: the rows are a contiguous run of nob * 144 bytes on both sides
        double2* dst       = reinterpret_cast<double2*>(A.o_W + gbase * 18);
        const double2* src = reinterpret_cast<const double2*>(s_st);
        for (int i = lane; i < nob * 9; i += 64) dst[i] = src[i];
    }
}

// Back-substitution of the points with the point_wave work items: lane = observation forms W^T dc
// (three sums of six products), lane = point subtracts them from b_p in observation order and applies V^-1.
__device__ inline void trial_poses_block(const Arrays& A, int pb, int block);
__global__ __launch_bounds__(256) void update_wave(Arrays A, Opt O, int wave_blocks)
{
    __shared__ double s_t[4][64 * 3];
    if ((int)blockIdx.x >= wave_blocks)  // the workgroups behind the point work items update the poses (was a launch of its own)
    {
        trial_poses_block(A, blockIdx.y, (int)blockIdx.x - wave_blocks);
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pb   = blockIdx.y;
    const Prob pr  = A.prob[pb];
    const int w    = blockIdx.x * 4 + wave;
    if (w >= pr.n_wv) return;
    const int p0 = A.wv_pt[pr.wv_off + w], p1 = A.wv_pt[pr.wv_off + w + 1];
    const int sb = A.pt_start[pr.ptstart_off + p0], se = A.pt_start[pr.ptstart_off + p1];
    const int nob = se - sb, npt = p1 - p0;
    const double* x = A.x + pr.vec_off;
    // W^T dc = J_p^T (J_c dc): the Jacobians are rebuilt from the observation (48 coalesced bytes + cached gathers of
    // the point, the pose and dc) with the code of point_wave instead of fetching the 144-byte W row again -- the pass
    // was re-reading all of W (390 MB per 256 windows, 2.6 TB/s) for three numbers per observation.
    double t[3] = {0, 0, 0};
    if (lane < nob)
    {
        const size_t go = (size_t)pr.obs_off + sb + lane;
        const int c = A.o_cam[go], opt = A.o_pt[go], oimg = A.o_img[go], oorig = A.o_orig[go];
        const double2 uv = A.o_uv[go];
        const double odepth = A.o_depth[go], oweight = A.o_weight[go];
        const int cc = c < 0 ? 0 : c;
        const double* ptp  = A.pt + (size_t)(pr.pt_off + opt) * 3;
        const double pt[3] = {ptp[0], ptp[1], ptp[2]};
        const double* posep = A.pose + (size_t)(pr.img_off + oimg) * 7;
        double pose[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) pose[k] = posep[k];
        double xv[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) xv[a] = x[cc * 6 + a];
        const bool skip = A.outlier[oorig] != 0 || c < 0 || A.pt_const[pr.pt_off + opt];
        if (!skip)
        {
            double R[9], r[3], Jc[18], Jp[9];
            quat_to_R(pose, R);
            const int dim = obs_linearize<true>(pose, R, pt, pr.K, pr.bf, uv.x, uv.y, odepth, oweight, r, Jc, Jp);
            if (dim)
            {
                const double sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
                double sw;
                (void)huber_rho(sq, dim == 3 ? O.huber_stereo : O.huber_mono, sw);
                const double s2 = sw * sw;  // J_c and J_p each carry the IRLS scale
                double u[3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    u[k] = s2 * (Jc[6 * k] * xv[0] + Jc[6 * k + 1] * xv[1] + Jc[6 * k + 2] * xv[2] + Jc[6 * k + 3] * xv[3] + Jc[6 * k + 4] * xv[4] +
                                 Jc[6 * k + 5] * xv[5]);
#pragma unroll
                for (int b = 0; b < 3; ++b) t[b] = Jp[b] * u[0] + Jp[3 + b] * u[1] + Jp[6 + b] * u[2];
            }
        }
    }
    s_t[wave][lane * 3] = t[0];
    s_t[wave][lane * 3 + 1] = t[1];
    s_t[wave][lane * 3 + 2] = t[2];
    __builtin_amdgcn_wave_barrier();
    if (lane < npt)
    {
        const int p  = p0 + lane;
        const int gp = pr.pt_off + p;
        double* out  = A.pt_new + (size_t)gp * 3;
        const double* cur = A.pt + (size_t)gp * 3;
        if (A.pt_const[gp])
        {
            out[0] = cur[0];
            out[1] = cur[1];
            out[2] = cur[2];
        }
        else
        {
            double g[3] = {A.bp[(size_t)gp * 3], A.bp[(size_t)gp * 3 + 1], A.bp[(size_t)gp * 3 + 2]};
            const int a0 = A.pt_start[pr.ptstart_off + p] - sb, a1 = A.pt_start[pr.ptstart_off + p + 1] - sb;
            for (int a = a0; a < a1; ++a)
            {
                g[0] -= s_t[wave][a * 3];
                g[1] -= s_t[wave][a * 3 + 1];
                g[2] -= s_t[wave][a * 3 + 2];
            }
            const double* Vi = A.Vinv + (size_t)gp * 6;
            out[0] = cur[0] + (Vi[0] * g[0] + Vi[1] * g[1] + Vi[2] * g[2]);
            out[1] = cur[1] + (Vi[1] * g[0] + Vi[3] * g[1] + Vi[4] * g[2]);
            out[2] = cur[2] + (Vi[2] * g[0] + Vi[4] * g[1] + Vi[5] * g[2]);
        }
    }
}

// Robust cost at the trial state (MODE 1 of point_pass) with the point_wave work items: lane = observation
// evaluates the residual, lane = point adds the costs in observation order (bit-identical to point_pass<1>).
__global__ __launch_bounds__(256) void cost_wave(Arrays A, Opt O)
{
    __shared__ double s_c[4][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pb   = blockIdx.y;
    const Prob pr  = A.prob[pb];
    const int w    = blockIdx.x * 4 + wave;
    if (w >= pr.n_wv) return;
    const int p0 = A.wv_pt[pr.wv_off + w], p1 = A.wv_pt[pr.wv_off + w + 1];
    const int sb = A.pt_start[pr.ptstart_off + p0], se = A.pt_start[pr.ptstart_off + p1];
    const int nob = se - sb, npt = p1 - p0;
    double cost = 0.0;
    if (lane < nob)
    {
        // observation-indexed loads first, then the gathers through them (two dependent round trips, not four); the
        // outlier flag only masks the result
        const size_t go = (size_t)pr.obs_off + sb + lane;
        const int lp = A.o_pt[go], oimg = A.o_img[go], oorig = A.o_orig[go];
        const double2 uv = A.o_uv[go];
        const double odepth = A.o_depth[go], oweight = A.o_weight[go];
        const double* ptp   = A.pt_new + (size_t)(pr.pt_off + lp) * 3;
        const double pt[3]  = {ptp[0], ptp[1], ptp[2]};
        const double* posep = A.pose_new + ((size_t)pr.img_off + oimg) * 7;
        double pose[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) pose[k] = posep[k];
        if (!A.outlier[oorig])
        {
            double R[9], r[3], Jc[1], Jp[1];
            quat_to_R(pose, R);
            const int dim = obs_linearize<false>(pose, R, pt, pr.K, pr.bf, uv.x, uv.y, odepth, oweight, r, Jc, Jp);
            if (dim)
            {
                double sw;
                cost = huber_rho(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], dim == 3 ? O.huber_stereo : O.huber_mono, sw);
            }
        }
    }
    s_c[wave][lane] = cost;
    __builtin_amdgcn_wave_barrier();
    if (lane < npt)
    {
        const int p  = p0 + lane;
        const int a0 = A.pt_start[pr.ptstart_off + p] - sb, a1 = A.pt_start[pr.ptstart_off + p + 1] - sb;
        double c = 0.0;
        for (int a = a0; a < a1; ++a) c += s_c[wave][a];
        A.cost_pt_new[pr.pt_off + p] = c;
    }
}

// fixed-order sum of one double per thread over the workgroup: xor butterfly inside each wavefront
// (shuffles, no barrier), then the wavefront totals in order — 2 barriers instead of log2(THREADS)+1.
template <int THREADS>
__device__ __forceinline__ double block_sum(double v, double* red, int tid)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double t = red[0];
#pragma unroll
    for (int w = 1; w < THREADS / 64; ++w) t += red[w];
    __syncthreads();
    return t;
}

// two sums at once (same tree for each): one pair of barriers instead of two
template <int THREADS>
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red, int tid)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
    }
    if ((tid & 63) == 0)
    {
        red[tid >> 6]                = a;
        red[THREADS / 64 + (tid >> 6)] = b;
    }
    __syncthreads();
    double ta = red[0], tb = red[THREADS / 64];
#pragma unroll
    for (int w = 1; w < THREADS / 64; ++w)
    {
        ta += red[w];
        tb += red[THREADS / 64 + w];
    }
    __syncthreads();
    a = ta;
    b = tb;
}

// ---- relative pose constraints (IMU scenes): e = log(T2 T1^-1 rel^-1), r = W e ----
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

// r (6) and J1 = d r / d delta1 (6x6 row-major); d r / d delta2 = W
__device__ void rpc_linearize(const double* pose1, const double* pose2, const RpcMeta& c, double* r, double* J1)
{
    double T21[7], R21[9];
    {
        const double ax = pose2[0], ay = pose2[1], az = pose2[2], aw = pose2[3];
        const double bx = -pose1[0], by = -pose1[1], bz = -pose1[2], bw = pose1[3];
        T21[0] = aw * bx + ax * bw + ay * bz - az * by;
        T21[1] = aw * by - ax * bz + ay * bw + az * bx;
        T21[2] = aw * bz + ax * by - ay * bx + az * bw;
        T21[3] = aw * bw - ax * bx - ay * by - az * bz;
    }
    quat_to_R(T21, R21);
    T21[4] = pose2[4] - (R21[0] * pose1[4] + R21[1] * pose1[5] + R21[2] * pose1[6]);
    T21[5] = pose2[5] - (R21[3] * pose1[4] + R21[4] * pose1[5] + R21[5] * pose1[6]);
    T21[6] = pose2[6] - (R21[6] * pose1[4] + R21[7] * pose1[5] + R21[8] * pose1[6]);
    double e[6];
    se3_log_rel(T21, c.rel, e);
    const double wt = c.w_trans, wr = c.w_rot;
    for (int a = 0; a < 3; ++a)
    {
        r[a]     = wt * e[a];
        r[3 + a] = wr * e[3 + a];
    }
    const double tx = T21[4], ty = T21[5], tz = T21[6];
    const double K[9] = {0, -tz, ty, tz, 0, -tx, -ty, tx, 0};
    double KR[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) KR[i * 3 + j] = K[i * 3] * R21[j] + K[i * 3 + 1] * R21[3 + j] + K[i * 3 + 2] * R21[6 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
        {
            J1[i * 6 + j]           = -wt * R21[i * 3 + j];
            J1[i * 6 + 3 + j]       = -wt * KR[i * 3 + j];
            J1[(3 + i) * 6 + j]     = 0.0;
            J1[(3 + i) * 6 + 3 + j] = -wr * R21[i * 3 + j];
        }
}

// one thread per constraint.  trial == 0: residual, cost and the normal-equation terms at the current poses;
// trial == 1: cost at the trial poses.
__global__ __launch_bounds__(64) void rpc_pass(Arrays A, int trial)
{
    const int pb  = blockIdx.y;
    const Prob pr = A.prob[pb];
    const int k   = blockIdx.x * 64 + threadIdx.x;
    if (k >= pr.n_rpc) return;
    const RpcMeta m = A.rpc_meta[pr.rpc_off + k];
    const double* P = (trial ? A.pose_new : A.pose) + (size_t)pr.img_off * 7;
    double r[6], J1[36];
    rpc_linearize(P + (size_t)m.img1 * 7, P + (size_t)m.img2 * 7, m, r, J1);
    double cost = 0.0;
    for (int a = 0; a < 6; ++a) cost += r[a] * r[a];
    double* o = A.rpc_out + (size_t)(pr.rpc_off + k) * RPC_STRIDE;
    if (trial)
    {
        o[1] = cost;
        return;
    }
    o[0] = cost;
    const double w[6] = {m.w_trans, m.w_trans, m.w_trans, m.w_rot, m.w_rot, m.w_rot};
    for (int a = 0; a < 6; ++a)
    {
        double g = 0.0;
        for (int q = 0; q < 6; ++q) g += J1[q * 6 + a] * r[q];
        o[2 + a] = -g;            // b1 = -J1^T r
        o[8 + a] = -(w[a] * r[a]);  // b2 = -W r
    }
    int u = 14;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b)
        {
            double s2 = 0.0;
            for (int q = 0; q < 6; ++q) s2 += J1[q * 6 + a] * J1[q * 6 + b];
            o[u++] = s2;  // H11 upper
        }
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) o[35 + a * 6 + b] = J1[b * 6 + a] * w[b];  // H12 = J1^T W
}

// The end of a camera's pass over its observations: tot = U upper (21) | b_c (6) | sum Y b_p (6).  Adds the relative pose
// constraints incident to the camera, damps the diagonal, writes U and the reduced right-hand side.
__device__ inline void cam_finish(const Arrays& A, const Prob& pr, int pb, int c, double* tot)
{
    if (pr.n_rpc > 0)  // relative pose constraints incident to this camera (fixed order)
    {
        const int r0 = A.cam_rpc_start[pr.camrpc_off + c], r1 = A.cam_rpc_start[pr.camrpc_off + c + 1];
        for (int q = r0; q < r1; ++q)
        {
            const int item  = A.cam_rpc_items[q];
            const int k     = item >> 1;
            const double* o = A.rpc_out + (size_t)(pr.rpc_off + k) * RPC_STRIDE;
            if ((item & 1) == 0)
            {
                for (int u = 0; u < 21; ++u) tot[u] += o[14 + u];
                for (int a = 0; a < 6; ++a) tot[21 + a] += o[2 + a];
            }
            else
            {
                const RpcMeta m = A.rpc_meta[pr.rpc_off + k];
                const double w2[6] = {m.w_trans * m.w_trans, m.w_trans * m.w_trans, m.w_trans * m.w_trans,
                                      m.w_rot * m.w_rot,     m.w_rot * m.w_rot,     m.w_rot * m.w_rot};
                int u = 0;
                for (int a = 0; a < 6; ++a)
                    for (int b = a; b < 6; ++b, ++u)
                        if (a == b) tot[u] += w2[a];
                for (int a = 0; a < 6; ++a) tot[21 + a] += o[8 + a];
            }
        }
    }
    const double lambda = A.state[pb].lambda;
    double* U = A.U + (size_t)(pr.cam_off + c) * 36;
    int q = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b)
        {
            double v = tot[q++];
            if (a == b) v += lambda * clampd(v);
            U[a * 6 + b] = v;
            U[b * 6 + a] = v;
        }
    for (int a = 0; a < 6; ++a) A.rhs[pr.vec_off + c * 6 + a] = tot[21 + a] - tot[27 + a];
}

// a wave-uniform double into scalar registers
// (inline assembly: the builtin lets the compiler move the read in front of the arithmetic that produced the value and redo that
// arithmetic -- there is no scalar fp64 unit -- in vector registers)
__device__ __forceinline__ double uniform_f64(double v)
{
    int lo, hi;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(lo) : "v"(__double2loint(v)));
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(hi) : "v"(__double2hiint(v)));
    return __hiloint2double(hi, lo);
}

// Column sums of 33 values per lane over the 64 lanes of a wavefront, "transpose and add": every step pairs the lanes, each lane keeps
// one half of its values and hands the other half to its partner, so the number of live sums halves with the number of lanes that
// share them -- 17 + 9 + 5 + 3 + 2 + 1 = 37 additions instead of the 33 x 6 of a butterfly that carries every sum through all six
// steps.  Steps 1 and 2 are gfx950's v_permlane32_swap / v_permlane16_swap (two registers exchange halves / alternate rows: one
// instruction per 32-bit half, no select), steps 3-6 DPP moves inside a row of 16 (row_ror:8, row_half_mirror, quad_perm).  On return
// lane l holds the total of value idx = 17 b5 + 9 b4 + 5 b3 + 3 b2 + 2 b1 + b0 (the bits of l) when `ok` (the other lanes hold the
// zero padding of the odd splits).  The order of the additions is fixed (a tree over the lanes), like the butterfly's.
__device__ __forceinline__ double swap_add32(double a, double b)  // lanes < 32: a(l) + a(l + 32); lanes >= 32: b(l - 32) + b(l)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap_add16(double a, double b)  // even rows of 16: a(l) + a(l + 16); odd rows: b(l - 16) + b(l)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
template <int CTRL>
__device__ __forceinline__ double xchg_add(double lo, double hi, bool bit)  // bit = 0: lo(l) + lo(partner); bit = 1: hi(l) + hi(partner)
{
    const double keep = bit ? hi : lo, send = bit ? lo : hi;
    return keep + dpp_mov64<CTRL>(send);
}
__device__ __forceinline__ double wave_reduce33(const double (&acc)[33], int lane, int& idx, bool& ok)
{
    double a[17], b[9], c[5], d[3], e[2];
#pragma unroll
    for (int i = 0; i < 17; ++i) a[i] = swap_add32(acc[i], i + 17 < 33 ? acc[i + 17] : 0.0);
#pragma unroll
    for (int i = 0; i < 9; ++i) b[i] = swap_add16(a[i], i + 9 < 17 ? a[i + 9] : 0.0);
    const int b5 = (lane >> 5) & 1, b4 = (lane >> 4) & 1, b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1, b1 = (lane >> 1) & 1, b0 = lane & 1;
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = xchg_add<0x128>(b[i], i + 5 < 9 ? b[i + 5] : 0.0, b3 != 0);  // row_ror:8
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = xchg_add<0x141>(c[i], i + 3 < 5 ? c[i + 3] : 0.0, b2 != 0);  // row_half_mirror: l <-> 7 - l
#pragma unroll
    for (int i = 0; i < 2; ++i) e[i] = xchg_add<0x4E>(d[i], i + 2 < 3 ? d[i + 2] : 0.0, b1 != 0);   // quad_perm [2,3,0,1]
    const double f = xchg_add<0xB1>(e[0], e[1], b0 != 0);                                            // quad_perm [1,0,3,2]
    const int i3 = 2 * b1 + b0, i5 = 3 * b2 + i3, i9 = 5 * b3 + i5, i17 = 9 * b4 + i9;
    idx = 17 * b5 + i17;
    ok  = i3 < 3 && i5 < 5 && i9 < 9 && i17 < 17 && idx < 33;
    return f;
}

// CAM_THREADS threads per camera.  The 33 sums of a wavefront are reduced with a 6-step butterfly (~600 instructions), as much as
// linearising three observations: with many windows per launch ONE wavefront per camera (12 observations per thread for the
// benchmark window) is fastest -- 143 us (4 wavefronts) -> 102 (2) -> 86 (1) per 256 windows; a single window keeps 4 for latency.
template <int CAM_THREADS>
__global__ __launch_bounds__(CAM_THREADS, SNK_BA_CAM_WAVES) void cam_pass(Arrays A, Opt O, int nbx, int B)
{
    __shared__ double part[CAM_THREADS / 64][33];
    // nbx > 0 (batched windows): 1-D grid, all cameras of a window on ONE XCD like the other passes of the iteration (workgroup L runs on
    // XCD L % 8).  Every observation gathers 48 bytes of its point (position | V^-1 b_p, 96 KB per benchmark window, written by
    // schur_fused on that XCD); with the cameras of a window dealt over the eight XCDs each L2 fetched the window's points for itself:
    // counter bytes 1.64 MB per window and launch, half of them those gathers (round 5).
    int pb, c;
    if (nbx > 0)
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        pb = (slot / nbx) * 8 + xcd;
        c  = slot - (slot / nbx) * nbx;
        if (pb >= B) return;
    }
    else
    {
        pb = blockIdx.y;
        c  = blockIdx.x;
    }
    const Prob pr = A.prob[pb];
    if (c >= pr.nfc) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int s0 = A.cam_start[pr.camstart_off + c], s1 = A.cam_start[pr.camstart_off + c + 1];
    double acc[33];  // 21 (U upper) | 6 (b_c) | 6 (sum Y b_p)
#pragma unroll
    for (int k = 0; k < 33; ++k) acc[k] = 0.0;
    // J_c, r and Y b_p of an observation are REBUILT here from its static record (streamed in camera order), the pose
    // of this camera and one 48-byte gather per observation (point position of the linearisation | V^-1 b_p) with the
    // same code that point_wave ran -- instead of being written there per observation (224 bytes) and gathered back
    // here in camera order, which made this pass and point_wave bound by HBM / texture-address traffic.
    // The camera is the same for every lane: its rotation matrix and translation are kept in SCALAR registers (v_readfirstlane of the
    // wave-uniform values; 24 of them) -- as vector registers they were 24 of the kernel's 182, which kept it at two wavefronts per SIMD.
    double R[9];
    double pose[7] = {0, 0, 0, 1, 0, 0, 0};
    if (s1 > s0)
    {
        // the camera's image: that of its first observation (the packed records do not carry it)
        const double* pg = A.pose + (size_t)(pr.img_off + A.o_img[pr.obs_off + A.cam_items[pr.citem_off + s0]]) * 7;
#pragma unroll
        for (int k = 0; k < 7; ++k) pose[k] = pg[k];
    }
    quat_to_R(pose, R);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = uniform_f64(R[k]);
#pragma unroll
    for (int k = 4; k < 7; ++k) pose[k] = uniform_f64(pose[k]);
    // a thread runs ~3 observations: the record, the outlier flag and the 48-byte gather of the NEXT one are in flight
    // while the current one is linearised (record -> gather is a dependent pair of memory round trips)
    CamObs ob_n{};
    double pv_n[6] = {0, 0, 0, 0, 0, 0};
    unsigned char out_n = 1;
    auto fetch = [&](int s)
    {
        ob_n  = A.cs_obs[pr.citem_off + s];
        out_n = A.outlier[ob_n.orig];
        const double2* pv = reinterpret_cast<const double2*>(A.ptv + (size_t)(pr.pt_off + (ob_n.ptw & 0x7FFFFFFF)) * 6);
        const double2 a = pv[0], b = pv[1], c = pv[2];
        pv_n[0] = a.x; pv_n[1] = a.y; pv_n[2] = b.x; pv_n[3] = b.y; pv_n[4] = c.x; pv_n[5] = c.y;
    };
    if (s0 + tid < s1) fetch(s0 + tid);
    for (int s = s0 + tid; s < s1; s += CAM_THREADS)
    {
        const CamObs ob = ob_n;
        const bool skip = out_n != 0;
        const double pt[3] = {pv_n[0], pv_n[1], pv_n[2]};
        const double vb[3] = {pv_n[3], pv_n[4], pv_n[5]};
        if (s + CAM_THREADS < s1) fetch(s + CAM_THREADS);
        if (skip) continue;
        double r[3];
        ObsCore oc;
        const int dim = obs_core(pose, R, pt, pr.K, pr.bf, ob.u, ob.v, ob.depth, ob.weight, r, oc);
        if (!dim) continue;
        ObsMoments mo;
        {
            const double sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
            double sw;
            (void)huber_rho(sq, dim == 3 ? O.huber_stereo : O.huber_mono, sw);
            obs_moments(oc, r, sw * sw, mo);
        }
        const double X = oc.X, Y = oc.Y, Z = oc.Z;
        // U = J_c^T J_c = [ M, T ; T^T, Xc x T ], T[i] = Xc x M[:, i] (obs_core); upper triangle row by row, acc[1] (U01) stays 0
        const double T00 = Y * mo.M02, T01 = Z * mo.M00 - X * mo.M02, T02 = -Y * mo.M00;
        const double T10 = Y * mo.M12 - Z * mo.M11, T11 = -X * mo.M12, T12 = X * mo.M11;
        const double T20 = Y * mo.M22 - Z * mo.M12, T21 = Z * mo.M02 - X * mo.M22, T22 = X * mo.M12 - Y * mo.M02;
        acc[0] += mo.M00;
        acc[2] += mo.M02;
        acc[3] += T00;
        acc[4] += T01;
        acc[5] += T02;
        acc[6] += mo.M11;
        acc[7] += mo.M12;
        acc[8] += T10;
        acc[9] += T11;
        acc[10] += T12;
        acc[11] += mo.M22;
        acc[12] += T20;
        acc[13] += T21;
        acc[14] += T22;
        acc[15] += Y * T20 - Z * T10;  // B = Xc x (the columns of T)
        acc[16] += Y * T21 - Z * T11;
        acc[17] += Y * T22 - Z * T12;
        acc[18] += Z * T01 - X * T21;
        acc[19] += Z * T02 - X * T22;
        acc[20] += X * T12 - Y * T02;
        // b_c = -J_c^T r = -[ h ; Xc x h ]
        acc[21] -= mo.h0;
        acc[22] -= mo.h1;
        acc[23] -= mo.h2;
        acc[24] -= Y * mo.h2 - Z * mo.h1;
        acc[25] -= Z * mo.h0 - X * mo.h2;
        acc[26] -= X * mo.h1 - Y * mo.h0;
        if (ob.ptw < 0)  // the point is an unknown
        {
            // Y b_p = W V^-1 b_p = J_c^T (J_p (V^-1 b_p)) = [ f ; Xc x f ], f = M (R v)
            const double e0 = R[0] * vb[0] + R[1] * vb[1] + R[2] * vb[2];
            const double e1 = R[3] * vb[0] + R[4] * vb[1] + R[5] * vb[2];
            const double e2 = R[6] * vb[0] + R[7] * vb[1] + R[8] * vb[2];
            const double f0 = mo.M00 * e0 + mo.M02 * e2, f1 = mo.M11 * e1 + mo.M12 * e2, f2 = mo.M02 * e0 + mo.M12 * e1 + mo.M22 * e2;
            acc[27] += f0;
            acc[28] += f1;
            acc[29] += f2;
            acc[30] += Y * f2 - Z * f1;
            acc[31] += Z * f0 - X * f2;
            acc[32] += X * f1 - Y * f0;
        }
    }
    // fixed-order reduction: inside each wavefront (wave_reduce33), then the wavefronts in order
#if SNK_BA_CAM_BUTTERFLY  // build-time A/B: the xor butterfly of rounds 1-4, every sum through all six steps (~600 instructions)
#pragma unroll
    for (int k = 0; k < 33; ++k)
    {
        double v = acc[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        acc[k] = v;
    }
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 33; ++k) part[wave][k] = acc[k];
#else
    {
        int idx;
        bool ok;
        const double v = wave_reduce33(acc, lane, idx, ok);
        if (ok) part[wave][idx] = v;
    }
#endif
    __syncthreads();
    if (tid == 0)
    {
        double tot[33];
        for (int k = 0; k < 33; ++k)
        {
            tot[k] = part[0][k];
            for (int w = 1; w < CAM_THREADS / 64; ++w) tot[k] += part[w][k];
        }
        cam_finish(A, pr, pb, c, tot);
    }
}

// cam_pass for the batches whose linearisation kernel (schur_fused<3, true>) already left every camera's sums as one partial
// sum per work item: one wavefront per camera adds them in list order (lane = term, four partial sums in flight).
__global__ __launch_bounds__(64) void cam_sum(Arrays A)
{
    __shared__ double s_tot[CS_TERMS];
    const int pb  = blockIdx.y;
    const Prob pr = A.prob[pb];
    const int c   = blockIdx.x, lane = threadIdx.x;
    if (c >= pr.nfc) return;
    const int e0 = A.cc_start[pr.ccam_off + c], e1 = A.cc_start[pr.ccam_off + c + 1];
    const int t  = lane < CS_TERMS ? lane : 0;
    double acc   = 0.0;
    for (int k0 = 0; k0 < e1 - e0; k0 += 64)
    {
        const int chunk = min(64, e1 - e0 - k0);
        const int items = k0 + lane < e1 - e0 ? A.cc_items[e0 + k0 + lane] : 0;
        for (int k = 0; k < chunk; k += 4)
        {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                const int idx = __builtin_amdgcn_readlane(items, min(k + u, chunk - 1));
                v[u]          = A.cam_part[(size_t)idx * CS_TERMS + t];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k + u < chunk) acc += v[u];
        }
    }
    // the partial sums hold b_c (6) | U upper (21) | Y b_p (6); cam_finish wants U | b_c | Y b_p
    if (lane < CS_TERMS) s_tot[lane < 6 ? 21 + lane : (lane < 27 ? lane - 6 : lane)] = acc;
    __syncthreads();
    if (lane == 0)
    {
        double tot[CS_TERMS];
        for (int k = 0; k < CS_TERMS; ++k) tot[k] = s_tot[k];
        cam_finish(A, pr, pb, c, tot);
    }
}

// S block (c1, c2) = [c1 == c2] U - sum over co-observations of Y(c1) W(c2)^T.  One wavefront per
// block: lanes stride over the block's co-observation list, each accumulating a full 6x6 product in
// registers (36 + 36 operands from two contiguous 144-byte rows), then a fixed xor-butterfly sum.
// XCD-aware launch: workgroup L of the 1-D grid runs on XCD L % 8 (observed dispatch order; a speed
// matter only).  All workgroups of a window are given the same L % 8, so the window's Y / W rows (4.6 MB,
// each re-read ~4.5 times by different blocks) are served by ONE XCD's L2 instead of being pulled into all
// eight: windows w = 8 j + xcd live on XCD xcd.
// ---- point-major Schur pass ----
// schur_pass (below) walks one camera pair's co-observation list with lane = entry: every lane gathers its own two
// 144-byte W rows and 48 V^-1 bytes, so each of the 21 dwordx4 loads of a pass touches 64 different cache lines and the
// texture-address unit (about one line per clock) is the limit -- PMC: TA 75 % busy, VALU 22 %.  The rows of ONE point
// are contiguous (observations are sorted by point) and all of its k (k + 1) / 2 camera pairs need them, so here a
// wavefront takes points: the point's run of rows and its V^-1 are copied to LDS with consecutive lanes on consecutive
// 16-byte chunks (2-3 load instructions per point instead of 21 x pairs / 64), lane q < npairs multiplies the pair
// (ra, rb) it owns, and because all points of a work item share one camera set (SetItem) the lane's 36 sums stay in
// registers across the item.  The item's sums go to s_part; schur_sum adds the partial sums of a block in a fixed
// order and writes S.  The next two points' rows are in flight (two register sets) while one is multiplied.
constexpr int SET_CHUNK   = 50;  // points per work item (a set's points are cut into equal items of at most this many)
// Big batches (>= 256 problems, every set with <= 8 free cameras, default kernels): items of up to 128 points.  Every item
// writes one 288-byte partial sum per camera pair whatever its size: with 100 points per camera set (the benchmark window) one
// item per set instead of two halves what schur_fused writes and schur_sum reads back (425 MB per LM iteration of 1024 windows).
// Only schur_fused / update_cost walk such items (two list registers); the alternative paths keep <= 64.
constexpr int SET_CHUNK_BIG = 104;
constexpr int SET_MAX_RUN = 14;  // observations of a point (rows staged per point)
constexpr int SET_MAX_K   = 10;  // free-camera observations of a point: 55 pairs <= 64 lanes
constexpr int SET_SLOT    = SET_MAX_RUN * 144 + 48;  // LDS bytes of one staged point (2064)
constexpr int SET_QUADS   = (SET_SLOT / 16 + 63) / 64;  // load instructions per point (3)

// ROUNDS x 64 lanes >= 6 x pairs, QUADS x 64 lanes >= 16-byte chunks of a point: <4, 2> serves points with <= 8
// free-camera observations out of <= 13 (42 row units... 36 pairs), <6, 3> the limits above; fewer rounds = fewer
// registers = more wavefronts per SIMD, and this kernel lives on latency hiding.
template <int ROUNDS, int QUADS>
__global__ __launch_bounds__(256) void schur_set(Arrays A, int nbx, int B)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_stage[4][2][SET_SLOT];
    int pb, bx;
    if (B >= 16)  // batched windows: one XCD per window
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        pb = (slot / nbx) * 8 + xcd;
        bx = slot - (slot / nbx) * nbx;
    }
    else
    {
        pb = blockIdx.x / nbx;
        bx = blockIdx.x - pb * nbx;
    }
    if (pb >= B) return;
    const Prob pr  = A.prob[pb];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int it   = bx * 4 + wave;
    if (it >= pr.n_set) return;  // whole wavefront
    const SetItem si = A.set_items[pr.set_off + it];
    if (si.npairs == 0) return;  // linearisation-only item of schur_fused
    // the item's point list sits in one register per lane and is broadcast with v_readlane: no dependent index loads
    // in front of the row loads
    const int2 my_pt = lane < si.n_pts ? A.set_pts[si.pts_off + lane] : make_int2(0, 0);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(A.o_W + (size_t)pr.obs_off * 18);
    const unsigned char* Vb = reinterpret_cast<const unsigned char*>(A.Vinv + (size_t)pr.pt_off * 6);
    const int wchunks = si.run * 9, nch = wchunks + 3;  // 16-byte chunks of a point: rows | V^-1
    // Work unit = one ROW of one pair's 6 x 6 block: (W_a V^-1) row r times W_b^T, 27 multiply-adds.  With a pair per lane
    // 36 of 64 lanes work for the usual 8 observations per point; with 6 npairs units over ceil(units / 64) rounds 84 %
    // do.  Unit u = 64 round + lane -> pair u / 6, row u % 6; its 6 sums stay in registers across the item's points.
    static_assert(ROUNDS <= (SET_MAX_K * (SET_MAX_K + 1) / 2 * 6 + 63) / 64 && QUADS <= SET_QUADS, "limits");
    const int nunits = si.npairs * 6, nrounds = (nunits + 63) >> 6;
    int ua[ROUNDS], ub[ROUNDS];  // LDS byte offsets of row r of W_a and of W_b inside a staged point
#pragma unroll
    for (int t = 0; t < ROUNDS; ++t)
    {
        const int u = 64 * t + lane, q = (u * 10923) >> 16, r = u - 6 * q;  // u / 6, u % 6 (u < 384)
        const int pq = u < nunits ? A.set_pairs[si.pair_off + q] : 0;
        ua[t] = (pq & 255) * 144 + r * 24;
        ub[t] = (pq >> 8) * 144;
    }
    double acc[ROUNDS][6];
#pragma unroll
    for (int t = 0; t < ROUNDS; ++t)
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[t][c] = 0.0;

    auto fetch = [&](int n, uint4 (&st)[QUADS])
    {
        const int p  = __builtin_amdgcn_readlane(my_pt.x, n);
        const int s0 = __builtin_amdgcn_readlane(my_pt.y, n);
#pragma unroll
        for (int u = 0; u < QUADS; ++u)
        {
            const int ch = lane + 64 * u;
            st[u]        = uint4{0u, 0u, 0u, 0u};
            if (ch < nch)
                st[u] = *reinterpret_cast<const uint4*>(ch < wchunks ? Wb + (size_t)s0 * 144 + ch * 16 : Vb + (size_t)p * 48 + (ch - wchunks) * 16);
        }
    };
    auto consume = [&](int slot, uint4 (&st)[QUADS], int refill)
    {
        unsigned char* base = s_stage[wave][slot];
#pragma unroll
        for (int u = 0; u < QUADS; ++u)
            if (lane + 64 * u < nch) *reinterpret_cast<uint4*>(base + (lane + 64 * u) * 16) = st[u];
        if (refill < si.n_pts) fetch(refill, st);
        __builtin_amdgcn_wave_barrier();
        const double2* vv = reinterpret_cast<const double2*>(base + wchunks * 16);
        const double2 v01 = vv[0], v23 = vv[1], v45 = vv[2];
        const double v0 = v01.x, v1 = v01.y, v2 = v23.x, v3 = v23.y, v4 = v45.x, v5 = v45.y;
#pragma unroll
        for (int t = 0; t < ROUNDS; ++t)
            if (t < nrounds)  // wave-uniform; lanes past the last unit multiply row 0 of pair 0 and are never stored
            {
                const double* wa   = reinterpret_cast<const double*>(base + ua[t]);
                const double2* rwb = reinterpret_cast<const double2*>(base + ub[t]);
                const double w0 = wa[0], w1 = wa[1], w2 = wa[2];
                const double y0 = w0 * v0 + w1 * v1 + w2 * v2;
                const double y1 = w0 * v1 + w1 * v3 + w2 * v4;
                const double y2 = w0 * v2 + w1 * v4 + w2 * v5;
                double w[18];
#pragma unroll
                for (int q = 0; q < 9; ++q)
                {
                    const double2 b = rwb[q];
                    w[2 * q] = b.x; w[2 * q + 1] = b.y;
                }
#pragma unroll
                for (int c = 0; c < 6; ++c) acc[t][c] += y0 * w[c * 3] + y1 * w[c * 3 + 1] + y2 * w[c * 3 + 2];
            }
        __builtin_amdgcn_wave_barrier();
    };

    uint4 sa[QUADS], sb[QUADS];
    fetch(0, sa);
    if (si.n_pts > 1) fetch(1, sb);
    for (int n = 0; n < si.n_pts; n += 2)
    {
        // a register set is refilled right after its chunks went to LDS (inside consume): no copy, the refill is in
        // flight during the multiplications
        consume(0, sa, n + 2);
        if (n + 1 < si.n_pts) consume(1, sb, n + 3);
    }
#pragma unroll
    for (int t = 0; t < ROUNDS; ++t)
    {
        const int u = 64 * t + lane;
        if (u < nunits)
        {
            double2* out = reinterpret_cast<double2*>(A.s_part + (size_t)si.part_off * 36 + (size_t)u * 6);
            out[0] = make_double2(acc[t][0], acc[t][1]);
            out[1] = make_double2(acc[t][2], acc[t][3]);
            out[2] = make_double2(acc[t][4], acc[t][5]);
        }
    }
}

// ---- point-major Schur pass on the matrix cores ----
// The same work items, staging and partial sums as schur_set, but a point's products are ONE small matrix product on the
// MFMA pipe: with the point's k free rows ordered by camera index, M = (W V^-1) W^T is a 6k x 6k matrix whose upper
// triangle holds exactly the 6 x 6 blocks of the set's pairs.  It is covered by T(T + 1) / 2 tiles of
// v_mfma_f64_16x16x4_f64 (T = ceil(6k / 16): 3 for k <= 8, 4 for k <= 10; inner dimension 3 of 4), accumulated over the
// item's points in the accumulator registers.  Per point a lane reads T x 3 doubles of W from LDS (the vector form: ~70),
// does T x 3 multiply-adds for its element of W V^-1, and the 6 (10) matrix instructions carry the 27 multiply-adds per
// (pair, row) unit of schur_set.  fp64 throughput of the two pipes is the same on MI355X (tools/probes/f64_rate_probe.hip:
// 55.9 vs 49.3 TFLOP/s measured), so the gain is not arithmetic rate: it is the operand traffic -- the vector ALU and the
// LDS no longer carry the products -- and the two pipes running side by side.
// Operand / result layout of the instruction (probed on the hardware): lane l supplies A[l & 15][l >> 4] and
// B[l >> 4][l & 15], and receives D[(l >> 4) + 4 j][l & 15] in register j.
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int T, int QUADS>
__global__ __launch_bounds__(256) void schur_mfma(Arrays A, int nbx, int B)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_stage[4][2][SET_SLOT];
    int pb, bx;
    if (B >= 16)  // batched windows: one XCD per window
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        pb = (slot / nbx) * 8 + xcd;
        bx = slot - (slot / nbx) * nbx;
    }
    else
    {
        pb = blockIdx.x / nbx;
        bx = blockIdx.x - pb * nbx;
    }
    if (pb >= B) return;
    const Prob pr  = A.prob[pb];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int it   = bx * 4 + wave;
    if (it >= pr.n_set) return;  // whole wavefront
    const SetItem si = A.set_items[pr.set_off + it];
    if (si.npairs == 0) return;  // linearisation-only item of schur_fused
    const int2 my_pt = lane < si.n_pts ? A.set_pts[si.pts_off + lane] : make_int2(0, 0);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(A.o_W + (size_t)pr.obs_off * 18);
    const unsigned char* Vb = reinterpret_cast<const unsigned char*>(A.Vinv + (size_t)pr.pt_off * 6);
    const int wchunks = si.run * 9, nch = wchunks + 3;  // 16-byte chunks of a point: rows | V^-1
    const int nrows = si.nfree * 6;                      // rows of M
    const int kk = lane >> 4, mr = lane & 15;            // inner index / row (column) inside a tile of this lane's operands
    // LDS byte offset of row 16 t + mr of M inside a staged point (rows beyond 6k and the 4th inner index supply zeros)
    int roff[T];
    bool rok[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
    {
        const int m = 16 * t + mr, i = (m * 43) >> 8, r = m - 6 * i;  // m / 6, m % 6 (m < 64)
        rok[t]      = m < nrows && kk < 3;
        const int pos = m < nrows ? A.set_pairs[si.aux_off + i] : 0;
        roff[t]     = pos * 144 + r * 24;
    }
    double4_t acc[T * (T + 1) / 2];
#pragma unroll
    for (int q = 0; q < T * (T + 1) / 2; ++q) acc[q] = double4_t{0.0, 0.0, 0.0, 0.0};

    auto fetch = [&](int n, uint4 (&st)[QUADS])
    {
        const int p  = __builtin_amdgcn_readlane(my_pt.x, n);
        const int s0 = __builtin_amdgcn_readlane(my_pt.y, n);
#pragma unroll
        for (int u = 0; u < QUADS; ++u)
        {
            const int ch = lane + 64 * u;
            st[u]        = uint4{0u, 0u, 0u, 0u};
            if (ch < nch)
                st[u] = *reinterpret_cast<const uint4*>(ch < wchunks ? Wb + (size_t)s0 * 144 + ch * 16 : Vb + (size_t)p * 48 + (ch - wchunks) * 16);
        }
    };
    auto consume = [&](int slot, uint4 (&st)[QUADS], int refill)
    {
        unsigned char* base = s_stage[wave][slot];
#pragma unroll
        for (int u = 0; u < QUADS; ++u)
            if (lane + 64 * u < nch) *reinterpret_cast<uint4*>(base + (lane + 64 * u) * 16) = st[u];
        if (refill < si.n_pts) fetch(refill, st);
        __builtin_amdgcn_wave_barrier();
        const double* vv = reinterpret_cast<const double*>(base + wchunks * 16);
        // column kk of the symmetric V^-1 (entries 0 1 2 / 1 3 4 / 2 4 5)
        const double vc0 = vv[kk == 0 ? 0 : (kk == 1 ? 1 : 2)], vc1 = vv[kk == 0 ? 1 : (kk == 1 ? 3 : 4)], vc2 = vv[kk == 0 ? 2 : (kk == 1 ? 4 : 5)];
        double ya[T], wb[T];
#pragma unroll
        for (int t = 0; t < T; ++t)
        {
            const double* w = reinterpret_cast<const double*>(base + roff[t]);
            const double w0 = w[0], w1 = w[1], w2 = w[2];
            ya[t] = rok[t] ? w0 * vc0 + w1 * vc1 + w2 * vc2 : 0.0;                  // (W V^-1)[row][kk]
            wb[t] = rok[t] ? (kk == 0 ? w0 : (kk == 1 ? w1 : w2)) : 0.0;           // W[row][kk]
        }
        int q = 0;
#pragma unroll
        for (int ti = 0; ti < T; ++ti)
#pragma unroll
            for (int tj = ti; tj < T; ++tj, ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[ti], wb[tj], acc[q], 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
    };

    uint4 sa[QUADS], sb[QUADS];
    fetch(0, sa);
    if (si.n_pts > 1) fetch(1, sb);
    for (int n = 0; n < si.n_pts; n += 2)
    {
        consume(0, sa, n + 2);
        if (n + 1 < si.n_pts) consume(1, sb, n + 3);
    }
    // the tiles -> the item's partial sums (pair slot q: 36 doubles, row-major block (W_ra V^-1) W_rb^T)
    const int* tab = A.set_pairs + si.aux_off + si.nfree;
    double* part   = A.s_part + (size_t)si.part_off * 36;
    int q = 0;
#pragma unroll
    for (int ti = 0; ti < T; ++ti)
#pragma unroll
        for (int tj = ti; tj < T; ++tj, ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                const int m = 16 * ti + kk + 4 * j, n = 16 * tj + mr;
                if (m >= nrows || n >= nrows) continue;
                const int ia = (m * 43) >> 8, r = m - 6 * ia, ib = (n * 43) >> 8, c = n - 6 * ib;
                if (ia > ib) continue;  // lower triangle inside a diagonal tile: not a pair
                const int slot = tab[ia * si.nfree + ib];
                const double v = acc[q][j];
                part[(size_t)slot * 36 + r * 6 + c] = v;
                // a diagonal block that straddles two tiles has its lower-left corner in a tile that is not computed:
                // M is symmetric, the mirror element stands in (schur_sum symmetrises diagonal blocks anyway)
                if (ia == ib && ti != tj) part[(size_t)slot * 36 + c * 6 + r] = v;
            }
}

// ---- linearisation + point-major Schur pass in one kernel: W never reaches HBM ----
// point_wave writes W = J_c^T J_p (144 bytes per observation, 590 MB per LM iteration of 256 benchmark windows) only for
// the Schur pass to read it back.  Here a wavefront that owns a work item of schur_mfma linearises the item's points itself,
// 64 / run points at a time (lane = (point of the group, observation of the point); the same arithmetic, in the same order,
// as point_wave: phase 1 per observation, phase 2 per point in observation order), leaves the rows in LDS and multiplies
// them on the matrix cores right there.  It also writes what point_wave writes per POINT (cost, V^-1, b_p, position | V^-1 b_p),
// so point_wave is not launched at all; points without Schur products (constant points, points of constant cameras only)
// come as work items without pairs.  Observation inputs are read once (48 bytes + cached point / pose gathers).
constexpr int SF_GMAX = 16;  // points linearised at a time (64 / run, at most this many: bounds the per-point LDS arrays)
constexpr int SF_NC   = 10;  // per-observation contributions summed per point: V (6), b_p (3), cost
typedef unsigned int u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef unsigned int u32x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
struct SfObs  // first round of loads of a lane: its static observation record and its point (both known without a lookup)
{
    SetRec rec;
    double pt[3];
    bool act;
};
struct SfGather  // second round: through the record's indices (the camera of a lane is loop-invariant, see schur_fused)
{
    bool is_out;
};
// CS (camera sums, T == 3 only): the wavefront also emits what cam_pass computes -- per free camera b_c = -sum J_c^T r,
// U = sum J_c^T J_c, sum Y b_p -- as ONE partial sum per (work item, free camera of its set): all points of an item see the
// same cameras in the same run positions, so lane (point g, position a) of a group contributes to camera(a).  The observation
// is linearised here anyway; cam_pass re-read every record, gathered the point again and linearised a second time (294 us per
// 1024 windows).  33 terms per observation go through the contribution buffer eight at a time (lane = observation writes,
// lane = (free camera f, term k) sums the group's points in order and keeps a running sum per pass: five passes, five
// accumulators); cam_sum adds a camera's partial sums in list order and finishes like cam_pass (damping, U, rhs).
__device__ __forceinline__ double cam_term(int t, const double* Jc, const double* r)  // t < 27: b_c (6) | U upper (21)
{
    constexpr int PA[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
    constexpr int PB[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};
    if (t < 6) return -(Jc[t] * r[0] + Jc[6 + t] * r[1] + Jc[12 + t] * r[2]);
    const int a = PA[t - 6], b = PB[t - 6];
    return Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b] + Jc[12 + a] * Jc[12 + b];
}

template <int T, bool CS>
__global__ __launch_bounds__(256) void schur_fused(Arrays A, Opt O, int nbx, int B)
{
    static_assert(!CS || T == 3, "camera sums: at most 8 free cameras per set (lane = (camera, term of eight))");
    // per wavefront: the W rows of the group (64 x 18 doubles) | the observations' contributions (64 x 10) | the sums of
    // the group's points | their V^-1 | V^-1 b_p.  16.8 KB: two workgroups per CU, which is also what the registers allow.
    __shared__ __attribute__((aligned(16))) double s_buf[4][64 * 18 + 64 * SF_NC + SF_GMAX * (SF_NC + 9) + 4 + SF_GMAX * 3 + SF_GMAX * 4];
    int pb, bx;
    if (B >= 16)  // batched windows: one XCD per window
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        pb = (slot / nbx) * 8 + xcd;
        bx = slot - (slot / nbx) * nbx;
    }
    else
    {
        pb = blockIdx.x / nbx;
        bx = blockIdx.x - pb * nbx;
    }
    if (pb >= B) return;
    const Prob pr  = A.prob[pb];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int it   = bx * 4 + wave;
    if (it >= pr.n_set) return;  // whole wavefront
    const SetItem si = A.set_items[pr.set_off + it];
    const int2 my_pt = lane < si.n_pts ? A.set_pts[si.pts_off + lane] : make_int2(0, 0);
    // items of up to 128 points (big batches, SET_CHUNK_BIG): the second half of the list in a second register
    const int my_pt_hi = 64 + lane < si.n_pts ? A.set_pts[si.pts_off + 64 + lane].x : 0;
    auto pt_at = [&](int idx) -> int  // point idx of the item; both shuffles outside any branch (a shuffle reads 0 from inactive lanes)
    {
        const int lo = __shfl(my_pt.x, min(idx, 63)), hi = __shfl(my_pt_hi, min(max(idx - 64, 0), 63));
        return idx < 64 ? lo : hi;
    };
    double* s_w    = s_buf[wave];
    double* s_con  = s_w + 64 * 18;
    double* s_sum  = s_con + 64 * SF_NC;
    double* s_vi   = s_sum + SF_GMAX * SF_NC;
    double* s_zero = s_vi + SF_GMAX * 9;  // 4 zeros: what the operand lanes outside the matrix read
    double* s_vb   = s_zero + 4;          // V^-1 b_p of the group's points (camera sums)
    double* s_pp   = s_vb + SF_GMAX * 3;  // position | "is an unknown" of the group's points, left by the first observation lane of each point
    const int run  = si.run, G = min(64 / run, SF_GMAX);  // run <= SET_MAX_RUN = 14: at least 4 points per group
    const int lg   = lane / run;                           // point of the group (its observation is lane - lg * run)
    const double* poses = A.pose + (size_t)pr.img_off * 7;
    const double lambda = A.state[pb].lambda;
    if (lane < 4) s_zero[lane] = 0.0;

    const int nrows = si.nfree * 6;
    const int kk = lane >> 4, mr = lane & 15;
    // Operand addressing of the matrix products without selects: lane (row 16 t + mr, inner index kk) reads its row of W at
    // wrow[t] + point * wstep[t]; lanes outside the matrix (row >= 6k) have their pointer parked on four zeros with a zero step,
    // so they read zeros and need no masking.
    // The inner dimension of a product is FOUR wide and a point has three coordinates: four points share three products (round 5;
    // one point per product left a quarter of every v_mfma_f64_16x16x4 multiplying zeros).  Inner index 4 s + kk of product s of a
    // block of four points is (point gl[s], coordinate cs[s]) = ((4 s + kk) / 3, (4 s + kk) % 3); V^-1 is kept as the full 3 x 3
    // matrix per point, so that the lane's column is at 3 * (4 s + kk) doubles from the block's start -- a per-lane base and
    // compile-time offsets.
    const double* wrow0[T];
    int wstep[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
    {
        const int m = 16 * t + mr, i = (m * 43) >> 8, r = m - 6 * i;
        const bool ok = m < nrows;
        const int pos = ok ? A.set_pairs[si.aux_off + i] : 0;
        wrow0[t]    = ok ? s_w + pos * 18 + r * 3 : s_zero;
        wstep[t]    = ok ? run * 18 : 0;
    }
    const int gl[3] = {kk == 3 ? 1 : 0, kk >= 2 ? 2 : 1, kk >= 1 ? 3 : 2};
    const int cs[3] = {kk == 3 ? 0 : kk, kk == 2 ? 0 : (kk == 3 ? 1 : kk + 1), kk == 0 ? 2 : kk - 1};
    double4_t acc[T * (T + 1) / 2];
#pragma unroll
    for (int q = 0; q < T * (T + 1) / 2; ++q) acc[q] = double4_t{0.0, 0.0, 0.0, 0.0};
    // camera sums: lane = (free camera f of the set, term k of the pass); cs_row = f's run position
    const int cs_f = lane >> 3, cs_k = lane & 7;
    const bool cs_on = CS && cs_f < si.nfree;
    const int cs_row = cs_on ? A.set_pairs[si.aux_off + cs_f] : 0;
    double cs_acc[CS ? CS_PASSES : 1];
#pragma unroll
    for (int q = 0; q < (CS ? CS_PASSES : 1); ++q) cs_acc[q] = 0.0;
    // one pass: every lane's eight terms through the contribution buffer, summed over the group's points in order
    auto cs_pass = [&](int pass, const double* term, int gc)
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) s_con[lane * 8 + k] = term[k];
        __builtin_amdgcn_wave_barrier();
        if (cs_on)
        {
            const double* q = s_con + cs_row * 8 + cs_k;
            double sum = 0.0;
            for (int g = 0; g < gc; ++g) sum += q[g * run * 8];  // (all reads first, then the adds: measured slower, 1437 -> 1460 / 1504 us)
            cs_acc[pass] += sum;
        }
        __builtin_amdgcn_wave_barrier();
    };

    // The two dependent rounds of loads of a group are issued one group ahead: round 1 (indexed by the observation) at the
    // top of the previous group's work, round 2 (point, pose, flags: through round 1's indices) in front of its matrix
    // products -- only two wavefronts fit a SIMD here, so the latency has to be hidden inside the wavefront.
    const char* rec_base = reinterpret_cast<const char*>(A.set_obs + si.rec_off);
    const unsigned n_rec = (unsigned)(si.n_pts * run);
    auto load1 = [&](int n0, SfObs& o)
    {
        const int gc = min(G, si.n_pts - n0);
        o.act        = lg < gc;
        // 32-bit offsets from uniform bases: one address instruction per load
        const unsigned ri = min((unsigned)(n0 * run + lane), n_rec - 1u);
        const char* rp = rec_base + ri * 40u;  // 40-byte records: 8-byte aligned
        const u32x4_a8 r0 = *reinterpret_cast<const u32x4_a8*>(rp), r1 = *reinterpret_cast<const u32x4_a8*>(rp + 16);
        const u32x2_a8 r2 = *reinterpret_cast<const u32x2_a8*>(rp + 32);
        const int px   = pt_at(n0 + (o.act ? lg : 0));
        const double* ptp = reinterpret_cast<const double*>(reinterpret_cast<const char*>(A.pt) + (unsigned)(pr.pt_off + px) * 24u);
        o.pt[0] = ptp[0]; o.pt[1] = ptp[1]; o.pt[2] = ptp[2];
        o.rec.u      = __hiloint2double((int)r0.y, (int)r0.x);
        o.rec.v      = __hiloint2double((int)r0.w, (int)r0.z);
        o.rec.depth  = __hiloint2double((int)r1.y, (int)r1.x);
        o.rec.weight = __hiloint2double((int)r1.w, (int)r1.z);
        o.rec.orig = (int)r2.x; o.rec.pk = (int)r2.y;
    };
    // Second round (through the record's indices): the outlier flag of every observation -- and the camera.  All points of a work item
    // see the same FREE cameras in the same run positions and lane -> (point of the group, position) is the same in every group, so a
    // lane's camera is loop-invariant unless its position holds a constant camera that differs from point to point: its rotation matrix
    // and translation are formed ONCE in front of the loop (camR) and again only by the lanes whose image changes (a dependent round
    // trip then, at the end of a group) -- instead of seven gathered loads and a quaternion -> matrix conversion per observation and group.
    double camR[9], camT[7] = {0, 0, 0, 1, 0, 0, 0};  // camT[4..6] = the translation (obs_core reads pose[4..6])
    auto load_cam = [&](int img)
    {
        const double* posep = reinterpret_cast<const double*>(reinterpret_cast<const char*>(poses) + (unsigned)img * 56u);
        double q[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) q[k] = posep[k];
        quat_to_R(q, camR);
        camT[4] = q[4]; camT[5] = q[5]; camT[6] = q[6];
    };
    auto load2 = [&](const SfObs& o, SfGather& g) { g.is_out = A.outlier[(unsigned)o.rec.orig] != 0; };
    // one group: `ob` / `gt` are its inputs (already loaded), `obn` / `gtn` receive the next group's
    auto process = [&](int n0, const SfObs& ob, const SfGather& gt, SfObs& obn, SfGather& gtn)
    {
        const int gc    = min(G, si.n_pts - n0);  // points of this group (wave-uniform)
        const bool more = n0 + G < si.n_pts;      // wave-uniform
        if (more) load1(n0 + G, obn);
        // ---- phase 1: lane = observation: residual, Jacobians, its terms of V / b_p / cost, its row W = Jc^T Jp ----
        {
            double con[SF_NC];
#pragma unroll
            for (int k = 0; k < SF_NC; ++k) con[k] = 0.0;
            bool cpl = false;  // the observation couples a free camera with a free point: it has a row of W
            double r[3] = {0.0, 0.0, 0.0};
            bool lin = false;  // the observation is active in this iteration (what cam_pass sums)
            ObsCore oc{};      // kept for the camera-sum variant, which rebuilds J_c from it (cam_term)
            double sw_obs = 0.0;
            double N[9];       // N = M R; W = [N ; Xc x N] (see obs_core)
            double Xc[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 9; ++k) N[k] = 0.0;
            if (ob.act && !gt.is_out && !(SNK_SF_SKIP & 4))
            {
                const double* R = camR;
                const int dim = obs_core(camT, R, ob.pt, pr.K, pr.bf, ob.rec.u, ob.rec.v, ob.rec.depth, ob.rec.weight, r, oc);
                lin           = dim != 0;
                if (dim)
                {
                    const double sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
                    double sw;
                    con[9] = huber_rho(sq, dim == 3 ? O.huber_stereo : O.huber_mono, sw);
                    ObsMoments mo;
                    obs_moments(oc, r, sw * sw, mo);
                    sw_obs = sw;
                    if (ob.rec.ptfree())
                    {
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                        {
                            N[j]     = mo.M00 * R[j] + mo.M02 * R[6 + j];
                            N[3 + j] = mo.M11 * R[3 + j] + mo.M12 * R[6 + j];
                            N[6 + j] = mo.M02 * R[j] + mo.M12 * R[3 + j] + mo.M22 * R[6 + j];
                        }
                        // V = J_p^T J_p = R^T N (upper triangle), b_p = -R^T h
                        con[0] = R[0] * N[0] + R[3] * N[3] + R[6] * N[6];
                        con[1] = R[0] * N[1] + R[3] * N[4] + R[6] * N[7];
                        con[2] = R[0] * N[2] + R[3] * N[5] + R[6] * N[8];
                        con[3] = R[1] * N[1] + R[4] * N[4] + R[7] * N[7];
                        con[4] = R[1] * N[2] + R[4] * N[5] + R[7] * N[8];
                        con[5] = R[2] * N[2] + R[5] * N[5] + R[8] * N[8];
#pragma unroll
                        for (int b = 0; b < 3; ++b) con[6 + b] = -(R[b] * mo.h0 + R[3 + b] * mo.h1 + R[6 + b] * mo.h2);
                        cpl   = ob.rec.cam() >= 0;
                        Xc[0] = oc.X; Xc[1] = oc.Y; Xc[2] = oc.Z;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < SF_NC; ++k) s_con[lane * SF_NC + k] = con[k];
            // phase 2b wants the point's position and constancy per POINT: the point's first observation lane has both in registers
            // (round 5: they used to be two dependent global loads in front of phase 2b, one memory round trip per group)
            if (ob.act && lane == lg * run)
            {
                s_pp[lg * 4 + 0] = ob.pt[0];
                s_pp[lg * 4 + 1] = ob.pt[1];
                s_pp[lg * 4 + 2] = ob.pt[2];
                s_pp[lg * 4 + 3] = ob.rec.ptfree() ? 1.0 : 0.0;
            }
            if (si.nfree != 0)
            {
                // the observation's row of W = J_c^T J_p: rows 0-2 = N, rows 3-5 = Xc x (the columns of N); zero unless it couples
#pragma unroll
                for (int b = 0; b < 3; ++b)
                {
                    const double n0 = cpl ? N[b] : 0.0, n1 = cpl ? N[3 + b] : 0.0, n2 = cpl ? N[6 + b] : 0.0;
                    s_w[lane * 18 + 0 * 3 + b] = n0;
                    s_w[lane * 18 + 1 * 3 + b] = n1;
                    s_w[lane * 18 + 2 * 3 + b] = n2;
                    s_w[lane * 18 + 3 * 3 + b] = Xc[1] * n2 - Xc[2] * n1;
                    s_w[lane * 18 + 4 * 3 + b] = Xc[2] * n0 - Xc[0] * n2;
                    s_w[lane * 18 + 5 * 3 + b] = Xc[0] * n1 - Xc[1] * n0;
                }
            }
            __builtin_amdgcn_wave_barrier();

            // ---- phase 2a: lane = (point of the group, term): the point's sums in observation order ----
            // (a lane's items lane, lane + 64, lane + 128 of the gc * 10 <= 160 are summed side by side: the sums are chains of dependent
            // LDS reads and additions, and the second / third pass over 16 lanes used to cost as much as the first over 64)
            if (!(SNK_SF_SKIP & 8))
            {
                const int n_items = gc * SF_NC;
                const double* q[3];
                bool has[3];
#pragma unroll
                for (int u = 0; u < 3; ++u)
                {
                    const int idx = min(lane + 64 * u, n_items - 1);
                    const int g = (idx * 205) >> 11, comp = idx - g * SF_NC;  // idx / 10 (idx < 160)
                    q[u]   = s_con + g * run * SF_NC + comp;
                    has[u] = lane + 64 * u < n_items;
                }
                double sum[3] = {0.0, 0.0, 0.0};
                if (n_items > 128)  // wave-uniform (more than 12 points per group: runs of <= 5)
                    for (int a = 0; a < run; ++a)
                    {
#pragma unroll
                        for (int u = 0; u < 3; ++u) sum[u] += q[u][a * SF_NC];
                    }
                else
                    for (int a = 0; a < run; ++a)
                    {
#pragma unroll
                        for (int u = 0; u < 2; ++u) sum[u] += q[u][a * SF_NC];
                    }
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (has[u]) s_sum[lane + 64 * u] = sum[u];
            }
            __builtin_amdgcn_wave_barrier();
            // ---- camera sums, the 27 terms of J_c and r (the contribution buffer is free again; J_c dies here) ----
            if constexpr (CS)
            {
                if (si.nfree != 0)
                {
                    double Jc[18];
                    core_to_Jc(oc, sw_obs, Jc);
#pragma unroll
                    for (int k = 0; k < 3; ++k) r[k] *= sw_obs;
#pragma unroll
                    for (int pass = 0; pass < 4; ++pass)
                    {
                        double term[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) term[k] = pass * 8 + k < 27 && lin ? cam_term(pass * 8 + k, Jc, r) : 0.0;
                        cs_pass(pass, term, gc);
                    }
                }
            }
        }
        if (more) load2(obn, gtn);
        // ---- phase 2b: lane = point of the group: damping, V^-1, the per-point outputs of point_wave ----
        const int p2 = pt_at(min(n0 + lane, si.n_pts - 1));  // outside the branch
        if (lane < gc && !(SNK_SF_SKIP & 2))
        {
            const int gp2 = pr.pt_off + p2;
            const bool pfree = s_pp[lane * 4 + 3] != 0.0;
            const double* sm = s_sum + lane * SF_NC;
            double V[6] = {sm[0], sm[1], sm[2], sm[3], sm[4], sm[5]};
            const double bp[3] = {sm[6], sm[7], sm[8]};
            A.cost_pt[gp2] = sm[9];
            double Vi[6] = {0, 0, 0, 0, 0, 0};
            double vb[3] = {0, 0, 0};
            if (pfree)
            {
                V[0] += lambda * clampd(V[0]);
                V[3] += lambda * clampd(V[3]);
                V[5] += lambda * clampd(V[5]);
                const double a = V[0], b = V[1], cc = V[2], d = V[3], e = V[4], f = V[5];
                const double Aa = d * f - e * e, Bb = cc * e - b * f, Cc = b * e - cc * d;
                const double det = a * Aa + b * Bb + cc * Cc;
                const double id  = det == 0.0 ? 0.0 : 1.0 / det;
                Vi[0] = Aa * id;
                Vi[1] = Bb * id;
                Vi[2] = Cc * id;
                Vi[3] = (a * f - cc * cc) * id;
                Vi[4] = (b * cc - a * e) * id;
                Vi[5] = (a * d - b * b) * id;
#pragma unroll
                for (int k = 0; k < 6; ++k) A.Vinv[(size_t)gp2 * 6 + k] = Vi[k];
                A.bp[(size_t)gp2 * 3 + 0] = bp[0];
                A.bp[(size_t)gp2 * 3 + 1] = bp[1];
                A.bp[(size_t)gp2 * 3 + 2] = bp[2];
                vb[0] = Vi[0] * bp[0] + Vi[1] * bp[1] + Vi[2] * bp[2];
                vb[1] = Vi[1] * bp[0] + Vi[3] * bp[1] + Vi[4] * bp[2];
                vb[2] = Vi[2] * bp[0] + Vi[4] * bp[1] + Vi[5] * bp[2];
            }
            {
                double* vo = s_vi + lane * 9;  // the full symmetric matrix: column c = row c = three consecutive doubles
                vo[0] = Vi[0]; vo[1] = Vi[1]; vo[2] = Vi[2];
                vo[3] = Vi[1]; vo[4] = Vi[3]; vo[5] = Vi[4];
                vo[6] = Vi[2]; vo[7] = Vi[4]; vo[8] = Vi[5];
            }
            if (CS)
            {
                s_vb[lane * 3 + 0] = vb[0];
                s_vb[lane * 3 + 1] = vb[1];
                s_vb[lane * 3 + 2] = vb[2];
            }
            {
                const double px = s_pp[lane * 4], py = s_pp[lane * 4 + 1], pz = s_pp[lane * 4 + 2];
                double* pv        = A.ptv + (size_t)gp2 * 6;
                pv[0] = px; pv[1] = py; pv[2] = pz;
                pv[3] = vb[0]; pv[4] = vb[1]; pv[5] = vb[2];
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (CS && si.nfree != 0)
        {
            // ---- camera sums, Y b_p = W (V^-1 b_p): the lane's row of W (zero unless the observation couples) times its point's vector ----
            const int gl = min(lg, gc - 1);
            const double b0 = s_vb[gl * 3], b1 = s_vb[gl * 3 + 1], b2 = s_vb[gl * 3 + 2];
            double term[8];
#pragma unroll
            for (int a = 0; a < 6; ++a) term[a] = lg < gc ? s_w[lane * 18 + a * 3] * b0 + s_w[lane * 18 + a * 3 + 1] * b1 + s_w[lane * 18 + a * 3 + 2] * b2 : 0.0;
            term[6] = term[7] = 0.0;
            cs_pass(4, term, gc);
        }

        // ---- phase 3: (W V^-1) W^T of every point of the group on the matrix cores ----
        if (si.nfree != 0 && !(SNK_SF_SKIP & 1))  // wave-uniform; a work item without pairs is linearisation only
        {
            const double* wr[T];
#pragma unroll
            for (int t = 0; t < T; ++t) wr[t] = wrow0[t];
            const double* vq = s_vi + 3 * kk;
            int g = 0;
            for (; g + 4 <= gc; g += 4, vq += 36)  // four points, three products per tile pair
            {
#pragma unroll
                for (int s = 0; s < 3; ++s)
                {
                    const double vc0 = vq[12 * s], vc1 = vq[12 * s + 1], vc2 = vq[12 * s + 2];  // column cs[s] of the symmetric V^-1 of point gl[s]
                    double ya[T], wb[T];
#pragma unroll
                    for (int t = 0; t < T; ++t)
                    {
                        const double* w = wr[t] + gl[s] * wstep[t];
                        ya[t] = w[0] * vc0 + w[1] * vc1 + w[2] * vc2;  // (W V^-1)[row][coordinate]
                        wb[t] = w[cs[s]];                               // W[row][coordinate]
                    }
                    int q = 0;
#pragma unroll
                    for (int ti = 0; ti < T; ++ti)
#pragma unroll
                        for (int tj = ti; tj < T; ++tj, ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[ti], wb[tj], acc[q], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < T; ++t) wr[t] += 4 * wstep[t];
            }
            // what is left of a group whose points are no multiple of four: one point per product, the fourth inner index reads zeros
            const bool on   = kk < 3;
            const int ksel  = on ? kk : 0;
            for (; g < gc; ++g, vq += 9)
            {
                const double* vt = on ? vq : s_zero;
                const double vc0 = vt[0], vc1 = vt[1], vc2 = vt[2];
                double ya[T], wb[T];
#pragma unroll
                for (int t = 0; t < T; ++t)
                {
                    const double* w = on ? wr[t] : s_zero;
                    ya[t] = w[0] * vc0 + w[1] * vc1 + w[2] * vc2;
                    wb[t] = w[ksel];
                    wr[t] += wstep[t];
                }
                int q = 0;
#pragma unroll
                for (int ti = 0; ti < T; ++ti)
#pragma unroll
                    for (int tj = ti; tj < T; ++tj, ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[ti], wb[tj], acc[q], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // a lane whose position holds another image in the next group (constant cameras only) forms that camera's matrix now
        if (more && obn.act && obn.rec.img() != ob.rec.img()) load_cam(obn.rec.img());  // (lanes active in a group were active in every group before it)
    };
    // two register sets, used alternately: no copies between the groups
    SfObs oa, obb;
    SfGather ga, gb;
    load1(0, oa);
    load_cam(oa.rec.img());
    load2(oa, ga);
    for (int n0 = 0; n0 < si.n_pts; n0 += 2 * G)
    {
        process(n0, oa, ga, obb, gb);
        if (n0 + G < si.n_pts) process(n0 + G, obb, gb, oa, ga);
    }
    if (si.nfree == 0) return;
    if (cs_on)
    {
        double* cp = A.cam_part + (size_t)(si.cpart_off + cs_f) * CS_TERMS;
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) cp[pass * 8 + cs_k] = cs_acc[pass];
        if (cs_k < 3) cp[24 + cs_k] = cs_acc[3];
        if (cs_k < 6) cp[27 + cs_k] = cs_acc[4];
    }
    // The slot of every camera pair of the set, staged in LDS first (the contribution buffer is free): looked up in global memory
    // element by element, each of the up to 24 + 8 stores below waited for its own dependent load -- and, the counter being shared, for
    // the store before it -- a chain of ~28 memory round trips at the end of every wavefront (round 5: a quarter of a wavefront's life).
    const int* tab = A.set_pairs + si.aux_off + si.nfree;
    int* s_tab     = reinterpret_cast<int*>(s_con);
    {
        const int n2 = si.nfree * si.nfree;  // nfree <= 10 (SET_MAX_K): at most two entries per lane
        if (lane < n2) s_tab[lane] = tab[lane];
        if (lane + 64 < n2) s_tab[lane + 64] = tab[lane + 64];
    }
    __builtin_amdgcn_wave_barrier();
    double* part   = A.s_part + (size_t)si.part_off * 36;
    int q = 0;
#pragma unroll
    for (int ti = 0; ti < T; ++ti)
#pragma unroll
        for (int tj = ti; tj < T; ++tj, ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                const int m = 16 * ti + kk + 4 * j, n = 16 * tj + mr;
                if (m >= nrows || n >= nrows) continue;
                const int ia = (m * 43) >> 8, r = m - 6 * ia, ib = (n * 43) >> 8, c = n - 6 * ib;
                if (ia > ib) continue;
                const int slot = s_tab[ia * si.nfree + ib];
                const double v = acc[q][j];
                part[(size_t)slot * 36 + r * 6 + c] = v;
                if (ia == ib && ti != tj) part[(size_t)slot * 36 + c * 6 + r] = v;
            }
}

// ---- back-substitution of the points + robust cost at the trial state, one kernel over the work items of schur_fused ----
// update_wave and cost_wave are short-lived wavefronts behind a chain of five dependent loads (work item -> point range ->
// observation indices -> point / pose gathers -> per-point data): 81 % of their wavefront cycles wait (PMC r02f).  Here a
// wavefront walks a whole set item (<= 64 points) group by group with the next group's records in flight, reads the static
// observation records in the order it walks them (coalesced), and the trial cost is evaluated right after the group's new
// points exist (in LDS) -- the observation, point and pose loads are shared by the two passes.  Needs pose_new:
// update_pass(images) runs BEFORE this kernel.  Same arithmetic and summation order as update_wave / cost_wave.
__global__ __launch_bounds__(256, SNK_BA_UC_WAVES) void update_cost(Arrays A, Opt O, int nbx, int B)
{
    // per wavefront: t = J_p^T (J_c dc) per observation (64 x 3) | cost per observation (64) | b_p - sum t per point (16 x 3) | new points (16 x 3)
    __shared__ double s_buf[4][64 * 3 + 64 + SF_GMAX * 6];
    int pb, bx;
    if (B >= 16)  // batched windows: one XCD per window
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        pb = (slot / nbx) * 8 + xcd;
        bx = slot - (slot / nbx) * nbx;
    }
    else
    {
        pb = blockIdx.x / nbx;
        bx = blockIdx.x - pb * nbx;
    }
    if (pb >= B) return;
    const Prob pr  = A.prob[pb];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int it   = bx * 4 + wave;
    if (it >= pr.n_set) return;  // whole wavefront
    const SetItem si = A.set_items[pr.set_off + it];
    const int2 my_pt = lane < si.n_pts ? A.set_pts[si.pts_off + lane] : make_int2(0, 0);
    // items of up to 128 points (big batches, SET_CHUNK_BIG): the second half of the list in a second register
    const int my_pt_hi = 64 + lane < si.n_pts ? A.set_pts[si.pts_off + 64 + lane].x : 0;
    auto pt_at = [&](int idx) -> int  // point idx of the item; both shuffles outside any branch (a shuffle reads 0 from inactive lanes)
    {
        const int lo = __shfl(my_pt.x, min(idx, 63)), hi = __shfl(my_pt_hi, min(max(idx - 64, 0), 63));
        return idx < 64 ? lo : hi;
    };
    double* s_t  = s_buf[wave];
    double* s_c  = s_t + 64 * 3;
    double* s_g  = s_c + 64;
    double* s_pn = s_g + SF_GMAX * 3;
    const int run = si.run, G = min(64 / run, SF_GMAX);
    const int lg  = lane / run;
    const char* rec_base = reinterpret_cast<const char*>(A.set_obs + si.rec_off);
    const unsigned n_rec = (unsigned)(si.n_pts * run);
    const double* x = A.x + pr.vec_off;

    auto load1 = [&](int n0, SfObs& o)
    {
        const int gc = min(G, si.n_pts - n0);
        o.act        = lg < gc;
        const unsigned ri = min((unsigned)(n0 * run + lane), n_rec - 1u);
        const char* rp = rec_base + ri * 40u;  // 40-byte records: 8-byte aligned
        const u32x4_a8 r0 = *reinterpret_cast<const u32x4_a8*>(rp), r1 = *reinterpret_cast<const u32x4_a8*>(rp + 16);
        const u32x2_a8 r2 = *reinterpret_cast<const u32x2_a8*>(rp + 32);
        const int px   = pt_at(n0 + (o.act ? lg : 0));
        const double* ptp = reinterpret_cast<const double*>(reinterpret_cast<const char*>(A.pt) + (unsigned)(pr.pt_off + px) * 24u);
        o.pt[0] = ptp[0]; o.pt[1] = ptp[1]; o.pt[2] = ptp[2];
        o.rec.u      = __hiloint2double((int)r0.y, (int)r0.x);
        o.rec.v      = __hiloint2double((int)r0.w, (int)r0.z);
        o.rec.depth  = __hiloint2double((int)r1.y, (int)r1.x);
        o.rec.weight = __hiloint2double((int)r1.w, (int)r1.z);
        o.rec.orig = (int)r2.x; o.rec.pk = (int)r2.y;
    };
    SfObs ob;
    load1(0, ob);
    for (int n0 = 0; n0 < si.n_pts; n0 += G)
    {
        const int gc    = min(G, si.n_pts - n0);
        const bool more = n0 + G < si.n_pts;
        // second round for this group (pose at the linearisation point and at the trial state, dc of the camera, outlier flag);
        // first round for the next one
        double pose[7], posen[7], xv[6];
        {
            const double* posep = reinterpret_cast<const double*>(reinterpret_cast<const char*>(A.pose) + (unsigned)(pr.img_off + ob.rec.img()) * 56u);
            const double* posnp = reinterpret_cast<const double*>(reinterpret_cast<const char*>(A.pose_new) + (unsigned)(pr.img_off + ob.rec.img()) * 56u);
#pragma unroll
            for (int k = 0; k < 7; ++k) pose[k] = posep[k];
#pragma unroll
            for (int k = 0; k < 7; ++k) posen[k] = posnp[k];
            const double* xc = x + (ob.rec.cam() < 0 ? 0 : ob.rec.cam()) * 6;
#pragma unroll
            for (int a = 0; a < 6; ++a) xv[a] = xc[a];
        }
        const bool is_out = A.outlier[(unsigned)ob.rec.orig] != 0;
        // per point of the group: b_p by lanes (point, component), V^-1 / position / constancy by lane = point
        const int p2  = pt_at(min(n0 + lane, si.n_pts - 1));
        const int gp2 = pr.pt_off + p2;
        double bpv = 0.0;
        {
            const int g = (lane * 171) >> 9, b = lane - 3 * g;  // lane / 3
            const int pg = pt_at(min(n0 + g, si.n_pts - 1));    // outside the branch
            if (lane < gc * 3) bpv = A.bp[(size_t)(pr.pt_off + pg) * 3 + b];
        }
        double Vi[6] = {0, 0, 0, 0, 0, 0}, cur[3] = {0, 0, 0};
        bool pconst = true;
        if (lane < gc)
        {
#pragma unroll
            for (int k = 0; k < 6; ++k) Vi[k] = A.Vinv[(size_t)gp2 * 6 + k];
            cur[0] = A.pt[(size_t)gp2 * 3]; cur[1] = A.pt[(size_t)gp2 * 3 + 1]; cur[2] = A.pt[(size_t)gp2 * 3 + 2];
            pconst = A.pt_const[gp2] != 0;
        }
        SfObs obn = ob;
        if (more) load1(n0 + G, obn);

        // ---- lane = observation: t = J_p^T (J_c dc) (update_wave) ----
        double t[3] = {0, 0, 0};
        if (ob.act && !(is_out || ob.rec.cam() < 0 || !ob.rec.ptfree()))
        {
            double R[9], r[3];
            quat_to_R(pose, R);
            ObsCore oc;
            const int dim = obs_core(pose, R, ob.pt, pr.K, pr.bf, ob.rec.u, ob.rec.v, ob.rec.depth, ob.rec.weight, r, oc);
            if (dim)
            {
                const double sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
                double sw;
                (void)huber_rho(sq, dim == 3 ? O.huber_stereo : O.huber_mono, sw);
                const double s2 = sw * sw;  // J_c and J_p each carry the IRLS scale
                // J_p^T (J_c dc) = R^T M d with d = dt + dr x Xc (obs_core): u_k = m_k . d, g = sum_k m_k u_k
                const double d0 = xv[0] + (xv[4] * oc.Z - xv[5] * oc.Y);
                const double d1 = xv[1] + (xv[5] * oc.X - xv[3] * oc.Z);
                const double d2 = xv[2] + (xv[3] * oc.Y - xv[4] * oc.X);
                const double a2 = dim == 3 ? oc.a0 : 0.0;
                const double u0 = s2 * (oc.a0 * d0 + oc.c0 * d2), u1 = s2 * (oc.b1 * d1 + oc.c1 * d2), u2 = s2 * (a2 * d0 + oc.c2 * d2);
                const double g0 = oc.a0 * u0 + a2 * u2, g1 = oc.b1 * u1, g2 = oc.c0 * u0 + oc.c1 * u1 + oc.c2 * u2;
#pragma unroll
                for (int b = 0; b < 3; ++b) t[b] = R[b] * g0 + R[3 + b] * g1 + R[6 + b] * g2;
            }
        }
        s_t[lane * 3] = t[0];
        s_t[lane * 3 + 1] = t[1];
        s_t[lane * 3 + 2] = t[2];
        __builtin_amdgcn_wave_barrier();
        // ---- lane = (point, component): g = b_p - sum of t in observation order ----
        if (lane < gc * 3)
        {
            const int g = (lane * 171) >> 9, b = lane - 3 * g;
            double v = bpv;
            for (int a = 0; a < run; ++a) v -= s_t[(g * run + a) * 3 + b];
            s_g[lane] = v;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- lane = point: the new position ----
        if (lane < gc)
        {
            double out[3] = {cur[0], cur[1], cur[2]};
            if (!pconst)
            {
                const double g0 = s_g[lane * 3], g1 = s_g[lane * 3 + 1], g2 = s_g[lane * 3 + 2];
                out[0] = cur[0] + (Vi[0] * g0 + Vi[1] * g1 + Vi[2] * g2);
                out[1] = cur[1] + (Vi[1] * g0 + Vi[3] * g1 + Vi[4] * g2);
                out[2] = cur[2] + (Vi[2] * g0 + Vi[4] * g1 + Vi[5] * g2);
            }
            double* po = A.pt_new + (size_t)gp2 * 3;
            po[0] = out[0]; po[1] = out[1]; po[2] = out[2];
            s_pn[lane * 3] = out[0]; s_pn[lane * 3 + 1] = out[1]; s_pn[lane * 3 + 2] = out[2];
        }
        __builtin_amdgcn_wave_barrier();
        // ---- lane = observation: robust cost at the trial state (cost_wave) ----
        double cost = 0.0;
        if (ob.act && !is_out)
        {
            const double ptn[3] = {s_pn[lg * 3], s_pn[lg * 3 + 1], s_pn[lg * 3 + 2]};
            double R[9], r[3], Jc[1], Jp[1];
            quat_to_R(posen, R);
            const int dim = obs_linearize<false>(posen, R, ptn, pr.K, pr.bf, ob.rec.u, ob.rec.v, ob.rec.depth, ob.rec.weight, r, Jc, Jp);
            if (dim)
            {
                double sw;
                cost = huber_rho(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], dim == 3 ? O.huber_stereo : O.huber_mono, sw);
            }
        }
        s_c[lane] = cost;
        __builtin_amdgcn_wave_barrier();
        if (lane < gc)
        {
            double c = 0.0;
            for (int a = 0; a < run; ++a) c += s_c[lane * run + a];
            A.cost_pt_new[gp2] = c;
        }
        __builtin_amdgcn_wave_barrier();
        ob = obn;
    }
}

// S(c1, c2) = U(c1) [c1 == c2] - sum of the block's partial sums (fixed order) - relative-pose cross terms; lane = element.
// A wavefront takes SUM_NB consecutive blocks of the window and walks their lists side by side (round 5: one block per wavefront was a
// chain of three dependent memory round trips -- list bounds, list, partial sums -- for a handful of additions, 370 000 wavefronts per
// launch of 1024 windows, half of them lower-triangle blocks that left at once).  The additions of a block keep the order of its list.
constexpr int SUM_NB = 4;  // measured per 1024 windows: 1 block per wavefront 118 us, 4: 90 us, 8: 151 us (profiles/r05/r05Q_, r05R_)
__global__ __launch_bounds__(256) void schur_sum(Arrays A, int nbx, int B)
{
    int pb, bx;
    if (B >= 16)
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        pb = (slot / nbx) * 8 + xcd;
        bx = slot - (slot / nbx) * nbx;
    }
    else
    {
        pb = blockIdx.x / nbx;
        bx = blockIdx.x - pb * nbx;
    }
    if (pb >= B) return;
    const Prob pr  = A.prob[pb];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int blk0 = (bx * 4 + wave) * SUM_NB;
    const int nb   = pr.nfc * pr.nfc;
    if (blk0 >= nb) return;
    const int el = lane < 36 ? lane : 0, r = el / 6, c = el - 6 * r;
    int c1[SUM_NB], c2[SUM_NB], e0[SUM_NB], n[SUM_NB];
    bool on[SUM_NB];
    int nmax = 0;
#pragma unroll
    for (int i = 0; i < SUM_NB; ++i)
    {
        const int blk = blk0 + i;
        c1[i]         = blk / pr.nfc;
        c2[i]         = blk - c1[i] * pr.nfc;
        on[i]         = blk < nb && c2[i] >= c1[i];  // wave-uniform
        const int bq  = on[i] ? blk : 0;
        e0[i]         = A.cblk_start[pr.cblk_off + bq];
        n[i]          = on[i] ? A.cblk_start[pr.cblk_off + bq + 1] - e0[i] : 0;
    }
#pragma unroll
    for (int i = 0; i < SUM_NB; ++i) nmax = max(nmax, n[i]);
    double acc[SUM_NB];
#pragma unroll
    for (int i = 0; i < SUM_NB; ++i) acc[i] = 0.0;
    // the lists (one entry per lane) and then four partial sums per block at a time are in flight; the order of the adds is the list's
    for (int k0 = 0; k0 < nmax; k0 += 64)
    {
        int items[SUM_NB];
#pragma unroll
        for (int i = 0; i < SUM_NB; ++i) items[i] = k0 + lane < n[i] ? A.cblk_items[e0[i] + k0 + lane] : 0;
        const int cmax = min(64, nmax - k0);
        for (int k = 0; k < cmax; k += 4)
        {
            double v[SUM_NB][4];
#pragma unroll
            for (int i = 0; i < SUM_NB; ++i)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    const int idx = __builtin_amdgcn_readlane(items[i], min(k + u, 63));
                    v[i][u]       = k0 + k + u < n[i] ? A.s_part[(size_t)idx * 36 + el] : 0.0;
                }
#pragma unroll
            for (int i = 0; i < SUM_NB; ++i)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k0 + k + u < n[i]) acc[i] += v[i][u];
        }
    }
#pragma unroll
    for (int i = 0; i < SUM_NB; ++i)
    {
        if (!on[i]) continue;
        double* S  = A.S + pr.s_off + (size_t)(c1[i] * 6) * pr.n6 + c2[i] * 6;
        double* St = A.S + pr.s_off + (size_t)(c2[i] * 6) * pr.n6 + c1[i] * 6;
        if (c1[i] == c2[i])
        {
            // symmetrise the diagonal block (Y W^T of one camera is symmetric up to rounding)
            const double at = __shfl(acc[i], c * 6 + r);
            const double v  = A.U[(size_t)(pr.cam_off + c1[i]) * 36 + (r <= c ? r * 6 + c : c * 6 + r)] - 0.5 * (acc[i] + at);
            if (lane < 36) S[(size_t)r * pr.n6 + c] = v;
        }
        else
        {
            double a = acc[i];
            if (pr.n_rpc > 0)  // camera-camera terms J(c1)^T J(c2) of the relative pose constraints of this pair
            {
                int code = A.blk_rpc[pr.blkstart_off - pb + blk0 + i];
                while (code != 0)
                {
                    const int k = (code - 1) >> 1, tr = (code - 1) & 1;
                    const double* H = A.rpc_out + (size_t)(pr.rpc_off + k) * RPC_STRIDE + 35;
                    a -= tr ? H[c * 6 + r] : H[r * 6 + c];
                    code = A.rpc_next[pr.rpc_off + k];
                }
            }
            if (lane < 36)
            {
                S[(size_t)r * pr.n6 + c]  = -a;
                St[(size_t)c * pr.n6 + r] = -a;
            }
        }
    }
}

// WPB = wavefronts per block: 1 -- four blocks per workgroup, one wavefront each (batches, big scenes); 4 -- the whole workgroup
// on one block (a single local window: ~190 upper blocks of ~380 entries each leave most of the chip idle and every wavefront
// with six dependent trips through its list; four wavefronts per block make that two, measured 32 -> 15 us per LM iteration).
template <int WPB>
__global__ __launch_bounds__(256) void schur_pass(Arrays A, int zero_rows, int nbx, int B)
{
    __shared__ double s_red[WPB > 1 ? WPB * 36 : 1];
    int pb, bx;
    if (B >= 16)  // batched windows: one XCD per window
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        pb = (slot / nbx) * 8 + xcd;
        bx = slot - (slot / nbx) * nbx;
    }
    else  // few big problems (global BA): spread every problem over all XCDs
    {
        pb = blockIdx.x / nbx;
        bx = blockIdx.x - pb * nbx;
    }
    if (pb >= B) return;
    const Prob pr  = A.prob[pb];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int blk  = WPB == 1 ? bx * 4 + wave : bx;
    if (blk >= pr.nfc * pr.nfc) return;
    const int c1 = blk / pr.nfc, c2 = blk - c1 * pr.nfc;
    if (c2 < c1) return;  // S is symmetric: the lower blocks are written as transposes of the upper ones
    const int e0 = A.blk_start[pr.blkstart_off + blk] + (WPB == 1 ? 0 : wave * 64), e1 = A.blk_start[pr.blkstart_off + blk + 1];
    constexpr int STEP = 64 * WPB;
    double acc[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = 0.0;
    // zero_rows: the linearisation wrote zero rows for inactive observations (point_wave), so their
    // products vanish and the dependent activity lookup is skipped; the next list entry is fetched before
    // the current rows are consumed (one latency level per iteration instead of three)
    int4 en_next = e0 + lane < e1 ? A.blk_ent[pr.ent_off + e0 + lane] : make_int4(0, 0, 0, 0);
    for (int k = e0 + lane; k < e1; k += STEP)
    {
        const int4 en = en_next;
        if (k + STEP < e1) en_next = A.blk_ent[pr.ent_off + k + STEP];
        const int g1 = pr.obs_off + en.x, g2 = pr.obs_off + en.y;
        if (!zero_rows && (A.o_r[(size_t)g1 * 4 + 3] == 0.0 || A.o_r[(size_t)g2 * 4 + 3] == 0.0)) continue;
        // Y(c1) = W(c1) V^-1 is rebuilt from W and the point's 6 V^-1 entries instead of being stored per
        // observation: the rows the blocks of a window re-read are then W only (2.3 MB, fits one XCD's L2)
        const double2* Ap = reinterpret_cast<const double2*>(A.o_W + (size_t)g1 * 18);
        const double2* Wp = reinterpret_cast<const double2*>(A.o_W + (size_t)g2 * 18);
        const double2* Vp = reinterpret_cast<const double2*>(A.Vinv + (size_t)(pr.pt_off + en.z) * 6);
        double wa[18], w[18], y[18];
#pragma unroll
        for (int q = 0; q < 9; ++q)
        {
            const double2 a = Ap[q], b = Wp[q];
            wa[2 * q] = a.x; wa[2 * q + 1] = a.y;
            w[2 * q] = b.x; w[2 * q + 1] = b.y;
        }
        const double2 v01 = Vp[0], v23 = Vp[1], v45 = Vp[2];
        const double v0 = v01.x, v1 = v01.y, v2 = v23.x, v3 = v23.y, v4 = v45.x, v5 = v45.y;
#pragma unroll
        for (int a = 0; a < 6; ++a)
        {
            const double w0 = wa[a * 3], w1 = wa[a * 3 + 1], w2 = wa[a * 3 + 2];
            y[a * 3]     = w0 * v0 + w1 * v1 + w2 * v2;
            y[a * 3 + 1] = w0 * v1 + w1 * v3 + w2 * v4;
            y[a * 3 + 2] = w0 * v2 + w1 * v4 + w2 * v5;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[r * 6 + c] += y[r * 3] * w[c * 3] + y[r * 3 + 1] * w[c * 3 + 1] + y[r * 3 + 2] * w[c * 3 + 2];
    }
#pragma unroll
    for (int k = 0; k < 36; ++k)
    {
        double v = acc[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        acc[k] = v;
    }
    if (WPB > 1)  // the wavefronts' sums in a fixed order (the whole workgroup is on this block: no early return above diverged)
    {
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < 36; ++k) s_red[wave * 36 + k] = acc[k];
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int k = 0; k < 36; ++k)
        {
            double v = s_red[k];
#pragma unroll
            for (int w = 1; w < WPB; ++w) v += s_red[w * 36 + k];
            acc[k] = v;
        }
    }
    if (lane == 0)
    {
        double* S       = A.S + pr.s_off + (size_t)(c1 * 6) * pr.n6 + c2 * 6;
        double* St      = A.S + pr.s_off + (size_t)(c2 * 6) * pr.n6 + c1 * 6;
        const double* U = A.U + (size_t)(pr.cam_off + c1) * 36;
        if (c1 == c2)
        {
            // symmetrise the diagonal block (Y W^T of one camera is symmetric up to rounding)
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = r; c < 6; ++c)
                {
                    const double v = U[r * 6 + c] - 0.5 * (acc[r * 6 + c] + acc[c * 6 + r]);
                    S[(size_t)r * pr.n6 + c] = v;
                    S[(size_t)c * pr.n6 + r] = v;
                }
        }
        else
        {
            if (pr.n_rpc > 0)  // camera-camera terms J(c1)^T J(c2) of the relative pose constraints of this pair
            {
                int code = A.blk_rpc[pr.blkstart_off - pb + blk];
                while (code != 0)
                {
                    const int k = (code - 1) >> 1, tr = (code - 1) & 1;
                    const double* H = A.rpc_out + (size_t)(pr.rpc_off + k) * RPC_STRIDE + 35;
                    for (int r = 0; r < 6; ++r)
                        for (int c = 0; c < 6; ++c) acc[r * 6 + c] -= tr ? H[c * 6 + r] : H[r * 6 + c];
                    code = A.rpc_next[pr.rpc_off + k];
                }
            }
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c)
                {
                    S[(size_t)r * pr.n6 + c]  = -acc[r * 6 + c];
                    St[(size_t)c * pr.n6 + r] = -acc[r * 6 + c];
                }
        }
    }
}

// 6x6 SPD inverse by Cholesky (one thread); falls back to the clamped diagonal
__device__ void inv6_spd(const double* Ain, int lda, double* Ai)
{
    double L[36];
    bool ok = true;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j)
        {
            double s = Ain[i * lda + j];
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j)
            {
                if (s <= 0.0)
                {
                    ok = false;
                    s  = 1.0;
                }
                L[i * 6 + i] = sqrt(s);
            }
            else
                L[i * 6 + j] = s / L[j * 6 + j];
        }
    if (!ok)
    {
        for (int k = 0; k < 36; ++k) Ai[k] = 0.0;
        for (int a = 0; a < 6; ++a) Ai[a * 7] = 1.0 / clampd(Ain[a * lda + a]);
        return;
    }
    for (int c = 0; c < 6; ++c)
    {
        double y[6], x[6];
        for (int i = 0; i < 6; ++i)
        {
            double s = i == c ? 1.0 : 0.0;
            for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k];
            y[i] = s / L[i * 6 + i];
        }
        for (int i = 5; i >= 0; --i)
        {
            double s = y[i];
            for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
            x[i] = s / L[i * 6 + i];
        }
        for (int i = 0; i < 6; ++i) Ai[i * 6 + c] = x[i];
    }
}

constexpr int PCG_THREADS = 256;
// dynamic LDS: r, z, p, Ap (n6 each) | partial sums (4 * n6) | Minv (nfc*36) | S (n6*n6, when s_in_lds)
// S_IN_LDS is a template parameter, not an argument: with `S = s_in_lds ? Sl : Sg` the matvec read S through a generic
// pointer, i.e. FLAT loads that resolve the LDS aperture per access (texture-address unit 36 % busy in a kernel that
// should not touch it); the instantiations read it with ds_read / global_load.
template <bool S_IN_LDS>
__global__ __launch_bounds__(PCG_THREADS) void pcg_solve(Arrays A, Opt O)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[PCG_THREADS];
    const int pb  = blockIdx.x;
    const Prob pr = A.prob[pb];
    const int n6 = pr.n6, nfc = pr.nfc;
    const int tid = threadIdx.x;
    double* r  = reinterpret_cast<double*>(smem_raw);
    double* z  = r + n6;
    double* p  = z + n6;
    double* Ap = p + n6;
    double* ps = Ap + n6;       // [4][n6] partial sums of the split matvec
    double* xs = ps + 4 * n6;   // the solution stays in LDS until the end (a global read-modify-write per iteration cost a memory round trip)
    double* Mi = xs + n6;
    double* Sl = Mi + nfc * 36;
    const double* Sg  = A.S + pr.s_off;
    const double* rhs = A.rhs + pr.vec_off;
    double* x         = A.x + pr.vec_off;
    if (n6 == 0) return;
    if constexpr (S_IN_LDS)
    {
        // S to LDS with sixteen 16-byte loads in flight per thread (n6 is even and s_off a multiple of n6^2: 16-byte aligned), then the
        // block inverses read their 6x6 blocks from LDS: element by element from global memory the copy and the 36 dependent loads
        // of a Cholesky were ~25 us of a 70 us single-window solve
        const double2* Sg2 = reinterpret_cast<const double2*>(Sg);
        double2* Sl2       = reinterpret_cast<double2*>(Sl);
        const int n2       = (n6 * n6) >> 1;
        for (int i0 = tid; i0 < n2; i0 += 16 * PCG_THREADS)
        {
            double2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = i0 + u * PCG_THREADS < n2 ? Sg2[i0 + u * PCG_THREADS] : double2{0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (i0 + u * PCG_THREADS < n2) Sl2[i0 + u * PCG_THREADS] = v[u];
        }
        __syncthreads();
        for (int c = tid; c < nfc; c += PCG_THREADS) inv6_spd(Sl + (c * 6) * n6 + c * 6, n6, Mi + c * 36);
    }
    else
        for (int c = tid; c < nfc; c += PCG_THREADS) inv6_spd(Sg + (size_t)(c * 6) * n6 + c * 6, n6, Mi + c * 36);
    double part = 0.0;
    for (int q = tid; q < n6; q += PCG_THREADS)
    {
        const double v = rhs[q];
        r[q] = v;
        xs[q] = 0.0;
        part += v * v;
    }
    const double bnorm2 = block_sum<PCG_THREADS>(part, red, tid);
    part = 0.0;
    for (int q = tid; q < n6; q += PCG_THREADS)
    {
        const int c = q / 6, a = q - c * 6;
        double s = 0.0;
        for (int b = 0; b < 6; ++b) s += Mi[c * 36 + a * 6 + b] * r[c * 6 + b];
        z[q] = s;
        p[q] = s;
        part += r[q] * s;
    }
    double rz = block_sum<PCG_THREADS>(part, red, tid);
    double rn2 = bnorm2;  // |r|^2 of the current residual (x = 0: r = rhs); later iterations get it with r.z in one reduction
    const double stop2 = O.pcg_tol * O.pcg_tol * bnorm2;
    if constexpr (S_IN_LDS)
        if (n6 <= 128 && !O.pcg_general)
        {
            // Local-BA sized systems (<= 21 free cameras).  Every wavefront holds the WHOLE iteration state in registers (two
            // rows per lane: r, p, x, its rows of the block-Jacobi inverse) and runs the same arithmetic on it, so the four
            // wavefronts never have to tell each other anything except their quarter of S p: each multiplies its column
            // quarter (two rows of S per ds_read2st64_b64 -- a wavefront can only keep ~16 LDS operations outstanding, the
            // number of instructions is what counts), the four partial products meet in a double-buffered LDS block, and that
            // is the ONE workgroup barrier of the iteration.  The private copies of r and p that the preconditioner and the
            // product read back from LDS are per wavefront (program order, no barrier), the scalar products are DPP sums.
            // The general form below pays seven barriers and five LDS trees per iteration: 3.4 us against ~1.
            __syncthreads();
            const int wave = tid >> 6, lane = tid & 63;
            const bool v0 = lane < n6, v1 = lane + 64 < n6;
            const int q0 = v0 ? lane : 0, q1 = v1 ? lane + 64 : 0;
            double r0 = v0 ? r[q0] : 0.0, r1 = v1 ? r[q1] : 0.0;
            double p0 = v0 ? p[q0] : 0.0, p1 = v1 ? p[q1] : 0.0;
            double x0 = 0.0, x1 = 0.0;
            const int c0 = q0 / 6, c1 = q1 / 6;
            double m0[6], m1[6];
#pragma unroll
            for (int b = 0; b < 6; ++b)
            {
                m0[b] = v0 ? Mi[c0 * 36 + (q0 - c0 * 6) * 6 + b] : 0.0;
                m1[b] = v1 ? Mi[c1 * 36 + (q1 - c1 * 6) * 6 + b] : 0.0;
            }
            __syncthreads();                  // everybody has read r / p: the vector block is re-used
            double* rw = r + wave * n6;       // r, z, p, Ap: 4 * n6 doubles -> one private r per wavefront
            double* pw = ps + wave * n6;      // ps: 4 * n6 doubles -> one private p per wavefront
            double* ex = Sl + n6 * n6 + 128;  // [2][4][128] partial products (behind S and its read padding)
            if (v0) pw[q0] = p0;
            if (v1) pw[q1] = p1;
            // this wavefront's column quarter (even bounds: p is read two columns at a time)
            const int cb = (((n6 + 3) >> 2) + 1) & ~1;
            const int ub = min(wave * cb, n6), ue = min(ub + cb, n6);
            const double* Sq = Sl + lane;
            int iters = 0, par = 0;
            for (int k = 0; k < O.max_pcg; ++k)
            {
                if (rn2 <= stop2) break;  // the same decision in every wavefront (identical arithmetic)
                double s0 = 0.0, s1 = 0.0, t0 = 0.0, t1 = 0.0;
                struct Blk
                {
                    double a[8], b[8], pu[8];
                };
                auto load = [&](Blk& B, int u)
                {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                    {
                        const double* row = Sq + (u + e) * n6;
                        B.a[e]            = row[0];
                        B.b[e]            = row[64];  // rows that do not exist read what follows (padding) and are masked
                    }
#pragma unroll
                    for (int e = 0; e < 8; e += 2)
                    {
                        const double2 pv = *reinterpret_cast<const double2*>(pw + u + e);  // 16-byte aligned: n6, u even
                        B.pu[e]          = pv.x;
                        B.pu[e + 1]      = pv.y;
                    }
                };
                auto proc = [&](const Blk& B)
                {
#pragma unroll
                    for (int e = 0; e < 8; e += 2)
                    {
                        s0 += B.a[e] * B.pu[e];
                        s1 += B.b[e] * B.pu[e];
                        t0 += B.a[e + 1] * B.pu[e + 1];
                        t1 += B.b[e + 1] * B.pu[e + 1];
                    }
                };
                const int nblk = (ue - ub) >> 3;
                Blk B0, B1;
                if (nblk > 0) load(B0, ub);
                for (int kb = 0; kb < nblk; kb += 2)
                {
                    if (kb + 1 < nblk) load(B1, ub + 8 * (kb + 1));
                    proc(B0);
                    if (kb + 1 < nblk)
                    {
                        if (kb + 2 < nblk) load(B0, ub + 8 * (kb + 2));
                        proc(B1);
                    }
                }
                for (int u = ub + 8 * nblk; u < ue; ++u)
                {
                    const double pu = pw[u];
                    s0 += Sq[u * n6] * pu;
                    s1 += Sq[u * n6 + 64] * pu;
                }
                double* exw = ex + par * 512;
                exw[wave * 128 + lane]      = v0 ? s0 + t0 : 0.0;
                exw[wave * 128 + 64 + lane] = v1 ? s1 + t1 : 0.0;
                __syncthreads();
                s0  = (exw[lane] + exw[128 + lane]) + (exw[256 + lane] + exw[384 + lane]);
                s1  = (exw[64 + lane] + exw[192 + lane]) + (exw[320 + lane] + exw[448 + lane]);
                par ^= 1;
                const double pAp = wave_sum64_dpp(p0 * s0 + p1 * s1);  // p is 0 in rows that do not exist
                if (pAp <= 0.0) break;
                const double alpha = rz / pAp;
                x0 += alpha * p0;
                x1 += alpha * p1;
                r0 = v0 ? r0 - alpha * s0 : 0.0;
                r1 = v1 ? r1 - alpha * s1 : 0.0;
                if (v0) rw[q0] = r0;
                if (v1) rw[q1] = r1;
                __builtin_amdgcn_wave_barrier();
                double z0 = 0.0, z1 = 0.0;
#pragma unroll
                for (int b = 0; b < 6; ++b)
                {
                    z0 += m0[b] * rw[c0 * 6 + b];
                    z1 += m1[b] * rw[c1 * 6 + b];
                }
                const double rz_new = wave_sum64_dpp(r0 * z0 + r1 * z1);
                rn2                 = wave_sum64_dpp(r0 * r0 + r1 * r1);
                const double beta   = rz_new / rz;
                rz                  = rz_new;
                p0                  = z0 + beta * p0;
                p1                  = z1 + beta * p1;
                if (v0) pw[q0] = p0;
                if (v1) pw[q1] = p1;
                __builtin_amdgcn_wave_barrier();
                ++iters;
            }
            if (wave == 0)
            {
                if (v0) x[q0] = x0;
                if (v1) x[q1] = x1;
                if (lane == 0) A.state[pb].pcg_iters += iters;
            }
            return;
        }
    // matvec split: `parts` threads share a row, each a contiguous column chunk (fixed combine order)
    int parts = PCG_THREADS / n6;
    parts     = parts < 1 ? 1 : (parts > 4 ? 4 : parts);
    const int chunk = (n6 + parts - 1) / parts;
    int iters = 0;
    for (int k = 0; k < O.max_pcg; ++k)
    {
        if (rn2 <= stop2) break;
        for (int t = tid; t < n6 * parts; t += PCG_THREADS)
        {
            const int pi = t / n6, q = t - pi * n6;
            const int t0 = pi * chunk, t1 = min(t0 + chunk, n6);
            double s = 0.0;
            int u = t0;
            for (; u + 8 <= t1; u += 8)  // 16 reads in flight; the adds keep their order
            {
                double a[8], b[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                {
                    a[e] = S_IN_LDS ? Sl[(u + e) * n6 + q] : Sg[(size_t)(u + e) * n6 + q];  // column q == row q (symmetric)
                    b[e] = p[u + e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) s += a[e] * b[e];
            }
            for (; u < t1; ++u) s += (S_IN_LDS ? Sl[u * n6 + q] : Sg[(size_t)u * n6 + q]) * p[u];
            ps[pi * n6 + q] = s;
        }
        __syncthreads();
        part = 0.0;
        for (int q = tid; q < n6; q += PCG_THREADS)
        {
            double s = ps[q];
            for (int pi = 1; pi < parts; ++pi) s += ps[pi * n6 + q];
            Ap[q] = s;
            part += p[q] * s;
        }
        const double pAp = block_sum<PCG_THREADS>(part, red, tid);
        if (pAp <= 0.0) break;
        const double alpha = rz / pAp;
        for (int q = tid; q < n6; q += PCG_THREADS)
        {
            xs[q] += alpha * p[q];
            r[q] -= alpha * Ap[q];
        }
        __syncthreads();
        part = 0.0;
        double part_rr = 0.0;
        for (int q = tid; q < n6; q += PCG_THREADS)
        {
            const int c = q / 6, a = q - c * 6;
            double s = 0.0;
            for (int b = 0; b < 6; ++b) s += Mi[c * 36 + a * 6 + b] * r[c * 6 + b];
            z[q] = s;
            part += r[q] * s;
            part_rr += r[q] * r[q];
        }
        block_sum2<PCG_THREADS>(part, part_rr, red, tid);
        const double rz_new = part;
        rn2                 = part_rr;
        const double beta   = rz_new / rz;
        rz                  = rz_new;
        for (int q = tid; q < n6; q += PCG_THREADS) p[q] = z[q] + beta * p[q];
        __syncthreads();
        ++iters;
    }
    for (int q = tid; q < n6; q += PCG_THREADS) x[q] = xs[q];
    if (tid == 0) A.state[pb].pcg_iters += iters;
}

// Local-BA sized systems (n6 <= 128), S in REGISTERS.  Same algorithm, same summation order as the replicated
// four-wavefront loop of pcg_solve<true> (bit-identical results), but each wavefront keeps its column quarter of S -- two
// rows per lane x <= 32 columns = <= 64 doubles -- in VGPRs, loaded once straight from global memory: the 58 ds_reads of S
// per lane and iteration are gone (what is left reads p, a broadcast), and without the 104 KB of S in LDS two workgroups
// share a compute unit (the loop is a dependency chain: a second one fills its bubbles).  LDS: r, z, p, Ap | ps | xs |
// Minv | diagonal blocks | [2][4][128] partial products = 27 KB for 19 free cameras.
constexpr int PCG_SMALL_COLS = 32;
__global__ __launch_bounds__(PCG_THREADS, 2) void pcg_small(Arrays A, Opt O)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[PCG_THREADS];
    const int pb  = blockIdx.x;
    const Prob pr = A.prob[pb];
    const int n6 = pr.n6, nfc = pr.nfc;
    const int tid = threadIdx.x;
    double* r  = reinterpret_cast<double*>(smem_raw);
    double* z  = r + n6;
    double* p  = z + n6;
    double* ps = p + 2 * n6;   // behind Ap's slot (the vector block r, z, p, Ap becomes the wavefronts' private r)
    double* Mi = ps + 5 * n6;  // ps [4][n6] + xs [n6] (xs unused here: x lives in registers)
    double* Sd = Mi + nfc * 36;
    double* ex = Sd + nfc * 36;  // [2][4][128]
    const double* Sg  = A.S + pr.s_off;
    const double* rhs = A.rhs + pr.vec_off;
    double* x         = A.x + pr.vec_off;
    if (n6 == 0) return;
    const int wave = tid >> 6, lane = tid & 63;
    const bool v0 = lane < n6, v1 = lane + 64 < n6;
    // this wavefront's column quarter (even bounds, as in pcg_solve) into registers; issued first, consumed after the set-up
    const int cb = (((n6 + 3) >> 2) + 1) & ~1;
    const int ub = min(wave * cb, n6), ue = min(ub + cb, n6);
    const int ncols = ue - ub, nb8 = (ncols >> 3) << 3;
    double sa[PCG_SMALL_COLS], sb[PCG_SMALL_COLS];
#pragma unroll
    for (int j = 0; j < PCG_SMALL_COLS; ++j)
    {
        const double* row = Sg + (size_t)(ub + j) * n6 + lane;  // column q == row q (symmetric): coalesced along the row
        sa[j] = j < ncols && v0 ? row[0] : 0.0;
        sb[j] = j < ncols && v1 ? row[64] : 0.0;
    }
    // the diagonal blocks through LDS (one load per thread instead of 36 dependent ones in the Cholesky), then the block inverses
    for (int t = tid; t < nfc * 36; t += PCG_THREADS)
    {
        const int c = t / 36, e = t - c * 36, i = e / 6, j = e - i * 6;
        Sd[t] = Sg[(size_t)(c * 6 + i) * n6 + c * 6 + j];
    }
    __syncthreads();
    for (int c = tid; c < nfc; c += PCG_THREADS) inv6_spd(Sd + c * 36, 6, Mi + c * 36);
    double part = 0.0;
    for (int q = tid; q < n6; q += PCG_THREADS)
    {
        const double v = rhs[q];
        r[q] = v;
        part += v * v;
    }
    const double bnorm2 = block_sum<PCG_THREADS>(part, red, tid);
    part = 0.0;
    for (int q = tid; q < n6; q += PCG_THREADS)
    {
        const int c = q / 6, a = q - c * 6;
        double sacc = 0.0;
        for (int b = 0; b < 6; ++b) sacc += Mi[c * 36 + a * 6 + b] * r[c * 6 + b];
        z[q] = sacc;
        p[q] = sacc;
        part += r[q] * sacc;
    }
    double rz = block_sum<PCG_THREADS>(part, red, tid);
    double rn2 = bnorm2;
    const double stop2 = O.pcg_tol * O.pcg_tol * bnorm2;
    __syncthreads();
    const int q0 = v0 ? lane : 0, q1 = v1 ? lane + 64 : 0;
    double r0 = v0 ? r[q0] : 0.0, r1 = v1 ? r[q1] : 0.0;
    double p0 = v0 ? p[q0] : 0.0, p1 = v1 ? p[q1] : 0.0;
    double x0 = 0.0, x1 = 0.0;
    const int c0 = q0 / 6, c1 = q1 / 6;
    double m0[6], m1[6];
#pragma unroll
    for (int b = 0; b < 6; ++b)
    {
        m0[b] = v0 ? Mi[c0 * 36 + (q0 - c0 * 6) * 6 + b] : 0.0;
        m1[b] = v1 ? Mi[c1 * 36 + (q1 - c1 * 6) * 6 + b] : 0.0;
    }
    __syncthreads();              // everybody has read r / p: the vector block is re-used
    double* rw = r + wave * n6;   // r, z, p, Ap: 4 * n6 doubles -> one private r per wavefront
    double* pw = ps + wave * n6;  // ps: 4 * n6 doubles -> one private p per wavefront
    if (v0) pw[q0] = p0;
    if (v1) pw[q1] = p1;
    __builtin_amdgcn_wave_barrier();
    int iters = 0, par = 0;
    for (int k = 0; k < O.max_pcg; ++k)
    {
        if (rn2 <= stop2) break;  // the same decision in every wavefront (identical arithmetic)
        double s0 = 0.0, s1 = 0.0, t0 = 0.0, t1 = 0.0;
        const double* pq = pw + ub;
#pragma unroll
        for (int j = 0; j < PCG_SMALL_COLS; j += 2)
        {
            if (j < ncols)  // ub and ncols are even: p two columns at a time, 16-byte aligned
            {
                const double2 pv = *reinterpret_cast<const double2*>(pq + j);
                s0 += sa[j] * pv.x;
                s1 += sb[j] * pv.x;
                if (j + 1 < nb8)  // inside the blocks of eight the odd columns have their own accumulator (pcg_solve's order)
                {
                    t0 += sa[j + 1] * pv.y;
                    t1 += sb[j + 1] * pv.y;
                }
                else if (j + 1 < ncols)
                {
                    s0 += sa[j + 1] * pv.y;
                    s1 += sb[j + 1] * pv.y;
                }
            }
        }
        double* exw = ex + par * 512;
        exw[wave * 128 + lane]      = v0 ? s0 + t0 : 0.0;
        exw[wave * 128 + 64 + lane] = v1 ? s1 + t1 : 0.0;
        __syncthreads();
        s0  = (exw[lane] + exw[128 + lane]) + (exw[256 + lane] + exw[384 + lane]);
        s1  = (exw[64 + lane] + exw[192 + lane]) + (exw[320 + lane] + exw[448 + lane]);
        par ^= 1;
        const double pAp = wave_sum64_dpp(p0 * s0 + p1 * s1);  // p is 0 in rows that do not exist
        if (pAp <= 0.0) break;
        const double alpha = rz / pAp;
        x0 += alpha * p0;
        x1 += alpha * p1;
        r0 = v0 ? r0 - alpha * s0 : 0.0;
        r1 = v1 ? r1 - alpha * s1 : 0.0;
        if (v0) rw[q0] = r0;
        if (v1) rw[q1] = r1;
        __builtin_amdgcn_wave_barrier();
        double z0 = 0.0, z1 = 0.0;
#pragma unroll
        for (int b = 0; b < 6; ++b)
        {
            z0 += m0[b] * rw[c0 * 6 + b];
            z1 += m1[b] * rw[c1 * 6 + b];
        }
        const double rz_new = wave_sum64_dpp(r0 * z0 + r1 * z1);
        rn2                 = wave_sum64_dpp(r0 * r0 + r1 * r1);
        const double beta   = rz_new / rz;
        rz                  = rz_new;
        p0                  = z0 + beta * p0;
        p1                  = z1 + beta * p1;
        if (v0) pw[q0] = p0;
        if (v1) pw[q1] = p1;
        __builtin_amdgcn_wave_barrier();
        ++iters;
    }
    if (wave == 0)
    {
        if (v0) x[q0] = x0;
        if (v1) x[q1] = x1;
        if (lane == 0) A.state[pb].pcg_iters += iters;
    }
}

// ---- PCG for reduced systems that do not fit one workgroup's LDS (global BA: hundreds of keyframes) ----
// Same algorithm as pcg_solve, spread over the chip: vectors live in HBM, S p is a (row chunk x column
// part) grid of workgroups with the column parts combined in fixed order, the scalar products are
// per-workgroup partial sums combined in fixed order by every workgroup that needs them (deterministic,
// no atomics).  One PCG iteration = 4 launches (matvec, combine, update, direction); convergence is
// re-derived from the partial sums by every workgroup, the last launch of an iteration latches it.
struct PcgLarge
{
    double* r;     // [tot_vec]
    double* z;
    double* p;
    double* Ap;
    double* Minv;  // [tot_cam][36]
    double* ps;    // [parts][tot_vec]
    double* prr;   // [2][B][G] partial r.r   (double-buffered by iteration parity)
    double* prz;   // [2][B][G] partial r.z
    double* ppap;  // [B][G]    partial p.Ap
    double* scal;  // [B][4]    stop2, done, -, -
    int G, parts, tot_vec, B;
    // the one-launch form (pcgl_persist, one problem): p double-buffered, per-workgroup partial sums, the grid barrier's counter
    double* p2;       // [tot_vec]  the other direction buffer
    double* wrr;      // [2][PERSIST_WGS]  partial r.r  (by iteration parity)
    double* wrz;      // [2][PERSIST_WGS]  partial r.z
    double* wpap;     // [PERSIST_WGS]     partial p.Ap
    unsigned* bar;    // [BAR_WORDS] the grid barrier's phase flags: per workgroup, per group of eight, per group generation (zeroed by pcgl_init)
    int bar_flat;     // SNK_BA_FLAT_BARRIER=1: every workgroup polls every flag (the round-5 barrier)
    int persist_one;  // pcgl_persist1 (one grid barrier per PCG iteration; r, z, p private in LDS) instead of pcgl_persist
    int persist_wgs;  // workgroups of the launch (all resident: cooperative launch)
    int persist_rows;  // pcgl_persist_reg: rows of S per workgroup (8 or 16)
    int timing;       // SNK_BA_PCG_TIMING=1 (diagnostic): workgroup 0 of pcgl_persist_reg adds its cycles per phase to ps[0..5]
};
constexpr int PERSIST_WGS_MAX = 1024;
static inline int snk_env_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return e && atoi(e) > 0 ? atoi(e) : dflt;
}

__device__ __forceinline__ double wave_sum64(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ double sum_partials(const double* a, int g)
{
    double t = 0.0;
    for (int i = 0; i < g; ++i) t += a[i];
    return t;
}

// one thread per free camera (64 per workgroup): Minv, r = rhs, x = 0, z = p = Minv r, partial r.r / r.z
__global__ __launch_bounds__(64) void pcgl_init(Arrays A, Opt O, PcgLarge W)
{
    const int pb  = blockIdx.y;
    const Prob pr = A.prob[pb];
    const int c   = blockIdx.x * 64 + threadIdx.x;
    double rr = 0.0, rz = 0.0;
    if (c < pr.nfc)
    {
        const int n6 = pr.n6;
        double* Mi   = W.Minv + (size_t)(pr.cam_off + c) * 36;
        inv6_spd(A.S + pr.s_off + (size_t)(c * 6) * n6 + c * 6, n6, Mi);
        double rv[6];
        for (int a = 0; a < 6; ++a)
        {
            rv[a] = A.rhs[pr.vec_off + c * 6 + a];
            rr += rv[a] * rv[a];
        }
        for (int a = 0; a < 6; ++a)
        {
            double sacc = 0.0;
            for (int b = 0; b < 6; ++b) sacc += Mi[a * 6 + b] * rv[b];
            const int q = pr.vec_off + c * 6 + a;
            W.r[q] = rv[a];
            W.z[q] = sacc;
            W.p[q] = sacc;
            A.x[q] = 0.0;
            rz += rv[a] * sacc;
        }
    }
    rr = wave_sum64(rr);
    rz = wave_sum64(rz);
    if (pb == 0 && blockIdx.x == 0 && W.bar)
        for (int i = threadIdx.x; i < PERSIST_WGS_MAX + 32 + 8 * 32 /* BAR_WORDS */; i += 64) W.bar[i] = 0u;  // the grid barrier's flags of pcgl_persist
    if (threadIdx.x == 0)
    {
        W.prr[(size_t)pb * W.G + blockIdx.x] = rr;  // buffer 0
        W.prz[(size_t)pb * W.G + blockIdx.x] = rz;
        if (blockIdx.x == 0)
        {
            W.scal[pb * 4 + 1] = 0.0;  // done
            W.scal[pb * 4 + 0] = -1.0;  // stop2 not known yet (needs all partials): derived in pcgl_matvec of iteration 0

        }
    }
}

// converged / broken?  buf = parity of the iteration.  stop2 = tol^2 * |rhs|^2 is latched by iteration 0.
__device__ __forceinline__ bool pcgl_stop(const PcgLarge& W, const Prob& pr, const Opt& O, int pb, int k, int g_used)
{
    if (pr.n6 == 0 || W.scal[pb * 4 + 1] != 0.0) return true;
    const double* prr = W.prr + ((size_t)(k & 1) * W.B + pb) * W.G;
    const double rn2  = sum_partials(prr, g_used);
    const double stop2 = k == 0 ? O.pcg_tol * O.pcg_tol * rn2 : W.scal[pb * 4 + 0];
    return rn2 <= stop2;
}

// ps[part][q] = sum over the part's columns u of S[u][q] p[u]  (S symmetric: column q == row q)
__global__ __launch_bounds__(256) void pcgl_matvec(Arrays A, Opt O, PcgLarge W, int k)
{
    const int pb  = blockIdx.y;
    const Prob pr = A.prob[pb];
    const int n6  = pr.n6;
    const int g_used = (pr.nfc + 63) / 64;
    if (pcgl_stop(W, pr, O, pb, k, g_used)) return;
    const int rowchunks = (n6 + 255) / 256;
    const int rc = blockIdx.x % rowchunks, part = blockIdx.x / rowchunks;
    if (part >= W.parts) return;
    const int q = rc * 256 + threadIdx.x;
    const int chunk = (n6 + W.parts - 1) / W.parts;
    const int u0 = part * chunk, u1 = min(u0 + chunk, n6);
    if (q >= n6) return;
    const double* S = A.S + pr.s_off;
    const double* p = W.p + pr.vec_off;
    double acc = 0.0;
    for (int u = u0; u < u1; ++u) acc += S[(size_t)u * n6 + q] * p[u];
    W.ps[(size_t)part * W.tot_vec + pr.vec_off + q] = acc;
}

// Ap = sum of the parts (fixed order); partial p.Ap
__global__ __launch_bounds__(64) void pcgl_combine(Arrays A, Opt O, PcgLarge W, int k)
{
    const int pb  = blockIdx.y;
    const Prob pr = A.prob[pb];
    const int g_used = (pr.nfc + 63) / 64;
    if ((int)blockIdx.x >= g_used || pcgl_stop(W, pr, O, pb, k, g_used)) return;
    const int c = blockIdx.x * 64 + threadIdx.x;
    double pap  = 0.0;
    if (c < pr.nfc)
        for (int a = 0; a < 6; ++a)
        {
            const int q = pr.vec_off + c * 6 + a;
            double sacc = W.ps[q];
            for (int pi = 1; pi < W.parts; ++pi) sacc += W.ps[(size_t)pi * W.tot_vec + q];
            W.Ap[q] = sacc;
            pap += W.p[q] * sacc;
        }
    pap = wave_sum64(pap);
    if (threadIdx.x == 0) W.ppap[(size_t)pb * W.G + blockIdx.x] = pap;
}

// x += alpha p, r -= alpha Ap, z = Minv r; partial r.r / r.z of the NEXT parity
__global__ __launch_bounds__(64) void pcgl_update(Arrays A, Opt O, PcgLarge W, int k)
{
    const int pb  = blockIdx.y;
    const Prob pr = A.prob[pb];
    const int g_used = (pr.nfc + 63) / 64;
    if ((int)blockIdx.x >= g_used || pcgl_stop(W, pr, O, pb, k, g_used)) return;
    const double pAp = sum_partials(W.ppap + (size_t)pb * W.G, g_used);
    if (pAp <= 0.0) return;  // pcgl_direction latches the break
    const double rz    = sum_partials(W.prz + ((size_t)(k & 1) * W.B + pb) * W.G, g_used);
    const double alpha = rz / pAp;
    const int c = blockIdx.x * 64 + threadIdx.x;
    double rr = 0.0, rzn = 0.0;
    if (c < pr.nfc)
    {
        const double* Mi = W.Minv + (size_t)(pr.cam_off + c) * 36;
        double rv[6];
        for (int a = 0; a < 6; ++a)
        {
            const int q = pr.vec_off + c * 6 + a;
            A.x[q] += alpha * W.p[q];
            rv[a] = W.r[q] - alpha * W.Ap[q];
            W.r[q] = rv[a];
            rr += rv[a] * rv[a];
        }
        for (int a = 0; a < 6; ++a)
        {
            double sacc = 0.0;
            for (int b = 0; b < 6; ++b) sacc += Mi[a * 6 + b] * rv[b];
            W.z[pr.vec_off + c * 6 + a] = sacc;
            rzn += rv[a] * sacc;
        }
    }
    rr  = wave_sum64(rr);
    rzn = wave_sum64(rzn);
    if (threadIdx.x == 0)
    {
        W.prr[((size_t)((k + 1) & 1) * W.B + pb) * W.G + blockIdx.x] = rr;
        W.prz[((size_t)((k + 1) & 1) * W.B + pb) * W.G + blockIdx.x] = rzn;
    }
}

// p = z + beta p; latches stop2 (iteration 0), convergence and the pAp <= 0 break; counts iterations
__global__ __launch_bounds__(64) void pcgl_direction(Arrays A, Opt O, PcgLarge W, int k)
{
    const int pb  = blockIdx.y;
    const Prob pr = A.prob[pb];
    const int g_used = (pr.nfc + 63) / 64;
    if ((int)blockIdx.x >= g_used || pr.n6 == 0 || W.scal[pb * 4 + 1] != 0.0) return;
    const double rn2   = sum_partials(W.prr + ((size_t)(k & 1) * W.B + pb) * W.G, g_used);
    const double stop2 = k == 0 ? O.pcg_tol * O.pcg_tol * rn2 : W.scal[pb * 4 + 0];
    const double pAp   = sum_partials(W.ppap + (size_t)pb * W.G, g_used);
    const bool stop    = rn2 <= stop2;
    const bool brk     = !stop && pAp <= 0.0;
    if (!stop && !brk)
    {
        const double rz   = sum_partials(W.prz + ((size_t)(k & 1) * W.B + pb) * W.G, g_used);
        const double rzn  = sum_partials(W.prz + ((size_t)((k + 1) & 1) * W.B + pb) * W.G, g_used);
        const double beta = rzn / rz;
        const int c       = blockIdx.x * 64 + threadIdx.x;
        if (c < pr.nfc)
            for (int a = 0; a < 6; ++a)
            {
                const int q = pr.vec_off + c * 6 + a;
                W.p[q]      = W.z[q] + beta * W.p[q];
            }
    }
    // scal (stop2, done) is latched by pcgl_latch, a separate launch: workgroups of this launch may
    // still be reading it.
}

// single-thread latch between iterations (runs after pcgl_direction of iteration k has completed)
__global__ void pcgl_latch(Arrays A, Opt O, PcgLarge W, int k)
{
    const int pb = blockIdx.x * blockDim.x + threadIdx.x;
    if (pb >= W.B) return;
    const Prob pr = A.prob[pb];
    if (pr.n6 == 0 || W.scal[pb * 4 + 1] != 0.0) return;
    const int g_used   = (pr.nfc + 63) / 64;
    const double rn2   = sum_partials(W.prr + ((size_t)(k & 1) * W.B + pb) * W.G, g_used);
    const double stop2 = k == 0 ? O.pcg_tol * O.pcg_tol * rn2 : W.scal[pb * 4 + 0];
    if (k == 0) W.scal[pb * 4 + 0] = stop2;
    const double pAp = sum_partials(W.ppap + (size_t)pb * W.G, g_used);
    if (rn2 <= stop2 || pAp <= 0.0)
        W.scal[pb * 4 + 1] = 1.0;
    else
        A.state[pb].pcg_iters += 1;
}

// ---- the same PCG as ONE launch (round 5; one problem -- a global scene is one problem) ----
// The multi-launch form above pays five launches per PCG iteration and ran `pcgl_combine` on ceil(cameras / 64) wavefronts: FullBA(4) on
// 300 keyframes was 160 iterations x 85 us = 13.5 of its 14.4 ms (profiles/r05/r05i_gba_kernel_stats.csv), none of it bandwidth.  Here
// all workgroups stay resident (cooperative launch) and the phases of an iteration are separated by a grid barrier:
//   1 matvec   (A p)[q] for the workgroup's rows q: S is symmetric, so row q is contiguous and one wavefront streams it against the
//              direction p = z + beta p_prev held in LDS; partial p.Ap per workgroup                                -- barrier
//   2 update   x += alpha p, r -= alpha Ap, z = Minv r (a thread per element, cameras never straddle workgroups),
//              partial r.r / r.z of the next parity                                                                 -- barrier
// Scalars (r.r, r.z, p.Ap, alpha, beta, the stopping test) are re-derived by every workgroup from the partial sums in a fixed
// order: deterministic, no floating-point atomics.  The barrier is an arrival counter in HBM: __syncthreads, one agent-scope
// release increment per workgroup, a spin on an agent-scope acquire load, __syncthreads.
constexpr int BAR_PER_XCD = PERSIST_WGS_MAX / 8;      // arrival flags of group x at bar[x * BAR_PER_XCD + (workgroup >> 3)]
constexpr int BAR_XFLAG   = PERSIST_WGS_MAX;          // the eight group flags, one 32-byte run
constexpr int BAR_GEN     = PERSIST_WGS_MAX + 32;     // the groups' generation words, 128 bytes apart
constexpr int BAR_WORDS   = BAR_GEN + 8 * 32;
static_assert(BAR_WORDS <= 2 * (PERSIST_WGS_MAX + 8), "the barrier's words live in the (PERSIST_WGS_MAX + 8) doubles behind wpap");

__device__ __forceinline__ void grid_barrier_flat(unsigned* bar, unsigned n_wgs, unsigned& phase)
{
    // One flag per workgroup, no read-modify-write: an arrival counter costs one device-scope atomic per workgroup on ONE address, and
    // those are served one after the other at the memory side (the XCDs' L2s are not coherent with each other): 512 arrivals took
    // ~45 us per barrier (profiles/r05/r05j_gba_persist_counter_barrier.txt).  Here a workgroup publishes its phase with a plain
    // device-scope store and its first wavefront polls everybody's flags (n_wgs / 64 coalesced loads per round).
    __syncthreads();
    ++phase;
    if (threadIdx.x < 64)
    {
        if (threadIdx.x == 0) __hip_atomic_store(bar + blockIdx.x, phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        for (;;)
        {
            bool ok = true;
            for (unsigned i = threadIdx.x; i < n_wgs; i += 64) ok = ok && __hip_atomic_load(bar + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= phase;
            if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
            __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);  // agent scope by default: the other workgroups' stores before their flags are visible
    }
    __syncthreads();
}

// The same barrier as a two-level tree (round 6; the round-5 review measured the flat form at ~8 us against the 4-6 us
// MI355X_MICROARCH.md gives for an XCD-hierarchical barrier): in the flat form every workgroup polls every flag, i.e. n_wgs pollers
// hammer the same few lines at the memory side, where same-line requests are served one after the other.  Here workgroup w belongs
// to group w % 8 (the XCD it runs on under the round-robin dispatch -- a speed matter only, any grouping is correct); its members
// publish their phase to the group's flag run, the group's leader (w < 8) waits for them, publishes the GROUP's flag, waits for the
// eight group flags and releases its members through the group's generation word.  Pollers per line: the leader on its members' run,
// eight leaders on the group flags, a group's members on their own generation word.  Still no read-modify-write anywhere.
__device__ __forceinline__ void grid_barrier_xcd(unsigned* bar, unsigned n_wgs, unsigned& phase)
{
    __syncthreads();
    ++phase;
    if (threadIdx.x < 64)
    {
        const unsigned lane = threadIdx.x, x = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const unsigned n_x = (n_wgs + 7u - x) >> 3;  // members of group x (workgroups x, x + 8, ...)
        const unsigned groups = n_wgs < 8u ? n_wgs : 8u;
        unsigned* flags = bar + x * BAR_PER_XCD;
        unsigned* gen   = bar + BAR_GEN + 32u * x;
        if (j != 0)
        {
            if (lane == 0)
            {
                __hip_atomic_store(flags + j, phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) __builtin_amdgcn_s_sleep(1);
            }
        }
        else
        {
            for (;;)  // the members of this group
            {
                bool ok = true;
                for (unsigned i = 1u + lane; i < n_x; i += 64) ok = ok && __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= phase;
                if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
                __builtin_amdgcn_s_sleep(1);
            }
            // (the members' data was written back by THEIR release stores before their flags became visible; this release covers the
            // leader's own)
            if (lane == 0) __hip_atomic_store(bar + BAR_XFLAG + x, phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            for (;;)  // the eight groups
            {
                const bool ok = lane >= groups || __hip_atomic_load(bar + BAR_XFLAG + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= phase;
                if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0 && n_x > 1u) __hip_atomic_store(gen, phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);  // agent scope: every workgroup's stores before its arrival are visible
    }
    __syncthreads();
}

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned n_wgs, unsigned& phase, int flat)
{
    if (flat)  // grid-uniform (SNK_BA_FLAT_BARRIER=1: A/B against the round-5 form)
        grid_barrier_flat(bar, n_wgs, phase);
    else
        grid_barrier_xcd(bar, n_wgs, phase);
}

// sum of n partials in a fixed order by one wavefront-sized group of threads: lane l adds entries l, l + 64, ... then a butterfly
__device__ __forceinline__ double sum_partials_wave(const double* a, int n, int lane)
{
    double t = 0.0;
    for (int i = lane; i < n; i += 64) t += a[i];
    return wave_sum64(t);
}

constexpr int PERSIST_THREADS = 256;
constexpr int PERSIST_CAMS    = PERSIST_THREADS / 6;  // 42 cameras (252 elements) per workgroup in the update phase

__global__ __launch_bounds__(PERSIST_THREADS) void pcgl_persist(Arrays A, Opt O, PcgLarge W)
{
    extern __shared__ __attribute__((aligned(16))) double sh_p[];  // the direction p of this iteration, all n6 entries
    __shared__ double sh_r[PERSIST_THREADS], sh_red[2][PERSIST_THREADS / 64];
    const Prob pr  = A.prob[0];
    const int n6   = pr.n6, nfc = pr.nfc;
    if (n6 == 0) return;  // every workgroup
    const int tid  = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NW   = gridDim.x;
    const int wg   = blockIdx.x;
    const double* S = A.S + pr.s_off;
    const int rpw   = (n6 + NW - 1) / NW;                            // rows of S (= entries of A p) per workgroup
    const int q0    = min(wg * rpw, n6), q1 = min(q0 + rpw, n6);
    const int n_upd = (nfc + PERSIST_CAMS - 1) / PERSIST_CAMS;       // workgroups of the update phase (= partial r.r / r.z entries)
    unsigned phase = 0;
    double stop2   = 0.0;
    // parity 0 partials of r.r / r.z come from pcgl_init (camera groups of 64): G entries
    int n_prev = (nfc + 63) / 64;
    const double* prr_prev = W.prr;  // [B = 1][G], buffer 0
    const double* prz_prev = W.prz;
    double rz_prev = 0.0, rz_cur = 0.0;
    int iters = 0;
    for (int k = 0; k < O.max_pcg; ++k)
    {
        const double* p_prev = (k & 1) ? W.p2 : W.p;  // the direction of iteration k - 1 (k = 0: z, written by pcgl_init into p)
        double* p_cur        = (k & 1) ? W.p : W.p2;
        // ---- scalars of this iteration: |r|^2, r.z (every thread computes the same values) ----
        const double rn2 = sum_partials_wave(prr_prev, n_prev, lane);
        rz_cur           = sum_partials_wave(prz_prev, n_prev, lane);
        if (k == 0) stop2 = O.pcg_tol * O.pcg_tol * rn2;
        if (rn2 <= stop2) break;  // grid-uniform
        const double beta = k == 0 ? 0.0 : rz_cur / rz_prev;
        // ---- 1. A p for the workgroup's rows.  S is symmetric: row q is contiguous, a wavefront streams it (16 bytes per lane) against
        // the direction in LDS, formed on the fly as z + beta p_prev; the rows' p entries are stored for the update / the next iteration ----
        for (int u = tid; u < n6; u += PERSIST_THREADS) sh_p[u] = W.z[u] + beta * p_prev[u];
        __syncthreads();
        {
            double pap = 0.0;
            const int n2 = n6 >> 1;  // n6 = 6 * cameras: even, rows start 16-byte aligned (one problem: s_off = 0)
            const double2* sp2 = reinterpret_cast<const double2*>(sh_p);
            for (int q = q0 + wave; q < q1; q += PERSIST_THREADS / 64)
            {
                const double2* row = reinterpret_cast<const double2*>(S + (size_t)q * n6);
                double acc = 0.0;
                int u = lane;
                for (; u + 192 < n2; u += 256)  // four loads in flight
                {
                    const double2 s0 = row[u], s1 = row[u + 64], s2 = row[u + 128], s3 = row[u + 192];
                    const double2 p0 = sp2[u], p1 = sp2[u + 64], p2 = sp2[u + 128], p3 = sp2[u + 192];
                    acc += s0.x * p0.x; acc += s0.y * p0.y;
                    acc += s1.x * p1.x; acc += s1.y * p1.y;
                    acc += s2.x * p2.x; acc += s2.y * p2.y;
                    acc += s3.x * p3.x; acc += s3.y * p3.y;
                }
                for (; u < n2; u += 64)
                {
                    const double2 s0 = row[u], p0 = sp2[u];
                    acc += s0.x * p0.x; acc += s0.y * p0.y;
                }
                acc = wave_sum64(acc);
                if (lane == 0)
                {
                    const double pq = sh_p[q];
                    W.Ap[q]  = acc;
                    p_cur[q] = pq;
                    pap += pq * acc;
                }
            }
            if (lane == 0) sh_red[0][wave] = pap;
            __syncthreads();
            if (tid == 0) W.wpap[wg] = (sh_red[0][0] + sh_red[0][1]) + (sh_red[0][2] + sh_red[0][3]);
        }
        grid_barrier(W.bar, NW, phase, W.bar_flat);
        // ---- 2. update ----
        const double pAp = sum_partials_wave(W.wpap, NW, lane);
        if (pAp <= 0.0) break;  // grid-uniform (the reference's break: the step of this iteration is not applied)
        const double alpha = rz_cur / pAp;
        double* wrr = W.wrr + (size_t)((k + 1) & 1) * PERSIST_WGS_MAX;
        double* wrz = W.wrz + (size_t)((k + 1) & 1) * PERSIST_WGS_MAX;
        for (int g = wg; g < n_upd; g += NW)  // normally one pass
        {
            const int cl = tid / 6, a = tid - cl * 6;  // camera inside the group, row
            const int c  = g * PERSIST_CAMS + cl;
            const bool on = cl < PERSIST_CAMS && c < nfc;
            const int q  = c * 6 + a;
            double rv    = 0.0;
            if (on)
            {
                A.x[q] += alpha * p_cur[q];
                rv     = W.r[q] - alpha * W.Ap[q];
                W.r[q] = rv;
            }
            sh_r[tid] = rv;
            __syncthreads();
            double zz = 0.0;
            if (on)
            {
                const double* Mi = W.Minv + (size_t)(pr.cam_off + c) * 36 + a * 6;
                for (int b = 0; b < 6; ++b) zz += Mi[b] * sh_r[cl * 6 + b];
                W.z[q] = zz;
            }
            const double rr = wave_sum64(rv * rv), rzn = wave_sum64(rv * zz);
            if (lane == 0) sh_red[0][wave] = rr, sh_red[1][wave] = rzn;
            __syncthreads();
            if (tid == 0)
            {
                wrr[g] = (sh_red[0][0] + sh_red[0][1]) + (sh_red[0][2] + sh_red[0][3]);
                wrz[g] = (sh_red[1][0] + sh_red[1][1]) + (sh_red[1][2] + sh_red[1][3]);
            }
            __syncthreads();
        }
        grid_barrier(W.bar, NW, phase, W.bar_flat);
        prr_prev = wrr;
        prz_prev = wrz;
        n_prev   = n_upd;
        rz_prev  = rz_cur;
        ++iters;
    }
    if (wg == 0 && tid == 0) A.state[0].pcg_iters += iters;
}

// sum over the 64 lanes on the vector ALU only (permlane swaps + DPP moves; every lane gets the same bits): the __shfl_xor butterfly goes
// through the LDS crossbar, ~100 cycles per dependent step
__device__ __forceinline__ double wave_sum64_valu(double v)
{
    v = swap_add32(v, v);
    v = swap_add16(v, v);
    v += dpp_mov64<0x128>(v);  // row_ror:8
    v += dpp_mov64<0x141>(v);  // row_half_mirror
    v += dpp_mov64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov64<0xB1>(v);   // quad_perm [1,0,3,2]
    return v;
}

// The same PCG with ONE grid barrier per iteration (round 6).  Of the two reductions of an iteration only p.Ap needs every workgroup's
// rows; r.r and r.z of the next iteration are functions of r - alpha A p, which every workgroup can form for ALL n6 entries itself once
// A p is known: 6 x 6 blocks against 299 cameras are ~11 k multiply-adds, a microsecond, against the ~6 us of a barrier.  So every
// workgroup keeps private copies of r, z and p in LDS (identical in all workgroups: same inputs, same instruction sequence -- the
// grid-uniform decisions stay grid-uniform), multiplies its rows of S, publishes A p and its share of p.Ap, waits at the ONE barrier,
// and then updates r, z, r.r, r.z for the whole vector redundantly.  A p and the p.Ap partials alternate between two buffers: a
// workgroup that is one iteration ahead (it cannot be two: the barrier) writes the other parity.  Same arithmetic per element as the
// two-barrier form; the sums r.r / r.z run in this kernel's own fixed order (256 strided partials, four wavefronts).
// LDS: 3 n6 doubles (n6 <= 6400: ~1060 free cameras; beyond that pcgl_persist).
__global__ __launch_bounds__(PERSIST_THREADS) void pcgl_persist1(Arrays A, Opt O, PcgLarge W)
{
    extern __shared__ __attribute__((aligned(16))) double sh_all[];
    __shared__ double sh_red[2][PERSIST_THREADS / 64];
    const Prob pr  = A.prob[0];
    const int n6   = pr.n6;
    if (n6 == 0) return;  // every workgroup
    double* sh_p = sh_all;             // the direction
    double* sh_r = sh_all + n6;        // the residual
    double* sh_z = sh_all + 2 * n6;    // the preconditioned residual
    const int tid  = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NW   = gridDim.x;
    const int wg   = blockIdx.x;
    const double* S = A.S + pr.s_off;
    const int rpw   = (n6 + NW - 1) / NW;
    const int q0    = min(wg * rpw, n6), q1 = min(q0 + rpw, n6);
    unsigned phase = 0;
    // The six preconditioner values of the thread's own entries u = tid + 256 j in registers for the launch (up to 4096 unknowns; round 6,
    // late: 42-84 L2 loads per thread and iteration before, beside a workgroup that is alone on its compute unit and has 512 registers
    // per lane); bigger systems read them from memory as before.
    constexpr int MK = 16;
    const bool m_regs = n6 <= MK * PERSIST_THREADS;  // grid-uniform
    double mrow[MK][6];
#pragma unroll
    for (int j = 0; j < MK; ++j)
    {
        const int u = tid + PERSIST_THREADS * j;
#pragma unroll
        for (int b = 0; b < 6; ++b) mrow[j][b] = m_regs && u < n6 ? W.Minv[(size_t)pr.cam_off * 36 + (size_t)u * 6 + b] : 0.0;
    }
    // block-wide sums of two values per thread, every thread gets both (fixed order: wavefront trees, then the four wavefronts)
    auto block_sum2 = [&](double a, double b, double& sa, double& sb)
    {
        a = wave_sum64_valu(a);
        b = wave_sum64_valu(b);
        __syncthreads();  // the previous round's readers are done
        if (lane == 0) sh_red[0][wave] = a, sh_red[1][wave] = b;
        __syncthreads();
        sa = (sh_red[0][0] + sh_red[0][1]) + (sh_red[0][2] + sh_red[0][3]);
        sb = (sh_red[1][0] + sh_red[1][1]) + (sh_red[1][2] + sh_red[1][3]);
    };
    // r, z from pcgl_init; |r|^2 and r.z in this kernel's order
    double rn2, rz_cur;
    {
        double a = 0.0, b = 0.0;
        for (int u = tid; u < n6; u += PERSIST_THREADS)
        {
            const double rv = W.r[u], zv = W.z[u];
            sh_r[u] = rv;
            sh_z[u] = zv;
            sh_p[u] = 0.0;
            a += rv * rv;
            b += rv * zv;
        }
        block_sum2(a, b, rn2, rz_cur);
    }
    const double stop2 = O.pcg_tol * O.pcg_tol * rn2;
    double rz_prev = 0.0;
    int iters = 0;
    for (int k = 0; k < O.max_pcg; ++k)
    {
        if (rn2 <= stop2) break;  // grid-uniform
        const double beta = k == 0 ? 0.0 : rz_cur / rz_prev;
        double* Ap   = (k & 1) ? W.p2 : W.Ap;                                   // the two A p buffers (p2 is free in this form)
        double* wpap = W.wrr + (size_t)(k & 1) * PERSIST_WGS_MAX;               // ... and the two partial-sum buffers
        for (int u = tid; u < n6; u += PERSIST_THREADS) sh_p[u] = sh_z[u] + beta * sh_p[u];
        __syncthreads();
        {
            double pap = 0.0;
            const int n2 = n6 >> 1;
            const double2* sp2 = reinterpret_cast<const double2*>(sh_p);
            // Two rows of the wavefront at a time, eight 16-byte loads per row in flight (sixteen per lane): the workgroup is alone on its CU
            // (one wavefront per SIMD), S comes out of the Infinity Cache at ~1 us per round trip, and with four loads in flight a
            // wavefront's three rows were a dozen dependent round trips of a 16 us iteration.  Every row is accumulated in the same order
            // as before (ascending u in steps of 64 per lane): bit-identical sums.
            constexpr int WV = PERSIST_THREADS / 64;
            auto row_dot = [&](const double2* __restrict__ ra, const double2* __restrict__ rb, double& acc_a, double& acc_b)
            {
                int u = lane;
                for (; u + 448 < n2; u += 512)
                {
                    double2 sa[8], sb[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) sa[k] = ra[u + 64 * k], sb[k] = rb[u + 64 * k];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                    {
                        const double2 pv = sp2[u + 64 * k];
                        acc_a += sa[k].x * pv.x; acc_a += sa[k].y * pv.y;
                        acc_b += sb[k].x * pv.x; acc_b += sb[k].y * pv.y;
                    }
                }
                for (; u + 192 < n2; u += 256)
                {
                    double2 sa[4], sb[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) sa[k] = ra[u + 64 * k], sb[k] = rb[u + 64 * k];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                    {
                        const double2 pv = sp2[u + 64 * k];
                        acc_a += sa[k].x * pv.x; acc_a += sa[k].y * pv.y;
                        acc_b += sb[k].x * pv.x; acc_b += sb[k].y * pv.y;
                    }
                }
                for (; u < n2; u += 64)
                {
                    const double2 s0 = ra[u], s1 = rb[u], pv = sp2[u];
                    acc_a += s0.x * pv.x; acc_a += s0.y * pv.y;
                    acc_b += s1.x * pv.x; acc_b += s1.y * pv.y;
                }
            };
            for (int q = q0 + wave; q < q1; q += 2 * WV)
            {
                const int qb    = q + WV;
                const bool two  = qb < q1;  // wave-uniform
                const double2* ra = reinterpret_cast<const double2*>(S + (size_t)q * n6);
                const double2* rb = reinterpret_cast<const double2*>(S + (size_t)(two ? qb : q) * n6);
                double acc_a = 0.0, acc_b = 0.0;
                row_dot(ra, rb, acc_a, acc_b);
                acc_a = wave_sum64_valu(acc_a);
                acc_b = wave_sum64_valu(acc_b);
                if (lane == 0)
                {
                    Ap[q] = acc_a;
                    pap += sh_p[q] * acc_a;
                    if (two)
                    {
                        Ap[qb] = acc_b;
                        pap += sh_p[qb] * acc_b;
                    }
                }
            }
            if (lane == 0) sh_red[0][wave] = pap;
            __syncthreads();
            if (tid == 0) wpap[wg] = (sh_red[0][0] + sh_red[0][1]) + (sh_red[0][2] + sh_red[0][3]);
        }
        grid_barrier(W.bar, NW, phase, W.bar_flat);
        // A p of the other workgroups comes from the memory side (~2 us): the first 16 entries per thread are requested BEFORE p.Ap is
        // summed, whose own loads and the break below would otherwise stand in front of them as a second round trip
        double apv[MK];
#pragma unroll
        for (int j = 0; j < MK; ++j)
        {
            const int u = tid + PERSIST_THREADS * j;
            apv[j]      = u < n6 ? Ap[u] : 0.0;
        }
        double pAp = 0.0;
        for (int i = lane; i < NW; i += 64) pAp += wpap[i];
        pAp = wave_sum64_valu(pAp);
        if (pAp <= 0.0) break;  // grid-uniform (the reference's break: the step of this iteration is not applied)
        const double alpha = rz_cur / pAp;
#pragma unroll
        for (int j = 0; j < MK; ++j)
        {
            const int u = tid + PERSIST_THREADS * j;
            if (u < n6) sh_r[u] -= alpha * apv[j];
        }
        for (int u = tid + MK * PERSIST_THREADS; u < n6; u += PERSIST_THREADS) sh_r[u] -= alpha * Ap[u];
        for (int q = q0 + tid; q < q1; q += PERSIST_THREADS) A.x[q] += alpha * sh_p[q];  // the rows' owner keeps the solution
        __syncthreads();
        double a = 0.0, b = 0.0;
        if (m_regs)
        {
            // all reads of r in front of the stores of z (one LDS array: the compiler keeps their order entry by entry otherwise); the
            // same six products in the same order as the loop below
            double zz[MK];
#pragma unroll
            for (int j = 0; j < MK; ++j)
            {
                const int u = min(tid + PERSIST_THREADS * j, n6 - 1);
                const double2* rc = reinterpret_cast<const double2*>(sh_r + (u / 6) * 6);  // the camera's six entries: 48 bytes, 16-byte aligned
                const double2 r01 = rc[0], r23 = rc[1], r45 = rc[2];
                double t = 0.0;
                t += mrow[j][0] * r01.x; t += mrow[j][1] * r01.y; t += mrow[j][2] * r23.x;
                t += mrow[j][3] * r23.y; t += mrow[j][4] * r45.x; t += mrow[j][5] * r45.y;
                zz[j] = t;
            }
#pragma unroll
            for (int j = 0; j < MK; ++j)
            {
                const int u = tid + PERSIST_THREADS * j;
                if (u < n6)
                {
                    const double rv = sh_r[u];
                    sh_z[u] = zz[j];
                    a += rv * rv;
                    b += rv * zz[j];
                }
            }
        }
        else
        for (int u = tid; u < n6; u += PERSIST_THREADS)
        {
            const int c = u / 6, ar = u - c * 6;
            const double* Mi = W.Minv + (size_t)(pr.cam_off + c) * 36 + ar * 6;
            const double* rc = sh_r + c * 6;
            double zz = 0.0;
            for (int bq = 0; bq < 6; ++bq) zz += Mi[bq] * rc[bq];
            sh_z[u] = zz;
            const double rv = sh_r[u];
            a += rv * rv;
            b += rv * zz;
        }
        rz_prev = rz_cur;
        block_sum2(a, b, rn2, rz_cur);
        ++iters;
    }
    if (wg == 0 && tid == 0) A.state[0].pcg_iters += iters;
}

// The one-barrier PCG with the workgroup's rows of S held in REGISTERS for the whole launch (round 6, late).  A launch runs up to
// max_pcg (40) iterations on the same S: pcgl_persist1 streamed its dozen rows per workgroup out of the Infinity Cache in every one of
// them (~2.7 us of a 15 us iteration, 25.9 MB per iteration for 300 keyframes).  A workgroup is alone on its compute unit (one
// wavefront per SIMD: 512 registers per lane, arch + acc), so R rows x 8 columns per thread (column u = tid + 256 k) fit the register
// file for n6 <= 2048 (~340 free cameras) with R = 8 (128 registers) or 16 (256: half of them parked in the acc half, a move per use);
// S is then read ONCE per launch.  A thread owns the entries u = tid + 256 j of r, z and p -- registers too, with the six
// preconditioner values of each (42 L2 loads per thread and iteration before) -- so it multiplies its columns against the entries of
// p it has just formed itself: no LDS read, no barrier in front of the product.  The R row sums are reduced by a transpose-and-add
// over the wavefront (vector ALU only) and a four-way sum through LDS; LDS otherwise holds the copies of p and r other threads read.
// Same algorithm, same grid barrier and buffers as pcgl_persist1; the sums run in this kernel's own fixed order.  16 rows per
// workgroup above 512 unknowns: barrier and exchange grow with the workgroups (113 instead of 225 for 300 keyframes: 5.2 instead of
// 7.8 us of an iteration).  Bigger systems keep pcgl_persist1 (SNK_BA_PERSIST_STREAM=1: always).
// Measured per iteration, 300 keyframes (SNK_BA_PCG_TIMING=1): product + row sums 1.2, barrier 3.4, A p exchange 1.8, preconditioner +
// r.r / r.z 0.7 us = 7.1 (pcgl_persist1: 15.3).  Also built and dropped: no barrier at all, exchange buffers whose entries validate
// themselves (a NaN pattern = "not written yet", three buffers in turn) -- every workgroup then polls 15 KB at the memory side and
// the wait costs what barrier + exchange cost (profiles/NOTES.md).
constexpr int PREG_MAX_N6 = 2048;
template <int R>
__global__ __launch_bounds__(PERSIST_THREADS) void pcgl_persist_reg(Arrays A, Opt O, PcgLarge W)
{
    constexpr int NT = PERSIST_THREADS, KC = PREG_MAX_N6 / NT, NWV = NT / 64;  // columns (and own entries) per thread, wavefronts
    static_assert(R == 8 || R == 16, "rows per workgroup");
    extern __shared__ __attribute__((aligned(16))) double sh_all[];
    __shared__ double sh_red[2][NWV], sh_acc[NWV][R];
    const Prob pr  = A.prob[0];
    const int n6   = pr.n6;
    if (n6 == 0) return;  // every workgroup
    double* sh_p = sh_all;             // the direction (read by the owners of other rows: p.Ap, x)
    double* sh_r = sh_all + n6;        // the residual (read by the six threads of a camera: z = Minv r)
    const int tid  = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NW   = gridDim.x;
    const int wg   = blockIdx.x;
    const int q0   = min(wg * R, n6), nrow = min(R, n6 - q0);
    unsigned phase = 0;
    const bool timing = W.timing != 0 && wg == 0 && tid == 0;
    unsigned long long tc[5] = {0, 0, 0, 0, 0}, t_at = timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
    auto stamp = [&](int slot)
    {
        if (!timing) return;
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        tc[slot] += now - t_at;
        t_at = now;
    };
    double s[R][KC];
    {
        const double* S = A.S + pr.s_off;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < KC; ++k)
            {
                const int u = tid + NT * k;
                s[r][k]     = r < nrow && u < n6 ? S[(size_t)(q0 + r) * n6 + u] : 0.0;
            }
    }
    // ... and so do the preconditioner rows of the thread's own entries (6 doubles each; 42 L2 loads per thread and iteration before)
    double mrow[KC][6];
#pragma unroll
    for (int j = 0; j < KC; ++j)
    {
        const int u = tid + NT * j;
#pragma unroll
        for (int b = 0; b < 6; ++b) mrow[j][b] = u < n6 ? W.Minv[(size_t)pr.cam_off * 36 + (size_t)u * 6 + b] : 0.0;  // 36 c + 6 a + b = 6 u + b
    }
    // block-wide sums of two values per thread, every thread gets both (fixed order: wavefront trees, then the four wavefronts)
    auto block_sum2 = [&](double a, double b, double& sa, double& sb)
    {
        a = wave_sum64_valu(a);
        b = wave_sum64_valu(b);
        __syncthreads();  // the previous round's readers are done
        if (lane == 0) sh_red[0][wave] = a, sh_red[1][wave] = b;
        __syncthreads();
        sa = sh_red[0][0], sb = sh_red[1][0];
#pragma unroll
        for (int w = 1; w < NWV; ++w) sa += sh_red[0][w], sb += sh_red[1][w];
    };
    // The thread's own entries u = tid + 256 j of r, z and p stay in registers from iteration to iteration (it is the only writer of
    // them and the only reader of z); LDS holds the copies other threads need.  All loads of a phase are issued before its stores:
    // reads and writes of the one LDS array in the same loop body would be kept in program order by the compiler, entry by entry.
    double rv[KC], zv[KC], pv[KC];
    double rn2, rz_cur;
    {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int j = 0; j < KC; ++j)
        {
            const int u = tid + NT * j;
            rv[j] = u < n6 ? W.r[u] : 0.0;
            zv[j] = u < n6 ? W.z[u] : 0.0;
            pv[j] = 0.0;
            a += rv[j] * rv[j];
            b += rv[j] * zv[j];
        }
        block_sum2(a, b, rn2, rz_cur);
    }
    const double stop2 = O.pcg_tol * O.pcg_tol * rn2;
    double rz_prev = 0.0;
    int iters = 0;
    stamp(0);  // set-up: S into registers, the preconditioner into LDS, r / z
    for (int k = 0; k < O.max_pcg; ++k)
    {
        if (rn2 <= stop2) break;  // grid-uniform
        const double beta = k == 0 ? 0.0 : rz_cur / rz_prev;
        double* Ap   = (k & 1) ? W.p2 : W.Ap;                                   // the two A p buffers (p2 is free in this form)
        double* wpap = W.wrr + (size_t)(k & 1) * PERSIST_WGS_MAX;               // ... and the two partial-sum buffers
#pragma unroll
        for (int j = 0; j < KC; ++j)
        {
            const int u = tid + NT * j;
            pv[j]       = zv[j] + beta * pv[j];
            if (u < n6) sh_p[u] = pv[j];
        }
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
        {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < KC; ++j) t += s[r][j] * pv[j];
            acc[r] = t;
        }
        // The row sums over the wavefront as a reduce-scatter (cam_pass's "transpose and add"): halve the rows a lane carries while
        // doubling the lanes a value covers (permlane32_swap, permlane16_swap, row_ror:8, ...), then DPP steps on the one row left --
        // 10 vector-ALU exchanges for 8 rows instead of the 48 LDS-crossbar round trips of eight butterflies (~100 cycles each,
        // dependent: 2 of an iteration's 13 us, SNK_BA_PCG_TIMING).  Row r ends up in the lanes whose upper bits are r.
        if constexpr (R == 8)
        {
            double w4[4], w2[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) w4[i] = swap_add32(acc[i], acc[i + 4]);
#pragma unroll
            for (int i = 0; i < 2; ++i) w2[i] = swap_add16(w4[i], w4[i + 2]);
            double w1 = xchg_add<0x128>(w2[0], w2[1], (lane & 8) != 0);  // row_ror:8
            w1 += dpp_mov64<0x141>(w1);  // row_half_mirror
            w1 += dpp_mov64<0x4E>(w1);   // quad_perm [2,3,0,1]
            w1 += dpp_mov64<0xB1>(w1);   // quad_perm [1,0,3,2]
            if ((lane & 7) == 0) sh_acc[wave][lane >> 3] = w1;
        }
        else
        {
            double w8[8], w4[4], w2[2];
#pragma unroll
            for (int i = 0; i < 8; ++i) w8[i] = swap_add32(acc[i], acc[i + 8]);
#pragma unroll
            for (int i = 0; i < 4; ++i) w4[i] = swap_add16(w8[i], w8[i + 4]);
#pragma unroll
            for (int i = 0; i < 2; ++i) w2[i] = xchg_add<0x128>(w4[i], w4[i + 2], (lane & 8) != 0);  // row_ror:8
            double w1 = xchg_add<0x141>(w2[0], w2[1], (lane & 4) != 0);                           // row_half_mirror: l <-> 7 - l
            w1 += dpp_mov64<0x4E>(w1);   // quad_perm [2,3,0,1]
            w1 += dpp_mov64<0xB1>(w1);   // quad_perm [1,0,3,2]
            if ((lane & 3) == 0) sh_acc[wave][lane >> 2] = w1;  // row r = bits 5..2 of the lane
        }
        __syncthreads();
        if (wave == 0)
        {
            double pap = 0.0, v = 0.0;
            if (lane < nrow)
            {
                v = sh_acc[0][lane];
#pragma unroll
                for (int w = 1; w < NWV; ++w) v += sh_acc[w][lane];
                pap = sh_p[q0 + lane] * v;
            }
            pap = wave_sum64_valu(pap);
            if (lane < nrow) Ap[q0 + lane] = v;
            if (lane == 0) wpap[wg] = pap;
        }
        stamp(1);  // direction, product, row sums
        double apv[KC];
        double pAp = 0.0;
        grid_barrier(W.bar, NW, phase, W.bar_flat);
        stamp(2);  // barrier
        // A p of the other workgroups comes from the memory side (~2 us): requested BEFORE p.Ap is summed, whose own loads and the
        // break below would otherwise stand in front of it as a second round trip
#pragma unroll
        for (int j = 0; j < KC; ++j)
        {
            const int u = tid + NT * j;
            apv[j]      = u < n6 ? Ap[u] : 0.0;
        }
        for (int i = lane; i < NW; i += 64) pAp += wpap[i];
        pAp = wave_sum64_valu(pAp);
        if (pAp <= 0.0) break;  // grid-uniform (the reference's break: the step of this iteration is not applied)
        const double alpha = rz_cur / pAp;
        if (tid < nrow) A.x[q0 + tid] += alpha * sh_p[q0 + tid];  // the rows' owner keeps the solution
#pragma unroll
        for (int j = 0; j < KC; ++j)
        {
            const int u = tid + NT * j;
            rv[j] -= alpha * apv[j];
            if (u < n6) sh_r[u] = rv[j];
        }
        __syncthreads();
        stamp(3);  // p.Ap from the partial sums, r -= alpha A p (A p from the other workgroups)
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int j = 0; j < KC; ++j)
        {
            const int u = min(tid + NT * j, n6 - 1);
            const double2* rc = reinterpret_cast<const double2*>(sh_r + (u / 6) * 6);  // the camera's six entries: 48 bytes, 16-byte aligned
            const double2 r01 = rc[0], r23 = rc[1], r45 = rc[2];
            zv[j] = ((mrow[j][0] * r01.x + mrow[j][1] * r01.y) + (mrow[j][2] * r23.x + mrow[j][3] * r23.y)) + (mrow[j][4] * r45.x + mrow[j][5] * r45.y);
            a += rv[j] * rv[j];
            b += rv[j] * zv[j];
        }
        rz_prev = rz_cur;
        block_sum2(a, b, rn2, rz_cur);
        stamp(4);  // z = Minv r, r.r, r.z
        ++iters;
    }
    if (timing)
    {
        for (int i = 0; i < 5; ++i) W.ps[i] += (double)tc[i];
        W.ps[5] += (double)iters;
    }
    if (wg == 0 && tid == 0) A.state[0].pcg_iters += iters;
}

static inline const void* persist_reg_kernel(int rows)
{
    return rows == 16 ? reinterpret_cast<const void*>(pcgl_persist_reg<16>) : reinterpret_cast<const void*>(pcgl_persist_reg<8>);
}

// pose <- exp(delta) * pose
__device__ void se3_update(const double* pose, const double* d, double* out)
{
    const double wx = d[3], wy = d[4], wz = d[5];
    const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
    double B, Cc, qd[4];
    if (th < 1e-8)
    {
        B  = 0.5 - th2 / 24.0;
        Cc = 1.0 / 6.0 - th2 / 120.0;
        const double h = 0.5 - th2 / 48.0;
        qd[0] = h * wx; qd[1] = h * wy; qd[2] = h * wz; qd[3] = 1.0 - th2 / 8.0;
    }
    else
    {
        const double s = sin(th), c = cos(th);
        B  = (1.0 - c) / th2;
        Cc = (th - s) / (th2 * th);
        const double sh = sin(0.5 * th) / th;
        qd[0] = sh * wx; qd[1] = sh * wy; qd[2] = sh * wz; qd[3] = cos(0.5 * th);
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
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    out[0] = q[0] / n; out[1] = q[1] / n; out[2] = q[2] / n; out[3] = q[3] / n;
    out[4] = Rd[0] * tx + Rd[1] * ty + Rd[2] * tz + tdx;
    out[5] = Rd[3] * tx + Rd[4] * ty + Rd[5] * tz + tdy;
    out[6] = Rd[6] * tx + Rd[7] * ty + Rd[8] * tz + tdz;
}

// the trial pose of image i: exp(dc) * pose for a free camera, the pose itself for a constant one
__device__ inline void trial_pose(const Arrays& A, const Prob& pr, int i)
{
    const int gi = pr.img_off + i;
    const int c  = A.cam_idx[gi];
    const double* cur = A.pose + (size_t)gi * 7;
    double* out       = A.pose_new + (size_t)gi * 7;
    if (c < 0)
        for (int k = 0; k < 7; ++k) out[k] = cur[k];
    else
        se3_update(cur, A.x + pr.vec_off + c * 6, out);
}

// update_wave's workgroups behind the point work items (blockIdx.x >= wave_blocks): the trial poses, 256 images each
__device__ inline void trial_poses_block(const Arrays& A, int pb, int block)
{
    const Prob pr = A.prob[pb];
    const int i   = block * 256 + threadIdx.x;
    if (i < pr.ni) trial_pose(A, pr, i);
}

// threads [0, np): back-substitution dp = V^-1 (b_p - sum W^T dc); threads [np, np + ni): trial poses
__global__ __launch_bounds__(128) void update_pass(Arrays A, int images_only)
{
    const int pb  = blockIdx.y;
    const Prob pr = A.prob[pb];
    const int t   = blockIdx.x * 128 + threadIdx.x + (images_only ? pr.np : 0);  // update_wave does the points
    const double* x = A.x + pr.vec_off;
    if (t < pr.np)
    {
        const int gp = pr.pt_off + t;
        double* out  = A.pt_new + (size_t)gp * 3;
        const double* cur = A.pt + (size_t)gp * 3;
        if (A.pt_const[gp])
        {
            out[0] = cur[0];
            out[1] = cur[1];
            out[2] = cur[2];
            return;
        }
        double g[3] = {A.bp[(size_t)gp * 3], A.bp[(size_t)gp * 3 + 1], A.bp[(size_t)gp * 3 + 2]};
        const int s0 = A.pt_start[pr.ptstart_off + t], s1 = A.pt_start[pr.ptstart_off + t + 1];
        for (int s = s0; s < s1; ++s)
        {
            const int go = pr.obs_off + s;
            const int c  = A.o_cam[go];
            if (c < 0 || A.o_r[(size_t)go * 4 + 3] == 0.0) continue;
            const double* Wp = A.o_W + (size_t)go * 18;
            const double* xc = x + c * 6;
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int a = 0; a < 6; ++a) g[b] -= Wp[a * 3 + b] * xc[a];
        }
        const double* Vi = A.Vinv + (size_t)gp * 6;
        out[0] = cur[0] + (Vi[0] * g[0] + Vi[1] * g[1] + Vi[2] * g[2]);
        out[1] = cur[1] + (Vi[1] * g[0] + Vi[3] * g[1] + Vi[4] * g[2]);
        out[2] = cur[2] + (Vi[2] * g[0] + Vi[4] * g[1] + Vi[5] * g[2]);
    }
    else if (t < pr.np + pr.ni)
        trial_pose(A, pr, t - pr.np);
}

constexpr int ACC_THREADS = 256;
constexpr int ACC_TILE    = ACC_THREADS * 14;  // entries of each cost array per LDS tile of accept_pass's tiled sums (2 x 28 KB: under the 64 KB a launch gets without asking)
__global__ __launch_bounds__(ACC_THREADS) void accept_pass(Arrays A, int only_marked, int no_copy, int tiled /* launched with 2 * ACC_TILE doubles of LDS */)
{
    __shared__ double red[ACC_THREADS];
    __shared__ int s_acc;
    const int pb  = blockIdx.x;
    if (only_marked && A.state[pb].marked == 0) return;
    const Prob pr = A.prob[pb];
    const int tid = threadIdx.x;
    // fixed-order sums: contiguous chunk per thread, then the tree
    const int chunk = (pr.np + ACC_THREADS - 1) / ACC_THREADS;
    double c0 = 0.0, c1 = 0.0;
    int k_from = 0;
    if (tiled)
    {
        // Big problems (a global scene: 15 000 points, 59 per thread).  A thread's chunk is contiguous, so the direct loads below are
        // strided across the lanes -- 64 cache lines per load instruction, and all of it on ONE compute unit: 28 us of sums.  Here the
        // workgroup loads both arrays tile by tile, coalesced (the next tile's loads in flight meanwhile), and every thread adds the
        // entries of its chunk that lie in the tile out of LDS -- the same additions in the same order.
        extern __shared__ __attribute__((aligned(16))) double acc_tile[];
        double* ta = acc_tile, *tb = acc_tile + ACC_TILE;
        const int lo = tid * chunk, hi = min(lo + chunk, pr.np);
        double va[ACC_TILE / ACC_THREADS], vb[ACC_TILE / ACC_THREADS];
        auto fetch = [&](int t0)
        {
#pragma unroll
            for (int u = 0; u < ACC_TILE / ACC_THREADS; ++u)
            {
                const int p = t0 + u * ACC_THREADS + tid;
                va[u]       = p < pr.np ? A.cost_pt[pr.pt_off + p] : 0.0;
                vb[u]       = p < pr.np ? A.cost_pt_new[pr.pt_off + p] : 0.0;
            }
        };
        fetch(0);
        for (int t0 = 0; t0 < pr.np; t0 += ACC_TILE)
        {
            __syncthreads();  // the previous tile's readers are done
#pragma unroll
            for (int u = 0; u < ACC_TILE / ACC_THREADS; ++u) ta[u * ACC_THREADS + tid] = va[u], tb[u * ACC_THREADS + tid] = vb[u];
            __syncthreads();
            if (t0 + ACC_TILE < pr.np) fetch(t0 + ACC_TILE);
            for (int p = max(lo, t0); p < min(hi, t0 + ACC_TILE); ++p)
            {
                c0 += ta[p - t0];
                c1 += tb[p - t0];
            }
        }
        k_from = chunk;  // nothing left for the direct loop
    }
    for (int k0 = k_from; k0 < chunk; k0 += 4)  // four loads of each array in flight, added in index order (one at a time: `chunk` dependent round trips)
    {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
            const int p   = tid * chunk + k0 + u;
            const bool ok = k0 + u < chunk && p < pr.np;
            a[u]          = ok ? A.cost_pt[pr.pt_off + p] : 0.0;
            b[u]          = ok ? A.cost_pt_new[pr.pt_off + p] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
            c0 += a[u];
            c1 += b[u];
        }
    }
    double cost     = block_sum<ACC_THREADS>(c0, red, tid);
    double cost_new = block_sum<ACC_THREADS>(c1, red, tid);
    for (int k = 0; k < pr.n_rpc; ++k)  // few (<= one per keyframe pair), fixed order, same value in every thread
    {
        const double* o = A.rpc_out + (size_t)(pr.rpc_off + k) * RPC_STRIDE;
        cost += o[0];
        cost_new += o[1];
    }
    State& st = A.state[pb];
    if (tid == 0)
    {
        if (st.iter == 0) st.cost_initial = cost;
        const int acc = cost_new < cost ? 1 : 0;
        st.cost_new   = cost_new;
        st.accepted   = acc;
        if (acc)
        {
            st.cost   = cost_new;
            st.lambda = st.lambda * (1.0 / 3.0);
            st.vfac   = 2.0;
        }
        else
        {
            st.cost = cost;
            st.lambda *= st.vfac;
            st.vfac *= 2.0;
        }
        st.iter += 1;
        s_acc = acc;
    }
    __syncthreads();
    if (!s_acc || no_copy) return;  // no_copy: accept_copy follows (big single problems: the copy on many workgroups)
    {
        // eight loads in flight per thread (one at a time the 6000 doubles of a window's points were 24 dependent round trips)
        const double* src = A.pt_new + (size_t)pr.pt_off * 3;
        double* dst       = A.pt + (size_t)pr.pt_off * 3;
        const int n       = pr.np * 3;
        int k_lo          = tid;
        for (; k_lo + 31 * ACC_THREADS < n; k_lo += 32 * ACC_THREADS)  // big problems: 32 loads in flight (45 000 doubles: 6 rounds instead of 22)
        {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = src[k_lo + u * ACC_THREADS];
#pragma unroll
            for (int u = 0; u < 32; ++u) dst[k_lo + u * ACC_THREADS] = v[u];
        }
        for (int k0 = k_lo; k0 < n; k0 += 8 * ACC_THREADS)
        {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k0 + u * ACC_THREADS < n ? src[k0 + u * ACC_THREADS] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + u * ACC_THREADS < n) dst[k0 + u * ACC_THREADS] = v[u];
        }
    }
    for (int k = tid; k < pr.ni * 7; k += ACC_THREADS) A.pose[(size_t)pr.img_off * 7 + k] = A.pose_new[(size_t)pr.img_off * 7 + k];
}

// The accepted state of a BIG problem copied by many workgroups (a global scene: 15 000 points + 300 poses = 377 KB read and written;
// on accept_pass's one workgroup that was most of its 41 us -- one compute unit's worth of memory requests in flight).
__global__ __launch_bounds__(256) void accept_copy(Arrays A, int only_marked)
{
    const int pb = blockIdx.y;
    if (only_marked && A.state[pb].marked == 0) return;
    if (!A.state[pb].accepted) return;
    const Prob pr = A.prob[pb];
    const int n_pt = pr.np * 3, n = n_pt + pr.ni * 7;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256)
    {
        if (k < n_pt)
            A.pt[(size_t)pr.pt_off * 3 + k] = A.pt_new[(size_t)pr.pt_off * 3 + k];
        else
            A.pose[(size_t)pr.img_off * 7 + (k - n_pt)] = A.pose_new[(size_t)pr.img_off * 7 + (k - n_pt)];
    }
}

__global__ void begin_solve(State* st, int n, double lambda_init, int only_marked)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (only_marked && st[i].marked == 0) return;  // a conditional extra iteration (select_marked) leaves unmarked problems alone
    st[i].lambda    = lambda_init;
    st[i].vfac      = 2.0;
    st[i].iter      = 0;
    st[i].pcg_iters = 0;
    st[i].accepted  = 0;
    if (!only_marked) st[i].marked = 0;
}
}  // namespace
}  // namespace snk

using namespace snk;

// The lists a scene hand-over builds on the host live in PINNED vectors that belong to the handle and keep their capacity from call to
// call: hipMemcpyAsync from pageable memory stages and synchronises (33 uploads cost 0.23 ms per local-BA scene), from pinned memory
// it is an enqueue; and a new scene per keyframe no longer allocates and first-touches a megabyte of host memory.
template <typename T>
struct PinnedAlloc
{
    using value_type = T;
    PinnedAlloc() = default;
    template <typename U>
    PinnedAlloc(const PinnedAlloc<U>&) {}
    T* allocate(size_t n)
    {
        void* p = nullptr;
        if (hipHostMalloc(&p, n * sizeof(T), hipHostMallocDefault) != hipSuccess) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t) { (void)hipHostFree(p); }
    // resize() default-initialises (no zero fill): every list is written in full right after it is sized, and zeroing hundreds of
    // megabytes of pinned memory first was host time of a batch hand-over; resize(n, value) still fills
    template <typename U>
    void construct(U* p) noexcept
    {
        ::new (static_cast<void*>(p)) U;
    }
    template <typename U, typename... Args>
    void construct(U* p, Args&&... args)
    {
        ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
    }
    template <typename U>
    bool operator==(const PinnedAlloc<U>&) const { return true; }
    template <typename U>
    bool operator!=(const PinnedAlloc<U>&) const { return false; }
};
template <typename T>
using pvec = std::vector<T, PinnedAlloc<T>>;

struct BaLists
{
    pvec<Prob> probs;
    pvec<double> pose, pt, ouv2, odepth, oweight;
    pvec<unsigned char> ptc, optfree;
    pvec<SetItem> setitems;
    pvec<int2> setpts;
    pvec<int> setpairs, cblkstart, cblkitems, ccstart, ccitems;
    pvec<int> camidx, ptstart, oimg, ocam, oorig, camstart, camitems, blkstart, optidx, wvpt, rpcnext, camrpcstart, camrpcitems, blkrpc;
    pvec<RpcMeta> rpcmeta;
    pvec<int4> blkent;
    pvec<State> states;
    void clear()
    {
        probs.clear(), pose.clear(), pt.clear(), ouv2.clear(), odepth.clear(), oweight.clear(), ptc.clear(), optfree.clear();
        setitems.clear(), setpts.clear(), setpairs.clear(), cblkstart.clear(), cblkitems.clear();
        camidx.clear(), ptstart.clear(), oimg.clear(), ocam.clear(), oorig.clear(), camstart.clear(), camitems.clear();
        blkstart.clear(), optidx.clear(), wvpt.clear(), rpcnext.clear(), camrpcstart.clear(), camrpcitems.clear(), blkrpc.clear();
        rpcmeta.clear(), blkent.clear(), ccstart.clear(), ccitems.clear();
    }
};

// The host threads of a batch hand-over, kept by the handle (round 6, late).  snk_ba_set_problems runs seven threaded passes over the
// problems of a batch; with std::thread created and joined per pass that was 7 x 31 creations (~20 us each, issued one after the
// other) inside a 22 ms hand-over.  Workers park on a condition variable between passes and end with the handle.
constexpr int BA_FILL_CHUNKS = 4;   // chunks of problems the fill pass of a batch hand-over runs (and uploads) in
struct HostPool
{
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, cv_done;
    const std::function<void()>* job = nullptr;
    unsigned gen = 0;
    int n_run = 0, n_left = 0;
    bool stop = false;
    void worker(int idx)
    {
        unsigned seen = 0;
        for (;;)
        {
            const std::function<void()>* j = nullptr;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                if (idx < n_run) j = job;
            }
            if (j)
            {
                (*j)();  // the passes catch their own exceptions
                std::lock_guard<std::mutex> lk(m);
                if (--n_left == 0) cv_done.notify_one();
            }
        }
    }
    // runs `work` on up to `helpers` pool threads and on the caller; returns when all of them are done
    void run(int helpers, const std::function<void()>& work)
    {
        try
        {
            while ((int)th.size() < helpers) th.emplace_back(&HostPool::worker, this, (int)th.size());
        }
        catch (...)
        {
        }  // thread creation failed: the threads that exist (and the caller) do the work
        const int n = std::min(helpers, (int)th.size());
        if (n > 0)
        {
            std::lock_guard<std::mutex> lk(m);
            job    = &work;
            n_run  = n;
            n_left = n;
            ++gen;
        }
        if (n > 0) cv.notify_all();
        work();
        if (n > 0)
        {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return n_left == 0; });
            job = nullptr;
        }
    }
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};

struct snk_ba : HandleBase
{
    BaLists lists;
    HostPool pool;    // host threads of the batch hand-over
    HostBuf h_stage;  // pinned staging of the small per-call transfers (outlier masks)
    DevBuf d_becnt;   // per (camera, 64-item chunk, camera) counters of the device-built block entries
    DevBuf d_probcond;  // the problem table of a conditional extra iteration (select_marked)
    DevBuf d_campart, d_ccstart, d_ccitems;  // per (work item, free camera) sums of schur_fused<3, true> and the per-camera lists of them
    bool state_fresh = false;                // the device state is the uploaded initial one (no solve since the hand-over)
    bool blk_built = true;                   // the camera-pair block entries of the current problem set exist on the device (see ba_sets_will_run)
    bool cam_sums_ok = false;                // every observation of a free camera belongs to a work item with pairs (no constant point seen by a free camera)
    snk_ba_options opt{};
    int count = 0;
    std::vector<Prob> probs;
    int tot_img = 0, tot_pt = 0, tot_obs = 0, tot_cam = 0, tot_orig = 0, tot_vec = 0;
    long long tot_s = 0;
    int max_np = 0, max_nfc = 0, max_n6 = 0, max_ni = 0, max_set_items = 0;
    bool set_ok = false, set_small = false;
    int set_k_max = 0, set_run_max = 0;
    DevBuf d_setitems, d_setpts, d_setpairs, d_setobs, d_cblkstart, d_cblkitems, d_spart;
    DevBuf d_prob, d_state, d_pose, d_pose_new, d_pose0, d_pt, d_pt_new, d_pt0, d_ptc, d_camidx, d_ptstart, d_oimg, d_ocam,
        d_optfree, d_ouv, d_odepth, d_oweight, d_oorig, d_outlier, d_csobs, d_r, d_W, d_ptv, d_Vinv, d_bp, d_cost,
        d_cost_new, d_U, d_camstart, d_camitems, d_blkstart, d_blkent, d_S, d_rhs, d_x, d_chi2, d_pcgw, d_optidx, d_wvpt, d_rpcmeta, d_rpcnext, d_camrpcstart, d_camrpcitems, d_blkrpc, d_rpcout;
    int max_rpc = 0;
    int max_wv = 0;
    bool point_wave_ok = false;  // every problem has a point_wave work list (no point with > 64 observations)
    PcgLarge pcgw{};     // work arrays of the multi-workgroup PCG (only when the reduced system exceeds the LDS)
    bool pcg_large = false;
    Arrays arr{};
    std::vector<int> orig_off, orig_n;
    std::map<int, hipGraphExec_t> graphs;  // LM launch sequence captured per iteration count
    std::map<int, int> plain_runs;         // solves issued with plain launches since set_problems, per iteration count
    void drop_graphs()
    {
        for (auto& g : graphs) (void)hipGraphExecDestroy(g.second);
        graphs.clear();
        plain_runs.clear();
    }
};

namespace
{
Opt make_opt(const snk_ba_options& o)
{
    Opt d;
    d.max_pcg      = o.max_pcg_iterations;
    static const bool pcg_general = getenv("SNK_BA_PCG_GENERAL") != nullptr;
    d.pcg_general  = pcg_general ? 1 : 0;
    d.pcg_tol      = o.pcg_tol;
    d.huber_mono   = o.huber_mono;
    d.huber_stereo = o.huber_stereo;
    d.lambda_init  = o.lambda_init > 0.0 ? o.lambda_init : 1e-4;
    d.chi2_mono = d.chi2_stereo = 0.0;
    return d;
}

// The hand-over's lists reach the device with ONE kernel that reads the pinned host vectors over the bus (hipHostMalloc memory is
// device-visible) and writes the device arrays: 33 separate copies cost ~6 us each on the copy engine whatever their size.
constexpr int COPY_TAB_MAX = 48;
struct CopyTab
{
    const void* src[COPY_TAB_MAX];
    void* dst[COPY_TAB_MAX];
    unsigned bytes[COPY_TAB_MAX];
    int n;
};
// ---- scene lists built on the device -------------------------------------------------------------------------------------------
// The static per-camera observation records (what cam_pass streams) are a gather of the sorted observation arrays through the
// camera lists: 40 bytes per observation that neither the host loop nor the bus has to touch.
// Batches (round 6): three of the sorted observation arrays and the camera lists are functions of arrays that are on the device anyway --
// 13 of the 53 bytes per observation a hand-over used to carry over the bus (210 MB of a 1024-window batch's 1.07 GB).
//   o_pt[s]     = the point whose run [pt_start[p], pt_start[p + 1]) holds s (binary search),
//   o_cam[s]    = cam_idx[o_img[s]],   o_ptfree[s] = !pt_const[o_pt[s]]
__global__ __launch_bounds__(256) void derive_obs_fields(Arrays A, int* __restrict__ o_pt, int* __restrict__ o_cam, unsigned char* __restrict__ o_ptfree)
{
    const Prob pr = A.prob[blockIdx.y];
    const int s   = blockIdx.x * 256 + threadIdx.x;
    if (s >= pr.no) return;
    const int* ps = A.pt_start + pr.ptstart_off;
    int lo = 0, hi = pr.np;  // ps[lo] <= s < ps[hi]
    while (hi - lo > 1)
    {
        const int mid = (lo + hi) >> 1;
        if (ps[mid] <= s) lo = mid;
        else hi = mid;
    }
    const int go  = pr.obs_off + s;
    o_pt[go]      = lo;
    o_cam[go]     = A.cam_idx[pr.img_off + A.o_img[go]];
    o_ptfree[go]  = A.pt_const[pr.pt_off + lo] ? 0 : 1;
}
// cam_items of camera c = the observations s with o_cam[s] == c in ascending s (the host builder's order): one wavefront per (camera,
// problem) walks the observations 64 at a time and compacts by ballot.  cam_start comes from the host (nfc + 1 ints per problem).
__global__ __launch_bounds__(64) void derive_cam_items(Arrays A, const int* __restrict__ o_cam, int* __restrict__ cam_items)
{
    const Prob pr = A.prob[blockIdx.y];
    const int c   = blockIdx.x;
    if (c >= pr.nfc) return;
    const int lane = threadIdx.x;
    int at = pr.citem_off + A.cam_start[pr.camstart_off + c];
    for (int s0 = 0; s0 < pr.no; s0 += 64)
    {
        const int s    = s0 + lane;
        const bool hit = s < pr.no && o_cam[pr.obs_off + s] == c;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
        if (hit) cam_items[at + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = s;
        at += __popcll(m);
    }
}

__global__ __launch_bounds__(256) void gather_cam_records(Arrays A, CamObs* __restrict__ out)
{
    const Prob pr = A.prob[blockIdx.y];
    const int n   = A.cam_start[pr.camstart_off + pr.nfc];
    const int k   = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int go = pr.obs_off + A.cam_items[pr.citem_off + k];
    const double2 uv = A.o_uv[go];
    CamObs rec;
    rec.u = uv.x; rec.v = uv.y; rec.depth = A.o_depth[go]; rec.weight = A.o_weight[go];
    rec.ptw = A.o_pt[go] | (A.o_ptfree[go] ? (int)0x80000000u : 0); rec.orig = A.o_orig[go];
    out[pr.citem_off + k] = rec;
}

// The static observation records in the order schur_fused / update_cost walk them (work item, point of the item, observation of
// the point) are a gather as well: 48 bytes per observation that the host loop wrote one by one and the bus carried (786 KB per
// benchmark window -- half of a batch's upload).  One workgroup per work item.
__global__ __launch_bounds__(256) void gather_set_records(Arrays A, SetObs* __restrict__ out)
{
    const Prob pr = A.prob[blockIdx.y];
    if ((int)blockIdx.x >= pr.n_set) return;
    const SetItem si = A.set_items[pr.set_off + blockIdx.x];
    const int n      = si.n_pts * si.run;
    for (int idx = threadIdx.x; idx < n; idx += 256)
    {
        const int q = idx / si.run, a = idx - q * si.run;
        const int go = pr.obs_off + A.set_pts[si.pts_off + q].y + a;
        const double2 uv = A.o_uv[go];
        SetObs rec;
        rec.u = uv.x; rec.v = uv.y; rec.depth = A.o_depth[go]; rec.weight = A.o_weight[go];
        rec.orig = A.o_orig[go]; rec.pk = set_pack(A.o_img[go], A.o_cam[go], A.o_ptfree[go]);
        out[(size_t)si.rec_off + idx] = rec;
    }
}

// The camera-pair block entries of schur_pass -- for every upper block (c1, c2) the co-observations (observation of c1,
// observation of c2, point) in ascending order -- built by three launches instead of a host loop over every pair of every
// point (half of a scene hand-over's host time, and 1.1 MB over the bus for a 20 x 2000 x 8 window).  Requirements (checked
// on the host, which keeps its own builder for the other scenes): <= BE_MAX_CAMS free cameras, no camera twice on a point.
// One wavefront per (camera c1, 64 items of its list): lane = one observation a of c1; the free cameras >= c1 in its point's
// run are a bit mask (BE_WORDS x 64 bits in registers); ballot(c2 in mask) ranks the lane inside the chunk for block (c1, c2).
// Order inside a block = list order of c1 = ascending observation index = ascending point: the host builder's order.
constexpr int BE_WORDS = 8, BE_MAX_CAMS = 64 * BE_WORDS;
__device__ inline void block_entry_item(const Arrays& A, const Prob& pr, int c1, int chunk, int lane, int& a, int& p, int& r0, int& r1,
                                        unsigned long long (&mask)[BE_WORDS])
{
    const int s0 = A.cam_start[pr.camstart_off + c1], s1 = A.cam_start[pr.camstart_off + c1 + 1];
    const int k  = s0 + chunk * 64 + lane;
    a = -1, p = 0, r0 = 0, r1 = 0;
#pragma unroll
    for (int w = 0; w < BE_WORDS; ++w) mask[w] = 0ull;
    if (k >= s1) return;
    const int s  = A.cam_items[pr.citem_off + k];
    const int go = pr.obs_off + s;
    if (!A.o_ptfree[go]) return;  // constant points produce no Schur products
    a  = s;
    p  = A.o_pt[go];
    r0 = A.pt_start[pr.ptstart_off + p], r1 = A.pt_start[pr.ptstart_off + p + 1];
    for (int c = r0; c < r1; ++c)
    {
        const int cc = A.o_cam[pr.obs_off + c];
        if (cc < c1) continue;
        const unsigned long long bit = 1ull << (cc & 63);
#pragma unroll
        for (int w = 0; w < BE_WORDS; ++w)
            if (w == (cc >> 6)) mask[w] |= bit;
    }
}

__global__ __launch_bounds__(64) void block_entries_count(Arrays A, int* __restrict__ cnt)
{
    const Prob pr = A.prob[blockIdx.y];
    const int w0  = blockIdx.x;
    if (pr.be_nch <= 0 || w0 >= pr.nfc * pr.be_nch) return;
    const int c1 = w0 / pr.be_nch, chunk = w0 - c1 * pr.be_nch, lane = threadIdx.x;
    int a, p, r0, r1;
    unsigned long long mask[BE_WORDS];
    block_entry_item(A, pr, c1, chunk, lane, a, p, r0, r1, mask);
    int* out = cnt + pr.becnt_off + (size_t)(c1 * pr.be_nch + chunk) * pr.nfc;
#pragma unroll
    for (int w = 0; w < BE_WORDS; ++w)
    {
        if (w * 64 >= pr.nfc) break;  // uniform
        int mine = 0;
        if (__builtin_amdgcn_ballot_w64(mask[w] != 0ull) != 0ull)  // uniform: most words of a big scene are empty
            for (int b = 0; b < 64; ++b)
            {
                const unsigned long long m = __builtin_amdgcn_ballot_w64((mask[w] >> b) & 1ull);
                if (lane == b) mine = __popcll(m);
            }
        if (w * 64 + lane < pr.nfc) out[w * 64 + lane] = mine;  // (cameras below c1 are in no mask: zeros)
    }
}

// per problem: chunk counts -> chunk bases (in place), block totals -> blk_start (exclusive scan over the nfc * nfc blocks)
constexpr int BE_SCAN_THREADS = 1024;
__global__ __launch_bounds__(BE_SCAN_THREADS) void block_entries_scan(Arrays A, int* __restrict__ cnt, int* __restrict__ blk_start)
{
    __shared__ int s_wave[BE_SCAN_THREADS / 64];
    __shared__ int s_run;
    const Prob pr = A.prob[blockIdx.x];
    const int nb = pr.nfc * pr.nfc, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += BE_SCAN_THREADS)
    {
        const int blk = base + tid;
        int tot = 0;
        if (blk < nb && pr.be_nch > 0)
        {
            const int c1 = blk / pr.nfc, c2 = blk - c1 * pr.nfc;
            if (c2 >= c1)  // the lower blocks have no entries (and their counters were never written)
                for (int ch = 0; ch < pr.be_nch; ++ch)
                {
                    int* q = cnt + pr.becnt_off + (size_t)(c1 * pr.be_nch + ch) * pr.nfc + c2;
                    const int v = *q;
                    *q = tot;
                    tot += v;
                }
        }
        // inclusive scan: inside the wavefront by shuffles, then the wavefront totals
        int incl = tot;
        incl = wave_scan_incl_dpp(incl);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        const int run = s_run;
        if (blk < nb) blk_start[pr.blkstart_off + blk] = run + before + incl - tot;
        __syncthreads();
        if (tid == BE_SCAN_THREADS - 1) s_run = run + before + incl;
        __syncthreads();
    }
    if (tid == 0) blk_start[pr.blkstart_off + nb] = s_run;
}

__global__ __launch_bounds__(64) void block_entries_fill(Arrays A, const int* __restrict__ cnt, int4* __restrict__ blk_ent)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char be_smem[];
    const Prob pr = A.prob[blockIdx.y];
    const int w0  = blockIdx.x;
    if (pr.be_nch <= 0 || w0 >= pr.nfc * pr.be_nch) return;
    unsigned long long* s_ball = reinterpret_cast<unsigned long long*>(be_smem);  // [words * 64] ballots per camera c2
    const int nwords = (pr.nfc + 63) >> 6;
    int* s_base = reinterpret_cast<int*>(s_ball + nwords * 64);                     // [words * 64] first entry of (c1, chunk, c2)
    const int c1 = w0 / pr.be_nch, chunk = w0 - c1 * pr.be_nch, lane = threadIdx.x;
    int a, p, r0, r1;
    unsigned long long mask[BE_WORDS];
    block_entry_item(A, pr, c1, chunk, lane, a, p, r0, r1, mask);
    const int* cin = cnt + pr.becnt_off + (size_t)(c1 * pr.be_nch + chunk) * pr.nfc;
#pragma unroll
    for (int w = 0; w < BE_WORDS; ++w)
    {
        if (w * 64 >= pr.nfc) break;  // uniform
        unsigned long long mine = 0ull;
        if (__builtin_amdgcn_ballot_w64(mask[w] != 0ull) != 0ull)
            for (int b = 0; b < 64; ++b)
            {
                const unsigned long long m = __builtin_amdgcn_ballot_w64((mask[w] >> b) & 1ull);
                if (lane == b) mine = m;
            }
        const int c2 = w * 64 + lane;
        s_ball[c2] = mine;
        s_base[c2] = c2 >= c1 && c2 < pr.nfc ? A.blk_start[pr.blkstart_off + c1 * pr.nfc + c2] + cin[c2] : 0;
    }
    __syncthreads();
    if (a < 0) return;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int c = r0; c < r1; ++c)
    {
        const int cc = A.o_cam[pr.obs_off + c];
        if (cc < c1) continue;
        blk_ent[(size_t)pr.ent_off + s_base[cc] + __popcll(s_ball[cc] & below)] = make_int4(a, c, p, 0);
    }
}

// The extra iteration of SolveLocalScene runs only for scenes whose chi-square pass marked something
// (LocalBundleAdjustment.cpp:399).  Instead of reading the count back and deciding on the host, the iteration is enqueued
// behind the pass with THIS table of problems: an unmarked problem appears with every size zero, so each kernel of the
// iteration finds nothing to do for it (the same way an empty scene in a batch does); begin_solve and accept_pass, which touch
// the per-problem state whatever the sizes, take the condition as an argument.
__global__ __launch_bounds__(64) void select_marked(const Prob* __restrict__ prob, Prob* __restrict__ out, State* __restrict__ st, int n,
                                                    double lambda_init)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    Prob p = prob[i];
    if (st[i].marked == 0)
        p.ni = p.np = p.no = p.nfc = p.n6 = p.n_wv = p.n_rpc = p.n_set = p.be_nch = 0;
    else  // what begin_solve does (but the count stays: the kernels of the iteration test it)
    {
        st[i].lambda    = lambda_init;
        st[i].vfac      = 2.0;
        st[i].iter      = 0;
        st[i].pcg_iters = 0;
        st[i].accepted  = 0;
    }
    out[i] = p;
}

__global__ __launch_bounds__(256) void copy_table_kernel(CopyTab T)
{
    const int e = blockIdx.y;
    const unsigned nb = T.bytes[e], nq = nb >> 4;
    const uint4* s4 = static_cast<const uint4*>(T.src[e]);  // both sides are at least 256-byte aligned (hipHostMalloc / hipMalloc)
    uint4* d4       = static_cast<uint4*>(T.dst[e]);
    if (s4 == nullptr)  // a buffer that starts as zeros
    {
        for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < nq; i += gridDim.x * 256u) d4[i] = make_uint4(0u, 0u, 0u, 0u);
        if (blockIdx.x == 0)
            for (unsigned i = (nq << 4) + threadIdx.x; i < nb; i += 256u) static_cast<unsigned char*>(T.dst[e])[i] = 0;
        return;
    }
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < nq; i += gridDim.x * 256u) d4[i] = s4[i];
    if (blockIdx.x == 0)
    {
        const unsigned char* sb = static_cast<const unsigned char*>(T.src[e]);
        unsigned char* db       = static_cast<unsigned char*>(T.dst[e]);
        for (unsigned i = (nq << 4) + threadIdx.x; i < nb; i += 256u) db[i] = sb[i];
    }
}
template <typename V>
int upload(DevBuf& b, const V& v, CopyTab& tab)
{
    using T = typename V::value_type;
    int rc  = b.reserve(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (rc != SNK_OK) return rc;
    if (v.empty()) return SNK_OK;
    SNK_REQUIRE(tab.n < COPY_TAB_MAX && v.size() * sizeof(T) < (1ull << 32), "scene list too large for the upload table");
    tab.src[tab.n]   = v.data();
    tab.dst[tab.n]   = b.p;
    tab.bytes[tab.n] = (unsigned)(v.size() * sizeof(T));
    ++tab.n;
    return SNK_OK;
}
}  // namespace

// One launch sequence, two ways to issue it: plain launches on the handle's stream, or kernel nodes appended to an
// explicitly built hipGraph (a linear chain).  No stream capture anywhere: capture is process-visible state that any
// other seam's thread (FeatureDetection || Preprocess || Tracking || LBA all run at once) can
// invalidate with an allocation or a synchronous copy; hipGraphAddKernelNode touches nothing but this handle's graph.
struct Launcher
{
    hipStream_t st      = nullptr;
    hipGraph_t graph    = nullptr;  // non-null: record instead of launching
    hipGraphNode_t last = nullptr;
    hipError_t err      = hipSuccess;
    int line            = 0;
    bool cooperative    = false;  // the next launch is a cooperative one (all workgroups resident: grid barriers inside); plain launches only
    template <typename... P, typename... A>
    void operator()(int at, void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, A&&... a)
    {
        static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
        if (err != hipSuccess) return;
        std::tuple<P...> args{static_cast<P>(a)...};
        void* ptrs[sizeof...(P)];
        fill(ptrs, args, std::index_sequence_for<P...>{});
        if (cooperative)
        {
            cooperative = false;
            err         = graph ? hipErrorNotSupported : hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, (unsigned)lds, st);
        }
        else if (!graph)
            err = hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, st);
        else
        {
            hipKernelNodeParams np{};
            np.func           = reinterpret_cast<void*>(kernel);
            np.gridDim        = grid;
            np.blockDim       = block;
            np.sharedMemBytes = (unsigned)lds;
            np.kernelParams   = ptrs;
            np.extra          = nullptr;
            hipGraphNode_t node = nullptr;
            err = hipGraphAddKernelNode(&node, graph, last ? &last : nullptr, last ? 1 : 0, &np);
            last = node;
        }
        if (err != hipSuccess) line = at;
    }
    template <typename T, size_t... I>
    static void fill(void** ptrs, T& args, std::index_sequence<I...>)
    {
        ((ptrs[I] = &std::get<I>(args)), ...);
    }
};
#define LAUNCH(...) L(__LINE__, __VA_ARGS__)

// Will the LM sequence of this problem set run the point-major kernels (schur_fused / schur_mfma + update_cost over the camera-set work
// items)?  Decided by the hand-over's results and by switches that are read once per process: the hand-over asks too, because the
// camera-pair block entries are only read by the block-major pass (schur_pass) -- building them for a 1024-window batch that never
// runs it was 1.4 ms of device time behind every hand-over (block_entries_count 0.40 + block_entries_fill 1.03 ms, round 6 trace).
static bool ba_sets_will_run(const snk_ba* h)
{
    static const bool no_wave       = getenv("SNK_BA_NO_POINT_WAVE") != nullptr;
    static const bool no_set        = getenv("SNK_BA_NO_SCHUR_SET") != nullptr;
    static const long long set_min  = getenv("SNK_BA_SCHUR_SET_MIN_ITEMS") ? atoll(getenv("SNK_BA_SCHUR_SET_MIN_ITEMS")) : 256;
    return h->max_nfc > 0 && h->point_wave_ok && !no_wave && h->set_ok && !no_set && (long long)h->max_set_items * h->count >= set_min;
}

extern "C" {

int snk_ba_create(const snk_ba_options* options, int device, void* stream, snk_ba** out)
{
    SNK_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    SNK_REQUIRE(options != nullptr, "options is NULL");
    SNK_REQUIRE(options->max_iterations >= 0 && options->max_pcg_iterations >= 0, "negative iteration count");
    SNK_REQUIRE(options->huber_mono > 0.0 && options->huber_stereo > 0.0, "Huber thresholds must be > 0");
    snk_ba* h = new snk_ba();
    h->opt    = *options;
    int rc    = h->init(device, stream);
    if (rc != SNK_OK)
    {
        delete h;
        return rc;
    }
    *out = h;
    return SNK_OK;
}

int snk_ba_destroy(snk_ba* h)
{
    if (!h) return SNK_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);  // an upload of a hand-over (it reads the handle's pinned lists) or a solve may still be in flight
    // every device buffer of the handle (the struct's DevBuf members)
    DevBuf* all[] = {&h->d_setitems, &h->d_setpts, &h->d_setpairs, &h->d_setobs, &h->d_cblkstart, &h->d_cblkitems, &h->d_spart,
                     &h->d_prob, &h->d_state, &h->d_pose, &h->d_pose_new, &h->d_pose0, &h->d_pt, &h->d_pt_new,
                     &h->d_pt0, &h->d_ptc, &h->d_camidx, &h->d_ptstart, &h->d_oimg, &h->d_ocam, &h->d_optfree,
                     &h->d_ouv, &h->d_odepth, &h->d_oweight, &h->d_oorig, &h->d_outlier, &h->d_csobs, &h->d_r,
                     &h->d_W, &h->d_ptv, &h->d_Vinv, &h->d_bp, &h->d_cost, &h->d_cost_new, &h->d_U, &h->d_camstart,
                     &h->d_camitems, &h->d_blkstart, &h->d_blkent, &h->d_S, &h->d_rhs, &h->d_x, &h->d_chi2,
                     &h->d_pcgw, &h->d_optidx, &h->d_wvpt, &h->d_rpcmeta, &h->d_rpcnext, &h->d_camrpcstart,
                     &h->d_camrpcitems, &h->d_blkrpc, &h->d_rpcout, &h->d_becnt, &h->d_probcond, &h->d_campart, &h->d_ccstart, &h->d_ccitems};
    for (DevBuf* b : all) b->release();
    h->h_stage.release();
    h->drop_graphs();
    h->fini();
    delete h;
    return SNK_OK;
}

int snk_ba_sync(snk_ba* h)
{
    SNK_REQUIRE(h != nullptr, "ba is NULL");
    SNK_HIP_CHECK(hipStreamSynchronize(h->stream));
    return SNK_OK;
}

int snk_ba_set_problems(snk_ba* h, const snk_ba_problem* problems, int count)
{
    SNK_REQUIRE(h != nullptr, "ba is NULL");
    SNK_REQUIRE(count >= 1 && count <= 65535 && problems != nullptr, "count must be 1..65535");
    for (int b = 0; b < count; ++b)  // the packed observation records hold the image index in 15 bits (SetObs)
        SNK_REQUIRE(problems[b].n_img <= SET_MAX_IMG, "a problem has more than 32767 images");
    SNK_HIP_CHECK(hipSetDevice(h->device));
    SNK_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->drop_graphs();
    const auto t_begin = std::chrono::steady_clock::now();

    BaLists& LS = h->lists;  // pinned, capacity kept from the previous scene
    LS.clear();
    auto& probs = LS.probs;
    probs.resize((size_t)count);
    auto &pose = LS.pose, &pt = LS.pt, &ouv2 = LS.ouv2, &odepth = LS.odepth, &oweight = LS.oweight;
    auto &ptc = LS.ptc, &optfree = LS.optfree;
    auto& setitems = LS.setitems;
    auto& setpts   = LS.setpts;
    auto &setpairs = LS.setpairs, &cblkstart = LS.cblkstart, &cblkitems = LS.cblkitems, &ccstart = LS.ccstart, &ccitems = LS.ccitems;
    int n_partials = 0, max_set_items = 0, max_set_pairs = 0, max_set_run = 0, max_set_k = 0;
    int n_cparts = 0;          // per (work item, free camera) partial sums of the camera pass
    long long n_setrec = 0;    // static observation records of the work items
    bool cam_sums_ok = true;   // no constant point is seen by a free camera (its observations are in no work item with pairs)
    bool set_ok = true;  // every problem can run the point-major Schur pass
    auto &camidx = LS.camidx, &ptstart = LS.ptstart, &oimg = LS.oimg, &ocam = LS.ocam, &oorig = LS.oorig, &camstart = LS.camstart,
         &camitems = LS.camitems, &blkstart = LS.blkstart, &optidx = LS.optidx, &wvpt = LS.wvpt, &rpcnext = LS.rpcnext,
         &camrpcstart = LS.camrpcstart, &camrpcitems = LS.camrpcitems, &blkrpc = LS.blkrpc;
    auto& rpcmeta = LS.rpcmeta;
    int max_rpc = 0;
    int max_wv = 0;
    bool wave_ok = true;
    auto& blkent = LS.blkent;
    {
        // the big lists are sized by the totals of the call: growing a pinned vector re-pins and copies it every time it doubles
        size_t t_img = 0, t_pt = 0, t_obs = 0;
        for (int b = 0; b < count; ++b)
        {
            t_img += (size_t)std::max(problems[b].n_img, 0);
            t_pt += (size_t)std::max(problems[b].n_pt, 0);
            t_obs += (size_t)std::max(problems[b].n_obs, 0);
        }
        pose.reserve(7 * t_img), pt.reserve(3 * t_pt), ptc.reserve(t_pt), camidx.reserve(t_img), ptstart.reserve(t_pt + (size_t)count);
        ouv2.reserve(2 * t_obs), odepth.reserve(t_obs), oweight.reserve(t_obs), optfree.reserve(t_obs), oimg.reserve(t_obs);
        ocam.reserve(t_obs), oorig.reserve(t_obs), optidx.reserve(t_obs), camitems.reserve(t_obs);
        // ... and so are the lists of the point-major pass when they will be built (batches, big scenes): on a 300-keyframe
        // scene the doubling of setobs / cblkstart / cblkitems through fresh pinned allocations was 17 of the 26 ms of a
        // first hand-over (the same scene again on the handle, capacities kept: 8 ms)
        size_t t_blk = 0;
        bool sets_likely = count >= 8 || getenv("SNK_BA_SCHUR_SET_MIN_ITEMS") != nullptr;
        for (int b = 0; b < count; ++b)
        {
            size_t nfc = 0;
            if (problems[b].img_const)
                for (int i = 0; i < problems[b].n_img; ++i) nfc += problems[b].img_const[i] ? 0 : 1;
            t_blk += nfc * nfc + 1;
            sets_likely |= problems[b].n_pt >= 8000;
        }
        cblkstart.reserve(t_blk);
        if (sets_likely) setpts.reserve(t_pt), cblkitems.reserve(2 * t_obs), ccitems.reserve(t_obs);
    }
    h->orig_off.assign((size_t)count, 0);
    h->orig_n.assign((size_t)count, 0);
    int img_off = 0, pt_off = 0, obs_off = 0, cam_off = 0, orig_off = 0, vec_off = 0;
    long long s_off = 0;
    int max_np = 0, max_nfc = 0, max_n6 = 0, max_ni = 0;

    // work items of up to 128 points: see SET_CHUNK_BIG.  Needs the default point-major kernels (the A/B switches that select
    // the others are read here as well) and at most 8 free observations per point in EVERY problem (else the batch runs
    // point_wave + schur_mfma<4>), which is known only after a look at all of them.
    // (decided behind the sizing pass below, which counts the free observations per point on the host threads: as a serial loop over
    // every observation of the batch in front of everything else it was ~20 of a 1024-window hand-over's 53 ms of list time, round 6)
    bool big_items = false;
    static const bool alt_paths = getenv("SNK_BA_NO_SCHUR_FUSED") || getenv("SNK_BA_NO_SCHUR_MFMA") || getenv("SNK_BA_NO_UPDATE_COST") ||
                                  getenv("SNK_BA_FUSED_K10") || getenv("SNK_BA_NO_SCHUR_SET") || getenv("SNK_BA_NO_POINT_WAVE") ||
                                  getenv("SNK_BA_NO_BIG_ITEMS");
    const bool want_big_items = count >= 256 && !alt_paths;
    size_t blkrpc_logical = 0;   // entries of blk_rpc up to the current problem (materialised only for problems with constraints)
    bool dev_entries_ok = true;  // every problem can have its block entries built on the device
    std::vector<long long> ent_bound((size_t)count, 0);
    int blkstart_total = 0, max_citems = 0;
    // camera-pair blocks on the host: co-observations of every ordered pair (dense block grid, empty blocks allowed).  The
    // builder for scenes the device kernels do not take (more than 512 free cameras, one camera twice on a point), and their checker.
    auto host_block_entries = [&](int b, pvec<int>& blkstart, pvec<int4>& blkent)
    {
        const snk_ba_problem& P = problems[b];
        const Prob& pr          = probs[(size_t)b];
        const int nfc           = pr.nfc;
        const int* pstart       = ptstart.data() + pr.ptstart_off;
        const int* s_cam        = ocam.data() + pr.obs_off;
        {
            const size_t nb = (size_t)nfc * nfc;
            std::vector<int> bs(nb + 1, 0);
            for (int p = 0; p < P.n_pt; ++p)
            {
                if (P.pt_const[p]) continue;
                for (int a = pstart[(size_t)p]; a < pstart[(size_t)p + 1]; ++a)
                {
                    if (s_cam[(size_t)a] < 0) continue;
                    for (int c = pstart[(size_t)p]; c < pstart[(size_t)p + 1]; ++c)  // upper blocks only: schur_pass never reads the others
                        if (s_cam[(size_t)c] >= s_cam[(size_t)a]) bs[(size_t)s_cam[(size_t)a] * nfc + s_cam[(size_t)c] + 1]++;
                }
            }
            for (size_t k = 0; k < nb; ++k) bs[k + 1] += bs[k];
            const size_t ent_at = blkent.size();
            blkent.resize(ent_at + (size_t)bs[nb]);  // filled in place (no second copy of a megabyte of entries)
            int4* ent = blkent.data() + ent_at;
            std::vector<int> fill(bs.begin(), bs.end() - 1);
            for (int p = 0; p < P.n_pt; ++p)
            {
                if (P.pt_const[p]) continue;
                for (int a = pstart[(size_t)p]; a < pstart[(size_t)p + 1]; ++a)
                {
                    if (s_cam[(size_t)a] < 0) continue;
                    for (int c = pstart[(size_t)p]; c < pstart[(size_t)p + 1]; ++c)
                        if (s_cam[(size_t)c] >= s_cam[(size_t)a])
                        {
                            int4 e;
                            e.x = a;
                            e.y = c;
                            e.z = p;
                            e.w = 0;
                            ent[(size_t)fill[(size_t)s_cam[(size_t)a] * nfc + s_cam[(size_t)c]]++] = e;
                        }
                }
            }
            blkstart.insert(blkstart.end(), bs.begin(), bs.end());
        }
    };
    static const bool prof_sections = getenv("SNK_BA_PROFILE_CREATE") != nullptr;
    long long sec_us[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto sec_t = std::chrono::steady_clock::now();
    auto mark = [&](int k)
    {
        if (!prof_sections) return;
        const auto now = std::chrono::steady_clock::now();
        sec_us[k] += (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(now - sec_t).count();
        sec_t = now;
    };
    // ---- sizing pass + fill pass over the problems, on several host threads for batches (round 4) ----
    // The values, the free-camera indices, the counting sort by point and the eight sorted observation arrays of a problem depend on nothing
    // but that problem, and they were half of a batch hand-over's list time on ONE core (1024 windows: 117 of 227 ms).  Pass 1 counts
    // (valid observations, free cameras) per problem, a prefix sum gives every problem its place in the shared lists, pass 2 writes the
    // places directly -- disjoint ranges, no locks.  The per-problem loop below then only READS these lists.  Same contents as the serial
    // builder (the loop's old code, run per problem): SNK_BA_CHECK_LISTS and the bit-identity tests of the variants suite cover it.
    for (int b = 0; b < count; ++b)
    {
        const snk_ba_problem& P = problems[b];
        SNK_REQUIRE(P.n_img >= 0 && P.n_pt >= 0 && P.n_obs >= 0, "negative problem size");
        SNK_REQUIRE(P.n_img == 0 || (P.pose && P.img_const), "NULL pose arrays");
        SNK_REQUIRE(P.n_pt == 0 || (P.pt && P.pt_const), "NULL point arrays");
        SNK_REQUIRE(P.n_obs == 0 || (P.obs_img && P.obs_pt && P.obs_uv && P.obs_depth && P.obs_weight), "NULL observation arrays");
        SNK_REQUIRE(P.n_rpc >= 0 && (P.n_rpc == 0 || P.rpc != nullptr), "bad relative pose constraints");
    }
    struct PreProb
    {
        int nfc, no;
        size_t img_at, pt_at, ps_at, obs_at;
        int orig_at;
        char dup;  // one camera twice on a point (device-built block entries are then off)
        char k_over8;  // a point with more than eight free observations (work items of up to 128 points are then off)
    };
    std::vector<PreProb> pre((size_t)count);
    static const int host_threads_env = getenv("SNK_BA_HOST_THREADS") ? atoi(getenv("SNK_BA_HOST_THREADS")) : 0;
    int n_threads = 1;
    if (count >= 16)
    {
        // up to 32 threads (round 6; 16 before): on the 256-thread hosts of the MI355X boxes a 1024-window hand-over builds its lists in 35
        // instead of 53 ms with 32, no faster with 64 (profiles/r06/r06i_ba_handover_threads_before.txt)
        n_threads = host_threads_env > 0 ? host_threads_env : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
        n_threads = std::min(n_threads, count / 8);
    }
    else if (host_threads_env > 0 && count >= 2)
        n_threads = std::min(host_threads_env, count);  // tests force the threaded form on small batches
    std::atomic<bool> worker_failed{false};
    int pf_lo = 0, pf_hi = count;  // the problems a parallel_for covers (the fill pass of a batch runs in chunks, see below)
    auto parallel_for = [&](auto&& body)
    {
        const int lo = pf_lo, hi = pf_hi;
        if (n_threads <= 1)
        {
            for (int b = lo; b < hi; ++b) body(b);
            return;
        }
        std::atomic<int> next{lo};
        // an exception in a worker (the vectors it grows: std::bad_alloc) must not reach std::terminate: it is caught, the remaining
        // work is abandoned and the caller turns worker_failed into an error code after the pass
        const std::function<void()> work = [&]()
        {
            try
            {
                for (;;)
                {
                    const int b0 = next.fetch_add(8);
                    if (b0 >= hi || worker_failed.load(std::memory_order_relaxed)) return;
                    for (int b = b0; b < std::min(b0 + 8, hi); ++b) body(b);
                }
            }
            catch (...)
            {
                worker_failed.store(true);
            }
        };
        static const bool no_pool = getenv("SNK_BA_NO_HOST_POOL") != nullptr;  // A/B: threads created and joined per pass (rounds 4-6)
        if (!no_pool)
        {
            h->pool.run(n_threads - 1, work);
            return;
        }
        std::vector<std::thread> pool;
        try
        {
            for (int t = 1; t < n_threads; ++t) pool.emplace_back(work);
        }
        catch (...)
        {
            worker_failed.store(true);  // thread creation failed: the threads that exist finish, this thread does the rest
        }
        work();
        for (auto& t : pool) t.join();
    };
    parallel_for([&](int b)
    {
        const snk_ba_problem& P = problems[b];
        PreProb& q = pre[(size_t)b];
        q.nfc = 0;
        for (int i = 0; i < P.n_img; ++i) q.nfc += P.img_const[i] ? 0 : 1;
        int no = 0;
        for (int o = 0; o < P.n_obs; ++o)
        {
            const int i = P.obs_img[o], p = P.obs_pt[o];
            if (i < 0 || i >= P.n_img || p < 0 || p >= P.n_pt) continue;
            if (P.img_const[i] && P.pt_const[p]) continue;  // reference LocalBundleAdjustment.cpp:286
            ++no;
        }
        q.no  = no;
        q.dup = 0;
        q.k_over8 = 0;
        if (want_big_items && P.n_pt > 0 && P.n_obs > 0)
        {
            // work items of up to 128 points need at most 8 free observations per point in EVERY problem
            std::vector<unsigned char> kfree((size_t)P.n_pt, 0);
            for (int o = 0; o < P.n_obs; ++o)
            {
                const int i = P.obs_img[o], p = P.obs_pt[o];
                if (i < 0 || i >= P.n_img || p < 0 || p >= P.n_pt || P.img_const[i]) continue;
                if (++kfree[(size_t)p] > 8) q.k_over8 = 1;
            }
        }
    });
    if (want_big_items)
    {
        big_items = true;
        for (int b = 0; b < count; ++b) big_items = big_items && !pre[(size_t)b].k_over8;
    }
    if (worker_failed.load())
    {
        set_error("snk_ba_set_problems: a list-building thread failed (out of host memory?)");
        return SNK_ERR_HIP;
    }
    {
        size_t a_img = 0, a_pt = 0, a_ps = 0, a_obs = 0;
        long long a_orig = 0;
        for (int b = 0; b < count; ++b)
        {
            PreProb& q = pre[(size_t)b];
            q.img_at = a_img, q.pt_at = a_pt, q.ps_at = a_ps, q.obs_at = a_obs, q.orig_at = (int)a_orig;
            a_img += (size_t)problems[b].n_img, a_pt += (size_t)problems[b].n_pt, a_ps += (size_t)problems[b].n_pt + 1, a_obs += (size_t)q.no;
            a_orig += problems[b].n_obs;
            SNK_REQUIRE(a_orig < (1ll << 31) && a_obs < ((size_t)1 << 31), "scene list too large (observations)");
        }
        pose.resize(7 * a_img), pt.resize(3 * a_pt), ptc.resize(a_pt), camidx.resize(a_img), ptstart.resize(a_ps);
        oimg.resize(a_obs), ocam.resize(a_obs), optfree.resize(a_obs), ouv2.resize(2 * a_obs), odepth.resize(a_obs), oweight.resize(a_obs);
        oorig.resize(a_obs), optidx.resize(a_obs);
    }
    auto fill_pass = [&](int b)
    {
        const snk_ba_problem& P = problems[b];
        PreProb& q = pre[(size_t)b];
        // values
        if (P.n_img) memcpy(pose.data() + 7 * q.img_at, &P.pose[0][0], (size_t)P.n_img * 7 * sizeof(double));
        if (P.n_pt) memcpy(pt.data() + 3 * q.pt_at, &P.pt[0][0], (size_t)P.n_pt * 3 * sizeof(double));
        for (int p = 0; p < P.n_pt; ++p) ptc[q.pt_at + (size_t)p] = P.pt_const[p] ? 1 : 0;
        // free cameras
        int* cidx = camidx.data() + q.img_at;
        int nfc   = 0;
        for (int i = 0; i < P.n_img; ++i) cidx[i] = P.img_const[i] ? -1 : nfc++;
        // valid observations, counting sort by point (stable: caller order inside a point)
        int* pstart = ptstart.data() + q.ps_at;
        for (int p = 0; p <= P.n_pt; ++p) pstart[p] = 0;
        std::vector<char> valid((size_t)P.n_obs, 0);
        for (int o = 0; o < P.n_obs; ++o)
        {
            const int i = P.obs_img[o], p = P.obs_pt[o];
            if (i < 0 || i >= P.n_img || p < 0 || p >= P.n_pt) continue;
            if (P.img_const[i] && P.pt_const[p]) continue;
            valid[(size_t)o] = 1;
            pstart[p + 1]++;
        }
        for (int p = 0; p < P.n_pt; ++p) pstart[p + 1] += pstart[p];
        // the sorted observation arrays, written in ONE pass over the caller's order: position = next free slot of the point
        const size_t obs_at = q.obs_at;
        int* q_img = oimg.data() + obs_at, *q_cam = ocam.data() + obs_at, *q_orig = oorig.data() + obs_at, *q_pt = optidx.data() + obs_at;
        unsigned char* q_free = optfree.data() + obs_at;
        double *q_uv = ouv2.data() + 2 * obs_at, *q_d = odepth.data() + obs_at, *q_w = oweight.data() + obs_at;
        // free cameras seen so far per point (device-built block entries: no camera twice on a point)
        const int seen_words = nfc <= BE_MAX_CAMS ? (nfc + 63) >> 6 : 0;
        std::vector<unsigned long long> seen((size_t)P.n_pt * (size_t)seen_words, 0ull);
        std::vector<int> fill(pstart, pstart + P.n_pt);
        for (int o = 0; o < P.n_obs; ++o)
        {
            if (!valid[(size_t)o]) continue;
            const int i = P.obs_img[o], p = P.obs_pt[o];
            const int sl = fill[(size_t)p]++;
            const int c  = cidx[i];
            if (c >= 0 && seen_words)
            {
                const unsigned long long bit = 1ull << (c & 63);
                unsigned long long& word     = seen[(size_t)p * (size_t)seen_words + (size_t)(c >> 6)];
                if (word & bit) q.dup = 1;
                word |= bit;
            }
            q_img[sl]  = i;
            q_cam[sl]  = c;
            q_free[sl] = P.pt_const[p] ? 0 : 1;
            q_uv[2 * sl]     = P.obs_uv[o][0];
            q_uv[2 * sl + 1] = P.obs_uv[o][1];
            q_d[sl]    = P.obs_depth[o];
            q_w[sl]    = P.obs_weight[o];
            q_orig[sl] = q.orig_at + o;
            q_pt[sl]   = p;
        }
    };
    // ---- batches: the observation arrays (0.65 of a batch's GB) go over the bus WHILE the lists are built (round 6): the fill pass runs in
    // four chunks of problems, every chunk's ranges of the arrays are sent as soon as they are written (the upload of a 1024-window batch is
    // ~15 ms of PCIe time that used to start when the last list was done), and pass 3 and the merge below build the rest meanwhile.
    // Everything sent here is final: the fill pass writes it in place and nothing below touches it.  o_cam, o_ptfree, o_pt and cam_items
    // are derived on the device (derive_obs_fields, derive_cam_items): reserved, not sent.
    const bool early_upload = count >= 16;
    static const int chunks_env = getenv("SNK_BA_FILL_CHUNKS") ? atoi(getenv("SNK_BA_FILL_CHUNKS")) : 0;  // A/B
    const int n_chunks      = early_upload && count >= 64 ? (chunks_env > 0 ? std::min(chunks_env, count / 16) : BA_FILL_CHUNKS) : 1;
    if (early_upload)
    {
        int rcE;
#define RSE(buf, vec, T) if ((rcE = h->buf.reserve(std::max<size_t>((vec).size(), 1) * sizeof(T))) != SNK_OK) return rcE
        RSE(d_pose, pose, double); RSE(d_pt, pt, double); RSE(d_ptc, ptc, unsigned char); RSE(d_camidx, camidx, int); RSE(d_ptstart, ptstart, int);
        RSE(d_oimg, oimg, int); RSE(d_ouv, ouv2, double); RSE(d_odepth, odepth, double); RSE(d_oweight, oweight, double); RSE(d_oorig, oorig, int);
        RSE(d_ocam, ocam, int); RSE(d_optfree, optfree, unsigned char); RSE(d_optidx, optidx, int);
#undef RSE
    }
    for (int ck = 0; ck < n_chunks; ++ck)
    {
        const int b0 = (int)((long long)count * ck / n_chunks), b1 = (int)((long long)count * (ck + 1) / n_chunks);
        pf_lo = b0, pf_hi = b1;
        parallel_for(fill_pass);
        pf_lo = 0, pf_hi = count;
        if (worker_failed.load())
        {
            set_error("snk_ba_set_problems: a list-building thread failed (out of host memory?)");
            return SNK_ERR_HIP;
        }
        if (!early_upload) continue;
        // this chunk's ranges of the arrays.  Range ends are rounded outwards to 16 bytes (the copy kernel moves 16-byte words): the few
        // bytes of a neighbouring chunk that go along are either final already or sent again, later on the same stream, by their own chunk
        CopyTab tabE;
        tabE.n = 0;
        auto part = [&](DevBuf& buf, const void* host, size_t elem, size_t e0, size_t e1, size_t total)
        {
            const size_t x0 = (e0 * elem) & ~(size_t)15, x1 = e1 >= total ? total * elem : std::min(total * elem, (e1 * elem + 15) & ~(size_t)15);
            if (x1 <= x0 || tabE.n >= COPY_TAB_MAX) return;
            tabE.src[tabE.n]   = static_cast<const char*>(host) + x0;
            tabE.dst[tabE.n]   = static_cast<char*>(buf.p) + x0;
            tabE.bytes[tabE.n] = (unsigned)(x1 - x0);
            ++tabE.n;
        };
        const size_t i0 = pre[(size_t)b0].img_at, p0 = pre[(size_t)b0].pt_at, s0 = pre[(size_t)b0].ps_at, o0 = pre[(size_t)b0].obs_at;
        const bool last = b1 >= count;
        const size_t i1 = last ? camidx.size() : pre[(size_t)b1].img_at, p1 = last ? ptc.size() : pre[(size_t)b1].pt_at,
                     s1 = last ? ptstart.size() : pre[(size_t)b1].ps_at, o1 = last ? oimg.size() : pre[(size_t)b1].obs_at;
        SNK_REQUIRE((o1 - o0 + 1) * 16 < (1ull << 32) && (p1 - p0 + 1) * 24 < (1ull << 32) && (i1 - i0 + 1) * 56 < (1ull << 32),
                    "scene list too large for the upload table");
        part(h->d_pose, pose.data(), 56, i0, i1, camidx.size());
        part(h->d_camidx, camidx.data(), 4, i0, i1, camidx.size());
        part(h->d_pt, pt.data(), 24, p0, p1, ptc.size());
        part(h->d_ptc, ptc.data(), 1, p0, p1, ptc.size());
        part(h->d_ptstart, ptstart.data(), 4, s0, s1, ptstart.size());
        part(h->d_oimg, oimg.data(), 4, o0, o1, oimg.size());
        part(h->d_ouv, ouv2.data(), 16, o0, o1, oimg.size());
        part(h->d_odepth, odepth.data(), 8, o0, o1, oimg.size());
        part(h->d_oweight, oweight.data(), 8, o0, o1, oimg.size());
        part(h->d_oorig, oorig.data(), 4, o0, o1, oimg.size());
        if (tabE.n > 0)
        {
            unsigned big = 0;
            for (int e = 0; e < tabE.n; ++e) big = std::max(big, tabE.bytes[e]);
            const int gxe = (int)std::min(256u, std::max(16u, big >> 16));
            hipLaunchKernelGGL(copy_table_kernel, dim3(gxe, tabE.n), dim3(256), 0, h->stream, tabE);
            SNK_LAUNCH_CHECK();
        }
    }
    mark(0);
    mark(1);
    // ---- pass 3: the camera lists and the point-major lists of every problem, built with PROBLEM-LOCAL offsets on the host threads; the
    // per-problem loop below appends them to the shared lists and relocates the offsets (positions in setpts / setpairs / cblkitems / ccitems,
    // partial-sum, camera-partial and record indices) by the running totals -- the same lists the serial builder wrote ----
    struct Built
    {
        std::vector<int> camstart, camitems;
        int be_nch          = 0;
        long long ent_bound = 0;
        bool ok = false, cam_sums_bad = false;
        std::vector<SetItem> items;
        std::vector<int2> ipts;
        std::vector<int> ipairs;
        int parts = 0, cparts = 0;
        long long recs = 0;
        int max_pairs = 0, max_run = 0, max_k = 0;
        std::vector<int> cblkstart, cblkitems, ccstart, ccitems;
        std::vector<int> wv;  // point_wave work items (first point of each, then n_pt); empty: a point has more than 64 observations
        bool wv_ok = false;
    };
    // where the per-problem loop below puts a problem's lists in the shared ones: the loop only takes the decisions and does the
    // arithmetic of the running totals; the element copies (with their relocations) run on the host threads afterwards (round 6: the
    // loop's push_back relocations were ~6 of a 1024-window hand-over's 23 ms of list time)
    struct MergeAt
    {
        size_t wvpt, camstart, camitems, ccstart, ccitems, setitems, setpts, setpairs, cblkstart, cblkitems;
        int base_parts, base_cparts;
        long long base_rec;
        bool set;
    };
    std::vector<MergeAt> at((size_t)count);
    size_t n_wvpt = 0, n_camstart = 0, n_camitems = 0, n_ccstart = 0, n_ccitems = 0, n_setitems = 0, n_setpts = 0, n_setpairs = 0, n_cblkstart = 0,
           n_cblkitems = 0;
    std::vector<Built> built((size_t)count);
    parallel_for([&](int b)
    {
        const snk_ba_problem& P = problems[b];
        const PreProb& pq       = pre[(size_t)b];
        Built& B                = built[(size_t)b];
        const int nfc = pq.nfc, no = pq.no;
        const int* const pstart = ptstart.data() + pq.ps_at;
        const int* const s_cam  = ocam.data() + pq.obs_at;
        // camera lists
        {
            std::vector<int> cs((size_t)nfc + 1, 0);
            for (int s = 0; s < no; ++s)
                if (s_cam[(size_t)s] >= 0) cs[(size_t)s_cam[(size_t)s] + 1]++;
            for (int c = 0; c < nfc; ++c) cs[(size_t)c + 1] += cs[(size_t)c];
            std::vector<int> items((size_t)cs[(size_t)nfc]);
            std::vector<int> fill(cs.begin(), cs.end() - 1);
            for (int s = 0; s < no; ++s)
                if (s_cam[(size_t)s] >= 0) items[(size_t)fill[(size_t)s_cam[(size_t)s]]++] = s;
            B.camstart.assign(cs.begin(), cs.end());
            B.camitems.swap(items);
            
            int longest = 0;
            for (int c = 0; c < nfc; ++c) longest = std::max(longest, cs[(size_t)c + 1] - cs[(size_t)c]);
            B.be_nch = ceil_div(longest, 64);
            // (the same list as static records -- what cam_pass streams -- is gathered on the device: gather_cam_records)
        }
        {
            // room for the block entries when the device builds them: every pair of a point's run is the most there can be
            long long bound = 0;
            for (int p = 0; p < P.n_pt; ++p)
            {
                const long long run = pstart[(size_t)p + 1] - pstart[(size_t)p];
                bound += run * run;
            }
            B.ent_bound = bound;
        }
        // point_wave work items: consecutive whole points with <= 64 observations in total
        {
            bool ok = true;
            std::vector<int>& wv = B.wv;
            int p = 0;
            while (p < P.n_pt && ok)
            {
                wv.push_back(p);
                int n = 0, q = p;
                while (q < P.n_pt && q - p < 64 && n + (pstart[(size_t)q + 1] - pstart[(size_t)q]) <= 64)
                {
                    n += pstart[(size_t)q + 1] - pstart[(size_t)q];
                    ++q;
                }
                if (q == p) ok = false;  // a point with more than 64 observations: point_pass handles the problem
                p = q;
            }
            if (ok) wv.push_back(P.n_pt);
            else wv.clear();
            B.wv_ok = ok;
        }
        // point-major Schur pass: points grouped by camera set, work items of <= SET_CHUNK points, per-block lists of
        // the partial sums they produce
        {
            const size_t nb = (size_t)nfc * nfc;
            // camera set -> group: a hash of the signature finds the candidate, the stored signature confirms it (a std::map keyed by
            // the vectors themselves was 60 ns per point, a third of a batch hand-over's list time)
            std::unordered_multimap<unsigned long long, int> gid;
            std::vector<std::vector<int>> gpts;
            std::vector<std::vector<int>> gsig;
            auto find_group = [&](const std::vector<int>& key) -> int
            {
                unsigned long long hsh = 1469598103934665603ull;
                for (int v : key) hsh = (hsh ^ (unsigned long long)(unsigned)v) * 1099511628211ull;
                auto range = gid.equal_range(hsh);
                for (auto it = range.first; it != range.second; ++it)
                    if (gsig[(size_t)it->second] == key) return it->second;
                gid.emplace(hsh, (int)gpts.size());
                gpts.emplace_back();
                gsig.push_back(key);
                return (int)gpts.size() - 1;
            };
            std::vector<int> sig;
            // The point-major kernels (schur_fused / schur_mfma / update_cost) are only chosen when the launch has enough work
            // items (max_set_items * count >= SNK_BA_SCHUR_SET_MIN_ITEMS, default 256): for the reference's per-keyframe
            // call -- ONE window of a few thousand points -- their lists are never used, and building + uploading them
            // (0.8 MB of records alone) was a quarter of the 0.9 ms a scene hand-over cost.  Built for batches and for big
            // single scenes (global BA); forced when the threshold is lowered by the environment (tests).
            static const bool sets_forced = getenv("SNK_BA_SCHUR_SET_MIN_ITEMS") != nullptr;
            bool ok = nfc > 0 && (count >= 8 || P.n_pt >= 8000 || sets_forced);
            // points that produce no Schur products (constant points, points seen by constant cameras only) still need their
            // linearisation (cost, V, b_p): they form groups of their own, keyed by their run length, with no pairs
            std::vector<int> plain_key;
            auto plain_group = [&](int p, int run)
            {
                // (a signature of `run` times -1 cannot be a set with free cameras: the plain group of that run length)
                plain_key.assign((size_t)run, -1);
                gpts[(size_t)find_group(plain_key)].push_back(p);
            };
            for (int p = 0; p < P.n_pt && ok; ++p)
            {
                const int a0 = pstart[(size_t)p], a1 = pstart[(size_t)p + 1];
                if (a1 - a0 > SET_MAX_RUN) ok = false;
                if (a1 == a0) continue;  // a point without observations: nothing to linearise (update_wave keeps it in place)
                if (P.pt_const[p])
                {
                    for (int a = a0; a < a1; ++a)
                        if (s_cam[(size_t)a] >= 0) B.cam_sums_bad = true;
                    plain_group(p, a1 - a0);
                    continue;
                }
                sig.clear();
                int k = 0;
                for (int a = a0; a < a1; ++a)
                {
                    const int c = s_cam[(size_t)a];
                    sig.push_back(c);
                    if (c < 0) continue;
                    ++k;
                    for (int b = a0; b < a; ++b)
                        if (s_cam[(size_t)b] == c) ok = false;  // one camera twice on a point: block-major pass only
                }
                if (k == 0)
                {
                    plain_group(p, a1 - a0);
                    continue;
                }
                if (k > SET_MAX_K || a1 - a0 > SET_MAX_RUN) ok = false;
                gpts[(size_t)find_group(sig)].push_back(p);
            }
            std::vector<std::vector<int>> contrib(nb);
            std::vector<std::vector<int>> ccontrib((size_t)nfc);  // per free camera: its partial sums in cam_part
            int cparts = 0;
            long long recs = 0;  // static observation records of the work items (gathered on the device: gather_set_records)
            std::vector<SetItem> items;
            std::vector<int2> ipts;
            std::vector<int> ipairs;
            int parts = 0;
            if (ok)
            {
                // groups in order of their first point (std::map order would do as well: any fixed order)
                for (size_t g = 0; g < gpts.size(); ++g)
                {
                    const std::vector<int>& sig = gsig[g];
                    const int pair_off = (int)((size_t)0 + ipairs.size());
                    std::vector<int> blocks;
                    for (size_t i = 0; i < sig.size(); ++i)
                        for (size_t j = i; j < sig.size(); ++j)
                        {
                            if (sig[i] < 0 || sig[j] < 0) continue;
                            const bool sw = sig[i] > sig[j];
                            const int ra = (int)(sw ? j : i), rb = (int)(sw ? i : j);
                            ipairs.push_back(ra | (rb << 8));
                            blocks.push_back(sig[(size_t)ra] * nfc + sig[(size_t)rb]);
                        }
                    const int npairs = (int)blocks.size();
                    // matrix-core form (schur_mfma): the point's free rows ordered by camera index, so that every pair
                    // (ra, rb) -- camera(ra) < camera(rb) -- lies in the upper triangle of Y W^T, and the slot of each
                    const int aux_off = (int)((size_t)0 + ipairs.size());
                    std::vector<int> fcams;  // the set's free cameras in ascending order (= the order of the k run positions)
                    {
                        std::vector<int> fpos;
                        for (size_t i = 0; i < sig.size(); ++i)
                            if (sig[i] >= 0) fpos.push_back((int)i);
                        std::sort(fpos.begin(), fpos.end(), [&](int a, int b) { return sig[(size_t)a] < sig[(size_t)b]; });
                        const int kf = (int)fpos.size();
                        for (int v : fpos) ipairs.push_back(v);
                        for (int v : fpos) fcams.push_back(sig[(size_t)v]);
                        for (int i = 0; i < kf; ++i)
                            for (int j = 0; j < kf; ++j)
                            {
                                int slot = -1;
                                if (i <= j)
                                    for (int q = 0; q < npairs; ++q)
                                        if (ipairs[(size_t)(pair_off - (int)(size_t)0) + (size_t)q] == (fpos[(size_t)i] | (fpos[(size_t)j] << 8))) slot = q;
                                ipairs.push_back(slot);
                            }
                        B.max_k = std::max(B.max_k, kf);
                    }
                    const size_t chunk    = big_items ? SET_CHUNK_BIG : SET_CHUNK;
                    const size_t n_in_set = gpts[g].size(), n_cuts = (n_in_set + chunk - 1) / chunk;
                    size_t cut = (n_in_set + n_cuts - 1) / n_cuts;  // equal items: a launch ends with its longest item
                    {
                        // schur_fused linearises 64 / run points at a time: whole groups of that many per item where possible
                        const size_t grp = std::min<size_t>(64 / std::max<size_t>(sig.size(), 1), 16);  // SF_GMAX
                        cut = std::min<size_t>((cut + grp - 1) / grp * grp, big_items ? 128 : 64);
                    }
                    for (size_t q0 = 0; q0 < n_in_set; q0 += cut)
                    {
                        SetItem si;
                        si.pts_off  = (int)((size_t)0 + ipts.size());
                        si.n_pts    = (int)std::min<size_t>(cut, n_in_set - q0);
                        si.pair_off = pair_off;
                        si.npairs   = npairs;
                        si.part_off = parts;
                        si.run      = (int)sig.size();
                        si.aux_off  = aux_off;
                        si.nfree    = 0;
                        si.rec_off  = (int)recs;
                        for (int v : sig) si.nfree += v >= 0 ? 1 : 0;
                        si.cpart_off = cparts;
                        for (int f = 0; f < si.nfree; ++f) ccontrib[(size_t)fcams[(size_t)f]].push_back(cparts + f);
                        cparts += si.nfree;
                        for (int q = 0; q < si.n_pts; ++q)
                        {
                            const int pp = gpts[g][q0 + (size_t)q];
                            ipts.push_back(make_int2(pp, pstart[(size_t)pp]));
                        }
                        recs += (long long)si.n_pts * si.run;
                        if (recs >= (1ll << 31)) ok = false;  // (would not be addressable by rec_off: the block-major pass then)
                        for (int q = 0; q < npairs; ++q) contrib[(size_t)blocks[(size_t)q]].push_back(parts + q);
                        parts += npairs;
                        items.push_back(si);
                        B.max_pairs = std::max(B.max_pairs, npairs);
                        B.max_run   = std::max(B.max_run, si.run);
                    }
                }
            }
            {
                int crun = 0;
                for (int c = 0; c < nfc; ++c)
                {
                    B.ccstart.push_back(crun);
                    if (ok)
                    {
                        B.ccitems.insert(B.ccitems.end(), ccontrib[(size_t)c].begin(), ccontrib[(size_t)c].end());
                        crun += (int)ccontrib[(size_t)c].size();
                    }
                }
                B.ccstart.push_back(crun);
            }
            B.ok = ok;
            if (ok)
            {
                B.items.swap(items);
                B.ipts.swap(ipts);
                B.ipairs.swap(ipairs);
                B.parts  = parts;
                B.cparts = cparts;
                B.recs   = recs;
            }
            int run = 0;
            B.cblkstart.resize(nb + 1);
            int* cb = B.cblkstart.data();
            for (size_t k = 0; k < nb; ++k)
            {
                cb[k] = run;
                if (ok && !contrib[k].empty())
                {
                    B.cblkitems.insert(B.cblkitems.end(), contrib[k].begin(), contrib[k].end());
                    run += (int)contrib[k].size();
                }
            }
            cb[nb] = run;
        }
    });
    if (worker_failed.load())
    {
        set_error("snk_ba_set_problems: a list-building thread failed (out of host memory?)");
        return SNK_ERR_HIP;
    }
    mark(3);
    for (int b = 0; b < count; ++b)
    {
        const snk_ba_problem& P = problems[b];
        mark(7);
        SNK_REQUIRE(P.n_img >= 0 && P.n_pt >= 0 && P.n_obs >= 0, "negative problem size");
        SNK_REQUIRE(P.n_img == 0 || (P.pose && P.img_const), "NULL pose arrays");
        SNK_REQUIRE(P.n_pt == 0 || (P.pt && P.pt_const), "NULL point arrays");
        SNK_REQUIRE(P.n_obs == 0 || (P.obs_img && P.obs_pt && P.obs_uv && P.obs_depth && P.obs_weight), "NULL observation arrays");
        SNK_REQUIRE(P.n_rpc >= 0 && (P.n_rpc == 0 || P.rpc != nullptr), "bad relative pose constraints");
        Prob& pr = probs[(size_t)b];
        memset(&pr, 0, sizeof(pr));
        pr.ni = P.n_img;
        pr.np = P.n_pt;
        for (int k = 0; k < 4; ++k) pr.K[k] = P.K[k];
        pr.bf       = P.bf;
        pr.img_off  = img_off;
        pr.pt_off   = pt_off;
        pr.obs_off  = obs_off;
        pr.cam_off  = cam_off;
        pr.orig_off = orig_off;
        pr.vec_off  = vec_off;
        pr.s_off    = s_off;
        h->orig_off[(size_t)b] = orig_off;
        h->orig_n[(size_t)b]   = P.n_obs;
        // values, free-camera indices, counting sort and the sorted observation arrays: written by the fill pass above
        const PreProb& pq = pre[(size_t)b];
        const int nfc     = pq.nfc;
        const int* const cidx = camidx.data() + pq.img_at;
        pr.nfc = nfc;
        pr.n6  = 6 * nfc;
        const int no = pq.no;
        pr.no        = no;
        pr.ptstart_off = (int)pq.ps_at;
        if (pq.dup) dev_entries_ok = false;
        const Built& B = built[(size_t)b];
        MergeAt& M     = at[(size_t)b];
        // point_wave work items (built in pass 3)
        pr.wv_off = (int)n_wvpt;
        pr.n_wv   = 0;
        M.wvpt    = n_wvpt;
        if (B.wv_ok)
        {
            pr.n_wv = (int)B.wv.size() - 1;
            n_wvpt += B.wv.size();
            max_wv = std::max(max_wv, pr.n_wv);
        }
        else
            wave_ok = false;
        mark(2);
        // camera lists (built in pass 3; positions and items are problem-local: appended as they are)
        pr.camstart_off = (int)n_camstart;
        pr.citem_off    = (int)n_camitems;
        M.camstart = n_camstart, M.camitems = n_camitems;
        n_camstart += B.camstart.size();
        n_camitems += B.camitems.size();
        max_citems = std::max(max_citems, (int)B.camitems.size());
        pr.be_nch  = B.be_nch;
        if (nfc > BE_MAX_CAMS) dev_entries_ok = false;
        ent_bound[(size_t)b] = B.ent_bound;
        pr.blkstart_off = blkstart_total;
        blkstart_total += nfc * nfc + 1;
        mark(3);
        mark(4);
        // point-major Schur pass (built in pass 3 with problem-local offsets): append, relocating by the running totals
        pr.set_off  = (int)n_setitems;
        pr.cblk_off = (int)n_cblkstart;
        pr.n_set    = 0;
        {
            if (B.cam_sums_bad) cam_sums_ok = false;
            max_set_k     = std::max(max_set_k, B.max_k);
            max_set_pairs = std::max(max_set_pairs, B.max_pairs);
            max_set_run   = std::max(max_set_run, B.max_run);
            pr.ccam_off = (int)n_ccstart;
            M.ccstart = n_ccstart, M.ccitems = n_ccitems;  // ccstart entries + base_cc (= M.ccitems), ccitems entries + base_cparts
            n_ccstart += B.ccstart.size();
            n_ccitems += B.ccitems.size();
            M.base_parts = n_partials, M.base_cparts = n_cparts, M.base_rec = n_setrec;
            // the batch's record / partial-sum counters are 32-bit on the device: a batch that would overflow them keeps the block-major pass
            // (what the serial builder of round 3 did), it is not an error
            const bool set_fits = n_setrec + B.recs < (1ll << 31) && (long long)n_partials + B.parts < (1ll << 31);
            M.set = B.ok && set_fits;
            M.setitems = n_setitems, M.setpts = n_setpts, M.setpairs = n_setpairs;
            if (M.set)
            {
                n_setitems += B.items.size();
                n_setpts += B.ipts.size();
                n_setpairs += B.ipairs.size();
                pr.n_set      = (int)B.items.size();
                max_set_items = std::max(max_set_items, pr.n_set);
            }
            else
                set_ok = false;
            M.cblkstart = n_cblkstart, M.cblkitems = n_cblkitems;  // cblkstart entries + base_cb (= M.cblkitems), cblkitems entries + base_parts
            n_cblkstart += B.cblkstart.size();
            n_cblkitems += B.cblkitems.size();
            if (M.set)
            {
                n_partials += B.parts;
                n_cparts += B.cparts;
                n_setrec += B.recs;
            }
        }
        mark(5);
        // relative pose constraints (IMU scenes): valid ones, per-camera incidence, per-block chains
        {
            pr.rpc_off    = (int)rpcmeta.size();
            pr.camrpc_off = (int)camrpcstart.size();
            std::vector<int> cs((size_t)nfc + 1, 0);
            std::vector<RpcMeta> mine;
            for (int k = 0; k < P.n_rpc; ++k)
            {
                const snk_ba_rpc& q = P.rpc[k];
                if (q.img1 < 0 || q.img2 < 0 || q.img1 >= P.n_img || q.img2 >= P.n_img || q.img1 == q.img2) continue;
                if (P.img_const[q.img1] && P.img_const[q.img2]) continue;
                SNK_REQUIRE(q.weight_rotation >= 0.0 && q.weight_translation >= 0.0, "negative constraint weight");
                RpcMeta m;
                m.img1 = q.img1; m.img2 = q.img2;
                m.c1 = cidx[(size_t)q.img1]; m.c2 = cidx[(size_t)q.img2];
                for (int t = 0; t < 7; ++t) m.rel[t] = q.rel_pose[t];
                m.w_rot = q.weight_rotation; m.w_trans = q.weight_translation;
                mine.push_back(m);
                if (m.c1 >= 0) cs[(size_t)m.c1 + 1]++;
                if (m.c2 >= 0) cs[(size_t)m.c2 + 1]++;
            }
            pr.n_rpc = (int)mine.size();
            max_rpc  = std::max(max_rpc, pr.n_rpc);
            for (int c = 0; c < nfc; ++c) cs[(size_t)c + 1] += cs[(size_t)c];
            const int item_base = (int)camrpcitems.size();
            std::vector<int> items((size_t)cs[(size_t)nfc]), fill(cs.begin(), cs.end() - 1);
            // the per-block chains are only read for problems that HAVE constraints: the others advance the offset and write nothing
            std::vector<int> brpc(mine.empty() ? 0 : (size_t)nfc * nfc, 0), nxt(mine.size(), 0);
            for (int k = 0; k < (int)mine.size(); ++k)
            {
                const RpcMeta& m = mine[(size_t)k];
                if (m.c1 >= 0) items[(size_t)fill[(size_t)m.c1]++] = k * 2;
                if (m.c2 >= 0) items[(size_t)fill[(size_t)m.c2]++] = k * 2 + 1;
                if (m.c1 >= 0 && m.c2 >= 0)
                {
                    // the upper block (lo, hi) holds J(lo)^T J(hi): H12 when img1 is `lo`, its transpose otherwise
                    const int lo = std::min(m.c1, m.c2), hi = std::max(m.c1, m.c2);
                    const int code = 1 + (k * 2 + (m.c1 == lo ? 0 : 1));
                    nxt[(size_t)k]                 = brpc[(size_t)lo * nfc + hi];
                    brpc[(size_t)lo * nfc + hi] = code;
                }
            }
            for (int c = 0; c <= nfc; ++c) camrpcstart.push_back(item_base + cs[(size_t)c]);
            camrpcitems.insert(camrpcitems.end(), items.begin(), items.end());
            if (!mine.empty())
            {
                blkrpc.resize(blkrpc_logical, 0);  // zeros for the problems without constraints in front of this one
                blkrpc.insert(blkrpc.end(), brpc.begin(), brpc.end());
            }
            blkrpc_logical += (size_t)nfc * nfc;
            rpcnext.insert(rpcnext.end(), nxt.begin(), nxt.end());
            rpcmeta.insert(rpcmeta.end(), mine.begin(), mine.end());
        }
        mark(6);
        img_off += P.n_img;
        pt_off += P.n_pt;
        obs_off += no;
        cam_off += nfc;
        orig_off += P.n_obs;
        vec_off += pr.n6;
        s_off += (long long)pr.n6 * pr.n6;
        max_np  = std::max(max_np, P.n_pt);
        max_ni  = std::max(max_ni, P.n_img);
        max_nfc = std::max(max_nfc, nfc);
        max_n6  = std::max(max_n6, pr.n6);
    }
    // ---- the element copies the loop above left out: every problem's lists into its ranges of the shared lists, relocated, on the host
    // threads (disjoint ranges, no locks) ----
    wvpt.resize(n_wvpt), camstart.resize(n_camstart), camitems.resize(n_camitems), ccstart.resize(n_ccstart), ccitems.resize(n_ccitems);
    setitems.resize(n_setitems), setpts.resize(n_setpts), setpairs.resize(n_setpairs), cblkstart.resize(n_cblkstart), cblkitems.resize(n_cblkitems);
    parallel_for([&](int b)
    {
        const Built& B   = built[(size_t)b];
        const MergeAt& M = at[(size_t)b];
        auto put = [](auto& dst, size_t pos, const auto& src)
        {
            if (!src.empty()) memcpy(dst.data() + pos, src.data(), src.size() * sizeof(src[0]));
        };
        if (B.wv_ok) put(wvpt, M.wvpt, B.wv);
        put(camstart, M.camstart, B.camstart);
        put(camitems, M.camitems, B.camitems);
        {
            int* d = ccstart.data() + M.ccstart;
            for (size_t k = 0; k < B.ccstart.size(); ++k) d[k] = B.ccstart[k] + (int)M.ccitems;
            d = ccitems.data() + M.ccitems;
            for (size_t k = 0; k < B.ccitems.size(); ++k) d[k] = B.ccitems[k] + M.base_cparts;
            d = cblkstart.data() + M.cblkstart;
            for (size_t k = 0; k < B.cblkstart.size(); ++k) d[k] = B.cblkstart[k] + (int)M.cblkitems;
            d = cblkitems.data() + M.cblkitems;
            for (size_t k = 0; k < B.cblkitems.size(); ++k) d[k] = B.cblkitems[k] + M.base_parts;
        }
        if (M.set)
        {
            SetItem* d = setitems.data() + M.setitems;
            for (size_t k = 0; k < B.items.size(); ++k)
            {
                SetItem si = B.items[k];
                si.pts_off += (int)M.setpts;
                si.pair_off += (int)M.setpairs;
                si.aux_off += (int)M.setpairs;
                si.part_off += M.base_parts;
                si.cpart_off += M.base_cparts;
                si.rec_off += (int)M.base_rec;
                d[k] = si;
            }
            put(setpts, M.setpts, B.ipts);
            put(setpairs, M.setpairs, B.ipairs);
        }
    });
    if (worker_failed.load())
    {
        set_error("snk_ba_set_problems: a list-building thread failed (out of host memory?)");
        return SNK_ERR_HIP;
    }
    const size_t pcg_lds = (size_t)max_n6 * 9 * 8 + (size_t)max_nfc * 36 * 8;
    // S (and the vectors) of the largest problem fit one workgroup's LDS -> one workgroup per problem;
    // otherwise the multi-workgroup PCG (measured: 120 keyframes 38 ms -> 9 ms, 600 keyframes 21 ms)
    h->pcg_large = pcg_lds + (size_t)max_n6 * max_n6 * 8 > 158 * 1024;
    h->probs.assign(probs.begin(), probs.end());
    h->count = count;
    h->tot_img = img_off; h->tot_pt = pt_off; h->tot_obs = obs_off; h->tot_cam = cam_off; h->tot_orig = orig_off;
    h->tot_vec = vec_off; h->tot_s = s_off;
    h->max_np = max_np; h->max_nfc = max_nfc; h->max_n6 = max_n6; h->max_ni = max_ni;
    h->max_wv = max_wv;
    h->max_rpc = max_rpc;
    h->point_wave_ok = wave_ok && max_wv > 0;

    // block entries: on the device when every problem qualifies, by the host builder otherwise
    static const bool host_entries = getenv("SNK_BA_HOST_ENTRIES") != nullptr;  // A/B and tests
    bool dev_entries = dev_entries_ok && !host_entries;
    if (dev_entries)
    {
        // The device builder counts into nfc x chunks x nfc ints per problem -- quadratic in the free cameras.  A global BA with
        // ~500 free cameras and one long camera list needs hundreds of megabytes of counters the host builder never allocates:
        // beyond a modest budget (64 MB; a batch of 1024 local windows needs 1.6 MB) the host builder takes over.
        long long cnt = 0;
        for (int b = 0; b < count; ++b) cnt += (long long)probs[(size_t)b].nfc * probs[(size_t)b].be_nch * probs[(size_t)b].nfc;
        static const long long budget = getenv("SNK_BA_BECNT_BUDGET") ? atoll(getenv("SNK_BA_BECNT_BUDGET")) : (64ll << 20);  // bytes; tests force the fallback
        if (cnt * (long long)sizeof(int) > budget) dev_entries = false;
    }
    long long ent_total = 0, becnt_total = 0;
    int max_be_waves = 0;
    if (dev_entries)
    {
        for (int b = 0; b < count; ++b)
        {
            Prob& pr = probs[(size_t)b];
            SNK_REQUIRE(ent_total + ent_bound[(size_t)b] < (1ll << 31), "scene list too large (block entries)");
            pr.ent_off   = (int)ent_total;
            pr.becnt_off = (int)becnt_total;
            ent_total += ent_bound[(size_t)b];
            becnt_total += (long long)pr.nfc * pr.be_nch * pr.nfc;
            SNK_REQUIRE(becnt_total < (1ll << 31), "scene list too large (block entry counters)");
            max_be_waves = std::max(max_be_waves, pr.nfc * pr.be_nch);
        }
    }
    else
    {
        long long bound = 0;
        for (int b = 0; b < count; ++b) bound += ent_bound[(size_t)b];
        LS.blkent.reserve((size_t)bound);  // one pinned allocation instead of a doubling chain
        LS.blkstart.reserve((size_t)blkstart_total);
        for (int b = 0; b < count; ++b)
        {
            probs[(size_t)b].be_nch  = 0;
            probs[(size_t)b].ent_off = (int)blkent.size();
            host_block_entries(b, LS.blkstart, LS.blkent);
        }
    }
    h->probs.assign(probs.begin(), probs.end());  // again: with the block-entry offsets
    mark(4);
    int rc;
    hipStream_t st = h->stream;
    const auto t_lists = std::chrono::steady_clock::now();
    CopyTab tab;
    tab.n = 0;
#define UP(buf, vec) if ((rc = upload(h->buf, vec, tab)) != SNK_OK) return rc
    UP(d_prob, probs);
    // batches: the second and third copies of the poses / points (the reset state, the trial points) are device-to-device copies behind
    // the upload instead of two more trips over the bus (round 6: 100 MB of a 1024-window hand-over's 1.1 GB); a single window keeps the
    // one launch
    const bool dup_on_device = count >= 16;
    CopyTab tab2;
    tab2.n = 0;
    auto dup = [&](DevBuf& dst, const DevBuf& src, size_t bytes) -> int
    {
        int rc2 = dst.reserve(std::max<size_t>(bytes, 1));
        if (rc2 != SNK_OK || bytes == 0) return rc2;
        SNK_REQUIRE(tab2.n < COPY_TAB_MAX && bytes < (1ull << 32), "scene list too large for the upload table");
        tab2.src[tab2.n] = src.p, tab2.dst[tab2.n] = dst.p, tab2.bytes[tab2.n] = (unsigned)bytes;
        ++tab2.n;
        return SNK_OK;
    };
    if (!early_upload) { UP(d_pose, pose); }
    if (!dup_on_device) { UP(d_pose0, pose); }
    else if ((rc = dup(h->d_pose0, h->d_pose, pose.size() * sizeof(double))) != SNK_OK) return rc;
    if (!early_upload) { UP(d_pt, pt); }
    if (!dup_on_device) { UP(d_pt0, pt); }
    else if ((rc = dup(h->d_pt0, h->d_pt, pt.size() * sizeof(double))) != SNK_OK) return rc;
    if (!early_upload)
    {
        UP(d_ptc, ptc);
        UP(d_camidx, camidx);
        UP(d_ptstart, ptstart);
        UP(d_oimg, oimg);
        UP(d_ocam, ocam);
        UP(d_optfree, optfree);
        UP(d_ouv, ouv2);
        UP(d_odepth, odepth);
        UP(d_oweight, oweight);
        UP(d_oorig, oorig);
    }
    UP(d_camstart, camstart);
    if (!early_upload) { UP(d_camitems, camitems); }
    else if ((rc = h->d_camitems.reserve(std::max<size_t>(camitems.size(), 1) * sizeof(int))) != SNK_OK) return rc;
    if ((rc = h->d_csobs.reserve(std::max<size_t>(camitems.size(), 1) * sizeof(CamObs))) != SNK_OK) return rc;  // gather_cam_records
    UP(d_setitems, setitems);
    if ((rc = h->d_setobs.reserve((size_t)std::max<long long>(n_setrec, 1) * sizeof(SetObs))) != SNK_OK) return rc;  // gather_set_records
    UP(d_setpts, setpts);
    UP(d_setpairs, setpairs);
    UP(d_cblkstart, cblkstart);
    UP(d_cblkitems, cblkitems);
    UP(d_ccstart, LS.ccstart);
    UP(d_ccitems, LS.ccitems);
    if (dev_entries)
    {
        if ((rc = h->d_blkstart.reserve((size_t)std::max(blkstart_total, 1) * sizeof(int))) != SNK_OK) return rc;
        if ((rc = h->d_blkent.reserve((size_t)std::max<long long>(ent_total, 1) * sizeof(int4))) != SNK_OK) return rc;
        if ((rc = h->d_becnt.reserve((size_t)std::max<long long>(becnt_total, 1) * sizeof(int))) != SNK_OK) return rc;
    }
    else
    {
        UP(d_blkstart, blkstart);
        UP(d_blkent, blkent);
    }
    if (!early_upload) { UP(d_optidx, optidx); }
    UP(d_wvpt, wvpt);
    UP(d_rpcmeta, rpcmeta);
    UP(d_rpcnext, rpcnext);
    UP(d_camrpcstart, camrpcstart);
    UP(d_camrpcitems, camrpcitems);
    UP(d_blkrpc, blkrpc);
    if (!dup_on_device) { UP(d_pt_new, pt); }  // points without observations stay put
    else if ((rc = dup(h->d_pt_new, h->d_pt, pt.size() * sizeof(double))) != SNK_OK) return rc;
    const auto t_up = std::chrono::steady_clock::now();
    const size_t nobs = (size_t)std::max(obs_off, 1), npt = (size_t)std::max(pt_off, 1);
#define RS(buf, bytes) if ((rc = h->buf.reserve(bytes)) != SNK_OK) return rc
    RS(d_state, (size_t)count * sizeof(State));
    RS(d_pose_new, (size_t)std::max(img_off, 1) * 7 * 8);
    RS(d_pt_new, npt * 3 * 8);
    RS(d_outlier, (size_t)std::max(orig_off, 1));
    RS(d_chi2, (size_t)std::max(orig_off, 1) * 8);
    RS(d_r, nobs * 4 * 8);
    RS(d_W, nobs * 18 * 8);
    RS(d_ptv, npt * 6 * 8);
    RS(d_spart, (size_t)std::max(n_partials, 1) * 36 * 8);
    RS(d_campart, (size_t)std::max(n_cparts, 1) * CS_TERMS * 8);
    h->cam_sums_ok = cam_sums_ok;
    h->set_ok = set_ok && max_set_items > 0;
    h->max_set_items = max_set_items;
    h->set_small     = max_set_pairs * 6 <= 4 * 64 && max_set_run * 9 + 3 <= 2 * 64;
    h->set_k_max     = max_set_k;
    h->set_run_max   = max_set_run;
    RS(d_Vinv, npt * 6 * 8);
    RS(d_bp, npt * 3 * 8);
    RS(d_cost, npt * 8);
    RS(d_cost_new, npt * 8);
    RS(d_U, (size_t)std::max(cam_off, 1) * 36 * 8);
    RS(d_S, (size_t)std::max<long long>(s_off, 1) * 8);
    RS(d_rhs, (size_t)std::max(vec_off, 1) * 8);
    RS(d_x, (size_t)std::max(vec_off, 1) * 8);
    RS(d_rpcout, std::max<size_t>(rpcmeta.size(), 1) * RPC_STRIDE * 8);
    if (h->pcg_large)
    {
        PcgLarge& W = h->pcgw;
        W.G         = std::max(1, ceil_div(max_nfc, 64));
        W.B         = count;
        W.tot_vec   = vec_off;
        // enough (row chunk x column part) workgroups to fill 256 CUs a few times over
        const int rowchunks = std::max(1, ceil_div(max_n6, 256));
        W.parts             = std::min(64, std::max(1, ceil_div(1024, rowchunks * count)));
        const size_t nv = (size_t)std::max(vec_off, 1), ng = (size_t)count * W.G;
        const size_t doubles = 4 * nv + (size_t)std::max(cam_off, 1) * 36 + (size_t)std::max(W.parts, PERSIST_WGS_MAX / rowchunks + 1) * nv + 5 * ng + (size_t)count * 4 +
                               nv + 6 * (size_t)PERSIST_WGS_MAX + 8;
        RS(d_pcgw, doubles * 8);
        double* w = h->d_pcgw.as<double>();
        W.r = w;            w += nv;
        W.z = w;            w += nv;
        W.p = w;            w += nv;
        W.Ap = w;           w += nv;
        W.Minv = w;         w += (size_t)std::max(cam_off, 1) * 36;
        W.ps = w;           w += (size_t)std::max(W.parts, PERSIST_WGS_MAX / rowchunks + 1) * nv;
        W.prr = w;          w += 2 * ng;
        W.prz = w;          w += 2 * ng;
        W.ppap = w;         w += ng;
        W.scal = w;         w += (size_t)count * 4;
        W.p2 = w;           w += nv;
        W.wrr = w;          w += 2 * (size_t)PERSIST_WGS_MAX;
        W.wrz = w;          w += 2 * (size_t)PERSIST_WGS_MAX;
        W.wpap = w;         w += (size_t)PERSIST_WGS_MAX;
        W.bar = reinterpret_cast<unsigned*>(w);  // (PERSIST_WGS_MAX + 8) doubles = 2064 words >= BAR_WORDS
        W.bar_flat = getenv("SNK_BA_FLAT_BARRIER") != nullptr ? 1 : 0;
        W.timing   = getenv("SNK_BA_PCG_TIMING") != nullptr ? 1 : 0;
        // one problem, a cooperative launch the device can hold: two workgroups per compute unit (SNK_BA_PCGL_LAUNCHES=1: the
        // multi-launch form, A/B and the fallback for batches of large problems)
        W.persist_wgs = 0;
        static const bool launches_env = getenv("SNK_BA_PCGL_LAUNCHES") != nullptr;
        if (count == 1 && !launches_env)
        {
            hipDeviceProp_t prop;
            int coop = 0, per_cu = 0;
            if (hipGetDeviceProperties(&prop, h->device) == hipSuccess &&
                hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, h->device) == hipSuccess && coop &&
                (per_cu = 1) >= 1)
            {
                // one workgroup per compute unit: the barrier's cost grows with the participants (measured 7.0 ms per FullBA(4) with 256, 10.4 with 512)
                // workgroups: a dozen rows of S each, at most one per compute unit -- the barrier's cost grows with the participants (FullBA(4) on 300
                // keyframes, n6 = 1794: 5.0 / 4.6 / 5.9 / 5.0 ms with 256 / 128 / 64 / 32 workgroups, 10.4 with 512 in the three-barrier form)
                const int wgs = std::min(PERSIST_WGS_MAX, snk_env_int("SNK_BA_PERSIST_WGS", std::min(prop.multiProcessorCount, std::max(16, ceil_div(max_n6, 12)))));
                const bool fits = (size_t)max_n6 * 8 <= 150 * 1024 && ceil_div(max_nfc, PERSIST_CAMS) <= PERSIST_WGS_MAX;  // p in LDS; partial-sum slots
                if (wgs >= 1 && fits && set_max_lds_once(reinterpret_cast<const void*>(pcgl_persist), 150 * 1024) == SNK_OK)
                {
                    // all workgroups must be resident at once (grid barriers inside): ask the runtime what this kernel's registers and THIS
                    // problem's LDS allow per compute unit instead of assuming one (round-5 advisor) -- other CU / LDS configurations,
                    // partitioned devices
                    int resident = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, reinterpret_cast<const void*>(pcgl_persist), PERSIST_THREADS, (size_t)max_n6 * 8) == hipSuccess &&
                        resident >= 1)
                        W.persist_wgs = std::min(wgs, resident * prop.multiProcessorCount);
                    // the one-barrier form needs 3 n6 doubles of LDS (SNK_BA_PERSIST_TWO_BARRIERS=1: A/B, the round-5 form)
                    W.persist_one = 0;
                    if (W.persist_wgs > 0 && (size_t)max_n6 * 24 <= 150 * 1024 && getenv("SNK_BA_PERSIST_TWO_BARRIERS") == nullptr &&
                        set_max_lds_once(reinterpret_cast<const void*>(pcgl_persist1), 150 * 1024) == SNK_OK &&
                        hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, reinterpret_cast<const void*>(pcgl_persist1), PERSIST_THREADS, (size_t)max_n6 * 24) == hipSuccess &&
                        resident >= 1 && resident * prop.multiProcessorCount >= W.persist_wgs)
                        W.persist_one = 1;
                    // ... and with S in registers when the system is small enough (8 rows x 2048 columns per workgroup, one workgroup per 8 rows)
                    // 16 rows per workgroup above 512 unknowns (the barrier and the exchange of A p grow with the workgroups: 225 against 113
                    // for 300 keyframes: 9.4 -> 7.1 us per PCG iteration; 90 against 45 for 120 keyframes: 6.45 -> 5.7), 8 below; SNK_BA_PERSIST_REG_ROWS / _THREADS: A/B
                    static const int rows_env = snk_env_int("SNK_BA_PERSIST_REG_ROWS", 0);
                    const int reg_rows     = rows_env == 8 || rows_env == 16 ? rows_env : (max_n6 > 512 ? 16 : 8);
                    const void* reg_kernel = persist_reg_kernel(reg_rows);
                    const int wgs_reg = ceil_div(std::max(max_n6, 1), reg_rows);
                    if (W.persist_one == 1 && max_n6 <= PREG_MAX_N6 && getenv("SNK_BA_PERSIST_STREAM") == nullptr && getenv("SNK_BA_PERSIST_WGS") == nullptr &&
                        set_max_lds_once(reg_kernel, 150 * 1024) == SNK_OK &&
                        hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, reg_kernel, PERSIST_THREADS, (size_t)max_n6 * 16) == hipSuccess &&
                        resident >= 1 && resident * prop.multiProcessorCount >= wgs_reg && wgs_reg <= PERSIST_WGS_MAX)
                    {
                        W.persist_rows = reg_rows;
                        W.persist_one = 2;
                        W.persist_wgs = wgs_reg;
                    }
                }
            }
            (void)hipGetLastError();
        }
    }
#undef RS
#undef UP
    // ... and the buffers that start as zeros are entries of the same table (source NULL): nine fill launches of ~5 us each
    // stood between the upload and the first kernel of the solve
    auto zero = [&](DevBuf& b, size_t bytes) -> int
    {
        SNK_REQUIRE(tab.n < COPY_TAB_MAX && bytes < (1ull << 32), "scene list too large for the upload table");
        tab.src[tab.n] = nullptr, tab.dst[tab.n] = b.p, tab.bytes[tab.n] = (unsigned)bytes;
        ++tab.n;
        return SNK_OK;
    };
    if ((rc = zero(h->d_outlier, (size_t)std::max(orig_off, 1))) != SNK_OK) return rc;
    {
        // the state starts as begin_solve would leave it (the first solve of the scene then needs no launch for that)
        auto& states = LS.states;
        State s0{};
        s0.lambda = make_opt(h->opt).lambda_init;
        s0.vfac   = 2.0;
        states.assign((size_t)count, s0);
        SNK_REQUIRE(tab.n < COPY_TAB_MAX, "scene list too large for the upload table");
        tab.src[tab.n] = states.data(), tab.dst[tab.n] = h->d_state.p, tab.bytes[tab.n] = (unsigned)(states.size() * sizeof(State));
        ++tab.n;
        h->state_fresh = true;
    }
    if ((rc = zero(h->d_r, nobs * 4 * 8)) != SNK_OK) return rc;
    if ((rc = zero(h->d_x, (size_t)std::max(vec_off, 1) * 8)) != SNK_OK) return rc;
    // points without observations are in no work item of schur_fused: their cost, V^-1 and b_p are zero once and for all
    if ((rc = zero(h->d_cost, npt * 8)) != SNK_OK) return rc;
    if ((rc = zero(h->d_cost_new, npt * 8)) != SNK_OK) return rc;
    if ((rc = zero(h->d_Vinv, npt * 6 * 8)) != SNK_OK) return rc;
    if ((rc = zero(h->d_bp, npt * 3 * 8)) != SNK_OK) return rc;
    {
        // enough workgroups per array to keep the bus busy: one per 64 KB of the largest list, 16 .. 256
        unsigned big = 0;
        for (int e = 0; e < tab.n; ++e) big = std::max(big, tab.bytes[e]);
        // (more workgroups per array do not shorten it: 16.3 / 18.3 / 17.1 / 15.2 us with one per 64 / 16 / 4 / 1 KB, r03ag)
        const int gx = (int)std::min(256u, std::max(16u, big >> 16));
        hipLaunchKernelGGL(copy_table_kernel, dim3(gx, tab.n), dim3(256), 0, st, tab);
        SNK_LAUNCH_CHECK();
        if (tab2.n > 0)
        {
            hipLaunchKernelGGL(copy_table_kernel, dim3(gx, tab2.n), dim3(256), 0, st, tab2);  // stream-ordered behind the upload
            SNK_LAUNCH_CHECK();
        }
    }
    if (!h->pcg_large)
    {
        // process-wide, once, to the most the kernels can use (enqueue_lm keeps S in LDS only when it fits 158 KB)
        if ((rc = set_max_lds_once(reinterpret_cast<const void*>(pcg_solve<true>), 158 * 1024)) != SNK_OK) return rc;
        if ((rc = set_max_lds_once(reinterpret_cast<const void*>(pcg_solve<false>), 158 * 1024)) != SNK_OK) return rc;
    }

    Arrays& A   = h->arr;
    A.prob      = h->d_prob.as<Prob>();
    A.state     = h->d_state.as<State>();
    A.pose      = h->d_pose.as<double>();
    A.pose_new  = h->d_pose_new.as<double>();
    A.pt        = h->d_pt.as<double>();
    A.pt_new    = h->d_pt_new.as<double>();
    A.pt_const  = h->d_ptc.as<unsigned char>();
    A.cam_idx   = h->d_camidx.as<int>();
    A.pt_start  = h->d_ptstart.as<int>();
    A.o_img     = h->d_oimg.as<int>();
    A.o_cam     = h->d_ocam.as<int>();
    A.o_ptfree  = h->d_optfree.as<unsigned char>();
    A.o_uv      = h->d_ouv.as<double2>();
    A.o_depth   = h->d_odepth.as<double>();
    A.o_weight  = h->d_oweight.as<double>();
    A.o_orig    = h->d_oorig.as<int>();
    A.o_pt      = h->d_optidx.as<int>();
    A.wv_pt     = h->d_wvpt.as<int>();
    A.rpc_meta  = h->d_rpcmeta.as<RpcMeta>();
    A.rpc_out   = h->d_rpcout.as<double>();
    A.cam_rpc_start = h->d_camrpcstart.as<int>();
    A.cam_rpc_items = h->d_camrpcitems.as<int>();
    A.blk_rpc   = h->d_blkrpc.as<int>();
    A.rpc_next  = h->d_rpcnext.as<int>();
    A.outlier   = h->d_outlier.as<unsigned char>();
    A.o_r       = h->d_r.as<double>();
    A.o_W       = h->d_W.as<double>();
    A.ptv       = h->d_ptv.as<double>();
    A.cs_obs    = h->d_csobs.as<CamObs>();
    A.set_items = h->d_setitems.as<SetItem>();
    A.set_obs   = h->d_setobs.as<SetObs>();
    A.set_pts   = h->d_setpts.as<int2>();
    A.set_pairs = h->d_setpairs.as<int>();
    A.cc_start   = h->d_ccstart.as<int>();
    A.cc_items   = h->d_ccitems.as<int>();
    A.cam_part   = h->d_campart.as<double>();
    A.cblk_start = h->d_cblkstart.as<int>();
    A.cblk_items = h->d_cblkitems.as<int>();
    A.s_part    = h->d_spart.as<double>();
    A.Vinv      = h->d_Vinv.as<double>();
    A.bp        = h->d_bp.as<double>();
    A.cost_pt   = h->d_cost.as<double>();
    A.cost_pt_new = h->d_cost_new.as<double>();
    A.U         = h->d_U.as<double>();
    A.cam_start = h->d_camstart.as<int>();
    A.cam_items = h->d_camitems.as<int>();
    A.blk_start = h->d_blkstart.as<int>();
    A.blk_ent   = h->d_blkent.as<int4>();
    A.S         = h->d_S.as<double>();
    A.rhs       = h->d_rhs.as<double>();
    A.x         = h->d_x.as<double>();
    A.chi2      = h->d_chi2.as<double>();
    // the lists the device builds from the uploaded ones (stream ordered behind copy_table_kernel)
    if (early_upload)
    {
        if (obs_off > 0)
        {
            int max_no = 0;
            for (int b = 0; b < count; ++b) max_no = std::max(max_no, probs[(size_t)b].no);
            hipLaunchKernelGGL(derive_obs_fields, dim3(ceil_div(std::max(max_no, 1), 256), count), dim3(256), 0, st, A, h->d_optidx.as<int>(), h->d_ocam.as<int>(),
                               h->d_optfree.as<unsigned char>());
            SNK_LAUNCH_CHECK();
        }
        if (max_nfc > 0 && max_citems > 0)
        {
            hipLaunchKernelGGL(derive_cam_items, dim3(max_nfc, count), dim3(64), 0, st, A, (const int*)h->d_ocam.as<int>(), h->d_camitems.as<int>());
            SNK_LAUNCH_CHECK();
        }
    }
    if (max_citems > 0)
    {
        hipLaunchKernelGGL(gather_cam_records, dim3(ceil_div(max_citems, 256), count), dim3(256), 0, st, A, h->d_csobs.as<CamObs>());
        SNK_LAUNCH_CHECK();
    }
    if (max_set_items > 0)
    {
        hipLaunchKernelGGL(gather_set_records, dim3(max_set_items, count), dim3(256), 0, st, A, h->d_setobs.as<SetObs>());
        SNK_LAUNCH_CHECK();
    }
    // batches that will run the point-major kernels never read the block entries (SNK_BA_CHECK_LISTS=1 builds and checks them anyway)
    static const bool check_lists_be = getenv("SNK_BA_CHECK_LISTS") != nullptr;
    const bool skip_entries = dev_entries && count >= 16 && ba_sets_will_run(h) && !check_lists_be;
    h->blk_built = !skip_entries;
    if (dev_entries && !skip_entries)
    {
        if (max_be_waves > 0)
        {
            hipLaunchKernelGGL(block_entries_count, dim3(max_be_waves, count), dim3(64), 0, st, A, h->d_becnt.as<int>());
            SNK_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(block_entries_scan, dim3(count), dim3(BE_SCAN_THREADS), 0, st, A, h->d_becnt.as<int>(), h->d_blkstart.as<int>());
        SNK_LAUNCH_CHECK();
        if (max_be_waves > 0)
        {
            const size_t be_lds = (size_t)((max_nfc + 63) / 64) * 64 * 12;  // ballots (8 B) + bases (4 B) per camera
            hipLaunchKernelGGL(block_entries_fill, dim3(max_be_waves, count), dim3(64), be_lds, st, A, h->d_becnt.as<int>(), h->d_blkent.as<int4>());
            SNK_LAUNCH_CHECK();
        }
    }
    const auto t_rs = std::chrono::steady_clock::now();
    static const bool prof = getenv("SNK_BA_PROFILE_CREATE") != nullptr;  // host-side cost of a scene hand-over, in microseconds
    static const bool check_lists = getenv("SNK_BA_CHECK_LISTS") != nullptr;  // tests / fuzzers: device-built lists against the host builder
    if (check_lists)
    {
        // the host builder's lists (never uploaded here) against what the kernels wrote
        SNK_HIP_CHECK(hipStreamSynchronize(st));
        std::vector<CamObs> d_rec(camitems.size());
        if (!d_rec.empty()) SNK_HIP_CHECK(hipMemcpy(d_rec.data(), h->d_csobs.p, d_rec.size() * sizeof(CamObs), hipMemcpyDeviceToHost));
        for (int b = 0; b < count; ++b)
        {
            const snk_ba_problem& P = problems[b];
            const Prob& pr          = probs[(size_t)b];
            const int n_items       = camstart[(size_t)pr.camstart_off + (size_t)pr.nfc];
            for (int k = 0; k < n_items; ++k)
            {
                const int s = camitems[(size_t)pr.citem_off + (size_t)k], o = oorig[(size_t)pr.obs_off + (size_t)s] - pr.orig_off;
                const CamObs& r = d_rec[(size_t)pr.citem_off + (size_t)k];
                const bool same = r.u == P.obs_uv[o][0] && r.v == P.obs_uv[o][1] && r.depth == P.obs_depth[o] && r.weight == P.obs_weight[o] &&
                                  (r.ptw & 0x7FFFFFFF) == P.obs_pt[o] && r.orig == pr.orig_off + o &&
                                  (r.ptw < 0 ? 1 : 0) == (P.pt_const[P.obs_pt[o]] ? 0 : 1);
                SNK_REQUIRE(same, "SNK_BA_CHECK_LISTS: a device-gathered camera record differs from the caller's observation");
            }
        }
        {
            // the work items' observation records (gather_set_records) against the caller's arrays
            std::vector<SetObs> d_sr((size_t)n_setrec);
            if (!d_sr.empty()) SNK_HIP_CHECK(hipMemcpy(d_sr.data(), h->d_setobs.p, d_sr.size() * sizeof(SetObs), hipMemcpyDeviceToHost));
            for (int b = 0; b < count; ++b)
            {
                const snk_ba_problem& P = problems[b];
                const Prob& pr          = probs[(size_t)b];
                for (int it = 0; it < pr.n_set; ++it)
                {
                    const SetItem& si = LS.setitems[(size_t)pr.set_off + (size_t)it];
                    for (int q = 0; q < si.n_pts; ++q)
                        for (int a = 0; a < si.run; ++a)
                        {
                            const int2 pp = LS.setpts[(size_t)si.pts_off + (size_t)q];
                            const int s = pp.y + a, o = oorig[(size_t)pr.obs_off + (size_t)s] - pr.orig_off;
                            const SetObs& r = d_sr[(size_t)si.rec_off + (size_t)q * si.run + (size_t)a];
                            const bool same = r.u == P.obs_uv[o][0] && r.v == P.obs_uv[o][1] && r.depth == P.obs_depth[o] &&
                                              r.weight == P.obs_weight[o] && r.orig == pr.orig_off + o &&
                                              r.pk == set_pack(P.obs_img[o], ocam[(size_t)pr.obs_off + (size_t)s], P.pt_const[pp.x] ? 0 : 1) &&
                                              P.obs_pt[o] == pp.x;
                            SNK_REQUIRE(same, "SNK_BA_CHECK_LISTS: a device-gathered work-item record differs from the caller's observation");
                        }
                }
            }
        }
        if (dev_entries)
        {
            std::vector<int> d_bs((size_t)blkstart_total);
            std::vector<int4> d_ent((size_t)ent_total);
            SNK_HIP_CHECK(hipMemcpy(d_bs.data(), h->d_blkstart.p, d_bs.size() * sizeof(int), hipMemcpyDeviceToHost));
            if (!d_ent.empty()) SNK_HIP_CHECK(hipMemcpy(d_ent.data(), h->d_blkent.p, d_ent.size() * sizeof(int4), hipMemcpyDeviceToHost));
            pvec<int> h_bs;
            pvec<int4> h_ent;
            for (int b = 0; b < count; ++b)
            {
                const Prob& pr  = probs[(size_t)b];
                const size_t nb = (size_t)pr.nfc * pr.nfc, bs_at = h_bs.size(), ent_at = h_ent.size();
                host_block_entries(b, h_bs, h_ent);
                for (size_t k = 0; k <= nb; ++k)
                    SNK_REQUIRE(d_bs[(size_t)pr.blkstart_off + k] == h_bs[bs_at + k], "SNK_BA_CHECK_LISTS: device-built block starts differ from the host builder's");
                SNK_REQUIRE((long long)h_bs[bs_at + nb] <= ent_bound[(size_t)b], "SNK_BA_CHECK_LISTS: block entries exceed their bound");
                for (int k = 0; k < h_bs[bs_at + nb]; ++k)
                {
                    const int4 d = d_ent[(size_t)pr.ent_off + (size_t)k], w = h_ent[ent_at + (size_t)k];
                    SNK_REQUIRE(d.x == w.x && d.y == w.y && d.z == w.z, "SNK_BA_CHECK_LISTS: device-built block entries differ from the host builder's");
                }
            }
        }
    }
    if (prof)
    {
        SNK_HIP_CHECK(hipStreamSynchronize(st));
        const auto t_end = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return (long long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        fprintf(stderr, "[snk_ba_set_problems] lists %lld us, uploads %lld us, reserve+memset %lld us, sync %lld us\n", us(t_begin, t_lists),
                us(t_lists, t_up), us(t_up, t_rs), us(t_rs, t_end));
        fprintf(stderr, "[snk_ba_set_problems] lists in us: values+sort %lld, observation arrays %lld, wave items %lld, camera lists %lld, "
                        "block entries %lld, point sets %lld (grouping %lld, work items %lld, block lists %lld), constraints %lld, rest %lld\n",
                sec_us[0] / 1000, sec_us[1] / 1000, sec_us[2] / 1000, sec_us[3] / 1000, sec_us[4] / 1000,
                (sec_us[5] + sec_us[8] + sec_us[9]) / 1000, sec_us[8] / 1000, sec_us[9] / 1000, sec_us[5] / 1000, sec_us[6] / 1000, sec_us[7] / 1000);
    }
    return SNK_OK;
}

int snk_ba_set_problem(snk_ba* h, const snk_ba_problem* problem)
{
    return snk_ba_set_problems(h, problem, 1);
}

int snk_ba_set_outliers(snk_ba* h, int problem, const uint8_t* obs_outlier)
{
    SNK_REQUIRE(h != nullptr && h->count > 0, "no problem set");
    SNK_REQUIRE(problem >= 0 && problem < h->count, "problem index out of range");
    SNK_HIP_CHECK(hipSetDevice(h->device));
    unsigned char* dst = h->d_outlier.as<unsigned char>() + h->orig_off[(size_t)problem];
    const size_t n     = (size_t)h->orig_n[(size_t)problem];
    if (n == 0) return SNK_OK;
    if (obs_outlier)
    {
        // through the handle's pinned buffer: a copy from the caller's pageable array stages and synchronises inside the runtime
        // (0.08 ms for 16 KB); the stream is idle between the calls of a local-BA sequence, the wait below is for safety
        SNK_HIP_CHECK(hipStreamSynchronize(h->stream));
        int rc = h->h_stage.reserve(n);
        if (rc != SNK_OK) return rc;
        memcpy(h->h_stage.p, obs_outlier, n);
        SNK_HIP_CHECK(hipMemcpyAsync(dst, h->h_stage.p, n, hipMemcpyHostToDevice, h->stream));
    }
    else
        SNK_HIP_CHECK(hipMemsetAsync(dst, 0, n, h->stream));
    return SNK_OK;  // stream ordered: the next solve / residuals call of this handle sees the mask
}

int snk_ba_reset(snk_ba* h)
{
    SNK_REQUIRE(h != nullptr && h->count > 0, "no problem set");
    SNK_HIP_CHECK(hipSetDevice(h->device));
    SNK_HIP_CHECK(hipMemcpyAsync(h->d_pose.p, h->d_pose0.p, (size_t)h->tot_img * 7 * 8, hipMemcpyDeviceToDevice, h->stream));
    SNK_HIP_CHECK(hipMemcpyAsync(h->d_pt.p, h->d_pt0.p, (size_t)h->tot_pt * 3 * 8, hipMemcpyDeviceToDevice, h->stream));
    return SNK_OK;
}

constexpr int BA_GRAPH_CHAINS = 1;  // chains a recorded batch sequence is split into by default (measured: see enqueue_lm)
static int enqueue_lm(snk_ba* h, int iterations, Launcher& L, bool only_marked = false)
{
    const Opt O = make_opt(h->opt);
    const int B_all = h->count;
    Arrays A_all    = h->arr;
    const int cond = only_marked ? 1 : 0;
    if (only_marked)
    {
        int rc = h->d_probcond.reserve((size_t)B_all * sizeof(Prob));
        if (rc != SNK_OK) return rc;
        LAUNCH(select_marked, dim3(ceil_div(B_all, 64)), dim3(64), 0, h->d_prob.as<Prob>(), h->d_probcond.as<Prob>(), h->d_state.as<State>(), B_all,
               O.lambda_init);
        A_all.prob = h->d_probcond.as<Prob>();
    }
    else if (h->state_fresh && L.graph == nullptr)
        ;  // the first solve after a hand-over: the uploaded state IS what begin_solve writes (one launch less on the keyframe path)
    else
        LAUNCH(begin_solve, dim3(ceil_div(B_all, 64)), dim3(64), 0, h->d_state.as<State>(), B_all, O.lambda_init, 0);
    if (L.graph == nullptr) h->state_fresh = false;
    // A recorded sequence of a big batch is recorded as SEVERAL chains over disjoint ranges of the windows (windows are independent; a
    // kernel finds its window through A.prob / A.state only, so a range is those two pointers moved): the graph's branches run side by
    // side, and the latency-bound kernels of one range (pcg_small: one workgroup per window walking its PCG iterations; schur_sum; accept_pass)
    // fill the issue slots the bandwidth-bound ones of another leave.  Same kernels, same arithmetic per window.  SNK_BA_GRAPH_CHAINS=n.
    static const int chains_env = getenv("SNK_BA_GRAPH_CHAINS") ? atoi(getenv("SNK_BA_GRAPH_CHAINS")) : 0;
    int n_chain = 1;
    if (L.graph != nullptr && !h->pcg_large && B_all >= 128) n_chain = chains_env > 0 ? std::min(chains_env, B_all / 64) : BA_GRAPH_CHAINS;
    hipGraphNode_t const root = L.last;
    for (int chain = 0; chain < n_chain; ++chain)
    {
    const int b_lo = chain == 0 ? 0 : ((int)((long long)B_all * chain / n_chain) & ~7);
    const int b_hi = chain == n_chain - 1 ? B_all : ((int)((long long)B_all * (chain + 1) / n_chain) & ~7);
    const int B    = b_hi - b_lo;
    Arrays A       = A_all;
    A.prob += b_lo;
    A.state += b_lo;
    L.last = root;
    const dim3 gpt(std::max(1, ceil_div(h->max_np, 128)), B);
    size_t pcg_lds        = (size_t)h->max_n6 * 9 * 8 + (size_t)h->max_nfc * 36 * 8;
    const size_t s_bytes  = (size_t)h->max_n6 * h->max_n6 * 8;
    // + 128 doubles read padding behind S (rows lane and lane + 64 are read unmasked) + [2][4][128] partial products
    const int s_in_lds    = pcg_lds + s_bytes + 1024 + 8192 <= 158 * 1024 ? 1 : 0;
    if (s_in_lds) pcg_lds += s_bytes + 1024 + 8192;
    for (int it = 0; it < iterations; ++it)
    {
        static const bool no_wave = getenv("SNK_BA_NO_POINT_WAVE") != nullptr;
        // (SNK_BA_NO_SCHUR_SET, SNK_BA_SCHUR_SET_MIN_ITEMS: ba_sets_will_run -- a single small window has too few work items to fill the
        // chip, the block-major pass is quicker there)
        static const bool no_mfma  = getenv("SNK_BA_NO_SCHUR_MFMA") != nullptr;   // A/B: the vector-ALU form (schur_set)
        static const bool no_fused = getenv("SNK_BA_NO_SCHUR_FUSED") != nullptr;  // A/B: point_wave + schur_mfma through W in HBM
        // (the whole batch decides, not the range of a chain: the hand-over left out the block entries on the same answer)
        const bool use_set = ba_sets_will_run(h);
        if (!use_set && !h->blk_built)
        {
            set_error("bundle adjustment: the block-major pass was chosen but the hand-over did not build its lists (internal)");
            return SNK_ERR_HIP;
        }
        // points with 9 or 10 free observations need a fourth tile row: 260 registers, one wavefront per SIMD -- there the
        // linearisation stays in point_wave and schur_mfma<4> reads W (measured on the 300-keyframe global BA: 14.7 vs 15.5 ms);
        // SNK_BA_FUSED_K10=1 forces the fused form (tests)
        static const bool fused_k10 = getenv("SNK_BA_FUSED_K10") != nullptr;
        const bool fused   = use_set && !no_mfma && !no_fused && (h->set_k_max <= 8 || fused_k10);
        const int nsx      = ceil_div(std::max(h->max_set_items, 1), 4);
        // SNK_BA_CAM_SUMS=1: the camera pass as per-item partial sums out of schur_fused<3, true> + cam_sum instead of cam_pass.
        // Built because cam_pass linearises every observation a second time (294 us per 1024 windows); measured: schur_fused
        // 1127 -> 1437 us for the five passes through the contribution buffer, cam_sum 29 us -- 46 us SLOWER per LM iteration
        // (profiles/r03/r03aa_*).  Kept selectable and tested, not the default.
        static const bool want_cam_sums = getenv("SNK_BA_CAM_SUMS") != nullptr;
        const bool cam_sums = fused && h->set_k_max <= 8 && h->cam_sums_ok && want_cam_sums;
        if (fused)
        {
            // linearisation + Schur products in one kernel (W stays in LDS); it writes what point_wave writes per point
            if (h->set_k_max <= 8 && cam_sums)
                LAUNCH((schur_fused<3, true>), dim3(nsx * 8 * ceil_div(B, 8)), dim3(256), 0, A, O, nsx, B);
            else if (h->set_k_max <= 8)
                LAUNCH((schur_fused<3, false>), dim3(nsx * 8 * ceil_div(B, 8)), dim3(256), 0, A, O, nsx, B);
            else
                LAUNCH((schur_fused<4, false>), dim3(nsx * 8 * ceil_div(B, 8)), dim3(256), 0, A, O, nsx, B);
        }
        else if (h->point_wave_ok && !no_wave)
            LAUNCH(point_wave, dim3(h->max_wv, B), dim3(64), 0, A, O);
        else
            LAUNCH(point_pass<0>, gpt, dim3(128), 0, A, O);
        if (h->max_nfc > 0)
        {
            if (h->max_rpc > 0) LAUNCH(rpc_pass, dim3(ceil_div(h->max_rpc, 64), B), dim3(64), 0, A, 0);
            if (cam_sums)
                LAUNCH(cam_sum, dim3(h->max_nfc, B), dim3(64), 0, A);
            else if (B >= 16)
            {
                static const bool cam_2d = getenv("SNK_BA_CAM_GRID_2D") != nullptr;  // A/B: the round-1..4 grid (cameras of a window over all XCDs)
                if (cam_2d) LAUNCH(cam_pass<64>, dim3(h->max_nfc, B), dim3(64), 0, A, O, 0, B);
                else LAUNCH(cam_pass<64>, dim3(8 * h->max_nfc * ceil_div(B, 8)), dim3(64), 0, A, O, h->max_nfc, B);
            }
            else
                LAUNCH(cam_pass<256>, dim3(h->max_nfc, B), dim3(256), 0, A, O, 0, B);
            {
                const int nbx = ceil_div(h->max_nfc * h->max_nfc, 4);  // schur_pass: a block per wavefront
                const int nbs = ceil_div(h->max_nfc * h->max_nfc, 4 * SUM_NB);  // schur_sum: SUM_NB blocks per wavefront
                if (use_set)
                {
                    const bool q2 = h->set_run_max * 9 + 3 <= 2 * 64;
                    if (fused)
                        ;  // the partial sums are there already
                    else if (!no_mfma && h->set_k_max <= 8)
                    {
                        if (q2) LAUNCH((schur_mfma<3, 2>), dim3(nsx * 8 * ceil_div(B, 8)), dim3(256), 0, A, nsx, B);
                        else LAUNCH((schur_mfma<3, 3>), dim3(nsx * 8 * ceil_div(B, 8)), dim3(256), 0, A, nsx, B);
                    }
                    else if (!no_mfma)
                        LAUNCH((schur_mfma<4, 3>), dim3(nsx * 8 * ceil_div(B, 8)), dim3(256), 0, A, nsx, B);
                    else if (h->set_small)
                        LAUNCH((schur_set<4, 2>), dim3(nsx * 8 * ceil_div(B, 8)), dim3(256), 0, A, nsx, B);
                    else
                        LAUNCH((schur_set<6, 3>), dim3(nsx * 8 * ceil_div(B, 8)), dim3(256), 0, A, nsx, B);
                    LAUNCH(schur_sum, dim3(nbs * 8 * ceil_div(B, 8)), dim3(256), 0, A, nbs, B);
                }
                else
                {
                    static const bool wide_off = getenv("SNK_BA_NO_SCHUR_WIDE") != nullptr;  // A/B
                    const int nb = h->max_nfc * h->max_nfc;
                    if (B < 16 && (long long)nb * B <= 8192 && !wide_off)
                        LAUNCH(schur_pass<4>, dim3(nb * B), dim3(256), 0, A, h->point_wave_ok && !no_wave ? 1 : 0, nb, B);
                    else
                        LAUNCH(schur_pass<1>, dim3(nbx * 8 * ceil_div(B, 8)), dim3(256), 0, A,
                                           h->point_wave_ok && !no_wave ? 1 : 0, nbx, B);
                }
            }
            static const bool pcg_in_lds = getenv("SNK_BA_PCG_LDS") != nullptr;  // A/B: S in LDS (pcg_solve<true>) also for local-BA sizes
            if (!h->pcg_large && h->max_n6 <= 128 && !O.pcg_general && !pcg_in_lds)
                LAUNCH(pcg_small, dim3(B), dim3(PCG_THREADS), ((size_t)h->max_n6 * 9 + (size_t)h->max_nfc * 72) * 8 + 8192, A, O);
            else if (!h->pcg_large)
                if (s_in_lds)
                    LAUNCH(pcg_solve<true>, dim3(B), dim3(PCG_THREADS), pcg_lds, A, O);
                else
                    LAUNCH(pcg_solve<false>, dim3(B), dim3(PCG_THREADS), pcg_lds, A, O);
            else
            {
                const PcgLarge& W = h->pcgw;
                const dim3 gcam(W.G, B);
                const dim3 gmv(ceil_div(h->max_n6, 256) * W.parts, B);
                LAUNCH(pcgl_init, gcam, dim3(64), 0, A, O, W);
                bool persisted = false;
                if (W.persist_wgs > 0 && B == 1 && L.graph == nullptr && L.err == hipSuccess)
                {
                    static const bool fail_hook = getenv("SNK_BA_PERSIST_FAIL") != nullptr;  // tests: the runtime refuses the cooperative launch
                    L.cooperative = true;
                    if (fail_hook)
                        L.cooperative = false, L.err = hipErrorCooperativeLaunchTooLarge;
                    else
                    {
                        if (W.persist_one == 2 && W.persist_rows == 16)
                            LAUNCH(pcgl_persist_reg<16>, dim3(W.persist_wgs), dim3(PERSIST_THREADS), (size_t)h->max_n6 * 16, A, O, W);
                        else if (W.persist_one == 2)
                            LAUNCH(pcgl_persist_reg<8>, dim3(W.persist_wgs), dim3(PERSIST_THREADS), (size_t)h->max_n6 * 16, A, O, W);
                        else if (W.persist_one)
                            LAUNCH(pcgl_persist1, dim3(W.persist_wgs), dim3(PERSIST_THREADS), (size_t)h->max_n6 * 24, A, O, W);
                        else
                            LAUNCH(pcgl_persist, dim3(W.persist_wgs), dim3(PERSIST_THREADS), (size_t)h->max_n6 * 8, A, O, W);
                    }
                    persisted = L.err == hipSuccess;
                    if (!persisted)
                    {
                        // the runtime refused the cooperative launch (co-residency under another CU / LDS configuration, a partitioned device, a
                        // driver without cooperative queues): nothing ran, pcgl_init's state is what the launch sequence below starts from too --
                        // this handle uses that sequence from now on
                        if (getenv("SNK_DEBUG")) fprintf(stderr, "snake_hip: cooperative PCG launch refused (%s); multi-launch PCG\n", hipGetErrorString(L.err));
                        (void)hipGetLastError();
                        L.err               = hipSuccess;
                        h->pcgw.persist_wgs = 0;
                    }
                }
                if (!persisted)
                for (int k = 0; k < O.max_pcg; ++k)
                {
                    LAUNCH(pcgl_matvec, gmv, dim3(256), 0, A, O, W, k);
                    LAUNCH(pcgl_combine, gcam, dim3(64), 0, A, O, W, k);
                    LAUNCH(pcgl_update, gcam, dim3(64), 0, A, O, W, k);
                    LAUNCH(pcgl_direction, gcam, dim3(64), 0, A, O, W, k);
                    LAUNCH(pcgl_latch, dim3(ceil_div(B, 64)), dim3(64), 0, A, O, W, k);
                }
            }
        }
        static const bool no_uc = getenv("SNK_BA_NO_UPDATE_COST") != nullptr;  // A/B: update_wave + cost_wave
        if (use_set && !no_uc)
        {
            // the work items of the set cover every point that has observations (schur_fused's linearisation-only items included)
            LAUNCH(update_pass, dim3(std::max(1, ceil_div(h->max_ni, 128)), B), dim3(128), 0, A, 1);
            LAUNCH(update_cost, dim3(nsx * 8 * ceil_div(B, 8)), dim3(256), 0, A, O, nsx, B);
        }
        else if (h->point_wave_ok && !no_wave)
        {
            const dim3 gwv(ceil_div(h->max_wv, 4), B);
            LAUNCH(update_wave, dim3(gwv.x + std::max(1, ceil_div(h->max_ni, 256)), B), dim3(256), 0, A, O, (int)gwv.x);
            LAUNCH(cost_wave, gwv, dim3(256), 0, A, O);
        }
        else
        {
            LAUNCH(update_pass, dim3(std::max(1, ceil_div(h->max_np + h->max_ni, 128)), B), dim3(128), 0, A, 0);
            LAUNCH(point_pass<1>, gpt, dim3(128), 0, A, O);
        }
        if (h->max_rpc > 0 && h->max_nfc > 0) LAUNCH(rpc_pass, dim3(ceil_div(h->max_rpc, 64), B), dim3(64), 0, A, 1);
        // big problems in small numbers: the copy of the accepted state on its own workgroups (one workgroup per problem otherwise)
        const bool split_copy = B <= 4 && h->max_np >= 4096;
        LAUNCH(accept_pass, dim3(B), dim3(ACC_THREADS), split_copy ? 2 * ACC_TILE * sizeof(double) : 0, A, cond, split_copy ? 1 : 0, split_copy ? 1 : 0);
        if (split_copy) LAUNCH(accept_copy, dim3(std::min(128, ceil_div(h->max_np * 3 + h->max_ni * 7, 512)), B), dim3(256), 0, A, cond);
    }
    }  // chain
    if (L.err != hipSuccess)
    {
        set_error("bundle adjustment launch failed: %s (%s:%d)", hipGetErrorString(L.err), __FILE__, L.line);
        return SNK_ERR_HIP;
    }
    return SNK_OK;
}

// The LM loop is a fixed launch sequence (all decisions are taken on the device).  A sequence that is issued again for
// the same problem set and iteration count is replayed as ONE graph launch instead of 7+ launches per iteration; the
// graph is built explicitly, node by node (see Launcher), never by stream capture.  The reference's local-BA call
// pattern is a NEW scene per keyframe solved once or twice (LocalBundleAdjustment.cpp:353-413), where building and
// instantiating a graph costs more than it saves: the first solve with a given iteration count after set_problems
// uses plain launches, a repeat builds the graph (measured: DESIGN.md section 4).  SNK_BA_NO_GRAPH=1: never,
// SNK_BA_GRAPH_FIRST=1: already on the first use.
int snk_ba_solve_async(snk_ba* h, int iterations)
{
    SNK_REQUIRE(h != nullptr && h->count > 0, "no problem set");
    SNK_REQUIRE(iterations >= 0, "negative iteration count");
    SNK_HIP_CHECK(hipSetDevice(h->device));
    static const bool no_graph    = getenv("SNK_BA_NO_GRAPH") != nullptr;
    static const bool graph_first = getenv("SNK_BA_GRAPH_FIRST") != nullptr;
    Launcher direct;
    direct.st = h->stream;
    // a cooperative launch is not a graph node: scenes solved by the one-launch PCG (global BA) always take plain launches -- seven per
    // LM iteration there, against milliseconds of work
    if (no_graph || iterations == 0 || (h->pcg_large && h->pcgw.persist_wgs > 0 && h->count == 1)) return enqueue_lm(h, iterations, direct);
    auto it = h->graphs.find(iterations);
    if (it == h->graphs.end())
    {
        if (!graph_first && h->plain_runs[iterations]++ == 0) return enqueue_lm(h, iterations, direct);
        Launcher rec;
        rec.st = h->stream;
        if (hipGraphCreate(&rec.graph, 0) != hipSuccess)
        {
            (void)hipGetLastError();
            return enqueue_lm(h, iterations, direct);
        }
        const int rc       = enqueue_lm(h, iterations, rec);
        hipGraphExec_t exec = nullptr;
        hipError_t ei       = hipErrorUnknown;
        if (rc == SNK_OK) ei = hipGraphInstantiate(&exec, rec.graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(rec.graph);
        if (rc != SNK_OK || ei != hipSuccess || exec == nullptr)
        {
            // nothing has been enqueued yet (recording only adds nodes): issue the sequence directly instead
            (void)hipGetLastError();
            return enqueue_lm(h, iterations, direct);
        }
        it = h->graphs.emplace(iterations, exec).first;
    }
    SNK_HIP_CHECK(hipGraphLaunch(it->second, h->stream));
    return SNK_OK;
}

int snk_ba_solve(snk_ba* h, int iterations, double* cost_initial, double* cost_final)
{
    const bool pcg_timing = h->pcg_large && h->pcgw.timing && h->pcgw.persist_one == 2;  // SNK_BA_PCG_TIMING=1 (diagnostic)
    if (pcg_timing) SNK_HIP_CHECK(hipMemsetAsync(h->pcgw.ps, 0, 6 * sizeof(double), h->stream));
    int rc = snk_ba_solve_async(h, iterations);
    if (rc != SNK_OK) return rc;
    std::vector<State> st((size_t)h->count);
    SNK_HIP_CHECK(hipMemcpyAsync(st.data(), h->d_state.p, st.size() * sizeof(State), hipMemcpyDeviceToHost, h->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (pcg_timing)
    {
        double t[6];
        SNK_HIP_CHECK(hipMemcpy(t, h->pcgw.ps, sizeof(t), hipMemcpyDeviceToHost));
        const double it = std::max(t[5], 1.0);  // 100 MHz ticks of workgroup 0, summed over the launches of this solve
        fprintf(stderr, "[pcgl_persist_reg] %d PCG iterations; set-up %.1f us per launch-sum; per iteration: product %.2f, barrier %.2f, A p exchange %.2f, "
                        "preconditioner + sums %.2f us\n", (int)t[5], t[0] * 0.01, t[1] * 0.01 / it, t[2] * 0.01 / it, t[3] * 0.01 / it, t[4] * 0.01 / it);
    }
    for (int b = 0; b < h->count; ++b)
    {
        if (cost_initial) cost_initial[b] = st[(size_t)b].cost_initial;
        if (cost_final) cost_final[b] = st[(size_t)b].cost;
    }
    return SNK_OK;
}

int snk_ba_solve_local_scene(snk_ba* h, int problem, double chi2_mono, double chi2_stereo, int extra_iterations, uint8_t* obs_outlier,
                             int* n_marked, double* cost_initial, double* cost_final, double (*pose)[7], double (*pt)[3])
{
    SNK_REQUIRE(h != nullptr && h->count > 0, "no problem set");
    SNK_REQUIRE(problem >= 0 && problem < h->count && n_marked != nullptr, "bad arguments");
    SNK_REQUIRE(chi2_mono > 0.0 && chi2_stereo > 0.0 && extra_iterations >= 0, "thresholds must be positive, extra_iterations >= 0");
    SNK_HIP_CHECK(hipSetDevice(h->device));
    int rc;
    {
        // The pinned staging buffer is sized for everything this call reads back BEFORE anything is enqueued: growing it frees the
        // old allocation, which an earlier snk_ba_set_outliers may still be copying from (its H2D copy is asynchronous) -- so a
        // growth waits for the stream first, and no reserve() below can reallocate behind an enqueued kernel or cost a second wait.
        const Prob& pr0   = h->probs[(size_t)problem];
        const size_t need = (size_t)pr0.ni * 56 + (size_t)pr0.np * 24 + (size_t)h->orig_n[(size_t)problem] +
                            (size_t)h->count * sizeof(State) + 4 * (256 + 16) + 128;
        if (need > h->h_stage.bytes)
        {
            SNK_HIP_CHECK(hipStreamSynchronize(h->stream));
            if ((rc = h->h_stage.reserve(need)) != SNK_OK) return rc;
        }
    }
    rc = snk_ba_solve_async(h, h->opt.max_iterations);  // initAndSolve
    if (rc != SNK_OK) return rc;
    Opt O         = make_opt(h->opt);
    O.chi2_mono   = chi2_mono;
    O.chi2_stereo = chi2_stereo;
    // chi-square pass at the solved state, marking on the device in the same kernel (the count and the costs of this moment stay in the state)
    hipLaunchKernelGGL(point_pass<3>, dim3(std::max(1, ceil_div(h->max_np, 128)), h->count), dim3(128), 0, h->stream, h->arr, O);
    SNK_LAUNCH_CHECK();
    // The extra iteration(s) (:399-410) run for the problems that had something marked.  Default: enqueued right behind the
    // pass, conditional on the device (select_marked) -- the whole call is ONE synchronisation.  SNK_BA_LOCAL_SYNC=1 (A/B), and
    // scenes on the multi-workgroup PCG (whose launch sequence sizes itself on the host): read the count back and decide here.
    static const bool sync_env = getenv("SNK_BA_LOCAL_SYNC") != nullptr;
    const bool on_device       = !sync_env && !h->pcg_large;
    State st{};
    bool have_state = false;
    if (extra_iterations > 0 && on_device)
    {
        Launcher direct;
        direct.st = h->stream;
        if ((rc = enqueue_lm(h, extra_iterations, direct, true)) != SNK_OK) return rc;
    }
    else if (extra_iterations > 0)
    {
        // Host-side decision (the multi-workgroup PCG, SNK_BA_LOCAL_SYNC): the SAME per-problem rule as the device path -- every
        // problem of the batch that had something marked gets the extra iteration(s), the others are left as they are.  The states of
        // all problems are read back; the iteration itself runs on select_marked's table like on the device path (a problem without
        // marks appears with every size zero, and the multi-workgroup PCG's kernels skip a problem with n6 == 0).
        const size_t sb = (size_t)h->count * sizeof(State);
        if ((rc = h->h_stage.reserve(64 + sb)) != SNK_OK) return rc;  // no-op: sized at the top
        SNK_HIP_CHECK(hipMemcpyAsync(h->h_stage.p, h->d_state.as<State>(), sb, hipMemcpyDeviceToHost, h->stream));
        SNK_HIP_CHECK(hipStreamSynchronize(h->stream));
        const State* all = h->h_stage.as<State>();
        bool any = false;
        for (int b = 0; b < h->count; ++b) any = any || all[b].marked > 0;
        memcpy(&st, all + problem, sizeof(State));
        have_state = true;
        if (any)
        {
            Launcher direct;
            direct.st = h->stream;
            if ((rc = enqueue_lm(h, extra_iterations, direct, true)) != SNK_OK) return rc;
        }
    }
    // One kernel writes the results into the pinned buffer (device visible) instead of one copy-engine transfer each; its
    // 16-byte loads want aligned sources: every piece is copied from the 16-byte boundary below it.
    const Prob& pr = h->probs[(size_t)problem];
    const size_t n = (size_t)h->orig_n[(size_t)problem];
    const char* src[4] = {reinterpret_cast<const char*>(h->d_pose.as<double>() + (size_t)pr.img_off * 7),
                          reinterpret_cast<const char*>(h->d_pt.as<double>() + (size_t)pr.pt_off * 3),
                          reinterpret_cast<const char*>(h->d_outlier.as<unsigned char>() + h->orig_off[(size_t)problem]),
                          reinterpret_cast<const char*>(h->d_state.as<State>() + problem)};
    void* dst[4]          = {pose, pt, obs_outlier, &st};
    const size_t bytes[4] = {pose ? (size_t)pr.ni * 56 : 0, pt ? (size_t)pr.np * 24 : 0, obs_outlier ? n : 0, have_state ? 0 : sizeof(State)};
    size_t at[4], shift[4], total = 0;
    for (int k = 0; k < 4; ++k)
    {
        shift[k] = (size_t)(reinterpret_cast<uintptr_t>(src[k]) & 15u);
        at[k]    = total;
        total += (bytes[k] + shift[k] + 255) & ~(size_t)255;
    }
    if ((rc = h->h_stage.reserve(total + 64)) != SNK_OK) return rc;
    char* hs = h->h_stage.as<char>();
    CopyTab tab;
    tab.n = 0;
    for (int k = 0; k < 4; ++k)
    {
        if (!bytes[k]) continue;
        tab.src[tab.n] = src[k] - shift[k], tab.dst[tab.n] = hs + at[k], tab.bytes[tab.n] = (unsigned)(bytes[k] + shift[k]);
        ++tab.n;
    }
    if (tab.n > 0)
    {
        hipLaunchKernelGGL(copy_table_kernel, dim3(4, tab.n), dim3(256), 0, h->stream, tab);
        SNK_LAUNCH_CHECK();
    }
    SNK_HIP_CHECK(hipStreamSynchronize(h->stream));
    for (int k = 0; k < 4; ++k)
        if (bytes[k]) memcpy(dst[k], hs + at[k] + shift[k], bytes[k]);
    *n_marked = st.marked;
    if (cost_initial) *cost_initial = st.first_cost_initial;  // the reference returns the FIRST solve's costs (LocalBundleAdjustment.cpp:412)
    if (cost_final) *cost_final = st.first_cost;
    return SNK_OK;
}

int snk_ba_get_state(snk_ba* h, int problem, double (*pose)[7], double (*pt)[3], int* pcg_iterations)
{
    SNK_REQUIRE(h != nullptr && h->count > 0, "no problem set");
    SNK_REQUIRE(problem >= 0 && problem < h->count, "problem index out of range");
    SNK_HIP_CHECK(hipSetDevice(h->device));
    const Prob& pr = h->probs[(size_t)problem];
    if (pose && pr.ni)
        SNK_HIP_CHECK(hipMemcpyAsync(pose, h->d_pose.as<double>() + (size_t)pr.img_off * 7, (size_t)pr.ni * 56,
                                     hipMemcpyDeviceToHost, h->stream));
    if (pt && pr.np)
        SNK_HIP_CHECK(hipMemcpyAsync(pt, h->d_pt.as<double>() + (size_t)pr.pt_off * 3, (size_t)pr.np * 24,
                                     hipMemcpyDeviceToHost, h->stream));
    State st{};
    SNK_HIP_CHECK(hipMemcpyAsync(&st, h->d_state.as<State>() + problem, sizeof(State), hipMemcpyDeviceToHost, h->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (pcg_iterations) *pcg_iterations = st.pcg_iters;
    return SNK_OK;
}

int snk_ba_residuals(snk_ba* h, int problem, double* chi2_per_obs)
{
    SNK_REQUIRE(h != nullptr && h->count > 0, "no problem set");
    SNK_REQUIRE(problem >= 0 && problem < h->count && chi2_per_obs != nullptr, "bad arguments");
    SNK_HIP_CHECK(hipSetDevice(h->device));
    const Opt O = make_opt(h->opt);
    SNK_HIP_CHECK(hipMemsetAsync(h->d_chi2.p, 0, (size_t)std::max(h->tot_orig, 1) * 8, h->stream));
    hipLaunchKernelGGL(point_pass<2>, dim3(std::max(1, ceil_div(h->max_np, 128)), h->count), dim3(128), 0, h->stream, h->arr, O);
    SNK_LAUNCH_CHECK();
    const size_t n = (size_t)h->orig_n[(size_t)problem];
    if (n)
        SNK_HIP_CHECK(hipMemcpyAsync(chi2_per_obs, h->d_chi2.as<double>() + h->orig_off[(size_t)problem], n * 8,
                                     hipMemcpyDeviceToHost, h->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(h->stream));
    return SNK_OK;
}
}
