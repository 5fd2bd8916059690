// Feature grid and projection-guided matchers for gfx950 — replaces
//   Preprocess::computeFeatureGrid / FeatureGrid2::create         (reference Snake/Preprocess/Preprocess.cpp:244-266)
//   Features::GetFeaturesInArea*                                  (reference Snake/Map/Features.cpp:13-76)
//   SnakeORBMatcher::SearchByProjectionFrameFrame2 (coarse)       (reference Snake/Tracking/SnakeORBMatcher.cpp:191-354)
//   SnakeORBMatcher::SearchByProjection2 (fine)                   (reference Snake/Tracking/SnakeORBMatcher.cpp:365-526)
//   SnakeORBMatcher::SearchByProjectionFrameToKeyframe            (reference Snake/Tracking/SnakeORBMatcher.cpp:71-188)
//
// Mapping to the hardware.  The grid orders features x-major by 20-px cell, so the candidates of a
// search window are, per cell column, ONE contiguous index range — no per-point candidate vectors:
// one wavefront per local-map point projects the point (all lanes redundantly, fp64), walks the
// <= (2r/20 + 1) cell columns of its window, the lanes stride over each range, evaluate the gates +
// the 256-bit Hamming distance and keep the two smallest packed keys dist << 20 | index (the
// reference's "strict <, first candidate wins" order is ascending feature index), and a 6-step
// xor-shuffle network merges them.  The reference's serial first-claimant-wins pass becomes an
// integer atomicMin of the local-map index per claimed feature (order independent, deterministic).
// The keyframe matcher is greedy-sequential by definition (a feature taken by point i is gone for
// point i+1), so one wavefront walks the points in order with the taken mask in LDS.
#include "matcher_handle.hpp"

namespace snk
{
namespace
{
using u8  = unsigned char;
using u32 = unsigned int;
using u64 = unsigned long long;

constexpr u32 PJ_IDX_BITS = 20;
constexpr u32 PJ_IDX_MASK = (1u << PJ_IDX_BITS) - 1u;
constexpr u32 PJ_INF_KEY  = (256u << PJ_IDX_BITS) | PJ_IDX_MASK;

struct FrameDev
{
    int n, cols, rows;
    const snk_kp64* kps;
    const uint4* desc;
    const float* right_points;
    const u8* taken;
    const int* cell_start;
    double min_x, min_y, max_x, max_y;
};

struct CamDev
{
    double fx, fy, cx, cy, bf;
    double R[9], t[3], campos[3];
};

struct ScalesDev
{
    float s[32];
    int n;
    double log_f, s_last;
};

__device__ __forceinline__ double det_log(double x)
{
    int e;
    double m = frexp(x, &e);
    if (m < 0.70710678118654752440)
    {
        m *= 2.0;
        e -= 1;
    }
    const double z = (m - 1.0) / (m + 1.0), z2 = z * z;
    double s = 1.0 / 21.0;
#pragma unroll  // fully: 1 / k are then constants (the same values the division gives at run time)
    for (int k = 19; k >= 1; k -= 2) s = s * z2 + 1.0 / (double)k;
    return 2.0 * z * s + (double)e * 0.69314718055994530942;
}

template <int N>
__device__ __forceinline__ double det_exp_terms(double s, double f)
{
    if constexpr (N >= 1)
        return det_exp_terms<N - 1>(1.0 + div_const<N>(s * f), f);
    else
        return s;
}
__device__ __forceinline__ double det_exp(double y)
{
    const double k = floor(y * 1.44269504088896340736 + 0.5);
    const double f = y - k * 0.69314718055994530942;
    // s = 1 + s f / n for n = 16 ... 1
    return ldexp(det_exp_terms<16>(1.0, f), (int)k);
}

__device__ __forceinline__ int cell_coord(double p, double lo, int n)
{
    const int c = (int)floor(div_const<20>(p - lo));
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__device__ __forceinline__ void split_desc(const uint64_t* d, uint4& a, uint4& c)
{
    a = make_uint4((u32)d[0], (u32)(d[0] >> 32), (u32)d[1], (u32)(d[1] >> 32));
    c = make_uint4((u32)d[2], (u32)(d[2] >> 32), (u32)d[3], (u32)(d[3] >> 32));
}

__device__ __forceinline__ void insert2(u32& k1, u32& k2, u32 key)
{
    const u32 hi = k1 < key ? key : k1;
    k1           = k1 < key ? k1 : key;
    k2           = k2 < hi ? k2 : hi;
}
__device__ __forceinline__ void merge2(u32& k1, u32& k2, u32 o1, u32 o2)
{
    const u32 lo = k1 < o1 ? k1 : o1, hi = k1 < o1 ? o1 : k1, m = k2 < o2 ? k2 : o2;
    k2 = hi < m ? hi : m;
    k1 = lo;
}

// Candidate scan of one search window by one wavefront.  MODE 0: radius only; 1: octave window;
// 2: predicted scale.  `taken` may point to LDS (keyframe matcher) or global memory.
// The window's cell range, packed (cx0 | cx1 << 16, cy0 | cy1 << 16) -- cols * rows < 65535 at every entry point: four fp64
// divisions that the batched matchers evaluate once per point in the lane = point phase instead of in every window scan.
__device__ __forceinline__ uint2 window_cells(const FrameDev& F, double ipx, double ipy, double r)
{
    const int cx0 = cell_coord(ipx - r, F.min_x, F.cols), cx1 = cell_coord(ipx + r, F.min_x, F.cols);
    const int cy0 = cell_coord(ipy - r, F.min_y, F.rows), cy1 = cell_coord(ipy + r, F.min_y, F.rows);
    return make_uint2((u32)cx0 | ((u32)cx1 << 16), (u32)cy0 | ((u32)cy1 << 16));
}
template <int MODE, int LANES = 64>
__device__ __forceinline__ void scan_window_cells(const FrameDev& F, const u8* taken, double ipx, double ipy, double disp, uint2 cells,
                                                  double r, double r2, int min_oct, int max_oct, double pred, const uint4& qa,
                                                  const uint4& qc, int lane, u32& k1, u32& k2)
{
    const int cx0 = (int)(cells.x & 0xFFFFu), cx1 = (int)(cells.x >> 16), cy0 = (int)(cells.y & 0xFFFFu), cy1 = (int)(cells.y >> 16);
    for (int cx = cx0; cx <= cx1; ++cx)
    {
        const int lo = F.cell_start[cx * F.rows + cy0], hi = F.cell_start[cx * F.rows + cy1 + 1];
        for (int pid = lo + lane; pid < hi; pid += LANES)  // `lane` = index inside the group of LANES lanes that shares the window
        {
            const snk_kp64 kp = F.kps[pid];
            if (MODE == 1 && (kp.octave < min_oct || kp.octave > max_oct)) continue;
            if (MODE == 2 && fabs(pred - (double)kp.octave) > 1.0) continue;
            const double dx = kp.x - ipx, dy = kp.y - ipy;
            if (!(dx * dx + dy * dy < r2)) continue;
            if (taken[pid]) continue;
            const float rp = F.right_points[pid];
            if (rp > 0)
            {
                double er = fabs(disp - (double)rp);
                if (MODE == 0) er = (double)(float)er;  // `const float er` in SearchByProjectionFrameToKeyframe
                if (er > r * 0.5) continue;
            }
            const uint4 ta = F.desc[(size_t)pid * 2], tc = F.desc[(size_t)pid * 2 + 1];
            const u32 d    = (u32)hamming256(qa, qc, ta, tc);
            if (d < 256u) insert2(k1, k2, (d << PJ_IDX_BITS) | (u32)pid);
        }
    }
#pragma unroll
    for (int off = LANES / 2; off >= 1; off >>= 1)
    {
        const u32 o1 = __shfl_xor(k1, off), o2 = __shfl_xor(k2, off);
        merge2(k1, k2, o1, o2);
    }
}
template <int MODE, int LANES = 64>
__device__ __forceinline__ void scan_window(const FrameDev& F, const u8* taken, double ipx, double ipy, double z, double bf,
                                            double r, double r2, int min_oct, int max_oct, double pred, const uint4& qa,
                                            const uint4& qc, int lane, u32& k1, u32& k2)
{
    scan_window_cells<MODE, LANES>(F, taken, ipx, ipy, ipx - bf / z, window_cells(F, ipx, ipy, r), r, r2, min_oct, max_oct, pred, qa,
                                   qc, lane, k1, k2);
}

// The next (up to) NG set bits of `todo`, one per group of 64 / NG lanes: group g gets the g-th; -1 = none.  Clears them.
template <int NG>
__device__ __forceinline__ int take_groups(u64& todo, int grp)
{
    int res = -1;
#pragma unroll
    for (int k = 0; k < NG; ++k)
    {
        const int s = todo ? __builtin_ctzll(todo) : -1;
        todo &= todo - 1;  // 0 stays 0
        res = grp == k ? s : res;
    }
    return res;
}
__device__ __forceinline__ double shfl_d(double v, int src)
{
    return __hiloint2double(__shfl(__double2hiint(v), src), __shfl(__double2loint(v), src));
}
__device__ __forceinline__ uint4 shfl_u4(const uint4& v, int src)
{
    return make_uint4((u32)__shfl((int)v.x, src), (u32)__shfl((int)v.y, src), (u32)__shfl((int)v.z, src), (u32)__shfl((int)v.w, src));
}

// Lanes that share one window scan in the batched matchers.  A frame has ~1 feature per grid cell, a window a handful of
// candidates per grid column: measured per 256 frames (coarse 1500 + fine 10 000 points + refinement) 16 lanes x 4 points at a time
// 1.08 ms, 8 x 8 0.92, 4 x 16 0.86, 2 x 32 0.86, and ONE lane per window (every lane scans its own point's window, nothing is
// broadcast: 17 lane shuffles per round of points gone) 0.84 -- the default.  Any power of two up to 16 works.
constexpr int PJ_GROUP = 1;
// group broadcasts: with one lane per window a point's numbers are already where they are needed
__device__ __forceinline__ double gsh_d(double v, int src) { return PJ_GROUP == 1 ? v : shfl_d(v, src); }
__device__ __forceinline__ float gsh_f(float v, int src) { return PJ_GROUP == 1 ? v : __shfl(v, src); }
__device__ __forceinline__ int gsh_i(int v, int src) { return PJ_GROUP == 1 ? v : __shfl(v, src); }
__device__ __forceinline__ uint4 gsh_u4(const uint4& v, int src) { return PJ_GROUP == 1 ? v : shfl_u4(v, src); }

// Batch of grid-ordered frames resident on the device (the arrays snk_feature_grid_batch_dev / snk_stereo_match_batch_dev
// leave behind): per-feature arrays are [batch][cap], cell_start is [batch][cols * rows + 1].
struct FramesDev
{
    int cap, cols, rows;
    const int* n;
    const snk_kp64* kps;
    const uint4* desc;
    const float* right_points;
    const u8* taken;
    const int* cell_start;
    double min_x, min_y, max_x, max_y;
};

__device__ __forceinline__ FrameDev frame_of(const FramesDev& B, int b)
{
    FrameDev F;
    F.n            = min(max(B.n[b], 0), B.cap);  // device-side counts cannot be validated by the host: never past the frame's slab / LDS carve
    F.cols         = B.cols;
    F.rows         = B.rows;
    F.kps          = B.kps + (size_t)b * B.cap;
    F.desc         = B.desc + (size_t)b * B.cap * 2;
    F.right_points = B.right_points + (size_t)b * B.cap;
    F.taken        = B.taken + (size_t)b * B.cap;
    F.cell_start   = B.cell_start + (size_t)b * (B.cols * B.rows + 1);
    F.min_x = B.min_x; F.min_y = B.min_y; F.max_x = B.max_x; F.max_y = B.max_y;
    return F;
}

// A frame's arrays copied into the workgroup's LDS (dynamic, 16-byte aligned): every wavefront of the workgroup matches points of
// the same frame and every window scan is a chain of dependent gathers (cell range -> keypoint -> flags -> descriptor) -- out of
// LDS those cost tens of cycles instead of L2 round trips.  ~61 bytes per feature + 4 per grid cell: a 1 000-feature frame is 65 KB.
__device__ __forceinline__ size_t frame_lds_bytes(int cap, int ncell1)
{
    return (((size_t)cap * 24 + 15) & ~(size_t)15) + (size_t)cap * 32 + (((size_t)cap * 4 + 15) & ~(size_t)15) + (((size_t)cap + 15) & ~(size_t)15) +
           (size_t)ncell1 * 4;
}
__device__ __forceinline__ FrameDev stage_frame_lds(const FrameDev& G, unsigned char* lds, int nthreads)
{
    const int n = G.n, ncell1 = G.cols * G.rows + 1;
    unsigned char* p_kps  = lds;
    unsigned char* p_desc = p_kps + (((size_t)n * 24 + 15) & ~(size_t)15);
    unsigned char* p_rp   = p_desc + (size_t)n * 32;
    unsigned char* p_tk   = p_rp + (((size_t)n * 4 + 15) & ~(size_t)15);
    unsigned char* p_cs   = p_tk + (((size_t)n + 15) & ~(size_t)15);
    const int tid = threadIdx.x;
    // (unroll pragmas below: left alone the compiler unrolls these copies eight-fold -- the byte and float copies are one or two
    // iterations per thread anyway -- and their addresses and loads in flight push two or three values of the callers, which hold 64
    // registers for two workgroups per CU, into scratch)
    {
        const u32* src = reinterpret_cast<const u32*>(G.kps);
        u32* dst       = reinterpret_cast<u32*>(p_kps);
#pragma unroll 2
        for (int i = tid; i < n * 6; i += nthreads) dst[i] = src[i];
    }
    {
        const uint4* src = G.desc;
        uint4* dst       = reinterpret_cast<uint4*>(p_desc);
#pragma unroll 2
        for (int i = tid; i < n * 2; i += nthreads) dst[i] = src[i];
    }
    {
        const float* src = G.right_points;
        float* dst       = reinterpret_cast<float*>(p_rp);
#pragma unroll 1
        for (int i = tid; i < n; i += nthreads) dst[i] = src[i];
        u8* dt = p_tk;
#pragma unroll 1
        for (int i = tid; i < n; i += nthreads) dt[i] = G.taken[i];
    }
    {
        int* dst = reinterpret_cast<int*>(p_cs);
#pragma unroll 1
        for (int i = tid; i < ncell1; i += nthreads) dst[i] = G.cell_start[i];
    }
    FrameDev F     = G;
    F.kps          = reinterpret_cast<const snk_kp64*>(p_kps);
    F.desc         = reinterpret_cast<const uint4*>(p_desc);
    F.right_points = reinterpret_cast<const float*>(p_rp);
    F.taken        = p_tk;
    F.cell_start   = reinterpret_cast<const int*>(p_cs);
    __syncthreads();
    return F;
}

// Broadcast of one lane's value to the wavefront (the lane index is wave-uniform: it comes from a scalar bit scan).
__device__ __forceinline__ double bcast_d(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ u32 bcast_u(u32 v, int src) { return (u32)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ uint4 bcast_u4(const uint4& v, int src)
{
    return make_uint4(bcast_u(v.x, src), bcast_u(v.y, src), bcast_u(v.z, src), bcast_u(v.w, src));
}

// The matchers work on 64 local-map points per wavefront in two phases.  Phase 1, lane = point: projection, culls, search
// radius (fp64, with the fixed-order log / exp series of the fine matcher) -- ONCE per point; the first version had all 64 lanes
// of a wavefront repeat this per point and was bound by exactly that (790 vector instructions per point, VALU 100 % busy,
// PMC r02t).  Phase 2: PJ_GROUP lanes share one window scan; with PJ_GROUP = 1 (the measured optimum, r02u) every lane scans
// the window of its own point and nothing is broadcast.  Same arithmetic per point as before.

// coarse (SnakeORBMatcher.cpp:221-318): 64 points starting at i0; results to best[] / bins[] (indexed like pts)
__device__ __forceinline__ void coarse_wave64(const FrameDev& F, const CamDev& C, const ScalesDev& S, const snk_lm_coarse* __restrict__ pts,
                                              int m, int i0, int ppw, float th, int feature_error, int direction, int lane,
                                              int* __restrict__ best, int* __restrict__ bins, int* claim = nullptr)
{
    const int i   = i0 + lane;
    const bool in = lane < ppw && i < m;  // ppw points per wavefront: 64 for large batches, fewer when the points are few
    const snk_lm_coarse* lp = pts + (in ? i : i0);
    const double px = lp->pos[0], py = lp->pos[1], pz = lp->pos[2];
    const int oct  = lp->octave;
    const float ang = lp->angle;
    uint4 qa, qc;
    split_desc(lp->desc, qa, qc);
    const double pcx = C.R[0] * px + C.R[1] * py + C.R[2] * pz + C.t[0];
    const double pcy = C.R[3] * px + C.R[4] * py + C.R[5] * pz + C.t[1];
    const double z   = C.R[6] * px + C.R[7] * py + C.R[8] * pz + C.t[2];
    const double ipx = C.fx * pcx / z + C.cx, ipy = C.fy * pcy / z + C.cy;
    bool ok = in && z > 0 && ipx >= F.min_x && ipx < F.max_x && ipy >= F.min_y && ipy < F.max_y;
    if (ok)
    {
        const double POx = C.campos[0] - px, POy = C.campos[1] - py, POz = C.campos[2] - pz;
        const double dist    = sqrt(POx * POx + POy * POy + POz * POz);
        const double viewCos = (POx * lp->normal[0] + POy * lp->normal[1] + POz * lp->normal[2]) / dist;
        ok = !(viewCos < 0.5);
    }
    int lvl = oct;
    lvl     = lvl < 0 ? 0 : (lvl >= S.n ? S.n - 1 : lvl);
    float r = th;
    r *= S.s[lvl];
    // the window's cell range and the expected disparity, once per point (lanes that fail the culls compute on harmless values)
    const uint2 cells = window_cells(F, ok ? ipx : F.min_x, ok ? ipy : F.min_y, (double)r);
    const double disp = ipx - C.bf / (ok ? z : 1.0);
    // results go through the wavefront's LDS slice (served in program order): default "no match", the group leaders overwrite
    __shared__ int s_res[16][2][64];  // up to 16 wavefronts per workgroup (the frame-resident kernels)
    int* my_res = s_res[(threadIdx.x >> 6) & 15][0];
    int* my_bin = s_res[(threadIdx.x >> 6) & 15][1];
    my_res[lane] = -1;
    my_bin[lane] = 0;
    // Phase 2: a window holds a few dozen candidates, so 16 lanes share one and four points are scanned at a time
    const int grp = lane / PJ_GROUP, sub = lane % PJ_GROUP;
    u64 todo = __builtin_amdgcn_ballot_w64(ok);
    for (int pass = 0; PJ_GROUP == 1 ? pass < 1 : todo != 0; ++pass)
    {
        const int src  = PJ_GROUP == 1 ? (ok ? lane : -1) : take_groups<(PJ_GROUP == 1 ? 1 : 64 / PJ_GROUP)>(todo, grp);
        const bool has = src >= 0;
        const int sl   = has ? src : 0;
        const double bx = gsh_d(ipx, sl), by = gsh_d(ipy, sl), bdisp = gsh_d(disp, sl);
        const uint2 bcells = make_uint2((u32)gsh_i((int)cells.x, sl), (u32)gsh_i((int)cells.y, sl));
        const float br  = gsh_f(r, sl);
        const int boct  = gsh_i(oct, sl);
        const float bang = gsh_f(ang, sl);
        const uint4 ba = gsh_u4(qa, sl), bc = gsh_u4(qc, sl);
        if (has)
        {
            int mn, mx;
            if (direction == 1) { mn = boct - 1; mx = 100; }
            else if (direction == 2) { mn = 0; mx = boct; }
            else { mn = boct - 1; mx = boct + 1; }
            u32 k1 = PJ_INF_KEY, k2 = PJ_INF_KEY;
            scan_window_cells<1, PJ_GROUP>(F, F.taken, bx, by, bdisp, bcells, (double)br, (double)br * (double)br, mn, mx, 0.0, ba, bc, sub, k1,
                                     k2);
            const int bd = (int)(k1 >> PJ_IDX_BITS);
            if (bd <= feature_error && k1 != PJ_INF_KEY)
            {
                int res   = (int)(k1 & PJ_IDX_MASK);
                float rot = bang - F.kps[res].angle;
                if (rot < 0.0f) rot += 360.0f;
                int b = (int)roundf(rot * (1.0f / 30));
                if (b == 30) b = 0;
                // the reference asserts 0 <= bin < HISTO_LENGTH (:316); angles outside [0, 360) or NaN have no bin:
                // such a point is left unmatched instead of indexing outside the histogram
                if (!(b >= 0 && b < 30)) { res = -1; b = 0; }
                if (sub == 0)
                {
                    my_res[src] = res;
                    my_bin[src] = b;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (in)
    {
        const int res = my_res[lane];
        best[i] = res;
        bins[i] = my_bin[lane];
        if (claim != nullptr && res >= 0 && !F.taken[res]) atomicMin(&claim[res], i);  // the fused resolution (coarse_frame_kernel)
    }
}

// coarse: best[i] = feature index or -1, bins[i] = rotation bin
__global__ __launch_bounds__(256) void coarse_kernel(FrameDev F, CamDev C, ScalesDev S, const snk_lm_coarse* __restrict__ pts,
                                                     int m, int ppw, float th, int feature_error, int direction,
                                                     int* __restrict__ best, int* __restrict__ bins)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i0 = (blockIdx.x * 4 + wave) * ppw;
    if (i0 >= m) return;
    coarse_wave64(F, C, S, pts, m, i0, ppw, th, feature_error, direction, lane, best, bins);
}

// batched form: blockIdx.y = frame; points [batch][m_cap], counts m_dev[batch]
__global__ __launch_bounds__(256) void coarse_batch_kernel(FramesDev Fb, const CamDev* __restrict__ cams, ScalesDev S,
                                                           const snk_lm_coarse* __restrict__ pts, const int* __restrict__ m_dev,
                                                           int m_cap, int ppw, float th, int feature_error, int direction,
                                                           int* __restrict__ best, int* __restrict__ bins)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.y, i0 = (blockIdx.x * 4 + wave) * ppw;
    const int m = min(m_dev[b], m_cap);
    if (i0 >= m) return;
    const FrameDev F = frame_of(Fb, b);
    const CamDev C   = cams[b];
    coarse_wave64(F, C, S, pts + (size_t)b * m_cap, m, i0, ppw, th, feature_error, direction, lane, best + (size_t)b * m_cap,
                  bins + (size_t)b * m_cap);
}

// fine (SnakeORBMatcher.cpp:388-512): 64 points starting at i0; best[], visible[] and pts[].valid (in place).
// WRITE_VALID = false: the records stay untouched.  The flag the reference leaves in lmp.valid equals visible[]: a point keeps valid = 1
// exactly when it passes every cull, which is where IncreaseVisible() is called (:431).  Written back, it is one byte into each 96-byte
// record -- a 32-byte sector written per point, six times the bytes of the results proper.
template <bool WRITE_VALID = true>
__device__ __forceinline__ void fine_wave64(const FrameDev& F, const CamDev& C, const ScalesDev& S, snk_lm_fine* __restrict__ pts, int m,
                                            int i0, int ppw, float th, float ratio, int lane, int* __restrict__ best, u8* __restrict__ visible,
                                            int* claim = nullptr)
{
    const int i   = i0 + lane;
    const bool in = lane < ppw && i < m;
    const snk_lm_fine* lp = pts + (in ? i : i0);
    const double px = lp->pos[0], py = lp->pos[1], pz = lp->pos[2];
    uint4 qa, qc;
    split_desc(lp->desc, qa, qc);
    u8 vis = 0, valid = in ? lp->valid : (u8)0;
    double ipx = 0, ipy = 0, z = 1, prediction = 0;
    float r = 0;
    bool scan = false;
    if (valid)
    {
        const double pcx = C.R[0] * px + C.R[1] * py + C.R[2] * pz + C.t[0];
        const double pcy = C.R[3] * px + C.R[4] * py + C.R[5] * pz + C.t[1];
        z   = C.R[6] * px + C.R[7] * py + C.R[8] * pz + C.t[2];
        ipx = C.fx * pcx / z + C.cx;
        ipy = C.fy * pcy / z + C.cy;
        if (z < 0 || !(ipx >= F.min_x && ipx < F.max_x && ipy >= F.min_y && ipy < F.max_y))
            valid = 0;
        else
        {
            const double POx = C.campos[0] - px, POy = C.campos[1] - py, POz = C.campos[2] - pz;
            const double dist = sqrt(POx * POx + POy * POy + POz * POz);
            int rl            = lp->reference_scale_level;
            rl                = rl < 0 ? 0 : (rl >= S.n ? S.n - 1 : rl);
            const double sref = (double)S.s[rl];
            const double max_dist = 1.2 * (double)lp->reference_depth * sref;
            const double min_dist = 0.8 * (double)lp->reference_depth * sref / S.s_last;
            const double viewCos  = (POx * lp->normal[0] + POy * lp->normal[1] + POz * lp->normal[2]) / dist;
            if (dist < min_dist || dist > max_dist || viewCos < 0.5)
                valid = 0;
            else
            {
                vis = 1;
                const float vcf = (float)viewCos;
                r               = (double)vcf > 0.998 ? 2.5f : 4.0f;
                if (th != 1.0f) r *= th;
                prediction = (double)lp->reference_scale_level + det_log((double)lp->reference_depth / dist) / S.log_f;
                if (prediction < 0.0) prediction = 0.0;
                if (prediction > (double)(S.n - 1)) prediction = (double)(S.n - 1);
                r    = (float)((double)r * det_exp(prediction * S.log_f));
                scan = true;
            }
        }
    }
    const uint2 cells = window_cells(F, scan ? ipx : F.min_x, scan ? ipy : F.min_y, (double)r);
    const double disp = ipx - C.bf / (scan ? z : 1.0);
    __shared__ int s_resf[16][64];
    int* my_res = s_resf[(threadIdx.x >> 6) & 15];
    my_res[lane] = -1;
    const int grp = lane / PJ_GROUP, sub = lane % PJ_GROUP;
    u64 todo = __builtin_amdgcn_ballot_w64(scan);
    for (int pass = 0; PJ_GROUP == 1 ? pass < 1 : todo != 0; ++pass)
    {
        const int src  = PJ_GROUP == 1 ? (scan ? lane : -1) : take_groups<(PJ_GROUP == 1 ? 1 : 64 / PJ_GROUP)>(todo, grp);
        const bool has = src >= 0;
        const int sl   = has ? src : 0;
        const double bx = gsh_d(ipx, sl), by = gsh_d(ipy, sl), bdisp = gsh_d(disp, sl), bpred = gsh_d(prediction, sl);
        const uint2 bcells = make_uint2((u32)gsh_i((int)cells.x, sl), (u32)gsh_i((int)cells.y, sl));
        const float br  = gsh_f(r, sl);
        const uint4 ba = gsh_u4(qa, sl), bc = gsh_u4(qc, sl);
        if (has)
        {
            u32 k1 = PJ_INF_KEY, k2 = PJ_INF_KEY;
            scan_window_cells<2, PJ_GROUP>(F, F.taken, bx, by, bdisp, bcells, (double)br, (double)br * (double)br, 0, 0, bpred, ba, bc, sub, k1,
                                     k2);
            const int bd = (int)(k1 >> PJ_IDX_BITS);
            if (k1 != PJ_INF_KEY && bd <= 100)
            {
                const int bi  = (int)(k1 & PJ_IDX_MASK);
                const int bd2 = (int)(k2 >> PJ_IDX_BITS);
                const int l1  = F.kps[bi].octave;
                const int l2  = k2 != PJ_INF_KEY ? F.kps[k2 & PJ_IDX_MASK].octave : -1;
                if (!(l1 == l2 && (float)bd > ratio * (float)bd2) && sub == 0) my_res[src] = bi;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (in)
    {
        const int res = my_res[lane];
        best[i]       = res;
        visible[i]    = vis;
        if (WRITE_VALID) pts[i].valid = valid;
        if (claim != nullptr && res >= 0 && !F.taken[res]) atomicMin(&claim[res], i);  // the fused resolution (fine_frame_kernel)
    }
}

__global__ __launch_bounds__(256) void fine_kernel(FrameDev F, CamDev C, ScalesDev S, snk_lm_fine* __restrict__ pts, int m, int ppw,
                                                   float th, float ratio, int* __restrict__ best, u8* __restrict__ visible)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i0 = (blockIdx.x * 4 + wave) * ppw;
    if (i0 >= m) return;
    fine_wave64<false>(F, C, S, pts, m, i0, ppw, th, ratio, lane, best, visible);  // the host entry point sets pts[].valid from visible[]
}

template <bool WRITE_VALID>
__global__ __launch_bounds__(256) void fine_batch_kernel(FramesDev Fb, const CamDev* __restrict__ cams, ScalesDev S,
                                                         snk_lm_fine* __restrict__ pts, const int* __restrict__ m_dev, int m_cap, int ppw,
                                                         float th, float ratio, int* __restrict__ best, u8* __restrict__ visible)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.y, i0 = (blockIdx.x * 4 + wave) * ppw;
    const int m = min(m_dev[b], m_cap);
    if (i0 >= m) return;
    const FrameDev F = frame_of(Fb, b);
    const CamDev C   = cams[b];
    fine_wave64<WRITE_VALID>(F, C, S, pts + (size_t)b * m_cap, m, i0, ppw, th, ratio, lane, best + (size_t)b * m_cap, visible + (size_t)b * m_cap);
}

// first claimant (lowest local-map index) of every feature wins; coarse additionally applies the
// rotation-histogram filter.  One workgroup per frame.
// CLAIM: int* in global memory (any number of features) or in LDS (the batched kernel: the claims of a frame's features fit, and
// an LDS atomic per local-map point instead of a global one is most of the kernel's time)
// Second half of the resolution: the claims are in (claim[f] = lowest local-map index that wants feature f); every thread of the
// workgroup calls it.  Starts with a barrier, so it may follow the claiming pass directly.
template <typename CLAIM>
__device__ __forceinline__ void resolve_finish(const int* __restrict__ best, const int* __restrict__ bins, int m, const u8* taken, CLAIM claim,
                                               int with_rotation, int* __restrict__ match_idx, int* __restrict__ n_out)
{
    __shared__ int hist[30];
    __shared__ int keep[3];
    __shared__ int count;
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (tid < 30) hist[tid] = 0;
    if (tid == 0) count = 0;
    __syncthreads();
    for (int i = tid; i < m; i += nthr)
    {
        const int b   = best[i];
        const bool win = b >= 0 && !taken[b] && claim[b] == i;
        match_idx[i]   = win ? b : -1;
        if (win && with_rotation) atomicAdd(&hist[bins[i]], 1);
    }
    __syncthreads();
    if (tid == 0)
    {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        if (with_rotation)
        {
            int max1 = 0, max2 = 0, max3 = 0;  // ComputeThreeMaxima, SnakeORBMatcher.cpp:27-68
            for (int k = 0; k < 30; k++)
            {
                const int s = hist[k];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = k; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = k; }
                else if (s > max3) { max3 = s; ind3 = k; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        }
        keep[0] = ind1;
        keep[1] = ind2;
        keep[2] = ind3;
    }
    __syncthreads();
    int local = 0;
    for (int i = tid; i < m; i += nthr)
    {
        int v = match_idx[i];
        if (v >= 0 && with_rotation)
        {
            const int b = bins[i];
            if (b != keep[0] && b != keep[1] && b != keep[2])
            {
                v            = -1;
                match_idx[i] = -1;
            }
        }
        local += v >= 0 ? 1 : 0;
    }
    atomicAdd(&count, local);
    __syncthreads();
    if (tid == 0) *n_out = count;
}
template <typename CLAIM>
__device__ __forceinline__ void resolve_body(const int* __restrict__ best, const int* __restrict__ bins, int m,
                                             const u8* __restrict__ taken, CLAIM claim, int n_feat,
                                             int with_rotation, int* __restrict__ match_idx, int* __restrict__ n_out)
{
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int f = tid; f < n_feat; f += nthr) claim[f] = 0x7FFFFFFF;
    __syncthreads();
    for (int i = tid; i < m; i += nthr)
    {
        const int b = best[i];
        if (b >= 0 && !taken[b]) atomicMin(&claim[b], i);
    }
    resolve_finish(best, bins, m, taken, claim, with_rotation, match_idx, n_out);
}

__global__ __launch_bounds__(256) void resolve_kernel(const int* __restrict__ best, const int* __restrict__ bins, int m,
                                                      const u8* __restrict__ taken, int* __restrict__ claim, int n_feat,
                                                      int with_rotation, int* __restrict__ match_idx, int* __restrict__ n_out)
{
    resolve_body(best, bins, m, taken, claim, n_feat, with_rotation, match_idx, n_out);
}

// LDS_CLAIM: dynamic LDS of Fb.cap ints holds the claims (frames up to RESOLVE_LDS_FEATURES features); 1024 threads per frame
constexpr int RESOLVE_LDS_FEATURES = 32768;
template <bool LDS_CLAIM>
__global__ __launch_bounds__(1024) void resolve_batch_kernel(const int* __restrict__ best, const int* __restrict__ bins,
                                                             const int* __restrict__ m_dev, int m_cap, FramesDev Fb,
                                                             int* __restrict__ claim, int with_rotation,
                                                             int* __restrict__ match_idx, int* __restrict__ n_out)
{
    extern __shared__ __attribute__((aligned(16))) int s_claim[];
    const int b = blockIdx.x;
    const int m = min(m_dev[b], m_cap);
    if constexpr (LDS_CLAIM)
        resolve_body(best + (size_t)b * m_cap, bins ? bins + (size_t)b * m_cap : nullptr, m, Fb.taken + (size_t)b * Fb.cap, s_claim,
                     min(Fb.n[b], Fb.cap), with_rotation, match_idx + (size_t)b * m_cap, n_out + b);
    else
        resolve_body(best + (size_t)b * m_cap, bins ? bins + (size_t)b * m_cap : nullptr, m, Fb.taken + (size_t)b * Fb.cap,
                     claim + (size_t)b * Fb.cap, min(Fb.n[b], Fb.cap), with_rotation, match_idx + (size_t)b * m_cap, n_out + b);
    // entries past the frame's point count read as "no match"
    for (int i = m + threadIdx.x; i < m_cap; i += blockDim.x) match_idx[(size_t)b * m_cap + i] = -1;
}

// Frame-resident forms of the two batched matchers: 1024 threads (16 wavefronts, 1024 points) per workgroup, the frame in LDS.
// Eight wavefronts per SIMD (<= 64 VGPRs, three dwords spilled outside the scan): TWO workgroups per CU.  At the 68 / 69 VGPRs the
// kernels take unbounded only one workgroup fits a CU -- four wavefronts per SIMD against a window scan that is a chain of dependent
// LDS reads.  Round 5, tracking leg of bench.py (1024 frames): 853 k -> 911 k frames/s (profiles/r05/r05z_track_experiments.json).
#ifndef SNK_TRACK_FRAME_WAVES_PER_EU
#define SNK_TRACK_FRAME_WAVES_PER_EU 8
#endif
// FUSED resolution (claim_off >= 0, only with ONE workgroup per frame, which is what large batches get): the claims of the frame's
// features live in LDS behind the frame (claim_off bytes into the carve), every wavefront claims as it produces a point's best feature, and
// after the last chunk the workgroup finishes like resolve_batch_kernel does (first claimant wins, coarse: rotation histogram) -- the
// separate launch, its three passes over best[] from HBM and its barriers are gone.  A thread resolves the points it matched itself
// (point i is thread i mod 1024 in both halves).  Round 5: resolve_batch_kernel was 26 + 52 us of the 1.13 ms chain per 1024 frames.
__global__ __launch_bounds__(1024, SNK_TRACK_FRAME_WAVES_PER_EU) void coarse_frame_kernel(FramesDev Fb, const CamDev* __restrict__ cams, ScalesDev S,
                                                            const snk_lm_coarse* __restrict__ pts, const int* __restrict__ m_dev,
                                                            int m_cap, float th, int feature_error, int direction,
                                                            int* __restrict__ best, int* __restrict__ bins, int claim_off,
                                                            int* __restrict__ match_idx, int* __restrict__ n_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char frame_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int m = min(m_dev[b], m_cap);
    const bool fused = claim_off >= 0;
    if (!fused && blockIdx.x * 1024 >= m) return;  // whole workgroup
    int* claim = fused ? reinterpret_cast<int*>(frame_lds + claim_off) : nullptr;
    const FrameDev G = frame_of(Fb, b);
    if (fused)
        for (int f = threadIdx.x; f < G.n; f += 1024) claim[f] = 0x7FFFFFFF;  // the barrier at the end of the staging covers it
    // The frame's camera (20 doubles) in LDS, not in scalar registers: as a by-value copy it took 40 of the ~100 scalar registers, the
    // allocator parked scalars in lanes of vector registers and vector registers in scratch (12 bytes per lane, round-5 review);
    // the projection reads it once per point as broadcast LDS reads.
    __shared__ CamDev s_cam;
    if (threadIdx.x < sizeof(CamDev) / 8) reinterpret_cast<double*>(&s_cam)[threadIdx.x] = reinterpret_cast<const double*>(cams + b)[threadIdx.x];
    const FrameDev F = stage_frame_lds(G, frame_lds, 1024);  // once per workgroup; the chunks of its share follow (its closing barrier covers s_cam)
    const CamDev& C = s_cam;
    for (int chunk = blockIdx.x; chunk * 1024 < m; chunk += gridDim.x)
    {
        const int i0 = (chunk * 16 + wave) * 64;
        if (i0 >= m) break;
        coarse_wave64(F, C, S, pts + (size_t)b * m_cap, m, i0, 64, th, feature_error, direction, lane, best + (size_t)b * m_cap,
                      bins + (size_t)b * m_cap, claim);
    }
    if (fused)
    {
        resolve_finish(best + (size_t)b * m_cap, bins + (size_t)b * m_cap, m, F.taken, claim, 1, match_idx + (size_t)b * m_cap, n_out + b);
        for (int i = m + threadIdx.x; i < m_cap; i += 1024) match_idx[(size_t)b * m_cap + i] = -1;
    }
}

template <bool WRITE_VALID>
__global__ __launch_bounds__(1024, SNK_TRACK_FRAME_WAVES_PER_EU) void fine_frame_kernel(FramesDev Fb, const CamDev* __restrict__ cams, ScalesDev S,
                                                          snk_lm_fine* __restrict__ pts, const int* __restrict__ m_dev, int m_cap, float th,
                                                          float ratio, int* __restrict__ best, u8* __restrict__ visible, int claim_off,
                                                          int* __restrict__ match_idx, int* __restrict__ n_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char frame_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int m = min(m_dev[b], m_cap);
    const bool fused = claim_off >= 0;
    if (!fused && blockIdx.x * 1024 >= m) return;  // whole workgroup
    // The frame (65 KB for 1000 features) is staged ONCE per workgroup and the workgroup walks the chunks blockIdx.x, + gridDim.x, ...
    // of the frame's local map: with one chunk per workgroup the ten workgroups of a 10 000-point local map each staged the same
    // frame -- 40 % of a workgroup's input bytes (round 4; grid.x is chosen by the launch, SNK_TRACK_FRAME_WGS).
    int* claim = fused ? reinterpret_cast<int*>(frame_lds + claim_off) : nullptr;
    const FrameDev G = frame_of(Fb, b);
    if (fused)
        for (int f = threadIdx.x; f < G.n; f += 1024) claim[f] = 0x7FFFFFFF;
    // see coarse_frame_kernel: the camera in LDS instead of 40 scalar registers -- and here the pyramid scales too (37 more: the fine
    // matcher indexes them per lane and reads log_f / s_last per point); 44 bytes of scratch per lane before
    __shared__ CamDev s_cam;
    __shared__ ScalesDev s_scales;
    if (threadIdx.x < sizeof(CamDev) / 8) reinterpret_cast<double*>(&s_cam)[threadIdx.x] = reinterpret_cast<const double*>(cams + b)[threadIdx.x];
    if (threadIdx.x == 64) s_scales = S;
    const FrameDev F = stage_frame_lds(G, frame_lds, 1024);
    const CamDev& C = s_cam;
    const ScalesDev& SL = s_scales;
    for (int chunk = blockIdx.x; chunk * 1024 < m; chunk += gridDim.x)
    {
        const int i0 = (chunk * 16 + wave) * 64;
        if (i0 >= m) break;  // chunks ascend: nothing further for this wavefront (the barriers of the fused resolution come after the loop)
        fine_wave64<WRITE_VALID>(F, C, SL, pts + (size_t)b * m_cap, m, i0, 64, th, ratio, lane, best + (size_t)b * m_cap, visible + (size_t)b * m_cap,
                                 claim);
    }
    if (fused)
    {
        resolve_finish(best + (size_t)b * m_cap, (const int*)nullptr, m, F.taken, claim, 0, match_idx + (size_t)b * m_cap, n_out + b);
        for (int i = m + threadIdx.x; i < m_cap; i += 1024) match_idx[(size_t)b * m_cap + i] = -1;
    }
}

// CamDev of every frame of a batch from its pose (same operations, in the same order, as make_cam on the host)
__global__ void cam_batch_kernel(snk_camera cam, const double* __restrict__ poses, int batch, CamDev* __restrict__ out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double* pose = poses + (size_t)b * 7;
    CamDev c;
    c.fx = cam.fx; c.fy = cam.fy; c.cx = cam.cx; c.cy = cam.cy; c.bf = cam.bf;
    const double x = pose[0], y = pose[1], z = pose[2], w = pose[3];
    double* R = c.R;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
    c.t[0] = pose[4]; c.t[1] = pose[5]; c.t[2] = pose[6];
    c.campos[0] = -(R[0] * c.t[0] + R[3] * c.t[1] + R[6] * c.t[2]);
    c.campos[1] = -(R[1] * c.t[0] + R[4] * c.t[1] + R[7] * c.t[2]);
    c.campos[2] = -(R[2] * c.t[0] + R[5] * c.t[1] + R[8] * c.t[2]);
    out[b] = c;
}

// taken[b][match_idx[b][i]] = 1: the adaptor's mvpMapPoints[idx] = mp after a matcher call, kept on the device
__global__ void mark_taken_kernel(const int* __restrict__ match_idx, const int* __restrict__ m_dev, int m_cap, u8* __restrict__ taken,
                                  int cap)
{
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(m_dev[b], m_cap)) return;
    const int f = match_idx[(size_t)b * m_cap + i];
    if (f >= 0 && f < cap) taken[(size_t)b * cap + f] = 1;
}

// greedy sequential keyframe matcher: one wavefront, taken mask in LDS
constexpr int KF_MAX_FEATURES = 49152;
__global__ __launch_bounds__(64) void keyframe_kernel(FrameDev F, CamDev C, const double* __restrict__ pos,
                                                      const uint4* __restrict__ desc, const u8* __restrict__ skip, int m,
                                                      float th, int feature_error, int* __restrict__ match_idx,
                                                      int* __restrict__ n_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char taken2[];
    const int lane = threadIdx.x;
    for (int f = lane; f < F.n; f += 64) taken2[f] = F.taken[f];
    __builtin_amdgcn_wave_barrier();
    int matches = 0;
    const double r = (double)th, r2 = (double)(th * th);
    for (int i = 0; i < m; ++i)
    {
        int result = -1;
        if (!skip[i])
        {
            const double* p  = pos + (size_t)i * 3;
            const double pcx = C.R[0] * p[0] + C.R[1] * p[1] + C.R[2] * p[2] + C.t[0];
            const double pcy = C.R[3] * p[0] + C.R[4] * p[1] + C.R[5] * p[2] + C.t[1];
            const double z   = C.R[6] * p[0] + C.R[7] * p[1] + C.R[8] * p[2] + C.t[2];
            const double ipx = C.fx * pcx / z + C.cx, ipy = C.fy * pcy / z + C.cy;
            if (z > 0 && ipx >= F.min_x && ipx < F.max_x && ipy >= F.min_y && ipy < F.max_y)
            {
                const uint4 qa = desc[(size_t)i * 2], qc = desc[(size_t)i * 2 + 1];
                u32 k1 = PJ_INF_KEY, k2 = PJ_INF_KEY;
                scan_window<0>(F, taken2, ipx, ipy, z, C.bf, r, r2, 0, 0, 0.0, qa, qc, lane, k1, k2);
                if (k1 != PJ_INF_KEY && (int)(k1 >> PJ_IDX_BITS) <= feature_error) result = (int)(k1 & PJ_IDX_MASK);
            }
        }
        if (result >= 0)
        {
            if (lane == 0) taken2[result] = 1;
            matches++;
        }
        if (lane == 0) match_idx[i] = result;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) *n_out = matches;
}

// ---- feature grid: sort (cell << 16 | index) in LDS, derive the permutation and the cell ranges ----
// one workgroup per image; n_dev == nullptr: single image with n_host features
__global__ __launch_bounds__(256) void grid_kernel(const snk_kp64* __restrict__ kps, const int* __restrict__ n_dev, int n_host,
                                                   int cap, double min_x, double min_y, int cols, int rows,
                                                   int* __restrict__ perm, int* __restrict__ order, int* __restrict__ cell_start)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    u32* keys     = reinterpret_cast<u32*>(gsm);
    const int tid = threadIdx.x;
    const int b   = blockIdx.x;
    int n         = n_dev ? n_dev[b] : n_host;
    n             = n < cap ? n : cap;
    int n_pow2    = 2;
    while (n_pow2 < n) n_pow2 <<= 1;
    kps += (size_t)b * cap;
    perm += (size_t)b * cap;
    if (order) order += (size_t)b * cap;
    const int ncell = cols * rows;
    cell_start += (size_t)b * (ncell + 1);
    for (int i = tid; i < n_pow2; i += 256)
    {
        u32 k = 0xFFFFFFFFu;
        if (i < n)
        {
            const snk_kp64 kp = kps[i];
            const int cell    = cell_coord(kp.x, min_x, cols) * rows + cell_coord(kp.y, min_y, rows);
            k                 = ((u32)cell << 16) | (u32)i;
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= n_pow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1)
        {
            for (int t = tid; t < (n_pow2 >> 1); t += 256)
            {
                const int i   = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int ixj = i | j;
                const bool up = (i & k) == 0;
                const u32 x = keys[i], y = keys[ixj];
                if ((x > y) == up)
                {
                    keys[i]   = y;
                    keys[ixj] = x;
                }
            }
            __syncthreads();
        }
    for (int p = tid; p < n; p += 256)
    {
        const u32 k       = keys[p];
        perm[k & 0xFFFFu] = p;
        if (order) order[p] = (int)(k & 0xFFFFu);
        const int c  = (int)(k >> 16);
        const int cp = p == 0 ? -1 : (int)(keys[p - 1] >> 16);
        for (int q = cp + 1; q <= c; ++q) cell_start[q] = p;  // first feature of cell c (and of the empty cells before it)
    }
    const int last = n == 0 ? -1 : (int)(keys[n - 1] >> 16);
    for (int q = last + 1 + tid; q <= ncell; q += 256) cell_start[q] = n;
}

// The same grid by counting instead of sorting: cell histogram (LDS atomics), exclusive scan = cell_start, members dropped into their
// cell's segment in arrival order, final position of a feature = segment start + number of members with a smaller index (cells hold
// one or two features on average).  Five barriers instead of the 55 compare-exchange stages of the bitonic network: 22 -> ~5 us for
// the one frame of a per-frame call (the network's time is latency, not work).  Same perm / order / cell_start.  LDS: 2 (ncell + 1)
// + 2 n ints (GRID_COUNT_LDS_MAX bounds it; larger grids keep the network).
constexpr int GRID_COUNT_LDS_MAX = 96 * 1024;
__global__ __launch_bounds__(256) void grid_count_kernel(const snk_kp64* __restrict__ kps, const int* __restrict__ n_dev, int n_host, int cap,
                                                         double min_x, double min_y, int cols, int rows, int* __restrict__ perm,
                                                         int* __restrict__ order, int* __restrict__ cell_start)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    __shared__ int s_wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b   = blockIdx.x;
    int n         = n_dev ? n_dev[b] : n_host;
    n             = n < cap ? n : cap;
    kps += (size_t)b * cap;
    perm += (size_t)b * cap;
    if (order) order += (size_t)b * cap;
    const int ncell = cols * rows;
    cell_start += (size_t)b * (ncell + 1);
    int* start  = reinterpret_cast<int*>(gsm);  // [ncell + 1]: counts, then exclusive starts
    int* fill   = start + ncell + 1;            // [ncell + 1]: next free slot of every cell
    int* cellof = fill + ncell + 1;             // [n]
    int* seg    = cellof + n;                   // [n]: members of every cell, arrival order
    for (int c = tid; c <= ncell; c += 256) start[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256)
    {
        const snk_kp64 kp = kps[i];
        const int cell    = cell_coord(kp.x, min_x, cols) * rows + cell_coord(kp.y, min_y, rows);
        cellof[i]         = cell;
        atomicAdd(&start[cell], 1);
    }
    __syncthreads();
    // exclusive scan of the ncell counts: a contiguous chunk per thread, the 256 chunk sums scanned by wavefront shuffles
    {
        const int chunk = (ncell + 256) / 256;  // covers index ncell too
        const int c0 = tid * chunk, c1 = min(c0 + chunk, ncell + 1);
        int sum = 0;
        for (int c = c0; c < c1; ++c) sum += start[c];
        int inc = sum;
        inc = wave_scan_incl_dpp(inc);
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        int base = inc - sum;
        for (int w = 0; w < wave; ++w) base += s_wsum[w];
        for (int c = c0; c < c1; ++c)
        {
            const int v = start[c];
            start[c]    = base;
            fill[c]     = base;
            cell_start[c] = base;  // start[ncell] = n: the end marker
            base += v;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) seg[atomicAdd(&fill[cellof[i]], 1)] = i;
    __syncthreads();
    for (int i = tid; i < n; i += 256)
    {
        const int c = cellof[i], s0 = start[c], s1 = start[c + 1];
        int r = s0;
        for (int q = s0; q < s1; ++q) r += seg[q] < i ? 1 : 0;
        perm[i] = r;
        if (order) order[r] = i;
    }
}
bool grid_count_fits(int n, int ncell) { return ((size_t)2 * (ncell + 1) + (size_t)2 * n) * 4 <= (size_t)GRID_COUNT_LDS_MAX; }
bool grid_use_network()
{
    static const bool v = getenv("SNK_GRID_NETWORK") != nullptr;  // A/B, tests: the bitonic network also where the counting form fits
    return v;
}

// out[p] = in[order[p]] for the rectified keypoints and the descriptors (Preprocess.cpp:254-260)
__global__ __launch_bounds__(256) void reorder_kernel(const int* __restrict__ order, const int* __restrict__ n_dev, int cap,
                                                      const snk_kp64* __restrict__ kin, const uint4* __restrict__ din,
                                                      snk_kp64* __restrict__ kout, uint4* __restrict__ dout)
{
    const int b = blockIdx.y;
    int n       = n_dev[b];
    n           = n < cap ? n : cap;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const size_t base = (size_t)b * cap;
    const int src     = order[base + p];
    kout[base + p]           = kin[base + src];
    dout[(base + p) * 2]     = din[(base + src) * 2];
    dout[(base + p) * 2 + 1] = din[(base + src) * 2 + 1];
}

// ---- local-mapping matchers (keyframe rate): every point / feature is independent -------------------

__device__ __forceinline__ u32 wave_min_u32(u32 v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
        const u32 o = __shfl_xor(v, off);
        v           = o < v ? o : v;
    }
    return v;
}

// MappingORBMatcher::Fuse, LocalMap<FusionPoint> overload (reference
// Snake/LocalMapping/MappingORBMatcher.cpp:359-480).  One wavefront per point; best[i] = feature or -1.
__global__ __launch_bounds__(256) void fuse_kernel(FrameDev F, CamDev C, ScalesDev S, const snk_fusion_point* __restrict__ pts,
                                                   const u8* __restrict__ mask, int m, float th, float obs_factor,
                                                   int feature_th, int* __restrict__ best)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= m) return;
    int result = -1;
    const snk_fusion_point lmp = pts[i];
    bool go = !(mask && !mask[i]);
    double ipx = 0, ipy = 0, z = 1, dist = 1;
    if (go)
    {
        const double pcx = C.R[0] * lmp.pos[0] + C.R[1] * lmp.pos[1] + C.R[2] * lmp.pos[2] + C.t[0];
        const double pcy = C.R[3] * lmp.pos[0] + C.R[4] * lmp.pos[1] + C.R[5] * lmp.pos[2] + C.t[1];
        z                = C.R[6] * lmp.pos[0] + C.R[7] * lmp.pos[1] + C.R[8] * lmp.pos[2] + C.t[2];
        go               = !(z <= 0);
        if (go)
        {
            ipx = C.fx * pcx / z + C.cx;
            ipy = C.fy * pcy / z + C.cy;
            go  = ipx >= F.min_x && ipx < F.max_x && ipy >= F.min_y && ipy < F.max_y;
        }
    }
    if (go)
    {
        const double POx = C.campos[0] - lmp.pos[0], POy = C.campos[1] - lmp.pos[1], POz = C.campos[2] - lmp.pos[2];
        dist             = sqrt(POx * POx + POy * POy + POz * POz);
        int rl           = lmp.reference_scale_level;
        rl               = rl < 0 ? 0 : (rl >= S.n ? S.n - 1 : rl);
        const double sref     = (double)S.s[rl];
        const double max_dist = 1.2 * (double)lmp.reference_depth * sref;
        const double min_dist = 0.8 * (double)lmp.reference_depth * sref / S.s_last;
        go = !(dist < min_dist || dist > max_dist) &&
             !(POx * lmp.normal[0] + POy * lmp.normal[1] + POz * lmp.normal[2] < 0.5 * dist);
    }
    if (go)
    {
        const float of     = lmp.observations <= 2 ? obs_factor : 1.0f;
        const float radius = of * th;
        const float gate   = (th * th) * of;
        double prediction  = (double)lmp.reference_scale_level + det_log((double)lmp.reference_depth / dist) / S.log_f;
        if (prediction < 0.0) prediction = 0.0;
        if (prediction > (double)(S.n - 1)) prediction = (double)(S.n - 1);
        const double r = (double)radius, r2 = r * r, ur = ipx - C.bf / z;
        uint4 qa, qc;
        split_desc(lmp.desc, qa, qc);
        const int cx0 = cell_coord(ipx - r, F.min_x, F.cols), cx1 = cell_coord(ipx + r, F.min_x, F.cols);
        const int cy0 = cell_coord(ipy - r, F.min_y, F.rows), cy1 = cell_coord(ipy + r, F.min_y, F.rows);
        u32 k1 = PJ_INF_KEY;
        for (int cx = cx0; cx <= cx1; ++cx)
        {
            const int lo = F.cell_start[cx * F.rows + cy0], hi = F.cell_start[cx * F.rows + cy1 + 1];
            for (int pid = lo + lane; pid < hi; pid += 64)
            {
                const snk_kp64 kp = F.kps[pid];
                if (fabs(prediction - (double)kp.octave) > 1.0) continue;
                const double ax = kp.x - ipx, ay = kp.y - ipy;  // area query: (kp.point - position).squaredNorm() < r2
                if (!(ax * ax + ay * ay < r2)) continue;
                const double dx = ipx - kp.x, dy = ipy - kp.y;
                double e2       = dx * dx + dy * dy;
                const float rp  = F.right_points[pid];
                if (rp > 0)
                {
                    const double dr = ur - (double)rp;
                    e2 += dr * dr;
                }
                if (e2 > (double)gate) continue;
                const uint4 ta = F.desc[(size_t)pid * 2], tc = F.desc[(size_t)pid * 2 + 1];
                const u32 d    = (u32)hamming256(qa, qc, ta, tc);
                const u32 key  = (d << PJ_IDX_BITS) | (u32)pid;
                k1             = key < k1 ? key : k1;
            }
        }
        k1 = wave_min_u32(k1);
        if (k1 != PJ_INF_KEY && (int)(k1 >> PJ_IDX_BITS) <= feature_th && (k1 >> PJ_IDX_BITS) < 256u) result = (int)(k1 & PJ_IDX_MASK);
    }
    if (lane == 0) best[i] = result;
}

struct TriDev
{
    double R1[9], t1[3], R2[9], t2[3];
    double fx, fy, cx, cy;
    double E[9];
    double th_chi2;
    int grid_rows, grid_cols, feature_distance;
};

// MappingORBMatcher::SearchForTriangulationProject (reference
// Snake/LocalMapping/MappingORBMatcher.cpp:168-249).  One wavefront per feature of keyframe 1.
__global__ __launch_bounds__(256) void triangulate_kernel(FrameDev F, TriDev T, const double* __restrict__ grid,
                                                          const snk_kp64* __restrict__ kps1, const double* __restrict__ np1,
                                                          const uint4* __restrict__ desc1, const u8* __restrict__ has1, int n1,
                                                          const double* __restrict__ np2, int* __restrict__ out)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= n1) return;
    int result = -1;
    if (!has1[i])
    {
        const snk_kp64 kp = kps1[i];
        const int cx = cell_coord(kp.x, F.min_x, F.cols), cy = cell_coord(kp.y, F.min_y, F.rows);
        const int gr = cy / 4 < T.grid_rows ? cy / 4 : T.grid_rows - 1, gc = cx / 4 < T.grid_cols ? cx / 4 : T.grid_cols - 1;
        const double z   = grid[gr * T.grid_cols + gc];
        const double pc0 = (kp.x - T.cx) / T.fx * z, pc1 = (kp.y - T.cy) / T.fy * z;
        const double d0 = pc0 - T.t1[0], d1 = pc1 - T.t1[1], d2 = z - T.t1[2];
        const double w0 = T.R1[0] * d0 + T.R1[3] * d1 + T.R1[6] * d2;
        const double w1 = T.R1[1] * d0 + T.R1[4] * d1 + T.R1[7] * d2;
        const double w2 = T.R1[2] * d0 + T.R1[5] * d1 + T.R1[8] * d2;
        const double p0 = T.R2[0] * w0 + T.R2[1] * w1 + T.R2[2] * w2 + T.t2[0];
        const double p1 = T.R2[3] * w0 + T.R2[4] * w1 + T.R2[5] * w2 + T.t2[1];
        const double p2 = T.R2[6] * w0 + T.R2[7] * w1 + T.R2[8] * w2 + T.t2[2];
        const double ipx = T.fx * p0 / p2 + T.cx, ipy = T.fy * p1 / p2 + T.cy;
        if (ipx >= F.min_x && ipx < F.max_x && ipy >= F.min_y && ipy < F.max_y)
        {
            const double x = np1[2 * i], y = np1[2 * i + 1];
            const double l0 = T.E[0] * x + T.E[1] * y + T.E[2], l1 = T.E[3] * x + T.E[4] * y + T.E[5],
                         l2 = T.E[6] * x + T.E[7] * y + T.E[8];
            const double ln = l0 * l0 + l1 * l1;
            const uint4 qa = desc1[(size_t)i * 2], qc = desc1[(size_t)i * 2 + 1];
            const double r = 20.0, r2 = 400.0;
            const int cx0 = cell_coord(ipx - r, F.min_x, F.cols), cx1 = cell_coord(ipx + r, F.min_x, F.cols);
            const int cy0 = cell_coord(ipy - r, F.min_y, F.rows), cy1 = cell_coord(ipy + r, F.min_y, F.rows);
            // "dist > bestDist -> continue" keeps the LAST candidate of minimal distance: key = dist | reversed index
            u32 k1 = PJ_INF_KEY;
            for (int c = cx0; c <= cx1; ++c)
            {
                const int lo = F.cell_start[c * F.rows + cy0], hi = F.cell_start[c * F.rows + cy1 + 1];
                for (int pid = lo + lane; pid < hi; pid += 64)
                {
                    const snk_kp64 k2 = F.kps[pid];
                    const double ax = k2.x - ipx, ay = k2.y - ipy;
                    if (!(ax * ax + ay * ay < r2)) continue;
                    if (F.taken[pid]) continue;
                    const double dd     = np2[2 * pid] * l0 + np2[2 * pid + 1] * l1 + l2;
                    const double disepi = dd * dd / ln;
                    if (disepi > T.th_chi2) continue;
                    const uint4 ta = F.desc[(size_t)pid * 2], tc = F.desc[(size_t)pid * 2 + 1];
                    const int d    = hamming256(qa, qc, ta, tc);
                    if (d > T.feature_distance || d > 50) continue;  // TH_LOW
                    const u32 key = ((u32)d << PJ_IDX_BITS) | (PJ_IDX_MASK - (u32)pid);
                    k1            = key < k1 ? key : k1;
                }
            }
            k1 = wave_min_u32(k1);
            if (k1 != PJ_INF_KEY) result = (int)(PJ_IDX_MASK - (k1 & PJ_IDX_MASK));
        }
    }
    if (lane == 0) out[i] = result;
}

struct EpiDev
{
    double E[9];
    double th_chi2;
    int feature_distance;
};

// Epipolar line of a keyframe-1 feature in keyframe 2 and the squared distance of a keyframe-2 feature to it, operation
// for operation as the oracle (and triangulate_kernel) evaluate them.
__device__ __forceinline__ void epi_line(const EpiDev& T, double x, double y, double& l0, double& l1, double& l2, double& ln)
{
    l0 = T.E[0] * x + T.E[1] * y + T.E[2];
    l1 = T.E[3] * x + T.E[4] * y + T.E[5];
    l2 = T.E[6] * x + T.E[7] * y + T.E[8];
    ln = l0 * l0 + l1 * l1;
}

// MappingORBMatcher::SearchForTriangulation2 (reference Snake/LocalMapping/MappingORBMatcher.cpp:14-99).  The host
// intersects the two bag-of-words feature vectors; an item = (feature of keyframe 1, [lo, hi) of the matching node's
// list in feat2).  Node lists are short (a handful of features), so 16 lanes share one item.
__global__ __launch_bounds__(256) void triangulate_bow_kernel(EpiDev T, const int4* __restrict__ items, int n_items,
                                                              const double* __restrict__ np1, const uint4* __restrict__ desc1,
                                                              const double* __restrict__ np2, const uint4* __restrict__ desc2,
                                                              const u8* __restrict__ has2, const int* __restrict__ feat2,
                                                              int* __restrict__ out)
{
    const int item = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const int4 it  = items[item < n_items ? item : n_items - 1];
    const int idx1 = it.x, lo = it.y, hi = it.z;
    double l0, l1, l2, ln;
    epi_line(T, np1[2 * idx1], np1[2 * idx1 + 1], l0, l1, l2, ln);
    const uint4 qa = desc1[(size_t)idx1 * 2], qc = desc1[(size_t)idx1 * 2 + 1];
    // "dist > bestDist -> continue" keeps the LAST candidate of minimal distance: key = dist | reversed list position
    u32 k1 = PJ_INF_KEY;
    for (int v = lo + sub; v < hi; v += 16)
    {
        const int idx2 = feat2[v];
        if (has2[idx2]) continue;
        const uint4 ta = desc2[(size_t)idx2 * 2], tc = desc2[(size_t)idx2 * 2 + 1];
        const int d    = hamming256(qa, qc, ta, tc);
        if (d > T.feature_distance || d > 50) continue;  // TH_LOW
        const double dd = np2[2 * idx2] * l0 + np2[2 * idx2 + 1] * l1 + l2;
        if (!(dd * dd / ln < T.th_chi2)) continue;
        const u32 key = ((u32)d << PJ_IDX_BITS) | (PJ_IDX_MASK - (u32)(v - lo));
        k1            = key < k1 ? key : k1;
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1)
    {
        const u32 o = __shfl_xor(k1, off);
        k1          = o < k1 ? o : k1;
    }
    if (sub == 0 && item < n_items) out[item] = k1 != PJ_INF_KEY ? feat2[lo + (int)(PJ_IDX_MASK - (k1 & PJ_IDX_MASK))] : -1;
}

// MappingORBMatcher::SearchForTriangulationBF (reference Snake/LocalMapping/MappingORBMatcher.cpp:102-165).  One
// wavefront per feature of keyframe 1 against every feature of keyframe 2: epipolar gate first, then Hamming.
__global__ __launch_bounds__(256) void triangulate_bf_kernel(EpiDev T, const double* __restrict__ np1, const uint4* __restrict__ desc1,
                                                             const u8* __restrict__ has1, int n1, const double* __restrict__ np2,
                                                             const uint4* __restrict__ desc2, const u8* __restrict__ has2, int n2,
                                                             int* __restrict__ out)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= n1) return;
    int result = -1;
    if (!has1[i])
    {
        double l0, l1, l2, ln;
        epi_line(T, np1[2 * i], np1[2 * i + 1], l0, l1, l2, ln);
        const uint4 qa = desc1[(size_t)i * 2], qc = desc1[(size_t)i * 2 + 1];
        u32 k1 = PJ_INF_KEY;
        for (int j = lane; j < n2; j += 64)
        {
            if (has2[j]) continue;
            const double dd = np2[2 * j] * l0 + np2[2 * j + 1] * l1 + l2;
            if (dd * dd / ln > T.th_chi2) continue;
            const uint4 ta = desc2[(size_t)j * 2], tc = desc2[(size_t)j * 2 + 1];
            const int d    = hamming256(qa, qc, ta, tc);
            if (d > T.feature_distance || d > 50) continue;  // TH_LOW
            const u32 key = ((u32)d << PJ_IDX_BITS) | (PJ_IDX_MASK - (u32)j);
            k1            = key < k1 ? key : k1;
        }
        k1 = wave_min_u32(k1);
        if (k1 != PJ_INF_KEY) result = (int)(PJ_IDX_MASK - (k1 & PJ_IDX_MASK));
    }
    if (lane == 0) out[i] = result;
}

// DeferredMapper::Relink, per-observation search (reference Snake/Optimizer/DeferredMapper.cpp:61-138).  The query disc
// (0.8 px) touches 1-4 grid cells with a handful of features: 16 lanes per query.
__global__ __launch_bounds__(256) void relink_kernel(FrameDev F, CamDev C, const snk_relink_query* __restrict__ qs, int n,
                                                     float radius, double out2, int feature_threshold, int* __restrict__ action,
                                                     int* __restrict__ best)
{
    const int q = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const snk_relink_query& Q = qs[q < n ? q : n - 1];
    const int i      = Q.feature;
    const double pcx = C.R[0] * Q.pos[0] + C.R[1] * Q.pos[1] + C.R[2] * Q.pos[2] + C.t[0];
    const double pcy = C.R[3] * Q.pos[0] + C.R[4] * Q.pos[1] + C.R[5] * Q.pos[2] + C.t[1];
    const double z   = C.R[6] * Q.pos[0] + C.R[7] * Q.pos[1] + C.R[8] * Q.pos[2] + C.t[2];
    const double ipx = C.fx * pcx / z + C.cx, ipy = C.fy * pcy / z + C.cy;
    const snk_kp64 ki = F.kps[i];
    const double ex = ipx - ki.x, ey = ipy - ki.y;
    const double rep2 = ex * ex + ey * ey;
    int act = 0, res = -1;
    if (z <= 0 || rep2 > out2)
        act = 1;  // also covers a NaN / infinite projection of a point on the camera plane only through z <= 0
    u32 k1 = PJ_INF_KEY;
    if (act == 0)
    {
        uint4 qa, qc;
        split_desc(Q.desc, qa, qc);
        int fd = hamming256(qa, qc, F.desc[(size_t)i * 2], F.desc[(size_t)i * 2 + 1]);
        if (fd == 0 && Q.has_alt)
        {
            uint4 aa, ac;
            split_desc(Q.alt_desc, aa, ac);
            fd = hamming256(qa, qc, aa, ac);
        }
        const double r = (double)radius, r2 = (double)(radius * radius), ur = ipx - C.bf / z;
        const int cx0 = cell_coord(ipx - r, F.min_x, F.cols), cx1 = cell_coord(ipx + r, F.min_x, F.cols);
        const int cy0 = cell_coord(ipy - r, F.min_y, F.rows), cy1 = cell_coord(ipy + r, F.min_y, F.rows);
        // "featureDist2 < bestDist" keeps the FIRST candidate of minimal distance: key = dist | index (grid order ==
        // iteration order)
        for (int cx = cx0; cx <= cx1; ++cx)
        {
            const int lo = F.cell_start[cx * F.rows + cy0], hi = F.cell_start[cx * F.rows + cy1 + 1];
            for (int pid = lo + sub; pid < hi; pid += 16)
            {
                const snk_kp64 kp = F.kps[pid];
                const double ax = kp.x - ipx, ay = kp.y - ipy;
                if (!(ax * ax + ay * ay < r2) || pid == i) continue;
                const double dx = ipx - kp.x, dy = ipy - kp.y;
                if (dx * dx + dy * dy > rep2) continue;
                const float rp = F.right_points[pid];
                if (rp > 0)
                {
                    const double er = ur - (double)rp;
                    if (er * er > rep2 * 2.0) continue;
                }
                const int d2 = hamming256(qa, qc, F.desc[(size_t)pid * 2], F.desc[(size_t)pid * 2 + 1]);
                if (!(d2 < feature_threshold && d2 < fd)) continue;
                const u32 key = ((u32)d2 << PJ_IDX_BITS) | (u32)pid;
                k1            = key < k1 ? key : k1;
            }
        }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1)
    {
        const u32 o = __shfl_xor(k1, off);
        k1          = o < k1 ? o : k1;
    }
    if (k1 != PJ_INF_KEY)
    {
        act = 2;
        res = (int)(k1 & PJ_IDX_MASK);
    }
    if (sub == 0 && q < n)
    {
        action[q] = act;
        best[q]   = res;
    }
}

// Points per wavefront of the projection matchers: a wavefront does the per-point geometry for `ppw` points at once and, with
// PJ_GROUP = 1, every lane then scans its own point's window.  Many points per wavefront fill the lanes (throughput of the
// batched forms), few spread a single frame's points over the chip: for the host calls (one frame, the reference's call
// pattern) ONE point per wavefront is the lowest latency -- measured on MI355X with the frame bound (tools/latency_track.py,
// profiles/r03/r03b_latency_track.log): SearchByProjectionFrameFrame2 with 1500 points 0.089 / 0.097 / 0.104 / 0.113 ms and
// SearchByProjection2 with 10 000 points 0.785 / 0.797 / 0.800 / 0.907 ms at 1 / 4 / 16 / 64 points per wavefront -- a lane's
// serial window scan is the critical path either way, and more wavefronts only add parallel scans.
constexpr int FRAME_LDS_MAX = 144 * 1024;  // LDS carve of the frame-resident matchers (one workgroup per CU above 80 KB)
size_t frame_lds_host(int cap, int ncell1)
{
    return (((size_t)cap * 24 + 15) & ~(size_t)15) + (size_t)cap * 32 + (((size_t)cap * 4 + 15) & ~(size_t)15) + (((size_t)cap + 15) & ~(size_t)15) +
           (size_t)ncell1 * 4 + 64;
}
bool no_frame_lds()
{
    static const bool v = getenv("SNK_TRACK_NO_FRAME_LDS") != nullptr;  // A/B: the batched matchers with global-memory gathers
    return v;
}

// Workgroups per frame of the frame-resident matchers: every workgroup stages the frame in LDS once and then walks its share of the
// local map's 1024-point chunks, so as few as keep the chip's 512 workgroup slots (two 65 KB frames per CU) filled about twice.
// Measured on the tracking leg of bench.py (1024 frames x 10 000 points, profiles/r04/r04g_track_wgs.log): 10 workgroups per frame
// (one per chunk, the round-3 form) 537 k frames/s, 5: 582 k, 3: 595 k, 2: 610 k, 1: 611 k.  SNK_TRACK_FRAME_WGS forces a value.
int frame_wgs(int chunks, int batch)
{
    static const int forced = getenv("SNK_TRACK_FRAME_WGS") ? atoi(getenv("SNK_TRACK_FRAME_WGS")) : 0;  // A/B, tests
    int wgs = batch > 0 ? (1024 + batch - 1) / batch : chunks;
    if (forced >= 1) wgs = forced;
    return wgs < 1 ? 1 : (wgs > chunks ? chunks : wgs);
}

// Byte offset of the claims inside the LDS carve of a frame-resident matcher that resolves its matches itself, -1 = separate resolution
// (several workgroups per frame, or the claims do not fit behind the frame).  SNK_TRACK_NO_FUSED_RESOLVE=1: always separate (A/B, tests).
int fused_claim_offset(int wgs, size_t flds, int cap)
{
    static const bool off = getenv("SNK_TRACK_NO_FUSED_RESOLVE") != nullptr;
    const size_t at = (flds + 15) & ~(size_t)15;
    return (off || wgs != 1 || at + (size_t)cap * 4 > (size_t)FRAME_LDS_MAX) ? -1 : (int)at;
}

int points_per_wave(long long total_points)
{
    static const int forced = getenv("SNK_TRACK_PPW") ? atoi(getenv("SNK_TRACK_PPW")) : 0;  // tests: force a value
    if (forced >= 1 && forced <= 64) return forced;
    const long long want = total_points / 8192;
    return (int)(want < 1 ? 1 : (want > 64 ? 64 : want));
}

int make_cam(const snk_camera* cam, const double* pose, CamDev* c)
{
    SNK_REQUIRE(cam != nullptr && pose != nullptr, "camera / pose is NULL");
    c->fx = cam->fx; c->fy = cam->fy; c->cx = cam->cx; c->cy = cam->cy; c->bf = cam->bf;
    const double x = pose[0], y = pose[1], z = pose[2], w = pose[3];
    double* R = c->R;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
    c->t[0] = pose[4]; c->t[1] = pose[5]; c->t[2] = pose[6];
    c->campos[0] = -(R[0] * c->t[0] + R[3] * c->t[1] + R[6] * c->t[2]);
    c->campos[1] = -(R[1] * c->t[0] + R[4] * c->t[1] + R[7] * c->t[2]);
    c->campos[2] = -(R[2] * c->t[0] + R[5] * c->t[1] + R[8] * c->t[2]);
    return SNK_OK;
}

// series evaluated on the host exactly like det_log (same operation order, no FMA)
double host_det_log(double x)
{
    int e;
    double m = frexp(x, &e);
    if (m < 0.70710678118654752440)
    {
        m *= 2.0;
        e -= 1;
    }
    const double z = (m - 1.0) / (m + 1.0), z2 = z * z;
    double s = 1.0 / 21.0;
    for (int k = 19; k >= 1; k -= 2) s = s * z2 + 1.0 / (double)k;
    return 2.0 * z * s + (double)e * 0.69314718055994530942;
}

int make_scales(const float* level_scale, int n_levels, ScalesDev* s)
{
    SNK_REQUIRE(level_scale != nullptr && n_levels >= 1 && n_levels <= 32, "level_scale / n_levels (1..32)");
    for (int i = 0; i < 32; ++i) s->s[i] = i < n_levels ? level_scale[i] : level_scale[n_levels - 1];
    s->n      = n_levels;
    s->log_f  = host_det_log(n_levels > 1 ? (double)level_scale[1] / (double)level_scale[0] : 1.2);
    s->s_last = (double)level_scale[n_levels - 1];
    return SNK_OK;
}

void grid_dims(const snk_grid_bounds* b, int* cols, int* rows)
{
    int c = (int)ceil((b->max_x - b->min_x) / 20.0), r = (int)ceil((b->max_y - b->min_y) / 20.0);
    *cols = c < 1 ? 1 : c;
    *rows = r < 1 ? 1 : r;
}

// byte offsets of the arrays of a frame of n features inside one device buffer
struct FrameLayout
{
    size_t o_desc, o_rp, o_tk, o_cs, total;
};
FrameLayout frame_layout(size_t n, size_t nc)
{
    FrameLayout L;
    L.o_desc = (n * sizeof(snk_kp64) + 15) & ~(size_t)15;
    L.o_rp   = L.o_desc + n * 32;
    L.o_tk   = L.o_rp + n * 4;
    L.o_cs   = (L.o_tk + n + 15) & ~(size_t)15;
    L.total  = L.o_cs + nc * 4;
    return L;
}
void frame_pointers(char* d, const FrameLayout& L, FrameDev* F)
{
    F->kps          = reinterpret_cast<const snk_kp64*>(d);
    F->desc         = reinterpret_cast<const uint4*>(d + L.o_desc);
    F->right_points = reinterpret_cast<const float*>(d + L.o_rp);
    F->taken        = reinterpret_cast<const u8*>(d + L.o_tk);
    F->cell_start   = reinterpret_cast<const int*>(d + L.o_cs);
}

int upload_frame_to(snk_matcher* m, DevBuf& buf, const snk_frame_view* f, FrameDev* F)
{
    SNK_REQUIRE(f->n >= 0 && f->n < (int)PJ_IDX_MASK && f->cols >= 1 && f->rows >= 1, "bad frame view sizes");
    SNK_REQUIRE((long long)f->cols * f->rows < 65535, "grid too large (window cell ranges are packed in 16 bits)");
    SNK_REQUIRE(f->n == 0 || (f->kps && f->desc && f->right_points && f->taken), "NULL frame arrays");
    SNK_REQUIRE(f->cell_start != nullptr, "cell_start is NULL");
    const size_t n = (size_t)f->n, nc = (size_t)f->cols * f->rows + 1;
    const FrameLayout L = frame_layout(n, nc);
    int rc = buf.reserve(L.total + 16);
    if (rc != SNK_OK) return rc;
    char* d = buf.as<char>();
    if (n)
    {
        SNK_HIP_CHECK(hipMemcpyAsync(d, f->kps, n * sizeof(snk_kp64), hipMemcpyHostToDevice, m->stream));
        SNK_HIP_CHECK(hipMemcpyAsync(d + L.o_desc, f->desc, n * 32, hipMemcpyHostToDevice, m->stream));
        SNK_HIP_CHECK(hipMemcpyAsync(d + L.o_rp, f->right_points, n * 4, hipMemcpyHostToDevice, m->stream));
        SNK_HIP_CHECK(hipMemcpyAsync(d + L.o_tk, f->taken, n, hipMemcpyHostToDevice, m->stream));
    }
    SNK_HIP_CHECK(hipMemcpyAsync(d + L.o_cs, f->cell_start, nc * 4, hipMemcpyHostToDevice, m->stream));
    F->n = f->n; F->cols = f->cols; F->rows = f->rows;
    frame_pointers(d, L, F);
    F->min_x = f->bounds.min_x; F->min_y = f->bounds.min_y; F->max_x = f->bounds.max_x; F->max_y = f->bounds.max_y;
    return SNK_OK;
}

// The frame a matcher call looks at: a host view is uploaded into the handle's scratch (every call); frame == NULL means the
// frame bound with snk_match_bind_frame, which is already on the device.
int upload_frame(snk_matcher* m, const snk_frame_view* f, FrameDev* F)
{
    if (f == nullptr)
    {
        SNK_REQUIRE(m->view_valid, "frame view is NULL and no frame is bound (snk_match_bind_frame)");
        F->n = m->view_n; F->cols = m->view_cols; F->rows = m->view_rows;
        frame_pointers(m->view.as<char>(), frame_layout((size_t)m->view_n, (size_t)m->view_cols * m->view_rows + 1), F);
        F->min_x = m->view_bounds[0]; F->min_y = m->view_bounds[1]; F->max_x = m->view_bounds[2]; F->max_y = m->view_bounds[3];
        return SNK_OK;
    }
    return upload_frame_to(m, m->aux, f, F);
}
}  // namespace
}  // namespace snk

using namespace snk;

extern "C" {

int snk_feature_grid(snk_matcher* m, const snk_kp64* undistorted, int n, const snk_grid_bounds* bounds, int32_t* perm,
                     int32_t* cell_start, int* cols_out, int* rows_out)
{
    SNK_REQUIRE(m != nullptr && bounds != nullptr && cell_start != nullptr, "NULL argument");
    SNK_REQUIRE(n >= 0 && n <= 16384, "feature count must be 0..16384");
    SNK_REQUIRE(n == 0 || (undistorted && perm), "NULL buffer");
    int cols, rows;
    grid_dims(bounds, &cols, &rows);
    SNK_REQUIRE((long long)cols * rows < 65535, "grid too large");
    if (cols_out) *cols_out = cols;
    if (rows_out) *rows_out = rows;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    const size_t nc = (size_t)cols * rows + 1;
    int rc;
    if ((rc = m->aux.reserve((size_t)(n > 0 ? n : 1) * sizeof(snk_kp64))) != SNK_OK) return rc;
    if ((rc = m->aux2.reserve((size_t)(n > 0 ? n : 1) * 4 + nc * 4)) != SNK_OK) return rc;
    int n_pow2 = 1;
    while (n_pow2 < n) n_pow2 <<= 1;
    if (n_pow2 < 2) n_pow2 = 2;
    int* d_perm = m->aux2.as<int>();
    int* d_cs   = d_perm + (n > 0 ? n : 1);
    if ((rc = m->h_in.reserve((size_t)(n > 0 ? n : 1) * sizeof(snk_kp64))) != SNK_OK) return rc;
    if ((rc = m->h_res.reserve((size_t)(n > 0 ? n : 1) * 4 + nc * 4)) != SNK_OK) return rc;
    if (n)
    {
        memcpy(m->h_in.p, undistorted, (size_t)n * sizeof(snk_kp64));  // pinned staging: see matcher_handle.hpp
        SNK_HIP_CHECK(hipMemcpyAsync(m->aux.p, m->h_in.p, (size_t)n * sizeof(snk_kp64), hipMemcpyHostToDevice, m->stream));
    }
    if (grid_count_fits(n, cols * rows) && !grid_use_network())
    {
        if ((rc = set_max_lds_once(reinterpret_cast<const void*>(grid_count_kernel), GRID_COUNT_LDS_MAX)) != SNK_OK) return rc;
        hipLaunchKernelGGL(grid_count_kernel, dim3(1), dim3(256), ((size_t)2 * (cols * rows + 1) + (size_t)2 * n) * 4, m->stream,
                           m->aux.as<snk_kp64>(), (const int*)nullptr, n, n > 0 ? n : 1, bounds->min_x, bounds->min_y, cols, rows, d_perm,
                           (int*)nullptr, d_cs);
    }
    else
    {
        if ((rc = set_max_lds_once(reinterpret_cast<const void*>(grid_kernel), 16384 * 4)) != SNK_OK) return rc;
        hipLaunchKernelGGL(grid_kernel, dim3(1), dim3(256), (size_t)n_pow2 * 4, m->stream, m->aux.as<snk_kp64>(), (const int*)nullptr,
                           n, n > 0 ? n : 1, bounds->min_x, bounds->min_y, cols, rows, d_perm, (int*)nullptr, d_cs);
    }
    SNK_LAUNCH_CHECK();
    // permutation | cell starts are adjacent in aux2: one copy back
    const size_t np1 = (size_t)(n > 0 ? n : 1);
    SNK_HIP_CHECK(hipMemcpyAsync(m->h_res.p, d_perm, (np1 + nc) * 4, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    if (n) memcpy(perm, m->h_res.p, (size_t)n * 4);
    memcpy(cell_start, m->h_res.as<int>() + np1, nc * 4);
    return SNK_OK;
}

int snk_feature_grid_batch_dev(snk_matcher* m, const snk_grid_bounds* bounds, const snk_kp64* kps_dev,
                               const uint64_t* desc_dev, const int32_t* n_dev, int cap, int batch, snk_kp64* kps_out_dev,
                               uint64_t* desc_out_dev, int32_t* perm_dev, int32_t* cell_start_dev)
{
    SNK_REQUIRE(m != nullptr && bounds != nullptr, "NULL argument");
    SNK_REQUIRE(batch >= 0 && cap >= 1 && cap <= 16384, "cap must be 1..16384");
    SNK_REQUIRE(kps_dev && desc_dev && n_dev && kps_out_dev && desc_out_dev && perm_dev && cell_start_dev, "NULL device buffer");
    int cols, rows;
    grid_dims(bounds, &cols, &rows);
    SNK_REQUIRE((long long)cols * rows < 65535, "grid too large");
    if (batch == 0) return SNK_OK;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    int rc;
    if ((rc = m->aux2.reserve((size_t)batch * cap * 4)) != SNK_OK) return rc;
    int cap_pow2 = 2;
    while (cap_pow2 < cap) cap_pow2 <<= 1;
    if (grid_count_fits(cap, cols * rows) && !grid_use_network())
    {
        if ((rc = set_max_lds_once(reinterpret_cast<const void*>(grid_count_kernel), GRID_COUNT_LDS_MAX)) != SNK_OK) return rc;
        hipLaunchKernelGGL(grid_count_kernel, dim3(batch), dim3(256), ((size_t)2 * (cols * rows + 1) + (size_t)2 * cap) * 4, m->stream, kps_dev,
                           n_dev, 0, cap, bounds->min_x, bounds->min_y, cols, rows, perm_dev, m->aux2.as<int>(), cell_start_dev);
    }
    else
    {
        if ((rc = set_max_lds_once(reinterpret_cast<const void*>(grid_kernel), 16384 * 4)) != SNK_OK) return rc;
        hipLaunchKernelGGL(grid_kernel, dim3(batch), dim3(256), (size_t)cap_pow2 * 4, m->stream, kps_dev, n_dev, 0, cap, bounds->min_x,
                           bounds->min_y, cols, rows, perm_dev, m->aux2.as<int>(), cell_start_dev);
    }
    hipLaunchKernelGGL(reorder_kernel, dim3(ceil_div(cap, 256), batch), dim3(256), 0, m->stream, m->aux2.as<int>(), n_dev, cap,
                       kps_dev, reinterpret_cast<const uint4*>(desc_dev), kps_out_dev, reinterpret_cast<uint4*>(desc_out_dev));
    SNK_LAUNCH_CHECK();
    return SNK_OK;
}

int snk_match_bind_frame(snk_matcher* m, const snk_frame_view* frame)
{
    SNK_REQUIRE(m != nullptr, "matcher is NULL");
    m->view_valid = false;
    if (frame == nullptr) return SNK_OK;  // unbind
    SNK_HIP_CHECK(hipSetDevice(m->device));
    FrameDev F;
    int rc = upload_frame_to(m, m->view, frame, &F);
    if (rc != SNK_OK) return rc;
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));  // the caller's arrays may change after the call
    m->view_n = frame->n; m->view_cols = frame->cols; m->view_rows = frame->rows;
    m->view_bounds[0] = frame->bounds.min_x; m->view_bounds[1] = frame->bounds.min_y;
    m->view_bounds[2] = frame->bounds.max_x; m->view_bounds[3] = frame->bounds.max_y;
    m->view_valid = true;
    return SNK_OK;
}

int snk_match_bound_taken(snk_matcher* m, const uint8_t* taken)
{
    SNK_REQUIRE(m != nullptr && m->view_valid, "no frame is bound (snk_match_bind_frame)");
    SNK_REQUIRE(m->view_n == 0 || taken != nullptr, "taken is NULL");
    SNK_HIP_CHECK(hipSetDevice(m->device));
    const FrameLayout L = frame_layout((size_t)m->view_n, (size_t)m->view_cols * m->view_rows + 1);
    return copy_sync(m->view.as<char>() + L.o_tk, taken, (size_t)m->view_n, hipMemcpyHostToDevice, m->stream);
}

int snk_match_project_coarse(snk_matcher* m, const snk_frame_view* frame, const snk_camera* cam, const double pose[7],
                             const snk_lm_coarse* pts, int n_pts, float th, int feature_error, int direction,
                             const float* level_scale, int n_levels, int32_t* match_idx, int* n_matches)
{
    SNK_REQUIRE(m != nullptr && n_matches != nullptr, "NULL argument");
    *n_matches = 0;
    SNK_REQUIRE(n_pts >= 0 && (n_pts == 0 || (pts && match_idx)), "bad point arrays");
    SNK_REQUIRE(direction >= 0 && direction <= 2, "direction must be 0 (none), 1 (forward) or 2 (backward)");
    CamDev C;
    ScalesDev S;
    FrameDev F;
    int rc;
    if ((rc = make_cam(cam, pose, &C)) != SNK_OK) return rc;
    if ((rc = make_scales(level_scale, n_levels, &S)) != SNK_OK) return rc;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    if ((rc = upload_frame(m, frame, &F)) != SNK_OK) return rc;
    if (n_pts == 0) return SNK_OK;
    const size_t np = (size_t)n_pts;
    // q: points | out: best, bins, match | t: claim | cnt: count
    if ((rc = m->q.reserve(np * sizeof(snk_lm_coarse))) != SNK_OK) return rc;
    if ((rc = m->out.reserve(np * 12 + 64)) != SNK_OK) return rc;
    if ((rc = m->t.reserve((size_t)(F.n > 0 ? F.n : 1) * 4)) != SNK_OK) return rc;
    if ((rc = m->cnt.reserve(64)) != SNK_OK) return rc;
    if ((rc = m->h_in.reserve(np * sizeof(snk_lm_coarse))) != SNK_OK) return rc;
    if ((rc = m->h_res.reserve(np * 4 + 4)) != SNK_OK) return rc;
    int* d_best = m->out.as<int>();
    int* d_bins = d_best + np;
    int* d_match = d_bins + np;  // match[np] | count: one block, one copy back
    memcpy(m->h_in.p, pts, np * sizeof(snk_lm_coarse));
    SNK_HIP_CHECK(hipMemcpyAsync(m->q.p, m->h_in.p, np * sizeof(snk_lm_coarse), hipMemcpyHostToDevice, m->stream));
    const int ppw = points_per_wave(n_pts);
    hipLaunchKernelGGL(coarse_kernel, dim3(ceil_div(n_pts, 4 * ppw)), dim3(256), 0, m->stream, F, C, S, m->q.as<snk_lm_coarse>(), n_pts, ppw,
                       th, feature_error, direction, d_best, d_bins);
    hipLaunchKernelGGL(resolve_kernel, dim3(1), dim3(256), 0, m->stream, d_best, d_bins, n_pts, F.taken, m->t.as<int>(), F.n, 1,
                       d_match, d_match + np);
    SNK_LAUNCH_CHECK();
    SNK_HIP_CHECK(hipMemcpyAsync(m->h_res.p, d_match, np * 4 + 4, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    memcpy(match_idx, m->h_res.p, np * 4);
    memcpy(n_matches, m->h_res.as<char>() + np * 4, 4);
    return SNK_OK;
}

int snk_match_project_fine(snk_matcher* m, const snk_frame_view* frame, const snk_camera* cam, const double pose[7],
                           snk_lm_fine* pts, int n_pts, float th, float ratio, const float* level_scale, int n_levels,
                           int32_t* match_idx, uint8_t* visible, int* n_matches)
{
    SNK_REQUIRE(m != nullptr && n_matches != nullptr, "NULL argument");
    *n_matches = 0;
    SNK_REQUIRE(n_pts >= 0 && (n_pts == 0 || (pts && match_idx && visible)), "bad point arrays");
    CamDev C;
    ScalesDev S;
    FrameDev F;
    int rc;
    if ((rc = make_cam(cam, pose, &C)) != SNK_OK) return rc;
    if ((rc = make_scales(level_scale, n_levels, &S)) != SNK_OK) return rc;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    if ((rc = upload_frame(m, frame, &F)) != SNK_OK) return rc;
    if (n_pts == 0) return SNK_OK;
    const size_t np = (size_t)n_pts;
    if ((rc = m->q.reserve(np * sizeof(snk_lm_fine))) != SNK_OK) return rc;
    const size_t res_b = np * 4 + 4 + np;  // match[np] | count | visible[np]: one block, one copy back
    if ((rc = m->out.reserve(np * 4 + res_b + 64)) != SNK_OK) return rc;
    if ((rc = m->t.reserve((size_t)(F.n > 0 ? F.n : 1) * 4)) != SNK_OK) return rc;
    if ((rc = m->cnt.reserve(64)) != SNK_OK) return rc;
    if ((rc = m->h_in.reserve(np * sizeof(snk_lm_fine))) != SNK_OK) return rc;
    if ((rc = m->h_res.reserve(res_b)) != SNK_OK) return rc;
    int* d_best  = m->out.as<int>();
    int* d_match = d_best + np;
    int* d_cnt   = d_match + np;
    u8* d_vis    = reinterpret_cast<u8*>(d_cnt + 1);
    memcpy(m->h_in.p, pts, np * sizeof(snk_lm_fine));
    SNK_HIP_CHECK(hipMemcpyAsync(m->q.p, m->h_in.p, np * sizeof(snk_lm_fine), hipMemcpyHostToDevice, m->stream));
    const int ppw = points_per_wave(n_pts);
    hipLaunchKernelGGL(fine_kernel, dim3(ceil_div(n_pts, 4 * ppw)), dim3(256), 0, m->stream, F, C, S, m->q.as<snk_lm_fine>(), n_pts, ppw, th,
                       ratio, d_best, d_vis);
    hipLaunchKernelGGL(resolve_kernel, dim3(1), dim3(256), 0, m->stream, d_best, (const int*)nullptr, n_pts, F.taken,
                       m->t.as<int>(), F.n, 0, d_match, d_cnt);
    SNK_LAUNCH_CHECK();
    SNK_HIP_CHECK(hipMemcpyAsync(m->h_res.p, d_match, res_b, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    const char* hr = m->h_res.as<char>();
    memcpy(match_idx, hr, np * 4);
    memcpy(n_matches, hr + np * 4, 4);
    memcpy(visible, hr + np * 4 + 4, np);
    // lmp.valid after the search is `visible` (a point keeps valid = 1 exactly when it passes every cull, which is where
    // IncreaseVisible() is called, SnakeORBMatcher.cpp:397-431): the 96-byte records are not copied back for one byte each
    for (size_t i = 0; i < np; ++i) pts[i].valid = visible[i];
    return SNK_OK;
}

namespace
{
int make_frames(const snk_frames_dev* f, FramesDev* F)
{
    SNK_REQUIRE(f != nullptr, "frames is NULL");
    SNK_REQUIRE(f->batch >= 0 && f->cap >= 1 && f->cap < (int)PJ_IDX_MASK, "bad batch / cap");
    SNK_REQUIRE(f->n && f->kps && f->desc && f->right_points && f->taken && f->cell_start, "NULL device array in frames");
    grid_dims(&f->bounds, &F->cols, &F->rows);
    SNK_REQUIRE((long long)F->cols * F->rows < 65535, "grid too large");
    F->cap          = f->cap;
    F->n            = f->n;
    F->kps          = f->kps;
    F->desc         = reinterpret_cast<const uint4*>(f->desc);
    F->right_points = f->right_points;
    F->taken        = f->taken;
    F->cell_start   = f->cell_start;
    F->min_x = f->bounds.min_x; F->min_y = f->bounds.min_y; F->max_x = f->bounds.max_x; F->max_y = f->bounds.max_y;
    return SNK_OK;
}

// scratch of the batched matchers: aux = CamDev[batch], out = best | bins ([batch][m_cap] each), t = claim [batch][cap]
int batch_scratch(snk_matcher* m, int batch, int m_cap, int cap, const snk_camera* cam, const double* poses_dev, CamDev** cams,
                  int** best, int** bins, int** claim)
{
    int rc;
    if ((rc = m->aux.reserve((size_t)batch * sizeof(CamDev))) != SNK_OK) return rc;
    if ((rc = m->out.reserve((size_t)batch * m_cap * 8)) != SNK_OK) return rc;
    if ((rc = m->t.reserve((size_t)batch * cap * 4)) != SNK_OK) return rc;
    *cams  = m->aux.as<CamDev>();
    *best  = m->out.as<int>();
    *bins  = *best + (size_t)batch * m_cap;
    *claim = m->t.as<int>();
    hipLaunchKernelGGL(cam_batch_kernel, dim3(ceil_div(batch, 64)), dim3(64), 0, m->stream, *cam, poses_dev, batch, *cams);
    return SNK_OK;
}
}  // namespace

static int launch_resolve_batch(snk_matcher* m, int batch, int cap, const int* best, const int* bins, const int* n_pts_dev, int pts_cap,
                                const FramesDev& F, int* claim, int with_rotation, int* match_idx_dev, int* n_matches_dev)
{
    if (cap <= RESOLVE_LDS_FEATURES)
    {
        int rc = set_max_lds_once(reinterpret_cast<const void*>(resolve_batch_kernel<true>), RESOLVE_LDS_FEATURES * (int)sizeof(int) + 1024);
        if (rc != SNK_OK) return rc;
        hipLaunchKernelGGL(resolve_batch_kernel<true>, dim3(batch), dim3(1024), (size_t)cap * sizeof(int), m->stream, best, bins, n_pts_dev,
                           pts_cap, F, claim, with_rotation, match_idx_dev, n_matches_dev);
    }
    else
        hipLaunchKernelGGL(resolve_batch_kernel<false>, dim3(batch), dim3(1024), 0, m->stream, best, bins, n_pts_dev, pts_cap, F, claim,
                           with_rotation, match_idx_dev, n_matches_dev);
    SNK_LAUNCH_CHECK();
    return SNK_OK;
}

int snk_match_project_coarse_batch_dev(snk_matcher* m, const snk_frames_dev* frames, const snk_camera* cam,
                                       const double* poses_dev, const snk_lm_coarse* pts_dev, const int32_t* n_pts_dev,
                                       int pts_cap, float th, int feature_error, int direction, const float* level_scale,
                                       int n_levels, int32_t* match_idx_dev, int32_t* n_matches_dev)
{
    SNK_REQUIRE(m != nullptr && cam != nullptr, "NULL argument");
    SNK_REQUIRE(poses_dev && pts_dev && n_pts_dev && match_idx_dev && n_matches_dev, "NULL device buffer");
    SNK_REQUIRE(pts_cap >= 1, "pts_cap must be >= 1");
    SNK_REQUIRE(direction >= 0 && direction <= 2, "direction must be 0 (none), 1 (forward) or 2 (backward)");
    FramesDev F;
    ScalesDev S;
    int rc;
    if ((rc = make_frames(frames, &F)) != SNK_OK) return rc;
    if ((rc = make_scales(level_scale, n_levels, &S)) != SNK_OK) return rc;
    const int batch = frames->batch;
    if (batch == 0) return SNK_OK;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    CamDev* cams;
    int *best, *bins, *claim;
    if ((rc = batch_scratch(m, batch, pts_cap, F.cap, cam, poses_dev, &cams, &best, &bins, &claim)) != SNK_OK) return rc;
    const int ppw = points_per_wave((long long)pts_cap * batch);
    const size_t flds = frame_lds_host(F.cap, F.cols * F.rows + 1);
    if (ppw == 64 && flds <= FRAME_LDS_MAX && !no_frame_lds())
    {
        if ((rc = set_max_lds_once(reinterpret_cast<const void*>(coarse_frame_kernel), FRAME_LDS_MAX)) != SNK_OK) return rc;
        const int wgs       = frame_wgs(ceil_div(pts_cap, 1024), batch);
        const int claim_off = fused_claim_offset(wgs, flds, F.cap);
        hipLaunchKernelGGL(coarse_frame_kernel, dim3(wgs, batch), dim3(1024), claim_off >= 0 ? (size_t)claim_off + (size_t)F.cap * 4 : flds, m->stream, F,
                           (const CamDev*)cams, S, pts_dev, n_pts_dev, pts_cap, th, feature_error, direction, best, bins, claim_off, match_idx_dev,
                           n_matches_dev);
        SNK_LAUNCH_CHECK();
        if (claim_off >= 0) return SNK_OK;  // resolved in the kernel
    }
    else
        hipLaunchKernelGGL(coarse_batch_kernel, dim3(ceil_div(pts_cap, 4 * ppw), batch), dim3(256), 0, m->stream, F, (const CamDev*)cams, S,
                           pts_dev, n_pts_dev, pts_cap, ppw, th, feature_error, direction, best, bins);
    if ((rc = launch_resolve_batch(m, batch, frames->cap, (const int*)best, (const int*)bins, n_pts_dev, pts_cap, F, claim, 1, match_idx_dev,
                                   n_matches_dev)) != SNK_OK)
        return rc;
    return SNK_OK;
}

}  // extern "C"
static int fine_batch_impl(snk_matcher* m, const snk_frames_dev* frames, const snk_camera* cam, const double* poses_dev,
                           snk_lm_fine* pts_dev, const int32_t* n_pts_dev, int pts_cap, float th, float ratio,
                           const float* level_scale, int n_levels, int32_t* match_idx_dev, uint8_t* visible_dev,
                           int32_t* n_matches_dev, bool write_valid)
{
    SNK_REQUIRE(m != nullptr && cam != nullptr, "NULL argument");
    SNK_REQUIRE(poses_dev && pts_dev && n_pts_dev && match_idx_dev && visible_dev && n_matches_dev, "NULL device buffer");
    SNK_REQUIRE(pts_cap >= 1, "pts_cap must be >= 1");
    FramesDev F;
    ScalesDev S;
    int rc;
    if ((rc = make_frames(frames, &F)) != SNK_OK) return rc;
    if ((rc = make_scales(level_scale, n_levels, &S)) != SNK_OK) return rc;
    const int batch = frames->batch;
    if (batch == 0) return SNK_OK;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    CamDev* cams;
    int *best, *bins, *claim;
    if ((rc = batch_scratch(m, batch, pts_cap, F.cap, cam, poses_dev, &cams, &best, &bins, &claim)) != SNK_OK) return rc;
    SNK_HIP_CHECK(hipMemsetAsync(visible_dev, 0, (size_t)batch * pts_cap, m->stream));
    const int ppw = points_per_wave((long long)pts_cap * batch);
    const size_t flds = frame_lds_host(F.cap, F.cols * F.rows + 1);
    if (ppw == 64 && flds <= FRAME_LDS_MAX && !no_frame_lds())
    {
        const void* fk = write_valid ? reinterpret_cast<const void*>(fine_frame_kernel<true>) : reinterpret_cast<const void*>(fine_frame_kernel<false>);
        if ((rc = set_max_lds_once(fk, FRAME_LDS_MAX)) != SNK_OK) return rc;
        const int wgs = frame_wgs(ceil_div(pts_cap, 1024), batch);
        const int claim_off = fused_claim_offset(wgs, flds, F.cap);
        const size_t lds    = claim_off >= 0 ? (size_t)claim_off + (size_t)F.cap * 4 : flds;
        if (write_valid)
            hipLaunchKernelGGL(fine_frame_kernel<true>, dim3(wgs, batch), dim3(1024), lds, m->stream, F, (const CamDev*)cams, S, pts_dev,
                               n_pts_dev, pts_cap, th, ratio, best, visible_dev, claim_off, match_idx_dev, n_matches_dev);
        else
            hipLaunchKernelGGL(fine_frame_kernel<false>, dim3(wgs, batch), dim3(1024), lds, m->stream, F, (const CamDev*)cams, S, pts_dev,
                               n_pts_dev, pts_cap, th, ratio, best, visible_dev, claim_off, match_idx_dev, n_matches_dev);
        if (claim_off >= 0)
        {
            SNK_LAUNCH_CHECK();
            return SNK_OK;  // resolved in the kernel
        }
    }
    else if (write_valid)
        hipLaunchKernelGGL(fine_batch_kernel<true>, dim3(ceil_div(pts_cap, 4 * ppw), batch), dim3(256), 0, m->stream, F, (const CamDev*)cams,
                           S, pts_dev, n_pts_dev, pts_cap, ppw, th, ratio, best, visible_dev);
    else
        hipLaunchKernelGGL(fine_batch_kernel<false>, dim3(ceil_div(pts_cap, 4 * ppw), batch), dim3(256), 0, m->stream, F, (const CamDev*)cams,
                           S, pts_dev, n_pts_dev, pts_cap, ppw, th, ratio, best, visible_dev);
    if ((rc = launch_resolve_batch(m, batch, frames->cap, (const int*)best, (const int*)nullptr, n_pts_dev, pts_cap, F, claim, 0,
                                   match_idx_dev, n_matches_dev)) != SNK_OK)
        return rc;
    return SNK_OK;
}
extern "C"
{
int snk_match_project_fine_batch_dev(snk_matcher* m, const snk_frames_dev* frames, const snk_camera* cam, const double* poses_dev,
                                     snk_lm_fine* pts_dev, const int32_t* n_pts_dev, int pts_cap, float th, float ratio,
                                     const float* level_scale, int n_levels, int32_t* match_idx_dev, uint8_t* visible_dev,
                                     int32_t* n_matches_dev)
{
    return fine_batch_impl(m, frames, cam, poses_dev, pts_dev, n_pts_dev, pts_cap, th, ratio, level_scale, n_levels, match_idx_dev,
                           visible_dev, n_matches_dev, true);
}

int snk_match_project_fine_batch_ro_dev(snk_matcher* m, const snk_frames_dev* frames, const snk_camera* cam, const double* poses_dev,
                                        const snk_lm_fine* pts_dev, const int32_t* n_pts_dev, int pts_cap, float th, float ratio,
                                        const float* level_scale, int n_levels, int32_t* match_idx_dev, uint8_t* visible_dev,
                                        int32_t* n_matches_dev)
{
    return fine_batch_impl(m, frames, cam, poses_dev, const_cast<snk_lm_fine*>(pts_dev), n_pts_dev, pts_cap, th, ratio, level_scale,
                           n_levels, match_idx_dev, visible_dev, n_matches_dev, false);
}

int snk_match_mark_taken_batch_dev(snk_matcher* m, const int32_t* match_idx_dev, const int32_t* n_pts_dev, int pts_cap, int batch,
                                   uint8_t* taken_dev, int cap)
{
    SNK_REQUIRE(m != nullptr && match_idx_dev && n_pts_dev && taken_dev, "NULL argument");
    SNK_REQUIRE(batch >= 0 && pts_cap >= 1 && cap >= 1, "bad sizes");
    if (batch == 0) return SNK_OK;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    hipLaunchKernelGGL(mark_taken_kernel, dim3(ceil_div(pts_cap, 256), batch), dim3(256), 0, m->stream, match_idx_dev, n_pts_dev,
                       pts_cap, taken_dev, cap);
    SNK_LAUNCH_CHECK();
    return SNK_OK;
}

int snk_match_project_keyframe(snk_matcher* m, const snk_frame_view* frame, const snk_camera* cam, const double pose[7],
                               const double (*positions)[3], const uint64_t (*descriptors)[4], const uint8_t* skip, int n_pts,
                               float th, int feature_error, int32_t* match_idx, int* n_matches)
{
    SNK_REQUIRE(m != nullptr && n_matches != nullptr, "NULL argument");
    *n_matches = 0;
    SNK_REQUIRE(n_pts >= 0 && (n_pts == 0 || (positions && descriptors && skip && match_idx)), "bad point arrays");
    CamDev C;
    FrameDev F;
    int rc;
    if ((rc = make_cam(cam, pose, &C)) != SNK_OK) return rc;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    if ((rc = upload_frame(m, frame, &F)) != SNK_OK) return rc;
    SNK_REQUIRE(F.n <= KF_MAX_FEATURES, "too many features for the keyframe matcher");
    if (n_pts == 0) return SNK_OK;
    const size_t np = (size_t)n_pts;
    const size_t o_desc = np * 24, o_skip = o_desc + np * 32, total = o_skip + np;
    if ((rc = m->q.reserve(total)) != SNK_OK) return rc;
    if ((rc = m->out.reserve(np * 4)) != SNK_OK) return rc;
    if ((rc = m->cnt.reserve(64)) != SNK_OK) return rc;
    char* d = m->q.as<char>();
    SNK_HIP_CHECK(hipMemcpyAsync(d, positions, np * 24, hipMemcpyHostToDevice, m->stream));
    SNK_HIP_CHECK(hipMemcpyAsync(d + o_desc, descriptors, np * 32, hipMemcpyHostToDevice, m->stream));
    SNK_HIP_CHECK(hipMemcpyAsync(d + o_skip, skip, np, hipMemcpyHostToDevice, m->stream));
    if ((rc = set_max_lds_once(reinterpret_cast<const void*>(keyframe_kernel), KF_MAX_FEATURES + 16)) != SNK_OK) return rc;
    hipLaunchKernelGGL(keyframe_kernel, dim3(1), dim3(64), (size_t)((F.n + 15) & ~15) + 16, m->stream, F, C,
                       reinterpret_cast<const double*>(d), reinterpret_cast<const uint4*>(d + o_desc),
                       reinterpret_cast<const u8*>(d + o_skip), n_pts, th, feature_error, m->out.as<int>(), m->cnt.as<int>());
    SNK_LAUNCH_CHECK();
    SNK_HIP_CHECK(hipMemcpyAsync(match_idx, m->out.p, np * 4, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipMemcpyAsync(n_matches, m->cnt.p, 4, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    return SNK_OK;
}

int snk_match_fuse(snk_matcher* m, const snk_frame_view* frame, const snk_camera* cam, const double pose[7],
                   const snk_fusion_point* pts, const uint8_t* point_mask, int n_pts, float th, float obs_factor, int feature_th,
                   const float* level_scale, int n_levels, int32_t* best_idx, int* n_fused)
{
    SNK_REQUIRE(m != nullptr && n_fused != nullptr, "NULL argument");
    *n_fused = 0;
    SNK_REQUIRE(n_pts >= 0 && (n_pts == 0 || (pts && best_idx)), "bad point arrays");
    CamDev C;
    ScalesDev S;
    FrameDev F;
    int rc;
    if ((rc = make_cam(cam, pose, &C)) != SNK_OK) return rc;
    if ((rc = make_scales(level_scale, n_levels, &S)) != SNK_OK) return rc;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    if ((rc = upload_frame(m, frame, &F)) != SNK_OK) return rc;
    if (n_pts == 0) return SNK_OK;
    const size_t np = (size_t)n_pts, o_mask = np * sizeof(snk_fusion_point);
    if ((rc = m->q.reserve(o_mask + np + 16)) != SNK_OK) return rc;
    if ((rc = m->out.reserve(np * 4)) != SNK_OK) return rc;
    char* d = m->q.as<char>();
    SNK_HIP_CHECK(hipMemcpyAsync(d, pts, o_mask, hipMemcpyHostToDevice, m->stream));
    if (point_mask) SNK_HIP_CHECK(hipMemcpyAsync(d + o_mask, point_mask, np, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(fuse_kernel, dim3(ceil_div(n_pts, 4)), dim3(256), 0, m->stream, F, C, S,
                       reinterpret_cast<const snk_fusion_point*>(d), point_mask ? reinterpret_cast<const u8*>(d + o_mask) : nullptr,
                       n_pts, th, obs_factor, feature_th, m->out.as<int>());
    SNK_LAUNCH_CHECK();
    SNK_HIP_CHECK(hipMemcpyAsync(best_idx, m->out.p, np * 4, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    int cnt = 0;
    for (int i = 0; i < n_pts; ++i) cnt += best_idx[i] >= 0 ? 1 : 0;
    *n_fused = cnt;
    return SNK_OK;
}

int snk_match_triangulation_project(snk_matcher* m, const double* depth_grid, int grid_rows, int grid_cols, const double pose1[7],
                                    const double pose2[7], const snk_camera* cam, const snk_kp64* kps1, const double (*np1)[2],
                                    const uint64_t (*desc1)[4], const uint8_t* has_mp1, int n1, const snk_frame_view* frame2,
                                    const double (*np2)[2], const double E12[9], float epipolar_distance, int feature_distance,
                                    int32_t* match_idx2, int* n_matches)
{
    SNK_REQUIRE(m != nullptr && n_matches != nullptr, "NULL argument");
    *n_matches = 0;
    SNK_REQUIRE(depth_grid != nullptr && grid_rows >= 1 && grid_cols >= 1 && E12 != nullptr, "bad depth grid / E");
    SNK_REQUIRE(n1 >= 0 && (n1 == 0 || (kps1 && np1 && desc1 && has_mp1 && match_idx2)), "bad keyframe-1 arrays");
    // frame2 == NULL: the frame bound with snk_match_bind_frame (already on the device), like the other matchers
    SNK_REQUIRE(frame2 != nullptr || m->view_valid, "frame2 is NULL and no frame is bound (snk_match_bind_frame)");
    const int n2_frame = frame2 ? frame2->n : m->view_n;
    SNK_REQUIRE(n2_frame == 0 || np2 != nullptr, "bad keyframe-2 arrays");
    CamDev C1, C2;
    FrameDev F;
    int rc;
    if ((rc = make_cam(cam, pose1, &C1)) != SNK_OK) return rc;
    if ((rc = make_cam(cam, pose2, &C2)) != SNK_OK) return rc;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    if ((rc = upload_frame(m, frame2, &F)) != SNK_OK) return rc;
    if (n1 == 0) return SNK_OK;
    TriDev T;
    memcpy(T.R1, C1.R, sizeof(T.R1));
    memcpy(T.t1, C1.t, sizeof(T.t1));
    memcpy(T.R2, C2.R, sizeof(T.R2));
    memcpy(T.t2, C2.t, sizeof(T.t2));
    T.fx = cam->fx; T.fy = cam->fy; T.cx = cam->cx; T.cy = cam->cy;
    memcpy(T.E, E12, sizeof(T.E));
    const double th_chi1 = (double)epipolar_distance / cam->fx;
    T.th_chi2            = th_chi1 * th_chi1;
    T.grid_rows = grid_rows; T.grid_cols = grid_cols; T.feature_distance = feature_distance;
    const size_t n = (size_t)n1, n2 = (size_t)n2_frame, ng = (size_t)grid_rows * grid_cols;
    const size_t o_np1 = n * sizeof(snk_kp64), o_d1 = o_np1 + n * 16, o_h1 = o_d1 + n * 32, o_np2 = (o_h1 + n + 15) & ~(size_t)15,
                 o_grid = o_np2 + n2 * 16, total = o_grid + ng * 8;
    if ((rc = m->q.reserve(total + 16)) != SNK_OK) return rc;
    if ((rc = m->out.reserve(n * 4)) != SNK_OK) return rc;
    char* d = m->q.as<char>();
    SNK_HIP_CHECK(hipMemcpyAsync(d, kps1, n * sizeof(snk_kp64), hipMemcpyHostToDevice, m->stream));
    SNK_HIP_CHECK(hipMemcpyAsync(d + o_np1, np1, n * 16, hipMemcpyHostToDevice, m->stream));
    SNK_HIP_CHECK(hipMemcpyAsync(d + o_d1, desc1, n * 32, hipMemcpyHostToDevice, m->stream));
    SNK_HIP_CHECK(hipMemcpyAsync(d + o_h1, has_mp1, n, hipMemcpyHostToDevice, m->stream));
    if (n2) SNK_HIP_CHECK(hipMemcpyAsync(d + o_np2, np2, n2 * 16, hipMemcpyHostToDevice, m->stream));
    SNK_HIP_CHECK(hipMemcpyAsync(d + o_grid, depth_grid, ng * 8, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(triangulate_kernel, dim3(ceil_div(n1, 4)), dim3(256), 0, m->stream, F, T,
                       reinterpret_cast<const double*>(d + o_grid), reinterpret_cast<const snk_kp64*>(d),
                       reinterpret_cast<const double*>(d + o_np1), reinterpret_cast<const uint4*>(d + o_d1),
                       reinterpret_cast<const u8*>(d + o_h1), n1, reinterpret_cast<const double*>(d + o_np2), m->out.as<int>());
    SNK_LAUNCH_CHECK();
    SNK_HIP_CHECK(hipMemcpyAsync(match_idx2, m->out.p, n * 4, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    int cnt = 0;
    for (int i = 0; i < n1; ++i) cnt += match_idx2[i] >= 0 ? 1 : 0;
    *n_matches = cnt;
    return SNK_OK;
}

namespace
{
int check_bow(const snk_bow_features* b, int n)
{
    SNK_REQUIRE(b != nullptr && b->n_nodes >= 0, "bag-of-words feature vector is NULL");
    if (b->n_nodes == 0) return SNK_OK;
    SNK_REQUIRE(b->node_id && b->node_start && b->node_start[0] >= 0, "NULL bag-of-words arrays");
    for (int k = 0; k < b->n_nodes; ++k)
    {
        SNK_REQUIRE(b->node_start[k + 1] >= b->node_start[k], "node_start must be non-decreasing");
        SNK_REQUIRE(k == 0 || b->node_id[k] > b->node_id[k - 1], "node ids must be strictly ascending");
    }
    SNK_REQUIRE(b->node_start[b->n_nodes] == b->node_start[0] || b->features != nullptr, "NULL bag-of-words feature list");
    for (int v = b->node_start[0]; v < b->node_start[b->n_nodes]; ++v)
        SNK_REQUIRE(b->features[v] >= 0 && b->features[v] < n, "bag-of-words feature index out of range");
    return SNK_OK;
}

// np | desc | has of one keyframe, packed into `buf`
struct KfDev
{
    const double* np;
    const uint4* desc;
    const u8* has;
};
int upload_kf(snk_matcher* m, snk::DevBuf& buf, const double (*np)[2], const uint64_t (*desc)[4], const uint8_t* has, int n,
              size_t extra, KfDev* K, char** extra_ptr)
{
    const size_t nn = (size_t)n, o_d = nn * 16, o_h = o_d + nn * 32, o_x = (o_h + nn + 15) & ~(size_t)15;
    int rc = buf.reserve(o_x + extra + 16);
    if (rc != SNK_OK) return rc;
    char* d = buf.as<char>();
    if (n)
    {
        SNK_HIP_CHECK(hipMemcpyAsync(d, np, nn * 16, hipMemcpyHostToDevice, m->stream));
        SNK_HIP_CHECK(hipMemcpyAsync(d + o_d, desc, nn * 32, hipMemcpyHostToDevice, m->stream));
        SNK_HIP_CHECK(hipMemcpyAsync(d + o_h, has, nn, hipMemcpyHostToDevice, m->stream));
    }
    K->np   = reinterpret_cast<const double*>(d);
    K->desc = reinterpret_cast<const uint4*>(d + o_d);
    K->has  = reinterpret_cast<const u8*>(d + o_h);
    if (extra_ptr) *extra_ptr = d + o_x;
    return SNK_OK;
}
}  // namespace

int snk_match_triangulation_bow(snk_matcher* m, const snk_camera* cam, const double E12[9], const double (*np1)[2],
                                const uint64_t (*desc1)[4], const uint8_t* has_mp1, int n1, const snk_bow_features* bow1,
                                const double (*np2)[2], const uint64_t (*desc2)[4], const uint8_t* has_mp2, int n2,
                                const snk_bow_features* bow2, float epipolar_distance, int feature_distance, int32_t (*pairs)[2],
                                int* n_matches)
{
    SNK_REQUIRE(m != nullptr && n_matches != nullptr, "NULL argument");
    *n_matches = 0;
    SNK_REQUIRE(cam != nullptr && E12 != nullptr, "camera / E is NULL");
    SNK_REQUIRE(n1 >= 0 && n1 < (int)PJ_IDX_MASK && (n1 == 0 || (np1 && desc1 && has_mp1)), "bad keyframe-1 arrays");
    SNK_REQUIRE(n2 >= 0 && n2 < (int)PJ_IDX_MASK && (n2 == 0 || (np2 && desc2 && has_mp2)), "bad keyframe-2 arrays");
    int rc;
    if ((rc = check_bow(bow1, n1)) != SNK_OK) return rc;
    if ((rc = check_bow(bow2, n2)) != SNK_OK) return rc;
    // intersect the two ascending node lists (:34-98); one item per unmatched keyframe-1 feature of a common node,
    // in the order the reference visits them
    std::vector<int4> items;
    for (int a = 0, b = 0; a < bow1->n_nodes && b < bow2->n_nodes;)
    {
        if (bow1->node_id[a] == bow2->node_id[b])
        {
            const int lo = bow2->node_start[b], hi = bow2->node_start[b + 1];
            if (hi > lo)
                for (int u = bow1->node_start[a]; u < bow1->node_start[a + 1]; ++u)
                    if (!has_mp1[bow1->features[u]]) items.push_back(make_int4(bow1->features[u], lo, hi, 0));
            ++a;
            ++b;
        }
        else if (bow1->node_id[a] < bow2->node_id[b])
            ++a;
        else
            ++b;
    }
    const int n_items = (int)items.size();
    if (n_items == 0) return SNK_OK;
    SNK_REQUIRE(pairs != nullptr, "pairs is NULL");
    SNK_HIP_CHECK(hipSetDevice(m->device));
    const size_t nf2 = (size_t)bow2->node_start[bow2->n_nodes];
    KfDev K1, K2;
    char *x1 = nullptr, *x2 = nullptr;
    if ((rc = upload_kf(m, m->q, np1, desc1, has_mp1, n1, (size_t)n_items * sizeof(int4), &K1, &x1)) != SNK_OK) return rc;
    if ((rc = upload_kf(m, m->t, np2, desc2, has_mp2, n2, nf2 * 4, &K2, &x2)) != SNK_OK) return rc;
    if ((rc = m->out.reserve((size_t)n_items * 4)) != SNK_OK) return rc;
    SNK_HIP_CHECK(hipMemcpyAsync(x1, items.data(), (size_t)n_items * sizeof(int4), hipMemcpyHostToDevice, m->stream));
    SNK_HIP_CHECK(hipMemcpyAsync(x2, bow2->features, nf2 * 4, hipMemcpyHostToDevice, m->stream));
    EpiDev T;
    memcpy(T.E, E12, sizeof(T.E));
    const double th_chi1 = (double)(epipolar_distance * 2) / cam->fx;  // :20
    T.th_chi2            = th_chi1 * th_chi1;
    T.feature_distance   = feature_distance;
    hipLaunchKernelGGL(triangulate_bow_kernel, dim3(ceil_div(n_items, 16)), dim3(256), 0, m->stream, T,
                       reinterpret_cast<const int4*>(x1), n_items, K1.np, K1.desc, K2.np, K2.desc, K2.has,
                       reinterpret_cast<const int*>(x2), m->out.as<int>());
    SNK_LAUNCH_CHECK();
    std::vector<int> best((size_t)n_items);
    SNK_HIP_CHECK(hipMemcpyAsync(best.data(), m->out.p, (size_t)n_items * 4, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    int cnt = 0;
    for (int k = 0; k < n_items; ++k)
        if (best[k] >= 0)
        {
            pairs[cnt][0] = items[k].x;
            pairs[cnt][1] = best[k];
            ++cnt;
        }
    *n_matches = cnt;
    return SNK_OK;
}

int snk_match_triangulation_bf(snk_matcher* m, const snk_camera* cam, const double E12[9], const double (*np1)[2],
                               const uint64_t (*desc1)[4], const uint8_t* has_mp1, int n1, const double (*np2)[2],
                               const uint64_t (*desc2)[4], const uint8_t* has_mp2, int n2, int feature_distance,
                               int32_t* match_idx2, int* n_matches)
{
    SNK_REQUIRE(m != nullptr && n_matches != nullptr, "NULL argument");
    *n_matches = 0;
    SNK_REQUIRE(cam != nullptr && E12 != nullptr, "camera / E is NULL");
    SNK_REQUIRE(n1 >= 0 && (n1 == 0 || (np1 && desc1 && has_mp1 && match_idx2)), "bad keyframe-1 arrays");
    SNK_REQUIRE(n2 >= 0 && n2 < (int)PJ_IDX_MASK && (n2 == 0 || (np2 && desc2 && has_mp2)), "bad keyframe-2 arrays");
    if (n1 == 0) return SNK_OK;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    int rc;
    KfDev K1, K2;
    if ((rc = upload_kf(m, m->q, np1, desc1, has_mp1, n1, 0, &K1, nullptr)) != SNK_OK) return rc;
    if ((rc = upload_kf(m, m->t, np2, desc2, has_mp2, n2, 0, &K2, nullptr)) != SNK_OK) return rc;
    if ((rc = m->out.reserve((size_t)n1 * 4)) != SNK_OK) return rc;
    EpiDev T;
    memcpy(T.E, E12, sizeof(T.E));
    const double th_chi1 = 10 / cam->fx;  // :107 (the epipolarDistance argument is not used by the reference)
    T.th_chi2            = th_chi1 * th_chi1;
    T.feature_distance   = feature_distance;
    hipLaunchKernelGGL(triangulate_bf_kernel, dim3(ceil_div(n1, 4)), dim3(256), 0, m->stream, T, K1.np, K1.desc, K1.has, n1, K2.np,
                       K2.desc, K2.has, n2, m->out.as<int>());
    SNK_LAUNCH_CHECK();
    SNK_HIP_CHECK(hipMemcpyAsync(match_idx2, m->out.p, (size_t)n1 * 4, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    int cnt = 0;
    for (int i = 0; i < n1; ++i) cnt += match_idx2[i] >= 0 ? 1 : 0;
    *n_matches = cnt;
    return SNK_OK;
}

int snk_match_relink(snk_matcher* m, const snk_frame_view* frame, const snk_camera* cam, const double pose[7],
                     const snk_relink_query* queries, int n, float radius, double outlier_threshold, int feature_threshold,
                     int32_t* action, int32_t* best_idx, int* n_changed)
{
    SNK_REQUIRE(m != nullptr && n_changed != nullptr, "NULL argument");
    *n_changed = 0;
    SNK_REQUIRE(n >= 0 && (n == 0 || (queries && action && best_idx)), "bad query arrays");
    SNK_REQUIRE(radius >= 0.f && outlier_threshold >= 0.0, "bad thresholds");
    CamDev C;
    FrameDev F;
    int rc;
    if ((rc = make_cam(cam, pose, &C)) != SNK_OK) return rc;
    SNK_REQUIRE(frame != nullptr, "frame view is NULL");
    for (int q = 0; q < n; ++q) SNK_REQUIRE(queries[q].feature >= 0 && queries[q].feature < frame->n, "query feature out of range");
    SNK_HIP_CHECK(hipSetDevice(m->device));
    if ((rc = upload_frame(m, frame, &F)) != SNK_OK) return rc;
    if (n == 0) return SNK_OK;
    const size_t nq = (size_t)n;
    if ((rc = m->q.reserve(nq * sizeof(snk_relink_query))) != SNK_OK) return rc;
    if ((rc = m->out.reserve(nq * 8)) != SNK_OK) return rc;
    SNK_HIP_CHECK(hipMemcpyAsync(m->q.p, queries, nq * sizeof(snk_relink_query), hipMemcpyHostToDevice, m->stream));
    int* d_action = m->out.as<int>();
    int* d_best   = d_action + nq;
    hipLaunchKernelGGL(relink_kernel, dim3(ceil_div(n, 16)), dim3(256), 0, m->stream, F, C, m->q.as<snk_relink_query>(), n, radius,
                       outlier_threshold * outlier_threshold, feature_threshold, d_action, d_best);
    SNK_LAUNCH_CHECK();
    SNK_HIP_CHECK(hipMemcpyAsync(action, d_action, nq * 4, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipMemcpyAsync(best_idx, d_best, nq * 4, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    int cnt = 0;
    for (int q = 0; q < n; ++q) cnt += action[q] != 0 ? 1 : 0;
    *n_changed = cnt;
    return SNK_OK;
}
}
