// ORB extractor for gfx950 — replaces Saiga::ORBExtractor / ORBExtractorGPU::Detect
// (reference call sites Snake/Preprocess/FeatureDetector.cpp:31-41,119,124,149,154).
//
// Semantics: "snk-orb v1"  — the published ORB-SLAM2 extractor with every
// unspecified step fixed in integer / explicitly ordered float arithmetic.  The whole library is
// built with -ffp-contract=off, so the few float expressions below round exactly like their
// scalar restatement.
//
// Pipeline (all kernels batched over B images; level / FAST / describe run on XCD-aware 1-D grids so that
// an image's whole chain stays on one XCD's L2; a batch can be split into launch chains on two streams):
//   level_kernel       one streaming pass per level (L launches): a wavefront walks down a <= 244-px column
//                      strip with the 7-row window in registers; 7x7 Gaussian of level l (v_alignbyte +
//                      v_dot4_u32_u8 + v_dot2_u32_u16) and the 11-bit fixed-point bilinear down-scale to
//                      level l+1 in the same pass (resize_kernel when scale > 2 or the level is < 8 px)
//   fast_kernel        one WAVEFRONT per ~30x30 FAST cell: image tile -> its LDS slice (aligned dwords),
//                      quick opposite-pair bound on every pixel, compaction, exact FAST-9/16 score of the
//                      survivors in packed 16-bit lanes, in-cell 3x3 NMS, ini/min threshold fallback,
//                      candidates ranked by strength
//   distribute_kernel  one workgroup per (image, level): quadtree distribution on sorted
//                      subdivision keys (bitonic sort in LDS + histogram of common-prefix lengths)
//   describe_kernel    one wavefront per 4 keypoints, all loads issued up front: integer IC moments
//                      (v_dot4 against a disc table), polynomial atan2, 256 steered BRIEF tests read from an
//                      LDS copy of the blurred patch, 4 ballots assemble the 256-bit descriptor
#include "common.hpp"

#include <array>
#include <vector>

namespace snk
{
namespace
{
using u8  = unsigned char;
using u16 = unsigned short;
using u32 = unsigned int;
using u64 = unsigned long long;

constexpr int MAX_LEVELS     = 16;
constexpr int EDGE_THRESHOLD = 19;
constexpr int MIN_BORDER     = 16;
constexpr int CELL_W         = 30;
constexpr int CELL_SLOTS     = 64;   // strongest candidates kept per cell
constexpr int KEY_DIGITS     = 16;
constexpr int DEFAULT_LEVEL_CAP = 8192;
#ifndef SNK_DIST_MIN_WAVES
#define SNK_DIST_MIN_WAVES 1  // build-time A/B: 8 = distribute_kernel must fit four 512-thread workgroups per CU (<= 64 VGPRs, <= 80 SGPRs)
#endif
#ifndef SNK_FAST_MIN_WAVES
#define SNK_FAST_MIN_WAVES 1  // build-time A/B: wavefronts per SIMD the register allocator must leave room for in fast_kernel
#endif
constexpr int FAST_CPW_DEFAULT  = 8;     // FAST cells per wavefront for big launches (SNK_ORB_FAST_CPW; measured in DESIGN.md section 5)

constexpr signed char k_pattern[1024] = {
#include "brief_pattern_31.inc"
};
constexpr int k_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

// BRIEF test pairs as floats: entry b = (x0, y0, x1, y1) of bit b (lanes keep 4 entries in registers).
struct PatternTab
{
    float v[256][4];
};
constexpr PatternTab make_pattern_tab()
{
    PatternTab t{};
    for (int b = 0; b < 256; ++b)
        for (int e = 0; e < 4; ++e) t.v[b][e] = (float)k_pattern[4 * b + e];
    return t;
}
__device__ const PatternTab c_pattern_f = make_pattern_tab();

// Intensity-centroid weights of the radius-15 disc for aligned-dword reads.  The 31 x 31 window of a
// keypoint starts sh (0..3) bytes into its first dword; item = row * 9 + dword (9 dwords per row).
// Entry [sh][item] = (wx, m): per byte j, m_j = 1 inside the disc else 0 and wx_j = (ux + 15) * m_j with
// ux the column offset -> m10 = sum dot4(p, wx) - 15 * sum dot4(p, m), m01 = sum vy * dot4(p, m).
constexpr int MOM_ITEMS = 31 * 9, MOM_PAD = 320;  // padded to 5 x 64 lanes, zero weights
struct MomentTab
{
    u32 v[4][MOM_PAD][2];
};
constexpr MomentTab make_moment_tab()
{
    MomentTab t{};
    for (int sh = 0; sh < 4; ++sh)
        for (int item = 0; item < MOM_ITEMS; ++item)
        {
            const int row = item / 9, d = item % 9;
            const int vy = row - 15, av = vy < 0 ? -vy : vy;
            u32 wx = 0, m = 0;
            for (int j = 0; j < 4; ++j)
            {
                const int ux = 4 * d + j - sh - 15, au = ux < 0 ? -ux : ux;
                if (au <= k_umax[av])
                {
                    wx |= (u32)(ux + 15) << (8 * j);
                    m |= 1u << (8 * j);
                }
            }
            t.v[sh][item][0] = wx;
            t.v[sh][item][1] = m;
        }
    return t;
}
__device__ const MomentTab c_moment = make_moment_tab();

struct LevelInfo
{
    int w, h, pitch;
    long long img_stride;  // bytes between consecutive images of this level
    u8* base;              // level buffer (levels >= 1); level 0 comes from the caller
    u8* blur;              // blurred level (all levels), same pitch / stride as the level buffers
    int strip_stride, n_strips, n_bands, unit_off;  // streaming pass: column strips x row bands of this level
    int store_end;         // the last strip writes the blurred row up to this column (>= w: into the row padding, whole 64-byte sectors)
    const int* ymap;      // [h] destination row of level l+1 whose upper source row is this row, or -1
    const int* strip_dx;  // [n_strips + 1] first destination dword (4 px) of level l+1 owned by each strip
    int ncols, nrows, wcell, hcell;
    int cell_off;          // first cell of this level in the per-image cell arrays
    int nfeat;             // features wanted on this level
    int slot_off, slot_cap;  // per-image slots for the selected keypoints of this level
    int nroots;
    float scale;
    const int* xofs;  // resize tables (device): source column / weight per destination column
    const int* xw1;
    const int* yofs;
    const int* yw1;
    long long blur_stride;  // bytes between consecutive images of the BLURRED level: pitch x (h rounded up to 4) -- room for the tiled layout
};

// Tiled layout of the blurred planes (round 6, an EXPERIMENT -- see orb_blur_mode): a 128-byte line holds 32 pixels x 4 rows, so the
// 37 x 40-pixel patch describe_kernel reads per keypoint touches ~23 lines instead of ~51 (measured upper bound with contiguous
// patches: 1.39 -> 1.135 ms, profiles/r06/r06o_*).  Byte of pixel (x, y): (y >> 2) * 4 * pitch + (x >> 5) * 128 + (y & 3) * 32 + (x & 31).
__device__ __forceinline__ u32 blur_tiled_x(int x) { return (u32)(((x >> 5) << 7) + (x & 31)); }
__device__ __forceinline__ u32 blur_tiled_y(int y, int pitch) { return (u32)((y >> 2) * 4 * pitch + ((y & 3) << 5)); }

// what fast_kernel reads of a level: the head of LevelInfo, taken with one 32-byte load
struct LevelHead
{
    int w, h, pitch, pad_;
    long long img_stride;
    u8* base;
};
static_assert(sizeof(LevelHead) == 32 && offsetof(LevelInfo, w) == 0 && offsetof(LevelInfo, h) == 4 && offsetof(LevelInfo, pitch) == 8 &&
                  offsetof(LevelInfo, img_stride) == 16 && offsetof(LevelInfo, base) == 24,
              "LevelHead mirrors the first 32 bytes of LevelInfo");

struct Layout
{
    int n_levels;
    int total_cells;
    int total_slots;
    int total_units;
    int level_cap;
    // per-wavefront LDS slice of fast_kernel (bytes; sized from the largest cell of the layout)
    int f_tile_pitch_dw, f_s_pitch, f_off_s, f_off_surv, f_off_list, f_off_cnt, f_lds_wave, f_surv_cap;
    const int4* cell_tab;  // [total_cells] (x0, y0, cw | ch << 16, level) of every FAST cell: one scalar load instead of a level search and two integer divisions per cell
    LevelInfo lv[MAX_LEVELS];
};

// ------------------------------------------------------------------------------------------------
// pyramid
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_kernel(const u8* __restrict__ src, int spitch, long long sstride, int sw,
                                                     int sh, u8* __restrict__ dst, int dpitch, long long dstride, int dw,
                                                     int dh, const int* __restrict__ xofs, const int* __restrict__ xw1,
                                                     const int* __restrict__ yofs, const int* __restrict__ yw1)
{
    const int b  = blockIdx.z;
    const int y  = blockIdx.y;
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x4 >= dw) return;
    const u8* s  = src + (long long)b * sstride;
    const int sy = yofs[y], wy1 = yw1[y], wy0 = 2048 - wy1;
    const int sy1 = sy + 1 < sh ? sy + 1 : sh - 1;
    const u8* r0  = s + (long long)sy * spitch;
    const u8* r1  = s + (long long)sy1 * spitch;
    u32 packed    = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        const int x = x4 + k;
        if (x < dw)
        {
            const int sx = xofs[x], wx1 = xw1[x], wx0 = 2048 - wx1;
            const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            const int v   = ((int)r0[sx] * wx0 + (int)r0[sx1] * wx1) * wy0 + ((int)r1[sx] * wx0 + (int)r1[sx1] * wx1) * wy1;
            packed |= (u32)((v + (1 << 21)) >> 22) << (8 * k);
        }
    }
    u8* d = dst + (long long)b * dstride + (long long)y * dpitch + x4;
    if (x4 + 3 < dw)
        *reinterpret_cast<u32*>(d) = packed;  // dpitch is a multiple of 64 and x4 of 4: aligned
    else
        for (int k = 0; k < 4 && x4 + k < dw; ++k) d[k] = (u8)(packed >> (8 * k));
}

// ------------------------------------------------------------------------------------------------
// tile loader: tile[r][d] (dwords, row pitch PITCH_DW) = image bytes (ys + r, xs + 4d .. xs + 4d + 3),
// xs a multiple of 4; coordinates outside the image are reflect-101'd.  A dword that lies inside the
// image row is fetched with one aligned 32-bit load when the image base / pitch are 4-aligned.
// Thread -> (row, dword) by shift/mask (no integer division).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int i, int n)
{
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// The loads of a batch are all issued before the first LDS store, so a wavefront keeps BATCH
// memory requests in flight instead of paying one full memory latency per dword.
template <int BATCH>
__device__ __forceinline__ void load_tile(u32* tile, int pitch_dw, const u8* __restrict__ src, int pitch, int w, int h, int xs,
                                          int ys, int ndw, int nrows, bool aligned, int tid, int nthreads)
{
    const int lc    = ndw <= 8 ? 3 : (ndw <= 16 ? 4 : (ndw <= 32 ? 5 : 6));
    const int total = nrows << lc;
    if (aligned && xs >= 0 && xs + 4 * ndw <= w && ys >= 0 && ys + nrows <= h)
    {
        // interior tile (tile-uniform test): branch-free aligned dword loads, BATCH requests in flight.
        // Items are numbered densely (row = item / ndw via an exact float reciprocal): a power-of-two row
        // length would waste 6 of 16 slots for the usual 10-dword FAST tile and cost a second batch.
        const int dense   = nrows * ndw;
        const float inv_n = 1.0f / (float)ndw;
        for (int i0 = tid; i0 < dense; i0 += nthreads * BATCH)
        {
            u32 v[BATCH];
            int at[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k)
            {
                const int i = min(i0 + k * nthreads, dense - 1);
                const int r = (int)(((float)i + 0.5f) * inv_n), d = i - r * ndw;
                at[k]       = r * pitch_dw + d;
                v[k]        = *reinterpret_cast<const u32*>(src + (u32)((ys + r) * pitch + xs + 4 * d));
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k)
                if (i0 + k * nthreads < dense) tile[at[k]] = v[k];
        }
        return;
    }
    for (int i0 = tid; i0 < total; i0 += nthreads * BATCH)
    {
        u32 v[BATCH];
        int dst[BATCH];  // LDS dword index, -1: nothing to store, -2: border / unaligned (slow path)
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
        {
            const int i = i0 + k * nthreads;
            const int r = i >> lc, d = i & ((1 << lc) - 1);
            dst[k] = -1;
            v[k]   = 0;
            if (i < total && d < ndw)
            {
                const int x = xs + 4 * d;
                dst[k]      = r * pitch_dw + d;
                if (aligned && x >= 0 && x + 3 < w)
                    v[k] = *reinterpret_cast<const u32*>(src + (long long)reflect101(ys + r, h) * pitch + x);
                else
                    dst[k] = -2 - dst[k];
            }
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
        {
            if (dst[k] <= -2)
            {
                const int i = i0 + k * nthreads;
                const int r = i >> lc, d = i & ((1 << lc) - 1);
                const int x = xs + 4 * d;
                const u8* rp = src + (long long)reflect101(ys + r, h) * pitch;
                v[k] = (u32)rp[reflect101(x, w)] | ((u32)rp[reflect101(x + 1, w)] << 8) | ((u32)rp[reflect101(x + 2, w)] << 16) |
                       ((u32)rp[reflect101(x + 3, w)] << 24);
                dst[k] = -2 - dst[k];
            }
            if (dst[k] >= 0) tile[dst[k]] = v[k];
        }
    }
}

// Interior FAST tile with a row pitch of NQ 16-byte quads: item i = (row, quad) lands at LDS byte 16 * i, so a lane
// moves 16 bytes with a handful of address instructions (the dword loader above pays ~25 per dword, three of them
// quarter-rate 32-bit multiplies).  Global addresses are only 4-byte aligned; quads right of the tile are skipped.
typedef u32 u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef u32 u32x4_a16 __attribute__((ext_vector_type(4), aligned(16)));

template <int NQ>
__device__ __forceinline__ void load_tile_quads(u32* tile, const u8* __restrict__ src, int pitch, int xs, int ys, int ndw, int nrows,
                                                int lane)
{
    static_assert(NQ == 3 || NQ == 4, "row pitch of 48 or 64 bytes");
    constexpr int B = 3;
    const int items = nrows * NQ;
    const u8* base  = src + (u32)(__mul24(ys, pitch) + xs);
    u32x4_a16* tq   = reinterpret_cast<u32x4_a16*>(tile);
    for (int i0 = lane; i0 < items; i0 += 64 * B)
    {
        u32x4_a4 v[B];
        bool ok[B];
#pragma unroll
        for (int k = 0; k < B; ++k)
        {
            const int i = i0 + 64 * k;
            const int r = NQ == 4 ? i >> 2 : (i * 171) >> 9;  // i / 3 for i < 512
            const int c = i - r * NQ;
            ok[k]       = i < items && 4 * c < ndw;
            v[k]        = u32x4_a4{0, 0, 0, 0};
            if (ok[k]) v[k] = *reinterpret_cast<const u32x4_a4*>(base + (u32)(__mul24(r, pitch) + 16 * c));
        }
#pragma unroll
        for (int k = 0; k < B; ++k)
            if (ok[k]) tq[i0 + 64 * k] = v[k];
    }
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16 score: S = max over the 16 arcs of 9 of min(ring - c), and of min(c - ring).
// corner(t) <=> S > t.  Sliding-window minima by doubling (2,4,8,+1), two pixels per lane in packed
// 16-bit lanes (v_pk_min_i16 / v_pk_max_i16).
// ------------------------------------------------------------------------------------------------
typedef short short2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ short2_t pk_min(short2_t a, short2_t b)
{
    return __builtin_elementwise_min(a, b);
}
__device__ __forceinline__ short2_t pk_max(short2_t a, short2_t b)
{
    return __builtin_elementwise_max(a, b);
}

__device__ __forceinline__ short2_t fast_score16_pk(const short2_t (&d)[16])
{
    short2_t a2[16], a4[16], b2[16], b4[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        a2[i] = pk_min(d[i], d[(i + 1) & 15]);
        b2[i] = pk_max(d[i], d[(i + 1) & 15]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        a4[i] = pk_min(a2[i], a2[(i + 2) & 15]);
        b4[i] = pk_max(b2[i], b2[(i + 2) & 15]);
    }
    short2_t bright = {-1000, -1000}, dark = {1000, 1000};
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        const short2_t a9 = pk_min(pk_min(a4[i], a4[(i + 4) & 15]), d[(i + 8) & 15]);
        const short2_t b9 = pk_max(pk_max(b4[i], b4[(i + 4) & 15]), d[(i + 8) & 15]);
        bright            = pk_max(bright, a9);
        dark              = pk_min(dark, b9);
    }
    return pk_max(bright, -dark);
}

// The same score on RAW ring bytes, two pixels per lane, with gfx950's three-input packed minimum / maximum:
// min over a 9-window of (ring - c) = (min over the window of ring) - c, so the network runs on the bytes themselves.
// A byte b in a 16-bit half is the f16 DENORMAL b * 2^-24; positive f16 values order like their bit patterns and the
// kernel runs with f16 denormals preserved (the compiler default, .amdhsa_float_denorm_mode_16_64 = 3; checked on the
// hardware by tools/probes/pk_min3_probe.hip), so v_pk_minimum3_f16 / v_pk_maximum3_f16 are exact integer min3 / max3 of
// the two halves.  9-window = 3 windows of 3: 16 + 16 + 8 instructions per polarity instead of 16 + 16 + 32 + 16.
__device__ __forceinline__ u32 pk_min3(u32 a, u32 b, u32 c)
{
    u32 o;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}
__device__ __forceinline__ u32 pk_max3(u32 a, u32 b, u32 c)
{
    u32 o;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 pku_min(u32 a, u32 b)
{
    return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b)));
}
__device__ __forceinline__ u32 pku_max(u32 a, u32 b)
{
    return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b)));
}
__device__ __forceinline__ short2_t pk_sub(u32 a, u32 b)
{
    return __builtin_bit_cast(short2_t, a) - __builtin_bit_cast(short2_t, b);
}

// r[i]: ring byte i of pixel 0 | ring byte i of pixel 1 << 16; c likewise for the centres.  Returns the two scores.
__device__ __forceinline__ short2_t fast_score16_raw(const u32 (&r)[16], u32 c)
{
    u32 a3[16], b3[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        a3[i] = pk_min3(r[i], r[(i + 1) & 15], r[(i + 2) & 15]);
        b3[i] = pk_max3(r[i], r[(i + 1) & 15], r[(i + 2) & 15]);
    }
    u32 a9[16], b9[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        a9[i] = pk_min3(a3[i], a3[(i + 3) & 15], a3[(i + 6) & 15]);
        b9[i] = pk_max3(b3[i], b3[(i + 3) & 15], b3[(i + 6) & 15]);
    }
    // max of the 16 window minima / min of the 16 window maxima: 16 -> 6 -> 2 -> 1
    const u32 hi = pk_max3(pk_max3(pk_max3(a9[0], a9[1], a9[2]), pk_max3(a9[3], a9[4], a9[5]), pk_max3(a9[6], a9[7], a9[8])),
                           pk_max3(pk_max3(a9[9], a9[10], a9[11]), pk_max3(a9[12], a9[13], a9[14]), a9[15]), a9[15]);
    const u32 lo = pk_min3(pk_min3(pk_min3(b9[0], b9[1], b9[2]), pk_min3(b9[3], b9[4], b9[5]), pk_min3(b9[6], b9[7], b9[8])),
                           pk_min3(pk_min3(b9[9], b9[10], b9[11]), pk_min3(b9[12], b9[13], b9[14]), b9[15]), b9[15]);
    return pk_max(pk_sub(hi, c), pk_sub(c, lo));
}

// XCD-aware grids: workgroup L (linear index, x fastest) runs on XCD L % 8 (observed dispatch order, a speed matter only).  All
// workgroups of an image get the same L % 8 in every kernel of the chain, so what one stage writes for an
// image (next level, blurred level, candidates) is read by the next stage through the same XCD's L2.
__device__ __forceinline__ bool xcd_image_map(int gx, int batch, int& b, int& bx)
{
    // 2-D grids (x fastest in the dispatch order): for batches of >= 16 images grid.x = 8 * gx, so the linear workgroup
    // index is congruent to blockIdx.x mod 8 and image 8 * blockIdx.y + (blockIdx.x & 7) stays on one XCD -- shifts and
    // masks only (the 1-D form needed an integer division per wavefront, ~35 instructions of the ~860 of a FAST cell)
    if (batch >= 16)
    {
        b  = blockIdx.y * 8 + (blockIdx.x & 7);
        bx = blockIdx.x >> 3;
    }
    else
    {
        b  = blockIdx.y;
        bx = blockIdx.x;
    }
    (void)gx;
    return b < batch;
}
static inline dim3 xcd_grid(int gx, int batch) { return batch >= 16 ? dim3(8 * gx, (batch + 7) / 8) : dim3(gx, batch); }

// One WAVEFRONT per FAST cell (4 cells per workgroup, no workgroup barriers: the phases of a cell
// only communicate through that wavefront's own LDS slice, which the LDS serves in program order).
// Phase A runs the cheap opposite-pair bound on every pixel and compacts the ~10 % that can still
// be corners; phase B evaluates the exact score only for those, densely, two per lane; NMS and the
// threshold fallback then walk the compacted list.  LDS slice per wavefront (sizes from the layout):
// image tile | score map with zero ring | quick-test survivors | NMS survivors | 3 counters.
// NQ = 3 / 4: compile-time tile row pitch of 48 / 64 bytes and score-map pitch of 40 / 64 (what the layout picks
// for cells up to 36 / 52 pixels wide): row addressing by shifts and the quad loader; NQ = 0: pitches from the layout.
template <int NQ, bool LOOP>
__global__ __launch_bounds__(256, SNK_FAST_MIN_WAVES) void fast_kernel(Layout L, const u8* __restrict__ img0, int pitch0, long long stride0,
                                                   int aligned0, int ini_th, int min_th, u32* __restrict__ cand,
                                                   u16* __restrict__ cell_cnt, int gx, int batch, int dbg_stop, int cpw,
                                                   int* __restrict__ queue_reset /* NULL or the distribute queue's counter: zeroed here instead of by a fill launch */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (queue_reset && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *queue_reset = 0;
    int b, bxi;
    if (!xcd_image_map(gx, batch, b, bxi)) return;
    // cpw cells per wavefront, one after the other (cells bxi * 4 * cpw + wave + 4 k): a cell is ~7 us of work, and with one cell per
    // wavefront the workgroup turnover (launch, teardown) left 6.2 of the 8 wavefront slots of a SIMD filled on average
    // (PMC r03g: wave cycles / busy cycles); longer-lived workgroups keep the slots occupied.  Nothing is carried from cell to cell.
    // LOOP = false: the one-cell form without the loop (53 registers, 8 wavefronts per SIMD; the loop form keeps more values live:
    // 71 registers, 7 wavefronts)
    for (int kc = 0; kc < (LOOP ? cpw : 1); ++kc)
    {
    const int cid  = LOOP ? (bxi * cpw + kc) * 4 + wave : bxi * 4 + wave;
    if (cid >= L.total_cells) return;  // whole wavefront (cells ascend with kc)
    __builtin_amdgcn_wave_barrier();   // the previous cell's LDS reads are done (one wavefront: program order)
    unsigned char* slice = fsm + wave * L.f_lds_wave;
    u32* tile_dw      = reinterpret_cast<u32*>(slice);
    u8* S             = slice + L.f_off_s;
    u16* surv         = reinterpret_cast<u16*>(slice + L.f_off_surv);
    u32* list         = reinterpret_cast<u32*>(slice + L.f_off_list);
    const int TPD = NQ ? NQ * 4 : L.f_tile_pitch_dw, TP = TPD * 4, SP = NQ == 3 ? 40 : (NQ == 4 ? 64 : L.f_s_pitch);

    // The cell's entry by a SCALAR load (the index is wave-uniform; as a vector load it was an L2 round trip per cell in front of
    // everything else), and the five fields of the level the cell needs -- w, h, pitch, img_stride, base: the first 32 bytes of
    // LevelInfo -- by ONE scalar load (read field by field they were four dependent scalar-load round trips per cell).
    int4 ct;
    {
        const int4* cp = L.cell_tab + cid;
        asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ct) : "s"(cp) : "memory");
    }
    const int l   = ct.w;
    const LevelHead lv = *reinterpret_cast<const LevelHead*>(&L.lv[l]);
    const int x0 = ct.x, y0 = ct.y;
    const int cw = (int)(short)(ct.z & 0xFFFF), ch = (int)(short)((unsigned)ct.z >> 16);
    const long long cell_index = (long long)b * L.total_cells + cid;
    if (cw <= 0 || ch <= 0)
    {
        if (lane == 0) cell_cnt[cell_index] = 0;
        if (LOOP) continue;
        return;
    }
    const u8* src      = l == 0 ? img0 + (long long)b * stride0 : lv.base + (long long)b * lv.img_stride;
    const int pitch    = l == 0 ? pitch0 : lv.pitch;
    const bool aligned = l == 0 ? aligned0 != 0 : true;

    // image tile with the 3-pixel ring halo (always inside the image: cells start at x,y >= 19)
    const int xs  = (x0 - 3) & ~3;
    const int sh  = (x0 - 3) - xs;  // tile byte column of cell pixel px is px + 3 + sh
    const int ndw = (sh + cw + 6 + 3) >> 2;
    // (tile-uniform) interior test of the quad loader: the last quad read of a row ends inside that row
    if (NQ != 0 && aligned && xs >= 0 && xs + 16 * ((ndw + 3) >> 2) <= lv.w && y0 - 3 >= 0 && y0 + ch + 3 <= lv.h)
    {
        if constexpr (NQ != 0) load_tile_quads<NQ>(tile_dw, src, pitch, xs, y0 - 3, ndw, ch + 6, lane);
    }
    else
        load_tile<6>(tile_dw, TPD, src, pitch, lv.w, lv.h, xs, y0 - 3, ndw, ch + 6, aligned, lane, 64);
    for (int i = lane; i < (((ch + 2) * SP + 15) >> 4); i += 64) reinterpret_cast<u32x4_a16*>(S)[i] = u32x4_a16{0, 0, 0, 0};
    __builtin_amdgcn_wave_barrier();
    const u8* tile = reinterpret_cast<const u8*>(tile_dw);
    if (dbg_stop == 1) return;  // SNK_ORB_FAST_STOP (timing experiments only, results are then meaningless): after the tile load

    // phase B: exact score of the survivors, two per lane
    // Returns how many of the ns survivors score above min_th; those are moved to the front of the list (in place: a round writes
    // no further than it has read), so that the non-maximum pass only walks real corners -- about a third of the survivors.
    auto score_survivors = [&](int ns) -> int
    {
    int n2 = 0;
    for (int j0 = 0; j0 < ns; j0 += 128)  // wave-uniform rounds (ballots inside)
    {
        const int j = j0 + lane * 2;
        const bool a0 = j < ns, a1 = j + 1 < ns;
        const int e0 = surv[a0 ? j : ns - 1], e1 = surv[a1 ? j + 1 : ns - 1];
        const int px0 = e0 & 63, py0 = e0 >> 6, px1 = e1 & 63, py1 = e1 >> 6;
        const u8* t0 = tile + (py0 + 3) * TP + px0 + 3 + sh;
        const u8* t1 = tile + (py1 + 3) * TP + px1 + 3 + sh;
        short2_t s;
        if constexpr (NQ != 0)
        {
            u32 r[16];
#define RING(i, dx, dy) r[i] = (u32)t0[(dy)*TP + (dx)] | ((u32)t1[(dy)*TP + (dx)] << 16);
            RING(0, 0, 3) RING(1, 1, 3) RING(2, 2, 2) RING(3, 3, 1) RING(4, 3, 0) RING(5, 3, -1) RING(6, 2, -2) RING(7, 1, -3)
            RING(8, 0, -3) RING(9, -1, -3) RING(10, -2, -2) RING(11, -3, -1) RING(12, -3, 0) RING(13, -3, 1) RING(14, -2, 2)
            RING(15, -1, 3)
#undef RING
            s = fast_score16_raw(r, (u32)t0[0] | ((u32)t1[0] << 16));
        }
        else
        {
            const short v0 = t0[0], v1 = t1[0];
            short2_t d[16];
#define RING(i, dx, dy) d[i] = short2_t{(short)(t0[(dy)*TP + (dx)] - v0), (short)(t1[(dy)*TP + (dx)] - v1)};
            RING(0, 0, 3) RING(1, 1, 3) RING(2, 2, 2) RING(3, 3, 1) RING(4, 3, 0) RING(5, 3, -1) RING(6, 2, -2) RING(7, 1, -3)
            RING(8, 0, -3) RING(9, -1, -3) RING(10, -2, -2) RING(11, -3, -1) RING(12, -3, 0) RING(13, -3, 1) RING(14, -2, 2)
            RING(15, -1, 3)
#undef RING
            s = fast_score16_pk(d);
        }
        if (a0) S[(py0 + 1) * SP + px0 + 1] = (u8)(s.x < 0 ? 0 : s.x);
        if (a1) S[(py1 + 1) * SP + px1 + 1] = (u8)(s.y < 0 ? 0 : s.y);
        const bool k0 = a0 && (int)s.x > min_th, k1 = a1 && (int)s.y > min_th;
        const u64 m0 = __builtin_amdgcn_ballot_w64(k0), m1 = __builtin_amdgcn_ballot_w64(k1);
        const int c0 = __popcll(m0);
        if (k0) surv[n2 + __builtin_amdgcn_mbcnt_hi((u32)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m0, 0u))] = (u16)e0;
        if (k1) surv[n2 + c0 + __builtin_amdgcn_mbcnt_hi((u32)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m1, 0u))] = (u16)e1;
        n2 += c0 + __popcll(m1);
    }
    return n2;
    };

    // phase A: every 9-arc holds one pixel of each opposite pair -> S <= min_i max(d_i, d_{i+8}).  The loops
    // are wave-uniform (lanes outside the cell are predicated), so the running survivor count lives in a
    // scalar register: position = count + prefix population of the ballot, no LDS atomic.
    // The survivor list holds surv_cap entries (half of the largest cell, >= 256), not one per pixel: the LDS slice of a
    // wavefront decides how many cells a compute unit works on at once.  A cell with more survivors than that (noise, dense
    // texture) scores the list whenever the next step might not fit and starts it again; the non-maximum pass then walks the
    // score map instead of the list.  Same scores, same corners, either way.
    int ns = 0;
    bool spilled        = false;
    const int surv_cap = L.f_surv_cap;
    auto flush = [&]()
    {
        __builtin_amdgcn_wave_barrier();
        (void)score_survivors(ns);
        __builtin_amdgcn_wave_barrier();
        ns      = 0;
        spilled = true;
    };
    const int lpy = lane >> 5, lpx = lane & 31;
#ifdef SNK_FAST_PHASE_A_DWORDS
    if constexpr (NQ != 0)
    {
        // Round 6 (the round-5 review's item 1a), built and MEASURED SLOWER -- an experiment build (-DSNK_FAST_PHASE_A_DWORDS), not the
        // default: same-box A/B FAST 1.80 (byte form below) against 1.96 ms (this form), step 192.2 against 187.0 k frames/s
        // (profiles/r06/r06g_ab_fast_phase_a_dwords_negative.txt); bit-exact in the extractor fuzzers (2 836 + 383 cases).  It removes 65
        // of a cell's ~160 LDS instructions and adds ten v_perm plus two more ballot / prefix / branch groups per step: the kernel is
        // bound by vector issue, not by the LDS pipe.  (Also: a step can add up to R x cw survivors, so it needs f_surv_cap >= that --
        // true for the layout's own choice, not for the test's forced 128.)
        // The quick test on whole tile DWORDS.  The byte form below read every operand with
        // ds_read_u8 -- ten LDS instructions per 128 pixels, ~80 of a cell's ~160, in a kernel whose LDS pipe is as busy as its vector
        // ALU (one LDS instruction per three vector ones, SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS 0.48).  Here a lane owns one aligned
        // dword of a tile row = four pixels: it reads that dword with its two neighbours and the dwords three rows up and down (five
        // dword reads, the compiler pairs them into ds_read2_b32: three LDS instructions per ~250 pixels), unpacks to the packed 16-bit
        // form with v_perm_b32 (ten, what the byte reads did for free) and evaluates the same bound
        //   ub = max(min(max(r0, r8), max(r4, r12)) - c, c - max(min(r0, r8), min(r4, r12)))
        // on two register pairs.  Lanes = ncol dword columns x R rows (ncol = the dwords the cell's columns touch, 8..10 for 30-pixel
        // cells; R = 64 / ncol rows per step).  Same survivors (as a set: the order of the list does not matter downstream).
        const int c0   = (3 + sh) >> 2;
        const int ncol = ((3 + sh + cw - 1) >> 2) - c0 + 1;                      // wave-uniform
        const int R    = 64 / ncol;                                              // rows per step (ncol <= 16 for cells up to 52 pixels)
        const int lr   = (lane * ((65536 + ncol - 1) / ncol)) >> 16;             // lane / ncol (exact for lane < 64, ncol <= 16)
        const int lc   = lane - lr * ncol;
        const int px0  = 4 * (c0 + lc) - 3 - sh;                                 // cell column of the dword's first pixel (may be < 0)
        const bool lane_on = lr < R;
        const u32* trow = tile_dw + 3 * TPD + (c0 + lc);                         // the dword of cell row 0
        const int step_max = R * cw;                                             // survivors one step can add
        bool okx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) okx[j] = lane_on && px0 + j >= 0 && px0 + j < cw;
        for (int y0 = 0; y0 < ch; y0 += R)
        {
            if (ns + step_max > surv_cap) flush();  // wave-uniform
            const int py     = y0 + lr;
            const bool rowok = py < ch;
            const u32* q     = trow + min(py, ch - 1) * TPD;                    // rows past the cell are clamped (their results are masked)
            const u32 up = q[-3 * TPD], dn = q[3 * TPD], dm = q[-1], d = q[0], dp = q[1];
            u32 ubw[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
            {
                // pixels 2 hh, 2 hh + 1 of the dword in the 16-bit halves
                const u32 sel  = hh ? 0x0c030c02u : 0x0c010c00u;
                const u32 c    = __builtin_amdgcn_perm(0u, d, sel);
                const u32 r0   = __builtin_amdgcn_perm(0u, dn, sel);
                const u32 r8   = __builtin_amdgcn_perm(0u, up, sel);
                const u32 r4   = __builtin_amdgcn_perm(dp, d, hh ? 0x0c060c05u : 0x0c040c03u);   // bytes j + 3 of {dp : d}
                const u32 r12  = __builtin_amdgcn_perm(d, dm, hh ? 0x0c040c03u : 0x0c020c01u);   // bytes j + 1 of {d : dm}
                const u32 hi   = pku_min(pku_max(r0, r8), pku_max(r4, r12));
                const u32 lo   = pku_max(pku_min(r0, r8), pku_min(r4, r12));
                ubw[hh]        = __builtin_bit_cast(u32, pk_max(pk_sub(hi, c), pk_sub(c, lo)));
            }
            const u32 code0 = ((u32)py << 6) + (u32)px0;  // (py << 6) | px of pixel 0 when it is valid; pixel j: + j
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                const u32 w   = ubw[j >> 1];
                const bool sv = (j & 1) ? (int)w > ((min_th << 16) | 0xFFFF) : (int)(short)(w & 0xFFFFu) > min_th;
                const bool on = sv && okx[j] && rowok;
                const u64 m   = __builtin_amdgcn_ballot_w64(on);
                if (on) surv[ns + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u))] = (u16)(code0 + (u32)j);
                ns += __popcll(m);
            }
        }
        (void)lpy;
        (void)lpx;
    }
    else
#endif
    if constexpr (NQ != 0)
    {
        // Compile-time pitch: FOUR rows of 32 pixels per step, two pixels per lane in the 16-bit halves of a register (rows
        // py0 + lpy and py0 + lpy + 2), the bound evaluated on the raw bytes with packed 16-bit min / max:
        //   ub = max(min(max(r0, r8), max(r4, r12)) - c, c - max(min(r0, r8), min(r4, r12)))
        // 5 packs + 9 packed operations + 2 compares per 128 pixels (the one-pixel form: 14 per 64).
        for (int px0 = 0; px0 < cw; px0 += 32)
        {
            const int px   = px0 + lpx;
            const u8* t    = tile + (lpy + 3) * TP + px + 3 + sh;
            u32 code       = ((u32)lpy << 6) | (u32)px;  // (py << 6) | px of the low half; the high half is two rows below
            const bool inx = px < cw;
            const u64 mx   = __builtin_amdgcn_ballot_w64(inx);
            auto step = [&](bool ok_lo, u64 m_lo, bool ok_hi, u64 m_hi)
            {
                const u32 c   = (u32)t[0] | ((u32)t[2 * TP] << 16);
                const u32 r0  = (u32)t[3 * TP] | ((u32)t[5 * TP] << 16);
                const u32 r8  = (u32)t[-3 * TP] | ((u32)t[-TP] << 16);
                const u32 r4  = (u32)t[3] | ((u32)t[2 * TP + 3] << 16);
                const u32 r12 = (u32)t[-3] | ((u32)t[2 * TP - 3] << 16);
                const u32 hi  = pku_min(pku_max(r0, r8), pku_max(r4, r12));
                const u32 lo  = pku_max(pku_min(r0, r8), pku_min(r4, r12));
                const short2_t ub = pk_max(pk_sub(hi, c), pk_sub(c, lo));
                const bool s_lo = (int)ub.x > min_th;
                const bool s_hi = __builtin_bit_cast(int, ub) > ((min_th << 16) | 0xFFFF);  // high half > min_th
                const u64 ml = m_lo & __builtin_amdgcn_ballot_w64(s_lo);
                if (ok_lo && s_lo) surv[ns + __builtin_amdgcn_mbcnt_hi((u32)(ml >> 32), __builtin_amdgcn_mbcnt_lo((u32)ml, 0u))] = (u16)code;
                ns += __popcll(ml);
                const u64 mh = m_hi & __builtin_amdgcn_ballot_w64(s_hi);
                if (ok_hi && s_hi)
                    surv[ns + __builtin_amdgcn_mbcnt_hi((u32)(mh >> 32), __builtin_amdgcn_mbcnt_lo((u32)mh, 0u))] = (u16)(code + 128u);
                ns += __popcll(mh);
            };
            int py0 = 0;
            for (; py0 + 4 <= ch; py0 += 4, t += 4 * TP, code += 256u)
            {
                if (ns + 128 > surv_cap) flush();  // wave-uniform
                step(inx, mx, inx, mx);
            }
            const int rem = ch - py0;  // 0..3 rows left: low halves hold rows py0 + lpy, high halves py0 + lpy + 2
            if (rem > 0)
            {
                if (ns + 128 > surv_cap) flush();
                const u64 lanes_lo = rem >= 2 ? ~0ull : 0xFFFFFFFFull, lanes_hi = rem >= 3 ? 0xFFFFFFFFull : 0ull;
                step(inx && lpy < rem, mx & lanes_lo, inx && lpy + 2 < rem, mx & lanes_hi);
            }
        }
    }
    else
    {
        for (int px0 = 0; px0 < cw; px0 += 32)
        {
            // two rows of 32 pixels per step; a lane walks down its column: address and (py << 6 | px) code advance by
            // constants, "inside the cell" is one compare of the code and a loop-invariant column mask.  Rows
            // below / columns right of the cell read whatever follows in the wavefront's slice; they are masked.
            const int px  = px0 + lpx;
            const u8* t   = tile + (lpy + 3) * TP + px + 3 + sh;
            u32 code      = ((u32)lpy << 6) | (u32)px;  // (py << 6) | px
            const bool inx = px < cw;
            const u64 mx   = __builtin_amdgcn_ballot_w64(inx);
            const u64 mlo  = mx & 0xFFFFFFFFull;  // odd cell height: the last step only has its first row
            auto step = [&](bool row_ok, u64 mrow)
            {
                const int v  = t[0];
                const int d0 = t[3 * TP] - v, d8 = t[-3 * TP] - v, d4 = t[3] - v, d12 = t[-3] - v;
                const int ub_b = min(max(d0, d8), max(d4, d12));
                const int ub_d = min(max(-d0, -d8), max(-d4, -d12));
                const int ub   = max(ub_b, ub_d);
                const bool sv  = row_ok && ub > min_th;
                // ballot of the compare ANDed with the loop-invariant mask in scalar registers (a ballot of `sv` costs two
                // more vector instructions per step)
                const u64 m = mrow & __builtin_amdgcn_ballot_w64(ub > min_th);
                if (sv) surv[ns + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u))] = (u16)code;
                ns += __popcll(m);
            };
            int py0 = 0;
            for (; py0 + 2 <= ch; py0 += 2, t += 2 * TP, code += 128u)
            {
                if (ns + 64 > surv_cap) flush();  // wave-uniform
                step(inx, mx);
            }
            if (py0 < ch)
            {
                if (ns + 64 > surv_cap) flush();
                step(inx && lpy == 0, mlo);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (dbg_stop == 2) return;  // after the quick test

    const int ns2 = score_survivors(ns);  // corners' positions are now surv[0 .. ns2)
    __builtin_amdgcn_wave_barrier();

    if (dbg_stop == 3) return;  // after the exact scores
    // 3x3 non-max suppression (strict) among scores above min_th (all of them are survivors); the list of
    // corners is appended the same way (wave-uniform loop, ballot prefix, counts in scalar registers)
    int nl = 0, nini = 0;
    const int n_items = spilled ? ch * 64 : ns2;  // spilled: every (row, lane = column) of the cell
    for (int j0 = 0; j0 < n_items; j0 += 64)
    {
        const int j  = j0 + lane;
        int e;
        bool valid;
        if (spilled)
        {
            e     = j0 | min(lane, cw - 1);  // (py << 6) | px
            valid = lane < cw;
        }
        else
        {
            e     = surv[min(j, ns2 - 1)];
            valid = j < ns2;
        }
        const int px = e & 63, py = e >> 6;
        const u8* s  = &S[(py + 1) * SP + px + 1];
        // all nine reads in one go (a short-circuit chain is nine dependent LDS round trips under divergent branches)
        const int v = s[0], n0 = s[-1], n1 = s[1], n2 = s[-SP - 1], n3 = s[-SP], n4 = s[-SP + 1], n5 = s[SP - 1], n6 = s[SP],
                  n7 = s[SP + 1];
        const int nmax  = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
        const bool keep = valid && v > min_th && v > nmax;
        const u64 m = __builtin_amdgcn_ballot_w64(keep);
        // strength key: higher score first, then smaller y, then smaller x
        if (keep)
            list[nl + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u))] =
                ((u32)v << 12) | ((u32)(63 - py) << 6) | (u32)(63 - px);
        nl += __popcll(m);
        nini += __popcll(__builtin_amdgcn_ballot_w64(keep && v > ini_th));
    }
    __builtin_amdgcn_wave_barrier();
    if (dbg_stop == 4) return;  // after the non-maximum suppression
    const bool ini = nini > 0;
    const u32 thr  = ((u32)(ini ? ini_th : min_th) << 12) | 0xFFFu;  // key > thr  <=>  score > threshold
    const int n    = ini ? nini : nl;
    u32* out = cand + cell_index * CELL_SLOTS;
    if (n <= CELL_SLOTS)
    {
        // The usual cell: every candidate of the effective threshold has a slot, so the slots need no order -- the order only matters
        // when somebody has to CUT a cell to its k strongest, i.e. when the level holds more candidates than its budget; distribute_kernel
        // ranks such cells itself (rank_cut_cells, noise-like images only).  Compaction by ballot prefix instead of the all-pairs rank
        // (~70 of the ~570 vector instructions of a cell).
        int at = 0;
        for (int i0 = 0; i0 < nl; i0 += 64)  // nl may exceed 64 when the ini threshold applies (the list holds everything above min_th)
        {
            const int i    = i0 + lane;
            const u32 k    = list[min(i, nl - 1)];
            const bool put = i < nl && k > thr;
            const u64 m    = __builtin_amdgcn_ballot_w64(put);
            if (put) out[at + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u))] = k;
            at += __popcll(m);
        }
    }
    else
    {
    // more candidates than slots: rank them by strength, keep the CELL_SLOTS strongest (in that order)
    for (int i = lane; i < nl; i += 64)
    {
        const u32 k = list[i];
        if (k <= thr) continue;
        int r = 0;
        int j = 0;
        for (; j + 4 <= nl; j += 4)  // four broadcast reads in flight
        {
            const u32 a0 = list[j], a1 = list[j + 1], a2 = list[j + 2], a3 = list[j + 3];
            r += (a0 > k ? 1 : 0) + (a1 > k ? 1 : 0) + (a2 > k ? 1 : 0) + (a3 > k ? 1 : 0);
        }
        for (; j < nl; ++j) r += list[j] > k ? 1 : 0;
        if (r < CELL_SLOTS) out[r] = k;
    }
    }
    if (lane == 0) cell_cnt[cell_index] = (u16)(n > 65535 ? 65535 : n);
    }
}

static inline int fast_quads(const Layout& L)
{
    return L.f_tile_pitch_dw == 12 && L.f_s_pitch == 40 ? 3 : (L.f_tile_pitch_dw == 16 && L.f_s_pitch == 64 ? 4 : 0);
}
static inline const void* fast_kernel_for(const Layout& L, bool loop)
{
    const int fq = fast_quads(L);
    if (loop)
        return fq == 3 ? reinterpret_cast<const void*>(fast_kernel<3, true>)
                       : (fq == 4 ? reinterpret_cast<const void*>(fast_kernel<4, true>) : reinterpret_cast<const void*>(fast_kernel<0, true>));
    return fq == 3 ? reinterpret_cast<const void*>(fast_kernel<3, false>)
                   : (fq == 4 ? reinterpret_cast<const void*>(fast_kernel<4, false>) : reinterpret_cast<const void*>(fast_kernel<0, false>));
}

// ------------------------------------------------------------------------------------------------
// Harris response of the FAST candidates ("orb.response" = 1; include/snake_hip.h).  One wavefront per FAST
// cell, one lane per candidate slot: the 9 x 9 pixels around the corner come from the level image (L2-hot: fast_kernel has just
// read the cell) as three aligned dwords per row, the 3 x 3 Sobel derivatives over the 7 x 7 block are separable integer sums, the
// response is OpenCV's float expression with every operation rounded in its written order (the file is compiled with
// -ffp-contract=off).  Written as an order-preserving uint32 next to the candidate; distribute_kernel ranks with it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 harris_rank(float r)
{
    const u32 u = __builtin_bit_cast(u32, r);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float harris_unrank(u32 k)
{
    return __builtin_bit_cast(float, (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__global__ __launch_bounds__(256) void harris_kernel(Layout L, const u8* __restrict__ img0, int pitch0, long long stride0, int aligned0,
                                                     const u32* __restrict__ cand, const u16* __restrict__ cell_cnt,
                                                     u32* __restrict__ cand_h, int gx, int batch)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int b, bxi;
    if (!xcd_image_map(gx, batch, b, bxi)) return;
    const int cid = bxi * 4 + wave;
    if (cid >= L.total_cells) return;
    const long long cell_index = (long long)b * L.total_cells + cid;
    const int cnt = min((int)cell_cnt[cell_index], CELL_SLOTS);
    if (lane >= cnt) return;
    // The cell's entry by a SCALAR load (the index is wave-uniform; as a vector load it was an L2 round trip per cell in front of
    // everything else), and the five fields of the level the cell needs -- w, h, pitch, img_stride, base: the first 32 bytes of
    // LevelInfo -- by ONE scalar load (read field by field they were four dependent scalar-load round trips per cell).
    int4 ct;
    {
        const int4* cp = L.cell_tab + cid;
        asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ct) : "s"(cp) : "memory");
    }
    const int l   = ct.w;
    const LevelHead lv = *reinterpret_cast<const LevelHead*>(&L.lv[l]);
    const u32 k   = cand[cell_index * CELL_SLOTS + lane];
    const int x   = ct.x + 63 - (int)(k & 63u), y = ct.y + 63 - (int)((k >> 6) & 63u);
    const u8* src      = l == 0 ? img0 + (long long)b * stride0 : lv.base + (long long)b * lv.img_stride;
    const int pitch    = l == 0 ? pitch0 : lv.pitch;
    const bool aligned = l == 0 ? aligned0 != 0 : true;
    // rows y - 4 .. y + 4, columns x - 4 .. x + 4 (corners lie >= 19 px inside the level): rw[r][j] = bytes 4 j .. 4 j + 3 of the row
    u32 rw[9][3];
    const int xa = (x - 4) & ~3, sh = (x - 4) & 3;
#pragma unroll
    for (int r = 0; r < 9; ++r)
    {
        const u8* rp = src + (long long)(y - 4 + r) * pitch;
        if (aligned)
        {
            const u32 d0 = *reinterpret_cast<const u32*>(rp + xa), d1 = *reinterpret_cast<const u32*>(rp + xa + 4),
                      d2 = *reinterpret_cast<const u32*>(rp + xa + 8);
            // the funnel shift takes its byte count from a register: sh is per lane
            const u64 q01 = ((u64)d1 << 32) | d0, q12 = ((u64)d2 << 32) | d1;
            rw[r][0] = (u32)(q01 >> (8 * sh));
            rw[r][1] = (u32)(q12 >> (8 * sh));
            rw[r][2] = d2 >> (8 * sh);
        }
        else
        {
#pragma unroll
            for (int j = 0; j < 3; ++j)
            {
                u32 v = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * j + e < 9) v |= (u32)rp[x - 4 + 4 * j + e] << (8 * e);
                rw[r][j] = v;
            }
        }
    }
    auto px = [&](int r, int c) -> int { return (int)((rw[r][c >> 2] >> (8 * (c & 3))) & 0xFFu); };
    // separable Sobel: dxr = p[c + 1] - p[c - 1], sxr = p[c - 1] + 2 p[c] + p[c + 1] per row; Ix = dxr(r - 1) + 2 dxr(r) + dxr(r + 1),
    // Iy = sxr(r + 1) - sxr(r - 1) -- the integers of the written 3 x 3 form
    int a = 0, bb = 0, c = 0;
#pragma unroll
    for (int cc = 1; cc <= 7; ++cc)
    {
        int dxr[9], sxr[9];
#pragma unroll
        for (int r = 0; r < 9; ++r)
        {
            const int p0 = px(r, cc - 1), p1 = px(r, cc), p2 = px(r, cc + 1);
            dxr[r] = p2 - p0;
            sxr[r] = p0 + 2 * p1 + p2;
        }
#pragma unroll
        for (int r = 1; r <= 7; ++r)
        {
            const int Ix = dxr[r - 1] + 2 * dxr[r] + dxr[r + 1];
            const int Iy = sxr[r + 1] - sxr[r - 1];
            a += Ix * Ix;
            bb += Iy * Iy;
            c += Ix * Iy;
        }
    }
    const float fa = (float)a, fb = (float)bb, fc = (float)c;
    const float s4 = 0x1.bb9da2p-52f;  // (1 / (4 * 7 * 255))^4 in float, multiplied left to right
    const float det = fa * fb - fc * fc;
    const float tr  = fa + fb;
    const float kt  = 0.04f * tr * tr;
    cand_h[cell_index * CELL_SLOTS + lane] = harris_rank((det - kt) * s4);
}

// The same response with the candidates of HARRIS_NC consecutive cells packed densely into the lanes (round 6).  A FAST cell of the
// benchmark images holds ~6 candidates (3 700 per image over ~650 cells): one wavefront per cell with one lane per slot ran its ~630
// vector instructions for a tenth of its lanes, and 27 scattered dword loads per lane -- 3.2 ms per 2048 images, the longest kernel of
// a batch under "orb.response" = 1.  Here a wavefront takes 16 cells: the counts of the cells are scanned across lanes 0..15, lane i
// of a round finds (cell, slot) of the group's i-th candidate in the scan, the level of its cell by a select over the levels (the
// cells of a group may lie on two levels), and reads each of the 9 rows with ONE 12-byte load.  Same integers, same float expression.
constexpr int HARRIS_NC = 16;
typedef u32 u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
__global__ __launch_bounds__(256) void harris_dense_kernel(Layout L, const u8* __restrict__ img0, int pitch0, long long stride0, int aligned0,
                                                           const u32* __restrict__ cand, const u16* __restrict__ cell_cnt,
                                                           u32* __restrict__ cand_h, int gx, int batch)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int b, bxi;
    if (!xcd_image_map(gx, batch, b, bxi)) return;
    const int cid0 = (bxi * 4 + wave) * HARRIS_NC;
    if (cid0 >= L.total_cells) return;
    const int ncell         = min(HARRIS_NC, L.total_cells - cid0);
    const long long cell0   = (long long)b * L.total_cells + cid0;
    const int cnt           = lane < ncell ? min((int)cell_cnt[cell0 + lane], CELL_SLOTS) : 0;
    int inc = cnt;  // inclusive scan over lanes 0 .. 15 (the other lanes hold 0 and are not read)
#pragma unroll
    for (int d = 1; d < HARRIS_NC; d <<= 1)
    {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    const int total = __builtin_amdgcn_readlane(inc, HARRIS_NC - 1);
    const int exc   = inc - cnt;
    for (int base = 0; base < total; base += 64)  // wave-uniform rounds
    {
        const int i    = base + lane;
        const bool act = i < total;
        const int ii   = act ? i : total - 1;
        int j = 0;  // the candidate's cell: how many cells end at or before it
#pragma unroll
        for (int t = 0; t < HARRIS_NC - 1; ++t) j += __builtin_amdgcn_readlane(inc, t) <= ii ? 1 : 0;
        const int slot = ii - __shfl(exc, j);
        const int4 ct  = L.cell_tab[cid0 + j];
        const int l    = ct.w;
        const long long cell_index = cell0 + j;
        const u32 k   = cand[cell_index * CELL_SLOTS + slot];
        const int x   = ct.x + 63 - (int)(k & 63u), y = ct.y + 63 - (int)((k >> 6) & 63u);
        const u8* src = img0 + (long long)b * stride0;
        int pitch     = pitch0;
        for (int q = 1; q < L.n_levels; ++q)  // wave-uniform loop, per-lane select
        {
            const LevelHead lv = *reinterpret_cast<const LevelHead*>(&L.lv[q]);
            if (l == q)
            {
                src   = lv.base + (long long)b * lv.img_stride;
                pitch = lv.pitch;
            }
        }
        const bool aligned = l == 0 ? aligned0 != 0 : true;
        // rows y - 4 .. y + 4, columns x - 4 .. x + 4 (corners lie >= 19 px inside the level): rw[r][j] = bytes 4 j .. 4 j + 3 of the row
        u32 rw[9][3];
        const int xa = (x - 4) & ~3, sh = (x - 4) & 3;
        const u8* rp0 = src + (long long)(y - 4) * pitch;
        if (__builtin_amdgcn_ballot_w64(!aligned) == 0)
        {
            u32x3_a4 d[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) d[r] = *reinterpret_cast<const u32x3_a4*>(rp0 + r * pitch + xa);
#pragma unroll
            for (int r = 0; r < 9; ++r)
            {
                const u64 q01 = ((u64)d[r].y << 32) | d[r].x, q12 = ((u64)d[r].z << 32) | d[r].y;
                rw[r][0] = (u32)(q01 >> (8 * sh));
                rw[r][1] = (u32)(q12 >> (8 * sh));
                rw[r][2] = d[r].z >> (8 * sh);
            }
        }
        else
        {
#pragma unroll
            for (int r = 0; r < 9; ++r)
#pragma unroll
                for (int jj = 0; jj < 3; ++jj)
                {
                    u32 v = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * jj + e < 9) v |= (u32)rp0[r * pitch + x - 4 + 4 * jj + e] << (8 * e);
                    rw[r][jj] = v;
                }
        }
        auto px = [&](int r, int c) -> int { return (int)((rw[r][c >> 2] >> (8 * (c & 3))) & 0xFFu); };
        int a = 0, bb = 0, c = 0;
#pragma unroll
        for (int cc = 1; cc <= 7; ++cc)
        {
            int dxr[9], sxr[9];
#pragma unroll
            for (int r = 0; r < 9; ++r)
            {
                const int p0 = px(r, cc - 1), p1 = px(r, cc), p2 = px(r, cc + 1);
                dxr[r] = p2 - p0;
                sxr[r] = p0 + 2 * p1 + p2;
            }
#pragma unroll
            for (int r = 1; r <= 7; ++r)
            {
                const int Ix = dxr[r - 1] + 2 * dxr[r] + dxr[r + 1];
                const int Iy = sxr[r + 1] - sxr[r - 1];
                a += Ix * Ix;
                bb += Iy * Iy;
                c += Ix * Iy;
            }
        }
        const float fa = (float)a, fb = (float)bb, fc = (float)c;
        const float s4 = 0x1.bb9da2p-52f;  // (1 / (4 * 7 * 255))^4 in float, multiplied left to right
        const float det = fa * fb - fc * fc;
        const float tr  = fa + fb;
        const float kt  = 0.04f * tr * tr;
        if (act) cand_h[cell_index * CELL_SLOTS + slot] = harris_rank((det - kt) * s4);
    }
}

// ------------------------------------------------------------------------------------------------
// 7x7 Gaussian {18,33,49,56,49,33,18}/256 of every level (what the descriptors sample), streaming:
// one WAVEFRONT walks down a column strip of 62 x 4 output pixels (lane = one aligned dword per row,
// lanes 0 / 63 are the 3-pixel halo) through a band of 64 rows with the 7-row window held in
// registers — no LDS staging, no halo re-reads inside the band, one coalesced 256-byte load and one
// 248-byte store per row.  Per row: neighbour dwords by lane shuffle, horizontal pass with
// v_alignbyte + v_dot4_u32_u8 (exact 16-bit rows), vertical pass over the register window, one rounding.
// ------------------------------------------------------------------------------------------------
constexpr int SM_BH        = 64;         // output rows per band (big launches); BH + 6 rows are streamed per band, a multiple of 14
                                          // so that the 7-row window index stays static.  Small launches (the per-frame calls: one or two
                                          // images) use bands of 22 or 8 rows: a wavefront walks its band row by row behind a chain of
                                          // load latencies, and a 752x480 level with 64-row bands is 32 wavefronts of 70 rows each --
                                          // 22 us per level on an otherwise empty chip (profiles/r04/r04j_*); same arithmetic per pixel.
constexpr int SM_LANES_OUT = 61;  // strip <= 244 columns: 8 loaded columns remain to the right for the down-scale taps

// The same pass also produces level l+1 (make_next): whenever the stream holds source rows sy, sy+1
// of a destination row, the two raw rows go to a 512-byte LDS buffer of the wavefront and every lane
// interpolates 4 destination pixels (its column taps / weights are loop-invariant registers).  Each
// level is therefore read from HBM exactly once for blur + pyramid.
__device__ __forceinline__ u32 wave_shr1(u32 v) { return __builtin_amdgcn_mov_dpp(v, 0x138, 0xf, 0xf, true); }  // lane i <- i-1
__device__ __forceinline__ u32 wave_shl1(u32 v) { return __builtin_amdgcn_mov_dpp(v, 0x130, 0xf, 0xf, true); }  // lane i <- i+1
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 dot2(u32 pair, u32 w, u32 acc)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pair), __builtin_bit_cast(u16x2, w), acc, false);
}

// ALIGNED = false: level 0 of an input whose base / pitch is not a multiple of 4 (byte loads; its own instantiation so
// that the usual one does not carry the byte-column registers).
#ifndef SNK_LEVEL_MIN_WAVES
#define SNK_LEVEL_MIN_WAVES 1  // build-time A/B: wavefronts per SIMD the register allocator must leave room for in level_kernel
#endif
template <bool ALIGNED, int BH>
__global__ __launch_bounds__(256, SNK_LEVEL_MIN_WAVES) void level_kernel(Layout L, int l, const u8* __restrict__ img0, int pitch0, long long stride0,
                                                    int make_next, int gx, int batch, int n_bands,
                                                    int blur_mode /* 0: the blurred level row-major, 2: tiled 32 x 4 (describe_kernel's LDS-DMA form reads that),
                                                                     1: not written at all (SNK_ORB_BLUR_IN_DESCRIBE=1, experiment) */)
{
    constexpr int SM_ROWS = BH + 6;
    static_assert(BH <= 64, "lane r of the wavefront keeps the row map of source row yb0 + r");
    __shared__ u32 rowbuf[4][64];  // one raw row per wavefront (down-scale taps)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int b, bxi;
    if (!xcd_image_map(gx, batch, b, bxi)) return;
    const LevelInfo& lv = L.lv[l];
    const int u    = bxi * 4 + wave;
    if (u >= lv.n_strips * n_bands) return;  // whole wavefront
    const int band  = u / lv.n_strips, strip = u - band * lv.n_strips;
    const int sx0   = strip * lv.strip_stride;
    const int sx1   = strip == lv.n_strips - 1 ? lv.w : sx0 + lv.strip_stride;  // the last strip takes the rest of the row (<= 244 columns)
    const int xl    = sx0 - 4 + 4 * lane;  // image column of this lane's dword
    const int yb0   = band * BH, yb1 = min(yb0 + BH, lv.h);
    const bool out_lane = lane >= 1 && lane <= SM_LANES_OUT && xl < (strip == lv.n_strips - 1 ? lv.store_end : sx1);
    const u8* src      = l == 0 ? img0 + (long long)b * stride0 : lv.base + (long long)b * lv.img_stride;
    // 32-bit offsets inside one image (pitch * h < 2^31), scalars pinned in SGPRs
    const int pitch    = __builtin_amdgcn_readfirstlane(l == 0 ? pitch0 : lv.pitch);
    const int bpitch   = __builtin_amdgcn_readfirstlane(lv.pitch);
    constexpr bool aligned = ALIGNED;
    const bool col_ok  = aligned && xl >= 0 && xl + 3 < lv.w;
    // aligned dword inside the row pitch; the dword holding column w - 1 may reach up to 3 bytes into the row padding
    const int xsafe    = min(max(xl, 0), (lv.w - 1) & ~3);
    // lanes right of the image only matter up to column w + 2 (halo of the last pixel); clamping keeps
    // reflect101 inside its domain for strips much narrower than the wavefront (w < 126)
    const int xr0 = reflect101(min(xl, lv.w + 2), lv.w), xr1 = reflect101(min(xl + 1, lv.w + 2), lv.w),
              xr2 = reflect101(min(xl + 2, lv.w + 2), lv.w), xr3 = reflect101(min(xl + 3, lv.w + 2), lv.w);
    // Border lanes (left halo of the first strip, lanes at / right of column w - 1) need reflect101 pixels.  Those
    // are bytes of at most two other lanes' dwords of the same row: two ds_bpermute pulls and a byte select per row
    // (per-lane source lanes / selector fixed for the whole band; interior lanes pull themselves) instead of
    // dependent byte loads from global memory.
    int pull_a = 4 * lane, pull_b = 4 * lane;
    u32 pull_sel = 0x03020100u;
    if (aligned && !col_ok)
    {
        const int org = sx0 - 4;
        const int l0 = (xr0 - org) >> 2, l1 = (xr1 - org) >> 2, l2 = (xr2 - org) >> 2, l3 = (xr3 - org) >> 2;
        const int la = min(min(l0, l1), min(l2, l3)), lb = max(max(l0, l1), max(l2, l3));
        pull_a   = 4 * la;
        pull_b   = 4 * lb;
        pull_sel = (u32)((l0 == la ? 0 : 4) + (xr0 & 3)) | ((u32)((l1 == la ? 0 : 4) + (xr1 & 3)) << 8) |
                   ((u32)((l2 == la ? 0 : 4) + (xr2 & 3)) << 16) | ((u32)((l3 == la ? 0 : 4) + (xr3 & 3)) << 24);
    }
    const bool border_strip = aligned && __any(!col_ok);  // wave-uniform
    u8* blur = lv.blur + (long long)b * lv.blur_stride;
    const bool no_blur_store = blur_mode == 1, blur_tiled = blur_mode == 2;
    const u32 xl_store       = blur_tiled ? blur_tiled_x(max(xl, 0)) : (u32)xl;  // the lane's column part of its store address (loop-invariant)
    const u32 W0123 = 18u | (33u << 8) | (49u << 16) | (56u << 24);
    const u32 W456  = 49u | (33u << 8) | (18u << 16);

    // ---- loop-invariant set-up of the down-scale.  A strip owns the destination dwords whose first
    // pixel has its source column inside the strip; lane i produces dword g0 + i.  The taps of the other
    // three pixels reach at most 3 * scale + 1 columns further, inside the 8 extra loaded columns
    // (scale <= 2).  Dwords are written whole: columns >= w of the last one land in the row padding.
    const LevelInfo& nx = L.lv[make_next ? l + 1 : l];
    u8* nbase           = make_next ? nx.base + (long long)b * nx.img_stride : nullptr;
    const int npitch    = __builtin_amdgcn_readfirstlane(nx.pitch);
    const u8* rbytes    = reinterpret_cast<const u8*>(&rowbuf[0][0]);
    int so[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    u32 wx[4] = {0, 0, 0, 0};  // (2048 - w1) | w1 << 16
    bool own  = false;
    int xdst  = 0;
    if (make_next)
    {
        const int g = lv.strip_dx[strip] + lane;
        own         = g < lv.strip_dx[strip + 1];
        xdst        = 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const int x  = min(xdst + j, nx.w - 1);
            const int sx = own ? nx.xofs[x] : sx0;
            const u32 w1 = own ? (u32)nx.xw1[x] : 0u;
            so[j]        = 256 * wave + sx - (sx0 - 4);
            s1[j]        = 256 * wave + min(sx + 1, lv.w - 1) - (sx0 - 4);
            wx[j]        = (2048u - w1) | (w1 << 16);
        }
    }
    // horizontally interpolated rows + 1024 (<= 255 * 2048 + 1024 < 2^24), slot = row parity in the stream
    u32 ah[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    // lane r keeps the destination row / y weight of source row yb0 + r (read back with v_readlane:
    // dependent scalar loads inside the row loop would serialise it)
    int dy_tab = -1, wy_tab = 0;
    if (make_next && yb0 + lane < yb1)
    {
        dy_tab = lv.ymap[yb0 + lane];
        if (dy_tab >= 0) wy_tab = nx.yw1[dy_tab];
    }
    // The two tables come from global loads issued before the row loop.  Inside the loop the compiler can no longer
    // tell how many memory operations were issued after them and guards every v_readlane of them with
    // s_waitcnt vmcnt(0) -- which also waits for all row loads in flight and for the previous rows' stores.
    // Passing them through an empty asm makes them plain register values (one wait here, none in the loop).
    asm volatile("" : "+v"(dy_tab), "+v"(wy_tab));

    // vertical window: pr[s][c] = H(row a) | H(row a + 1) << 16 of column c, written in slot (a + 1 - (yb0 - 3)) % 7
    u32 pr[7][4];
#pragma unroll
    for (int k = 0; k < 7; ++k) pr[k][0] = pr[k][1] = pr[k][2] = pr[k][3] = 0;

    static_assert(SM_ROWS % 14 == 0, "two 7-row blocks per iteration keep the row-parity slots static");
    // Rolling prefetch: row k + 7 is requested into the register row k just left (branch-free: every lane loads an aligned
    // dword from a clamped column, rows past the band are clamped too; lanes on the image border are patched per row, border
    // strips only), so seven row loads are ALWAYS in flight -- the first version requested a block of seven rows, waited for
    // them, worked through them and only then asked for the next seven.
    auto row_load = [&](int k) -> u32
    {
        const int y = min(yb0 - 3 + k, lv.h + 2);
        if (aligned) return *reinterpret_cast<const u32*>(src + (u32)(reflect101(y, lv.h) * pitch) + (u32)xsafe);
        const u8* rp = src + reflect101(y, lv.h) * pitch;
        return (u32)rp[xr0] | ((u32)rp[xr1] << 8) | ((u32)rp[xr2] << 16) | ((u32)rp[xr3] << 24);
    };
    u32 dn[7];
#pragma unroll
    for (int kk = 0; kk < 7; ++kk) dn[kk] = row_load(kk);
    for (int k00 = 0; k00 < SM_ROWS; k00 += 14)
#pragma unroll
    for (int half = 0; half < 2; ++half)
    {
        const int k0 = k00 + 7 * half;
#pragma unroll
        for (int kk = 0; kk < 7; ++kk)
        {
            const int k = k0 + kk;
            const int y = yb0 - 3 + k;
            u32 d  = dn[kk];
            dn[kk] = row_load(min(k + 7, SM_ROWS - 1));
            if (y <= yb1 + 2)  // wave-uniform
            {
                if (border_strip)
                {
                    const u32 pa = (u32)__builtin_amdgcn_ds_bpermute(pull_a, (int)d), pb = (u32)__builtin_amdgcn_ds_bpermute(pull_b, (int)d);
                    d            = __builtin_amdgcn_perm(pb, pa, pull_sel);
                }
                // ---- next pyramid level.  Row y is interpolated horizontally once (ac, v_dot2 of the
                // tap pair with the weight pair); a destination row whose source rows are (y-1, y)
                // combines it with the previous row's (ap).  With the y weights pre-multiplied by 4
                // (sum 2^13) the +1024 carried by ap / ac becomes the rounding term 2^23 and the pixel is
                // byte 3 of the sum (<= 255 * 2^24 + 2^23).
                if (make_next && y >= yb0 && y <= yb1)  // wave-uniform
                {
                    rowbuf[wave][lane] = d;
                    __builtin_amdgcn_wave_barrier();
                    u32* ac       = ah[(half + kk) & 1];
                    const u32* ap = ah[(half + kk + 1) & 1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) ac[j] = dot2((u32)rbytes[so[j]] | ((u32)rbytes[s1[j]] << 16), wx[j], 1024u);
                    __builtin_amdgcn_wave_barrier();
                    const int dy = y > yb0 ? __builtin_amdgcn_readlane(dy_tab, y - 1 - yb0) : -1;
                    if (dy >= 0)  // wave-uniform
                    {
                        const u32 wy1 = 4u * (u32)__builtin_amdgcn_readlane(wy_tab, y - 1 - yb0), wy0 = 8192u - wy1;
                        u32 v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = __umul24(ap[j], wy0) + __umul24(ac[j], wy1);
                        const u32 lo = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0703u);  // byte 3 of v0, v1
                        const u32 hi = __builtin_amdgcn_perm(v[3], v[2], 0x07030c0cu);  // byte 3 of v2, v3 in the top half
                        if (own) *reinterpret_cast<u32*>(nbase + (u32)(dy * npitch) + (u32)xdst) = lo | hi;
                    }
                }
                // ---- horizontal pass: stream bytes [dl | d | dr], output j centred on byte 4 + j ----
                const u32 dl = wave_shr1(d), dr = wave_shl1(d);
                const u32 q0 = __builtin_amdgcn_alignbyte(d, dl, 1), q1 = __builtin_amdgcn_alignbyte(d, dl, 2),
                          q2 = __builtin_amdgcn_alignbyte(d, dl, 3);
                const u32 r0 = __builtin_amdgcn_alignbyte(dr, d, 1), r1 = __builtin_amdgcn_alignbyte(dr, d, 2),
                          r2 = __builtin_amdgcn_alignbyte(dr, d, 3);
                u32 hc[4];
                hc[0] = __builtin_amdgcn_udot4(q0, W0123, __builtin_amdgcn_udot4(r0, W456, 0u, false), false);
                hc[1] = __builtin_amdgcn_udot4(q1, W0123, __builtin_amdgcn_udot4(r1, W456, 0u, false), false);
                hc[2] = __builtin_amdgcn_udot4(q2, W0123, __builtin_amdgcn_udot4(r2, W456, 0u, false), false);
                hc[3] = __builtin_amdgcn_udot4(d, W0123, __builtin_amdgcn_udot4(dr, W456, 0u, false), false);
                // ---- vertical pass of output row yo = y - 3: rows yo-3 .. yo+3 = pairs (y-6,y-5) (y-4,y-3)
                // (y-2,y-1) from slots kk-5, kk-3, kk-1 and the row just computed; byte 2 of the rounded
                // sum (<= 256 * 65280 + 32768 < 2^24) is the pixel ----
                const int yo = y - 3;
                if (k >= 6 && yo < yb1)
                {
                    u32 a[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                    {
                        u32 t = __umul24(hc[c], 18u) + 32768u;
                        t     = dot2(pr[(kk + 6) % 7][c], 49u | (33u << 16), t);
                        t     = dot2(pr[(kk + 4) % 7][c], 49u | (56u << 16), t);
                        a[c]  = dot2(pr[(kk + 2) % 7][c], 18u | (33u << 16), t);
                    }
                    if (out_lane && !no_blur_store)
                    {
                        const u32 lo = __builtin_amdgcn_perm(a[1], a[0], 0x0c0c0602u);
                        const u32 hi = __builtin_amdgcn_perm(a[3], a[2], 0x06020c0cu);
                        *reinterpret_cast<u32*>(blur + (blur_tiled ? blur_tiled_y(yo, bpitch) : (u32)(yo * bpitch)) + xl_store) = lo | hi;
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) pr[kk][c] = __builtin_amdgcn_alignbit(hc[c], pr[(kk + 6) % 7][c], 16);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// quadtree distribution
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 point_key(int x, int y, int W, int H, int nroots)
{
    // x, W < 2^16 and nroots <= 255: every product below stays under 2^24, 32-bit division suffices (the 64-bit one is
    // ~5 times the instructions and this runs once per candidate)
    const u32 ux = (u32)x, uw = (u32)W, un = (u32)nroots;
    const int root = (int)((ux * un) / uw);
    int x0 = (int)(((u32)root * uw + un - 1u) / un), x1 = (int)(((u32)(root + 1) * uw + un - 1u) / un);
    int y0 = 0, y1 = H;
    u64 key = (u64)root;
    // A node's extent at least halves (rounded up) per level, so after nd = ceil(log2(max(W, H))) levels it is one pixel in both
    // axes and every further digit is 0 (the midpoint of [x, x + 1) is x + 1): the loop stops there, the rest is a shift.
    const int ext = max(max(W, H), 2);
    const int nd  = min(KEY_DIGITS, 32 - __clz(ext - 1));
    for (int d = 0; d < nd; ++d)
    {
        const int mx = x0 + (x1 - x0 + 1) / 2, my = y0 + (y1 - y0 + 1) / 2;
        const int cx = x >= mx, cy = y >= my;
        x0 = cx ? mx : x0;
        x1 = cx ? x1 : mx;
        y0 = cy ? my : y0;
        y1 = cy ? y1 : my;
        key = (key << 2) | (u64)(cx + 2 * cy);
    }
    return key << (2 * (KEY_DIGITS - nd));  // 8 + 32 bits
}

// Bitonic sort of n_pow2 keys in LDS (ascending), all threads of the block participate.  Every wavefront
// owns a contiguous segment of n_pow2 / waves keys: compare-exchange stages whose partner distance stays
// inside the segment need no workgroup barrier (the LDS serves a wavefront in program order), only the
// stages that cross segments do -- 9 barriers instead of 66 for 2048 keys and 8 wavefronts.
__device__ void bitonic_sort(u64* a, int n_pow2, int tid, int nthreads)
{
    const int nw = nthreads >> 6, wave = tid >> 6, lane = tid & 63;
    const int seg = n_pow2 / nw;  // keys per wavefront segment
    if (seg < 128)
    {
        for (int k = 2; k <= n_pow2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1)
            {
                for (int t = tid; t < (n_pow2 >> 1); t += nthreads)
                {
                    const int i   = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int ixj = i | j;
                    const bool up = (i & k) == 0;
                    const u64 x = a[i], y = a[ixj];
                    if ((x > y) == up)
                    {
                        a[i]   = y;
                        a[ixj] = x;
                    }
                }
                __syncthreads();
            }
        return;
    }
    const int base = wave * seg;
    for (int k = 2; k <= n_pow2; k <<= 1)
    {
        int j = k >> 1;
        if (j >= seg) __syncthreads();  // the previous segment-local stages are visible to every wavefront
        for (; j >= seg; j >>= 1)      // stages across segments
        {
            for (int t = tid; t < (n_pow2 >> 1); t += nthreads)
            {
                const int i   = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int ixj = i | j;
                const bool up = (i & k) == 0;
                const u64 x = a[i], y = a[ixj];
                if ((x > y) == up)
                {
                    a[i]   = y;
                    a[ixj] = x;
                }
            }
            __syncthreads();
        }
        for (; j > 0; j >>= 1)  // stages inside the wavefront's own segment
        {
            for (int t = lane; t < (seg >> 1); t += 64)
            {
                const int i   = base + (((t & ~(j - 1)) << 1) | (t & (j - 1)));
                const int ixj = i | j;
                const bool up = (i & k) == 0;
                const u64 x = a[i], y = a[ixj];
                if ((x > y) == up)
                {
                    a[i]   = y;
                    a[ixj] = x;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
}

// block-wide inclusive scan of one int per thread (512 threads = 8 waves)
constexpr int DIST_THREADS = 512;
__device__ int block_scan_incl(int v, int tid, int* wave_tot /* >= 8 */)
{
    const int lane = tid & 63, wave = tid >> 6;
    int x = v;
    x = wave_scan_incl_dpp(x);
    if (lane == 63) wave_tot[wave] = x;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    __syncthreads();
    return x + base;
}

// Sort of n distinct subdivision keys (step 3 of distribute_body) by BUCKET + RANK instead of a comparison network.  The bitonic
// network does 1024 compare-exchanges in each of 66 stages for 2048 keys and is bound by exactly that work (51 k cycles per
// workgroup with four workgroups per compute unit, a third of the kernel; holding the keys in registers with lane shuffles did not
// help: 59 k).  The keys are quadtree paths, so their leading digits spread the points evenly: bucket = root and the first T digits
// (at most nb_max buckets), one LDS atomic per key counts the buckets, one scan places them, one more atomic per key lists the
// members, and a key's final position is its bucket's start plus the number of smaller keys in the bucket (a handful; at most the
// candidates of the few FAST cells a bucket's region overlaps).  Distinct keys: the result does not depend on the atomics' order.
// tmp: cap * 4 bytes (the node list of the careful phase, free at this point): u16 idx[n] | int hist[nb + 1].
__device__ void bucket_rank_sort(u64* keys, int n, int cap, int nroots, unsigned char* tmp, int tid, int* wave_tot)
{
    u16* idx  = reinterpret_cast<u16*>(tmp);
    int* hist = reinterpret_cast<int*>(tmp + 2 * cap);
    const int nb_max = min(1023, cap / 2 - 1);
    int T = 4;
    while (T > 0 && (nroots << (2 * T)) > nb_max) --T;
    const int nb    = nroots << (2 * T);  // nroots <= 255 <= nb_max for every carve this is called with (cap >= 512)
    const int shift = 13 + 32 - 2 * T;    // key = (root << 32 | 16 digits) << 13 | slot
    for (int i = tid; i <= nb; i += DIST_THREADS) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += DIST_THREADS) atomicAdd(&hist[(int)(keys[i] >> shift)], 1);
    __syncthreads();
    {
        // exclusive scan of the bucket counts, two buckets per thread
        const int b0 = 2 * tid, c0 = b0 < nb ? hist[b0] : 0, c1 = b0 + 1 < nb ? hist[b0 + 1] : 0;
        const int incl = block_scan_incl(c0 + c1, tid, wave_tot);
        if (b0 < nb) hist[b0] = incl - c0 - c1;
        if (b0 + 1 < nb) hist[b0 + 1] = incl - c1;
    }
    __syncthreads();
    for (int i = tid; i < n; i += DIST_THREADS) idx[atomicAdd(&hist[(int)(keys[i] >> shift)], 1)] = (u16)i;
    __syncthreads();
    // hist[b] is now the END of bucket b.  At most four keys per thread (n <= 2048): rank them, then move them.
    u64 mine[4];
    int dest[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
        const int i = tid + u * DIST_THREADS;
        mine[u] = 0;
        dest[u] = -1;
        if (i < n)
        {
            const u64 k = keys[i];
            const int b = (int)(k >> shift);
            const int s = b > 0 ? hist[b - 1] : 0, e = hist[b];
            int r       = s;
            int q       = s;
            for (; q + 2 <= e; q += 2)  // two members in flight
            {
                const u64 a0 = keys[idx[q]], a1 = keys[idx[q + 1]];
                r += (a0 < k ? 1 : 0) + (a1 < k ? 1 : 0);
            }
            if (q < e) r += keys[idx[q]] < k ? 1 : 0;
            mine[u] = k;
            dest[u] = r;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (dest[u] >= 0) keys[dest[u]] = mine[u];
    __syncthreads();
}

// Dynamic LDS carve (cap = level_cap, a power of two):
//   keys  u64[cap]      sort keys: subdivision key << 13 | candidate slot
//   nodes u64[cap/2]    careful-phase node list
//   px,py u16[cap]      candidate coordinates (level pixels)
//   sc    u8[cap]       scores
//   lcp   i8[cap + 1]   common-prefix length with the predecessor (-1 = different root)
//   fd    u8[cap]       final node depth of every sorted point
// Returns false (nothing written) when the level holds more candidates than this launch's LDS
// carve (lds_cap) can sort; the caller then queues the (image, level) for the large-LDS launch.
__device__ bool distribute_body(const Layout& L, int b, int l, int lds_cap, u32* cand /* slots of cut cells are re-ordered in place */,
                                const u16* __restrict__ cell_cnt, u32* __restrict__ sel /* [B][total_slots] x|y<<16 */,
                                u8* __restrict__ sel_score, int* __restrict__ sel_cnt,
                                int* __restrict__ cand_total /* debug: [B][levels] */,
                                unsigned long long* __restrict__ dbg_t = nullptr /* SNK_ORB_DIST_TIMING: [levels][16] cycle sums */,
                                u32* cand_h = nullptr /* "orb.response" = 1: Harris rank per candidate slot (moves with its slot) */,
                                u32* __restrict__ sel_resp = nullptr /* ... and the selected keypoints' ranks */,
                                u32* __restrict__ h_global = nullptr /* ranks of this workgroup's candidates when the LDS carve has no room (full-budget launch) */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int cap     = lds_cap;      // LDS carve of this launch
    const int sem_cap = L.level_cap;  // candidate budget of the definition
    u64* keys     = reinterpret_cast<u64*>(smem);
    u64* nodes    = keys + cap;
    u16* px       = reinterpret_cast<u16*>(nodes + cap / 2);
    u16* py       = px + cap;
    u8* sc        = reinterpret_cast<u8*>(py + cap);
    signed char* lcp = reinterpret_cast<signed char*>(sc + cap);
    u8* fd        = reinterpret_cast<u8*>(lcp + cap + 16);
    // Harris ranks of the gathered candidates: behind the carve (the launch adds 4 * cap bytes) or in global scratch
    u32* hv       = cand_h ? (h_global ? h_global : reinterpret_cast<u32*>(smem + ((cap * 8 + cap / 2 * 8 + cap * 2 * 2 + cap + (cap + 16) + cap + 15) & ~15)))
                           : nullptr;
    __shared__ int wave_tot[8];
    __shared__ int hist_lcp[20], hist_m[20];
    __shared__ int s_n, s_k, s_D, s_careful, s_size, s_nnodes, s_finish, s_jstar, s_out;

    const int tid = threadIdx.x;
    unsigned long long t_prev = dbg_t ? __builtin_readcyclecounter() : 0ull;
    auto mark = [&](int phase)  // SNK_ORB_DIST_TIMING (diagnostic): cycles of thread 0 since the previous mark
    {
        if (dbg_t && tid == 0)
        {
            const unsigned long long t = __builtin_readcyclecounter();
            atomicAdd(&dbg_t[l * 16 + phase], t - t_prev);
            t_prev = t;
        }
    };
    const LevelInfo& lv = L.lv[l];
    const int ncell = lv.ncols * lv.nrows;
    const int N     = lv.nfeat;
    const u16* cc   = cell_cnt + (long long)b * L.total_cells + lv.cell_off;
    u32* cd         = cand + ((long long)b * L.total_cells + lv.cell_off) * CELL_SLOTS;
    u32* chd        = cand_h ? cand_h + ((long long)b * L.total_cells + lv.cell_off) * CELL_SLOTS : nullptr;
    int* out_cnt    = sel_cnt + b * MAX_LEVELS + l;

    // ---- 1. per-cell budget k: largest k <= CELL_SLOTS with sum(min(cnt, k)) <= cap -------------
    if (tid == 0) s_k = CELL_SLOTS;
    __syncthreads();
    for (;;)
    {
        const int k = s_k;
        int part    = 0;
        for (int c = tid; c < ncell; c += DIST_THREADS) part += min((int)cc[c], k);
        const int incl = block_scan_incl(part, tid, wave_tot);
        if (tid == DIST_THREADS - 1) s_n = incl;
        __syncthreads();
        if (s_n <= sem_cap || k == 0) break;
        if (tid == 0) s_k = k - 1;
        __syncthreads();
    }
    const int kcell = s_k;
    const int n     = s_n;
    mark(0);
    if (n > cap) return false;
    // fast_kernel leaves the slots of a cell with <= CELL_SLOTS candidates in no particular order (cells with more hold their
    // CELL_SLOTS strongest, strongest first).  When the level budget cuts cells (kcell < CELL_SLOTS: more candidates on the level than
    // its budget -- noise, dense texture), a cell with kcell < count <= CELL_SLOTS is ranked here, in place, one wavefront per cell with a
    // lane per slot; keys are distinct, so the ranks are a permutation.  The Harris ranks ("orb.response" = 1) move with their slots.
    if (kcell < CELL_SLOTS)  // workgroup-uniform
    {
        u32* cdw = cd;
        u32* chw = chd;
        const int lane = tid & 63;
        for (int c = tid >> 6; c < ncell; c += DIST_THREADS / 64)  // wave-uniform
        {
            const int cnt = (int)cc[c];
            if (cnt <= kcell || cnt > CELL_SLOTS) continue;
            const u32 k = lane < cnt ? cdw[(long long)c * CELL_SLOTS + lane] : 0u;
            const u32 h = chw && lane < cnt ? chw[(long long)c * CELL_SLOTS + lane] : 0u;
            int r       = 0;
            for (int j = 0; j < cnt; ++j) r += (u32)__builtin_amdgcn_readlane((int)k, __builtin_amdgcn_readfirstlane(j)) > k ? 1 : 0;
            __builtin_amdgcn_wave_barrier();  // every lane holds its key before any slot is overwritten
            if (lane < cnt)
            {
                cdw[(long long)c * CELL_SLOTS + r] = k;
                if (chw) chw[(long long)c * CELL_SLOTS + r] = h;
            }
        }
        __syncthreads();  // the gather below reads the slots this workgroup has just rewritten (same CU, workgroup-scope ordering)
    }
    if (tid == 0 && cand_total) cand_total[b * MAX_LEVELS + l] = n;
    if (n == 0 || N <= 0)
    {
        if (tid == 0) *out_cnt = 0;
        return true;
    }
    int n_pow2 = 1;
    while (n_pow2 < n) n_pow2 <<= 1;

    // ---- 2. gather candidates (cell slots are strength-ordered: the first k are the k strongest) --
    {
        // cells are walked in chunks of DIST_THREADS with a running offset
        int base = 0;
        for (int c0 = 0; c0 < ncell; c0 += DIST_THREADS)
        {
            const int c   = c0 + tid;
            const int cnt = c < ncell ? min((int)cc[c], kcell) : 0;
            const int incl = block_scan_incl(cnt, tid, wave_tot);
            const int off  = base + incl - cnt;
            if (cnt > 0)
            {
                const int ci = c / lv.ncols, cj = c - ci * lv.ncols;
                const int x0 = EDGE_THRESHOLD + cj * lv.wcell, y0 = EDGE_THRESHOLD + ci * lv.hcell;
                // 32 slots in flight per thread (a full cell is two memory latencies; slot by slot it was 64, in eights 8)
                const uint4* cq = reinterpret_cast<const uint4*>(cd + (long long)c * CELL_SLOTS);
                for (int i = 0; i < cnt; i += 32)
                {
                    uint4 q[8];
#pragma unroll
                    for (int v = 0; v < 8; ++v) q[v] = i + 4 * v < cnt ? cq[(i >> 2) + v] : uint4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int v = 0; v < 8; ++v)
                    {
                        const u32 kk[4] = {q[v].x, q[v].y, q[v].z, q[v].w};
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (i + 4 * v + u < cnt)
                            {
                                const u32 k = kk[u];
                                const int o = off + i + 4 * v + u;
                                px[o] = (u16)(x0 + 63 - (int)(k & 63u));
                                py[o] = (u16)(y0 + 63 - (int)((k >> 6) & 63u));
                                sc[o] = (u8)(k >> 12);
                                if (hv) hv[o] = chd[(long long)c * CELL_SLOTS + i + 4 * v + u];
                            }
                    }
                }
            }
            if (tid == DIST_THREADS - 1) s_out = incl;
            __syncthreads();
            base += s_out;
            __syncthreads();
        }
    }
    mark(1);
    const int W = lv.w - 2 * MIN_BORDER, H = lv.h - 2 * MIN_BORDER;
    for (int i = tid; i < n_pow2; i += DIST_THREADS)
        keys[i] = i < n ? ((point_key(px[i] - MIN_BORDER, py[i] - MIN_BORDER, W, H, lv.nroots) << 13) | (u64)i) : ~0ull;
    __syncthreads();

    mark(2);
    // ---- 3. sort by subdivision key -------------------------------------------------------------
    if (n <= 4 * DIST_THREADS && cap >= 512)
        bucket_rank_sort(keys, n, cap, lv.nroots, reinterpret_cast<unsigned char*>(nodes), tid, wave_tot);
    else
        bitonic_sort(keys, n_pow2, tid, DIST_THREADS);  // the full-budget launch (levels with > 2048 candidates)
    mark(3);

    // ---- 4. common-prefix lengths and the per-depth node statistics ------------------------------
    if (tid < 20)
    {
        hist_lcp[tid] = 0;
        hist_m[tid]   = 0;
    }
    for (int i = tid; i <= n; i += DIST_THREADS)
    {
        int v = -1;
        if (i > 0 && i < n)
        {
            const u64 diff = (keys[i] >> 13) ^ (keys[i - 1] >> 13);  // 40-bit keys, never equal
            if ((diff >> 32) == 0)
            {
                const int hb = 63 - __clzll(diff);  // highest differing bit, 0..31
                v            = 15 - (hb >> 1);      // equal leading digits, 0..15
            }
        }
        lcp[i] = (signed char)v;
    }
    __syncthreads();
    for (int i = tid; i < n; i += DIST_THREADS)
    {
        if (i > 0) atomicAdd(&hist_lcp[lcp[i] + 1], 1);
        const int m = max((int)lcp[i], (int)lcp[i + 1]);  // lcp[0] = lcp[n] = -1
        atomicAdd(&hist_m[m + 1], 1);
    }
    __syncthreads();

    mark(4);
    // ---- 5. replay ORB-SLAM2's split passes on the statistics (one thread, <= 16 steps) -----------
    if (tid == 0)
    {
        // heads_d = 1 + #{i>=1 : lcp(i) < d};  singles_d = #{i : max(lcp(i), lcp(i+1)) < d}
        int cum_l = 0, cum_m = 0;
        int size_prev = 1 + hist_lcp[0];  // d = 0: roots that hold points
        cum_l = hist_lcp[0];
        cum_m = hist_m[0];
        int D = KEY_DIGITS, careful = 0, size_D = 0;
        bool done = false;
        for (int d = 1; d <= KEY_DIGITS && !done; ++d)
        {
            cum_l += hist_lcp[d];  // lcp value d-1
            cum_m += hist_m[d];
            const int size  = 1 + cum_l;
            const int multi = size - cum_m;
            if (size >= N || size == size_prev)
            {
                D = d; careful = 0; size_D = size; done = true;
            }
            else if (size + 3 * multi > N)
            {
                D = d; careful = 1; size_D = size; done = true;
            }
            size_prev = size;
            if (!done && d == KEY_DIGITS) { D = KEY_DIGITS; careful = 0; size_D = size; }
        }
        s_D       = D;
        s_careful = careful;
        s_size    = size_D;
        s_finish  = 0;
    }
    __syncthreads();
    const int D = s_D;
    for (int i = tid; i < n; i += DIST_THREADS) fd[i] = (u8)D;
    __syncthreads();

    mark(5);
    // ---- 6. careful phase: split the fullest nodes first until N nodes exist ----------------------
    if (s_careful && n <= 4 * DIST_THREADS)
    {
        // Rounds without a sorted node list (thread t owns positions 4 t .. 4 t + 3, at most four node heads, kept in registers): "the fullest nodes
        // first, ties by position, until N nodes exist" only needs the count class c* in which the running total reaches N --
        // a histogram of the new-node counts by point count and one suffix scan -- and, inside that class, a prefix over the
        // heads in position order.  Sorting the nodes (a bitonic network per round) was 27 % of the kernel.
        int* hist = reinterpret_cast<int*>(nodes);  // cap / 2 * 8 bytes = cap ints >= n - 1 entries (count c at c - 2)
        const int HN = n - 1;
        for (int dd = D; dd < KEY_DIGITS; ++dd)
        {
            for (int k = tid; k < HN; k += DIST_THREADS) hist[k] = 0;
            if (tid == 0) s_jstar = HN;  // "the total never reaches N"
            if (dbg_t && tid == 0) atomicAdd(&dbg_t[l * 16 + 14], 1ull);  // rounds
            __syncthreads();
            int cntv[4], delv[4];
            bool any = false;
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                const int i = 4 * tid + u;
                cntv[u] = delv[u] = 0;
                if (i < n && fd[i] == dd && (int)lcp[i] < dd)
                {
                    int e = i + 1, delta = 0;
                    for (;;)
                    {
                        // four entries per LDS round trip (lcp has 16 bytes of slack behind entry n; entries at or behind
                        // n are cut by the index test)
                        const int l0 = lcp[e], l1 = lcp[e + 1], l2 = lcp[e + 2], l3 = lcp[e + 3];
                        const bool c0 = e < n && l0 >= dd, c1 = c0 && e + 1 < n && l1 >= dd, c2 = c1 && e + 2 < n && l2 >= dd,
                                   c3 = c2 && e + 3 < n && l3 >= dd;
                        delta += (c0 && l0 == dd ? 1 : 0) + (c1 && l1 == dd ? 1 : 0) + (c2 && l2 == dd ? 1 : 0) + (c3 && l3 == dd ? 1 : 0);
                        e += (c0 ? 1 : 0) + (c1 ? 1 : 0) + (c2 ? 1 : 0) + (c3 ? 1 : 0);
                        if (!c3) break;
                    }
                    if (e - i > 1)
                    {
                        cntv[u] = e - i;
                        delv[u] = delta;
                        any     = true;
                        if (delta) atomicAdd(&hist[e - i - 2], delta);
                    }
                }
            }
            mark(8);
            if (!__syncthreads_or(any ? 1 : 0)) break;  // no node with more than one point left
            mark(9);
            // suffix sums over the count classes, fullest first: k = 0 is count n, entry HN - 1 - k
            int before_mine = 0;
            {
                int loc[4], sum = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    const int k = 4 * tid + u;
                    loc[u]      = k < HN ? hist[HN - 1 - k] : 0;
                    sum += loc[u];
                }
                const int incl = block_scan_incl(sum, tid, wave_tot);
                int run        = s_size + incl - sum;  // nodes before this thread's first class
                int found      = -1;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    if (found < 0 && loc[u] > 0 && run + loc[u] >= N) { found = 4 * tid + u; before_mine = run; }
                    run += loc[u];
                }
                if (found >= 0) atomicMin(&s_jstar, found);
                if (tid == DIST_THREADS - 1) s_out = run - s_size;  // new nodes if every node is split
            }
            __syncthreads();
            mark(10);
            const int kstar = s_jstar;
            if (kstar < HN && 4 * tid <= kstar && kstar < 4 * tid + 4) s_nnodes = before_mine;  // running total above class c*
            __syncthreads();
            const int cstar  = kstar < HN ? HN - 1 - kstar + 2 : 0;  // 0: every node is split
            const int before = kstar < HN ? s_nnodes : 0;
            int my_add = 0;
            bool split[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) split[u] = cntv[u] > 0 && (cstar == 0 || cntv[u] > cstar);
            if (cstar != 0)  // wave-uniform: the ties of class c* in position order, until the total reaches N
            {
                int d[4], sum = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    d[u] = cntv[u] == cstar ? delv[u] : 0;
                    sum += d[u];
                }
                int run = before + block_scan_incl(sum, tid, wave_tot) - sum;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    if (cntv[u] == cstar && run < N) split[u] = true;
                    run += d[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (split[u]) my_add += delv[u];
            mark(11);
            const int add = block_scan_incl(my_add, tid, wave_tot);
            if (tid == DIST_THREADS - 1)
            {
                s_finish = (s_size + add >= N || add == 0) ? 1 : 0;
                s_size += add;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (split[u])
                    for (int e = 4 * tid + u, e1 = e + cntv[u]; e < e1; ++e) fd[e] = (u8)(dd + 1);
            __syncthreads();
            mark(12);
            if (s_finish) break;
        }
    }
    else if (s_careful)
    {
        // the full-budget launch (more than 2048 candidates): sorted node list
        // round r works on nodes of depth dd = D + r whose points have fd == dd ("active")
        for (int dd = D; dd < KEY_DIGITS; ++dd)
        {
            if (tid == 0) s_nnodes = 0;
            __syncthreads();
            // a node head at depth dd: lcp(i) < dd; active nodes of this round have fd == dd and, for
            // rounds after the first, were created by a split in the previous round (fd was raised to dd)
            for (int i = tid; i < n; i += DIST_THREADS)
            {
                if (fd[i] != dd || (int)lcp[i] >= dd) continue;
                int e = i + 1, delta = 0;
                while (e < n && (int)lcp[e] >= dd)
                {
                    delta += (int)lcp[e] == dd ? 1 : 0;
                    ++e;
                }
                const int cnt = e - i;
                if (cnt > 1)
                {
                    // order: count descending, then position (= key prefix) ascending
                    const u64 nk = ((u64)(0xFFFF - cnt) << 32) | ((u64)i << 16) | (u64)delta;
                    nodes[atomicAdd(&s_nnodes, 1)] = nk;
                }
            }
            __syncthreads();
            const int m = s_nnodes;
            if (m == 0) break;
            int m_pow2 = 1;
            while (m_pow2 < m) m_pow2 <<= 1;
            for (int i = m + tid; i < m_pow2; i += DIST_THREADS) nodes[i] = ~0ull;
            __syncthreads();
            bitonic_sort(nodes, m_pow2, tid, DIST_THREADS);
            // running node count after splitting the first j+1 nodes; j* = first j reaching N
            if (tid == 0) s_jstar = m;  // "none"
            __syncthreads();
            int base = s_size;
            for (int j0 = 0; j0 < m; j0 += DIST_THREADS)
            {
                const int j     = j0 + tid;
                const int delta = j < m ? (int)(nodes[j] & 0xFFFFu) : 0;
                const int incl  = block_scan_incl(delta, tid, wave_tot);
                if (j < m && base + incl >= N) atomicMin(&s_jstar, j);
                if (tid == DIST_THREADS - 1) s_out = incl;
                __syncthreads();
                base += s_out;
                __syncthreads();
            }
            const int jstar = s_jstar;
            const int last  = jstar < m ? jstar : m - 1;  // nodes 0..last are split
            // size after this round
            {
                // nodes created by this round's splits: block-wide sum (one thread walking the list paid an LDS round trip
                // per node)
                int part = 0;
                for (int j = tid; j <= last; j += DIST_THREADS) part += (int)(nodes[j] & 0xFFFFu);
                const int add = block_scan_incl(part, tid, wave_tot);
                if (tid == DIST_THREADS - 1)
                {
                    s_finish = (s_size + add >= N || add == 0) ? 1 : 0;
                    s_size += add;
                }
            }
            for (int j = tid; j <= last; j += DIST_THREADS)
            {
                const int i = (int)((nodes[j] >> 16) & 0xFFFFu);
                int e       = i;
                do
                {
                    fd[e] = (u8)(dd + 1);
                    ++e;
                } while (e < n && (int)lcp[e] >= dd);
            }
            __syncthreads();
            if (s_finish) break;
        }
    }
    __syncthreads();

    mark(6);
    // ---- 7. one keypoint per final node: highest score, then smallest y, then smallest x ----------
    // Every point knows its node's output slot without looking for the node's head: slot = number of node heads at or before
    // it (the block scan of the head flags), and the winner of a node is an LDS atomic maximum of the packed priority
    // score | 0xFFFF - y | 0xFFFF - x, which also carries everything the output needs -- no walk along the node's points.
    {
        u32* out    = sel + (long long)b * L.total_slots + lv.slot_off;
        u8* out_sc  = sel_score + (long long)b * L.total_slots + lv.slot_off;
        // the priorities replace the keys in place; after that the node list and the coordinate arrays are dead and their
        // 8 * cap contiguous bytes (nodes | px | py) hold one winner slot per possible node (<= n <= cap)
        for (int i = tid; i < n; i += DIST_THREADS)
        {
            const int c = (int)(keys[i] & 0x1FFFu);
            keys[i]     = ((u64)(hv ? hv[c] : (u32)sc[c]) << 32) | ((u64)(0xFFFFu - py[c]) << 16) | (u64)(0xFFFFu - px[c]);
        }
        __syncthreads();
        unsigned long long* best = reinterpret_cast<unsigned long long*>(nodes);
        const int best_cap = min(n, lv.slot_cap);
        for (int k = tid; k < best_cap; k += DIST_THREADS) best[k] = 0ull;
        __syncthreads();
        int base = 0;
        for (int i0 = 0; i0 < n; i0 += DIST_THREADS)
        {
            const int i     = i0 + tid;
            const bool head = i < n && (int)lcp[i] < (int)fd[i];
            const int incl  = block_scan_incl(head ? 1 : 0, tid, wave_tot);
            const int pos   = base + incl - 1;
            if (i < n && pos < best_cap) atomicMax(&best[pos], (unsigned long long)keys[i]);
            if (tid == DIST_THREADS - 1) s_out = incl;
            __syncthreads();
            base += s_out;
            __syncthreads();
        }
        const int count = base < lv.slot_cap ? base : lv.slot_cap;
        for (int k = tid; k < count; k += DIST_THREADS)
        {
            const unsigned long long w = best[k];
            out[k]    = (0xFFFFu - (u32)(w & 0xFFFFu)) | ((0xFFFFu - (u32)((w >> 16) & 0xFFFFu)) << 16);
            out_sc[k] = (u8)(w >> 32);
            if (sel_resp) sel_resp[(long long)b * L.total_slots + lv.slot_off + k] = (u32)(w >> 32);
        }
        if (tid == 0) *out_cnt = count;
    }
    mark(7);
    if (dbg_t && tid == 0) atomicAdd(&dbg_t[l * 16 + 15], (unsigned long long)n);
    return true;
}

// small-LDS launch over every (level, image); levels that do not fit are queued
__global__ __launch_bounds__(DIST_THREADS, SNK_DIST_MIN_WAVES) void distribute_kernel(Layout L, int lds_cap, u32* cand,
                                                                  const u16* __restrict__ cell_cnt, u32* __restrict__ sel,
                                                                  u8* __restrict__ sel_score, int* __restrict__ sel_cnt,
                                                                  int* __restrict__ cand_total, int* __restrict__ queue,
                                                                  unsigned long long* __restrict__ dbg_t, u32* cand_h,
                                                                  u32* __restrict__ sel_resp)
{
    const int l = blockIdx.x, b = blockIdx.y;
    if (!distribute_body(L, b, l, lds_cap, cand, cell_cnt, sel, sel_score, sel_cnt, cand_total, dbg_t, cand_h, sel_resp) && threadIdx.x == 0)
        queue[1 + atomicAdd(&queue[0], 1)] = b * MAX_LEVELS + l;
}

// full-budget launch: a fixed set of workgroups drains the queue (normally empty)
__global__ __launch_bounds__(DIST_THREADS) void distribute_large_kernel(Layout L, u32* cand,
                                                                        const u16* __restrict__ cell_cnt,
                                                                        u32* __restrict__ sel, u8* __restrict__ sel_score,
                                                                        int* __restrict__ sel_cnt, int* __restrict__ cand_total,
                                                                        const int* __restrict__ queue, u32* cand_h,
                                                                        u32* __restrict__ sel_resp, u32* __restrict__ h_scratch)
{
    const int count = queue[0];
    for (int i = blockIdx.x; i < count; i += gridDim.x)
    {
        const int item = queue[1 + i];
        distribute_body(L, item / MAX_LEVELS, item % MAX_LEVELS, L.level_cap, cand, cell_cnt, sel, sel_score, sel_cnt, cand_total, nullptr,
                        cand_h, sel_resp, cand_h ? h_scratch + (size_t)blockIdx.x * L.level_cap : nullptr);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// orientation + descriptor
// ------------------------------------------------------------------------------------------------
// cv::fastAtan2's polynomial with a fixed operation order (degrees in [0, 360]).
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float p1 = 0x1.ca44dep+5f, p3 = -0x1.2aaddcp+4f, p5 = 0x1.1d3f7ep+3f, p7 = -0x1.4515b2p+1f;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay)
    {
        c  = ay / (ax + 0x1p-52f);
        c2 = c * c;
        a  = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    else
    {
        c  = ax / (ay + 0x1p-52f);
        c2 = c * c;
        a  = 90.0f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.0f - a;
    if (y < 0) a = 360.0f - a;
    return a;
}

// sine / cosine of degrees: exact octant reduction + Taylor polynomials on [0, 45] degrees.
__device__ __forceinline__ void sincos_deg(float deg, float& s_out, float& c_out)
{
    int q   = 0;
    float r = deg;
    if (r >= 360.0f) r -= 360.0f;
    if (r >= 270.0f) { r -= 270.0f; q = 3; }
    else if (r >= 180.0f) { r -= 180.0f; q = 2; }
    else if (r >= 90.0f) { r -= 90.0f; q = 1; }
    bool swap = false;
    if (r > 45.0f) { r = 90.0f - r; swap = true; }
    const float x  = r * 0.017453292519943295f;
    const float x2 = x * x;
    float s = x + x * x2 * (-1.6666667e-1f + x2 * (8.3333333e-3f + x2 * (-1.9841270e-4f + x2 * 2.7557319e-6f)));
    float c = 1.0f + x2 * (-0.5f + x2 * (4.1666667e-2f + x2 * (-1.3888889e-3f + x2 * 2.4801587e-5f)));
    if (swap) { const float t = s; s = c; c = t; }
    switch (q)
    {
        case 0: s_out = s; c_out = c; break;
        case 1: s_out = c; c_out = -s; break;
        case 2: s_out = -s; c_out = -c; break;
        default: s_out = -c; c_out = s; break;
    }
}

// sum over the 64 lanes, returned wave-uniform: DPP butterflies inside each row of 16 lanes, then the
// four row sums are read back with v_readlane
__device__ __forceinline__ int wave_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

// One wavefront describes DESC_KPW consecutive keypoints of a level.  The kernel is bound by load
// latency, not arithmetic, so every global load a wavefront needs is issued before the first use: for
// each of its keypoints the 31 x 9 raw dwords of the moment window and the 37 x 10 dwords of the blurred
// patch that the steered tests can reach (|rotated offset| <= 18).  Moments: v_dot4 against the disc
// table (LDS copy); the patch goes to the wavefront's LDS slice and the 512 test bytes are LDS reads;
// 4 ballots assemble the descriptor.
constexpr int DESC_KPW    = 4;
constexpr int PATCH_R     = 18, PATCH_DW = 12;                  // 37 rows x 12 dwords (10 needed): a row is 3 lanes x 16 bytes
constexpr int PATCH_QUADS = (2 * PATCH_R + 1) * 3;              // 111 sixteen-byte items -> 2 loads per lane
constexpr int PATCH_ITEMS = (2 * PATCH_R + 1) * PATCH_DW;
constexpr int MOM_TRIPLES = 31 * 3;                             // a moment-window row (9 dwords) is 3 lanes x 12 bytes -> 2 loads per lane
typedef u32 u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));

// BLUR_IN = true (SNK_ORB_BLUR_IN_DESCRIBE=1; the round-4 review's item 1c, an EXPERIMENT: results are right in the interior of a level and
// wrong within 21 pixels of its border, where the level-wide blur reflects and this one reads what is there): no blurred level at all --
// the descriptor wavefront loads the RAW 43 x 48-byte window of a keypoint (129 sixteen-byte items, three loads per lane), blurs it in LDS
// (horizontal pass: 430 (row, dword) items, 3 LDS reads + 6 v_alignbyte + 8 v_dot4 + 2 LDS writes each; vertical pass: 370 items, 14 LDS
// reads + 12 v_perm + 12 v_dot2 + 4 v_mad + packing + 1 LDS write each) and runs the steered tests on the result; level_kernel then skips
// the store of the blurred level.  Measured in profiles/r05/r05v_blur_in_describe.txt.
// DMA = true (round 6; the round-5 review's item 1b, SNK_ORB_DESC_DMA=1): the blurred patch goes from global memory straight into the
// wavefront's LDS slice with gfx950's LDS-DMA loads (global_load_lds_dwordx4: a lane names a global address, the 64 x 16 bytes of the
// instruction land side by side at the LDS address in M0) -- no patch registers (32 of the 94), no ds_write of the patch.  Two patch
// buffers per wavefront: the patches of keypoints 0 and 1 are requested up front with everything else, the patch of keypoint s + 2 is
// requested into the buffer keypoint s has just been tested from; counted s_waitcnt vmcnt keep the younger request in flight while a
// keypoint is tested, which is why the kernel's global STORES all come at its end (loads return in order among themselves, stores do not
// with respect to them).  Same bytes in the same LDS layout as the register path: identical results.
// (Sixteen wavefronts per workgroup -- the 10 KB moment table shared by more wavefronts: 8 instead of 6 per SIMD at the DMA form's 60
// registers -- was measured and is much slower, 2.0 against 1.37 ms: profiles/r06/r06d_ab_describe_16_wave_workgroups_negative.txt.)
// TILED = true (with DMA): the blurred plane is in the 32 x 4 tiled layout (blur_tiled_x / _y): the patch is 37 rows of FOUR 16-byte
// chunks at 16-pixel-aligned columns (no chunk straddles a tile; 40 needed pixels at any alignment fit 64), 148 items = three LDS-DMA
// instructions, 64-byte patch rows in LDS.
template <bool BLUR_IN, bool DMA = false, bool TILED = false>
__global__ __launch_bounds__(256) void describe_kernel(Layout L, const u8* __restrict__ img0, int pitch0,
                                                       long long stride0, int aligned0, const u32* __restrict__ sel,
                                                       const u8* __restrict__ sel_score, const int* __restrict__ sel_cnt,
                                                       snk_keypoint* __restrict__ kps, u64* __restrict__ desc,
                                                       int* __restrict__ n_out, int out_cap, int gx, int batch, int dbg_fake,
                                                       const u32* __restrict__ sel_resp /* "orb.response" = 1: Harris ranks, else NULL */)
{
    constexpr int PR     = BLUR_IN ? PATCH_R + 3 : PATCH_R;     // rows above / below the keypoint in the loaded window
    constexpr int PQUADS = (2 * PR + 1) * 3;                      // sixteen-byte items of the window
    constexpr int NB     = (PQUADS + 63) / 64;                    // loads per lane
    static_assert(!(BLUR_IN && DMA), "the LDS-DMA form reads the blurred level");
    static_assert(!TILED || DMA, "the tiled blurred plane is read by the LDS-DMA form only");
    constexpr int PROW   = TILED ? 16 : PATCH_DW;                 // dwords of a patch row in LDS
    constexpr int TITEMS = (2 * PATCH_R + 1) * 4;                 // TILED: 148 sixteen-byte items
    constexpr int NBT    = TILED ? (TITEMS + 63) / 64 : NB;       // LDS-DMA instructions per patch
    constexpr int PBUF   = TILED ? TITEMS * 4 : 64 * NB * 4;      // dwords of one LDS-DMA patch buffer (TILED: lanes past the last item are switched off)
    __shared__ uint2 mtab[4 * MOM_PAD];
    constexpr int WPB = 4;  // wavefronts per workgroup
    __shared__ __attribute__((aligned(16))) u32 patch[WPB][DMA ? 2 * PBUF : (2 * PR + 1) * PATCH_DW + 16];
    __shared__ u32 hbuf[BLUR_IN ? 4 : 1][BLUR_IN ? (2 * PR + 1) * 20 : 1];  // horizontally blurred rows, 40 pixels x 16 bit
    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // XCD-aware 1-D grid (workgroup L runs on XCD L % 8, a speed matter only): every workgroup of an image
    // gets the same L % 8, so the image's raw / blurred windows are served by one XCD's L2
    const int per_img = gx * L.n_levels;
    int b, rem;
    if (batch >= 16)
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        b   = (slot / per_img) * 8 + xcd;
        rem = slot - (slot / per_img) * per_img;
    }
    else
    {
        b   = blockIdx.x / per_img;
        rem = blockIdx.x - b * per_img;
    }
    if (b >= batch) return;
    const int l  = rem / gx;
    const int bx = rem - l * gx;
    const int* cnts = sel_cnt + b * MAX_LEVELS;
    int offset = 0, total = 0;
    for (int k = 0; k < L.n_levels; ++k)
    {
        const int c = cnts[k];
        offset += k < l ? c : 0;
        total += c;
    }
    if (bx == 0 && l == 0 && tid == 0) n_out[b] = total < out_cap ? total : out_cap;
    const LevelInfo& lv = L.lv[l];
    const int cnt_l     = cnts[l];
    if (bx * WPB * DESC_KPW >= cnt_l) return;  // whole workgroup (the grid is sized for the largest level)
    const int slot0     = (bx * WPB + wave) * DESC_KPW;
    const bool work     = slot0 < cnt_l && offset + slot0 < out_cap;  // whole wavefront

    const u8* src      = l == 0 ? img0 + (long long)b * stride0 : lv.base + (long long)b * lv.img_stride;
    const int pitch    = l == 0 ? pitch0 : lv.pitch;
    // dbg_fake == 2 (SNK_ORB_DESC_FAKE=2, timing experiment only, results meaningless): the patch from the RAW level, i.e. the rows the
    // moment window reads anyway -- the memory side of "blur inside describe_kernel" (one window per keypoint instead of two)
    const u8* bsrc     = (dbg_fake == 2 || BLUR_IN) ? src : lv.blur + (long long)b * lv.blur_stride;
    const int bpitch   = (dbg_fake == 2 || BLUR_IN) ? pitch : lv.pitch;
    const bool aligned = l == 0 ? aligned0 != 0 : true;

    // per-lane geometry, shared by the keypoints: byte offsets of its moment / patch items relative to the window
    // origins.  A lane moves 12 (moment window) or 16 (patch) bytes per load: the texture-address unit works through a
    // wavefront load 4 lanes per cycle however few bytes a lane asks for, and with one dword per lane the 44 loads of a
    // wavefront (11 per keypoint) kept it busy for ~700 cycles; now 16 loads carry the same rows.
    int moff[2], vyv[2], boff[NB];
    bool mok[2], bok[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k)
    {
        const int item = lane + 64 * k;
        bok[k]         = item < PQUADS;
        const int ib   = bok[k] ? item : PQUADS - 1;
        const int rb   = (ib * 171) >> 9;
        boff[k]        = (dbg_fake == 1 || dbg_fake == 4) ? 16 * ib : (rb - PR) * bpitch + 16 * (ib - rb * 3);  // 4: only the PATCH from one contiguous run (timing experiment)
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
    {
        const int item = lane + 64 * k;
        mok[k]         = item < MOM_TRIPLES;
        const int it   = mok[k] ? item : MOM_TRIPLES - 1;
        const int row  = (it * 171) >> 9;  // it / 3
        vyv[k]         = row - 15;
        // dbg_fake (SNK_ORB_DESC_FAKE=1, timing experiment only, results meaningless): the window's bytes from ONE contiguous run
        // behind the keypoint instead of 31 / 37 rows -- what the kernel would cost with ~46 instead of ~111 sectors per keypoint
        moff[k]        = dbg_fake == 1 ? 12 * it : (row - 15) * pitch + 12 * (it - row * 3);
    }

    // ---- issue every load of the wavefront; the copy of the moment table to LDS (and its barrier) comes after, so
    // that its latency overlaps theirs ----
    bool valid[DESC_KPW];
    int kxv[DESC_KPW], kyv[DESC_KPW], scv[DESC_KPW];
    u32x3_a4 dwv[DESC_KPW][2];
    u32x4_a4 bpv[DMA ? 1 : DESC_KPW][NB];
    const long long sbase = (long long)b * L.total_slots + lv.slot_off;
    // LDS-DMA request of keypoint s's blurred patch into buffer s & 1 of this wavefront (lanes past the last item are switched off)
    auto dma_patch = [&](int s)
    {
        if constexpr (DMA && TILED)
        {
            // item = (row r, chunk c): 16 bytes at column xb + 16 c (a multiple of 16: inside one 32-pixel tile) of row ky - 18 + r
            const int xb = (kxv[s] - PATCH_R) & ~15, y0 = kyv[s] - PATCH_R;
            u32* dst     = &patch[wave][PBUF * (s & 1)];
#pragma unroll
            for (int k = 0; k < NBT; ++k)
            {
                const int it = lane + 64 * k;
                if (it < TITEMS)
                {
                    const int r = it >> 2, c = it & 3;
                    const int x = min(xb + 16 * c, bpitch - 16);  // the fourth chunk may lie right of the plane: any bytes will do, inside the plane
                    const u8* g = bsrc + (size_t)(blur_tiled_y(y0 + r, bpitch) + blur_tiled_x(x));
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(dst + 256 * k), 16, 0, 0);
                }
            }
        }
        else if constexpr (DMA)
        {
            const int xb = (kxv[s] - PR) & ~3;
            const u8* bo = bsrc + ((long long)kyv[s] * bpitch + xb);
            u32* dst     = &patch[wave][PBUF * (s & 1)];
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (bok[k])
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bo + boff[k]),
                                                     (__attribute__((address_space(3))) void*)(dst + 256 * k), 16, 0, 0);
        }
    };
#pragma unroll
    for (int s = 0; s < DESC_KPW; ++s)
    {
        const int slot = slot0 + s;
        valid[s]       = work && slot < cnt_l && offset + slot < out_cap;
        kxv[s] = kyv[s] = scv[s] = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) dwv[s][k] = u32x3_a4{0u, 0u, 0u};
        if constexpr (!DMA)
        {
#pragma unroll
            for (int k = 0; k < NB; ++k) bpv[s][k] = u32x4_a4{0u, 0u, 0u, 0u};
        }
    }
    if (work)
    {
#pragma unroll
        for (int s = 0; s < DESC_KPW; ++s)
        {
            const int slot = slot0 + s;
            const u32 xy   = sel[sbase + (valid[s] ? slot : slot0)];
            scv[s]         = sel_resp ? (int)sel_resp[sbase + (valid[s] ? slot : slot0)] : (int)sel_score[sbase + (valid[s] ? slot : slot0)];
            kxv[s]         = (int)(xy & 0xFFFFu);
            kyv[s]         = (int)(xy >> 16);
        }
#pragma unroll
        for (int s = 0; s < DESC_KPW; ++s)
        {
            const int xa = (kxv[s] - 15) & ~3, xb = (kxv[s] - PR) & ~3;
            const u8* mo = src + ((long long)kyv[s] * pitch + xa);
            // (BLUR_IN: the window may reach 2 rows / columns outside the level; the row is kept inside the plane, what is read there is
            // not what the level-wide blur reflects -- see the kernel's header)
            const u8* bo = bsrc + ((long long)(BLUR_IN ? min(max(kyv[s], PR), lv.h - 1 - PR) : kyv[s]) * bpitch + (BLUR_IN ? max(xb, 0) : xb));
            // lanes past the last item repeat it (no divergent branch around a load); their moment weights are zeroed
            // below and their patch quads are not stored
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (aligned) dwv[s][k] = *reinterpret_cast<const u32x3_a4*>(mo + moff[k]);
            if constexpr (!DMA)
            {
#pragma unroll
                for (int k = 0; k < NB; ++k) bpv[s][k] = *reinterpret_cast<const u32x4_a4*>(bo + boff[k]);
            }
        }
        if constexpr (DMA)
        {
            // behind the moment-window loads (the moments are needed first): the first two patches
            if (valid[0]) dma_patch(0);
            if (valid[1]) dma_patch(1);
        }
    }
    {
        const uint2* g = reinterpret_cast<const uint2*>(&c_moment.v[0][0][0]);
        for (int i = tid; i < 4 * MOM_PAD; i += 64 * WPB) mtab[i] = g[i];
    }
    __syncthreads();
    if (!work) return;
    float pt[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) pt[k][e] = c_pattern_f.v[k * 64 + lane][e];

    // ---- orientation of the wavefront's keypoints: intensity-centroid moments over the radius-15 disc (integers),
    // then ONE evaluation of atan2 / sincos with the rows of 16 lanes working on different keypoints (the values are
    // wave-uniform per keypoint; evaluating them keypoint by keypoint costs every lane the same work four times)
    static_assert(DESC_KPW == 4, "one row of 16 lanes per keypoint");
    int m10v[DESC_KPW], m01v[DESC_KPW];
#pragma unroll
    for (int s = 0; s < DESC_KPW; ++s)
    {
        const int kx = kxv[s], ky = kyv[s];
        int m10 = 0, m01 = 0;
        if (aligned)
        {
            const int sh2 = (kx - 15) & 3;
            u32 sx = 0, s0 = 0;
#pragma unroll
            for (int k = 0; k < 2; ++k)
            {
                // item = (row, third of the row): its three dwords are dwords 3 * item .. 3 * item + 2 of the window
                const int it = mok[k] ? lane + 64 * k : 0;
                u32 sr       = 0;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                {
                    const uint2 w = mtab[sh2 * MOM_PAD + 3 * it + j];
                    const u32 d   = mok[k] ? dwv[s][k][j] : 0u;
                    sx            = __builtin_amdgcn_udot4(d, w.x, sx, false);
                    sr            = __builtin_amdgcn_udot4(d, w.y, sr, false);
                }
                s0 += sr;
                m01 += vyv[k] * (int)sr;
            }
            m10 = (int)sx - 15 * (int)s0;
        }
        else
        {
            // byte path: lane & 31 = column, lane >> 5 = row parity
            const int ux = (lane & 31) - 15;
            const int au = ux < 0 ? -ux : ux;
#pragma unroll 4
            for (int k = 0; k < 16; ++k)
            {
                const int vy = 2 * k + (lane >> 5) - 15;
                const int av = vy < 0 ? -vy : vy;
                if (vy <= 15 && ux <= 15 && au <= c_umax[av])
                {
                    const int p = src[(long long)(ky + vy) * pitch + kx + ux];
                    m10 += ux * p;
                    m01 += vy * p;
                }
            }
        }
        m10v[s] = wave_sum(m10);
        m01v[s] = wave_sum(m01);
    }
    float angle_l, sn_l, cs_l;
    {
        const int row = lane >> 4;
        const int a10 = row == 0 ? m10v[0] : (row == 1 ? m10v[1] : (row == 2 ? m10v[2] : m10v[3]));
        const int a01 = row == 0 ? m01v[0] : (row == 1 ? m01v[1] : (row == 2 ? m01v[2] : m01v[3]));
        angle_l       = fast_atan2_deg((float)a01, (float)a10);
        sincos_deg(angle_l, sn_l, cs_l);
    }

    u64 wordv[DMA ? DESC_KPW : 1][4];  // DMA: the descriptors wait for the stores at the end (wave-uniform: scalar registers)
#pragma unroll
    for (int s = 0; s < DESC_KPW; ++s)
    {
        if (!valid[s]) break;  // wave-uniform
        const int kx = kxv[s], ky = kyv[s];
        const u8* pb = reinterpret_cast<const u8*>(DMA ? &patch[wave][PBUF * (s & 1)] : &patch[wave][0]);
        if constexpr (DMA)
        {
            // the patch of keypoint s has landed when at most the younger request (keypoint s + 1's: NB instructions) is outstanding
            if (s + 1 < DESC_KPW && valid[s + 1])
                __builtin_amdgcn_s_waitcnt(0x0F70 | NBT);  // vmcnt(NBT), expcnt / lgkmcnt untouched
            else
                __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
            __builtin_amdgcn_wave_barrier();
        }
        else
        {
            // blurred patch -> LDS (in-order per wavefront: the previous keypoint's reads are done)
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (bok[k]) reinterpret_cast<u32x4_a16*>(patch[wave])[lane + 64 * k] = bpv[s][k];
        }
        if constexpr (BLUR_IN)
        {
            // 7 x 7 blur of the raw window in LDS: {18, 33, 49, 56, 49, 33, 18} / 256 per axis, exact 16-bit rows, one rounding -- the
            // arithmetic of level_kernel on a keypoint's own window.  Blurred dword j of a row = raw dword j + d0 (the raw window starts
            // 3 columns further left and is aligned separately).
            __builtin_amdgcn_wave_barrier();
            const u32 W0123 = 18u | (33u << 8) | (49u << 16) | (56u << 24), W456 = 49u | (33u << 8) | (18u << 16);
            const int d0    = (((kx - PATCH_R) & ~3) - ((kx - PR) & ~3)) >> 2;
            const u32* raw  = patch[wave];
            u32* hb         = hbuf[wave];
            for (int i = lane; i < (2 * PR + 1) * 10; i += 64)  // horizontal pass
            {
                const int row = (i * 205) >> 11, j = i - row * 10;  // i / 10
                const int at  = row * PATCH_DW + j + d0;
                const u32 dl = raw[max(at - 1, 0)], d = raw[at], dr = raw[at + 1];
                const u32 q0 = __builtin_amdgcn_alignbyte(d, dl, 1), q1 = __builtin_amdgcn_alignbyte(d, dl, 2), q2 = __builtin_amdgcn_alignbyte(d, dl, 3);
                const u32 r0 = __builtin_amdgcn_alignbyte(dr, d, 1), r1 = __builtin_amdgcn_alignbyte(dr, d, 2), r2 = __builtin_amdgcn_alignbyte(dr, d, 3);
                const u32 h0 = __builtin_amdgcn_udot4(q0, W0123, __builtin_amdgcn_udot4(r0, W456, 0u, false), false);
                const u32 h1 = __builtin_amdgcn_udot4(q1, W0123, __builtin_amdgcn_udot4(r1, W456, 0u, false), false);
                const u32 h2 = __builtin_amdgcn_udot4(q2, W0123, __builtin_amdgcn_udot4(r2, W456, 0u, false), false);
                const u32 h3 = __builtin_amdgcn_udot4(d, W0123, __builtin_amdgcn_udot4(dr, W456, 0u, false), false);
                hb[row * 20 + 2 * j]     = h0 | (h1 << 16);
                hb[row * 20 + 2 * j + 1] = h2 | (h3 << 16);
            }
            __builtin_amdgcn_wave_barrier();
            u32* outp = patch[wave];
            for (int i = lane; i < (2 * PATCH_R + 1) * 10; i += 64)  // vertical pass: blurred row r from horizontal rows r .. r + 6
            {
                const int row = (i * 205) >> 11, j = i - row * 10;
                u32 a[7][2];
#pragma unroll
                for (int t = 0; t < 7; ++t) a[t][0] = hb[(row + t) * 20 + 2 * j], a[t][1] = hb[(row + t) * 20 + 2 * j + 1];
                u32 px4[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                {
                    // pixel c of the dword: half (c & 1) of word (c >> 1); rows paired by v_perm for v_dot2
                    const u32 sel = (c & 1) ? 0x07060302u : 0x05040100u;
                    const u32 p01 = __builtin_amdgcn_perm(a[1][c >> 1], a[0][c >> 1], sel), p23 = __builtin_amdgcn_perm(a[3][c >> 1], a[2][c >> 1], sel),
                              p45 = __builtin_amdgcn_perm(a[5][c >> 1], a[4][c >> 1], sel);
                    const u32 h6  = (c & 1) ? a[6][c >> 1] >> 16 : a[6][c >> 1] & 0xFFFFu;
                    u32 t = __umul24(h6, 18u) + 32768u;
                    t     = dot2(p01, 18u | (33u << 16), t);
                    t     = dot2(p23, 49u | (56u << 16), t);
                    px4[c] = dot2(p45, 49u | (33u << 16), t);
                }
                const u32 lo = __builtin_amdgcn_perm(px4[1], px4[0], 0x0c0c0602u), hi = __builtin_amdgcn_perm(px4[3], px4[2], 0x06020c0cu);
                __builtin_amdgcn_wave_barrier();  // (the write below lands in the raw window: all lanes of this round have read)
                outp[row * PATCH_DW + j] = lo | hi;
            }
            __builtin_amdgcn_wave_barrier();
        }
        const float sn    = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sn_l), 16 * s));
        const float cs    = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cs_l), 16 * s));

        // 256 steered tests on the blurred patch; lane computes bits lane, lane+64, lane+128, lane+192
        const int pc = TILED ? PATCH_R * (4 * PROW) + PATCH_R + ((kx - PATCH_R) & 15)       // patch byte of the keypoint
                             : PATCH_R * (4 * PATCH_DW) + PATCH_R + ((kx - PATCH_R) & 3);
        u64 word[4];
        if constexpr (DMA)
        {
            // The eight test bytes of a lane by ds_read_u8 in ONE assembly block with its own lgkmcnt wait: behind an LDS-DMA request the
            // compiler guards every LDS read it knows of with s_waitcnt vmcnt(0) (it cannot tell the two patch buffers apart), which would
            // also wait for the NEXT keypoint's patch just requested -- the counted vmcnt above is the only wait these reads need.
            u32 adr[8], tb[8];
            const u32 lds0 = (u32)(size_t)(const __attribute__((address_space(3))) u8*)pb + (u32)pc;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                {
                    const float pxf = pt[k][2 * e], pyf = pt[k][2 * e + 1];
                    const int ry    = __float2int_rn(pxf * sn + pyf * cs);
                    const int rx    = __float2int_rn(pxf * cs - pyf * sn);
                    adr[2 * k + e]  = lds0 + (u32)(ry * (4 * PROW) + rx);
                }
            asm volatile("ds_read_u8 %0, %8\n\tds_read_u8 %1, %9\n\tds_read_u8 %2, %10\n\tds_read_u8 %3, %11\n\t"
                         "ds_read_u8 %4, %12\n\tds_read_u8 %5, %13\n\tds_read_u8 %6, %14\n\tds_read_u8 %7, %15\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(tb[0]), "=&v"(tb[1]), "=&v"(tb[2]), "=&v"(tb[3]), "=&v"(tb[4]), "=&v"(tb[5]), "=&v"(tb[6]), "=&v"(tb[7])
                         : "v"(adr[0]), "v"(adr[1]), "v"(adr[2]), "v"(adr[3]), "v"(adr[4]), "v"(adr[5]), "v"(adr[6]), "v"(adr[7])
                         : "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) word[k] = __ballot(tb[2 * k] < tb[2 * k + 1]);
        }
        else
        {
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            int t[2];
#pragma unroll
            for (int e = 0; e < 2; ++e)
            {
                const float pxf = pt[k][2 * e], pyf = pt[k][2 * e + 1];
                const int ry    = __float2int_rn(pxf * sn + pyf * cs);
                const int rx    = __float2int_rn(pxf * cs - pyf * sn);
                t[e]            = pb[pc + ry * (4 * PATCH_DW) + rx];
            }
            word[k] = __ballot(t[0] < t[1]);
        }
        }
        auto store_keypoint = [&](int s_, const u64 (&w)[4])
        {
            const float angle = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, angle_l), 16 * s_));
            if (lane == 0)
            {
                const int oi = offset + slot0 + s_;
                snk_keypoint kp;
                kp.x        = (float)kxv[s_] * lv.scale;
                kp.y        = (float)kyv[s_] * lv.scale;
                kp.size     = 31.0f * lv.scale;
                kp.angle    = angle;
                kp.response = sel_resp ? harris_unrank((u32)scv[s_]) : (float)(scv[s_] - 1);
                kp.octave   = l;
                kps[(long long)b * out_cap + oi] = kp;
                u64* d = desc + ((long long)b * out_cap + oi) * 4;
                d[0] = w[0];
                d[1] = w[1];
                d[2] = w[2];
                d[3] = w[3];
            }
        };
        if constexpr (DMA)
        {
#pragma unroll
            for (int k = 0; k < 4; ++k) wordv[s][k] = word[k];
            // every test byte of this buffer has been consumed by the ballots above: the patch of keypoint s + 2 may overwrite it
            if (s + 2 < DESC_KPW && valid[s + 2]) dma_patch(s + 2);
            if (s == DESC_KPW - 1 || !valid[s + 1 < DESC_KPW ? s + 1 : s])  // (the second operand is only read for s < DESC_KPW - 1)
            {
                // the last keypoint of the wavefront: all stores now
#pragma unroll
                for (int s2 = 0; s2 <= s; ++s2) store_keypoint(s2, wordv[s2]);
            }
        }
        else
            store_keypoint(s, word);
        (void)ky;
    }
}
}  // namespace
}  // namespace snk

using namespace snk;

struct snk_orb : HandleBase
{
    snk_orb_params params{};
    int width = 0, height = 0, max_batch = 0;
    bool configured = false;
    Layout lay{};
    DevBuf pyr[MAX_LEVELS];   // levels >= 1
    DevBuf blur[MAX_LEVELS];  // blurred levels (all)
    DevBuf tables;           // resize tables
    DevBuf cell_tab;         // FAST cell geometry
    DevBuf img0;             // level-0 staging for the host API
    DevBuf cand, cell_cnt, sel, sel_score, sel_cnt, cand_total, dist_queue;
    DevBuf cand_h, sel_resp, dist_h;  // "orb.response" = 1 only (reserved by the first call that runs under it)
    DevBuf out_kps;          // host-API staging: [count | pad to 64 B][cap keypoints][cap descriptors], fetched with ONE copy
    HostBuf h_out, h_img;    // pinned mirrors of out_kps / img0
    int pitch0_host = 0;
    size_t dist_lds = 0, dist_lds_small = 0;
    int dist_small_cap = 0;
    // optional per-stage timing with HIP events on the handle's stream (bench.py roofline leg)
    bool profiling = false;
    std::vector<std::array<hipEvent_t, 7>> ev_sets;  // pyramid | blur | fast | distribute | describe boundaries; [6] = start of the back half (staggered schedule)
    size_t ev_used = 0;
    // second stream: a batch is split in two halves whose launch chains overlap (the tail of one half's
    // launch runs beside the other half's kernels; latency-bound and VALU-bound stages share the CUs)
    static constexpr int MAX_PARTS = 4;
    hipStream_t stream2 = nullptr;  // non-null when the extra streams / events below exist
    hipStream_t extra[MAX_PARTS - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join_n[MAX_PARTS - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_join = nullptr;
    int split_min_batch = 8, parts = 1;  // launch chains per batch: 1 by default, snk_orb_set_chains / SNK_ORB_PARTS
    // staggered schedule (snk_orb_set_stagger / SNK_ORB_STAGGER): the batch is cut into `stagger` parts, the front halves run one
    // after the other on the handle's stream and every part's back half on the second stream as soon as its front half is done
    static constexpr int MAX_STAGGER = 16;
    int stagger = 0;
    hipEvent_t ev_front[MAX_STAGGER] = {};
    size_t part_ev[MAX_STAGGER]      = {};  // event set of each part of the current call (profiling)
};

static int compute_layout(snk_orb* o, int w, int h)
{
    const snk_orb_params& p = o->params;
    Layout& L               = o->lay;
    memset(&L, 0, sizeof(L));
    L.n_levels  = p.n_levels;
    L.level_cap = p.level_cap > 0 ? p.level_cap : DEFAULT_LEVEL_CAP;
    float scale[MAX_LEVELS];
    scale[0] = 1.0f;
    for (int l = 1; l < p.n_levels; ++l) scale[l] = scale[l - 1] * p.scale_factor;
    // ORB-SLAM2 constructor arithmetic (float)
    const float factor = 1.0f / p.scale_factor;
    float n_desired    = (float)p.nfeatures * (1.0f - factor) / (1.0f - (float)pow((double)factor, (double)p.n_levels));
    int sum            = 0;
    int cell_off = 0, slot_off = 0, tile_off = 0;
    for (int l = 0; l < p.n_levels; ++l)
    {
        LevelInfo& lv = L.lv[l];
        const float inv = 1.0f / scale[l];
        lv.w     = (int)lrintf((float)w * inv);
        lv.h     = (int)lrintf((float)h * inv);
        lv.scale = scale[l];
        if (l < p.n_levels - 1)
        {
            lv.nfeat = (int)lrintf(n_desired);
            sum += lv.nfeat;
            n_desired *= factor;
        }
        else
            lv.nfeat = p.nfeatures - sum > 0 ? p.nfeatures - sum : 0;
        const int width = lv.w - 2 * MIN_BORDER, height = lv.h - 2 * MIN_BORDER;
        lv.ncols = width > 0 ? width / CELL_W : 0;
        lv.nrows = height > 0 ? height / CELL_W : 0;
        if (lv.ncols < 1 || lv.nrows < 1 || lv.w < 2 * EDGE_THRESHOLD + 1 || lv.h < 2 * EDGE_THRESHOLD + 1)
        {
            lv.ncols = lv.nrows = 0;
            lv.wcell = lv.hcell = 1;
        }
        else
        {
            lv.wcell = (width + lv.ncols - 1) / lv.ncols;
            lv.hcell = (height + lv.nrows - 1) / lv.nrows;
        }
        lv.cell_off = cell_off;
        cell_off += lv.ncols * lv.nrows;
        int nroots = height > 0 ? (2 * width + height) / (2 * height) : 1;
        nroots     = nroots < 1 ? 1 : (nroots > 255 ? 255 : nroots);
        lv.nroots  = nroots;
        lv.slot_off = slot_off;
        lv.slot_cap = lv.nfeat + 3 > 4 * nroots ? lv.nfeat + 3 : 4 * nroots;
        slot_off += lv.slot_cap;
        lv.pitch      = (lv.w + 63) & ~63;
        lv.img_stride = (long long)lv.pitch * lv.h;
        lv.blur_stride = (long long)lv.pitch * ((lv.h + 3) & ~3);
        // streaming blur: balanced strips of <= 62 dwords, bands of 64 rows
        if (lv.w > 0 && lv.h > 0)
        {
            // Strips START on 64-byte boundaries of the row (stride 192 columns; the last strip takes what is left, up to the 244 columns a
            // wavefront can produce): a wavefront's row store then never shares a 64-byte sector with its neighbour strip's.  Round 5,
            // tools/probes/hbm_write_shapes.sh: rows written as 4 x 188-byte strips (the balanced layout of a 752-px level) reach 3.3 TB/s,
            // as 4 x 192-byte strips 5.1 TB/s.  SNK_ORB_BALANCED_STRIPS=1: the round-1..4 layout (equal strips of a multiple of 4 columns).
            static const bool balanced = getenv("SNK_ORB_BALANCED_STRIPS") != nullptr;
            if (balanced || lv.w <= 4 * SM_LANES_OUT)
            {
                lv.n_strips     = ceil_div(lv.w, 4 * SM_LANES_OUT);
                lv.strip_stride = ((ceil_div(lv.w, lv.n_strips) + 3) / 4) * 4;
            }
            else
            {
                lv.strip_stride = 192;
                lv.n_strips     = 1 + ceil_div(lv.w - 4 * SM_LANES_OUT, 192);
            }
            // The last strip's blurred store runs on to the end of the row pitch when its 61 lanes reach it (EuRoC levels 0 and 2: 576 + 192 =
            // 768, 384 + 192 = 576): whole 64-byte sectors instead of a ragged end; the padding columns hold reflected-border blur values
            // nobody reads.  +0.4 % end to end; one more 192-column strip wherever the rest exceeds 192 columns instead: -1.5 %
            // (profiles/r05/r05z_strip_store_end.json).
            lv.store_end = lv.w;
            if (!balanced && lv.w > 4 * SM_LANES_OUT && lv.pitch - (lv.n_strips - 1) * lv.strip_stride <= 4 * SM_LANES_OUT) lv.store_end = lv.pitch;
            lv.n_bands      = ceil_div(lv.h, SM_BH);
        }
        else  // a level scaled down to nothing (small image, many levels, large scale factor): no strips, no work
        {
            lv.w = lv.w > 0 ? lv.w : 0;
            lv.h = lv.h > 0 ? lv.h : 0;
            lv.n_strips = lv.n_bands = 0;
            lv.strip_stride = 4;
        }
        lv.unit_off     = tile_off;
        tile_off += lv.n_strips * lv.n_bands;
    }
    L.total_cells = cell_off;
    L.total_slots = slot_off;
    L.total_units = tile_off;
    {
        int mw = 1, mh = 1;
        for (int l = 0; l < p.n_levels; ++l)
            if (L.lv[l].ncols > 0)
            {
                mw = L.lv[l].wcell > mw ? L.lv[l].wcell : mw;
                mh = L.lv[l].hcell > mh ? L.lv[l].hcell : mh;
            }
        auto up16 = [](int v) { return (v + 15) & ~15; };
        L.f_tile_pitch_dw = (mw + 6 + 3 + 3) / 4;
        L.f_s_pitch       = (mw + 2 + 3) & ~3;
        // compile-time pitches (fast_kernel<3> / <4>) when the widest cell allows, see fast_quads()
        if (L.f_tile_pitch_dw <= 12 && L.f_s_pitch <= 40 && mh + 6 <= 128) { L.f_tile_pitch_dw = 12; L.f_s_pitch = 40; }
        else if (L.f_tile_pitch_dw <= 16 && L.f_s_pitch <= 64 && mh + 6 <= 128) { L.f_tile_pitch_dw = 16; L.f_s_pitch = 64; }
        const int tile_b  = up16((mh + 6) * L.f_tile_pitch_dw * 4);
        const int s_b     = up16((mh + 2) * L.f_s_pitch) + 16;  // zero-filled in 16-byte stores
        // survivors of the quick test kept at a time (fast_kernel scores and restarts the list when a cell has more); the
        // override is for the tests that force the spill path on ordinary images
        L.f_surv_cap = (mw * mh / 2 + 63) & ~63;  // measured: 0.60 / 0.55 / 0.50 / 0.485 / 0.54 ms at 320 / 384 / 512 / 640 / all pixels
        if (L.f_surv_cap < 256) L.f_surv_cap = 256;
        if (const char* e = getenv("SNK_ORB_FAST_SURV_CAP")) L.f_surv_cap = atoi(e) < 128 ? 128 : atoi(e);
        const int surv_b  = up16(L.f_surv_cap * 2);
        const int list_b  = up16(((mw + 1) / 2) * ((mh + 1) / 2) * 4);
        // the list of non-maximum-suppression survivors is written after the last read of the image tile: same bytes
        SNK_REQUIRE(list_b <= tile_b, "FAST: corner list larger than the tile it aliases");
        L.f_off_s    = tile_b;
        L.f_off_surv = L.f_off_s + s_b;
        L.f_off_list = 0;
        L.f_off_cnt  = L.f_off_surv + surv_b;
        L.f_lds_wave = L.f_off_cnt + 16;
    }
    return SNK_OK;
}

static void resize_tables(int src, int dst, int* ofs, int* w1)
{
    const double scale = (double)src / (double)dst;
    for (int d = 0; d < dst; ++d)
    {
        double f = ((double)d + 0.5) * scale - 0.5;
        int s    = (int)floor(f);
        f -= (double)s;
        if (s < 0)
        {
            s = 0;
            f = 0.0;
        }
        if (s >= src - 1)
        {
            s = src - 1;
            f = 0.0;
        }
        ofs[d] = s;
        w1[d]  = (int)lrint(f * 2048.0);
    }
}

// extra streams + fork / join events of the multi-chain schedules (best effort: without them every batch runs as one chain)
static bool ensure_extra_streams(snk_orb* o)
{
    if (o->stream2) return true;
    bool ok = hipEventCreateWithFlags(&o->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < snk_orb::MAX_PARTS - 1; ++i)
        ok = hipStreamCreateWithFlags(&o->extra[i], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&o->ev_join_n[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    o->stream2 = ok ? o->extra[0] : nullptr;
    return ok;
}

extern "C" {

int snk_orb_create(const snk_orb_params* params, int device, void* stream, snk_orb** out)
{
    SNK_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    SNK_REQUIRE(params != nullptr, "params is NULL");
    SNK_REQUIRE(params->n_levels >= 1 && params->n_levels <= MAX_LEVELS, "n_levels must be 1..16");
    SNK_REQUIRE(params->scale_factor > 1.0f, "scale_factor must be > 1");
    SNK_REQUIRE(params->nfeatures >= 1 && params->nfeatures <= 100000, "nfeatures must be 1..100000");
    SNK_REQUIRE(params->ini_th_fast >= 1 && params->ini_th_fast <= 254 && params->min_th_fast >= 1 &&
                    params->min_th_fast <= 254,
                "FAST thresholds must be 1..254");
    if (params->level_cap != 0)
    {
        const int c = params->level_cap;
        SNK_REQUIRE(c >= 256 && c <= 8192 && (c & (c - 1)) == 0, "level_cap must be a power of two in [256, 8192] (or 0)");
    }
    snk_orb* o = new snk_orb();
    o->params  = *params;
    int rc     = o->init(device, stream);
    if (rc != SNK_OK)
    {
        delete o;
        return rc;
    }
    // The extra streams / events of the multi-chain schedules are created when one is first asked for (ensure_extra_streams):
    // HIP spreads streams over a handful of hardware queues in creation order, and an extractor that never splits its batches
    // (the default, and every per-frame caller) should not take four of them -- the slots of a pipelined front-end would otherwise
    // land on the same hardware queue and run one behind the other.
    if (const char* e = getenv("SNK_ORB_PARTS"))
    {
        const int v = atoi(e);
        o->parts    = v < 1 ? 1 : (v > snk_orb::MAX_PARTS ? snk_orb::MAX_PARTS : v);
    }
    if (const char* e = getenv("SNK_ORB_STAGGER"))
    {
        const int v = atoi(e);
        o->stagger  = v >= 2 ? (v > snk_orb::MAX_STAGGER ? snk_orb::MAX_STAGGER : v) : 0;
    }
    if ((o->parts > 1 || o->stagger >= 2) && !ensure_extra_streams(o)) o->parts = 1, o->stagger = 0;
    *out = o;
    return SNK_OK;
}

int snk_orb_destroy(snk_orb* o)
{
    if (!o) return SNK_OK;
    (void)hipSetDevice(o->device);
    for (auto& b : o->pyr) b.release();
    for (auto& b : o->blur) b.release();
    o->tables.release();
    o->img0.release();
    o->cand.release();
    o->cell_cnt.release();
    o->cell_tab.release();
    o->sel.release();
    o->sel_score.release();
    o->sel_cnt.release();
    o->cand_total.release();
    o->dist_queue.release();
    o->cand_h.release();
    o->sel_resp.release();
    o->dist_h.release();
    o->out_kps.release();
    o->h_out.release();
    o->h_img.release();
    for (auto& e : o->ev_sets)
        for (auto& x : e) (void)hipEventDestroy(x);
    if (o->ev_fork) (void)hipEventDestroy(o->ev_fork);
    for (auto& e : o->ev_front)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : o->ev_join_n)
        if (e) (void)hipEventDestroy(e);
    for (auto& st : o->extra)
        if (st) (void)hipStreamDestroy(st);
    o->fini();
    delete o;
    return SNK_OK;
}

int snk_orb_sync(snk_orb* o)
{
    SNK_REQUIRE(o != nullptr, "orb is NULL");
    SNK_HIP_CHECK(hipStreamSynchronize(o->stream));
    return SNK_OK;
}

int snk_orb_configure(snk_orb* o, int width, int height, int max_batch)
{
    SNK_REQUIRE(o != nullptr, "orb is NULL");
    SNK_REQUIRE(width >= 1 && height >= 1 && width <= 16384 && height <= 16384, "image size must be 1..16384");
    SNK_REQUIRE(max_batch >= 1 && max_batch <= 65535, "max_batch must be 1..65535");
    SNK_HIP_CHECK(hipSetDevice(o->device));
    if (o->configured && o->width == width && o->height == height && o->max_batch >= max_batch) return SNK_OK;
    SNK_HIP_CHECK(hipStreamSynchronize(o->stream));
    int rc = compute_layout(o, width, height);
    if (rc != SNK_OK) return rc;
    Layout& L = o->lay;
    // resize tables
    size_t tab_ints = 0;
    for (int l = 1; l < L.n_levels; ++l)
        tab_ints += 2 * (size_t)(L.lv[l].w + L.lv[l].h) + (size_t)L.lv[l - 1].h + (size_t)L.lv[l - 1].n_strips + 1;
    if ((rc = o->tables.reserve((tab_ints + 4) * sizeof(int))) != SNK_OK) return rc;
    {
        int* host = (int*)malloc((tab_ints + 4) * sizeof(int));
        size_t at = 0;
        int* dev  = o->tables.as<int>();
        for (int l = 1; l < L.n_levels; ++l)
        {
            LevelInfo& lv = L.lv[l];
            LevelInfo& sv = L.lv[l - 1];
            lv.xofs = dev + at;
            lv.xw1  = dev + at + lv.w;
            int* xo = host + at;
            resize_tables(sv.w, lv.w, host + at, host + at + lv.w);
            at += 2 * (size_t)lv.w;
            lv.yofs = dev + at;
            lv.yw1  = dev + at + lv.h;
            int* yo = host + at;
            resize_tables(sv.h, lv.h, host + at, host + at + lv.h);
            at += 2 * (size_t)lv.h;
            // inverse maps used by the streaming pass of the SOURCE level l-1
            sv.ymap = dev + at;
            for (int r = 0; r < sv.h; ++r) host[at + r] = -1;
            for (int y = 0; y < lv.h; ++y) host[at + yo[y]] = y;  // source rows are distinct for scale > 1
            at += (size_t)sv.h;
            sv.strip_dx = dev + at;
            {
                // first destination dword (4 pixels) whose leading pixel samples from strip st or later
                const int n_dw = ceil_div(lv.w, 4);
                int g          = 0;
                for (int st = 0; st <= sv.n_strips; ++st)
                {
                    const int lim = st * sv.strip_stride;  // first source column of strip st
                    while (g < n_dw && xo[4 * g] < lim) ++g;
                    host[at + st] = st == sv.n_strips ? n_dw : g;
                }
            }
            at += (size_t)sv.n_strips + 1;
        }
        rc = copy_sync(o->tables.p, host, tab_ints * sizeof(int), hipMemcpyHostToDevice, o->stream);
        free(host);
        if (rc != SNK_OK) return rc;
    }
    for (int l = 1; l < L.n_levels; ++l)
    {
        if ((rc = o->pyr[l].reserve((size_t)L.lv[l].img_stride * max_batch + 64)) != SNK_OK) return rc;
        L.lv[l].base = o->pyr[l].as<u8>();
    }
    for (int l = 0; l < L.n_levels; ++l)
    {
        if ((rc = o->blur[l].reserve((size_t)L.lv[l].blur_stride * max_batch + 256)) != SNK_OK) return rc;
        L.lv[l].blur = o->blur[l].as<u8>();
    }
    {
        std::vector<int4> ct((size_t)(L.total_cells > 0 ? L.total_cells : 1));
        for (int l = 0; l < L.n_levels; ++l)
        {
            const LevelInfo& lv = L.lv[l];
            for (int c = 0; c < lv.ncols * lv.nrows; ++c)
            {
                const int ci = c / lv.ncols, cj = c - ci * lv.ncols;
                const int x0 = EDGE_THRESHOLD + cj * lv.wcell, y0 = EDGE_THRESHOLD + ci * lv.hcell;
                const int x1 = std::min(x0 + lv.wcell, lv.w - EDGE_THRESHOLD), y1 = std::min(y0 + lv.hcell, lv.h - EDGE_THRESHOLD);
                int4 e;
                e.x = x0;
                e.y = y0;
                e.z = (int)(((unsigned)(x1 - x0) & 0xFFFFu) | ((unsigned)(y1 - y0) << 16));
                e.w = l;
                ct[(size_t)lv.cell_off + c] = e;
            }
        }
        if ((rc = o->cell_tab.reserve(ct.size() * sizeof(int4))) != SNK_OK) return rc;
        if ((rc = copy_sync(o->cell_tab.p, ct.data(), ct.size() * sizeof(int4), hipMemcpyHostToDevice, o->stream)) != SNK_OK) return rc;
        L.cell_tab = o->cell_tab.as<int4>();
    }
    const size_t cells = (size_t)(L.total_cells > 0 ? L.total_cells : 1) * max_batch;
    const size_t slots = (size_t)(L.total_slots > 0 ? L.total_slots : 1) * max_batch;
    if ((rc = o->cand.reserve(cells * CELL_SLOTS * sizeof(u32))) != SNK_OK) return rc;
    if ((rc = o->cell_cnt.reserve(cells * sizeof(u16))) != SNK_OK) return rc;
    if ((rc = o->sel.reserve(slots * sizeof(u32))) != SNK_OK) return rc;
    if ((rc = o->sel_score.reserve(slots)) != SNK_OK) return rc;
    if ((rc = o->sel_cnt.reserve((size_t)max_batch * MAX_LEVELS * sizeof(int))) != SNK_OK) return rc;
    if ((rc = o->cand_total.reserve((size_t)max_batch * MAX_LEVELS * sizeof(int))) != SNK_OK) return rc;
    // distribute kernel dynamic LDS
    auto dist_lds_bytes = [](size_t cap) { return cap * 8 + cap / 2 * 8 + cap * 2 * 2 + cap + (cap + 16) + cap; };
    o->dist_small_cap = L.level_cap < 2048 ? L.level_cap : 2048;
    o->dist_lds       = dist_lds_bytes((size_t)L.level_cap);
    o->dist_lds_small = dist_lds_bytes((size_t)o->dist_small_cap);
    if ((rc = o->dist_queue.reserve(snk_orb::MAX_PARTS * ((size_t)max_batch * MAX_LEVELS + 1) * sizeof(int))) != SNK_OK) return rc;  // one per chain
    // process-wide, once per kernel, to the largest carve any handle can ask for (never per-handle sizes: two extractors
    // with different image sizes / level_cap would overwrite each other's limit)
    SNK_REQUIRE(4 * L.f_lds_wave <= LDS_MAX_BYTES, "FAST cells too large for the LDS (image too small for its cell grid?)");
    // the largest of: the 2048-candidate carve + the Harris ranks ("orb.response" = 1), the full-budget carve (small launches take it directly)
    if ((rc = set_max_lds_once(reinterpret_cast<const void*>(distribute_kernel), (int)std::max(dist_lds_bytes(2048) + 4 * 2048 + 16, dist_lds_bytes(8192)))) != SNK_OK) return rc;
    if ((rc = set_max_lds_once(reinterpret_cast<const void*>(distribute_large_kernel), (int)dist_lds_bytes(8192))) != SNK_OK) return rc;
    if ((rc = set_max_lds_once(fast_kernel_for(L, false), LDS_MAX_BYTES)) != SNK_OK) return rc;
    if ((rc = set_max_lds_once(fast_kernel_for(L, true), LDS_MAX_BYTES)) != SNK_OK) return rc;
    // host-API staging (snk_orb_detect): sized here so that the per-image call allocates nothing
    {
        const int dpitch = (width + 63) & ~63;
        const int cap    = L.total_slots > 0 ? L.total_slots : 1;
        if ((rc = o->img0.reserve((size_t)dpitch * height + 64)) != SNK_OK) return rc;
        if ((rc = o->out_kps.reserve((size_t)cap * (sizeof(snk_keypoint) + 32) + 128)) != SNK_OK) return rc;
        if ((rc = o->h_out.reserve((size_t)cap * (sizeof(snk_keypoint) + 32) + 128)) != SNK_OK) return rc;
        if ((rc = o->h_img.reserve((size_t)dpitch * height)) != SNK_OK) return rc;
    }
    o->width      = width;
    o->height     = height;
    o->max_batch  = max_batch;
    o->configured = true;
    return SNK_OK;
}

int snk_orb_max_keypoints(const snk_orb* o, int* out)
{
    SNK_REQUIRE(o != nullptr && out != nullptr, "NULL argument");
    if (!o->configured)
    {
        set_error("snk_orb_configure has not been called");
        return SNK_ERR_NOT_CONFIGURED;
    }
    *out = o->lay.total_slots;
    return SNK_OK;
}

// One launch chain over images [b0, b0 + batch) of a call on stream `st`: every per-image buffer is indexed
// relative to the chain's first image, so the bases are simply advanced by b0 images.
// stages: 1 = the front half (pyramid / blur passes + FAST cells), 2 = the back half (distribution + descriptors), 3 = both
// Layout of the blurred planes for this process (level_kernel writes it, describe_kernel reads it -- both ask here): 0 = row-major (the
// default), 2 = tiled 32 x 4 (SNK_ORB_BLUR_TILED=1, an experiment of round 6: built, bit-exact, MEASURED SLOWER -- describe_kernel
// 1.39 -> 1.32 ms, but level_kernel's stores, 16-byte pieces of 47 lines per row instead of two whole lines, cost 1.50 -> 1.74 ms:
// 191.6 -> 187.9 k frames/s, profiles/r06/r06p_ab_blur_tiled_negative.txt), 1 = no blurred plane (SNK_ORB_BLUR_IN_DESCRIBE=1).
static int orb_blur_mode()
{
    static const int mode = getenv("SNK_ORB_BLUR_IN_DESCRIBE") ? 1
                            : (getenv("SNK_ORB_BLUR_TILED") && !getenv("SNK_ORB_DESC_NO_DMA") && !getenv("SNK_ORB_DESC_FAKE")) ? 2 : 0;
    return mode;
}

static int run_part(snk_orb* o, hipStream_t st, int part, int b0, const u8* images_dev, int pitch, long long image_stride,
                    int batch, snk_keypoint* kps_dev, uint64_t* desc_dev, int32_t* n_dev, int out_cap, int stages = 3)
{
    Layout L = o->lay;
    for (int l = 0; l < L.n_levels; ++l)
    {
        if (L.lv[l].base) L.lv[l].base += (long long)b0 * L.lv[l].img_stride;
        if (L.lv[l].blur) L.lv[l].blur += (long long)b0 * L.lv[l].blur_stride;
    }
    images_dev += (long long)b0 * image_stride;
    kps_dev += (long long)b0 * out_cap;
    desc_dev += (long long)b0 * out_cap * 4;
    n_dev += b0;
    u32* d_cand     = o->cand.as<u32>() + (size_t)b0 * L.total_cells * CELL_SLOTS;
    u16* d_cellcnt  = o->cell_cnt.as<u16>() + (size_t)b0 * L.total_cells;
    u32* d_sel      = o->sel.as<u32>() + (size_t)b0 * L.total_slots;
    u8* d_selscore  = o->sel_score.as<u8>() + (size_t)b0 * L.total_slots;
    int* d_selcnt   = o->sel_cnt.as<int>() + (size_t)b0 * MAX_LEVELS;
    int* d_candtot  = o->cand_total.as<int>() + (size_t)b0 * MAX_LEVELS;
    int* d_queue    = o->dist_queue.as<int>() + (size_t)part * ((size_t)o->max_batch * MAX_LEVELS + 1);
    // "orb.response" = 1 (snk_set_definition): Harris response of the FAST candidates ranks the points of a quadtree node and is the
    // keypoints' response.  Its scratch is reserved by the first call that runs under it (a growing reserve synchronises the device).
    const bool harris = definition(DEF_ORB_RESPONSE) == 1;
    u32 *d_candh = nullptr, *d_selresp = nullptr, *d_disth = nullptr;
    const int large_workers = L.n_levels * batch < 256 ? L.n_levels * batch : 256;
    if (harris)
    {
        const size_t cells = (size_t)(L.total_cells > 0 ? L.total_cells : 1) * o->max_batch;
        const size_t slots = (size_t)(L.total_slots > 0 ? L.total_slots : 1) * o->max_batch;
        int rc;
        if ((rc = o->cand_h.reserve(cells * CELL_SLOTS * sizeof(u32))) != SNK_OK) return rc;
        if ((rc = o->sel_resp.reserve(slots * sizeof(u32))) != SNK_OK) return rc;
        if (o->dist_small_cap < L.level_cap &&
            (rc = o->dist_h.reserve((size_t)snk_orb::MAX_PARTS * 256 * L.level_cap * sizeof(u32))) != SNK_OK)
            return rc;
        d_candh   = o->cand_h.as<u32>() + (size_t)b0 * L.total_cells * CELL_SLOTS;
        d_selresp = o->sel_resp.as<u32>() + (size_t)b0 * L.total_slots;
        d_disth   = o->dist_h.as<u32>() ? o->dist_h.as<u32>() + (size_t)part * 256 * L.level_cap : nullptr;
    }
    std::array<hipEvent_t, 7>* ev = nullptr;
    if (o->profiling)
    {
        if (stages & 1)
        {
            if (o->ev_used == o->ev_sets.size())
            {
                std::array<hipEvent_t, 7> e{};
                for (auto& x : e) SNK_HIP_CHECK(hipEventCreate(&x));
                o->ev_sets.push_back(e);
            }
            o->part_ev[part] = o->ev_used++;
        }
        ev = &o->ev_sets[o->part_ev[part]];
        if (stages & 1) SNK_HIP_CHECK(hipEventRecord((*ev)[0], st));
    }
    if (stages & 1)
    {
    // one streaming pass per level: blur of level l + down-scale to level l+1 (the chain makes the
    // passes sequential; every level is read once)
    const int aligned0 = (reinterpret_cast<uintptr_t>(images_dev) % 4 == 0 && pitch % 4 == 0 && image_stride % 4 == 0) ? 1 : 0;
    static const bool unfused_env = getenv("SNK_ORB_UNFUSED") != nullptr;  // A/B: stand-alone resize_kernel + blur-only passes
    const bool fused_ok = o->params.scale_factor <= 2.0f && !unfused_env;  // the in-stream down-scale reaches 3 * scale + 1 columns right
    // levels below 8 x 8 (deep levels of small images) cannot hold a feature and are outside the domain of
    // the streaming pass's reflect-101 halo: they are only down-scaled (stand-alone kernel), never blurred
    auto tiny = [&](int l) { return L.lv[l].w < 8 || L.lv[l].h < 8; };
    for (int l = 0; l < L.n_levels; ++l)
    {
        const LevelInfo& lv = L.lv[l];
        const bool fused    = fused_ok && !tiny(l);
        if (lv.w <= 0 || lv.h <= 0) continue;  // a level scaled down to nothing (small image, many levels): no pixels, no cells
        if (l > 0 && (!fused_ok || tiny(l - 1)))
        {
            const LevelInfo& sv = L.lv[l - 1];
            dim3 grid(ceil_div(ceil_div(lv.w, 4), 256), lv.h, batch);
            hipLaunchKernelGGL(resize_kernel, grid, dim3(256), 0, st, l == 1 ? images_dev : sv.base, l == 1 ? pitch : sv.pitch,
                               l == 1 ? image_stride : sv.img_stride, sv.w, sv.h, lv.base, lv.pitch, lv.img_stride, lv.w, lv.h,
                               lv.xofs, lv.xw1, lv.yofs, lv.yw1);
            SNK_LAUNCH_CHECK();
        }
        if (tiny(l)) continue;
        {
            // rows per band: 64 when the launch fills the chip's 8192 wavefront slots anyway, else 22, else 8 (per-frame calls)
            static const int bh_env = getenv("SNK_ORB_LEVEL_BH") ? atoi(getenv("SNK_ORB_LEVEL_BH")) : 0;  // A/B, tests: 64 / 22 / 8
            const long long strips = (long long)lv.n_strips * batch;
            int bh = strips * ceil_div(lv.h, 64) >= 2048 ? 64 : (strips * ceil_div(lv.h, 22) >= 2048 ? 22 : 8);
            if (bh_env == 64 || bh_env == 22 || bh_env == 8) bh = bh_env;
            const int n_bands = ceil_div(lv.h, bh);
            const int gx = ceil_div(lv.n_strips * n_bands, 4);
            const bool al = !(l == 0 && !aligned0);
            auto lk = bh == 64 ? (al ? level_kernel<true, 64> : level_kernel<false, 64>)
                               : (bh == 22 ? (al ? level_kernel<true, 22> : level_kernel<false, 22>) : (al ? level_kernel<true, 8> : level_kernel<false, 8>));
            hipLaunchKernelGGL(lk, xcd_grid(gx, batch), dim3(256), 0, st, L, l, images_dev, pitch, image_stride,
                               fused && l + 1 < L.n_levels && L.lv[l + 1].w > 0 && L.lv[l + 1].h > 0 ? 1 : 0, gx, batch, n_bands,
                               orb_blur_mode());
        }
        SNK_LAUNCH_CHECK();
    }
    if (ev) SNK_HIP_CHECK(hipEventRecord((*ev)[1], st));
    SNK_LAUNCH_CHECK();
    if (ev) SNK_HIP_CHECK(hipEventRecord((*ev)[2], st));
    if (L.total_cells > 0)
    {
        // cells per wavefront: 1 for small launches (keep the chip filled), more once there are plenty of workgroups
        static const int cpw_env = getenv("SNK_ORB_FAST_CPW") ? atoi(getenv("SNK_ORB_FAST_CPW")) : 0;
        const long long cell_waves = (long long)L.total_cells * batch;
        int cpw = FAST_CPW_DEFAULT;
        while (cpw > 1 && cell_waves / cpw < 256 * 32 * 4) cpw >>= 1;  // at least four rounds of the chip's 8192 wavefront slots
        if (cpw_env >= 1 && cpw_env <= 64) cpw = cpw_env;               // forced (tests, measurements): whatever the launch size
        const int gx = ceil_div(L.total_cells, 4 * cpw);
        const int fq = fast_quads(L);
        static const int fast_stop = getenv("SNK_ORB_FAST_STOP") ? atoi(getenv("SNK_ORB_FAST_STOP")) : 0;  // timing experiments
        auto fk      = cpw > 1 ? (fq == 3 ? fast_kernel<3, true> : (fq == 4 ? fast_kernel<4, true> : fast_kernel<0, true>))
                               : (fq == 3 ? fast_kernel<3, false> : (fq == 4 ? fast_kernel<4, false> : fast_kernel<0, false>));
        // occupancy shaping (experiment, round 4): a workgroup that asks for at least SNK_ORB_FAST_LDS_WG bytes of LDS caps how many
        // FAST workgroups a CU holds, so that back-half workgroups of another stream (staggered schedule) find room beside them
        static const size_t lds_wg_env = getenv("SNK_ORB_FAST_LDS_WG") ? (size_t)atoll(getenv("SNK_ORB_FAST_LDS_WG")) : 0;
        const size_t fast_lds = std::max((size_t)4 * L.f_lds_wave, std::min(lds_wg_env, (size_t)LDS_MAX_BYTES));
        hipLaunchKernelGGL(fk, xcd_grid(gx, batch), dim3(256), fast_lds, st, L, images_dev, pitch,
                           image_stride, aligned0, o->params.ini_th_fast, o->params.min_th_fast, d_cand, d_cellcnt, gx, batch, fast_stop,
                           cpw, stages == 3 ? d_queue : (int*)nullptr);
        SNK_LAUNCH_CHECK();
    }
    if (ev) SNK_HIP_CHECK(hipEventRecord((*ev)[3], st));
    }
    if (!(stages & 2)) return SNK_OK;
    if (ev) SNK_HIP_CHECK(hipEventRecord((*ev)[6], st));
    {
    const int aligned0 = (reinterpret_cast<uintptr_t>(images_dev) % 4 == 0 && pitch % 4 == 0 && image_stride % 4 == 0) ? 1 : 0;
    // the queue's counter: reset by fast_kernel when this call enqueues the whole chain on one stream (the usual case: one launch less
    // per frame); the split schedules (front and back halves on different streams, chain slots shared by parts) keep the fill
    if (stages != 3 || L.total_cells <= 0) SNK_HIP_CHECK(hipMemsetAsync(d_queue, 0, sizeof(int), st));
    // SNK_ORB_DIST_TIMING=1 (diagnostic): cycle sums per phase and level, printed after a synchronisation
    static const bool dist_timing = getenv("SNK_ORB_DIST_TIMING") != nullptr;
    unsigned long long* d_dbg = nullptr;
    if (dist_timing)
    {
        SNK_HIP_CHECK(hipMalloc(&d_dbg, MAX_LEVELS * 16 * sizeof(unsigned long long)));
        SNK_HIP_CHECK(hipMemsetAsync(d_dbg, 0, MAX_LEVELS * 16 * sizeof(unsigned long long), st));
    }
    if (harris && L.total_cells > 0)
    {
        static const bool per_cell = getenv("SNK_ORB_HARRIS_PER_CELL") != nullptr;  // A/B: one wavefront per cell, one lane per slot (round 5)
        if (per_cell)
        {
            const int gxh = ceil_div(L.total_cells, 4);
            hipLaunchKernelGGL(harris_kernel, xcd_grid(gxh, batch), dim3(256), 0, st, L, images_dev, pitch, image_stride, aligned0, d_cand, d_cellcnt,
                               d_candh, gxh, batch);
        }
        else
        {
            const int gxh = ceil_div(ceil_div(L.total_cells, HARRIS_NC), 4);
            hipLaunchKernelGGL(harris_dense_kernel, xcd_grid(gxh, batch), dim3(256), 0, st, L, images_dev, pitch, image_stride, aligned0, d_cand,
                               d_cellcnt, d_candh, gxh, batch);
        }
        SNK_LAUNCH_CHECK();
    }
    // A launch of a few (image, level) workgroups (the per-frame calls) gives every workgroup the full-budget carve at once: nothing can
    // be queued, the drain launch is skipped (one launch less in the per-frame chain; the carve only limits workgroups per CU, and these
    // launches do not fill the chip).  Not under "orb.response" = 1: the Harris ranks do not fit beside the full carve.
    const bool one_dist_launch = !harris && o->dist_small_cap < L.level_cap && (long long)L.n_levels * batch <= 128;
    hipLaunchKernelGGL(distribute_kernel, dim3(L.n_levels, batch), dim3(DIST_THREADS),
                       one_dist_launch ? o->dist_lds : o->dist_lds_small + (harris ? 4 * (size_t)o->dist_small_cap + 16 : 0), st, L,
                       one_dist_launch ? L.level_cap : o->dist_small_cap, d_cand, d_cellcnt, d_sel, d_selscore,
                       d_selcnt, d_candtot, d_queue, d_dbg, d_candh, d_selresp);
    SNK_LAUNCH_CHECK();
    if (dist_timing)
    {
        unsigned long long h[MAX_LEVELS * 16];
        SNK_HIP_CHECK(hipStreamSynchronize(st));
        SNK_HIP_CHECK(hipMemcpy(h, d_dbg, sizeof(h), hipMemcpyDeviceToHost));
        SNK_HIP_CHECK(hipFree(d_dbg));
        for (int l = 0; l < L.n_levels; ++l)
        {
            fprintf(stderr, "[dist timing] level %d (avg candidates %.0f, %.2f careful rounds), cycles per workgroup:", l,
                    (double)h[l * 16 + 15] / batch, (double)h[l * 16 + 14] / batch);
            for (int ph = 0; ph < 13; ++ph) fprintf(stderr, " %.0f", (double)h[l * 16 + ph] / batch);
            fprintf(stderr, "\n");
        }
    }
    if (o->dist_small_cap < L.level_cap && !one_dist_launch)
    {
        hipLaunchKernelGGL(distribute_large_kernel, dim3(large_workers), dim3(DIST_THREADS), o->dist_lds, st, L,
                           d_cand, d_cellcnt, d_sel, d_selscore,
                           d_selcnt, d_candtot, d_queue, d_candh, d_selresp, d_disth);
        SNK_LAUNCH_CHECK();
    }
    if (ev) SNK_HIP_CHECK(hipEventRecord((*ev)[4], st));
    int max_slot = 1;
    for (int l = 0; l < L.n_levels; ++l) max_slot = L.lv[l].slot_cap > max_slot ? L.lv[l].slot_cap : max_slot;
    {
        const int nb = batch >= 16 ? 8 * ceil_div(batch, 8) : batch;
        static const bool blur_in = getenv("SNK_ORB_BLUR_IN_DESCRIBE") != nullptr;  // experiment (round 5), see describe_kernel<true>
        // round 6: the blurred patch by LDS-DMA (describe_kernel<false, true>; same-box A/B 1.457 -> 1.398 ms per 2048 images,
        // profiles/r06/r06c_ab_describe_lds_dma.txt).  SNK_ORB_DESC_NO_DMA=1: the register path
        static const bool dma = getenv("SNK_ORB_DESC_NO_DMA") == nullptr;
        const int dfake       = getenv("SNK_ORB_DESC_FAKE") ? atoi(getenv("SNK_ORB_DESC_FAKE")) : 0;
        const int gx          = ceil_div(max_slot, 4 * DESC_KPW);
        auto dk = describe_kernel<false, false, false>;
        if (blur_in) dk = describe_kernel<true, false, false>;
        else if (dma && orb_blur_mode() == 2) dk = describe_kernel<false, true, true>;
        else if (dma) dk = describe_kernel<false, true, false>;
        hipLaunchKernelGGL(dk, dim3(gx * L.n_levels * nb), dim3(256), 0, st, L, images_dev, pitch, image_stride, aligned0, d_sel, d_selscore, d_selcnt,
                           kps_dev, (u64*)desc_dev, n_dev, out_cap, gx, batch, dfake, d_selresp);
    }
    SNK_LAUNCH_CHECK();
    if (ev) SNK_HIP_CHECK(hipEventRecord((*ev)[5], st));
    }
    return SNK_OK;
}


// A batch of at least split_min_batch images runs as two half-batch chains on two streams (fork / join
// with events on the caller-visible stream); smaller batches as one chain.
static int run_pipeline(snk_orb* o, const u8* images_dev, int pitch, long long image_stride, int batch,
                        snk_keypoint* kps_dev, uint64_t* desc_dev, int32_t* n_dev, int out_cap)
{
    static const bool one_stream = getenv("SNK_ORB_ONE_STREAM") != nullptr;
    // Launch chains: one by default.  Two half-batch chains on two streams (snk_orb_set_chains) overlap the tails of one chain's launches
    // with the other's kernels: 186.0 / 192.1 / 192.9 / 188.4 k frames/s with 1 / 2 / 3 / 4 chains in one run, 187.6 k with 2 in the next
    // (profiles/r05/r05o_chains.txt, r05p_bench_two_chains_by_default.txt) -- the chains start together and mostly run the same stage at
    // the same time, so the gain is the tails only and a kernel's duration is then measured under a concurrent copy of itself.  Tried as
    // the default for big batches in round 5 and taken back.  Per-frame launches never split (and never create the extra streams).
    const int want_parts = o->parts;
    const bool want_split = !one_stream && batch >= o->split_min_batch && (want_parts > 1 || (o->stagger >= 2 && batch >= 2 * o->stagger));
    if (!want_split || !ensure_extra_streams(o))
        return run_part(o, o->stream, 0, 0, images_dev, pitch, image_stride, batch, kps_dev, desc_dev, n_dev, out_cap);
    if (o->stagger >= 2 && batch >= 2 * o->stagger)
    {
        // Front halves (VALU-bound: level passes, FAST) back to back on the handle's stream; the back half of part p (latency-bound:
        // distribution, descriptors) on the second stream beside the front half of part p + 1.  Parts are disjoint image ranges
        // with their own scratch, the only ordering is front(p) -> back(p).
        // INVARIANT: parts p and p + MAX_PARTS share the chain slot p % MAX_PARTS, i.e. one distribute queue (d_queue) and one
        // profiling event set.  Both are touched by the BACK half only (stages & 2 in run_part), and every back half is enqueued on
        // the single stream `sb`, so two users of a slot never overlap.  Moving back halves onto more than one stream requires
        // sizing dist_queue / part_ev by MAX_STAGGER and passing `p` as the part index.
        const int P = o->stagger;
        hipStream_t sf = o->stream, sb = o->extra[0];
        SNK_HIP_CHECK(hipEventRecord(o->ev_fork, sf));
        SNK_HIP_CHECK(hipStreamWaitEvent(sb, o->ev_fork, 0));
        int b0 = 0;
        for (int p = 0; p < P; ++p)
        {
            const int nb = (batch - b0) / (P - p);
            if (!o->ev_front[p]) SNK_HIP_CHECK(hipEventCreateWithFlags(&o->ev_front[p], hipEventDisableTiming));
            int rc = run_part(o, sf, p % snk_orb::MAX_PARTS, b0, images_dev, pitch, image_stride, nb, kps_dev, desc_dev, n_dev, out_cap, 1);
            if (rc != SNK_OK) return rc;
            SNK_HIP_CHECK(hipEventRecord(o->ev_front[p], sf));
            SNK_HIP_CHECK(hipStreamWaitEvent(sb, o->ev_front[p], 0));
            rc = run_part(o, sb, p % snk_orb::MAX_PARTS, b0, images_dev, pitch, image_stride, nb, kps_dev, desc_dev, n_dev, out_cap, 2);
            if (rc != SNK_OK) return rc;
            b0 += nb;
        }
        SNK_HIP_CHECK(hipEventRecord(o->ev_join_n[0], sb));
        SNK_HIP_CHECK(hipStreamWaitEvent(o->stream, o->ev_join_n[0], 0));
        return SNK_OK;
    }
    const int parts = want_parts < batch ? want_parts : batch;
    if (parts <= 1) return run_part(o, o->stream, 0, 0, images_dev, pitch, image_stride, batch, kps_dev, desc_dev, n_dev, out_cap);
    SNK_HIP_CHECK(hipEventRecord(o->ev_fork, o->stream));
    int b0 = 0;
    for (int p = 0; p < parts; ++p)
    {
        const int nb   = (batch - b0) / (parts - p);
        hipStream_t st = p == 0 ? o->stream : o->extra[p - 1];
        if (p > 0) SNK_HIP_CHECK(hipStreamWaitEvent(st, o->ev_fork, 0));
        const int rc = run_part(o, st, p, b0, images_dev, pitch, image_stride, nb, kps_dev, desc_dev, n_dev, out_cap);
        if (rc != SNK_OK) return rc;
        if (p > 0)
        {
            SNK_HIP_CHECK(hipEventRecord(o->ev_join_n[p - 1], st));
            SNK_HIP_CHECK(hipStreamWaitEvent(o->stream, o->ev_join_n[p - 1], 0));
        }
        b0 += nb;
    }
    return SNK_OK;
}

int snk_orb_detect_batch_dev(snk_orb* o, const uint8_t* images_dev, int pitch, size_t image_stride, int batch,
                             snk_keypoint* kps_dev, uint64_t* desc_dev, int32_t* n_dev, int out_cap)
{
    SNK_REQUIRE(o != nullptr, "orb is NULL");
    if (!o->configured)
    {
        set_error("snk_orb_configure has not been called");
        return SNK_ERR_NOT_CONFIGURED;
    }
    SNK_REQUIRE(batch >= 0 && batch <= o->max_batch, "batch exceeds the configured max_batch");
    SNK_REQUIRE(images_dev && kps_dev && desc_dev && n_dev, "NULL device buffer");
    SNK_REQUIRE(pitch >= o->width, "pitch smaller than the image width");
    SNK_REQUIRE(batch <= 1 || image_stride >= (size_t)pitch * (size_t)o->height, "image_stride smaller than pitch * height");
    SNK_REQUIRE(out_cap >= 1, "out_cap must be >= 1");
    if (batch == 0) return SNK_OK;
    SNK_HIP_CHECK(hipSetDevice(o->device));
    return run_pipeline(o, images_dev, pitch, (long long)image_stride, batch, kps_dev, desc_dev, n_dev, out_cap);
}

int snk_orb_detect(snk_orb* o, const uint8_t* img, int w, int h, int pitch, snk_keypoint* kps, uint64_t (*desc)[4],
                   int capacity, int* n_out)
{
    SNK_REQUIRE(o != nullptr, "orb is NULL");
    SNK_REQUIRE(n_out != nullptr, "n_out is NULL");
    *n_out = 0;
    SNK_REQUIRE(img != nullptr && w >= 1 && h >= 1 && pitch >= w, "bad image");
    SNK_REQUIRE(capacity >= 0 && (capacity == 0 || (kps && desc)), "bad output buffers");
    int rc;
    if (!o->configured || o->width != w || o->height != h)
        if ((rc = snk_orb_configure(o, w, h, o->max_batch > 0 ? o->max_batch : 1)) != SNK_OK) return rc;
    SNK_HIP_CHECK(hipSetDevice(o->device));
    const int dpitch = (w + 63) & ~63;
    const int cap    = o->lay.total_slots > 0 ? o->lay.total_slots : 1;
    // staging was sized by snk_orb_configure: [count, 64 B][cap keypoints][cap descriptors] on the device, a pinned
    // mirror on the host.  One upload, the launch chain, ONE download, one synchronisation -- all on the handle's stream.
    u8* dbase            = o->out_kps.as<u8>();
    snk_keypoint* d_kps  = reinterpret_cast<snk_keypoint*>(dbase + 64);
    const size_t desc_at = (64 + (size_t)cap * sizeof(snk_keypoint) + 63) & ~(size_t)63;
    uint64_t* d_desc     = reinterpret_cast<uint64_t*>(dbase + desc_at);
    const size_t out_len = desc_at + (size_t)cap * 32;
    for (int r = 0; r < h; ++r) memcpy(o->h_img.as<u8>() + (size_t)r * dpitch, img + (size_t)r * pitch, (size_t)w);
    SNK_HIP_CHECK(hipMemcpyAsync(o->img0.p, o->h_img.p, (size_t)dpitch * h, hipMemcpyHostToDevice, o->stream));
    rc = run_pipeline(o, o->img0.as<u8>(), dpitch, (long long)dpitch * h, 1, d_kps, d_desc, reinterpret_cast<int32_t*>(dbase), cap);
    if (rc != SNK_OK) return rc;
    SNK_HIP_CHECK(hipMemcpyAsync(o->h_out.p, dbase, out_len, hipMemcpyDeviceToHost, o->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(o->stream));
    const int n = *o->h_out.as<int32_t>();
    if (n > capacity)
    {
        set_error("capacity %d too small for %d keypoints (see snk_orb_max_keypoints)", capacity, n);
        *n_out = n;
        return SNK_ERR_CAPACITY;
    }
    if (n > 0)
    {
        memcpy(kps, o->h_out.as<u8>() + 64, (size_t)n * sizeof(snk_keypoint));
        memcpy(desc, o->h_out.as<u8>() + desc_at, (size_t)n * 32);
    }
    *n_out = n;
    return SNK_OK;
}

int snk_orb_set_profiling(snk_orb* o, int enable)
{
    SNK_REQUIRE(o != nullptr, "orb is NULL");
    SNK_HIP_CHECK(hipSetDevice(o->device));
    SNK_HIP_CHECK(hipStreamSynchronize(o->stream));
    o->profiling = enable != 0;
    o->ev_used   = 0;
    return SNK_OK;
}

int snk_orb_set_chains(snk_orb* o, int chains)
{
    SNK_REQUIRE(o != nullptr, "orb is NULL");
    SNK_REQUIRE(chains >= 1 && chains <= snk_orb::MAX_PARTS, "chains must be 1..4");
    SNK_HIP_CHECK(hipSetDevice(o->device));
    if (chains > 1 && !ensure_extra_streams(o))
    {
        set_error("the extra streams could not be created");
        return SNK_ERR_HIP;
    }
    SNK_HIP_CHECK(hipSetDevice(o->device));
    SNK_HIP_CHECK(hipStreamSynchronize(o->stream));
    o->parts = chains;
    return SNK_OK;
}

int snk_orb_set_stagger(snk_orb* o, int parts)
{
    SNK_REQUIRE(o != nullptr, "orb is NULL");
    SNK_REQUIRE(parts == 0 || (parts >= 2 && parts <= snk_orb::MAX_STAGGER), "parts must be 0 (off) or 2..16");
    SNK_HIP_CHECK(hipSetDevice(o->device));
    if (parts > 1 && !ensure_extra_streams(o))
    {
        set_error("the extra streams could not be created");
        return SNK_ERR_HIP;
    }
    SNK_HIP_CHECK(hipSetDevice(o->device));
    SNK_HIP_CHECK(hipStreamSynchronize(o->stream));
    o->stagger = parts;
    return SNK_OK;
}

int snk_orb_stage_times(snk_orb* o, float* ms /* 5: pyramid, blur, fast, distribute, describe */, int* n_calls)
{
    SNK_REQUIRE(o != nullptr && ms != nullptr && n_calls != nullptr, "NULL argument");
    SNK_HIP_CHECK(hipSetDevice(o->device));
    SNK_HIP_CHECK(hipStreamSynchronize(o->stream));  // joins the second stream's chain as well
    for (int k = 0; k < 5; ++k) ms[k] = 0.0f;
    for (size_t i = 0; i < o->ev_used; ++i)
        for (int k = 0; k < 5; ++k)
        {
            float t = 0.0f;
            // distribution: from the start of the back half ([6], its own stream in the staggered schedule) to its end
            SNK_HIP_CHECK(hipEventElapsedTime(&t, o->ev_sets[i][k == 3 ? 6 : k], o->ev_sets[i][k + 1]));
            ms[k] += t;
        }
    *n_calls   = (int)o->ev_used;
    o->ev_used = 0;
    return SNK_OK;
}

int snk_orb_debug_fetch(snk_orb* o, int what, int image, int level, void* out, size_t cap_bytes, size_t* n_bytes)
{
    SNK_REQUIRE(o != nullptr && n_bytes != nullptr, "NULL argument");
    *n_bytes = 0;
    if (!o->configured)
    {
        set_error("snk_orb_configure has not been called");
        return SNK_ERR_NOT_CONFIGURED;
    }
    const Layout& L = o->lay;
    SNK_REQUIRE(level >= 0 && level < L.n_levels && image >= 0 && image < o->max_batch, "bad image / level");
    SNK_HIP_CHECK(hipSetDevice(o->device));
    SNK_HIP_CHECK(hipStreamSynchronize(o->stream));
    const LevelInfo& lv = L.lv[level];
    const void* src     = nullptr;
    size_t bytes        = 0;
    switch (what)
    {
        case SNK_ORB_DEBUG_PYRAMID:  // level >= 1: h rows of `pitch` bytes
            SNK_REQUIRE(level >= 1, "level 0 is the caller's image");
            src   = lv.base + (long long)image * lv.img_stride;
            bytes = (size_t)lv.img_stride;
            break;
        case SNK_ORB_DEBUG_CELL_COUNTS:
            src   = o->cell_cnt.as<u16>() + (long long)image * L.total_cells + lv.cell_off;
            bytes = (size_t)lv.ncols * lv.nrows * sizeof(u16);
            break;
        case SNK_ORB_DEBUG_CELL_CANDIDATES:
            src   = o->cand.as<u32>() + ((long long)image * L.total_cells + lv.cell_off) * CELL_SLOTS;
            bytes = (size_t)lv.ncols * lv.nrows * CELL_SLOTS * sizeof(u32);
            break;
        case SNK_ORB_DEBUG_SELECTED:
            src   = o->sel.as<u32>() + (long long)image * L.total_slots + lv.slot_off;
            bytes = (size_t)lv.slot_cap * sizeof(u32);
            break;
        case SNK_ORB_DEBUG_SELECTED_COUNT:
            src   = o->sel_cnt.as<int>() + image * MAX_LEVELS + level;
            bytes = sizeof(int);
            break;
        case SNK_ORB_DEBUG_LEVEL_INFO:
        {
            int info[8] = {lv.w, lv.h, lv.pitch, lv.ncols, lv.nrows, lv.wcell, lv.hcell, lv.nfeat};
            SNK_REQUIRE(cap_bytes >= sizeof(info), "buffer too small");
            memcpy(out, info, sizeof(info));
            *n_bytes = sizeof(info);
            return SNK_OK;
        }
        default: SNK_REQUIRE(false, "unknown debug selector");
    }
    SNK_REQUIRE(out != nullptr && cap_bytes >= bytes, "buffer too small");
    if (int rc = copy_sync(out, src, bytes, hipMemcpyDeviceToHost, o->stream)) return rc;
    *n_bytes = bytes;
    return SNK_OK;
}
}
