// 256-bit Hamming matchers for gfx950: brute-force kNN-2 (+ ratio filter) and the epipolar
// row-band stereo matcher of Snake::Preprocess.
//
// Mapping to the hardware: one 64-lane wavefront owns QW query descriptors (held in registers,
// wave-uniform) and its lanes stride over the train set with coalesced 32-byte-per-lane loads
// (served by L1/L2: a 1000-descriptor train set is 32 KB).  Each lane keeps the two smallest
// packed keys (distance << 20 | index) per query; the wavefront then merges them with a
// 6-step xor-shuffle network.  Packing the index below the distance makes the reduction a plain
// unsigned min, and reproduces the sequential scan's "strict <, first index wins" tie-break
// exactly (the two lexicographically smallest (dist, idx) pairs).
#include "common.hpp"

namespace snk
{
namespace
{
using u32 = unsigned int;
using u64 = unsigned long long;

constexpr u32 BF_IDX_BITS = 20;
constexpr u32 BF_IDX_MASK = (1u << BF_IDX_BITS) - 1u;
constexpr u32 BF_INF_KEY  = ((u32)SNK_DIST_INF << BF_IDX_BITS) | BF_IDX_MASK;

__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// keep the two smallest of {k1 <= k2} U {o1 <= o2}
template <typename K>
__device__ __forceinline__ void merge2(K& k1, K& k2, K o1, K o2)
{
    K lo = k1 < o1 ? k1 : o1;
    K hi = k1 < o1 ? o1 : k1;
    K m  = k2 < o2 ? k2 : o2;
    k2   = hi < m ? hi : m;
    k1   = lo;
}

template <typename K>
__device__ __forceinline__ void insert2(K& k1, K& k2, K key)
{
    K hi = k1 < key ? key : k1;
    k1   = k1 < key ? k1 : key;
    k2   = k2 < hi ? k2 : hi;
}

template <int QW>
__global__ __launch_bounds__(256) void bf_knn2_kernel(const uint4* __restrict__ query, const int* __restrict__ nq_dev,
                                                      int nq_cap, int nq_host, const uint4* __restrict__ train,
                                                      const int* __restrict__ nt_dev, int nt_cap, int nt_host,
                                                      snk_knn2* __restrict__ out)
{
    const int b    = blockIdx.y;
    int nq         = nq_dev ? nq_dev[b] : nq_host;
    int nt         = nt_dev ? nt_dev[b] : nt_host;
    nq             = nq < nq_cap ? nq : nq_cap;
    nt             = nt < nt_cap ? nt : nt_cap;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int q0   = (blockIdx.x * 4 + wave) * QW;
    if (q0 >= nq) return;

    const uint4* qb = query + (size_t)b * nq_cap * 2;
    const uint4* tb = train + (size_t)b * nt_cap * 2;

    uint4 qa[QW], qc[QW];
    u32 k1[QW], k2[QW];
#pragma unroll
    for (int k = 0; k < QW; ++k)
    {
        int qi = q0 + k < nq ? q0 + k : nq - 1;
        qa[k]  = qb[(size_t)qi * 2];
        qc[k]  = qb[(size_t)qi * 2 + 1];
        k1[k]  = BF_INF_KEY;
        k2[k]  = BF_INF_KEY;
    }

    for (int j = lane; j < nt; j += 64)
    {
        const uint4 ta = tb[(size_t)j * 2];
        const uint4 tc = tb[(size_t)j * 2 + 1];
#pragma unroll
        for (int k = 0; k < QW; ++k)
        {
            u32 d   = (u32)hamming256(qa[k], qc[k], ta, tc);
            // d == 256 is "infinite" on the Snake side (strict '<' against an initial 256), never a neighbour
            u32 key = d < (u32)SNK_DIST_INF ? ((d << BF_IDX_BITS) | (u32)j) : BF_INF_KEY;
            insert2(k1[k], k2[k], key);
        }
    }

#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
#pragma unroll
        for (int k = 0; k < QW; ++k)
        {
            u32 o1 = __shfl_xor(k1[k], off);
            u32 o2 = __shfl_xor(k2[k], off);
            merge2(k1[k], k2[k], o1, o2);
        }
    }

    if (lane == 0)
    {
#pragma unroll
        for (int k = 0; k < QW; ++k)
        {
            if (q0 + k < nq)
            {
                snk_knn2 r;
                r.dist1 = (int)(k1[k] >> BF_IDX_BITS);
                r.idx1  = (k1[k] & BF_IDX_MASK) == BF_IDX_MASK ? -1 : (int)(k1[k] & BF_IDX_MASK);
                r.dist2 = (int)(k2[k] >> BF_IDX_BITS);
                r.idx2  = (k2[k] & BF_IDX_MASK) == BF_IDX_MASK ? -1 : (int)(k2[k] & BF_IDX_MASK);
                out[(size_t)b * nq_cap + q0 + k] = r;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// kNN-2 on the matrix cores.  The 256-bit Hamming distance is h = |q| + |t| - 2 (q . t) with q . t the dot product of
// the descriptors' bits -- an exact int8 GEMM.  The all-pairs scan is bound by the vector popcount rate (22 VALU
// instructions per pair in bf_knn2_kernel); v_mfma_i32_32x32x32_i8 computes 32 x 32 dot products over 32 bit-bytes in
// one instruction, leaving the vector ALU the top-2 bookkeeping (4 instructions per pair).
//   workgroup = 4 wavefronts = 4 blocks of 32 queries; train descriptors go by in tiles of 32.
//   bit -> byte expansion: byte c of dword ((w >> kc) & 0x01010101) is bit kc + 8 c of descriptor word w; lane group
//   g = lane >> 5 takes the g-th 128-bit half, MFMA step kc the shift kc.  Queries (B operand) are expanded once into
//   32 registers; every train tile is expanded once per workgroup into LDS (8 KiB, double buffered) and read as the
//   A operand by the four wavefronts.  C[row = train][col = query]: lane holds 16 trains of query lane & 31
//   (row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).
//   keys: |t| << 20 | train index (per tile in LDS, sentinel for rows past nt) minus dot << 21 orders a query's
//   candidates like distance << 20 | index (the query's own popcount is a constant, added at the end).
// ------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int BFM_SENTINEL = 0x7F000000;  // "no train": larger than any key, survives the subtraction of dot << 21 (dot = 0)

__device__ __forceinline__ v4i expand_bits(const uint4& h, int kc)
{
    const u32 M = 0x01010101u;
    return v4i{(int)((h.x >> kc) & M), (int)((h.y >> kc) & M), (int)((h.z >> kc) & M), (int)((h.w >> kc) & M)};
}
__device__ __forceinline__ int popc128(const uint4& h) { return __popc(h.x) + __popc(h.y) + __popc(h.z) + __popc(h.w); }
__device__ __forceinline__ void insert2s(int& k1, int& k2, int key)
{
    // k1 <= k2: the new second smallest is the median of (k1, k2, key) -- v_med3_i32 + v_min_i32, two instructions per
    // pair instead of three in the kernel's epilogue (which, not the matrix pipe, is what limits it)
    int m;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(m) : "v"(k1), "v"(k2), "v"(key));
    k2 = m;
    k1 = min(k1, key);
}

__global__ __launch_bounds__(256) void bf_knn2_mfma_kernel(const uint4* __restrict__ query, const int* __restrict__ nq_dev,
                                                           int nq_cap, int nq_host, const uint4* __restrict__ train,
                                                           const int* __restrict__ nt_dev, int nt_cap, int nt_host,
                                                           snk_knn2* __restrict__ out)
{
    __shared__ v4i As[2][8 * 2 * 32];  // [buffer][(kc * 2 + g) * 32 + train row]
    __shared__ __attribute__((aligned(16))) int ptk[2][32];
    const int b = blockIdx.y;
    int nq      = nq_dev ? nq_dev[b] : nq_host;
    int nt      = nt_dev ? nt_dev[b] : nt_host;
    nq          = nq < nq_cap ? nq : nq_cap;
    nt          = nt < nt_cap ? nt : nt_cap;
    if ((int)blockIdx.x * 128 >= nq) return;  // whole workgroup
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = lane & 31, g = lane >> 5;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const uint4* qb = query + (size_t)b * nq_cap * 2;
    const uint4* tb = train + (size_t)b * nt_cap * 2;

    // B operand: this lane's half of query q0 + j, all 8 shifts
    const int qi   = min(q0 + j, nq - 1);
    const uint4 qh = qb[(size_t)qi * 2 + g], qo = qb[(size_t)qi * 2 + (1 - g)];
    const int pq   = popc128(qh) + popc128(qo);
    v4i B[8];
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) B[kc] = expand_bits(qh, kc);

    // expansion role of this lane: train row e_i of the tile, half e_g, shifts 2 e_k and 2 e_k + 1
    const int e_i = wave * 8 + (lane & 7), e_g = (lane >> 3) & 1, e_k = lane >> 4;
    const int ntiles = (nt + 31) >> 5;
    auto fetch = [&](int t, uint4& h, uint4& o)
    {
        const int ti = t * 32 + e_i;
        h = uint4{0u, 0u, 0u, 0u};
        if (ti < nt) h = tb[(size_t)ti * 2 + e_g];
        (void)o;
    };
    auto expand = [&](int t, int buf, const uint4& h, const uint4&)
    {
        As[buf][((2 * e_k) * 2 + e_g) * 32 + e_i]     = expand_bits(h, 2 * e_k);
        As[buf][((2 * e_k + 1) * 2 + e_g) * 32 + e_i] = expand_bits(h, 2 * e_k + 1);
        // |t| of the row: this lane's half + the other half, which the lane 8 further (e_g = 1, same e_i, e_k) holds
        const int ph = popc128(h);
        const int pt = ph + __shfl_down(ph, 8);
        if (lane < 8)
        {
            const int ti  = t * 32 + e_i;
            ptk[buf][e_i] = ti < nt ? ((pt << BF_IDX_BITS) | ti) : BFM_SENTINEL;
        }
    };

    int k1 = BFM_SENTINEL, k2 = BFM_SENTINEL;
    uint4 h, o;
    if (ntiles > 0)
    {
        fetch(0, h, o);
        expand(0, 0, h, o);
        if (ntiles > 1) fetch(1, h, o);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t)
    {
        const int buf = t & 1;
        if (t + 1 < ntiles)
        {
            expand(t + 1, buf ^ 1, h, o);            // loaded one iteration ago
            if (t + 2 < ntiles) fetch(t + 2, h, o);  // lands during this iteration
        }
        v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        v4i a[8], pk[4];
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) a[kc] = As[buf][(kc * 2 + g) * 32 + j];  // all LDS reads in flight before the MFMA chain
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) pk[rq] = *reinterpret_cast<const v4i*>(&ptk[buf][8 * rq + 4 * g]);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kc], B[kc], acc, 0, 0, 0);
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)  // key' = (|t| << 20 | idx) - (dot << 21): one v_mad_i32_i24
                insert2s(k1, k2, __mul24(acc[4 * rq + rr], -(1 << (BF_IDX_BITS + 1))) + pk[rq][rr]);
        __syncthreads();
    }
    // the two halves of the wavefront hold disjoint trains of the same query
    {
        const int o1 = __shfl_xor(k1, 32), o2 = __shfl_xor(k2, 32);
        const int lo = min(k1, o1), hi = max(k1, o1), m = min(k2, o2);
        k1 = lo;
        k2 = min(hi, m);
    }
    if (lane < 32 && q0 + lane < nq)
    {
        snk_knn2 r;
        const int a1 = k1 >= BFM_SENTINEL - (1 << 30) ? -1 : k1 + (pq << BF_IDX_BITS);
        const int a2 = k2 >= BFM_SENTINEL - (1 << 30) ? -1 : k2 + (pq << BF_IDX_BITS);
        const bool v1 = a1 >= 0 && (a1 >> BF_IDX_BITS) < SNK_DIST_INF, v2 = a2 >= 0 && (a2 >> BF_IDX_BITS) < SNK_DIST_INF;
        r.dist1 = v1 ? a1 >> BF_IDX_BITS : SNK_DIST_INF;
        r.idx1  = v1 ? (int)((u32)a1 & BF_IDX_MASK) : -1;
        r.dist2 = v2 ? a2 >> BF_IDX_BITS : SNK_DIST_INF;
        r.idx2  = v2 ? (int)((u32)a2 & BF_IDX_MASK) : -1;
        out[(size_t)b * nq_cap + q0 + lane] = r;
    }
}

// Order-preserving compaction of the accepted (query, train) pairs; one workgroup per batch entry.
__global__ __launch_bounds__(256) void bf_filter_kernel(const snk_knn2* __restrict__ knn, const int* __restrict__ nq_dev,
                                                        int nq_cap, int nq_host, int threshold, float ratio, int th_strict,
                                                        int ratio_strict, int2* __restrict__ pairs, int* __restrict__ n_pairs)
{
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    const int b    = blockIdx.x;
    int nq         = nq_dev ? nq_dev[b] : nq_host;
    nq             = nq < nq_cap ? nq : nq_cap;
    const int tid  = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int start = 0; start < nq; start += 256)
    {
        const int i = start + tid;
        bool keep   = false;
        snk_knn2 k  = {};
        if (i < nq)
        {
            k    = knn[(size_t)b * nq_cap + i];
            // [DEFINED] operator strictness of filterMatches: definitions "bf_filter.threshold_strict" / "bf_filter.ratio_strict"
            const float rd2 = ratio * (float)k.dist2;
            keep = k.idx1 >= 0 && (th_strict ? k.dist1 < threshold : k.dist1 <= threshold) &&
                   (ratio_strict ? (float)k.dist1 < rd2 : (float)k.dist1 <= rd2);
        }
        const u64 mask   = __ballot(keep);
        const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(mask);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (keep) pairs[(size_t)b * nq_cap + off + prefix] = make_int2(i, k.idx1);
        __syncthreads();
        if (tid == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (tid == 0) n_pairs[b] = base_s;
}

struct LevelScales
{
    float s[32];
    int n;
    int iround_mode;  // definition "iround.mode" at the time of the call (make_scales)
};

// Saiga::iRound [DEFINED]: 0 = floor(x + 0.5), 1 = half away from zero, 2 = half to even (exact operations on both sides)
__device__ __forceinline__ int iround_d(double x, int mode)
{
    if (mode == 0) return (int)floor(x + 0.5);
    if (mode == 1) return (int)(x < 0.0 ? -floor(0.5 - x) : floor(x + 0.5));
    return (int)rint(x);
}

constexpr u64 ST_INF_KEY = (250ull << 40) | 0xFFFFFFFFFFull;
constexpr int ST_ROW_BIAS    = 4096;  // rounded rows are clamped to [-4096, 61439] for the index only
constexpr int ST_SORT_MAX    = 8192;  // right keypoints per image the in-LDS sort handles

// row-sorted index of the right keypoints of every image: (clamped rounded row + bias) << 16 | index
__device__ __forceinline__ void stereo_sort_network(const snk_kp64* __restrict__ rb, int nr, int iround_mode, u32* keys, u32* __restrict__ out)
{
    const int tid = threadIdx.x;
    int n_pow2    = 2;
    while (n_pow2 < nr) n_pow2 <<= 1;
    for (int i = tid; i < n_pow2; i += 256)
    {
        u32 k = 0xFFFFFFFFu;
        if (i < nr)
        {
            int row = iround_d(rb[i].y, iround_mode) + ST_ROW_BIAS;
            row     = row < 0 ? 0 : (row > 65535 ? 65535 : row);
            k       = ((u32)row << 16) | (u32)i;
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
    for (int i = tid; i < nr; i += 256) out[i] = keys[i];
}
__global__ __launch_bounds__(256) void stereo_sort_kernel(const snk_kp64* __restrict__ right, const int* __restrict__ nr_dev,
                                                          int nr_cap, int nr_host, int iround_mode, u32* __restrict__ row_sorted)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ssm[];
    const int b = blockIdx.x;
    int nr      = nr_dev ? nr_dev[b] : nr_host;
    nr          = nr < nr_cap ? nr : nr_cap;
    stereo_sort_network(right + (size_t)b * nr_cap, nr, iround_mode, reinterpret_cast<u32*>(ssm), row_sorted + (size_t)b * nr_cap);
}

// The same index by counting: the rows of a frame's right keypoints span the image height, so a histogram over [min row, max row]
// (LDS atomics), its scan and a rank inside each row's members replace the 55 compare-exchange stages of the network (18.6 -> ~5 us for
// the one frame of a per-frame call).  Identical output: keys ascending = (row, then index).  A frame whose rows span more than
// ST_COUNT_ROWS (rectification pushed keypoints far outside the image) runs the network inside the same launch.
constexpr int ST_COUNT_ROWS = 4096;
// n_matches / prefill_rp / prefill_dp (may be NULL): the frame's match counter is reset here and -- for the one-call front-end --
// right_points / depth get Frame::allocateTmp's -1000 (Snake/Map/Frame.cpp:25-26) here, instead of one fill launch each in front of
// a chain of a dozen short launches (a fill is a 4.5 us kernel of its own: profiles/r05/r05e_pipeline_trace_after.txt).
__global__ __launch_bounds__(256) void stereo_count_kernel(const snk_kp64* __restrict__ right, const int* __restrict__ nr_dev, int nr_cap,
                                                           int nr_host, int iround_mode, u32* __restrict__ row_sorted,
                                                           int* __restrict__ n_matches, float* __restrict__ prefill_rp,
                                                           float* __restrict__ prefill_dp, int prefill_n)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ssm[];
    if (n_matches && threadIdx.x == 0) n_matches[blockIdx.x] = 0;
    if (prefill_rp)
        for (int i = threadIdx.x; i < prefill_n; i += 256)
        {
            prefill_rp[(size_t)blockIdx.x * prefill_n + i] = -1000.0f;
            prefill_dp[(size_t)blockIdx.x * prefill_n + i] = -1000.0f;
        }
    __shared__ int s_min, s_max, s_wsum[4];
    int* start   = reinterpret_cast<int*>(ssm);      // [ST_COUNT_ROWS + 1]
    int* fill    = start + ST_COUNT_ROWS + 1;        // [ST_COUNT_ROWS + 1]
    int* rowof   = fill + ST_COUNT_ROWS + 1;         // [nr]
    int* seg     = rowof + nr_cap;                   // [nr]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nr      = nr_dev ? nr_dev[b] : nr_host;
    nr          = nr < nr_cap ? nr : nr_cap;
    const snk_kp64* rb = right + (size_t)b * nr_cap;
    if (tid == 0) s_min = 65535, s_max = 0;
    __syncthreads();
    int lmin = 65535, lmax = 0;
    for (int i = tid; i < nr; i += 256)
    {
        int row = iround_d(rb[i].y, iround_mode) + ST_ROW_BIAS;
        row     = row < 0 ? 0 : (row > 65535 ? 65535 : row);
        rowof[i] = row;
        lmin = min(lmin, row), lmax = max(lmax, row);
    }
    atomicMin(&s_min, lmin);
    atomicMax(&s_max, lmax);
    __syncthreads();
    const int r0 = s_min, span = nr > 0 ? s_max - s_min + 1 : 1;
    if (span > ST_COUNT_ROWS)  // whole workgroup
    {
        __syncthreads();  // everybody has read s_min / s_max; the network reuses the LDS from its start
        stereo_sort_network(rb, nr, iround_mode, reinterpret_cast<u32*>(ssm), row_sorted + (size_t)b * nr_cap);
        return;
    }
    for (int c = tid; c <= span; c += 256) start[c] = 0;
    __syncthreads();
    for (int i = tid; i < nr; i += 256) atomicAdd(&start[rowof[i] - r0], 1);
    __syncthreads();
    {
        const int chunk = (span + 256) / 256;
        const int c0 = tid * chunk, c1 = min(c0 + chunk, span + 1);
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
            base += v;
        }
    }
    __syncthreads();
    for (int i = tid; i < nr; i += 256) seg[atomicAdd(&fill[rowof[i] - r0], 1)] = i;
    __syncthreads();
    for (int i = tid; i < nr; i += 256)
    {
        const int c = rowof[i] - r0, s0 = start[c], s1 = start[c + 1];
        int r = s0;
        for (int q = s0; q < s1; ++q) r += seg[q] < i ? 1 : 0;
        row_sorted[(size_t)b * nr_cap + r] = ((u32)rowof[i] << 16) | (u32)i;
    }
}

// Snake::Preprocess::StereoMatching (reference Snake/Preprocess/Preprocess.cpp:161-240) with one
// wavefront per left keypoint.  The reference walks row buckets y-r..y+r in ascending row, each in
// index order, with strict '<'; that scan picks the lexicographic minimum of (dist, row, idx), so
// the lanes evaluate the gates for every right keypoint and min-reduce that packed key instead of
// building row buckets.
__global__ __launch_bounds__(256) void stereo_kernel(const snk_kp64* __restrict__ left, const uint4* __restrict__ dl,
                                                     const int* __restrict__ nl_dev, int nl_cap, int nl_host,
                                                     const snk_kp64* __restrict__ right, const uint4* __restrict__ dr,
                                                     const int* __restrict__ nr_dev, int nr_cap, int nr_host, double bf,
                                                     LevelScales ls, int relaxed, float* __restrict__ right_points,
                                                     float* __restrict__ depth, int* __restrict__ n_matches,
                                                     const u32* __restrict__ row_sorted)
{
    const int b    = blockIdx.y;
    int nl         = nl_dev ? nl_dev[b] : nl_host;
    int nr         = nr_dev ? nr_dev[b] : nr_host;
    nl             = nl < nl_cap ? nl : nl_cap;
    nr             = nr < nr_cap ? nr : nr_cap;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int i    = blockIdx.x * 4 + wave;
    if (i >= nl || nr <= 0) return;

    const snk_kp64* lb = left + (size_t)b * nl_cap;
    const snk_kp64* rb = right + (size_t)b * nr_cap;
    const uint4* dlb   = dl + (size_t)b * nl_cap * 2;
    const uint4* drb   = dr + (size_t)b * nr_cap * 2;

    const snk_kp64 kp = lb[i];
    const uint4 qa    = dlb[(size_t)i * 2];
    const uint4 qc    = dlb[(size_t)i * 2 + 1];
    const int y       = iround_d(kp.y, ls.iround_mode);
    int oct           = kp.octave;
    oct               = oct < 0 ? 0 : (oct >= ls.n ? ls.n - 1 : oct);
    const float r     = ceilf(2.0f * ls.s[oct]);
    const int ri      = (int)r;
    const float min_disp = 0.0f;
    const float max_disp = (float)(bf * 0.5);

    u64 k1 = ST_INF_KEY, k2 = ST_INF_KEY;
    // With the row-sorted index (biased rounded row << 16 | index, ascending) only the right keypoints
    // of rows y-r .. y+r are visited; every gate is still evaluated on the true values below.
    int scan_lo = 0, scan_hi = nr;
    const u32* srt = row_sorted ? row_sorted + (size_t)b * nr_cap : nullptr;
    if (srt)
    {
        const int lo_row = min(max(y - ri + ST_ROW_BIAS, 0), 65535), hi_row = min(max(y + ri + ST_ROW_BIAS, 0), 65535);
        const u32 lo_key = (u32)lo_row << 16, hi_key = ((u32)hi_row << 16) | 0xFFFFu;
        int a = 0, c = nr;
        while (a < c)
        {
            const int mid = (a + c) >> 1;
            if (srt[mid] < lo_key) a = mid + 1; else c = mid;
        }
        scan_lo = a;
        c       = nr;
        while (a < c)
        {
            const int mid = (a + c) >> 1;
            if (srt[mid] <= hi_key) a = mid + 1; else c = mid;
        }
        scan_hi = a;
    }
    for (int pos = scan_lo + lane; pos < scan_hi; pos += 64)
    {
        const int j       = srt ? (int)(srt[pos] & 0xFFFFu) : pos;
        const snk_kp64 kr = rb[j];
        const int yj      = iround_d(kr.y, ls.iround_mode);
        const int rel     = yj - (y - ri);
        if (rel < 0 || rel > 2 * ri) continue;
        const double disparity = kp.x - kr.x;
        if (disparity < (double)min_disp || disparity > (double)max_disp) continue;
        int doct = kp.octave - kr.octave;
        doct     = doct < 0 ? -doct : doct;
        if (doct > 1) continue;
        const uint4 ta = drb[(size_t)j * 2];
        const uint4 tc = drb[(size_t)j * 2 + 1];
        const int dist = hamming256(qa, qc, ta, tc);
        if (dist >= 250) continue;
        const u64 key = ((u64)dist << 40) | ((u64)rel << 24) | (u64)j;
        insert2(k1, k2, key);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
        u64 o1 = __shfl_xor(k1, off);
        u64 o2 = __shfl_xor(k2, off);
        merge2(k1, k2, o1, o2);
    }
    if (lane != 0) return;

    const int best_dist        = (int)(k1 >> 40);
    const int second_best_dist = (int)(k2 >> 40);
    if (best_dist > (relaxed ? 75 : 40)) return;
    if ((double)best_dist > (relaxed ? 0.9 : 0.7) * (double)second_best_dist) return;
    const int best_id  = (int)(k1 & 0xFFFFFFull);
    const snk_kp64 kb  = rb[best_id];
    const float angle1 = kp.angle;
    const float angle2 = kb.angle;
    const float rot    = fminf(fabsf(angle1 - angle2), fminf(fabsf((angle1 + 365.0f) - angle2), fabsf(angle1 - (angle2 + 365.0f))));
    if (rot > (relaxed ? 25.0f : 5.0f)) return;

    double right_point = kb.x;
    double disparity   = kp.x - right_point;
    if (disparity <= 0.001)
    {
        disparity   = 0.001;
        right_point = kp.x - disparity;
    }
    right_points[(size_t)b * nl_cap + i] = (float)right_point;
    depth[(size_t)b * nl_cap + i]        = (float)(bf / disparity);
    atomicAdd(&n_matches[b], 1);
}

// Same matcher with SIXTEEN lanes per left keypoint (16 keypoints per workgroup) for the indexed case:
// a row band holds ~15-30 candidates, so a whole wavefront per keypoint left three quarters of the lanes
// idle, and the two binary searches over the row index were ~20 dependent global loads per keypoint.  The
// workgroup stages the image's row index in LDS once, every 16-lane group searches it there, strides over
// its band and merges the two smallest keys with 4 xor steps inside the group.
__global__ __launch_bounds__(256) void stereo_kernel16(const snk_kp64* __restrict__ left, const uint4* __restrict__ dl,
                                                     const int* __restrict__ nl_dev, int nl_cap, int nl_host,
                                                     const snk_kp64* __restrict__ right, const uint4* __restrict__ dr,
                                                     const int* __restrict__ nr_dev, int nr_cap, int nr_host, double bf,
                                                     LevelScales ls, int relaxed, float* __restrict__ right_points,
                                                     float* __restrict__ depth, int* __restrict__ n_matches,
                                                     const u32* __restrict__ row_sorted)
{
    const int b    = blockIdx.y;
    int nl         = nl_dev ? nl_dev[b] : nl_host;
    int nr         = nr_dev ? nr_dev[b] : nr_host;
    nl             = nl < nl_cap ? nl : nl_cap;
    nr             = nr < nr_cap ? nr : nr_cap;
    extern __shared__ u32 s_srt[];
    const int lane = threadIdx.x & 15;
    const int i    = blockIdx.x * 16 + (threadIdx.x >> 4);
    {
        const u32* g = row_sorted + (size_t)b * nr_cap;
        for (int t = threadIdx.x; t < nr; t += 256) s_srt[t] = g[t];
    }
    __syncthreads();
    if (i >= nl || nr <= 0) return;

    const snk_kp64* lb = left + (size_t)b * nl_cap;
    const snk_kp64* rb = right + (size_t)b * nr_cap;
    const uint4* dlb   = dl + (size_t)b * nl_cap * 2;
    const uint4* drb   = dr + (size_t)b * nr_cap * 2;

    const snk_kp64 kp = lb[i];
    const uint4 qa    = dlb[(size_t)i * 2];
    const uint4 qc    = dlb[(size_t)i * 2 + 1];
    const int y       = iround_d(kp.y, ls.iround_mode);
    int oct           = kp.octave;
    oct               = oct < 0 ? 0 : (oct >= ls.n ? ls.n - 1 : oct);
    const float r     = ceilf(2.0f * ls.s[oct]);
    const int ri      = (int)r;
    const float min_disp = 0.0f;
    const float max_disp = (float)(bf * 0.5);

    u64 k1 = ST_INF_KEY, k2 = ST_INF_KEY;
    // With the row-sorted index (biased rounded row << 16 | index, ascending) only the right keypoints
    // of rows y-r .. y+r are visited; every gate is still evaluated on the true values below.
    int scan_lo = 0, scan_hi = nr;
    const u32* srt = s_srt;
    {
        const int lo_row = min(max(y - ri + ST_ROW_BIAS, 0), 65535), hi_row = min(max(y + ri + ST_ROW_BIAS, 0), 65535);
        const u32 lo_key = (u32)lo_row << 16, hi_key = ((u32)hi_row << 16) | 0xFFFFu;
        int a = 0, c = nr;
        while (a < c)
        {
            const int mid = (a + c) >> 1;
            if (srt[mid] < lo_key) a = mid + 1; else c = mid;
        }
        scan_lo = a;
        c       = nr;
        while (a < c)
        {
            const int mid = (a + c) >> 1;
            if (srt[mid] <= hi_key) a = mid + 1; else c = mid;
        }
        scan_hi = a;
    }
    for (int pos = scan_lo + lane; pos < scan_hi; pos += 16)
    {
        const int j       = (int)(srt[pos] & 0xFFFFu);
        const snk_kp64 kr = rb[j];
        const int yj      = iround_d(kr.y, ls.iround_mode);
        const int rel     = yj - (y - ri);
        if (rel < 0 || rel > 2 * ri) continue;
        const double disparity = kp.x - kr.x;
        if (disparity < (double)min_disp || disparity > (double)max_disp) continue;
        int doct = kp.octave - kr.octave;
        doct     = doct < 0 ? -doct : doct;
        if (doct > 1) continue;
        const uint4 ta = drb[(size_t)j * 2];
        const uint4 tc = drb[(size_t)j * 2 + 1];
        const int dist = hamming256(qa, qc, ta, tc);
        if (dist >= 250) continue;
        const u64 key = ((u64)dist << 40) | ((u64)rel << 24) | (u64)j;
        insert2(k1, k2, key);
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1)
    {
        u64 o1 = __shfl_xor(k1, off, 16);
        u64 o2 = __shfl_xor(k2, off, 16);
        merge2(k1, k2, o1, o2);
    }
    if (lane != 0) return;

    const int best_dist        = (int)(k1 >> 40);
    const int second_best_dist = (int)(k2 >> 40);
    if (best_dist > (relaxed ? 75 : 40)) return;
    if ((double)best_dist > (relaxed ? 0.9 : 0.7) * (double)second_best_dist) return;
    const int best_id  = (int)(k1 & 0xFFFFFFull);
    const snk_kp64 kb  = rb[best_id];
    const float angle1 = kp.angle;
    const float angle2 = kb.angle;
    const float rot    = fminf(fabsf(angle1 - angle2), fminf(fabsf((angle1 + 365.0f) - angle2), fabsf(angle1 - (angle2 + 365.0f))));
    if (rot > (relaxed ? 25.0f : 5.0f)) return;

    double right_point = kb.x;
    double disparity   = kp.x - right_point;
    if (disparity <= 0.001)
    {
        disparity   = 0.001;
        right_point = kp.x - disparity;
    }
    right_points[(size_t)b * nl_cap + i] = (float)right_point;
    depth[(size_t)b * nl_cap + i]        = (float)(bf / disparity);
    atomicAdd(&n_matches[b], 1);
}

// The same matcher for a BATCH of frames: ONE workgroup of 1024 threads per frame keeps the frame's whole right side
// in LDS -- x, rounded row, octave, angle, descriptor and a row-bucket index (56 bytes per keypoint) -- so that the
// candidate scan of a 16-lane group (band lookup, gates, Hamming) never leaves the CU.  stereo_kernel16 pays two
// dependent global gathers per candidate step plus the index copy per 16 left keypoints and is bound by their
// latency; here only the left keypoint / descriptor of the next round is a global load, issued one round ahead.
// The index is a counting sort by row (LDS histogram, one block scan, scatter): bucket = row - first row, clamped to
// ST_ROWS - 1, so a band is start[bucket(lo)] .. start[bucket(hi) + 1] without a search; the order inside a bucket is
// whatever the atomics gave -- the result is the minimum of (distance, row, index) keys and every gate is evaluated
// on the true values, so neither that order nor the clamping can change it.
// Replaces stereo_sort_kernel + stereo_kernel16 when the right side fits (nr_cap <= ST_FRAME_MAX).
constexpr int ST_FRAME_MAX = 2560;
constexpr int ST_ROWS      = 2048;
__host__ __device__ inline size_t stereo_frame_lds(int nr_cap) { return (size_t)nr_cap * 56 + (size_t)(ST_ROWS + 2) * 4 + 16; }

constexpr int ST_GROUP = 4;  // 16: 41.3 us, 4: 28.9, 2: 29.3, 1: 30.3 per 256 frames of 1000 + 1000 keypoints
__global__ __launch_bounds__(1024) void stereo_frame_kernel(const snk_kp64* __restrict__ left, const uint4* __restrict__ dl,
                                                            const int* __restrict__ nl_dev, int nl_cap, int nl_host,
                                                            const snk_kp64* __restrict__ right, const uint4* __restrict__ dr,
                                                            const int* __restrict__ nr_dev, int nr_cap, int nr_host,
                                                            double bf, LevelScales ls, int relaxed, float* __restrict__ right_points,
                                                            float* __restrict__ depth, int* __restrict__ n_matches)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    __shared__ int s_cnt, s_row0, s_wtot[16];
    const int b   = blockIdx.x;
    const int tid = threadIdx.x;
    int nl        = nl_dev ? nl_dev[b] : nl_host;
    int nr        = nr_dev ? nr_dev[b] : nr_host;
    nl            = nl < nl_cap ? nl : nl_cap;
    nr            = nr < nr_cap ? nr : nr_cap;
    if (nl <= 0 || nr <= 0) return;  // whole workgroup; n_matches[b] was cleared by the caller
    uint4* sdesc = reinterpret_cast<uint4*>(fsm);                          // [2 * nr_cap]
    double* sx   = reinterpret_cast<double*>(fsm + (size_t)nr_cap * 32);  // [nr_cap]
    int* syj     = reinterpret_cast<int*>(fsm + (size_t)nr_cap * 40);
    int* soct    = syj + nr_cap;
    float* sang  = reinterpret_cast<float*>(soct + nr_cap);
    int* srt     = reinterpret_cast<int*>(sang + nr_cap);  // [nr_cap] right indices grouped by row bucket
    int* start   = srt + nr_cap;                            // [ST_ROWS + 1]
    const snk_kp64* lb = left + (size_t)b * nl_cap;
    const snk_kp64* rb = right + (size_t)b * nr_cap;
    const uint4* dlb   = dl + (size_t)b * nl_cap * 2;
    const uint4* drb   = dr + (size_t)b * nr_cap * 2;

    if (tid == 0)
    {
        s_cnt  = 0;
        s_row0 = 65535;
    }
    for (int t = tid; t <= ST_ROWS; t += 1024) start[t] = 0;
    __syncthreads();
    for (int t = tid; t < 2 * nr; t += 1024) sdesc[t] = drb[t];
    constexpr int PER = (ST_FRAME_MAX + 1023) / 1024;
    int row[PER], rank[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u)
    {
        const int t = tid + 1024 * u;
        row[u]      = -1;
        if (t < nr)
        {
            const snk_kp64 kr = rb[t];
            sx[t]   = kr.x;
            syj[t]  = iround_d(kr.y, ls.iround_mode);
            soct[t] = kr.octave;
            sang[t] = kr.angle;
            int r   = syj[t] + ST_ROW_BIAS;  // the row of stereo_sort_kernel's index: the same rounding rule as syj and the left row
            row[u]  = r < 0 ? 0 : (r > 65535 ? 65535 : r);
            atomicMin(&s_row0, row[u]);
        }
    }
    __syncthreads();
    const int row0 = s_row0;
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if (row[u] >= 0)
        {
            row[u]  = min(row[u] - row0, ST_ROWS - 1);
            rank[u] = atomicAdd(&start[row[u]], 1);
        }
    __syncthreads();
    {
        // exclusive scan of the ST_ROWS bucket counts, two per thread
        const int a = start[2 * tid], c = start[2 * tid + 1];
        int x       = a + c;
        x = wave_scan_incl_dpp(x);
        if ((tid & 63) == 63) s_wtot[tid >> 6] = x;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += s_wtot[w];
        const int excl     = base + x - (a + c);
        start[2 * tid]     = excl;
        start[2 * tid + 1] = excl + a;
        if (tid == 1023) start[ST_ROWS] = base + x;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if (row[u] >= 0) srt[start[row[u]] + rank[u]] = tid + 1024 * u;
    __syncthreads();

    // ST_GROUP lanes share the band scan of one left keypoint (a band holds 10 - 20 candidates)
    constexpr int NGRP = 1024 / ST_GROUP;
    const int lane = tid % ST_GROUP, grp = tid / ST_GROUP;
    const float max_disp = (float)(bf * 0.5);
    snk_kp64 kp_n = lb[min(grp, nl - 1)];
    uint4 qa_n = dlb[(size_t)min(grp, nl - 1) * 2], qc_n = dlb[(size_t)min(grp, nl - 1) * 2 + 1];
    for (int i = grp; i < nl; i += NGRP)
    {
        const snk_kp64 kp = kp_n;
        const uint4 qa = qa_n, qc = qc_n;
        {
            const int in = min(i + NGRP, nl - 1);  // next round's keypoint is in flight during this one
            kp_n = lb[in];
            qa_n = dlb[(size_t)in * 2];
            qc_n = dlb[(size_t)in * 2 + 1];
        }
        const int y = iround_d(kp.y, ls.iround_mode);
        int oct     = kp.octave;
        oct         = oct < 0 ? 0 : (oct >= ls.n ? ls.n - 1 : oct);
        const int ri = (int)ceilf(2.0f * ls.s[oct]);
        const int lo_row  = min(max(y - ri + ST_ROW_BIAS, 0), 65535), hi_row = min(max(y + ri + ST_ROW_BIAS, 0), 65535);
        const int scan_lo = start[min(max(lo_row - row0, 0), ST_ROWS - 1)], scan_hi = start[min(max(hi_row - row0, 0), ST_ROWS - 1) + 1];
        u64 k1 = ST_INF_KEY, k2 = ST_INF_KEY;
        for (int pos = scan_lo + lane; pos < scan_hi; pos += ST_GROUP)
        {
            const int j   = srt[pos];
            const int rel = syj[j] - (y - ri);
            if (rel < 0 || rel > 2 * ri) continue;
            const double disparity = kp.x - sx[j];
            if (disparity < 0.0 || disparity > (double)max_disp) continue;
            int doct = kp.octave - soct[j];
            doct     = doct < 0 ? -doct : doct;
            if (doct > 1) continue;
            const int dist = hamming256(qa, qc, sdesc[2 * j], sdesc[2 * j + 1]);
            if (dist >= 250) continue;
            insert2(k1, k2, ((u64)dist << 40) | ((u64)rel << 24) | (u64)j);
        }
#pragma unroll
        for (int off = ST_GROUP / 2; off >= 1; off >>= 1)
        {
            const u64 o1 = __shfl_xor(k1, off, 16), o2 = __shfl_xor(k2, off, 16);
            merge2(k1, k2, o1, o2);
        }
        if (lane != 0) continue;
        const int best_dist = (int)(k1 >> 40), second_best_dist = (int)(k2 >> 40);
        if (best_dist > (relaxed ? 75 : 40)) continue;
        if ((double)best_dist > (relaxed ? 0.9 : 0.7) * (double)second_best_dist) continue;
        const int best_id  = (int)(k1 & 0xFFFFFFull);
        const float angle1 = kp.angle, angle2 = sang[best_id];
        const float rot    = fminf(fabsf(angle1 - angle2), fminf(fabsf((angle1 + 365.0f) - angle2), fabsf(angle1 - (angle2 + 365.0f))));
        if (rot > (relaxed ? 25.0f : 5.0f)) continue;
        double right_point = sx[best_id];
        double disparity   = kp.x - right_point;
        if (disparity <= 0.001)
        {
            disparity   = 0.001;
            right_point = kp.x - disparity;
        }
        right_points[(size_t)b * nl_cap + i] = (float)right_point;
        depth[(size_t)b * nl_cap + i]        = (float)(bf / disparity);
        atomicAdd(&s_cnt, 1);
    }
    __syncthreads();
    if (tid == 0) n_matches[b] = s_cnt;
}
}  // namespace
}  // namespace snk

using namespace snk;

#include "matcher_handle.hpp"

extern "C" {

int snk_matcher_destroy(snk_matcher* m);

int snk_matcher_create(int device, void* stream, snk_matcher** out)
{
    SNK_REQUIRE(out != nullptr, "out is NULL");
    *out           = nullptr;
    snk_matcher* m = new snk_matcher();
    int rc         = m->init(device, stream);
    // scratch for the per-frame (host pointer) calls is sized here -- 4 MB per buffer holds any call up to ~40 000
    // features / map points -- so that the seams allocate nothing while other seams' threads are running
    // (hipMalloc / hipFree synchronise the whole device); larger calls still grow the buffers once
    constexpr size_t SCRATCH = 4u << 20;
    for (snk::DevBuf* b : {&m->q, &m->t, &m->out, &m->aux, &m->aux2, &m->view})
        if (rc == SNK_OK) rc = b->reserve(SCRATCH);
    if (rc == SNK_OK) rc = m->cnt.reserve(256);
    if (rc == SNK_OK) rc = m->h_in.reserve(1u << 20);  // 10 000 local-map points of the fine matcher
    if (rc == SNK_OK) rc = m->h_res.reserve(256u << 10);
    if (rc != SNK_OK)
    {
        snk_matcher_destroy(m);
        return rc;
    }
    *out = m;
    return SNK_OK;
}

int snk_matcher_destroy(snk_matcher* m)
{
    if (!m) return SNK_OK;
    (void)hipSetDevice(m->device);
    m->q.release();
    m->t.release();
    m->out.release();
    m->aux.release();
    m->aux2.release();
    m->cnt.release();
    m->view.release();
    m->h_in.release();
    m->h_res.release();
    m->fini();
    delete m;
    return SNK_OK;
}

int snk_matcher_sync(snk_matcher* m)
{
    SNK_REQUIRE(m != nullptr, "matcher is NULL");
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    return SNK_OK;
}

static int launch_knn2(snk_matcher* m, const uint64_t* q, const int32_t* nq_dev, int nq_cap, int nq_host,
                       const uint64_t* t, const int32_t* nt_dev, int nt_cap, int nt_host, int batch, snk_knn2* out)
{
    if (batch <= 0 || nq_cap <= 0) return SNK_OK;
    // Matrix-core path once a query block (32) and a train tile (32) are mostly full; the vector kernel below for
    // small sets (SNK_BF_NO_MFMA=1 forces it, for A/B measurements).
    static const bool no_mfma = getenv("SNK_BF_NO_MFMA") != nullptr;
    if (!no_mfma && nq_cap >= 24 && nt_cap >= 24)
    {
        hipLaunchKernelGGL(bf_knn2_mfma_kernel, dim3(ceil_div(nq_cap, 128), batch), dim3(256), 0, m->stream, (const uint4*)q, nq_dev,
                           nq_cap, nq_host, (const uint4*)t, nt_dev, nt_cap, nt_host, out);
        SNK_LAUNCH_CHECK();
        return SNK_OK;
    }
    // 4 queries per wavefront once there is enough work to fill 256 CUs, else 1 (latency).
    const bool wide = (long long)batch * nq_cap >= 16384;
    const int qw    = wide ? 4 : 1;
    dim3 grid(ceil_div(nq_cap, 4 * qw), batch);
    if (wide)
        hipLaunchKernelGGL(bf_knn2_kernel<4>, grid, dim3(256), 0, m->stream, (const uint4*)q, nq_dev, nq_cap, nq_host,
                           (const uint4*)t, nt_dev, nt_cap, nt_host, out);
    else
        hipLaunchKernelGGL(bf_knn2_kernel<1>, grid, dim3(256), 0, m->stream, (const uint4*)q, nq_dev, nq_cap, nq_host,
                           (const uint4*)t, nt_dev, nt_cap, nt_host, out);
    SNK_LAUNCH_CHECK();
    return SNK_OK;
}

int snk_bf_knn2(snk_matcher* m, const uint64_t (*query)[4], int nq, const uint64_t (*train)[4], int nt, snk_knn2* out)
{
    SNK_REQUIRE(m != nullptr, "matcher is NULL");
    SNK_REQUIRE(nq >= 0 && nt >= 0, "negative count");
    SNK_REQUIRE(nt < (int)BF_IDX_MASK, "train set too large (max 2^20-2)");
    if (nq == 0) return SNK_OK;
    SNK_REQUIRE(query != nullptr && out != nullptr, "NULL buffer");
    SNK_REQUIRE(nt == 0 || train != nullptr, "NULL train buffer");
    SNK_HIP_CHECK(hipSetDevice(m->device));
    int rc;
    // query | train in ONE device block, through the pinned staging buffer: one upload, one download (see matcher_handle.hpp)
    const size_t qb = (size_t)nq * 32, tb = (size_t)(nt > 0 ? nt : 1) * 32, ob = (size_t)nq * sizeof(snk_knn2);
    if ((rc = m->q.reserve(qb + tb)) != SNK_OK) return rc;
    if ((rc = m->out.reserve(ob)) != SNK_OK) return rc;
    if ((rc = m->h_in.reserve(qb + tb)) != SNK_OK) return rc;
    if ((rc = m->h_res.reserve(ob)) != SNK_OK) return rc;
    memcpy(m->h_in.p, query, qb);
    if (nt > 0) memcpy(m->h_in.as<char>() + qb, train, (size_t)nt * 32);
    SNK_HIP_CHECK(hipMemcpyAsync(m->q.p, m->h_in.p, qb + (nt > 0 ? (size_t)nt * 32 : 0), hipMemcpyHostToDevice, m->stream));
    rc = launch_knn2(m, m->q.as<uint64_t>(), nullptr, nq, nq, reinterpret_cast<const uint64_t*>(m->q.as<char>() + qb), nullptr,
                     nt > 0 ? nt : 1, nt, 1, m->out.as<snk_knn2>());
    if (rc != SNK_OK) return rc;
    SNK_HIP_CHECK(hipMemcpyAsync(m->h_res.p, m->out.p, ob, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    memcpy(out, m->h_res.p, ob);
    return SNK_OK;
}

int snk_bf_knn2_batch_dev(snk_matcher* m, const uint64_t* query_dev, const int32_t* nq_dev, int nq_cap,
                          const uint64_t* train_dev, const int32_t* nt_dev, int nt_cap, int batch, snk_knn2* out_dev)
{
    SNK_REQUIRE(m != nullptr, "matcher is NULL");
    SNK_REQUIRE(batch >= 0 && nq_cap >= 0 && nt_cap >= 1, "bad sizes");
    SNK_REQUIRE(nt_cap < (int)BF_IDX_MASK, "train capacity too large (max 2^20-2)");
    SNK_REQUIRE(query_dev && train_dev && out_dev && nq_dev && nt_dev, "NULL device buffer");
    SNK_HIP_CHECK(hipSetDevice(m->device));
    return launch_knn2(m, query_dev, nq_dev, nq_cap, 0, train_dev, nt_dev, nt_cap, 0, batch, out_dev);
}

int snk_bf_filter(snk_matcher* m, const snk_knn2* knn, int nq, int threshold, float ratio, int32_t (*pairs)[2],
                  int* n_pairs)
{
    SNK_REQUIRE(m != nullptr, "matcher is NULL");
    SNK_REQUIRE(nq >= 0, "negative count");
    SNK_REQUIRE(n_pairs != nullptr, "n_pairs is NULL");
    *n_pairs = 0;
    if (nq == 0) return SNK_OK;
    SNK_REQUIRE(knn != nullptr && pairs != nullptr, "NULL buffer");
    SNK_HIP_CHECK(hipSetDevice(m->device));
    int rc;
    // aux: count (16 bytes) | pairs: ONE copy back (the count first and the n pairs after it were two round trips)
    const size_t kb = (size_t)nq * sizeof(snk_knn2), pb = 16 + (size_t)nq * 8;
    if ((rc = m->out.reserve(kb)) != SNK_OK) return rc;
    if ((rc = m->aux.reserve(pb)) != SNK_OK) return rc;
    if ((rc = m->h_in.reserve(kb)) != SNK_OK) return rc;
    if ((rc = m->h_res.reserve(pb)) != SNK_OK) return rc;
    memcpy(m->h_in.p, knn, kb);
    SNK_HIP_CHECK(hipMemcpyAsync(m->out.p, m->h_in.p, kb, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(bf_filter_kernel, dim3(1), dim3(256), 0, m->stream, m->out.as<snk_knn2>(), (const int*)nullptr,
                       nq, nq, threshold, ratio, definition(DEF_BF_FILTER_THRESHOLD_STRICT), definition(DEF_BF_FILTER_RATIO_STRICT),
                       reinterpret_cast<int2*>(m->aux.as<char>() + 16), m->aux.as<int>());
    SNK_LAUNCH_CHECK();
    SNK_HIP_CHECK(hipMemcpyAsync(m->h_res.p, m->aux.p, pb, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    const int n = *m->h_res.as<int>();
    if (n > 0) memcpy(pairs, m->h_res.as<char>() + 16, (size_t)n * 8);
    *n_pairs = n;
    return SNK_OK;
}

int snk_bf_filter_batch_dev(snk_matcher* m, const snk_knn2* knn_dev, const int32_t* nq_dev, int nq_cap, int batch,
                            int threshold, float ratio, int32_t* pairs_dev, int32_t* n_pairs_dev)
{
    SNK_REQUIRE(m != nullptr, "matcher is NULL");
    SNK_REQUIRE(batch >= 0 && nq_cap >= 0, "bad sizes");
    SNK_REQUIRE(knn_dev && nq_dev && pairs_dev && n_pairs_dev, "NULL device buffer");
    if (batch == 0) return SNK_OK;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    hipLaunchKernelGGL(bf_filter_kernel, dim3(batch), dim3(256), 0, m->stream, knn_dev, nq_dev, nq_cap, 0, threshold,
                       ratio, definition(DEF_BF_FILTER_THRESHOLD_STRICT), definition(DEF_BF_FILTER_RATIO_STRICT), (int2*)pairs_dev,
                       n_pairs_dev);
    SNK_LAUNCH_CHECK();
    return SNK_OK;
}

static int make_scales(const float* level_scale, int n_levels, LevelScales* ls)
{
    SNK_REQUIRE(level_scale != nullptr && n_levels >= 1 && n_levels <= 32, "level_scale / n_levels (1..32)");
    for (int i = 0; i < 32; ++i) ls->s[i] = i < n_levels ? level_scale[i] : level_scale[n_levels - 1];
    ls->n           = n_levels;
    ls->iround_mode = definition(DEF_IROUND_MODE);
    return SNK_OK;
}

int snk_stereo_match(snk_matcher* m, const snk_kp64* left, const uint64_t (*desc_left)[4], int nl,
                     const snk_kp64* right, const uint64_t (*desc_right)[4], int nr, double bf, const float* level_scale,
                     int n_levels, int relaxed, float* right_points, float* depth, int* n_matches)
{
    SNK_REQUIRE(m != nullptr, "matcher is NULL");
    SNK_REQUIRE(nl >= 0 && nr >= 0, "negative count");
    SNK_REQUIRE(n_matches != nullptr, "n_matches is NULL");
    SNK_REQUIRE(nr < (1 << 24), "right set too large");
    *n_matches = 0;
    LevelScales ls;
    int rc = make_scales(level_scale, n_levels, &ls);
    if (rc != SNK_OK) return rc;
    if (nl == 0 || nr == 0) return SNK_OK;
    SNK_REQUIRE(left && desc_left && right && desc_right && right_points && depth, "NULL buffer");
    SNK_HIP_CHECK(hipSetDevice(m->device));
    const size_t kl = (size_t)nl * sizeof(snk_kp64), kr = (size_t)nr * sizeof(snk_kp64);
    // ONE device block (aux): left kps | right kps | left descriptors | right descriptors | right_points | depth | count, filled through
    // the pinned staging buffer with one copy; right_points | depth | count come back with one copy (nine copies from / to pageable
    // memory before round 4)
    auto al16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_kr = al16(kl), o_dl = al16(o_kr + kr), o_dr = al16(o_dl + (size_t)nl * 32), o_rp = al16(o_dr + (size_t)nr * 32);
    const size_t o_dp = o_rp + (size_t)nl * 4, o_cnt = o_dp + (size_t)nl * 4, in_b = o_cnt + 4, res_b = in_b - o_rp;
    if ((rc = m->aux.reserve(in_b)) != SNK_OK) return rc;
    if ((rc = m->h_in.reserve(in_b)) != SNK_OK) return rc;
    if ((rc = m->h_res.reserve(res_b)) != SNK_OK) return rc;
    char* ab   = m->aux.as<char>();
    char* hb   = m->h_in.as<char>();
    float* rp  = reinterpret_cast<float*>(ab + o_rp);
    float* dp  = reinterpret_cast<float*>(ab + o_dp);
    int* d_cnt = reinterpret_cast<int*>(ab + o_cnt);
    const uint4* d_dl = reinterpret_cast<const uint4*>(ab + o_dl);
    const uint4* d_dr = reinterpret_cast<const uint4*>(ab + o_dr);
    memcpy(hb, left, kl);
    memcpy(hb + o_kr, right, kr);
    memcpy(hb + o_dl, desc_left, (size_t)nl * 32);
    memcpy(hb + o_dr, desc_right, (size_t)nr * 32);
    memcpy(hb + o_rp, right_points, (size_t)nl * 4);
    memcpy(hb + o_dp, depth, (size_t)nl * 4);
    memset(hb + o_cnt, 0, 4);
    SNK_HIP_CHECK(hipMemcpyAsync(ab, hb, in_b, hipMemcpyHostToDevice, m->stream));
    const u32* srt = nullptr;
    if (nr <= ST_SORT_MAX)
    {
        if ((rc = m->out.reserve((size_t)nr * 4)) != SNK_OK) return rc;
        int np2 = 2;
        while (np2 < nr) np2 <<= 1;
        if ((rc = set_max_lds_once(reinterpret_cast<const void*>(stereo_sort_kernel), ST_SORT_MAX * 4)) != SNK_OK) return rc;
        if ((rc = set_max_lds_once(reinterpret_cast<const void*>(stereo_kernel16), ST_SORT_MAX * 4)) != SNK_OK) return rc;
        static const bool sort_network = getenv("SNK_STEREO_SORT_NETWORK") != nullptr;  // A/B, tests: the bitonic network
        if (!sort_network)
        {
            const int nrc    = nr > 0 ? nr : 1;
            const size_t lds = std::max(((size_t)2 * (ST_COUNT_ROWS + 1) + (size_t)2 * nrc) * 4, (size_t)np2 * 4);
            if ((rc = set_max_lds_once(reinterpret_cast<const void*>(stereo_count_kernel), (2 * (ST_COUNT_ROWS + 1) + 2 * ST_SORT_MAX) * 4)) != SNK_OK)
                return rc;
            hipLaunchKernelGGL(stereo_count_kernel, dim3(1), dim3(256), lds, m->stream, (const snk_kp64*)(ab + o_kr), (const int*)nullptr, nrc, nr,
                               ls.iround_mode, m->out.as<u32>(), (int*)nullptr, (float*)nullptr, (float*)nullptr, 0);
        }
        else
            hipLaunchKernelGGL(stereo_sort_kernel, dim3(1), dim3(256), (size_t)np2 * 4, m->stream, (const snk_kp64*)(ab + o_kr),
                               (const int*)nullptr, nr, nr, ls.iround_mode, m->out.as<u32>());
        srt = m->out.as<u32>();
    }
    if (srt)
        hipLaunchKernelGGL(stereo_kernel16, dim3(ceil_div(nl, 16), 1), dim3(256), (size_t)nr * 4, m->stream, (const snk_kp64*)ab, d_dl,
                           (const int*)nullptr, nl, nl, (const snk_kp64*)(ab + o_kr), d_dr, (const int*)nullptr, nr, nr, bf, ls, relaxed, rp, dp,
                           d_cnt, srt);
    else
        hipLaunchKernelGGL(stereo_kernel, dim3(ceil_div(nl, 4), 1), dim3(256), 0, m->stream, (const snk_kp64*)ab, d_dl, (const int*)nullptr,
                           nl, nl, (const snk_kp64*)(ab + o_kr), d_dr, (const int*)nullptr, nr, nr, bf, ls, relaxed, rp, dp, d_cnt, srt);
    SNK_LAUNCH_CHECK();
    SNK_HIP_CHECK(hipMemcpyAsync(m->h_res.p, ab + o_rp, res_b, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    const char* hr = m->h_res.as<char>();
    memcpy(right_points, hr, (size_t)nl * 4);
    memcpy(depth, hr + (size_t)nl * 4, (size_t)nl * 4);
    memcpy(n_matches, hr + (size_t)nl * 8, 4);
    return SNK_OK;
}

int snk_stereo_match_batch_dev(snk_matcher* m, const snk_kp64* left_dev, const uint64_t* desc_left_dev,
                               const int32_t* nl_dev, int nl_cap, const snk_kp64* right_dev,
                               const uint64_t* desc_right_dev, const int32_t* nr_dev, int nr_cap, int batch, double bf,
                               const float* level_scale_host, int n_levels, int relaxed, float* right_points_dev,
                               float* depth_dev, int32_t* n_matches_dev)
{
    return snk::stereo_match_batch_dev_impl(m, left_dev, desc_left_dev, nl_dev, nl_cap, right_dev, desc_right_dev, nr_dev, nr_cap, batch, bf, level_scale_host,
                                            n_levels, relaxed, right_points_dev, depth_dev, n_matches_dev, false);
}
}  // extern "C"

// prefill: right_points / depth (nl_cap entries per frame) are set to -1000 by the call itself (Frame::allocateTmp) -- the one-call
// front-end's form; the public entry point leaves them to the caller (in / out arrays, include/snake_hip.h)
int snk::stereo_match_batch_dev_impl(snk_matcher* m, const snk_kp64* left_dev, const uint64_t* desc_left_dev, const int32_t* nl_dev, int nl_cap,
                                     const snk_kp64* right_dev, const uint64_t* desc_right_dev, const int32_t* nr_dev, int nr_cap, int batch, double bf,
                                     const float* level_scale_host, int n_levels, int relaxed, float* right_points_dev, float* depth_dev,
                                     int32_t* n_matches_dev, bool prefill)
{
    SNK_REQUIRE(m != nullptr, "matcher is NULL");
    SNK_REQUIRE(batch >= 0 && nl_cap >= 0 && nr_cap >= 1 && nr_cap < (1 << 24), "bad sizes");
    SNK_REQUIRE(left_dev && desc_left_dev && nl_dev && right_dev && desc_right_dev && nr_dev && right_points_dev &&
                    depth_dev && n_matches_dev,
                "NULL device buffer");
    LevelScales ls;
    int rc = make_scales(level_scale_host, n_levels, &ls);
    if (rc != SNK_OK) return rc;
    if (batch == 0 || nl_cap == 0) return SNK_OK;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    const u32* srt = nullptr;
    static const bool no_frame_kernel = getenv("SNK_STEREO_NO_FRAME_KERNEL") != nullptr;  // A/B measurements
    static const bool sort_network    = getenv("SNK_STEREO_SORT_NETWORK") != nullptr;     // A/B, tests: the bitonic network
    const bool frame_kernel = !no_frame_kernel && nr_cap <= ST_FRAME_MAX && batch >= 8;
    // the counting kernel of the small-batch path resets the counter and does the prefill itself; every other path pays the fills
    const bool count_path = !frame_kernel && nr_cap <= ST_SORT_MAX && !sort_network;
    if (!count_path)
    {
        SNK_HIP_CHECK(hipMemsetAsync(n_matches_dev, 0, (size_t)batch * sizeof(int), m->stream));
        if (prefill)
        {
            SNK_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(right_points_dev), 0xC47A0000u /* -1000.0f */, (size_t)batch * nl_cap, m->stream));
            SNK_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(depth_dev), 0xC47A0000u, (size_t)batch * nl_cap, m->stream));
        }
    }
    if (frame_kernel)
    {
        // enough frames to give every CU its own: one workgroup per frame, the right side resident in LDS
        if ((rc = set_max_lds_once(reinterpret_cast<const void*>(stereo_frame_kernel), (int)stereo_frame_lds(ST_FRAME_MAX))) != SNK_OK) return rc;
        hipLaunchKernelGGL(stereo_frame_kernel, dim3(batch), dim3(1024), stereo_frame_lds(nr_cap), m->stream, left_dev,
                           (const uint4*)desc_left_dev, nl_dev, nl_cap, 0, right_dev, (const uint4*)desc_right_dev, nr_dev, nr_cap,
                           0, bf, ls, relaxed, right_points_dev, depth_dev, n_matches_dev);
        SNK_LAUNCH_CHECK();
        return SNK_OK;
    }
    if (nr_cap <= ST_SORT_MAX)
    {
        if ((rc = m->out.reserve((size_t)batch * nr_cap * 4)) != SNK_OK) return rc;
        int np2 = 2;
        while (np2 < nr_cap) np2 <<= 1;
        if ((rc = set_max_lds_once(reinterpret_cast<const void*>(stereo_sort_kernel), ST_SORT_MAX * 4)) != SNK_OK) return rc;
        if ((rc = set_max_lds_once(reinterpret_cast<const void*>(stereo_kernel16), ST_SORT_MAX * 4)) != SNK_OK) return rc;
        if (!sort_network)
        {
            const size_t lds = std::max(((size_t)2 * (ST_COUNT_ROWS + 1) + (size_t)2 * nr_cap) * 4, (size_t)np2 * 4);
            if ((rc = set_max_lds_once(reinterpret_cast<const void*>(stereo_count_kernel), (2 * (ST_COUNT_ROWS + 1) + 2 * ST_SORT_MAX) * 4)) != SNK_OK)
                return rc;
            hipLaunchKernelGGL(stereo_count_kernel, dim3(batch), dim3(256), lds, m->stream, right_dev, nr_dev, nr_cap, 0, ls.iround_mode,
                               m->out.as<u32>(), n_matches_dev, prefill ? right_points_dev : (float*)nullptr, prefill ? depth_dev : (float*)nullptr, nl_cap);
        }
        else
            hipLaunchKernelGGL(stereo_sort_kernel, dim3(batch), dim3(256), (size_t)np2 * 4, m->stream, right_dev, nr_dev, nr_cap, 0,
                               ls.iround_mode, m->out.as<u32>());
        srt = m->out.as<u32>();
    }
    if (srt)
        hipLaunchKernelGGL(stereo_kernel16, dim3(ceil_div(nl_cap, 16), batch), dim3(256), (size_t)nr_cap * 4, m->stream, left_dev,
                           (const uint4*)desc_left_dev, nl_dev, nl_cap, 0, right_dev, (const uint4*)desc_right_dev, nr_dev,
                           nr_cap, 0, bf, ls, relaxed, right_points_dev, depth_dev, n_matches_dev, srt);
    else
        hipLaunchKernelGGL(stereo_kernel, dim3(ceil_div(nl_cap, 4), batch), dim3(256), 0, m->stream, left_dev,
                           (const uint4*)desc_left_dev, nl_dev, nl_cap, 0, right_dev, (const uint4*)desc_right_dev, nr_dev,
                           nr_cap, 0, bf, ls, relaxed, right_points_dev, depth_dev, n_matches_dev, srt);
    SNK_LAUNCH_CHECK();
    return SNK_OK;
}
