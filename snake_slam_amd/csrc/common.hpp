// Shared host-side plumbing for the C-ABI library: error reporting, handle base, grow-only
// device scratch.  gfx950 only; no CUDA paths.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "snake_hip.h"

namespace snk
{
void set_error(const char* fmt, ...);

#define SNK_HIP_CHECK(expr)                                                                      \
    do                                                                                           \
    {                                                                                            \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
        {                                                                                        \
            ::snk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SNK_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

#define SNK_REQUIRE(cond, msg)                        \
    do                                                \
    {                                                 \
        if (!(cond))                                  \
        {                                             \
            ::snk::set_error("invalid argument: %s", msg); \
            return SNK_ERR_INVALID_ARG;               \
        }                                             \
    } while (0)

#define SNK_LAUNCH_CHECK()                                                                          \
    do                                                                                              \
    {                                                                                               \
        hipError_t _e = hipGetLastError();                                                          \
        if (_e != hipSuccess)                                                                       \
        {                                                                                           \
            ::snk::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return SNK_ERR_HIP;                                                                     \
        }                                                                                           \
    } while (0)

// Grow-only device buffer.
struct DevBuf
{
    void* p      = nullptr;
    size_t bytes = 0;
    int reserve(size_t n)
    {
        if (n <= bytes) return SNK_OK;
        if (p) (void)hipFree(p);
        p     = nullptr;
        bytes = 0;
        size_t want = n + n / 4 + 256;
        SNK_HIP_CHECK(hipMalloc(&p, want));
        bytes = want;
        // SNK_DEBUG_POISON=1: fill fresh scratch with a pattern so that a kernel relying on
        // zero-initialised memory fails reproducibly instead of depending on the allocator's history
        static const bool poison = getenv("SNK_DEBUG_POISON") != nullptr;
        if (poison)
        {
            // on the calling thread's own stream (never the legacy stream: other seams run beside this one), complete
            // before the handle's stream touches the buffer
            SNK_HIP_CHECK(hipMemsetAsync(p, 0xCD, want, hipStreamPerThread));
            SNK_HIP_CHECK(hipStreamSynchronize(hipStreamPerThread));
        }
        return SNK_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p     = nullptr;
        bytes = 0;
    }
    template <typename T>
    T* as() const
    {
        return reinterpret_cast<T*>(p);
    }
};

// Grow-only pinned host staging buffer: device -> host results land here with ONE asynchronous copy on the handle's
// stream and are handed to the caller's (pageable) arrays with memcpy after the stream synchronisation.
struct HostBuf
{
    void* p      = nullptr;
    size_t bytes = 0;
    int reserve(size_t n)
    {
        if (n <= bytes) return SNK_OK;
        if (p) (void)hipHostFree(p);
        p     = nullptr;
        bytes = 0;
        size_t want = n + n / 4 + 256;
        SNK_HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
        bytes = want;
        return SNK_OK;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p     = nullptr;
        bytes = 0;
    }
    template <typename T>
    T* as() const
    {
        return reinterpret_cast<T*>(p);
    }
};

// Blocking copy on the handle's own stream.  The library never touches the legacy (NULL) stream: the seams are called
// from different OS threads at the same time and a legacy-stream call synchronises with -- and,
// during a stream capture, invalidates -- every other handle's work.
inline int copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t stream)
{
    if (bytes == 0) return SNK_OK;
    SNK_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, stream));
    SNK_HIP_CHECK(hipStreamSynchronize(stream));
    return SNK_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is per-(device, kernel), process-wide state: it is set ONCE per kernel and
// device (the current one -- callers have done hipSetDevice) to the largest carve the kernel supports (never from
// per-handle or per-call sizes, which would race between handles).
int set_max_lds_once(const void* kernel, int bytes);

constexpr int LDS_MAX_BYTES = 160 * 1024;  // gfx950: 160 KB of LDS per workgroup

// Run-time switches for the [DEFINED] choices that are pure comparisons / rounding rules: the arithmetic
// lives in the absent saiga, so a maintainer who can read it flips the definition with snk_set_definition (or the
// SNK_DEFINITIONS environment variable) instead of editing kernels.  Process-wide, read at every call, mirrored by the oracle
// (orc_set_definition).  Values and defaults: include/snake_hip.h.
enum DefKey
{
    DEF_BF_FILTER_THRESHOLD_STRICT = 0,  // filterMatches: 0 = keep d1 <= th (default), 1 = keep d1 < th
    DEF_BF_FILTER_RATIO_STRICT,          // filterMatches: 0 = keep d1 <= ratio * d2 (default), 1 = keep d1 < ratio * d2
    DEF_IROUND_MODE,                     // Saiga::iRound: 0 = floor(x + 0.5) (default), 1 = half away from zero, 2 = half to even
    DEF_ORB_RESPONSE,                    // ORB extractor: 0 = FAST-9 score ranks / is the response (default, ORB-SLAM2), 1 = Harris response (OpenCV's ORB form)
    DEF_COUNT
};
int definition(DefKey k);

struct HandleBase
{
    int device          = 0;
    hipStream_t stream  = nullptr;
    bool own_stream     = false;
    int init(int dev, void* user_stream);
    void fini();
};

inline int ceil_div(int a, int b)
{
    return (a + b - 1) / b;
}

#if defined(__HIPCC__)
// 1 / z and 1 / sqrt(s) from the hardware approximations plus two Newton steps each (5 / 8 instructions; the IEEE division and
// square root the compiler expands to are 14 - 18 each and a match needs three of them per Gauss-Newton step).  Within an ulp or
// two of the correctly rounded values for the magnitudes that occur (depths in metres, squared pixel errors); only for arithmetic
// that is specified by a tolerance ("snk-pose v1", "snk-ba v1": DESIGN.md sections 3c, 4), never where results are compared bit for bit.
__device__ __forceinline__ double rcp_nr(double z)
{
    double y = __builtin_amdgcn_rcp(z);
    y        = fma(y, fma(-z, y, 1.0), y);
    return fma(y, fma(-z, y, 1.0), y);
}
__device__ __forceinline__ double rsqrt_nr(double s)
{
    const double hs = 0.5 * s;
    double y        = __builtin_amdgcn_rsq(s);
    y               = fma(y, fma(-hs * y, y, 0.5), y);
    return fma(y, fma(-hs * y, y, 0.5), y);
}

// x / N for a small positive integer constant N with the bits of the IEEE division it replaces, in five instructions instead of the
// ~14 of the expanded division (the projection matchers evaluate 20 such divisions per local-map point: the 16 terms of det_exp and
// the four cell coordinates of a window).  y = RN(1 / N) is a compile-time constant, r = x - N q is exact in an FMA.
// q1 = RN(q0 + r0 y): q0 + r0 y = x / N - (x / N - q0)(1 - N y) lies within 2^-105 (relative) of x / N, so q1 is a faithful rounding; then
// Markstein's theorem (y the correctly rounded reciprocal, q faithful, N's significand not all ones) makes RN(q1 + r1 y) the
// correctly rounded quotient.  (x = -0 gives +0: every use adds the quotient to 1.0 or takes its floor.)  Checked against the
// division on the hardware over 4e9 random operands by tools/probes/div_const_probe.hip.
template <int N>
__device__ __forceinline__ double div_const(double x)
{
    constexpr double y = 1.0 / (double)N;
    constexpr double n = (double)N;
    double q = x * y;
    double r = __builtin_fma(-n, q, x);
    q        = __builtin_fma(r, y, q);
    r        = __builtin_fma(-n, q, x);
    return __builtin_fma(r, y, q);
}

// Inclusive prefix sum of one int per lane over the wavefront with six DPP additions (shifts by 1, 2, 4, 8 inside each row of 16
// lanes -- lanes without a source add `old` = 0 --, then lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3) instead of
// six __shfl_up = ds_bpermute round trips through the LDS crossbar.  Integer sums: the same values in any order.
__device__ __forceinline__ int wave_scan_incl_dpp(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return x;
}
// Sum over the 64 lanes of a double, returned to every lane, without the LDS: four DPP butterflies inside each row of 16 lanes
// (two 32-bit moves per step), then the four row sums are read back with v_readlane and added in a fixed order.  The
// __shfl_xor tree above is twelve dependent ds_bpermute round trips per sum.
template <int CTRL>
__device__ __forceinline__ double dpp_mov64(double v)
{
#if defined(SNK_DPP_OLD_INIT)  // A/B: the first form -- `old` = 0 without bound_ctrl costs a v_mov per half and step to initialise the destination
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
#else
    // every lane of these controls (quad_perm, row_mirror, row_half_mirror) has a valid source lane: no `old` value is needed
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
#endif
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane64(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// the four DPP steps only: every lane of a row of 16 holds the row's sum
__device__ __forceinline__ double row_sum64_dpp(double v)
{
    v += dpp_mov64<0xB1>(v);
    v += dpp_mov64<0x4E>(v);
    v += dpp_mov64<0x141>(v);
    v += dpp_mov64<0x140>(v);
    return v;
}
__device__ __forceinline__ double wave_sum64_dpp(double v)
{
    v += dpp_mov64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov64<0x141>(v);  // row_half_mirror
    v += dpp_mov64<0x140>(v);  // row_mirror
    return (readlane64(v, 0) + readlane64(v, 16)) + (readlane64(v, 32) + readlane64(v, 48));
}
#endif
}  // namespace snk
