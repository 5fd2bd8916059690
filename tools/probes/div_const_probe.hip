// div_const<N>(x) (snake_slam_amd/csrc/common.hpp) against the IEEE division x / N on the hardware, bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Iinclude -Isnake_slam_amd/csrc tools/probes/div_const_probe.hip -o /tmp/div_const_probe && /tmp/div_const_probe
// Operands: random significands x exponents in [-40, 40] (the matchers divide pixel differences and Taylor terms), plus significands
// next to 1, 2 and to the multiples of N (the hard cases of a quotient's rounding).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "common.hpp"

using namespace snk;

__device__ __forceinline__ uint64_t splitmix(uint64_t& s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int N>
__device__ void check(double x, unsigned long long* bad, unsigned long long* first)
{
    const double a = div_const<N>(x);
    const double b = x / (double)N;
    if (__double_as_longlong(a) != __double_as_longlong(b) && !(x == 0.0))
    {
        if (atomicAdd(bad + N, 1ull) == 0) first[N] = (unsigned long long)__double_as_longlong(x);
    }
}

template <int N>
__device__ void check_upto(double x, unsigned long long* bad, unsigned long long* first)
{
    check<N>(x, bad, first);
    if constexpr (N > 1) check_upto<N - 1>(x, bad, first);
}

__global__ void probe(unsigned long long* bad, unsigned long long* first, int rounds, uint64_t seed)
{
    uint64_t s = seed + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x1234567ull;
    for (int r = 0; r < rounds; ++r)
    {
        const uint64_t u = splitmix(s), v = splitmix(s);
        uint64_t mant    = u & 0xFFFFFFFFFFFFFull;
        const int mode   = (int)(v & 7);
        if (mode == 1) mant &= 0xFFull;                              // just above a power of two
        if (mode == 2) mant |= 0xFFFFFFFFFFF00ull;                   // just below
        if (mode == 3) mant = (mant & 0xFull) | ((v >> 20) % 21) * (0x10000000000000ull / 21);  // near k / 21 of the binade
        const int ex     = (int)((v >> 8) % 81) - 40;
        const uint64_t sg = (v >> 40) & 1;
        const uint64_t bits = (sg << 63) | ((uint64_t)(1023 + ex) << 52) | mant;
        const double x      = __longlong_as_double((long long)bits);
        check_upto<16>(x, bad, first);
        check<20>(x, bad, first);
    }
}

int main()
{
    unsigned long long *bad, *first;
    hipMalloc(&bad, 32 * 8);
    hipMalloc(&first, 32 * 8);
    hipMemset(bad, 0, 32 * 8);
    hipMemset(first, 0, 32 * 8);
    const int blocks = 4096, threads = 256, rounds = 4096;  // 4.3e9 operands, each through 17 divisors
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, bad, first, rounds, 12345ull);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    unsigned long long hb[32], hf[32];
    hipMemcpy(hb, bad, sizeof hb, hipMemcpyDeviceToHost);
    hipMemcpy(hf, first, sizeof hf, hipMemcpyDeviceToHost);
    unsigned long long total = 0;
    for (int n = 1; n <= 20; ++n)
        if (n <= 16 || n == 20)
        {
            total += hb[n];
            if (hb[n]) printf("N = %d: %llu mismatches, first x = 0x%016llx\n", n, hb[n], hf[n]);
        }
    printf("div_const probe: %.3g operands x 17 divisors, %llu mismatches\n", (double)blocks * threads * rounds, total);
    return total ? 1 : 0;
}
