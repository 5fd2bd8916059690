// Probe (gfx950): what a row-streaming wavefront (one row load + one row store per step, 7 loads in flight, bands of 64 rows with a
// 6-row halo -- the access pattern of level_kernel) reaches in bytes/s as a function of the bytes a LANE moves per access
// (4 = dword, 8 = dwordx2, 16 = dwordx4).    hipcc --offload-arch=gfx950 -O3 stream_width_probe.hip -o stream_width_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32;
template <int N>
struct Vec
{
    u32 v[N];
};
template <int N>
__global__ __launch_bounds__(256) void stream(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int pitch, int h,
                                              long long stride, int strips, int bands)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + wave;
    if (u >= strips * bands) return;
    const int band = u / strips, strip = u - band * strips;
    const unsigned char* s = src + blockIdx.y * stride + strip * (64 * 4 * N) + lane * (4 * N);
    unsigned char* d       = dst + blockIdx.y * stride + strip * (64 * 4 * N) + lane * (4 * N);
    const int y0 = band * 64;
    Vec<N> r[7];
    auto ld = [&](int k) { const int y = min(max(y0 - 3 + k, 0), h - 1); return *reinterpret_cast<const Vec<N>*>(s + (long long)y * pitch); };
#pragma unroll
    for (int k = 0; k < 7; ++k) r[k] = ld(k);
    for (int k0 = 0; k0 < 70; k0 += 7)
#pragma unroll
        for (int kk = 0; kk < 7; ++kk)
        {
            const int k = k0 + kk;
            Vec<N> x = r[kk];
            r[kk]    = ld(min(k + 7, 69));
            const int yo = y0 - 3 + k - 3;
            if (k >= 6 && yo < min(y0 + 64, h))
            {
#pragma unroll
                for (int i = 0; i < N; ++i) x.v[i] = x.v[i] * 3u + 1u;
                *reinterpret_cast<Vec<N>*>(d + (long long)yo * pitch) = x;
            }
        }
}
template <int N>
double run(const unsigned char* s, unsigned char* d, int w, int h, int B)
{
    const int strips = w / (64 * 4 * N), bands = (h + 63) / 64;
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    const dim3 grid((strips * bands + 3) / 4, B);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stream<N>, grid, dim3(256), 0, 0, s, d, w, h, (long long)w * h, strips, bands);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(stream<N>, grid, dim3(256), 0, 0, s, d, w, h, (long long)w * h, strips, bands);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return 2.0 * w * h * B * 10 / (ms * 1e-3) / 1e12;  // algorithmic read + write TB/s
}
int main()
{
    const int w = 1024, h = 512, B = 512;
    unsigned char *s, *d;
    hipMalloc(&s, (size_t)w * h * B);
    hipMalloc(&d, (size_t)w * h * B);
    hipMemset(s, 1, (size_t)w * h * B);
    printf("stream_width_probe (1024 x 512 x 512 images, read + write): 4 B/lane %.2f TB/s, 8 B/lane %.2f TB/s, 16 B/lane %.2f TB/s\n",
           run<1>(s, d, w, h, B), run<2>(s, d, w, h, B), run<4>(s, d, w, h, B));
    return 0;
}
