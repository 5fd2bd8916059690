// Probe (gfx950): (1) global_load_dwordx3 from a byte-aligned address returns the bytes at that address;
// (2) global_load_lds_dwordx4 from a 4-byte-aligned (not 16-byte-aligned) global address lands at LDS base + 16 * lane,
//     lanes switched off by EXEC write nothing.      hipcc --offload-arch=gfx950 -O3 lds_direct_probe.hip -o lds_direct_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned u32;
typedef u32 u32x3_a1 __attribute__((ext_vector_type(3), aligned(1)));
__global__ void k(const unsigned char* __restrict__ g, u32* out, int off)
{
    __shared__ __attribute__((aligned(16))) u32 buf[64 * 4 * 2];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) buf[i] = 0xDEADBEEFu;
    __syncthreads();
    const u32x3_a1 v = *reinterpret_cast<const u32x3_a1*>(g + off + 12 * lane);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 4 + 16 * lane),
                                     (__attribute__((address_space(3))) void*)(buf), 16, 0, 0);
    if (lane < 40)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 4 + 16 * lane + 1024),
                                         (__attribute__((address_space(3))) void*)(buf + 256), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    out[3 * lane] = v.x, out[3 * lane + 1] = v.y, out[3 * lane + 2] = v.z;
    for (int i = lane; i < 512; i += 64) out[192 + i] = buf[i];
}
int main()
{
    std::vector<unsigned char> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (unsigned char)(i * 7 + (i >> 8));
    unsigned char* d;
    u32* o;
    hipMalloc(&d, 4096);
    hipMalloc(&o, 4 * (192 + 512));
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    int bad = 0;
    for (int off = 0; off < 4; ++off)
    {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, off);
        std::vector<u32> r(192 + 512);
        hipMemcpy(r.data(), o, 4 * r.size(), hipMemcpyDeviceToHost);
        for (int i = 0; i < 192; ++i)
        {
            u32 w;
            memcpy(&w, &h[off + 4 * i], 4);
            bad += r[i] != w;
        }
        for (int i = 0; i < 512; ++i)
        {
            u32 w = 0xDEADBEEFu;
            if (i < 256) memcpy(&w, &h[4 + 4 * i], 4);
            else if (i < 256 + 160) memcpy(&w, &h[4 + 1024 + 4 * (i - 256)], 4);
            bad += r[192 + i] != w;
        }
    }
    printf("lds_direct_probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
