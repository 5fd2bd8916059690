// Hardware probe (gfx950): are f16 DENORMAL bit patterns (0x00rr, r = a byte) ordered correctly by
// v_pk_minimum3_f16 / v_pk_maximum3_f16 (i.e. not flushed to zero), and do ds_read_u8_d16 / _d16_hi fill the two halves
// of a register as documented?  fast_kernel's exact-score network relies on both.   hipcc --offload-arch=gfx950 -O2 pk_min3_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void probe(const unsigned* a, const unsigned* b, const unsigned* c, unsigned* mn, unsigned* mx, const unsigned char* bytes,
                      unsigned* packed)
{
    const int i = threadIdx.x;
    unsigned o1, o2;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(o1) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(o2) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    mn[i] = o1;
    mx[i] = o2;
    __shared__ unsigned char t[256];
    t[i]       = bytes[i];
    t[i + 64]  = bytes[i + 64];
    t[i + 128] = bytes[i + 128];
    t[i + 192] = bytes[i + 192];
    __syncthreads();
    unsigned p;
    const unsigned addr = (unsigned)(size_t)(t + i);  // LDS byte address of the lane (the LDS aperture's low 32 bits)
    asm volatile("ds_read_u8_d16 %0, %1 offset:3\n\tds_read_u8_d16_hi %0, %1 offset:131\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p) : "v"(addr) : "memory");
    packed[i] = p;
}

int main()
{
    unsigned ha[64], hb[64], hc[64], hmn[64], hmx[64], hp[64];
    unsigned char hbytes[256];
    srand(7);
    for (int i = 0; i < 256; ++i) hbytes[i] = (unsigned char)(rand() & 255);
    for (int i = 0; i < 64; ++i)
    {
        auto r = []() { return (unsigned)(rand() & 255); };
        ha[i] = r() | (r() << 16);
        hb[i] = r() | (r() << 16);
        hc[i] = r() | (r() << 16);
    }
    ha[0] = 0; hb[0] = 0x00010001u; hc[0] = 0x00ff00ffu;  // zero, smallest denormal, 255
    unsigned *a, *b, *c, *mn, *mx, *p;
    unsigned char* by;
    (void)hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&c, 256); hipMalloc(&mn, 256); hipMalloc(&mx, 256); hipMalloc(&p, 256); hipMalloc(&by, 256);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice); hipMemcpy(c, hc, 256, hipMemcpyHostToDevice);
    hipMemcpy(by, hbytes, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, c, mn, mx, by, p);
    hipMemcpy(hmn, mn, 256, hipMemcpyDeviceToHost); hipMemcpy(hmx, mx, 256, hipMemcpyDeviceToHost); hipMemcpy(hp, p, 256, hipMemcpyDeviceToHost);
    int bad = 0, bad_min = 0, bad_max = 0, bad_d16 = 0;
    for (int i = 0; i < 64; ++i)
    {
        auto m3 = [](unsigned x, unsigned y, unsigned z, bool mx) {
            unsigned lo[3] = {x & 0xffff, y & 0xffff, z & 0xffff}, hi[3] = {x >> 16, y >> 16, z >> 16};
            unsigned l = lo[0], h = hi[0];
            for (int k = 1; k < 3; ++k) { l = mx ? (lo[k] > l ? lo[k] : l) : (lo[k] < l ? lo[k] : l); h = mx ? (hi[k] > h ? hi[k] : h) : (hi[k] < h ? hi[k] : h); }
            return l | (h << 16);
        };
        if (hmn[i] != m3(ha[i], hb[i], hc[i], false)) ++bad, ++bad_min;
        if (hmx[i] != m3(ha[i], hb[i], hc[i], true)) ++bad, ++bad_max;
        if (hp[i] != ((unsigned)hbytes[i + 3] | ((unsigned)hbytes[i + 131] << 16)))
        {
            ++bad, ++bad_d16;
            if (bad_d16 < 4) printf("lane %d: d16 got %08x want %08x\n", i, hp[i], (unsigned)hbytes[i + 3] | ((unsigned)hbytes[i + 131] << 16));
        }
    }
    printf("min3 bad %d, max3 bad %d, d16 bad %d\n", bad_min, bad_max, bad_d16);
    printf("pk_min3_probe: %s (%d mismatches; lane 0: min %08x max %08x, d16 %08x)\n", bad ? "FAIL" : "PASS", bad, hmn[0], hmx[0], hp[0]);
    return bad ? 1 : 0;
}
