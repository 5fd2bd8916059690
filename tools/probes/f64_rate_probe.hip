// Hardware probe (gfx950): fp64 throughput of the vector ALU (v_fma_f64) against the matrix cores
// (v_mfma_f64_16x16x4_f64), and the operand / result layout of the latter.  Evidence for the choice of the Schur
// complement's arithmetic unit (north_star: "MFMA ... each choice evidenced").
//   hipcc --offload-arch=gfx950 -O2 f64_rate_probe.hip -o f64_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fma_rate(double* out, int iters)
{
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * 1e-3 + k;
    const double b = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], b, c);
    double s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void mfma_rate(double* out, int iters)
{
    double4_t acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = double4_t{0, 0, 0, 0};
    const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
    double s = 0;
    for (int k = 0; k < 4; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// D = A (16 x 4) * B (4 x 16): lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; receives D[(l >> 4) + 4 j][l & 15], j = 0..3
__global__ void mfma_layout(const double* A, const double* B, double* D)
{
    const int l = threadIdx.x;
    double4_t acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D[((l >> 4) + 4 * j) * 16 + (l & 15)] = acc[j];
}

int main()
{
    const int blocks = 256 * 8, iters = 20000;
    double* out;
    (void)hipMalloc(&out, blocks * 256 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; ++rep)
    {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(fma_rate, dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double fma_tf = 2.0 * blocks * 256.0 * 8 * iters / (ms * 1e-3) / 1e12;
    printf("v_fma_f64:               %.1f TFLOP/s  (%.3f ms)\n", fma_tf, ms);
    for (int rep = 0; rep < 2; ++rep)
    {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_rate, dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double mfma_tf = 2.0 * 16 * 16 * 4 * (double)blocks * 4 /*waves*/ * 4 * iters / (ms * 1e-3) / 1e12;
    printf("v_mfma_f64_16x16x4_f64:  %.1f TFLOP/s  (%.3f ms)  ratio %.2f\n", mfma_tf, ms, mfma_tf / fma_tf);
    // layout
    std::vector<double> hA(64), hB(64), hD(256), want(256, 0.0);
    for (int i = 0; i < 64; ++i) hA[i] = 1 + i * 0.5, hB[i] = 2 - i * 0.25;
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c)
            for (int k = 0; k < 4; ++k) want[r * 16 + c] += hA[r * 4 + k] * hB[k * 16 + c];
    double *dA, *dB, *dD;
    (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dD, 2048);
    (void)hipMemcpy(dA, hA.data(), 512, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(hD.data(), dD, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += hD[i] != want[i];
    printf("layout A[l&15][l>>4], B[l>>4][l&15], D[(l>>4)+4j][l&15]: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    return bad ? 1 : 0;
}
