// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of level_kernel (MI355X_MICROARCH.md, section HBM:
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel
// moves a KNOWN number of bytes over a 1.5 GB buffer (past the 256 MB Infinity Cache), in rows of `pitch` = 768 bytes like a 752-px level:
//   write_dword_256   a wavefront writes 64 lanes x 4 B = 256 aligned bytes per row          (3 strips cover 768 B)
//   write_dword_188   a wavefront writes 47 lanes x 4 B = 188 bytes per row at 188 * strip   (level_kernel's level-0 strips: 4 x 188 = 752)
//   write_quad_1024   16 B per lane, 1024 aligned bytes per wavefront instruction
//   read_dword_256    256 aligned bytes per row with one dword per lane (level_kernel's row loads), one dword written per wavefront
//   read_quad_1024    16 B per lane
// usage: hbm_counter_probe   (run under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE; prints the bytes each kernel moved)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(e)                                                                      \
    do                                                                                \
    {                                                                                 \
        hipError_t _e = (e);                                                          \
        if (_e != hipSuccess)                                                         \
        {                                                                             \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e));                   \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int PITCH = 768, ROWS_PER_BAND = 64;

// unit u = (band, strip): strip-major inside a band like level_kernel (adjacent strips of a band share a workgroup)
template <int STRIPS, int LANES, int STRIDE>
__global__ __launch_bounds__(256) void write_dword(unsigned char* buf, int n_units)
{
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (u >= n_units) return;
    const int band = u / STRIPS, strip = u - band * STRIPS;
    unsigned char* p = buf + (size_t)band * ROWS_PER_BAND * PITCH + strip * STRIDE + 4 * lane;
    if (lane < LANES)
        for (int r = 0; r < ROWS_PER_BAND; ++r) *reinterpret_cast<unsigned*>(p + (size_t)r * PITCH) = (unsigned)(u + r);
}
__global__ __launch_bounds__(256) void write_quad(uint4* buf, size_t n_quads)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_quads) buf[i] = uint4{1u, 2u, 3u, (unsigned)i};
}
__global__ __launch_bounds__(256) void read_dword(const unsigned char* buf, int n_units, unsigned* out)
{
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (u >= n_units) return;
    const int band = u / 3, strip = u - band * 3;
    const unsigned char* p = buf + (size_t)band * ROWS_PER_BAND * PITCH + strip * 256 + 4 * lane;
    unsigned s = 0;
    for (int r = 0; r < ROWS_PER_BAND; ++r) s += *reinterpret_cast<const unsigned*>(p + (size_t)r * PITCH);
    if (s == 0x12345678u) out[u] = s;  // never: keeps the loads
}
__global__ __launch_bounds__(256) void read_quad(const uint4* buf, size_t n_quads, unsigned* out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_quads) return;
    const uint4 v = buf[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) out[0] = v.x;
}

int main()
{
    const size_t bytes = (size_t)1536 << 20;
    unsigned char* buf;
    unsigned* out;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&out, 64 << 20));
    CHECK(hipMemset(buf, 1, bytes));
    const int bands = (int)(bytes / ((size_t)ROWS_PER_BAND * PITCH));
    for (int rep = 0; rep < 2; ++rep)
    {
        hipLaunchKernelGGL((write_dword<3, 64, 256>), dim3((bands * 3 + 3) / 4), dim3(256), 0, 0, buf, bands * 3);
        hipLaunchKernelGGL((write_dword<4, 47, 188>), dim3((bands * 4 + 3) / 4), dim3(256), 0, 0, buf, bands * 4);
        // round 5: the same 752 bytes of a row as strips that start on 64-byte / 32-byte boundaries, and the strip widths of levels 1-3
        hipLaunchKernelGGL((write_dword<4, 48, 192>), dim3((bands * 4 + 3) / 4), dim3(256), 0, 0, buf, bands * 4);
        hipLaunchKernelGGL((write_dword<3, 56, 224>), dim3((bands * 3 + 3) / 4), dim3(256), 0, 0, buf, bands * 3);
        hipLaunchKernelGGL((write_dword<3, 53, 212>), dim3((bands * 3 + 3) / 4), dim3(256), 0, 0, buf, bands * 3);
        hipLaunchKernelGGL((write_dword<3, 44, 176>), dim3((bands * 3 + 3) / 4), dim3(256), 0, 0, buf, bands * 3);
        hipLaunchKernelGGL((write_dword<2, 55, 220>), dim3((bands * 2 + 3) / 4), dim3(256), 0, 0, buf, bands * 2);
        hipLaunchKernelGGL((write_dword<2, 61, 244>), dim3((bands * 2 + 3) / 4), dim3(256), 0, 0, buf, bands * 2);
        hipLaunchKernelGGL((write_dword<2, 60, 240>), dim3((bands * 2 + 3) / 4), dim3(256), 0, 0, buf, bands * 2);
        hipLaunchKernelGGL(write_quad, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<uint4*>(buf), bytes / 16);
        hipLaunchKernelGGL(read_dword, dim3((bands * 3 + 3) / 4), dim3(256), 0, 0, buf, bands * 3, out);
        hipLaunchKernelGGL(read_quad, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<const uint4*>(buf), bytes / 16, out);
    }
    CHECK(hipDeviceSynchronize());
    printf("{\"bands\": %d, \"bytes\": {\"write_dword<3, 64, 256>\": %zu, \"write_dword<4, 47, 188>\": %zu, \"write_quad\": %zu, \"read_dword\": %zu, \"read_quad\": %zu}}\n", bands,
           (size_t)bands * ROWS_PER_BAND * 768, (size_t)bands * ROWS_PER_BAND * 752, bytes, (size_t)bands * ROWS_PER_BAND * 768, bytes);
    return 0;
}
