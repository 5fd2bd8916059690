// Error text, version and handle base of the C-ABI library.
#include "common.hpp"

#include <cstdarg>
#include <map>
#include <mutex>
#include <utility>

namespace snk
{
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int set_max_lds_once(const void* kernel, int bytes)
{
    // The attribute belongs to the kernel's function object ON THE CURRENT DEVICE (every entry point has called hipSetDevice
    // for its handle's device before it gets here), so the "already set" state is kept per (device, kernel): one process may
    // hold handles on several GPUs.
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> done;
    int dev = 0;
    SNK_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, kernel);
    auto it        = done.find(key);
    if (it != done.end() && it->second >= bytes) return SNK_OK;
    SNK_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done[key] = bytes;
    return SNK_OK;
}

int HandleBase::init(int dev, void* user_stream)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    {
        set_error("no HIP device visible (hipGetDeviceCount)");
        return SNK_ERR_NO_DEVICE;
    }
    if (dev < 0 || dev >= n)
    {
        set_error("device ordinal %d out of range [0,%d)", dev, n);
        return SNK_ERR_INVALID_ARG;
    }
    device = dev;
    SNK_HIP_CHECK(hipSetDevice(dev));
    if (user_stream)
    {
        stream     = reinterpret_cast<hipStream_t>(user_stream);
        own_stream = false;
    }
    else
    {
        SNK_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        own_stream = true;
    }
    return SNK_OK;
}

void HandleBase::fini()
{
    if (own_stream && stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
}
}  // namespace snk

extern "C" {
const char* snk_last_error(void)
{
    return snk::g_err;
}
const char* snk_version(void)
{
    return "snake_hip 0.1 (gfx950)";
}
int snk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
}
