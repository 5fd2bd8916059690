// Error text, version and handle base of the C-ABI library.
#include "common.hpp"

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

namespace
{
struct DefInfo
{
    const char* name;
    int lo, hi, dflt;
};
const DefInfo k_defs[DEF_COUNT] = {
    {"bf_filter.threshold_strict", 0, 1, 0},
    {"bf_filter.ratio_strict", 0, 1, 0},
    {"iround.mode", 0, 2, 0},
    {"orb.response", 0, 1, 0},
};
std::atomic<int> g_defs[DEF_COUNT];
std::once_flag g_defs_once;

int def_index(const char* key)
{
    if (!key) return -1;
    for (int i = 0; i < DEF_COUNT; ++i)
        if (strcmp(k_defs[i].name, key) == 0) return i;
    return -1;
}
void defs_init()
{
    std::call_once(g_defs_once,
                   []
                   {
                       for (int i = 0; i < DEF_COUNT; ++i) g_defs[i].store(k_defs[i].dflt);
                       // SNK_DEFINITIONS="iround.mode=2,bf_filter.ratio_strict=1": start-up values (bad entries are reported and ignored)
                       const char* e = getenv("SNK_DEFINITIONS");
                       if (!e) return;
                       std::string all(e);
                       size_t pos = 0;
                       while (pos < all.size())
                       {
                           size_t end = all.find(',', pos);
                           if (end == std::string::npos) end = all.size();
                           const std::string item = all.substr(pos, end - pos);
                           pos                    = end + 1;
                           const size_t eq        = item.find('=');
                           const int i            = eq == std::string::npos ? -1 : def_index(item.substr(0, eq).c_str());
                           const int v            = eq == std::string::npos ? 0 : atoi(item.c_str() + eq + 1);
                           if (i < 0 || v < k_defs[i].lo || v > k_defs[i].hi)
                               fprintf(stderr, "snake_hip: SNK_DEFINITIONS entry '%s' ignored\n", item.c_str());
                           else
                               g_defs[i].store(v);
                       }
                   });
}
}  // namespace

int definition(DefKey k)
{
    defs_init();
    return g_defs[k].load(std::memory_order_relaxed);
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
int snk_set_definition(const char* key, int value)
{
    snk::defs_init();
    const int i = snk::def_index(key);
    if (i < 0)
    {
        snk::set_error("snk_set_definition: unknown key '%s'", key ? key : "(null)");
        return SNK_ERR_INVALID_ARG;
    }
    if (value < snk::k_defs[i].lo || value > snk::k_defs[i].hi)
    {
        snk::set_error("snk_set_definition: %s must be %d..%d, got %d", key, snk::k_defs[i].lo, snk::k_defs[i].hi, value);
        return SNK_ERR_INVALID_ARG;
    }
    snk::g_defs[i].store(value);
    return SNK_OK;
}
int snk_get_definition(const char* key, int* value)
{
    snk::defs_init();
    const int i = snk::def_index(key);
    if (i < 0 || !value)
    {
        snk::set_error("snk_get_definition: unknown key '%s' / NULL output", key ? key : "(null)");
        return SNK_ERR_INVALID_ARG;
    }
    *value = snk::g_defs[i].load();
    return SNK_OK;
}
int snk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
}
