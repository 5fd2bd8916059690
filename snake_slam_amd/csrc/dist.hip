// Result gather across GPUs for a C++ Snake-SLAM process per GPU (BASELINE config 5): RCCL over xGMI, called
// directly -- no torch.distributed, no MPI.  The path shards over independent units (one sequence per GPU); nothing is exchanged
// while it runs.  At the end every rank holds a fixed-size block -- the TUM trajectory the reference writes per run
// (Snake/System/System.cpp:552-563: "timestamp tx ty tz qx qy qz qw" per frame) plus a few counters -- and ONE ncclAllGather puts all
// blocks on every rank.
//
// RCCL is opened with dlopen at the first snk_dist_* call that needs it, by SONAME (librccl.so.1): a process that already maps an
// RCCL (PyTorch bundles one) gets that copy, a plain C++ process gets /opt/rocm's through this library's RUNPATH.  A second RCCL
// next to PyTorch's would pull a second HIP runtime into the process (snake_slam_amd/_lib.py), and a link-time dependency would make
// the whole library unloadable on a box without RCCL; neither happens this way.  Single-GPU users never touch this file.
//
// One handle = one communicator + one stream + a device staging buffer and its pinned mirror; calls on a handle are serial (the
// threading contract of every seam).
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <mutex>
#include <thread>

#include "common.hpp"

using namespace snk;

namespace
{
// the slice of rccl.h this file uses (the header is not included: nothing here may need RCCL at link time)
typedef struct
{
    char internal[SNK_DIST_ID_BYTES];
} rccl_unique_id;
typedef void* rccl_comm;
enum
{
    RCCL_SUCCESS = 0,
    RCCL_UINT8   = 1,
    RCCL_INT64   = 4,
    RCCL_MAX     = 2
};
struct Rccl
{
    void* so = nullptr;
    int (*GetUniqueId)(rccl_unique_id*)                                                   = nullptr;
    int (*CommInitRank)(rccl_comm*, int, rccl_unique_id, int)                             = nullptr;
    int (*CommDestroy)(rccl_comm)                                                         = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm, hipStream_t)             = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t)        = nullptr;
    const char* (*GetErrorString)(int)                                                    = nullptr;
    int (*GetVersion)(int*)                                                               = nullptr;
    int (*CommCount)(rccl_comm, int*)                                                     = nullptr;  // optional: what RCCL itself says
    int (*CommUserRank)(rccl_comm, int*)                                                  = nullptr;  // optional
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_open()
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.so) return SNK_OK;
    // SNK_RCCL_LIB names the library to use; with SNK_RCCL_STRICT set nothing else is tried (a deployment that pins its RCCL build
    // wants an error, not another copy)
    const bool strict   = getenv("SNK_RCCL_STRICT") != nullptr && getenv("SNK_RCCL_LIB") != nullptr;
    const char* names[] = {getenv("SNK_RCCL_LIB"), strict ? nullptr : "librccl.so.1", strict ? nullptr : "librccl.so"};
    void* so            = nullptr;
    std::string tried;
    for (const char* n : names)
    {
        if (!n || !*n) continue;
        so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (so) break;
        const char* e = dlerror();  // one call: dlerror() clears the message it returns
        tried += std::string(n) + ": " + (e ? e : "?") + "; ";
    }
    if (!so)
    {
        set_error("RCCL not found (%s)", tried.c_str());
        return SNK_ERR_NO_DEVICE;
    }
    Rccl r;
    r.so = so;
#define SNK_SYM(field, name)                                                  \
    *reinterpret_cast<void**>(&r.field) = dlsym(so, name);                    \
    if (!r.field)                                                             \
    {                                                                         \
        set_error("RCCL: symbol %s missing", name);                           \
        dlclose(so);                                                          \
        return SNK_ERR_NO_DEVICE;                                             \
    }
    SNK_SYM(GetUniqueId, "ncclGetUniqueId")
    SNK_SYM(CommInitRank, "ncclCommInitRank")
    SNK_SYM(CommDestroy, "ncclCommDestroy")
    SNK_SYM(AllGather, "ncclAllGather")
    SNK_SYM(AllReduce, "ncclAllReduce")
    SNK_SYM(GetErrorString, "ncclGetErrorString")
    SNK_SYM(GetVersion, "ncclGetVersion")
#undef SNK_SYM
    *reinterpret_cast<void**>(&r.CommCount)    = dlsym(so, "ncclCommCount");
    *reinterpret_cast<void**>(&r.CommUserRank) = dlsym(so, "ncclCommUserRank");
    g_rccl = r;
    return SNK_OK;
}

#define SNK_RCCL_CHECK(expr)                                                                              \
    do                                                                                                    \
    {                                                                                                     \
        const int _r = (expr);                                                                            \
        if (_r != RCCL_SUCCESS)                                                                           \
        {                                                                                                 \
            set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__);     \
            return SNK_ERR_HIP;                                                                           \
        }                                                                                                 \
    } while (0)
}  // namespace

struct snk_dist : HandleBase
{
    rccl_comm comm = nullptr;
    int rank = 0, world = 1;
    DevBuf d_send, d_recv;
    HostBuf h_buf;  // [send | recv]
    std::string rendezvous;  // rank 0 of snk_dist_init_file: the file it wrote, removed in snk_dist_destroy
};

extern "C" {

int snk_dist_get_unique_id(uint8_t id[SNK_DIST_ID_BYTES])
{
    SNK_REQUIRE(id != nullptr, "id is NULL");
    int rc = rccl_open();
    if (rc != SNK_OK) return rc;
    rccl_unique_id u;
    memset(&u, 0, sizeof(u));
    SNK_RCCL_CHECK(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, SNK_DIST_ID_BYTES);
    return SNK_OK;
}

int snk_dist_init(const uint8_t id[SNK_DIST_ID_BYTES], int rank, int world, int device, snk_dist** out)
{
    SNK_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    SNK_REQUIRE(id != nullptr, "id is NULL");
    SNK_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank / world");
    int rc = rccl_open();
    if (rc != SNK_OK) return rc;
    snk_dist* d = new snk_dist();
    if ((rc = d->init(device, nullptr)) != SNK_OK)
    {
        delete d;
        return rc;
    }
    d->rank  = rank;
    d->world = world;
    rccl_unique_id u;
    memcpy(u.internal, id, SNK_DIST_ID_BYTES);
    const int r = g_rccl.CommInitRank(&d->comm, world, u, rank);  // collective: returns when all `world` ranks have called it
    if (r != RCCL_SUCCESS)
    {
        set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, g_rccl.GetErrorString(r));
        d->fini();
        delete d;
        return SNK_ERR_HIP;
    }
    *out = d;
    return SNK_OK;
}

// Rendezvous through a file every rank can see (one node: any local path): rank 0 writes the id to `<path>.tmp` and renames it
// to `path`; the others poll for it.  The reference has no launcher of its own, so this is what a per-GPU Snake-SLAM process uses
// when nothing else (MPI, a job scheduler's key-value store) hands the 128 bytes around.  Rank 0 removes the file in snk_dist_destroy.
int snk_dist_init_file(const char* path, int rank, int world, int device, double timeout_s, snk_dist** out)
{
    SNK_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    SNK_REQUIRE(path != nullptr && *path, "path");
    SNK_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank / world");
    uint8_t id[SNK_DIST_ID_BYTES];
    if (rank == 0)
    {
        // a file left behind by an earlier job (a crashed rank 0, a C caller that never destroyed its handle) would hand the other
        // ranks a dead id and hang them in ncclCommInitRank: remove it before the new id exists.  Ranks that read the stale file in
        // the window before this line is a launcher problem (start rank 0 first or use a per-job path); the header says so.
        (void)unlink(path);
        int rc = snk_dist_get_unique_id(id);
        if (rc != SNK_OK) return rc;
        const std::string tmp = std::string(path) + ".tmp";
        FILE* f               = fopen(tmp.c_str(), "wb");
        bool ok               = f != nullptr;
        if (f)
        {
            ok = fwrite(id, 1, sizeof(id), f) == sizeof(id);
            ok = (fclose(f) == 0) && ok;  // closed on every path
        }
        if (ok) ok = rename(tmp.c_str(), path) == 0;
        if (!ok)
        {
            (void)unlink(tmp.c_str());
            set_error("snk_dist_init_file: cannot write %s", path);
            return SNK_ERR_INVALID_ARG;
        }
    }
    else
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;)
        {
            struct stat st;
            if (stat(path, &st) == 0 && st.st_size == (off_t)sizeof(id))
            {
                FILE* f       = fopen(path, "rb");
                const bool ok = f && fread(id, 1, sizeof(id), f) == sizeof(id);
                if (f) fclose(f);
                if (ok) break;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > (timeout_s > 0 ? timeout_s : 60.0))
            {
                set_error("snk_dist_init_file: rank %d waited %.0f s for %s", rank, timeout_s > 0 ? timeout_s : 60.0, path);
                return SNK_ERR_INVALID_ARG;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
    }
    const int rc = snk_dist_init(id, rank, world, device, out);
    if (rank == 0)
    {
        if (rc == SNK_OK)
            (*out)->rendezvous = path;
        else
            (void)unlink(path);  // a failed init leaves nothing behind
    }
    return rc;
}

int snk_dist_destroy(snk_dist* d)
{
    if (!d) return SNK_OK;
    (void)hipSetDevice(d->device);
    (void)hipStreamSynchronize(d->stream);
    if (d->comm) (void)g_rccl.CommDestroy(d->comm);
    if (!d->rendezvous.empty()) (void)unlink(d->rendezvous.c_str());  // every rank has joined the communicator: the id is spent
    d->d_send.release();
    d->d_recv.release();
    d->h_buf.release();
    d->fini();
    delete d;
    return SNK_OK;
}

int snk_dist_rank(const snk_dist* d, int* rank, int* world)
{
    SNK_REQUIRE(d != nullptr, "dist is NULL");
    int r = d->rank, w = d->world;
    // what the COMMUNICATOR says (ncclCommCount / ncclCommUserRank), not what the caller passed in: a launcher check that all `world`
    // ranks really joined one RCCL communicator (tools/run_all_gpus.py, tests/cpp/dist_driver.cpp)
    if (d->comm && g_rccl.CommCount && g_rccl.CommUserRank)
    {
        SNK_RCCL_CHECK(g_rccl.CommCount(d->comm, &w));
        SNK_RCCL_CHECK(g_rccl.CommUserRank(d->comm, &r));
        if (w != d->world || r != d->rank)
        {
            set_error("snk_dist_rank: the communicator reports rank %d of %d, the handle was created as rank %d of %d", r, w, d->rank, d->world);
            return SNK_ERR_HIP;
        }
    }
    if (rank) *rank = r;
    if (world) *world = w;
    return SNK_OK;
}

// every rank's `bytes` bytes at send_dev -> recv_dev[r * bytes ...) on every rank, device memory, on the handle's stream; returns
// after the stream has drained (the blocks are results: the caller reads them next)
int snk_dist_all_gather_dev(snk_dist* d, const void* send_dev, size_t bytes, void* recv_dev)
{
    SNK_REQUIRE(d != nullptr && send_dev != nullptr && recv_dev != nullptr, "NULL argument");
    SNK_HIP_CHECK(hipSetDevice(d->device));
    if (bytes == 0) return SNK_OK;
    SNK_RCCL_CHECK(g_rccl.AllGather(send_dev, recv_dev, bytes, RCCL_UINT8, d->comm, d->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(d->stream));
    return SNK_OK;
}

// the same for host memory (what the reference holds: std::vector of poses): one pinned staging buffer, upload, ncclAllGather,
// download, one synchronisation
int snk_dist_all_gather(snk_dist* d, const void* send, size_t bytes, void* recv)
{
    SNK_REQUIRE(d != nullptr && (bytes == 0 || (send != nullptr && recv != nullptr)), "NULL argument");
    SNK_HIP_CHECK(hipSetDevice(d->device));
    if (bytes == 0) return SNK_OK;
    const size_t all = bytes * (size_t)d->world;
    int rc;
    if ((rc = d->d_send.reserve(bytes)) != SNK_OK) return rc;
    if ((rc = d->d_recv.reserve(all)) != SNK_OK) return rc;
    if ((rc = d->h_buf.reserve(bytes + all)) != SNK_OK) return rc;
    char* h = d->h_buf.as<char>();
    memcpy(h, send, bytes);
    SNK_HIP_CHECK(hipMemcpyAsync(d->d_send.p, h, bytes, hipMemcpyHostToDevice, d->stream));
    SNK_RCCL_CHECK(g_rccl.AllGather(d->d_send.p, d->d_recv.p, bytes, RCCL_UINT8, d->comm, d->stream));
    SNK_HIP_CHECK(hipMemcpyAsync(h + bytes, d->d_recv.p, all, hipMemcpyDeviceToHost, d->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(d->stream));
    memcpy(recv, h + bytes, all);
    return SNK_OK;
}

// max over the ranks of one int64 per rank (the longest trajectory: blocks are padded to it) -- also serves as a barrier
int snk_dist_max_i64(snk_dist* d, int64_t value, int64_t* out)
{
    SNK_REQUIRE(d != nullptr && out != nullptr, "NULL argument");
    SNK_HIP_CHECK(hipSetDevice(d->device));
    int rc;
    if ((rc = d->d_send.reserve(sizeof(int64_t))) != SNK_OK) return rc;
    if ((rc = d->h_buf.reserve(2 * sizeof(int64_t))) != SNK_OK) return rc;
    int64_t* h = d->h_buf.as<int64_t>();
    h[0]       = value;
    SNK_HIP_CHECK(hipMemcpyAsync(d->d_send.p, h, sizeof(int64_t), hipMemcpyHostToDevice, d->stream));
    SNK_RCCL_CHECK(g_rccl.AllReduce(d->d_send.p, d->d_send.p, 1, RCCL_INT64, RCCL_MAX, d->comm, d->stream));
    SNK_HIP_CHECK(hipMemcpyAsync(h + 1, d->d_send.p, sizeof(int64_t), hipMemcpyDeviceToHost, d->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(d->stream));
    *out = h[1];
    return SNK_OK;
}

int snk_dist_rccl_version(int* version)
{
    SNK_REQUIRE(version != nullptr, "NULL argument");
    int rc = rccl_open();
    if (rc != SNK_OK) return rc;
    SNK_RCCL_CHECK(g_rccl.GetVersion(version));
    return SNK_OK;
}
}
