// comm.hip -- the walker shards' communicator: RCCL over xGMI, inside libmcmc_hip.so.
//
// One process per GPU; the only data-path collective of the sampler is ONE all-reduce(sum) of
// the pooled sufficient statistics per learn / convergence checkpoint, which
// replaces the reference's gather -> root arithmetic -> broadcast round trip
// (cobaya/samplers/mcmc/mcmc.py:791-793 `mpi.array_gather`, :1005-1007 and :1021 `mpi.share`;
// cobaya/mpi.py:178-191).  The communicator is a process-level object (it must exist before
// the first engine: the job's seed is agreed with it, sampler.py:369-384) that engines attach
// with mcmc_hip_set_comm; the device checkpoint then queues `ncclAllReduce` IN PLACE on the
// engine's stream between its payload and solve kernels -- no host bounce, no host
// synchronisation, and no PyTorch anywhere on the path.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first communicator call): a
// single-GPU run never maps the 570 MB library, and a process that already holds RCCL (PyTorch
// ships the same SONAME) shares that copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/mcmc_hip.h"
#include "comm.h"

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error, version;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
thread_local std::string g_comm_error;   // failures before a communicator exists

void load_rccl()
{
    const char* names[] = {getenv("MCMC_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so",
                           "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        g_rccl.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.so) break;
        g_rccl.error = dlerror();
    }
    if (!g_rccl.so) return;
    auto sym = [](const char* s) { return dlsym(g_rccl.so, s); };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.GetVersion = (decltype(g_rccl.GetVersion))sym("ncclGetVersion");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce ||
        !g_rccl.GetErrorString) {
        g_rccl.error = "librccl lacks a symbol of the NCCL 2 API";
        dlclose(g_rccl.so);
        g_rccl.so = nullptr;
        return;
    }
    int v = 0;
    if (g_rccl.GetVersion && g_rccl.GetVersion(&v) == ncclSuccess) {
        char buf[64];
        snprintf(buf, sizeof buf, "RCCL %d.%d.%d", v / 10000, (v / 100) % 100, v % 100);
        g_rccl.version = buf;
    } else {
        g_rccl.version = "RCCL (version unknown)";
    }
}

bool have_rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.so != nullptr;
}

}  // namespace

struct mcmc_hip_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1, device = 0;
    hipStream_t stream = nullptr;        // host-buffer reductions (setup, counters, bench clock)
    double* pin = nullptr;               // pinned staging of those
    double* dev = nullptr;
    size_t cap = 0;
    std::string err;
};

namespace {

int cfail(mcmc_hip_comm* c, int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_comm_error = buf;
    return code;
}

int reserve(mcmc_hip_comm* c, size_t n)
{
    if (n <= c->cap) return MCMC_HIP_OK;
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->dev) (void)hipFree(c->dev);
    c->pin = c->dev = nullptr;
    c->cap = 0;
    size_t cap = 4096;
    while (cap < n) cap *= 2;
    if (hipHostMalloc((void**)&c->pin, sizeof(double) * cap, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&c->dev, sizeof(double) * cap) != hipSuccess)
        return cfail(c, MCMC_HIP_ERR_DEVICE, "communicator staging buffers (%zu doubles) could not be allocated", cap);
    c->cap = cap;
    return MCMC_HIP_OK;
}

}  // namespace

// ---- what capi.hip uses (comm.h) ----------------------------------------------------------------
int mcmc_comm_allreduce_on_stream(mcmc_hip_comm* c, double* dev, size_t n, int op, hipStream_t st)
{
    if (!c || !c->comm) return MCMC_HIP_ERR_STATE;
    const ncclResult_t r = g_rccl.AllReduce(dev, dev, n, ncclDouble, op == 1 ? ncclMax : ncclSum,
                                            c->comm, st);
    if (r != ncclSuccess)
        return cfail(c, MCMC_HIP_ERR_DEVICE, "ncclAllReduce of %zu doubles failed: %s", n,
                     g_rccl.GetErrorString(r));
    return MCMC_HIP_OK;
}

int mcmc_comm_size(const mcmc_hip_comm* c) { return c ? c->size : 1; }
int mcmc_comm_device(const mcmc_hip_comm* c) { return c ? c->device : -1; }
const char* mcmc_comm_error(const mcmc_hip_comm* c) { return c ? c->err.c_str() : g_comm_error.c_str(); }

// ---- C ABI ---------------------------------------------------------------------------------------
extern "C" {

const char* mcmc_hip_comm_version(void)
{
    return have_rccl() ? g_rccl.version.c_str() : "";
}

const char* mcmc_hip_comm_last_error(const mcmc_hip_comm* c) { return mcmc_comm_error(c); }

int mcmc_hip_comm_unique_id(uint8_t id[MCMC_HIP_COMM_ID_BYTES])
{
    static_assert(MCMC_HIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is an ncclUniqueId");
    if (!id) return cfail(nullptr, MCMC_HIP_ERR_ARG, "null argument");
    if (!have_rccl())
        return cfail(nullptr, MCMC_HIP_ERR_DEVICE, "RCCL could not be loaded: %s", g_rccl.error.c_str());
    ncclUniqueId u;
    const ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess)
        return cfail(nullptr, MCMC_HIP_ERR_DEVICE, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
    std::memcpy(id, u.internal, MCMC_HIP_COMM_ID_BYTES);
    return MCMC_HIP_OK;
}

int mcmc_hip_comm_create(const uint8_t id[MCMC_HIP_COMM_ID_BYTES], int32_t rank, int32_t n_ranks,
                         int32_t device, mcmc_hip_comm** out)
{
    if (!id || !out) return cfail(nullptr, MCMC_HIP_ERR_ARG, "null argument");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return cfail(nullptr, MCMC_HIP_ERR_ARG, "rank %d of %d", rank, n_ranks);
    if (!have_rccl())
        return cfail(nullptr, MCMC_HIP_ERR_DEVICE, "RCCL could not be loaded: %s", g_rccl.error.c_str());
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return cfail(nullptr, MCMC_HIP_ERR_DEVICE, "device %d is not available (%d visible)", device, ndev);
    if (hipSetDevice(device) != hipSuccess)
        return cfail(nullptr, MCMC_HIP_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    mcmc_hip_comm* c = new mcmc_hip_comm();
    c->rank = rank;
    c->size = n_ranks;
    c->device = device;
    ncclUniqueId u;
    std::memcpy(u.internal, id, MCMC_HIP_COMM_ID_BYTES);
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, n_ranks, u, rank);
    if (r != ncclSuccess) {
        cfail(nullptr, MCMC_HIP_ERR_DEVICE,
              "ncclCommInitRank(rank %d of %d, device %d) failed: %s (one process per GPU: two "
              "ranks cannot share a device)", rank, n_ranks, device, g_rccl.GetErrorString(r));
        delete c;
        return MCMC_HIP_ERR_DEVICE;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        cfail(nullptr, MCMC_HIP_ERR_DEVICE, "hipStreamCreate failed");
        g_rccl.CommDestroy(c->comm);
        delete c;
        return MCMC_HIP_ERR_DEVICE;
    }
    *out = c;
    return MCMC_HIP_OK;
}

void mcmc_hip_comm_destroy(mcmc_hip_comm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) g_rccl.CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->dev) (void)hipFree(c->dev);
    delete c;
}

int mcmc_hip_comm_rank(const mcmc_hip_comm* c) { return c ? c->rank : 0; }
int mcmc_hip_comm_size(const mcmc_hip_comm* c) { return c ? c->size : 1; }

int mcmc_hip_comm_allreduce(mcmc_hip_comm* c, double* buf, int64_t n, int32_t op)
{
    if (!c || !buf || n < 0 || (op != 0 && op != 1)) return cfail(c, MCMC_HIP_ERR_ARG, "invalid argument");
    if (n == 0) return MCMC_HIP_OK;
    if (hipSetDevice(c->device) != hipSuccess) return cfail(c, MCMC_HIP_ERR_DEVICE, "hipSetDevice failed");
    if (int rc = reserve(c, (size_t)n)) return rc;
    std::memcpy(c->pin, buf, sizeof(double) * (size_t)n);
    hipError_t e = hipMemcpyAsync(c->dev, c->pin, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        if (int rc = mcmc_comm_allreduce_on_stream(c, c->dev, (size_t)n, op, c->stream)) return rc;
        e = hipMemcpyAsync(c->pin, c->dev, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return cfail(c, MCMC_HIP_ERR_DEVICE, "all-reduce staging failed: %s", hipGetErrorString(e));
    std::memcpy(buf, c->pin, sizeof(double) * (size_t)n);
    return MCMC_HIP_OK;
}

int mcmc_hip_comm_allreduce_device(mcmc_hip_comm* c, uint64_t device_ptr, int64_t n, int32_t op,
                                   uint64_t stream)
{
    if (!c || !device_ptr || n < 0 || (op != 0 && op != 1)) return cfail(c, MCMC_HIP_ERR_ARG, "invalid argument");
    if (n == 0) return MCMC_HIP_OK;
    if (hipSetDevice(c->device) != hipSuccess) return cfail(c, MCMC_HIP_ERR_DEVICE, "hipSetDevice failed");
    return mcmc_comm_allreduce_on_stream(c, (double*)(uintptr_t)device_ptr, (size_t)n, op,
                                         stream ? (hipStream_t)(uintptr_t)stream : c->stream);
}

int mcmc_hip_comm_time_allreduce(mcmc_hip_comm* c, int64_t n, int32_t reps, double* us_per_call)
{
    if (!c || n < 1 || reps < 1 || !us_per_call) return cfail(c, MCMC_HIP_ERR_ARG, "invalid argument");
    if (hipSetDevice(c->device) != hipSuccess) return cfail(c, MCMC_HIP_ERR_DEVICE, "hipSetDevice failed");
    if (int rc = reserve(c, (size_t)n)) return rc;
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t e = hipMemsetAsync(c->dev, 0, sizeof(double) * (size_t)n, c->stream);
    if (e == hipSuccess) e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    int rc = MCMC_HIP_OK;
    if (e == hipSuccess) rc = mcmc_comm_allreduce_on_stream(c, c->dev, (size_t)n, 0, c->stream);   // (untimed: connects)
    if (e == hipSuccess && !rc) e = hipEventRecord(a, c->stream);
    for (int i = 0; i < reps && e == hipSuccess && !rc; ++i)
        rc = mcmc_comm_allreduce_on_stream(c, c->dev, (size_t)n, 0, c->stream);
    if (e == hipSuccess && !rc) e = hipEventRecord(b, c->stream);
    if (e == hipSuccess && !rc) e = hipEventSynchronize(b);
    float ms = 0.f;
    if (e == hipSuccess && !rc) e = hipEventElapsedTime(&ms, a, b);
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (rc) return rc;
    if (e != hipSuccess) return cfail(c, MCMC_HIP_ERR_DEVICE, "timing the all-reduce failed: %s", hipGetErrorString(e));
    *us_per_call = 1e3 * (double)ms / reps;
    return MCMC_HIP_OK;
}

}  // extern "C"
