// The library's own RCCL communicator: the all-reduces of the data-parallel step (SyncBatchNorm statistics, gradient buckets) enqueued from C
// on the stream of the kernels around them -- no process-group layer, no second stream, no host round trip (reference: Lightning's DDP
// strategy + sync_batchnorm, train.py:131-133,247; torch.distributed keeps the rendezvous, the parameter broadcast and the barriers).
// RCCL is taken from the process (torch has loaded its librccl.so.1 already) through dlopen / dlsym: this library has no link-time
// dependency on it, and a build without RCCL still loads -- leod_comm_* then report LEOD_ERR_UNSUPPORTED.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <string.h>

#include <mutex>
#include <string>

#include "common.hpp"

namespace {

typedef void* comm_t;
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueId_t)(UniqueId*);
typedef int (*CommInitRank_t)(comm_t*, int, UniqueId, int);
typedef int (*AllReduce_t)(const void*, void*, size_t, int, int, comm_t, hipStream_t);
typedef int (*CommDestroy_t)(comm_t);
typedef const char* (*GetErrorString_t)(int);

struct Rccl {
    void* handle = nullptr;
    GetUniqueId_t get_unique_id = nullptr;
    CommInitRank_t comm_init_rank = nullptr;
    AllReduce_t all_reduce = nullptr;
    CommDestroy_t comm_destroy = nullptr;
    GetErrorString_t error_string = nullptr;
    bool ok = false;
};

std::mutex g_mu;
Rccl g_rccl;
comm_t g_comm = nullptr;
int g_world = 0, g_rank = -1;
std::string g_err;

bool load_rccl() {
    if (g_rccl.ok) return true;
    if (g_rccl.handle == nullptr) {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (int pass = 0; pass < 2 && !g_rccl.handle; ++pass)        // first the copy the process already holds (torch's), then a fresh load
            for (const char* n : names) {
                g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (g_rccl.handle) break;
            }
        if (!g_rccl.handle) { g_err = "librccl.so.1 not found"; return false; }
    }
    g_rccl.get_unique_id = reinterpret_cast<GetUniqueId_t>(dlsym(g_rccl.handle, "ncclGetUniqueId"));
    g_rccl.comm_init_rank = reinterpret_cast<CommInitRank_t>(dlsym(g_rccl.handle, "ncclCommInitRank"));
    g_rccl.all_reduce = reinterpret_cast<AllReduce_t>(dlsym(g_rccl.handle, "ncclAllReduce"));
    g_rccl.comm_destroy = reinterpret_cast<CommDestroy_t>(dlsym(g_rccl.handle, "ncclCommDestroy"));
    g_rccl.error_string = reinterpret_cast<GetErrorString_t>(dlsym(g_rccl.handle, "ncclGetErrorString"));
    g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.all_reduce && g_rccl.comm_destroy;
    if (!g_rccl.ok) g_err = "RCCL symbols missing";
    return g_rccl.ok;
}

int fail(const char* what, int rc) {
    g_err = std::string(what) + ": " + (g_rccl.error_string ? g_rccl.error_string(rc) : "error") + " (" + std::to_string(rc) + ")";
    return LEOD_ERR_LAUNCH;
}

}  // namespace

// stands in for an all-reduce on a stream that is being captured: k_plan.hip turns the node into a collective op of the launch plan
__global__ void leod_comm_marker_kernel(void* buf, long count, int dtype) { (void)buf; (void)count; (void)dtype; }

LEOD_API const char* leod_comm_last_error() { return g_err.c_str(); }

// id128 <- a fresh RCCL unique id (rank 0 calls this and hands the 128 bytes to the other ranks)
LEOD_API int leod_comm_unique_id(char* id128) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!id128) return LEOD_ERR_ARG;
    if (!load_rccl()) return LEOD_ERR_UNSUPPORTED;
    UniqueId id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc != 0) return fail("ncclGetUniqueId", rc);
    memcpy(id128, id.internal, 128);
    return LEOD_OK;
}

// collective over all ranks: the communicator of this process on the CURRENT device
LEOD_API int leod_comm_init(const char* id128, int rank, int world) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!id128 || world < 1 || rank < 0 || rank >= world) return LEOD_ERR_ARG;
    if (g_comm) { g_err = "communicator already initialised"; return LEOD_ERR_ARG; }
    if (!load_rccl()) return LEOD_ERR_UNSUPPORTED;
    UniqueId id;
    memcpy(id.internal, id128, 128);
    comm_t c = nullptr;
    const int rc = g_rccl.comm_init_rank(&c, world, id, rank);
    if (rc != 0 || !c) return fail("ncclCommInitRank", rc);
    g_comm = c; g_world = world; g_rank = rank;
    return LEOD_OK;
}

LEOD_API int leod_comm_world() { return g_comm ? g_world : 0; }

// buf[count] <- sum over ranks, in place, ordered on `stream` like a kernel.  dtype: 0 float, 1 double, 2 bf16
LEOD_API int leod_comm_allreduce(void* buf, long count, int dtype, hipStream_t stream) {
    if (!g_comm) { g_err = "communicator not initialised"; return LEOD_ERR_UNSUPPORTED; }
    if (!buf || count < 0 || dtype < 0 || dtype > 2) return LEOD_ERR_ARG;
    if (count == 0) return LEOD_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) {
        hipLaunchKernelGGL(leod_comm_marker_kernel, dim3(1), dim3(1), 0, stream, buf, count, dtype);
        return leod_launch_status();
    }
    (void)hipGetLastError();
    static const int kType[3] = {7, 8, 9};                       // ncclFloat32, ncclFloat64, ncclBfloat16 (rccl.h)
    const int rc = g_rccl.all_reduce(buf, buf, (size_t)count, kType[dtype], 0 /* ncclSum */, g_comm, stream);
    if (rc != 0) { std::lock_guard<std::mutex> lk(g_mu); return fail("ncclAllReduce", rc); }
    return LEOD_OK;
}

LEOD_API int leod_comm_destroy() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm) return LEOD_OK;
    const int rc = g_rccl.comm_destroy(g_comm);
    g_comm = nullptr; g_world = 0; g_rank = -1;
    return rc == 0 ? LEOD_OK : fail("ncclCommDestroy", rc);
}
