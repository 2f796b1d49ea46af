// comm.hip - the one data-path collective of the ray-sharded step: an in-place sum (or mean) of the flat fp32 gradient buffer over
// RCCL / xGMI, behind the C ABI (nvfi_allreduce_grads).  RCCL is loaded with dlopen at first use, so libnvfi_hip.so has
// no link-time dependency on it (single-GPU hosts and the CPU-side symbol checks never touch it).
//
// Bootstrap is the caller's: rank 0 asks nvfi_comm_unique_id for the 128-byte id, ships it to the other ranks by whatever means the
// host program has (torch.distributed's store in nvfi_amd/dist.py; MPI, a file, a socket elsewhere), and every rank calls
// nvfi_comm_init.  One process per GPU; the communicator is bound to the device that is current at init time.
#include <dlfcn.h>
#include <string.h>
#include "common.h"

// The handful of NCCL / RCCL ABI items this file needs, declared here instead of #include <rccl/rccl.h>: the library is found with
// dlopen at run time, so a host without the RCCL development headers can still build libnvfi_hip.so (values as in nccl.h since 2.10:
// the enums are part of the stable ABI).
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclFloat32 = 7 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclAvg = 4 } ncclRedOp_t;
}

struct nvfi_comm { ncclComm_t comm; int world, rank; int no_avg; };

__global__ void k_scale_inplace(float* p, int64_t n, float s) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] *= s;
}

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
int load_rccl() {
    if (g_rccl.h) return 0;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return nvfi_fail(6, "RCCL not found (dlopen librccl.so): %s", dlerror());
#define SYM(field, name) *(void**)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) return nvfi_fail(6, "librccl.so lacks %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(AllReduce, "ncclAllReduce");
    SYM(CommDestroy, "ncclCommDestroy"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.h = h;
    return 0;
}
#define RCCLCK(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) return nvfi_fail(200 + (int)_r, "%s failed: %s", #expr, g_rccl.GetErrorString(_r)); } while (0)
}  // namespace

extern "C" int nvfi_comm_unique_id(void* id128) {
    if (load_rccl()) return 6;
    static_assert(sizeof(ncclUniqueId) == NVFI_UNIQUE_ID_BYTES, "RCCL unique id size");
    RCCLCK(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128)));
    return 0;
}
extern "C" int nvfi_comm_init(nvfi_comm** out, int world, int rank, const void* id128) {
    if (!out || world < 1 || rank < 0 || rank >= world) return nvfi_fail(2, "nvfi_comm_init: bad arguments (world=%d rank=%d)", world, rank);
    if (load_rccl()) return 6;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    nvfi_comm* c = new nvfi_comm{nullptr, world, rank, 0};
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { delete c; return nvfi_fail(200 + (int)r, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); }
    *out = c;
    return 0;
}
// in place: flat[0..count) <- sum over ranks (average != 0: mean).  Asynchronous on `stream`; one collective per call.
extern "C" int nvfi_allreduce_grads(nvfi_comm* c, float* flat, int64_t count, int average, void* stream) {
    if (!c) return nvfi_fail(2, "nvfi_allreduce_grads: NULL communicator");
    if (count <= 0) return 0;
    if (average && !c->no_avg) {
        const ncclResult_t r = g_rccl.AllReduce(flat, flat, (size_t)count, ncclFloat32, ncclAvg, c->comm, (hipStream_t)stream);
        if (r == ncclSuccess) return 0;
        if (r != ncclInvalidArgument) return nvfi_fail(200 + (int)r, "ncclAllReduce(avg) failed: %s", g_rccl.GetErrorString(r));
        c->no_avg = 1;      // an RCCL older than 2.10 has no ncclAvg: sum, then scale
    }
    RCCLCK(g_rccl.AllReduce(flat, flat, (size_t)count, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream));
    if (average) {
        hipLaunchKernelGGL(k_scale_inplace, dim3((unsigned)((count / 4 + 255) / 256 + 1)), dim3(256), 0, (hipStream_t)stream, flat, count, 1.f / (float)c->world);
        LAUNCHCK();
    }
    return 0;
}
extern "C" int nvfi_comm_destroy(nvfi_comm* c) {
    if (!c) return 0;
    if (g_rccl.h && c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return 0;
}
