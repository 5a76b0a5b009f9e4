// amdseg_allreduce_*: the gradient exchange of pure data parallelism for a host that is NOT PyTorch.
//
// The Python host of this repo exchanges gradients through torch.distributed (backend "nccl" = RCCL; spokennlp_amd/dp.py) because the
// process group, the launcher contract and accelerate / Trainer all live there.  A binder of include/amdseg.h without torch gets the same
// exchange from these five entry points: one explicit context (`amdseg_comm`: the RCCL communicator, a side stream, two events), buckets
// issued from the compute stream's point of view ("this slice is final now"), reduced in place over xGMI on the side stream, and one wait
// that makes the compute stream see every bucket.  That is exactly dp.GradBuckets' schedule (encoder layers last to first from inside
// backward, then the embedding tables, then the heads) -- the schedule stays the caller's.
//
// RCCL is bound at run time (dlopen, no link-time dependency): a process that already holds an RCCL (PyTorch ships one) shares it, a
// stand-alone binder gets /opt/rocm's.  Nothing here runs unless amdseg_allreduce_* is called.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/amdseg.h"

namespace {

typedef int nccl_result;                              // ncclResult_t (ncclSuccess == 0)
typedef struct ncclComm* nccl_comm;
struct nccl_uid { char internal[128]; };             // ncclUniqueId: NCCL_UNIQUE_ID_BYTES == AMDSEG_COMM_ID_BYTES
constexpr int kNcclSum = 0, kNcclFloat32 = 7, kNcclBfloat16 = 9;      // rccl.h: ncclRedOp_t / ncclDataType_t

struct RcclApi {
    void* handle = nullptr;
    nccl_result (*GetUniqueId)(nccl_uid*) = nullptr;
    nccl_result (*CommInitRank)(nccl_comm*, int, nccl_uid, int) = nullptr;
    nccl_result (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    nccl_result (*CommDestroy)(nccl_comm) = nullptr;
    const char* (*GetErrorString)(nccl_result) = nullptr;
    nccl_result (*CommCount)(const nccl_comm, int*) = nullptr;          // optional: the communicator's own rank count
};

// bound once per process; the only process-wide state of this file is the dlopen handle (a loaded library IS process-wide).  The binding is a
// function-local static initialised by bind(): C++11 makes that initialisation run exactly once and makes every concurrent first caller wait for
// it (two binder threads calling amdseg_allreduce_* at the same moment both see the finished table -- ADVICE r05); a partially bound library
// (a missing symbol) is closed again before the failure is reported.
RcclApi bind_rccl() {
    RcclApi api;
    const char* env = getenv("AMDSEG_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names)                      // an RCCL that is already in the process wins (one copy, one set of IPC handles)
        if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
    for (const char* n : names)
        if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) return api;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(h, "ncclCommCount"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) { dlclose(h); return RcclApi(); }
    api.handle = h;
    return api;
}
RcclApi* rccl() {
    static RcclApi api = bind_rccl();
    return api.handle ? &api : nullptr;
}

inline int comm_rc(nccl_result r) { return r == 0 ? AMDSEG_OK : AMDSEG_ERR_COMM_BASE + (int)r; }

}  // namespace

struct amdseg_comm {
    nccl_comm comm;
    hipStream_t side;           // every bucket is reduced here, never on the caller's compute stream
    hipEvent_t ready, done;
    int rank, world, device;
    size_t elements;            // elements issued since the last wait (bookkeeping for the caller: amdseg_allreduce_pending)
};

extern "C" {

int amdseg_allreduce_unique_id(void* id) {
    if (!id) return AMDSEG_ERR_ARG;
    RcclApi* r = rccl();
    if (!r) return AMDSEG_ERR_COMM_LIB;
    nccl_uid u;
    const nccl_result rc = r->GetUniqueId(&u);
    if (rc == 0) memcpy(id, u.internal, AMDSEG_COMM_ID_BYTES);
    return comm_rc(rc);
}

int amdseg_allreduce_init(amdseg_comm** out, const void* id, int rank, int world) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return AMDSEG_ERR_ARG;
    RcclApi* r = rccl();
    if (!r) return AMDSEG_ERR_COMM_LIB;
    amdseg_comm* c = static_cast<amdseg_comm*>(calloc(1, sizeof(amdseg_comm)));
    if (!c) return AMDSEG_ERR_ARG;
    hipError_t e = hipGetDevice(&c->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) { free(c); return (int)e; }
    nccl_uid u;
    memcpy(u.internal, id, AMDSEG_COMM_ID_BYTES);
    const nccl_result rc = r->CommInitRank(&c->comm, world, u, rank);          // collective: every rank of the world calls this
    if (rc != 0) {
        (void)hipEventDestroy(c->ready); (void)hipEventDestroy(c->done); (void)hipStreamDestroy(c->side);
        free(c);
        return comm_rc(rc);
    }
    c->rank = rank; c->world = world;
    *out = c;
    return AMDSEG_OK;
}

int amdseg_allreduce_bucket(amdseg_comm* c, void* buf, size_t n, int dtype, amdseg_stream_t compute_stream) {
    if (!c || (!buf && n) || (dtype != AMDSEG_F32 && dtype != AMDSEG_BF16)) return AMDSEG_ERR_ARG;
    if (!n) return AMDSEG_OK;
    RcclApi* r = rccl();
    if (!r) return AMDSEG_ERR_COMM_LIB;
    hipStream_t cs = static_cast<hipStream_t>(compute_stream);
    // the slice is final once everything queued on the compute stream so far has run: the side stream waits for exactly that point
    hipError_t e = hipEventRecord(c->ready, cs);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->side, c->ready, 0);
    if (e != hipSuccess) return (int)e;
    const nccl_result rc = r->AllReduce(buf, buf, n, dtype == AMDSEG_F32 ? kNcclFloat32 : kNcclBfloat16, kNcclSum, c->comm, c->side);
    if (rc == 0) c->elements += n;
    return comm_rc(rc);
}

int amdseg_allreduce_wait(amdseg_comm* c, amdseg_stream_t compute_stream) {
    if (!c) return AMDSEG_ERR_ARG;
    hipError_t e = hipEventRecord(c->done, c->side);
    if (e == hipSuccess) e = hipStreamWaitEvent(static_cast<hipStream_t>(compute_stream), c->done, 0);
    c->elements = 0;
    return e == hipSuccess ? AMDSEG_OK : (int)e;
}

int amdseg_allreduce_info(const amdseg_comm* c, int* rank, int* world, size_t* pending_elements) {
    if (!c) return AMDSEG_ERR_ARG;
    if (rank) *rank = c->rank;
    if (world) {                                            // what RCCL's communicator itself reports (ncclCommCount), not what init was told
        *world = c->world;
        RcclApi* r = rccl();
        int n = 0;
        if (r && r->CommCount && r->CommCount(c->comm, &n) == 0 && n > 0) *world = n;
    }
    if (pending_elements) *pending_elements = c->elements;
    return AMDSEG_OK;
}

int amdseg_allreduce_destroy(amdseg_comm* c) {
    if (!c) return AMDSEG_OK;
    RcclApi* r = rccl();
    (void)hipStreamSynchronize(c->side);
    nccl_result rc = r ? r->CommDestroy(c->comm) : 0;
    (void)hipEventDestroy(c->ready); (void)hipEventDestroy(c->done); (void)hipStreamDestroy(c->side);
    free(c);
    return comm_rc(rc);
}

}  // extern "C"

// (internal; amdseg_error_string routes the AMDSEG_ERR_COMM_* codes here)
__attribute__((visibility("hidden"))) const char* amdseg_comm_error_string_impl(int code) {
    if (code == AMDSEG_ERR_COMM_LIB) return "amdseg: librccl could not be loaded (set AMDSEG_RCCL_LIB)";
    RcclApi* r = rccl();
    if (r && r->GetErrorString && code > AMDSEG_ERR_COMM_BASE && code < AMDSEG_ERR_COMM_BASE + 100) return r->GetErrorString(code - AMDSEG_ERR_COMM_BASE);
    return nullptr;
}
