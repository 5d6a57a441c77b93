// In-library gradient exchange: ftmi_allreduce_{unique_id, init, bucket, wait, destroy} -- SURVEY section 8(b)'s minimum symbol set, replacing what
// `replicate(model, bucket_cap_mb=100)` does for the LoRA gradients in the reference (finetrainers/parallel/ptd.py:462-463): bucketed all-reduce (mean) of
// slices of the flat fp32 gradient buffer, issued as soon as a block range of the backward is final, on the library's own communication stream, overlapped with
// the rest of the backward; the compute stream joins before clip + AdamW.
//
// The collectives are RCCL's (one process per GPU; xGMI between the GPUs of a node).  librccl is NOT a link-time dependency of libftmi355.so: it is looked up
// with dlopen at the first ftmi_allreduce_* call (the soname torch already loaded wins), so the library loads -- and every other entry point works -- on a
// machine without RCCL, and these entry points fail loudly there.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "kernels.h"

namespace ftmi {
namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2.10: ncclAvg = 4, ncclFloat32 = 7)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { kNcclFloat32 = 7, kNcclSum = 0, kNcclAvg = 4 };
struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r = [] {
        Rccl q;
        void* h = nullptr;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return q;
        q.GetUniqueId = (decltype(q.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        q.CommInitRank = (decltype(q.CommInitRank))dlsym(h, "ncclCommInitRank");
        q.AllReduce = (decltype(q.AllReduce))dlsym(h, "ncclAllReduce");
        q.CommDestroy = (decltype(q.CommDestroy))dlsym(h, "ncclCommDestroy");
        q.GetErrorString = (decltype(q.GetErrorString))dlsym(h, "ncclGetErrorString");
        q.GetVersion = (decltype(q.GetVersion))dlsym(h, "ncclGetVersion");
        q.ok = q.GetUniqueId && q.CommInitRank && q.AllReduce && q.CommDestroy;
        return q;
    }();
    return r;
}

struct Exchange {
    ncclComm_t comm = nullptr;
    hipStream_t comm_st = nullptr;     // the library's communication stream: collectives never queue behind compute kernels
    hipEvent_t ready = nullptr;        // compute -> comm: "the bucket's gradients are final"
    hipEvent_t done = nullptr;         // comm -> compute: "every bucket issued so far has been reduced"
    int rank = 0, world = 1;
    long buckets = 0;
    bool pending = false;
    std::mutex mu;
};

int rccl_fail(const char* what, ncclResult_t rc) {
    char msg[256];
    snprintf(msg, sizeof msg, "%s: RCCL error %d (%s)", what, rc, rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
    return set_error(FTMI_ERR_LAUNCH, msg);
}

}  // namespace
}  // namespace ftmi

using namespace ftmi;

extern "C" {

int ftmi_allreduce_unique_id(void* id128) {
    if (!id128) return set_error(FTMI_ERR_INVALID, "ftmi_allreduce_unique_id: null buffer");
    if (!rccl().ok) return set_error(FTMI_ERR_UNSUPPORTED, "ftmi_allreduce: librccl.so not found (dlopen)");
    ncclUniqueId id;
    const ncclResult_t rc = rccl().GetUniqueId(&id);
    if (rc) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id128, &id, sizeof id);
    return 0;
}

int ftmi_allreduce_init(const void* id128, int rank, int world, ftmi_exchange* out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return set_error(FTMI_ERR_INVALID, "ftmi_allreduce_init: bad arguments");
    if (!rccl().ok) return set_error(FTMI_ERR_UNSUPPORTED, "ftmi_allreduce: librccl.so not found (dlopen)");
    Exchange* x = new Exchange();
    x->rank = rank;
    x->world = world;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    const ncclResult_t rc = rccl().CommInitRank(&x->comm, world, id, rank);  // collective: every rank of the job calls it with the same id
    if (rc) { delete x; return rccl_fail("ncclCommInitRank", rc); }
    if (hipStreamCreateWithFlags(&x->comm_st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&x->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x->done, hipEventDisableTiming) != hipSuccess) {
        rccl().CommDestroy(x->comm);
        delete x;
        return set_error(FTMI_ERR_LAUNCH, "ftmi_allreduce_init: stream / event creation failed");
    }
    *out = x;
    return 0;
}

int ftmi_allreduce_bucket(ftmi_exchange ex, float* grad, size_t count, int average, ftmi_stream compute_stream) {
    Exchange* x = (Exchange*)ex;
    if (!x || !grad) return set_error(FTMI_ERR_INVALID, "ftmi_allreduce_bucket: null argument");
    if (count == 0) return 0;
    std::lock_guard<std::mutex> lk(x->mu);
    // the gradients of this bucket are final in compute-stream order: the communication stream waits for exactly that point, then reduces in place
    if (hipEventRecord(x->ready, (hipStream_t)compute_stream) != hipSuccess || hipStreamWaitEvent(x->comm_st, x->ready, 0) != hipSuccess)
        return set_error(FTMI_ERR_LAUNCH, "ftmi_allreduce_bucket: event hand-over failed");
    const ncclResult_t rc = rccl().AllReduce(grad, grad, count, kNcclFloat32, average ? kNcclAvg : kNcclSum, x->comm, x->comm_st);
    if (rc) return rccl_fail("ncclAllReduce", rc);
    x->buckets += 1;
    x->pending = true;
    return 0;
}

int ftmi_allreduce_wait(ftmi_exchange ex, ftmi_stream compute_stream) {
    Exchange* x = (Exchange*)ex;
    if (!x) return set_error(FTMI_ERR_INVALID, "ftmi_allreduce_wait: null exchange");
    std::lock_guard<std::mutex> lk(x->mu);
    if (!x->pending) return 0;
    // device-side join: no host synchronisation -- the compute stream continues (clip + AdamW) once every bucket issued so far has been reduced
    if (hipEventRecord(x->done, x->comm_st) != hipSuccess || hipStreamWaitEvent((hipStream_t)compute_stream, x->done, 0) != hipSuccess)
        return set_error(FTMI_ERR_LAUNCH, "ftmi_allreduce_wait: event hand-over failed");
    x->pending = false;
    return 0;
}

long ftmi_allreduce_buckets_issued(ftmi_exchange ex) { return ex ? ((Exchange*)ex)->buckets : -1; }

int ftmi_allreduce_version(void) {
    int v = 0;
    if (!rccl().ok || !rccl().GetVersion || rccl().GetVersion(&v)) return -1;
    return v;
}

int ftmi_allreduce_destroy(ftmi_exchange ex) {
    Exchange* x = (Exchange*)ex;
    if (!x) return 0;
    hipStreamSynchronize(x->comm_st);
    if (x->comm) rccl().CommDestroy(x->comm);
    hipEventDestroy(x->ready);
    hipEventDestroy(x->done);
    hipStreamDestroy(x->comm_st);
    delete x;
    return 0;
}

}  // extern "C"
