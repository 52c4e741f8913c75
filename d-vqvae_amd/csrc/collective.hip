// RCCL all-gather of the generated MANO parameters behind the C ABI (`allgather_params(local[B/R,61]) -> [B,61]`).
// The reference has no counterpart (no distributed code at all): objects are independent, the batch shards contiguously
// over the ranks and this is the one exchange of the path.  One process per GPU; the communicator is created from a unique id
// that rank 0 makes (dvq_comm_unique_id) and the caller hands to every rank (the host mirror sends it through torch.distributed's
// store); the collective is enqueued on the caller's stream.  RCCL is resolved at the first call with dlopen -- the process
// usually has PyTorch's copy loaded already and must not get a second one -- so the library links and loads without it.
#include "dvq_internal.h"
#include <dlfcn.h>
#include <string.h>

// The six RCCL entry points used here, declared locally (rccl.h's own declarations, ABI-stable since NCCL 2.0): the library
// builds on a machine without the RCCL headers and, resolving the symbols by dlopen, loads on one without the library.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat32 = 7 } ncclDataType_t;
}

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

const Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        // the copy already in the process first (PyTorch's wheel bundles one); then the system one
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"}) {
            x.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (x.handle) break;
        }
        if (!x.handle)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                x.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (x.handle) break;
            }
        if (!x.handle) return x;
        x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.handle, "ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.handle, "ncclCommInitRank");
        x.AllGather = (decltype(x.AllGather))dlsym(x.handle, "ncclAllGather");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.handle, "ncclCommDestroy");
        x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.handle, "ncclGetErrorString");
        x.CommCount = (decltype(x.CommCount))dlsym(x.handle, "ncclCommCount");
        x.ok = x.GetUniqueId && x.CommInitRank && x.AllGather && x.CommDestroy && x.GetErrorString && x.CommCount;
        return x;
    }();
    return r;
}

int fail(const char* what, ncclResult_t e) {
    dvq_set_error("%s: RCCL error: %s", what, rccl().GetErrorString ? rccl().GetErrorString(e) : "?");
    return DVQ_ELAUNCH;
}

}  // namespace

extern "C" int dvq_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int dvq_comm_unique_id(void* id_out, size_t id_bytes) {
    DVQ_REQUIRE(id_out && id_bytes >= sizeof(ncclUniqueId), "comm_unique_id: need a %zu-byte buffer", sizeof(ncclUniqueId));
    if (!rccl().ok) { dvq_set_error("comm_unique_id: librccl not found"); return DVQ_ENODEVICE; }
    const ncclResult_t e = rccl().GetUniqueId(reinterpret_cast<ncclUniqueId*>(id_out));
    return e == ncclSuccess ? DVQ_OK : fail("comm_unique_id", e);
}

extern "C" int dvq_comm_init(const void* id, size_t id_bytes, int world, int rank, void** comm_out) {
    DVQ_REQUIRE(id && id_bytes >= sizeof(ncclUniqueId) && comm_out && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments");
    if (!rccl().ok) { dvq_set_error("comm_init: librccl not found"); return DVQ_ENODEVICE; }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    const ncclResult_t e = rccl().CommInitRank(&comm, world, uid, rank);       // on the calling thread's current device
    if (e != ncclSuccess) return fail("comm_init", e);
    *comm_out = comm;
    return DVQ_OK;
}

extern "C" int dvq_allgather_params(void* comm, const float* local, int64_t rows_per_rank, int cols, float* out, dvq_stream_t stream) {
    DVQ_REQUIRE(comm && rows_per_rank >= 0 && cols > 0, "allgather_params: bad arguments");
    if (rows_per_rank == 0) return DVQ_OK;
    DVQ_REQUIRE(local && out, "allgather_params: null pointer");
    if (!rccl().ok) { dvq_set_error("allgather_params: librccl not found"); return DVQ_ENODEVICE; }
    const ncclResult_t e = rccl().AllGather(local, out, (size_t)rows_per_rank * cols, ncclFloat32, (ncclComm_t)comm, (hipStream_t)stream);
    return e == ncclSuccess ? DVQ_OK : fail("allgather_params", e);
}

extern "C" int dvq_comm_count(void* comm, int* ranks_out) {
    DVQ_REQUIRE(comm && ranks_out, "comm_count: null pointer");
    if (!rccl().ok) { dvq_set_error("comm_count: librccl not found"); return DVQ_ENODEVICE; }
    const ncclResult_t e = rccl().CommCount((ncclComm_t)comm, ranks_out);      // what RCCL itself says the communicator spans
    return e == ncclSuccess ? DVQ_OK : fail("comm_count", e);
}

extern "C" int dvq_comm_destroy(void* comm) {
    if (!comm) return DVQ_OK;
    if (!rccl().ok) { dvq_set_error("comm_destroy: librccl not found"); return DVQ_ENODEVICE; }
    const ncclResult_t e = rccl().CommDestroy((ncclComm_t)comm);
    return e == ncclSuccess ? DVQ_OK : fail("comm_destroy", e);
}
