// The one-round reduce of DLSA through the C ABI: a single RCCL all-reduce (sum) of the rank's fp64 message
//   [ Sig_inv (p*p) | Sig_invMcoef (p) | coef (p) | ... ]     (reference: dlsa/dlsa.py:30-34, Spark groupby-sum + toPandas)
// for hosts that do not go through torch.distributed.  RCCL is resolved at run time (dlopen of the librccl the process
// already has, else the system one), so libdlsa_hip.so itself carries no link-time dependency on it and loads on a box
// without RCCL; the entry points then fail with DLSA_ERR_HIP and a message.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace dlsa {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = "symbols missing";      // dlerror() of the failed dlopen, captured once (a second dlerror() call returns NULL)
};

static RcclApi& rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api;
    tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {          // the copy already mapped into the process (torch's) wins
        api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (api.handle) break;
    }
    for (size_t i = 0; !api.handle && i < sizeof(names) / sizeof(names[0]); ++i) {
        api.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!api.handle) {
            const char* e = dlerror();
            if (e) snprintf(api.why, sizeof(api.why), "%s", e);
        }
    }
    if (!api.handle) return api;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.handle, "ncclAllReduce");
    api.CommCount = (decltype(api.CommCount))dlsym(api.handle, "ncclCommCount");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.GetErrorString;
    return api;
}

#define DLSA_RCCL_CHECK(expr)                                                                       \
    do {                                                                                            \
        ncclResult_t _r = (expr);                                                                   \
        if (_r != ncclSuccess) {                                                                    \
            dlsa::set_error("%s failed: %s", #expr, dlsa::rccl().GetErrorString(_r));               \
            return DLSA_ERR_HIP;                                                                    \
        }                                                                                           \
    } while (0)

static int need_rccl() {
    if (!rccl().ok) { set_error("RCCL (librccl.so) could not be loaded: %s", rccl().why); return DLSA_ERR_HIP; }
    return DLSA_OK;
}

}  // namespace dlsa

extern "C" {

int dlsa_comm_unique_id(char* id128) {
    using namespace dlsa;
    DLSA_REQUIRE(id128, "comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == DLSA_COMM_ID_BYTES, "ncclUniqueId size");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId id;
    DLSA_RCCL_CHECK(rccl().GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return DLSA_OK;
}

int dlsa_comm_init_rank(void** comm, int nranks, const char* id128, int rank) {
    using namespace dlsa;
    DLSA_REQUIRE(comm && id128 && nranks > 0 && rank >= 0 && rank < nranks, "comm_init_rank: bad argument");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    DLSA_RCCL_CHECK(rccl().CommInitRank(&c, nranks, id, rank));
    *comm = (void*)c;
    return DLSA_OK;
}

int dlsa_comm_destroy(void* comm) {
    using namespace dlsa;
    if (!comm) return DLSA_OK;
    if (int rc = need_rccl()) return rc;
    DLSA_RCCL_CHECK(rccl().CommDestroy((ncclComm_t)comm));
    return DLSA_OK;
}

int dlsa_allreduce_f64(void* rccl_comm, double* buf, int64_t count, void* stream) {
    using namespace dlsa;
    DLSA_REQUIRE(rccl_comm && buf && count > 0, "allreduce: bad argument");
    if (int rc = need_rccl()) return rc;
    DLSA_RCCL_CHECK(rccl().AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)rccl_comm, (hipStream_t)stream));
    return DLSA_OK;
}

}  // extern "C"
