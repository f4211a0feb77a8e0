// (e) multi-GPU: the per-layer all-gather of the propagated row blocks, on RCCL over xGMI.
//
// The reference has no collective; this is the exchange step the row partition adds.  RCCL is bound at
// run time (dlopen), not at link time: a single-GPU process never loads it, and inside a PyTorch-ROCm process the copy
// torch already loaded (torch/lib/librccl.so, SONAME librccl.so.1, linked against torch's HIP runtime) is the one used --
// a second RCCL/HIP pair from /opt/rocm would not see torch's device allocations.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "tgcn_internal.h"

namespace tgcn {
namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    Rccl &r = g_rccl;
    // already in the process (PyTorch-ROCm)?  then that copy; else torch's by file name via this library's RUNPATH
    // (build.py puts torch/lib first), else the system one
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (r.handle)
            break;
    }
    if (!r.handle) {
        for (const char *name : {"librccl.so", "librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle)
                break;
        }
    }
    if (!r.handle)
        return;
    r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(r.handle, "ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.all_gather = reinterpret_cast<decltype(r.all_gather)>(dlsym(r.handle, "ncclAllGather"));
    r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(r.handle, "ncclGetErrorString"));
    r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_gather && r.error_string;
}

const Rccl *rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.ok) {
        set_error("RCCL is not available: %s", g_rccl.handle ? "librccl lacks an expected symbol" : "librccl.so could not be loaded");
        return nullptr;
    }
    return &g_rccl;
}

int check_nccl(const Rccl *r, ncclResult_t res, const char *what)
{
    if (res == ncclSuccess)
        return TGCN_OK;
    set_error("%s: %s", what, r->error_string(res));
    return TGCN_ERR_HIP;
}

}  // namespace
}  // namespace tgcn

using namespace tgcn;

static_assert(TGCN_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "tgcn.h and rccl.h disagree on the id size");

extern "C" int tgcn_comm_unique_id(void *id_host)
{
    TGCN_REQUIRE(id_host, "id_host is NULL");
    const Rccl *r = rccl();
    if (!r)
        return TGCN_ERR_UNSUPPORTED;
    ncclUniqueId id;
    const int rc = check_nccl(r, r->get_unique_id(&id), "ncclGetUniqueId");
    if (rc == TGCN_OK)
        memcpy(id_host, &id, sizeof(id));
    return rc;
}

extern "C" int tgcn_comm_init_rank(tgcn_comm_t *comm, int32_t world, int32_t rank, const void *id_host)
{
    TGCN_REQUIRE(comm && id_host, "comm / id_host is NULL");
    TGCN_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank must be in [0, world)");
    const Rccl *r = rccl();
    if (!r)
        return TGCN_ERR_UNSUPPORTED;
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof(id));
    ncclComm_t c = nullptr;
    const int rc = check_nccl(r, r->comm_init_rank(&c, world, id, rank), "ncclCommInitRank");
    *comm = rc == TGCN_OK ? static_cast<tgcn_comm_t>(c) : nullptr;
    return rc;
}

extern "C" int tgcn_comm_destroy(tgcn_comm_t comm)
{
    if (!comm)
        return TGCN_OK;
    const Rccl *r = rccl();
    if (!r)
        return TGCN_ERR_UNSUPPORTED;
    return check_nccl(r, r->comm_destroy(static_cast<ncclComm_t>(comm)), "ncclCommDestroy");
}

extern "C" int tgcn_allgather_rows(tgcn_comm_t comm, const float *local, float *full, int64_t rows_local, int32_t d,
                                   tgcn_stream_t stream)
{
    TGCN_REQUIRE(comm, "comm is NULL");
    TGCN_REQUIRE(rows_local >= 0 && d > 0, "rows_local / d out of range");
    if (rows_local == 0)
        return TGCN_OK;
    TGCN_REQUIRE(local && full, "local / full is NULL");
    const Rccl *r = rccl();
    if (!r)
        return TGCN_ERR_UNSUPPORTED;
    // fp32 on the wire: anything narrower would break the 1e-4 bar and the 1-vs-P bit identity.
    // `local` may be the rank's own block inside `full` (RCCL's in-place form): no staging copy either way.
    return check_nccl(r, r->all_gather(local, full, (size_t)rows_local * (size_t)d, ncclFloat32, static_cast<ncclComm_t>(comm),
                                       static_cast<hipStream_t>(stream)),
                      "ncclAllGather");
}
