// Library-side RCCL: the collectives of the sharded paths (icp_rigid_run_dist, nicp_solve_dist) issued by the library
// itself on the context's stream, instead of through a host callback per collective ("RCCL over
// xGMI all-reduces the covariance / CG partial sums").  librccl is opened at run time, on the first icp_rccl_* call:
// libicp_hip.so has no link-time dependency on it, single-GPU hosts never load it, and inside a PyTorch process the
// loader hands back the copy PyTorch already mapped.  A communicator is the library's own (ncclCommInitRank from a
// unique id the host distributes), separate from whatever the host language uses.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.h"

struct icp_rccl {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

namespace icp {
namespace {
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

RcclApi &api() {
    static RcclApi a = [] {
        RcclApi r;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) return r;
#define ICP_SYM(field, sym) r.field = (decltype(r.field))dlsym(r.handle, sym)
        ICP_SYM(GetUniqueId, "ncclGetUniqueId");
        ICP_SYM(CommInitRank, "ncclCommInitRank");
        ICP_SYM(CommDestroy, "ncclCommDestroy");
        ICP_SYM(AllReduce, "ncclAllReduce");
        ICP_SYM(Send, "ncclSend");
        ICP_SYM(Recv, "ncclRecv");
        ICP_SYM(GroupStart, "ncclGroupStart");
        ICP_SYM(GroupEnd, "ncclGroupEnd");
        ICP_SYM(GetErrorString, "ncclGetErrorString");
#undef ICP_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
        return r;
    }();
    return a;
}

int need_api() {
    if (api().ok) return ICP_OK;
    set_error("librccl could not be opened (%s): the library-side collectives need RCCL; pass callbacks instead",
              api().handle ? "symbols missing" : dlerror());
    return ICP_ERR_UNSUPPORTED;
}

int check(ncclResult_t r, const char *what) {
    if (r == ncclSuccess) return ICP_OK;
    set_error("%s failed: %s", what, api().GetErrorString ? api().GetErrorString(r) : "RCCL error");
    return ICP_ERR_COMM;
}
}  // namespace

// buf[0..count) summed over the ranks in place, ordered on `stream` like a kernel launch (no host synchronisation)
int rccl_allreduce_sum(icp_rccl *rc, double *buf, int count, hipStream_t stream) {
    return check(api().AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, rc->comm, stream), "ncclAllReduce");
}

// Halo exchange: send_buf holds, grouped by destination rank, send_counts[q] blocks of `width` doubles for rank q;
// recv_buf receives recv_counts[q] blocks from rank q, grouped by source rank.  One grouped send/recv, on `stream`.
int rccl_exchange(icp_rccl *rc, const double *send_buf, const int64_t *send_counts, double *recv_buf, const int64_t *recv_counts,
                  int width, hipStream_t stream) {
    const RcclApi &a = api();
    if (int e = check(a.GroupStart(), "ncclGroupStart")) return e;
    int64_t so = 0, ro = 0;
    int err = ICP_OK;
    for (int q = 0; q < rc->world && !err; ++q) {
        const int64_t ns = send_counts[q] * width, nr = recv_counts[q] * width;
        if (ns > 0 && q != rc->rank) err = check(a.Send(send_buf + so, (size_t)ns, ncclDouble, q, rc->comm, stream), "ncclSend");
        if (nr > 0 && q != rc->rank && !err) err = check(a.Recv(recv_buf + ro, (size_t)nr, ncclDouble, q, rc->comm, stream), "ncclRecv");
        so += ns;
        ro += nr;
    }
    const int e2 = check(a.GroupEnd(), "ncclGroupEnd");
    return err ? err : e2;
}
}  // namespace icp

extern "C" int icp_rccl_unique_id(void *id_out) {
    ICP_REQUIRE(id_out != nullptr, "null pointer");
    if (int rc = icp::need_api()) return rc;
    static_assert(sizeof(ncclUniqueId) == ICP_RCCL_ID_BYTES, "ICP_RCCL_ID_BYTES must be RCCL's unique-id size");
    ncclUniqueId id;
    if (int rc = icp::check(icp::api().GetUniqueId(&id), "ncclGetUniqueId")) return rc;
    std::memcpy(id_out, &id, sizeof(id));
    return ICP_OK;
}

extern "C" int icp_rccl_create(icp_ctx *ctx, const void *id_bytes, int rank, int world, icp_rccl **out) {
    ICP_REQUIRE(ctx && id_bytes && out, "null pointer");
    ICP_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank / world");
    if (int rc = icp::need_api()) return rc;
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    icp_rccl *r = new icp_rccl();
    r->rank = rank;
    r->world = world;
    r->device = ctx->device;
    if (int rc = icp::check(icp::api().CommInitRank(&r->comm, world, id, rank), "ncclCommInitRank")) {
        delete r;
        return rc;
    }
    *out = r;
    return ICP_OK;
}

extern "C" int icp_rccl_destroy(icp_rccl *r) {
    if (!r) return ICP_OK;
    int rc = ICP_OK;
    if (r->comm && icp::api().ok) {
        (void)hipSetDevice(r->device);
        rc = icp::check(icp::api().CommDestroy(r->comm), "ncclCommDestroy");
    }
    delete r;
    return rc;
}
