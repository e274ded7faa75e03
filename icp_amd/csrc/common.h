// Internal helpers shared by the .hip translation units of libicp_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/icp_hip.h"

namespace icp {

void set_error(const char *fmt, ...);

#define ICP_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            icp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,  \
                           __LINE__);                                                        \
            return ICP_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

#define ICP_REQUIRE(cond, msg)                              \
    do {                                                    \
        if (!(cond)) {                                      \
            icp::set_error("%s: %s", __func__, msg);        \
            return ICP_ERR_ARG;                             \
        }                                                   \
    } while (0)

// Grow-only scratch buffer owned by the context.
struct Scratch {
    void *ptr = nullptr;
    size_t cap = 0;
};

constexpr size_t kSmallBytes = 8192;  // icp_ctx::small: RigidState + reduce buffer | find_transform scratch | the builds' statistics records
constexpr int kMaxPartials = 2048;   // upper bound on blocks of any partial-sum reduction
constexpr int kSumSlots = 16;        // doubles per block partial / per reduced vector

}  // namespace icp

struct icp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cu = 256;
    icp::Scratch split_d2, split_idx;   // closest: per-split partial results
    icp::Scratch partials;              // reductions: [kMaxPartials][kSumSlots] doubles
    icp::Scratch small;                 // small fp64 state (sums, means, R, t, flags)
    icp::Scratch work_a, work_b, work_c, work_d;  // general work arrays
    icp::Scratch grid_a, grid_b, grid_c, grid_d, grid_e;  // spatial grid over the target
    icp::Scratch sort_a, sort_b, sort_c, sort_d;          // Morton order of the queries
    // temporary storage of the stable code sort (stablesort.hip): one for a sorted source, one for a target above 4 M
    // points -- icp_rigid_run builds the two orders on two streams at the same time, they must not share it
    icp::Scratch sort_tmp, sort_tmp_tgt;
    icp::Scratch defer;                                   // grid search: queries handed to the wave-per-query pass
    icp::Scratch err_part;                                // fused rigid loop: per-block error sums of a search
    icp::Scratch coh;                                     // coherent rigid loop: per-query bound + reference position
    void *defer_cleared = nullptr;                        // buffer whose two counters are known to be in step
    int defer_parity = 0;                                 // which of the two counters the next search uses
    // nicp_solve replays its CG iterations as a hipGraph; graphs cannot be captured on the legacy
    // default stream (which `stream` may be), so they are captured and launched on this one
    hipStream_t side = nullptr;
    hipEvent_t side_ev = nullptr;
    hipEvent_t order_ev = nullptr;  // icp_ctx_set_stream: orders a new stream after the work queued on the previous one
    // this device's address of kabsch.hip's file-scope scratch array (icp_find_transform_noalloc)
    double *noalloc_scratch = nullptr;
    void *pinned = nullptr;  // 4 KB of pinned host memory: asynchronous read-back of the grid statistics (grid.hip)
    hipEvent_t stat_ev[2] = {nullptr, nullptr};  // "the statistics record of build slot 0 / 1 has reached the host"
};

namespace icp {
int ensure(icp_ctx *ctx, Scratch &s, size_t bytes);
int noalloc_init(icp_ctx *ctx);  // kabsch.hip
int stable_sort_codes(icp_ctx *ctx, Scratch &tmp, const uint32_t *codes_in, uint32_t *codes_out, int32_t *idx_tmp, int32_t *idx_out,
                      int64_t m, int code_bits);  // stablesort.hip

// spatial grid over a target cloud (grid.hip)
struct GridParams {
    double mn[3];
    double mx[3];
    double h;      // finest cell side
    double inv_h;
    int bits;      // cells per axis = 1 << bits
    int dim;
    int open;      // bit 2a / 2a+1: the first / last column of axis a is unbounded outwards (robust extents, grid.hip)
};
struct GridBuild {
    GridParams gp;
    double *spts = nullptr;     // targets in Morton order
    int32_t *sidx = nullptr;    // their original indices
    void *recs = nullptr;       // the same as 32-byte records {x, y, z, index} (double4)
    uint32_t *start = nullptr;  // [ncell+1] first sorted position of every finest cell
    uint32_t *inv = nullptr;    // [m] sorted position of every original index (for ICP_HINT_IDX)
    int64_t m = 0;
    bool usable = false;        // false: non-finite targets -> use the brute-force scan
};
// slot 0: target grid (grid_* scratch); slot 1: query ordering (sort_* scratch)
int grid_build(icp_ctx *ctx, const double *tgt, int64_t m, int dim, GridBuild &gb, int slot = 0);
// the two halves of a build (grid.hip): statistics + asynchronous read-back | everything after the read-back
struct GridPending {
    const double *pts = nullptr;
    int64_t m = 0;
    int dim = 0, ns = 0;
    bool stable = false, robust = false, adaptive = false;
    void *host = nullptr;   // pinned copy of the statistics record
    hipEvent_t ev = nullptr;  // recorded behind that copy: the second half waits for it, not for the whole stream
    Scratch *sa = nullptr, *sb = nullptr, *sc = nullptr, *sd = nullptr;
};
int grid_build_begin(icp_ctx *ctx, const double *tgt, int64_t m, int dim, GridBuild &gb, Scratch &sa, Scratch &sb,
                     Scratch &sc, Scratch &sd, bool stable, GridPending &pend);
int grid_build_finish(icp_ctx *ctx, const GridPending &pend, GridBuild &gb);
// same, into caller-owned scratch (persistent target handles)
int grid_build_into(icp_ctx *ctx, const double *tgt, int64_t m, int dim, GridBuild &gb, Scratch &sa, Scratch &sb,
                    Scratch &sc, Scratch &sd, bool stable = false);
// Fused rigid loop: the queries are src . R + t (R, t: device memory, read by the search kernels themselves) and the
// search also sums the distances to the hints = the previous iteration's error (grid.hip, grid_query_kernel<FUSED>).
// Out: where the per-block error sums are and how many there are (context scratch, valid until the next search).
struct FusedApply {
    const double *R, *t;
    double **err_part;
    int *n_err;
};
// Coherent rigid loop (grid.hip, grid_query_kernel<BND>): per query, a lower bound on the computed d2 to every target other
// than the one returned, and the position it was taken at.  A search given these writes them; a FUSED search given them
// first tries to prove every hinted record still nearest and searches only for the queries where it cannot.
struct Coherence {
    double *lb2;      // [n]
    double *sref;     // [n, dim]
    uint32_t *count;  // [blocks of the test pass] queries of each block that could not be proven unchanged this iteration
    uint32_t *gate;   // 0: few unproven rows (eight lanes each) | 1: too many (the whole-cloud search), set per iteration
};
int grid_query(icp_ctx *ctx, const GridBuild &gb, const double *src, int64_t n, const int32_t *qorder, bool le,
               const double *thr, int32_t *idx_out, double *d2_out, const int32_t *done, const int32_t *hint = nullptr,
               bool pos_mode = false, const FusedApply *fa = nullptr, const Coherence *coh = nullptr);
// internal flag of closest_dispatch: correspondences (and hints) are SORTED POSITIONS into GridBuild::spts /
// sidx instead of original indices (icp_rigid_run keeps them that way between iterations: the gathers of the
// reduction kernels then walk the Morton-sorted copy almost sequentially)
constexpr unsigned ICP_INTERNAL_POS = 0x10000u;
// hint_from (with ICP_HINT_IDX): read the hints from this array instead of idx_out (the sharded loop alternates two
// correspondence buffers)
int closest_dispatch(icp_ctx *ctx, const double *src, int64_t n, const double *tgt, int64_t m, int dim,
                     unsigned flags, int32_t *idx_out, double *d2_out, const int32_t *done,
                     const GridBuild *grid = nullptr, const int32_t *hint_from = nullptr, const FusedApply *fa = nullptr,
                     const Coherence *coh = nullptr);
bool grid_worthwhile(int64_t n, int64_t m, unsigned flags);
bool grid_fusable();  // false under the single-pass switches of the grid search: rigid loops then take the unfused form
// exact k nearest (k = 6 or 8) by the key fl(sqrt(d2)), ties to the lower index, through a built grid
int grid_knn(icp_ctx *ctx, const GridBuild &gb, const double *q, int64_t nq, int k, int32_t *idx_out, double *dist_out);

// library-side RCCL (rccl.hip): collectives enqueued on a stream by the library itself
int rccl_allreduce_sum(icp_rccl *rc, double *buf, int count, hipStream_t stream);
int rccl_exchange(icp_rccl *rc, const double *send_buf, const int64_t *send_counts, double *recv_buf, const int64_t *recv_counts,
                  int width, hipStream_t stream);

// 64-bit shuffle helpers (wave = 64 lanes)
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_down_f64(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_down(lo, delta, 64);
    hi = __shfl_down(hi, delta, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += shfl_down_f64(v, o);
    return v;  // valid in lane 0
}
}  // namespace icp
