// Brute-force closest-point search.
//
// Exact reference arithmetic: d2 = ((dx*dx)+(dy*dy))+(dz*dz), dx = t-s, every
// operation rounded to fp64, no FMA (this file is compiled with -ffp-contract=off);
// argmin with the LOWEST index on ties and "first NaN wins" (np.argmin,
// icp_3d.py:39-42).
//
// Two kernels:
//  * closest_tiles  -- many queries: one query (x Q) per lane, the target streamed
//    through LDS tiles that every lane reads by broadcast.  The hot loop costs
//    8 fp64 VALU ops + 1 compare per pair; the (rare) best-update runs in a
//    wave-uniform slow path, scanning in ascending target order with a strict
//    compare so the lowest index wins.
//  * closest_lanes  -- few queries: queries are wave-uniform, lanes stride over the
//    targets, and a wavefront-level lexicographic (d2, index) argmin combines lanes.
// Both may split the target range over blockIdx.y; split results are merged by
// closest_merge with the same lexicographic rule.
#include "common.h"

namespace icp {

constexpr int TB = 256;     // threads per block (4 waves)
constexpr int TILE = 1024;  // target points per LDS tile
constexpr int CH = 4;       // targets per fast-path chunk

struct Cand {
    double d;
    int32_t i;
};

// total order behind np.argmin: NaN before everything, then smaller d2, then lower index
__device__ __forceinline__ bool better(double da, int32_t ia, double db, int32_t ib) {
    const bool an = da != da, bn = db != db;
    if (an || bn) return (an && bn) ? (ia < ib) : an;
    return da < db || (da == db && ia < ib);
}

template <int DIM>
__device__ __forceinline__ double sqdist(const double *t, const double *s) {
    double dx = t[0] - s[0];
    double dy = t[1] - s[1];
    double acc = dx * dx + dy * dy;
    if (DIM == 3) {
        double dz = t[2] - s[2];
        acc = acc + dz * dz;
    }
    return acc;
}

// LE = false: argmin.  LE = true: lowest index with d2 <= thr_in[q] (second pass of
// the compare-sqrt mode); its "distance" output is 0 when found, +inf otherwise.
template <int DIM, int Q, bool LE>
__global__ __launch_bounds__(TB) void closest_tiles(const double *__restrict__ src, int64_t n,
                                                    const double *__restrict__ tgt, int64_t m,
                                                    int64_t split_len, const double *__restrict__ thr_in,
                                                    double *__restrict__ out_d2, int32_t *__restrict__ out_idx,
                                                    const int32_t *__restrict__ done) {
    if (done && *done) return;
    __shared__ double tile[TILE * DIM];
    constexpr int PRE = TILE * DIM / TB;
    const int tid = threadIdx.x;
    const int64_t q0 = (int64_t)blockIdx.x * (TB * Q) + tid;
    const int64_t j_begin = (int64_t)blockIdx.y * split_len;
    const int64_t j_end = (j_begin + split_len < m) ? j_begin + split_len : m;

    double s[Q][DIM], thr[Q], best[Q];
    int32_t bidx[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int64_t q = q0 + (int64_t)k * TB;
        const int64_t qq = q < n ? q : n - 1;
#pragma unroll
        for (int c = 0; c < DIM; ++c) s[k][c] = src[qq * DIM + c];
        best[k] = __builtin_inf();
        thr[k] = LE ? thr_in[qq] : __builtin_inf();
        bidx[k] = (int32_t)j_begin;
    }

    double pre[PRE];
    auto prefetch = [&](int64_t base) {
        const int64_t lim = (j_end - base) * DIM;
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const int e = tid + i * TB;
            pre[i] = e < lim ? tgt[base * DIM + e] : 0.0;
        }
    };
    prefetch(j_begin);

    for (int64_t base = j_begin; base < j_end; base += TILE) {
        const int cnt = (j_end - base) < TILE ? (int)(j_end - base) : TILE;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PRE; ++i) tile[tid + i * TB] = pre[i];
        __syncthreads();
        if (base + TILE < j_end) prefetch(base + TILE);

        auto slow = [&](int j) {
            double t[DIM];
#pragma unroll
            for (int c = 0; c < DIM; ++c) t[c] = tile[j * DIM + c];
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                const double d = sqdist<DIM>(t, s[k]);
                if (LE) {
                    if (d <= thr[k]) {
                        thr[k] = -1.0;
                        best[k] = 0.0;
                        bidx[k] = (int32_t)(base + j);
                    }
                } else {
                    if (!(d >= thr[k]) && thr[k] == thr[k]) {
                        thr[k] = d;
                        best[k] = d;
                        bidx[k] = (int32_t)(base + j);
                    }
                }
            }
        };

        int j = 0;
#pragma unroll 1
        for (; j + CH <= cnt; j += CH) {
            bool any = false;
#pragma unroll
            for (int c4 = 0; c4 < CH; ++c4) {
                double t[DIM];
#pragma unroll
                for (int c = 0; c < DIM; ++c) t[c] = tile[(j + c4) * DIM + c];
#pragma unroll
                for (int k = 0; k < Q; ++k) {
                    const double d = sqdist<DIM>(t, s[k]);
                    any |= LE ? (d <= thr[k]) : !(d >= thr[k]);
                }
            }
            if (__builtin_amdgcn_ballot_w64(any) != 0) {
#pragma unroll
                for (int c4 = 0; c4 < CH; ++c4) slow(j + c4);
            }
        }
        for (; j < cnt; ++j) slow(j);
    }

#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int64_t q = q0 + (int64_t)k * TB;
        if (q < n) {
            const int64_t o = (int64_t)blockIdx.y * n + q;
            out_idx[o] = bidx[k];
            if (out_d2) out_d2[o] = best[k];
        }
    }
}

// Few queries: QB wave-uniform queries per block, lanes stride over the targets.
template <int DIM, int QB, bool LE>
__global__ __launch_bounds__(TB) void closest_lanes(const double *__restrict__ src, int64_t n,
                                                    const double *__restrict__ tgt, int64_t m,
                                                    int64_t split_len, const double *__restrict__ thr_in,
                                                    double *__restrict__ out_d2, int32_t *__restrict__ out_idx,
                                                    const int32_t *__restrict__ done) {
    if (done && *done) return;
    __shared__ double red_d[QB][TB / 64];
    __shared__ int32_t red_i[QB][TB / 64];
    const int tid = threadIdx.x;
    const int64_t qbase = (int64_t)blockIdx.x * QB;
    const int64_t j_begin = (int64_t)blockIdx.y * split_len;
    const int64_t j_end = (j_begin + split_len < m) ? j_begin + split_len : m;

    double s[QB][DIM], thr[QB], best[QB];
    int32_t bidx[QB];
#pragma unroll
    for (int k = 0; k < QB; ++k) {
        const int64_t qq = (qbase + k) < n ? (qbase + k) : n - 1;
#pragma unroll
        for (int c = 0; c < DIM; ++c) s[k][c] = src[qq * DIM + c];
        best[k] = __builtin_inf();
        thr[k] = LE ? thr_in[qq] : __builtin_inf();
        bidx[k] = (int32_t)j_begin;
    }
    // lane-private ascending scan: strict compare keeps the lowest index per lane
    for (int64_t j = j_begin + tid; j < j_end; j += TB) {
        double t[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) t[c] = tgt[j * DIM + c];
#pragma unroll
        for (int k = 0; k < QB; ++k) {
            const double d = sqdist<DIM>(t, s[k]);
            if (LE) {
                if (d <= thr[k]) {
                    thr[k] = -1.0;
                    best[k] = 0.0;
                    bidx[k] = (int32_t)j;
                }
            } else {
                if (!(d >= thr[k]) && thr[k] == thr[k]) {
                    thr[k] = d;
                    best[k] = d;
                    bidx[k] = (int32_t)j;
                }
            }
        }
    }
    // wavefront-level argmin: lexicographic (d2, index), lowest index on ties.
    // A lane that saw no target (or no hit) holds (+inf, j_begin), which ties with the
    // best possible +inf candidate of this split and so never changes the result.
#pragma unroll
    for (int k = 0; k < QB; ++k) {
        double d = best[k];
        int32_t i = bidx[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double od = shfl_xor_f64(d, o);
            const int32_t oi = __shfl_xor(i, o, 64);
            if (better(od, oi, d, i)) {
                d = od;
                i = oi;
            }
        }
        if ((tid & 63) == 0) {
            red_d[k][tid >> 6] = d;
            red_i[k][tid >> 6] = i;
        }
    }
    __syncthreads();
    if (tid < QB && qbase + tid < n) {
        double d = red_d[tid][0];
        int32_t i = red_i[tid][0];
        for (int w = 1; w < TB / 64; ++w)
            if (better(red_d[tid][w], red_i[tid][w], d, i)) {
                d = red_d[tid][w];
                i = red_i[tid][w];
            }
        const int64_t o = (int64_t)blockIdx.y * n + qbase + tid;
        out_idx[o] = i;
        if (out_d2) out_d2[o] = d;
    }
}

// Merge split results: part_*[s][q] -> out[q]; splits cover ascending index ranges.
__global__ void closest_merge(const double *__restrict__ part_d2, const int32_t *__restrict__ part_idx,
                              int splits, int64_t n, double *__restrict__ out_d2,
                              int32_t *__restrict__ out_idx, const int32_t *__restrict__ done) {
    if (done && *done) return;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    double d = part_d2[q];
    int32_t i = part_idx[q];
    for (int s = 1; s < splits; ++s) {
        const double ds = part_d2[(int64_t)s * n + q];
        const int32_t is = part_idx[(int64_t)s * n + q];
        if (better(ds, is, d, i)) {
            d = ds;
            i = is;
        }
    }
    out_idx[q] = i;
    if (out_d2) out_d2[q] = d;
}

// compare-sqrt mode: thr[q] = largest double x with sqrt_rn(x) == sqrt_rn(d2min[q]).
// sqrt is monotone, and the preimage of one rounded root is a handful of doubles wide.
__global__ void sqrt_class_upper(const double *__restrict__ d2min, int64_t n, double *__restrict__ thr) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    double x = d2min[q];
    const double r = __dsqrt_rn(x);
    if (x == x && x < __builtin_inf()) {
        for (int it = 0; it < 16; ++it) {
            const double nx = __longlong_as_double(__double_as_longlong(x) + 1);  // next up (x >= 0)
            if (!(nx < __builtin_inf()) || __dsqrt_rn(nx) != r) break;
            x = nx;
        }
    }
    thr[q] = x;
}

template <int DIM, bool LE>
static int launch_closest(icp_ctx *ctx, const double *src, int64_t n, const double *tgt, int64_t m,
                          const double *thr, int32_t *idx_out, double *d2_out, const int32_t *done) {
    hipStream_t st = ctx->stream;
    const int want_blocks = ctx->num_cu * 4;
    const int64_t max_splits = (m + TILE - 1) / TILE;
    const bool tiles = n >= 16384;
    int64_t qblocks;
    int Q = 1;
    if (tiles) {
        if (n >= (int64_t)TB * 4 * want_blocks) Q = 4;
        else if (n >= (int64_t)TB * 2 * want_blocks) Q = 2;
        qblocks = (n + (int64_t)TB * Q - 1) / ((int64_t)TB * Q);
    } else {
        qblocks = (n + 3) / 4;
    }
    int64_t splits = 1;
    if (qblocks < want_blocks) {
        splits = (want_blocks + qblocks - 1) / qblocks;
        if (!tiles) splits *= 2;
        if (splits > max_splits) splits = max_splits;
        if (splits > 65535) splits = 65535;
    }
    int64_t split_len = (m + splits - 1) / splits;
    split_len = (split_len + TILE - 1) / TILE * TILE;
    splits = (m + split_len - 1) / split_len;

    double *p_d2 = d2_out;
    int32_t *p_idx = idx_out;
    if (splits > 1) {
        if (int rc = ensure(ctx, ctx->split_d2, (size_t)splits * n * sizeof(double))) return rc;
        p_d2 = (double *)ctx->split_d2.ptr;
    }
    if (splits > 1) {
        if (int rc = ensure(ctx, ctx->split_idx, (size_t)splits * n * sizeof(int32_t))) return rc;
        p_idx = (int32_t *)ctx->split_idx.ptr;
    }
    dim3 grid((unsigned)qblocks, (unsigned)splits);
    if (tiles) {
        if (Q == 4)
            hipLaunchKernelGGL((closest_tiles<DIM, 4, LE>), grid, dim3(TB), 0, st, src, n, tgt, m, split_len, thr, p_d2, p_idx, done);
        else if (Q == 2)
            hipLaunchKernelGGL((closest_tiles<DIM, 2, LE>), grid, dim3(TB), 0, st, src, n, tgt, m, split_len, thr, p_d2, p_idx, done);
        else
            hipLaunchKernelGGL((closest_tiles<DIM, 1, LE>), grid, dim3(TB), 0, st, src, n, tgt, m, split_len, thr, p_d2, p_idx, done);
    } else {
        hipLaunchKernelGGL((closest_lanes<DIM, 4, LE>), grid, dim3(TB), 0, st, src, n, tgt, m, split_len, thr, p_d2, p_idx, done);
    }
    ICP_HIP_CHECK(hipGetLastError());
    if (splits > 1) {
        hipLaunchKernelGGL(closest_merge, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p_d2, p_idx,
                           (int)splits, n, d2_out, idx_out, done);
        ICP_HIP_CHECK(hipGetLastError());
    }
    return ICP_OK;
}

// The grid pays off once the brute-force scan costs more than the build (a few launches and
// one host sync) -- roughly a billion pair evaluations.
bool grid_worthwhile(int64_t n, int64_t m, unsigned flags) {
    if (flags & ICP_FORCE_BRUTE) return false;
    if (flags & ICP_FORCE_GRID) return true;
    return m >= 4096 && (double)n * (double)m >= 1073741824.0;
}

// internal entry used by icp_closest, the fused ICP loop and nicp_run.  `grid` (optional) is a
// prebuilt grid over tgt; without it one is built here when that is worthwhile.
int closest_dispatch(icp_ctx *ctx, const double *src, int64_t n, const double *tgt, int64_t m, int dim,
                     unsigned flags, int32_t *idx_out, double *d2_out, const int32_t *done,
                     const GridBuild *grid, const int32_t *hint_from, const FusedApply *fa, const Coherence *coh) {
    GridBuild local;
    if (!grid && grid_worthwhile(n, m, flags)) {
        if (int rc = grid_build(ctx, tgt, m, dim, local)) return rc;
        grid = &local;
    }
    const bool use_grid = grid && grid->usable && !(flags & ICP_FORCE_BRUTE);
    const bool pos_mode = use_grid && (flags & ICP_INTERNAL_POS);
    if (fa && (!use_grid || (flags & ICP_COMPARE_SQRT))) {
        set_error("closest_dispatch: the fused form exists for the grid argmin search only");
        return ICP_ERR_ARG;
    }
    auto argmin = [&](double *d2) -> int {
        if (use_grid) return grid_query(ctx, *grid, src, n, nullptr, false, nullptr, idx_out, d2, done,
                                        (flags & ICP_HINT_IDX) ? (hint_from ? hint_from : idx_out) : nullptr, pos_mode, fa, coh);
        return dim == 3 ? launch_closest<3, false>(ctx, src, n, tgt, m, nullptr, idx_out, d2, done)
                        : launch_closest<2, false>(ctx, src, n, tgt, m, nullptr, idx_out, d2, done);
    };
    if (!(flags & ICP_COMPARE_SQRT)) return argmin(d2_out);
    // pass 1: min d2; pass 2: lowest index inside the same rounded-sqrt class
    if (int rc = ensure(ctx, ctx->work_a, (size_t)n * sizeof(double))) return rc;
    if (int rc = ensure(ctx, ctx->work_b, (size_t)n * sizeof(double))) return rc;
    double *dmin = (double *)ctx->work_a.ptr, *thr = (double *)ctx->work_b.ptr;
    if (int rc = argmin(dmin)) return rc;
    hipLaunchKernelGGL(sqrt_class_upper, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dmin, n, thr);
    ICP_HIP_CHECK(hipGetLastError());
    int rc;
    if (use_grid) rc = grid_query(ctx, *grid, src, n, nullptr, true, thr, idx_out, nullptr, done, nullptr, pos_mode);
    else rc = dim == 3 ? launch_closest<3, true>(ctx, src, n, tgt, m, thr, idx_out, nullptr, done)
                       : launch_closest<2, true>(ctx, src, n, tgt, m, thr, idx_out, nullptr, done);
    if (rc) return rc;
    if (d2_out) ICP_HIP_CHECK(hipMemcpyAsync(d2_out, dmin, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return ICP_OK;
}

}  // namespace icp

extern "C" int icp_closest(icp_ctx *ctx, const double *src, int64_t n, const double *tgt, int64_t m, int dim,
                           unsigned flags, int32_t *idx_out, double *d2_out) {
    ICP_REQUIRE(ctx && src && tgt && idx_out, "null pointer");
    ICP_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
    ICP_REQUIRE(n >= 0 && m > 0 && m < 0x7fffffff, "need n >= 0 and 0 < m < 2^31");
    if (n == 0) return ICP_OK;
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    return icp::closest_dispatch(ctx, src, n, tgt, m, dim, flags, idx_out, d2_out, nullptr);
}

// ------------------------------------------------------------------ persistent target index
struct icp_target {
    const double *tgt;
    int64_t m;
    int dim;
    icp::GridBuild grid;
    icp::Scratch a, b, c, d;
};

extern "C" int icp_target_create(icp_ctx *ctx, const double *tgt, int64_t m, int dim, icp_target **out) {
    ICP_REQUIRE(ctx && tgt && out, "null pointer");
    ICP_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
    ICP_REQUIRE(m > 0 && m < 0x7fffffff, "need 0 < m < 2^31");
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    icp_target *t = new icp_target();
    t->tgt = tgt;
    t->m = m;
    t->dim = dim;
    if (m >= 1024) {
        if (int rc = icp::grid_build_into(ctx, tgt, m, dim, t->grid, t->a, t->b, t->c, t->d)) {
            delete t;
            return rc;
        }
    }
    *out = t;
    return ICP_OK;
}

extern "C" int icp_target_destroy(icp_ctx *ctx, icp_target *t) {
    if (!t) return ICP_OK;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    icp::Scratch *all[] = {&t->a, &t->b, &t->c, &t->d};
    for (icp::Scratch *s : all)
        if (s->ptr) (void)hipFree(s->ptr);
    delete t;
    return ICP_OK;
}

extern "C" int icp_closest_target(icp_ctx *ctx, const icp_target *t, const double *src, int64_t n, unsigned flags,
                                  int32_t *idx_out, double *d2_out) {
    ICP_REQUIRE(ctx && t && src && idx_out, "null pointer");
    ICP_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return ICP_OK;
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    if (!t->grid.usable) flags |= ICP_FORCE_BRUTE;
    return icp::closest_dispatch(ctx, src, n, t->tgt, t->m, t->dim, flags, idx_out, d2_out, nullptr, &t->grid);
}

extern "C" int icp_morton_sort(icp_ctx *ctx, const double *pts, int64_t n, int dim, double *sorted_out, int32_t *order_out) {
    ICP_REQUIRE(ctx && pts && sorted_out && order_out, "null pointer");
    ICP_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
    ICP_REQUIRE(n > 0 && n < 0x7fffffff, "need 0 < n < 2^31");
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    icp::GridBuild g;
    if (int rc = icp::grid_build(ctx, pts, n, dim, g, 1)) return rc;
    if (!g.usable) {
        icp::set_error("icp_morton_sort: non-finite coordinates");
        return ICP_ERR_UNSUPPORTED;
    }
    ICP_HIP_CHECK(hipMemcpyAsync(sorted_out, g.spts, (size_t)n * dim * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    ICP_HIP_CHECK(hipMemcpyAsync(order_out, g.sidx, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
    return ICP_OK;
}
