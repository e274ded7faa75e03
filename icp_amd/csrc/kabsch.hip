// Kabsch step, transform application, error metric and the fused rigid-ICP loop
// (SURVEY.md section 8a rows a3, a4, a5).
//
// FindTransform (icp_3d.py:68-90) = column means + centred cross-covariance
// W[a][b] = sum_i (nn_i[a]-tbar[a]) (s_i[b]-sbar[b]) + SVD.  On the GPU the sums are a
// fixed-shape fp64 tree reduction (per-thread accumulators -> wave shuffle -> LDS ->
// per-block partials -> one final block), so results are deterministic run to run.
// To need only ONE pass over the points (and one all-reduce when sharded) the sums
// are taken about fixed shifts cs, ct and corrected exactly:
//     W = sum (nn-ct)(s-cs)^T - (1/N) sum(nn-ct) sum(s-cs)^T.
// The 3x3 (2x2) SVD is a one-sided Jacobi iteration run by a single lane; the
// reference applies no reflection fix (Q2) and neither does this.
#include <vector>

#include <chrono>
#include <cstdlib>

#include "common.h"

namespace icp {
extern long long *g_grid_dbg;
extern hipEvent_t *g_grid_events;

constexpr int RB = 256;  // threads per reduction block

// device-resident state of one fused ICP run
struct RigidState {
    double sums[kSumSlots];
    double R[9];
    double t[3];
    double cs[3];
    double ct[3];
    double err;
    double err_prev;
    int32_t done;
    int32_t iters;
};

// Row range of a block in the streaming kernels below.  Workgroups are dealt round-robin over the 8 XCDs (blocks b
// and b + 8 share an L2): XCD k takes the k-th contiguous eighth of the rows and its blocks stride through it, so
// the gathers tgt[idx[i]] of one L2 stay within about an eighth of the (Morton-coherent) target instead of all of it
// (2^20 points: both kernels -14 %; 2^24: within +-5 %).
struct RowWalk {
    int64_t begin, end, step;
};
__device__ __forceinline__ RowWalk row_walk(int64_t n, int block_threads) {
    RowWalk w;
    if ((gridDim.x & 7u) == 0 && n >= (int64_t)gridDim.x * block_threads) {
        const int64_t seg = (n + 7) / 8, i0 = (int64_t)(blockIdx.x & 7) * seg;
        w.begin = i0 + (int64_t)(blockIdx.x >> 3) * block_threads + threadIdx.x;
        w.end = i0 + seg < n ? i0 + seg : n;
        w.step = (int64_t)(gridDim.x >> 3) * block_threads;
    } else {
        w.begin = (int64_t)blockIdx.x * block_threads + threadIdx.x;
        w.end = n;
        w.step = (int64_t)gridDim.x * block_threads;
    }
    return w;
}

template <int NS>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NS], double *__restrict__ out) {
    __shared__ double red[RB / 64][NS];
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const double v = wave_sum(acc[k]);
        if ((tid & 63) == 0) red[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid < NS) {
        double v = red[0][tid];
#pragma unroll
        for (int w = 1; w < RB / 64; ++w) v += red[w][tid];
        out[tid] = v;
    }
}

// partial[b][0..15]: [sum(nn-ct) | sum(s-cs) | sum (nn-ct)(s-cs)^T | count]; nn = tgt[idx] or nn[i]
template <int DIM, bool GATHER>
__global__ __launch_bounds__(RB) void shard_sums_kernel(const double *__restrict__ src, const double *__restrict__ tgt,
                                                        const int32_t *__restrict__ idx, int64_t n,
                                                        const double *__restrict__ cs, const double *__restrict__ ct,
                                                        double *__restrict__ partial, const int32_t *__restrict__ done) {
    if (done && *done) return;
    double acc[kSumSlots];
#pragma unroll
    for (int k = 0; k < kSumSlots; ++k) acc[k] = 0.0;
    double c_s[DIM], c_t[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
        c_s[c] = cs[c];
        c_t[c] = ct[c];
    }
    const RowWalk rw = row_walk(n, RB);
    for (int64_t i = rw.begin; i < rw.end; i += rw.step) {
        const int64_t j = GATHER ? (int64_t)idx[i] : i;
        double s[DIM], t[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            s[c] = src[i * DIM + c] - c_s[c];
            t[c] = tgt[j * DIM + c] - c_t[c];
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            acc[c] += t[c];
            acc[3 + c] += s[c];
        }
#pragma unroll
        for (int a = 0; a < DIM; ++a)
#pragma unroll
            for (int b = 0; b < DIM; ++b) acc[6 + a * DIM + b] += t[a] * s[b];
        acc[15] += 1.0;
    }
    block_reduce_store<kSumSlots>(acc, partial + (int64_t)blockIdx.x * kSumSlots);
}

// column sums of one array -> slots 0..DIM-1, count in slot 15
template <int DIM>
__global__ __launch_bounds__(RB) void column_sums_kernel(const double *__restrict__ a, int64_t n,
                                                         double *__restrict__ partial) {
    double acc[kSumSlots];
#pragma unroll
    for (int k = 0; k < kSumSlots; ++k) acc[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += (int64_t)gridDim.x * RB) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) acc[c] += a[i * DIM + c];
        acc[15] += 1.0;
    }
    block_reduce_store<kSumSlots>(acc, partial + (int64_t)blockIdx.x * kSumSlots);
}

// sum over b = g, g + 16, g + 32, ... of partial[b][slot], added in that order.  The loads of a batch are all issued
// before the first add, so a thread pays the memory latency once per 16 rows, not once per row (512 block rows:
// 9 us -> 3 us for the single-block kernels below); the order of the additions, hence the result, is unchanged.
__device__ __forceinline__ double strided_partial_sum(const double *__restrict__ partial, int nb, int g, int slot) {
    constexpr int STEP = RB / kSumSlots, BATCH = 16;
    double v = 0.0;
    for (int b0 = g; b0 < nb; b0 += STEP * BATCH) {
        double t[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int b = b0 + u * STEP;
            t[u] = b < nb ? partial[(int64_t)b * kSumSlots + slot] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u)
            if (b0 + u * STEP < nb) v += t[u];
    }
    return v;
}

// out[0..nout-1] = sum_b partial[b][0..nout-1], fixed summation tree; slots < mean_slots are
// divided by the count in slot 15 (np.mean: sum / N)
__global__ __launch_bounds__(RB) void reduce_partials_kernel(const double *__restrict__ partial, int nb,
                                                             double *__restrict__ out, int nout, int mean_slots,
                                                             const int32_t *__restrict__ done) {
    if (done && *done) return;
    __shared__ double red[RB / kSumSlots][kSumSlots];
    const int tid = threadIdx.x, slot = tid & (kSumSlots - 1), g = tid / kSumSlots;
    const double v = strided_partial_sum(partial, nb, g, slot);
    red[g][slot] = v;
    __syncthreads();
    if (tid < kSumSlots) {
        double r = red[0][tid];
        for (int k = 1; k < RB / kSumSlots; ++k) r += red[k][tid];
        red[0][tid] = r;
    }
    __syncthreads();
    if (tid < kSumSlots) {
        double r = red[0][tid];
        if (tid < mean_slots) r = r / red[0][15];
        if (tid < nout) out[tid] = r;
    }
}

// block-wide: red16[k] = sum_b partial[b][k] (same tree as reduce_partials_kernel); all threads sync
__device__ __forceinline__ void block_reduce_partials(const double *__restrict__ partial, int nb, double *red16) {
    __shared__ double red[RB / kSumSlots][kSumSlots];
    const int tid = threadIdx.x, slot = tid & (kSumSlots - 1), g = tid / kSumSlots;
    const double v = strided_partial_sum(partial, nb, g, slot);
    red[g][slot] = v;
    __syncthreads();
    if (tid < kSumSlots) {
        double r = red[0][tid];
        for (int k = 1; k < RB / kSumSlots; ++k) r += red[k][tid];
        red16[tid] = r;
    }
    __syncthreads();
}

// One rotation of the one-sided Jacobi iteration on columns (P, Q) of A (and of V).  P and Q are template
// parameters and every loop below has constant bounds: A, V and U live in registers.  (With run-time column indices
// the compiler kept the matrices in scratch memory, 128 bytes per lane, and every access of this serial chain paid a
// memory round trip: the solve took 16 us on its one lane.)
template <int DIM, int P, int Q>
__device__ __forceinline__ bool jacobi_pair(double (&A)[DIM][DIM], double (&V)[DIM][DIM]) {
    double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
    for (int r = 0; r < DIM; ++r) {
        alpha += A[r][P] * A[r][P];
        beta += A[r][Q] * A[r][Q];
        gamma += A[r][P] * A[r][Q];
    }
    // columns orthogonal to rounding: |gamma| <= 2.3e-16 sqrt(alpha beta), tested on the squares
    if (gamma == 0.0 || gamma * gamma <= 5.29e-32 * (alpha * beta)) return false;
    // tan of the rotation angle: t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)) with zeta = (beta - alpha) / (2 gamma),
    // multiplied through by |2 gamma| (one square root and one division on this serial path, not two and two)
    const double g2 = 2.0 * gamma, dlt = beta - alpha;
    const double tt = ((dlt >= 0) == (g2 >= 0) ? fabs(g2) : -fabs(g2)) / (fabs(dlt) + sqrt(g2 * g2 + dlt * dlt));
    const double c = rsqrt(1.0 + tt * tt), s = c * tt;
#pragma unroll
    for (int r = 0; r < DIM; ++r) {
        const double ap = A[r][P], aq = A[r][Q];
        A[r][P] = c * ap - s * aq;
        A[r][Q] = s * ap + c * aq;
        const double vp = V[r][P], vq = V[r][Q];
        V[r][P] = c * vp - s * vq;
        V[r][Q] = s * vp + c * vq;
    }
    return true;
}

// 3-D, rank 2: the missing left vector J completes U to a frame of V's handedness (LAPACK's choice there is arbitrary;
// SURVEY.md section 7 "SVD degeneracy")
template <int J>
__device__ __forceinline__ void complete_column3(double (&U)[3][3], double sg) {
    constexpr int a = (J + 1) % 3, b = (J + 2) % 3;
    U[0][J] = sg * (U[1][a] * U[2][b] - U[2][a] * U[1][b]);
    U[1][J] = sg * (U[2][a] * U[0][b] - U[0][a] * U[2][b]);
    U[2][J] = sg * (U[0][a] * U[1][b] - U[1][a] * U[0][b]);
}

// One-sided Jacobi SVD of a DIMxDIM matrix; returns the reference's R = (U VT)^T = V U^T.
template <int DIM>
__device__ void polar_rotation(const double (&W)[DIM][DIM], double (&R)[DIM][DIM]) {
    double A[DIM][DIM], V[DIM][DIM];
    // R does not depend on the scale of W: bring the largest entry to [1, 2) by an exact power of two, so that the
    // products of squared column norms below can neither overflow nor underflow whatever the units of the clouds
    double wmax = 0.0;
#pragma unroll
    for (int a = 0; a < DIM; ++a)
#pragma unroll
        for (int b = 0; b < DIM; ++b) wmax = fmax(wmax, fabs(W[a][b]));
    int ex = 0;
    if (wmax > 0.0 && wmax <= 1.7e308) (void)frexp(wmax, &ex);
#pragma unroll
    for (int a = 0; a < DIM; ++a)
#pragma unroll
        for (int b = 0; b < DIM; ++b) {
            A[a][b] = ldexp(W[a][b], 1 - ex);
            V[a][b] = a == b ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = jacobi_pair<DIM, 0, 1>(A, V);
        if constexpr (DIM == 3) {
            rotated |= jacobi_pair<DIM, 0, 2>(A, V);
            rotated |= jacobi_pair<DIM, 1, 2>(A, V);
        }
        if (!rotated) break;
    }
    // U[:,j] = A[:,j]/sigma_j.  Rank-deficient W: the missing left vectors are completed to a
    // right-handed frame (LAPACK's choice there is arbitrary; SURVEY.md section 7 "SVD degeneracy").
    double U[DIM][DIM], sig[DIM], smax = 0;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        double s2 = 0;
#pragma unroll
        for (int r = 0; r < DIM; ++r) s2 += A[r][j] * A[r][j];
        sig[j] = sqrt(s2);
        smax = sig[j] > smax ? sig[j] : smax;
    }
    bool ok[DIM];
    int nok = 0;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        ok[j] = sig[j] > 1e-300 && sig[j] > 1e-14 * smax;
        nok += ok[j];
#pragma unroll
        for (int r = 0; r < DIM; ++r) U[r][j] = ok[j] ? A[r][j] / sig[j] : 0.0;
    }
    if (nok < DIM) {
        if constexpr (DIM == 3) {
            if (nok == 2) {
                // u_j = u_a x u_b and v_j = v_a x v_b would keep both frames right-handed; V is
                // already orthonormal, so only U needs completing (sign follows det V).
                const double detV = V[0][0] * (V[1][1] * V[2][2] - V[1][2] * V[2][1]) - V[0][1] * (V[1][0] * V[2][2] - V[1][2] * V[2][0]) +
                                    V[0][2] * (V[1][0] * V[2][1] - V[1][1] * V[2][0]);
                const double sg = detV < 0 ? -1.0 : 1.0;
                if (!ok[0]) complete_column3<0>(U, sg);
                else if (!ok[1]) complete_column3<1>(U, sg);
                else complete_column3<2>(U, sg);
            } else {
#pragma unroll
                for (int a = 0; a < DIM; ++a)
#pragma unroll
                    for (int b = 0; b < DIM; ++b) U[a][b] = V[a][b];  // W ~ 0: R = I
            }
        } else {
            if (nok == 1) {
                const double detV = V[0][0] * V[1][1] - V[0][1] * V[1][0];
                if (ok[0]) {   // complete column 1 from column 0
                    const double sg = detV < 0 ? -1.0 : 1.0;
                    U[0][1] = -sg * U[1][0];
                    U[1][1] = sg * U[0][0];
                } else {       // complete column 0 from column 1
                    const double sg = detV < 0 ? 1.0 : -1.0;
                    U[0][0] = -sg * U[1][1];
                    U[1][0] = sg * U[0][1];
                }
            } else {
#pragma unroll
                for (int a = 0; a < DIM; ++a)
#pragma unroll
                    for (int b = 0; b < DIM; ++b) U[a][b] = V[a][b];  // W ~ 0: R = I
            }
        }
    }
#pragma unroll
    for (int a = 0; a < DIM; ++a)
#pragma unroll
        for (int b = 0; b < DIM; ++b) {
            double r = 0;
#pragma unroll
            for (int j = 0; j < DIM; ++j) r += V[a][j] * U[b][j];
            R[a][b] = r;
        }
}

template <int DIM>
__device__ void kabsch_from_sums(const double *sums, const double *__restrict__ cs, const double *__restrict__ ct,
                                 double *__restrict__ R_out, double *__restrict__ t_out);

template <int DIM>
__global__ void kabsch_solve_kernel(const double *__restrict__ sums, const double *__restrict__ cs,
                                    const double *__restrict__ ct, double *__restrict__ R_out,
                                    double *__restrict__ t_out, const int32_t *__restrict__ done) {
    if (done && *done) return;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    kabsch_from_sums<DIM>(sums, cs, ct, R_out, t_out);
}

template <int DIM>
__device__ void kabsch_from_sums(const double *sums, const double *__restrict__ cs, const double *__restrict__ ct,
                                 double *__restrict__ R_out, double *__restrict__ t_out) {
    const double N = sums[15];
    double sn[DIM], ss[DIM], tbar[DIM], sbar[DIM], W[DIM][DIM], R[DIM][DIM];
    for (int c = 0; c < DIM; ++c) {
        sn[c] = sums[c];
        ss[c] = sums[3 + c];
        tbar[c] = ct[c] + sn[c] / N;
        sbar[c] = cs[c] + ss[c] / N;
    }
    for (int a = 0; a < DIM; ++a)
        for (int b = 0; b < DIM; ++b) W[a][b] = sums[6 + a * DIM + b] - sn[a] * ss[b] / N;
    polar_rotation<DIM>(W, R);
    for (int a = 0; a < DIM; ++a)
        for (int b = 0; b < DIM; ++b) R_out[a * DIM + b] = R[a][b];
    for (int c = 0; c < DIM; ++c) {
        double acc = 0;
        for (int k = 0; k < DIM; ++k) acc += sbar[k] * R[k][c];
        t_out[c] = tbar[c] - acc;
    }
}

// ss = src R + t; partial err = sum ||nn - ss||  (icp_3d.py:113-114, 59-62)
template <int DIM, int MODE>  // MODE 0: apply only; 1: apply + err vs tgt[idx]; 2: distance(a,b) only
__global__ __launch_bounds__(RB) void apply_err_kernel(const double *__restrict__ src, const double *__restrict__ tgt,
                                                       const int32_t *__restrict__ idx, int64_t n,
                                                       const double *__restrict__ Rm, const double *__restrict__ tv,
                                                       double *__restrict__ ss_out, double *__restrict__ partial,
                                                       const int32_t *__restrict__ done) {
    if (done && *done) return;
    double R[DIM][DIM], t[DIM];
    if (MODE != 2) {
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
#pragma unroll
            for (int b = 0; b < DIM; ++b) R[a][b] = Rm[a * DIM + b];
            t[a] = tv[a];
        }
    }
    double acc[1] = {0.0};
    const RowWalk rw = row_walk(n, RB);
    for (int64_t i = rw.begin; i < rw.end; i += rw.step) {
        double s[DIM], o[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) s[c] = src[i * DIM + c];
        if (MODE != 2) {
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                double v = s[0] * R[0][c];
#pragma unroll
                for (int k = 1; k < DIM; ++k) v += s[k] * R[k][c];
                o[c] = v + t[c];
                if (ss_out) ss_out[i * DIM + c] = o[c];
            }
        } else {
#pragma unroll
            for (int c = 0; c < DIM; ++c) o[c] = s[c];
        }
        if (MODE != 0) {
            const int64_t j = MODE == 1 ? (int64_t)idx[i] : i;
            double d2 = 0;
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                const double d = tgt[j * DIM + c] - o[c];
                d2 += d * d;
            }
            acc[0] += sqrt(d2);
        }
    }
    if (MODE != 0) block_reduce_store<1>(acc, partial + (int64_t)blockIdx.x * kSumSlots);
}

template <int DIM>
__global__ void gather_kernel(const double *__restrict__ tgt, const int32_t *__restrict__ idx, int64_t n,
                              double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t j = idx[i];
#pragma unroll
    for (int c = 0; c < DIM; ++c) out[i * DIM + c] = tgt[j * DIM + c];
}

// idx_out[order[i]] = idx_sorted[i]: undo the Morton ordering of the queries (order may be null) and,
// when the loop kept sorted target positions instead of indices, translate them (sidx non-null).
// In place (idx_out == idx_sorted) only without `order`.
__global__ void unsort_idx_kernel(const int32_t *idx_sorted, const int32_t *__restrict__ order,
                                  const int32_t *__restrict__ sidx, int64_t n, int32_t *idx_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int32_t v = idx_sorted[i];
        idx_out[order ? order[i] : i] = sidx ? sidx[v] : v;
    }
}

__global__ void rigid_init_kernel(RigidState *st) {
    st->err = 100000000.0;  // icp_3d.py:105
    st->err_prev = 100000000.0;
    st->done = 0;
    st->iters = 0;
}

// icp_3d.py:118-121
__global__ void rigid_finish_kernel(const RigidState *st, int dim, double *T, int32_t *iters_out) {
    const int d1 = dim + 1;
    for (int a = 0; a < d1; ++a)
        for (int b = 0; b < d1; ++b) T[a * d1 + b] = a == b ? 1.0 : 0.0;
    for (int a = 0; a < dim; ++a) {
        for (int b = 0; b < dim; ++b) T[a * d1 + b] = st->R[a * dim + b];
        T[dim * d1 + a] = st->t[a];
    }
    if (iters_out) *iters_out = st->iters;
}

constexpr int kRowBlocksMax = 512;  // 2 blocks per CU; the final single-block reduce stays short
static inline int nblocks_for(int64_t n) {
    int64_t nb = (n + RB - 1) / RB;
    if (nb > kRowBlocksMax) nb = kRowBlocksMax;
    if (nb < 1) nb = 1;
    return (int)nb;
}

static int reduce_to(icp_ctx *ctx, const double *part, int nb, double *out_dev, int nout, int mean_slots,
                     const int32_t *done) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(RB), 0, ctx->stream, part, nb, out_dev, nout, mean_slots,
                       done);
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

// `part` (the [kMaxPartials][kSumSlots] block-partial rows): null = the context's grow-only buffer
static int shard_sums(icp_ctx *ctx, const double *src, const double *tgt, const int32_t *idx, int64_t n, int dim,
                      const double *cs, const double *ct, double *sums_out, const int32_t *done, int *nb_out = nullptr,
                      double *part = nullptr) {
    if (!part) {
        if (int rc = ensure(ctx, ctx->partials, sizeof(double) * kMaxPartials * kSumSlots)) return rc;
        part = (double *)ctx->partials.ptr;
    }
    const int nb = nblocks_for(n);
    hipStream_t st = ctx->stream;
    if (dim == 3) {
        if (idx) hipLaunchKernelGGL((shard_sums_kernel<3, true>), dim3(nb), dim3(RB), 0, st, src, tgt, idx, n, cs, ct, part, done);
        else hipLaunchKernelGGL((shard_sums_kernel<3, false>), dim3(nb), dim3(RB), 0, st, src, tgt, idx, n, cs, ct, part, done);
    } else {
        if (idx) hipLaunchKernelGGL((shard_sums_kernel<2, true>), dim3(nb), dim3(RB), 0, st, src, tgt, idx, n, cs, ct, part, done);
        else hipLaunchKernelGGL((shard_sums_kernel<2, false>), dim3(nb), dim3(RB), 0, st, src, tgt, idx, n, cs, ct, part, done);
    }
    ICP_HIP_CHECK(hipGetLastError());
    if (nb_out) {  // caller reduces the partial rows itself (fused kernels)
        *nb_out = nb;
        return ICP_OK;
    }
    return reduce_to(ctx, part, nb, sums_out, kSumSlots, 0, done);
}

static int column_sums(icp_ctx *ctx, const double *a, int64_t n, int dim, double *out, int mean, double *part = nullptr) {
    if (!part) {
        if (int rc = ensure(ctx, ctx->partials, sizeof(double) * kMaxPartials * kSumSlots)) return rc;
        part = (double *)ctx->partials.ptr;
    }
    const int nb = nblocks_for(n);
    if (dim == 3) hipLaunchKernelGGL(column_sums_kernel<3>, dim3(nb), dim3(RB), 0, ctx->stream, a, n, part);
    else hipLaunchKernelGGL(column_sums_kernel<2>, dim3(nb), dim3(RB), 0, ctx->stream, a, n, part);
    ICP_HIP_CHECK(hipGetLastError());
    return reduce_to(ctx, part, nb, out, dim, mean ? dim : 0, nullptr);
}

static int kabsch_solve(icp_ctx *ctx, const double *sums, int dim, const double *cs, const double *ct, double *R,
                        double *t, const int32_t *done) {
    if (dim == 3) hipLaunchKernelGGL(kabsch_solve_kernel<3>, dim3(1), dim3(64), 0, ctx->stream, sums, cs, ct, R, t, done);
    else hipLaunchKernelGGL(kabsch_solve_kernel<2>, dim3(1), dim3(64), 0, ctx->stream, sums, cs, ct, R, t, done);
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

// mode 0 apply, 1 apply+err(gather), 2 distance
static int apply_err(icp_ctx *ctx, int mode, const double *src, const double *tgt, const int32_t *idx, int64_t n,
                     int dim, const double *R, const double *t, double *ss_out, double *err_out, const int32_t *done,
                     int *nb_out = nullptr) {
    if (int rc = ensure(ctx, ctx->partials, sizeof(double) * kMaxPartials * kSumSlots)) return rc;
    const int nb = nblocks_for(n);
    double *part = (double *)ctx->partials.ptr;
    hipStream_t st = ctx->stream;
#define ICP_LAUNCH_AE(D, M) \
    hipLaunchKernelGGL((apply_err_kernel<D, M>), dim3(nb), dim3(RB), 0, st, src, tgt, idx, n, R, t, ss_out, part, done)
    if (dim == 3) {
        if (mode == 0) ICP_LAUNCH_AE(3, 0);
        else if (mode == 1) ICP_LAUNCH_AE(3, 1);
        else ICP_LAUNCH_AE(3, 2);
    } else {
        if (mode == 0) ICP_LAUNCH_AE(2, 0);
        else if (mode == 1) ICP_LAUNCH_AE(2, 1);
        else ICP_LAUNCH_AE(2, 2);
    }
#undef ICP_LAUNCH_AE
    ICP_HIP_CHECK(hipGetLastError());
    if (mode == 0) return ICP_OK;
    if (nb_out) {
        *nb_out = nb;
        return ICP_OK;
    }
    // only slot 0 of each partial row is written in this mode
    return reduce_to(ctx, part, nb, err_out, 1, 0, done);
}

}  // namespace icp

using namespace icp;

#define ICP_COMMON_ARGS(ctx, dim)                                \
    ICP_REQUIRE(ctx != nullptr, "null context");                 \
    ICP_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");     \
    ICP_HIP_CHECK(hipSetDevice(ctx->device))

extern "C" int icp_gather_points(icp_ctx *ctx, const double *tgt, const int32_t *idx, int64_t n, int dim, double *nn_out) {
    ICP_COMMON_ARGS(ctx, dim);
    ICP_REQUIRE(tgt && idx && nn_out && n >= 0, "bad argument");
    if (n == 0) return ICP_OK;
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (dim == 3) hipLaunchKernelGGL(gather_kernel<3>, dim3(nb), dim3(256), 0, ctx->stream, tgt, idx, n, nn_out);
    else hipLaunchKernelGGL(gather_kernel<2>, dim3(nb), dim3(256), 0, ctx->stream, tgt, idx, n, nn_out);
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

extern "C" int icp_distance_sum(icp_ctx *ctx, const double *a, const double *b, int64_t n, int dim, double *out_dev) {
    ICP_COMMON_ARGS(ctx, dim);
    ICP_REQUIRE(a && b && out_dev && n >= 0, "bad argument");
    return apply_err(ctx, 2, b, a, nullptr, n, dim, nullptr, nullptr, nullptr, out_dev, nullptr);
}

extern "C" int icp_column_sums(icp_ctx *ctx, const double *a, int64_t n, int dim, double *out_dev) {
    ICP_COMMON_ARGS(ctx, dim);
    ICP_REQUIRE(a && out_dev && n > 0, "bad argument");
    if (int rc = ensure(ctx, ctx->small, kSmallBytes)) return rc;
    double *tmp = (double *)ctx->small.ptr + 256;
    if (int rc = column_sums(ctx, a, n, dim, tmp, 0)) return rc;
    ICP_HIP_CHECK(hipMemcpyAsync(out_dev, tmp, sizeof(double) * dim, hipMemcpyDeviceToDevice, ctx->stream));
    return ICP_OK;
}

extern "C" int icp_find_transform(icp_ctx *ctx, const double *src, const double *nn, int64_t n, int dim,
                                  double *R_out_dev, double *t_out_dev) {
    ICP_COMMON_ARGS(ctx, dim);
    ICP_REQUIRE(src && nn && R_out_dev && t_out_dev && n > 0, "bad argument");
    if (int rc = ensure(ctx, ctx->small, kSmallBytes)) return rc;
    double *base = (double *)ctx->small.ptr + 256;
    double *cs = base + 16, *ct = base + 32, *sums = base + 48;
    // pass 1: the two column means (icp_3d.py:69,71); pass 2: cross-covariance about them
    if (int rc = column_sums(ctx, src, n, dim, cs, 1)) return rc;
    if (int rc = column_sums(ctx, nn, n, dim, ct, 1)) return rc;
    if (int rc = shard_sums(ctx, src, nn, nullptr, n, dim, cs, ct, sums, nullptr)) return rc;
    return kabsch_solve(ctx, sums, dim, cs, ct, R_out_dev, t_out_dev, nullptr);
}

// ---- the entry point that allocates nothing (include/icp_hip.h: icp_find_transform_noalloc) ----
// Block-partial rows + the small state of one FindTransform, as a file-scope device array: every GPU the library runs
// on holds its own copy (66 048 bytes) until the process ends.  `scratch` is a host-side stand-in for the symbol, not
// a device address: host code only ever uses the per-device pointer that noalloc_init() obtains.
constexpr int kNoallocStateDoubles = 64;  // [cs 16 | ct 16 | sums 16 | spare 16]
constexpr size_t kNoallocDoubles = (size_t)icp::kRowBlocksMax * icp::kSumSlots + kNoallocStateDoubles;
static __device__ __align__(256) double scratch[kNoallocDoubles];

namespace icp {
// called by icp_ctx_create with the context's device current: each device has its own copy of the array
int noalloc_init(icp_ctx *ctx) {
    void *p = nullptr;
    ICP_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    ctx->noalloc_scratch = (double *)p;
    return ICP_OK;
}
}  // namespace icp

extern "C" int icp_find_transform_noalloc(icp_ctx *ctx, const double *src, const double *nn, int64_t n, int dim,
                                          double *R_out_dev, double *t_out_dev) {
    ICP_COMMON_ARGS(ctx, dim);
    ICP_REQUIRE(src && nn && R_out_dev && t_out_dev && n > 0, "bad argument");
    if (n > ICP_NOALLOC_MAX_ROWS) {
        set_error("icp_find_transform_noalloc: n = %lld exceeds the limit of %lld rows", (long long)n,
                  (long long)ICP_NOALLOC_MAX_ROWS);
        return ICP_ERR_UNSUPPORTED;
    }
    if (!ctx->noalloc_scratch) {
        set_error("icp_find_transform_noalloc: the context holds no address of the device scratch array");
        return ICP_ERR_HIP;
    }
    // same kernels, same launch shapes and same summation trees as icp_find_transform: identical bits.  Every block
    // writes its whole partial row before the single-block reductions read it, so nothing has to start at zero.
    double *part = ctx->noalloc_scratch;
    double *base = part + (size_t)kRowBlocksMax * kSumSlots;
    double *cs = base, *ct = base + 16, *sums = base + 32;
    if (int rc = column_sums(ctx, src, n, dim, cs, 1, part)) return rc;
    if (int rc = column_sums(ctx, nn, n, dim, ct, 1, part)) return rc;
    if (int rc = shard_sums(ctx, src, nn, nullptr, n, dim, cs, ct, sums, nullptr, nullptr, part)) return rc;
    return kabsch_solve(ctx, sums, dim, cs, ct, R_out_dev, t_out_dev, nullptr);
}

extern "C" int icp_apply_transform(icp_ctx *ctx, const double *src, int64_t n, int dim, const double *R_dev,
                                   const double *t_dev, double *out) {
    ICP_COMMON_ARGS(ctx, dim);
    ICP_REQUIRE(src && R_dev && t_dev && out && n >= 0, "bad argument");
    if (n == 0) return ICP_OK;
    return apply_err(ctx, 0, src, nullptr, nullptr, n, dim, R_dev, t_dev, out, nullptr, nullptr);
}

extern "C" int icp_shard_sums(icp_ctx *ctx, const double *src, const double *tgt, const int32_t *idx, int64_t n,
                              int dim, const double *cs_dev, const double *ct_dev, double *sums_out_dev) {
    ICP_COMMON_ARGS(ctx, dim);
    ICP_REQUIRE(src && tgt && idx && cs_dev && ct_dev && sums_out_dev && n >= 0, "bad argument");
    return shard_sums(ctx, src, tgt, idx, n, dim, cs_dev, ct_dev, sums_out_dev, nullptr);
}

extern "C" int icp_kabsch_solve(icp_ctx *ctx, const double *sums_dev, int dim, const double *cs_dev,
                                const double *ct_dev, double *R_out_dev, double *t_out_dev) {
    ICP_COMMON_ARGS(ctx, dim);
    ICP_REQUIRE(sums_dev && cs_dev && ct_dev && R_out_dev && t_out_dev, "null pointer");
    return kabsch_solve(ctx, sums_dev, dim, cs_dev, ct_dev, R_out_dev, t_out_dev, nullptr);
}

extern "C" int icp_shard_apply_err(icp_ctx *ctx, const double *src, const double *tgt, const int32_t *idx, int64_t n,
                                   int dim, const double *R_dev, const double *t_dev, double *ss_out, double *err_out_dev) {
    ICP_COMMON_ARGS(ctx, dim);
    ICP_REQUIRE(src && tgt && idx && R_dev && t_dev && err_out_dev && n >= 0, "bad argument");
    // the reduction writes its one result straight to the caller's slot (no staging copy: one launch less per iteration)
    return apply_err(ctx, 1, src, tgt, idx, n, dim, R_dev, t_dev, ss_out, err_out_dev, nullptr);
}

// The two one-time structures of a run: the search grid over the target and the Morton order of the source.  They are
// independent, and each is a chain of mostly small kernels with one host read-back in the middle, so the source's
// goes first, on the context's side stream, and keeps running there (counting, sorting, gathering) while the
// target's statistics kernels -- single-block sorts of a sample -- and its read-back happen on the main stream;
// the main stream waits for the side stream at the end.  2^20 x 2^20: set-up 0.78 -> see DESIGN.md 3.1b.
namespace icp {
template <typename Between>
static int build_grids(icp_ctx *ctx, const double *src, int64_t n, const double *tgt, int64_t m, int dim, unsigned flags,
                       GridBuild &grid, GridBuild &qsort, Between between) {
    // `between`: work of the caller that depends on neither structure (the rigid loop's two means), enqueued on the main
    // stream behind the target's statistics kernels: it runs while the host waits for the read-backs and enqueues the sorts
    if (!grid_worthwhile(n, m, flags)) return between();
    const bool sort_source = n >= 32768;
    bool side_ok = sort_source;
    if (side_ok && !ctx->side && hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) != hipSuccess) side_ok = false;
    if (side_ok && !ctx->side_ev && hipEventCreateWithFlags(&ctx->side_ev, hipEventDisableTiming) != hipSuccess) side_ok = false;
    if (!side_ok) (void)hipGetLastError();
    hipStream_t main_stream = ctx->stream;
    if (sort_source && side_ok) {
        // statistics of both clouds first (8 launches), then the two read-backs: while the host enqueues the ~30 launches
        // of the source's sort, the GPU is already through the target's sample statistics (set-up 0.62 -> see DESIGN.md 3.1b)
        GridPending ps, pt;
        ICP_HIP_CHECK(hipEventRecord(ctx->side_ev, main_stream));      // whatever produced src is queued on the main stream
        ICP_HIP_CHECK(hipStreamWaitEvent(ctx->side, ctx->side_ev, 0));
        if (int rc = grid_build_begin(ctx, tgt, m, dim, grid, ctx->grid_a, ctx->grid_b, ctx->grid_c, ctx->grid_d, false, pt)) return rc;
        ctx->stream = ctx->side;
        int rc = grid_build_begin(ctx, src, n, dim, qsort, ctx->sort_a, ctx->sort_b, ctx->sort_c, ctx->sort_d, true, ps);
        ctx->stream = main_stream;
        if (!rc) rc = between();
        ctx->stream = ctx->side;
        if (!rc) rc = grid_build_finish(ctx, ps, qsort);
        ctx->stream = main_stream;
        if (rc) return rc;
        if ((rc = grid_build_finish(ctx, pt, grid))) return rc;
        ICP_HIP_CHECK(hipEventRecord(ctx->side_ev, ctx->side));
        ICP_HIP_CHECK(hipStreamWaitEvent(main_stream, ctx->side_ev, 0));
        return ICP_OK;
    }
    {
        GridPending pt;
        if (int rc = grid_build_begin(ctx, tgt, m, dim, grid, ctx->grid_a, ctx->grid_b, ctx->grid_c, ctx->grid_d, false, pt)) return rc;
        if (int rc = between()) return rc;
        if (int rc = grid_build_finish(ctx, pt, grid)) return rc;
    }
    if (sort_source && grid.usable) {
        if (int rc = grid_build(ctx, src, n, dim, qsort, 1)) return rc;
    }
    return ICP_OK;
}
}  // namespace icp

// ---------------------------------------------------------------- source-sharded loop
// One rank's share of icp_3d.py:102-121 with the source rows split over ranks and the target replicated.  Per
// iteration ONE all-reduce of 17 doubles: the 16 shifted sums of this iteration and the error of the previous one,
// whose stop test (icp_3d.py:116) is therefore evaluated one Match late -- that Match goes into the other of two
// correspondence buffers and is discarded.  Every decision is taken on the device from all-reduced values (identical
// on every rank), so the host enqueues max_iter iterations and max_iter + 2 collectives without reading anything
// back, and every rank issues the same collectives whatever the stop test says.
namespace icp {

// buf[0..dim) = local column sums of the source shard, buf[3] = local row count
__global__ __launch_bounds__(RB) void dist_pack_means_kernel(const double *__restrict__ partial, int nb, double *__restrict__ buf) {
    __shared__ double tot[kSumSlots];
    block_reduce_partials(partial, nb, tot);
    if (threadIdx.x < 3) buf[threadIdx.x] = tot[threadIdx.x];
    if (threadIdx.x == 3) buf[3] = tot[15];
}
__global__ void dist_means_kernel(const double *__restrict__ buf, int dim, RigidState *st) {
    if (threadIdx.x < dim) st->cs[threadIdx.x] = buf[threadIdx.x] / buf[3];  // np.mean: sum / N (icp_3d.py:69)
}
// sum of the per-block error sums a fused search left (fixed tree: deterministic); valid in thread 0
__device__ __forceinline__ double block_sum_err(const double *__restrict__ err_part, int nerr, double *red) {
    const int tid = threadIdx.x;
    double v = 0.0;
    int b = tid;
    for (; b + 7 * RB < nerr; b += 8 * RB) {   // loads of a batch issued before the first add (same order of additions)
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = err_part[b + u * RB];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += x[u];
    }
    for (; b < nerr; b += RB) v += err_part[b];
    red[tid] = v;
    __syncthreads();
    for (int o = RB / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    return red[0];
}

// buf[0..15] = this rank's sums of iteration `it`, buf[16] = its error of iteration it - 1 (st->err, or -- fused
// loop -- the sum of the error partials the search of iteration `it` formed on the way)
__global__ __launch_bounds__(RB) void dist_pack_kernel(const double *__restrict__ partial, int nb, const RigidState *st,
                                                       int first, double *__restrict__ buf,
                                                       const double *__restrict__ err_part, int nerr) {
    __shared__ double tot[kSumSlots];
    __shared__ double red[RB];
    const int done = st->done;
    if (!done) block_reduce_partials(partial, nb, tot);
    double err = 0.0;
    if (!done && !first) err = err_part ? block_sum_err(err_part, nerr, red) : st->err;
    if (threadIdx.x < kSumSlots) buf[threadIdx.x] = done ? 0.0 : tot[threadIdx.x];
    if (threadIdx.x == 0) buf[kSumSlots] = err;
}

// One rank, fused loop: the reduction of the sums, the error of the previous iteration with its bookkeeping
// (icp_3d.py:114-117) and the solve of this one in a single launch.
template <int DIM>
__global__ __launch_bounds__(RB) void fused_solve_kernel(const double *__restrict__ partial, int nb,
                                                         const double *__restrict__ err_part, int nerr, RigidState *st,
                                                         int first, double tol, int use_tol, double *__restrict__ err_hist) {
    if (st->done) return;
    __shared__ double tot[kSumSlots];
    __shared__ double red[RB];
    double csr[DIM], ctr[DIM], e_prev = 0.0;
    int iters = 0;
    if (threadIdx.x == 0) {   // fetched before the sums they are combined with
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            csr[c] = st->cs[c];
            ctr[c] = st->ct[c];
        }
        e_prev = st->err_prev;
        iters = st->iters;
    }
    const double err = first ? 0.0 : block_sum_err(err_part, nerr, red);
    block_reduce_partials(partial, nb, tot);
    if (threadIdx.x != 0) return;
    if (!first) {
        st->err = err;
        st->err_prev = err;
        if (err_hist) err_hist[iters] = err;
        st->iters = iters + 1;
        if (use_tol && (e_prev - err) < tol) {
            st->done = 1;    // R, t stay those of the last executed iteration
            return;
        }
    }
    for (int k = 0; k < kSumSlots; ++k) st->sums[k] = tot[k];
    kabsch_from_sums<DIM>(tot, csr, ctr, st->R, st->t);
}
// after the all-reduce: end-of-iteration bookkeeping of iteration it - 1 (icp_3d.py:114-117), then the solve of `it`
template <int DIM>
__global__ void dist_solve_kernel(const double *__restrict__ buf, RigidState *st, int first, double tol, int use_tol,
                                  double *__restrict__ err_hist) {
    if (threadIdx.x || blockIdx.x || st->done) return;
    if (!first) {
        const double e = st->err_prev, err = buf[kSumSlots];
        st->err_prev = err;
        if (err_hist) err_hist[st->iters] = err;
        st->iters += 1;
        if (use_tol && (e - err) < tol) {
            st->done = 1;
            return;
        }
    }
    for (int k = 0; k < kSumSlots; ++k) st->sums[k] = buf[k];
    kabsch_from_sums<DIM>(st->sums, st->cs, st->ct, st->R, st->t);
}
// local error of this iteration -> st->err (reduced over ranks by the next iteration's collective)
__global__ __launch_bounds__(RB) void dist_err_kernel(const double *__restrict__ partial, int nb, RigidState *st) {
    if (st->done) return;
    __shared__ double red[RB];
    const int tid = threadIdx.x;
    double v = 0.0;
    for (int b = tid; b < nb; b += RB) v += partial[(int64_t)b * kSumSlots];
    red[tid] = v;
    __syncthreads();
    for (int o = RB / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) st->err = red[0];
}
// One rank: dist_err_kernel + dist_last_pack_kernel + dist_last_step_kernel + rigid_finish_kernel in one launch (the tail
// of a call was six dependent launches; same sums in the same order)
__global__ __launch_bounds__(RB) void rigid_tail_kernel(const double *__restrict__ partial, int nb, RigidState *st,
                                                        double *__restrict__ err_hist, int dim, double *__restrict__ T,
                                                        int32_t *__restrict__ iters_out) {
    __shared__ double red[RB];
    const int tid = threadIdx.x;
    const int done = st->done;
    if (!done) {
        double v = 0.0;
        for (int b = tid; b < nb; b += RB) v += partial[(int64_t)b * kSumSlots];
        red[tid] = v;
        __syncthreads();
        for (int o = RB / 2; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
    }
    if (tid != 0) return;
    int iters = st->iters;
    if (!done) {
        const double e = red[0];
        st->err = e;
        if (err_hist) err_hist[iters] = e;
        st->err_prev = e;
        iters += 1;
        st->iters = iters;
    }
    const int d1 = dim + 1;
    for (int a = 0; a < d1; ++a)
        for (int b = 0; b < d1; ++b) T[a * d1 + b] = a == b ? 1.0 : 0.0;
    for (int a = 0; a < dim; ++a) {
        for (int b = 0; b < dim; ++b) T[a * d1 + b] = st->R[a * dim + b];
        T[dim * d1 + a] = st->t[a];
    }
    if (iters_out) *iters_out = iters;
}
__global__ void dist_last_pack_kernel(const RigidState *st, double *__restrict__ buf) { buf[0] = st->done ? 0.0 : st->err; }
__global__ void dist_last_step_kernel(const double *__restrict__ buf, RigidState *st, double *__restrict__ err_hist) {
    if (st->done) return;
    if (err_hist) err_hist[st->iters] = buf[0];
    st->err_prev = buf[0];
    st->iters += 1;
}
// correspondences of the last EXECUTED iteration: buffer (iters - 1) & 1
__global__ void dist_unsort_idx_kernel(const int32_t *__restrict__ b0, const int32_t *__restrict__ b1, const RigidState *st,
                                       const int32_t *__restrict__ order, const int32_t *__restrict__ sidx, int64_t n,
                                       int32_t *__restrict__ idx_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int32_t v = (((st->iters - 1) & 1) ? b1 : b0)[i];
        idx_out[order ? order[i] : i] = sidx ? sidx[v] : v;
    }
}

static bool comm_active(const icp_comm *cm) { return cm && (cm->world > 1 || (cm->flags & ICP_COMM_FORCE)); }
static int comm_allreduce(icp_ctx *ctx, const icp_comm *cm, int count) {
    if (!comm_active(cm)) return ICP_OK;
    if (cm->rccl) return rccl_allreduce_sum(cm->rccl, cm->reduce_buf_dev, count, ctx->stream);  // on the stream: no host code
    if (int rc = cm->allreduce_sum(cm->user, count)) {
        set_error("icp_rigid_run_dist: the all-reduce callback failed (%d)", rc);
        return ICP_ERR_COMM;
    }
    return ICP_OK;
}

}  // namespace icp

namespace icp {
// icp_rigid_run (comm = null) and icp_rigid_run_dist.  Per iteration: search -> the shard's 16 sums -> [all-reduce] ->
// solve -> error; the error of iteration k is only needed by the stop test at the end of k, so it is reduced together
// with the sums of k + 1 and the test runs one Match late (that Match goes into the other of two correspondence
// buffers and is discarded).  Fused form (grid search, plain comparison; ICP_NO_FUSE keeps the other): the moved source
// is never written -- the search kernels apply R, t to the original rows themselves and add up the distances to their
// hints, which ARE the previous iteration's error terms -- so an iteration is search (2 launches) + sums + solve.
static int rigid_loop(icp_ctx *ctx, const double *src, int64_t n, const double *tgt, int64_t m, int dim, double tol,
                      int max_iter, int use_tol, unsigned flags, double *T_out_dev, double *err_hist_dev,
                      int32_t *iters_out_dev, int32_t *idx_out, const icp_comm *comm) {
    const bool dist = comm_active(comm);  // every sum goes through the (all-)reduce buffer
    if (int rc = ensure(ctx, ctx->small, kSmallBytes)) return rc;
    if (int rc = ensure(ctx, ctx->partials, sizeof(double) * kMaxPartials * kSumSlots)) return rc;
    if (int rc = ensure(ctx, ctx->work_d, (size_t)n * sizeof(int32_t) * 2)) return rc;
    // the target never changes inside the loop: build its search grid once.  With the grid, put the source into
    // Morton order once: lanes of a wave then search the same few cells, and a rigid motion keeps neighbours
    // neighbours, so the order stays good for every iteration.  Reductions run over the sorted copy (any order is a
    // valid sum order); only the user-visible correspondences are scattered back to the caller's order.
    GridBuild grid, qsort;
    const int32_t *order = nullptr;
    static const bool timing = getenv("ICP_RIGID_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tt0 = timing ? now() : 0.0;
    RigidState *st = (RigidState *)ctx->small.ptr;
    // a single rank needs no buffer from the caller: the 17 doubles live behind the RigidState block
    double *buf = dist ? comm->reduce_buf_dev : (double *)ctx->small.ptr + 128;
    static_assert(sizeof(RigidState) <= 128 * sizeof(double), "RigidState overlaps the local reduce buffer");
    double *part = (double *)ctx->partials.ptr;
    // The two means depend on neither the grid nor the Morton order, so they are enqueued between the halves of the
    // builds (behind the target's statistics kernels on the main stream) and run while the host waits for the
    // read-backs: 37 us off the critical path of a call.  Global mean of the ORIGINAL source rows (Q1) from (all-reduced)
    // column sums; the target is replicated, its mean is local.  (Their block partials, at most 512 rows, end below the
    // region the source's bounding-box kernels use on the side stream: grid_build_begin.)
    const double *src_rows = src;
    auto enqueue_means = [&]() -> int {
        hipStream_t s = ctx->stream;
        hipLaunchKernelGGL(rigid_init_kernel, dim3(1), dim3(1), 0, s, st);
        const int nb = nblocks_for(n);
        static_assert(kRowBlocksMax * kSumSlots <= 1024 * 8, "column-sum partials reach into the side stream's bounding-box partials");
        if (dim == 3) hipLaunchKernelGGL(column_sums_kernel<3>, dim3(nb), dim3(RB), 0, s, src_rows, n, part);
        else hipLaunchKernelGGL(column_sums_kernel<2>, dim3(nb), dim3(RB), 0, s, src_rows, n, part);
        hipLaunchKernelGGL(dist_pack_means_kernel, dim3(1), dim3(RB), 0, s, (const double *)part, nb, buf);
        ICP_HIP_CHECK(hipGetLastError());
        if (int rc = comm_allreduce(ctx, comm, 4)) return rc;
        hipLaunchKernelGGL(dist_means_kernel, dim3(1), dim3(64), 0, s, (const double *)buf, dim, st);
        return column_sums(ctx, tgt, m, dim, st->ct, 1);
    };
    if (int rc = build_grids(ctx, src, n, tgt, m, dim, flags, grid, qsort, enqueue_means)) return rc;
    if (timing) {
        const double t1 = now();
        (void)hipStreamSynchronize(ctx->stream);
        std::fprintf(stderr, "rigid_loop: build_grids host %.0f us, +sync %.0f us\n", t1 - tt0, now() - tt0);
    }
    if (!grid.usable) flags |= ICP_FORCE_BRUTE;
    if (grid.usable && qsort.usable) {
        src = qsort.spts;
        order = qsort.sidx;
    }
    // With the grid, correspondences stay SORTED TARGET POSITIONS inside the loop: the reductions then gather from the
    // Morton-sorted copy of the target (queries are in Morton order too, so neighbouring lanes read neighbouring
    // records) and last iteration's position is the next search's hint directly.
    const bool pos = grid.usable && !(flags & ICP_FORCE_BRUTE);
    const bool fused = pos && !(flags & (ICP_COMPARE_SQRT | ICP_NO_FUSE)) && grid_fusable();
    double *ss = nullptr;
    if (!fused) {
        if (int rc = ensure(ctx, ctx->work_c, (size_t)n * dim * sizeof(double))) return rc;
        ss = (double *)ctx->work_c.ptr;
    } else if (int rc = ensure(ctx, ctx->err_part, (size_t)(2 * ((n + 255) / 256) + 16) * sizeof(double))) {
        return rc;   // (an upper bound of the search's padded block count: no reallocation inside the loop)
    }
    int32_t *ib[2] = {(int32_t *)ctx->work_d.ptr, (int32_t *)ctx->work_d.ptr + n};
    hipStream_t s = ctx->stream;
    if (ss) ICP_HIP_CHECK(hipMemcpyAsync(ss, src, (size_t)n * dim * sizeof(double), hipMemcpyDeviceToDevice, s));  // icp_3d.py:104
    const int32_t *done = &st->done;
    const double *tgt_rows = pos ? grid.spts : tgt;
    // coherent loop (ICP_COHERENT, icp_hip.h; grid_query_kernel<BND>): per source row a bound and the position it was taken at
    static const int coh_env = getenv("ICP_COHERENT") ? atoi(getenv("ICP_COHERENT")) : -1;
    const bool coherent = fused && (coh_env >= 0 ? coh_env != 0 : (flags & ICP_COHERENT) != 0);
    Coherence coh = {nullptr, nullptr, nullptr, nullptr};
    if (coherent) {
        const size_t cnt_bytes = (((size_t)(n + 255) / 256 + 32) * sizeof(uint32_t) + 255) & ~(size_t)255;   // (gate | padded block count)
        if (int rc = ensure(ctx, ctx->coh, cnt_bytes + (size_t)n * (dim + 1) * sizeof(double))) return rc;
        coh.gate = (uint32_t *)ctx->coh.ptr;
        coh.count = coh.gate + 16;
        ICP_HIP_CHECK(hipMemsetAsync(coh.gate, 0, sizeof(uint32_t), ctx->stream));   // (the first coherent iteration: eight-lane form)
        coh.lb2 = (double *)((char *)ctx->coh.ptr + cnt_bytes);
        coh.sref = coh.lb2 + n;
    }
    flags = (flags & ~(unsigned)(ICP_HINT_IDX | ICP_NO_FUSE | ICP_COHERENT)) | (pos ? ICP_INTERNAL_POS : 0u);
    int tail_nb = 0;
    for (int it = 0; it < max_iter; ++it) {
        int32_t *idx = ib[it & 1];
        const int first = it == 0 ? 1 : 0;
        // from the second iteration on the previous correspondences are a tight starting bound
        double *err_part = nullptr;
        int nerr = 0;
        const FusedApply fa = {st->R, st->t, &err_part, &nerr};
        if (int rc = closest_dispatch(ctx, fused ? src : ss, n, tgt, m, dim, first ? flags : (flags | ICP_HINT_IDX), idx, nullptr,
                                      done, &grid, first ? nullptr : ib[(it - 1) & 1], (fused && !first) ? &fa : nullptr,
                                      coherent ? &coh : nullptr))
            return rc;
        int nb = 0;
        if (int rc = shard_sums(ctx, src, tgt_rows, idx, n, dim, st->cs, st->ct, st->sums, done, &nb)) return rc;
        if (!dist && fused) {
            if (dim == 3) hipLaunchKernelGGL(fused_solve_kernel<3>, dim3(1), dim3(RB), 0, s, (const double *)part, nb, (const double *)err_part, nerr, st, first, tol, use_tol, err_hist_dev);
            else hipLaunchKernelGGL(fused_solve_kernel<2>, dim3(1), dim3(RB), 0, s, (const double *)part, nb, (const double *)err_part, nerr, st, first, tol, use_tol, err_hist_dev);
        } else {
            hipLaunchKernelGGL(dist_pack_kernel, dim3(1), dim3(RB), 0, s, (const double *)part, nb, (const RigidState *)st, first, buf, (const double *)err_part, nerr);
            ICP_HIP_CHECK(hipGetLastError());
            if (int rc = comm_allreduce(ctx, comm, kSumSlots + 1)) return rc;
            if (dim == 3) hipLaunchKernelGGL(dist_solve_kernel<3>, dim3(1), dim3(64), 0, s, (const double *)buf, st, first, tol, use_tol, err_hist_dev);
            else hipLaunchKernelGGL(dist_solve_kernel<2>, dim3(1), dim3(64), 0, s, (const double *)buf, st, first, tol, use_tol, err_hist_dev);
        }
        ICP_HIP_CHECK(hipGetLastError());
        if (!fused || it == max_iter - 1) {   // (fused: only the last iteration's error needs a pass of its own)
            if (int rc = apply_err(ctx, 1, src, tgt_rows, idx, n, dim, st->R, st->t, ss, st->sums, done, &nb)) return rc;
            tail_nb = nb;
            if (dist || it != max_iter - 1) {   // (one rank: the last iteration's error is summed by rigid_tail_kernel)
                hipLaunchKernelGGL(dist_err_kernel, dim3(1), dim3(RB), 0, s, (const double *)part, nb, st);
                ICP_HIP_CHECK(hipGetLastError());
            }
        }
    }
    if (dist) {
        hipLaunchKernelGGL(dist_last_pack_kernel, dim3(1), dim3(1), 0, s, (const RigidState *)st, buf);
        ICP_HIP_CHECK(hipGetLastError());
        if (int rc = comm_allreduce(ctx, comm, 1)) return rc;
        hipLaunchKernelGGL(dist_last_step_kernel, dim3(1), dim3(1), 0, s, (const double *)buf, st, err_hist_dev);
    } else {   // (before the un-sorting, which picks the correspondence buffer by the final iteration count)
        hipLaunchKernelGGL(rigid_tail_kernel, dim3(1), dim3(RB), 0, s, (const double *)part, tail_nb, st, err_hist_dev, dim, T_out_dev, iters_out_dev);
    }
    if (idx_out)
        hipLaunchKernelGGL(dist_unsort_idx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const int32_t *)ib[0],
                           (const int32_t *)ib[1], (const RigidState *)st, order,
                           pos ? (const int32_t *)grid.sidx : (const int32_t *)nullptr, n, idx_out);
    if (dist) hipLaunchKernelGGL(rigid_finish_kernel, dim3(1), dim3(1), 0, s, (const RigidState *)st, dim, T_out_dev, iters_out_dev);
    ICP_HIP_CHECK(hipGetLastError());
    if (timing) {
        const double t1 = now();
        (void)hipStreamSynchronize(ctx->stream);
        std::fprintf(stderr, "rigid_loop: all enqueued at %.0f us, done at %.0f us (%d iterations)\n", t1 - tt0, now() - tt0, max_iter);
    }
    return ICP_OK;
}
}  // namespace icp

extern "C" int icp_rigid_run(icp_ctx *ctx, const double *src, int64_t n, const double *tgt, int64_t m, int dim,
                             double tol, int max_iter, int use_tol, unsigned flags, double *T_out_dev,
                             double *err_hist_dev, int32_t *iters_out_dev, int32_t *idx_out) {
    ICP_COMMON_ARGS(ctx, dim);
    ICP_REQUIRE(src && tgt && T_out_dev, "null pointer");
    ICP_REQUIRE(n > 0 && m > 0 && m < 0x7fffffff, "need n > 0 and 0 < m < 2^31");
    ICP_REQUIRE(max_iter > 0, "run = 0 leaves R undefined (icp_3d.py:118-120)");
    return rigid_loop(ctx, src, n, tgt, m, dim, tol, max_iter, use_tol, flags, T_out_dev, err_hist_dev, iters_out_dev, idx_out,
                      nullptr);
}

extern "C" int icp_rigid_run_dist(icp_ctx *ctx, const double *src, int64_t n, const double *tgt, int64_t m, int dim, double tol,
                                  int max_iter, int use_tol, unsigned flags, double *T_out_dev, double *err_hist_dev,
                                  int32_t *iters_out_dev, int32_t *idx_out, const icp_comm *comm) {
    ICP_COMMON_ARGS(ctx, dim);
    ICP_REQUIRE(src && tgt && T_out_dev && comm, "null pointer");
    ICP_REQUIRE(n > 0 && m > 0 && m < 0x7fffffff, "need n > 0 (every rank holds rows) and 0 < m < 2^31");
    ICP_REQUIRE(max_iter > 0, "run = 0 leaves R undefined (icp_3d.py:118-120)");
    ICP_REQUIRE(comm->world >= 1 && comm->rank >= 0 && comm->rank < comm->world, "bad rank / world");
    ICP_REQUIRE(!icp::comm_active(comm) || (comm->reduce_buf_dev && (comm->allreduce_sum || comm->rccl)), "comm buffer / callback missing");
    return rigid_loop(ctx, src, n, tgt, m, dim, tol, max_iter, use_tol, flags, T_out_dev, err_hist_dev, iters_out_dev, idx_out, comm);
}

extern "C" int icp_bench_grid_query(icp_ctx *ctx, const double *src, int64_t n, const double *tgt, int64_t m, int dim,
                                    int reps, double *ms_out_dev) {
    ICP_COMMON_ARGS(ctx, dim);
    ICP_REQUIRE(src && tgt && ms_out_dev && n > 0 && m > 0 && reps > 0, "bad argument");
    GridBuild grid, qsort;
    if (int rc = grid_build(ctx, tgt, m, dim, grid)) return rc;
    if (!grid.usable) {
        set_error("icp_bench_grid_query: target has non-finite coordinates");
        return ICP_ERR_UNSUPPORTED;
    }
    if (int rc = grid_build(ctx, src, n, dim, qsort, 1)) return rc;
    const double *q = qsort.usable ? qsort.spts : src;
    if (int rc = ensure(ctx, ctx->work_d, (size_t)n * sizeof(int32_t) * 2)) return rc;
    int32_t *idx = (int32_t *)ctx->work_d.ptr;
    hipStream_t s = ctx->stream;
    if (int rc = grid_query(ctx, grid, q, n, nullptr, false, nullptr, idx, nullptr, nullptr)) return rc;  // warm
    // whole search (both launches) back to back
    hipEvent_t e0, e1, ev[3];
    ICP_HIP_CHECK(hipEventCreate(&e0));
    ICP_HIP_CHECK(hipEventCreate(&e1));
    ICP_HIP_CHECK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r)
        if (int rc = grid_query(ctx, grid, q, n, nullptr, false, nullptr, idx, nullptr, nullptr)) return rc;
    ICP_HIP_CHECK(hipEventRecord(e1, s));
    ICP_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    ICP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    // the two kernels of one search separately (events between the launches)
    double ms_lane = 0.0, ms_defer = 0.0;
    for (int k = 0; k < 3; ++k) ICP_HIP_CHECK(hipEventCreate(&ev[k]));
    for (int r = 0; r < reps; ++r) {
        g_grid_events = ev;
        const int rc = grid_query(ctx, grid, q, n, nullptr, false, nullptr, idx, nullptr, nullptr);
        g_grid_events = nullptr;
        if (rc) return rc;
        ICP_HIP_CHECK(hipEventSynchronize(ev[2]));
        float a = 0.f, b2 = 0.f;
        ICP_HIP_CHECK(hipEventElapsedTime(&a, ev[0], ev[1]));
        ICP_HIP_CHECK(hipEventElapsedTime(&b2, ev[1], ev[2]));
        ms_lane += a;
        ms_defer += b2;
    }
    for (int k = 0; k < 3; ++k) (void)hipEventDestroy(ev[k]);
    if (const char *dpath = getenv("ICP_GRID_DEBUG")) {
        long long *d = nullptr;
        ICP_HIP_CHECK(hipMalloc((void **)&d, (size_t)n * 3 * sizeof(long long)));
        g_grid_dbg = d;
        int rc = grid_query(ctx, grid, q, n, nullptr, false, nullptr, idx, nullptr, nullptr);
        g_grid_dbg = nullptr;
        if (rc) return rc;
        std::vector<long long> h((size_t)n * 3);
        ICP_HIP_CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        (void)hipFree(d);
        if (FILE *f = fopen(dpath, "wb")) {
            fwrite(h.data(), sizeof(long long), h.size(), f);
            fclose(f);
        }
    }
    // the same search as iterations 2.. of the rigid loop run it: with the previous answer as the hint (grid_pool_kernel)
    double ms_h = 0.0, ms_hlane = 0.0, ms_hdefer = 0.0;
    {
        int32_t *idx2 = idx + n;   // (correspondences as sorted positions, as the loop keeps them between iterations)
        if (int rc = grid_query(ctx, grid, q, n, nullptr, false, nullptr, idx, nullptr, nullptr, nullptr, true)) return rc;
        if (int rc = grid_query(ctx, grid, q, n, nullptr, false, nullptr, idx2, nullptr, nullptr, idx, true)) return rc;  // warm
        hipEvent_t h0, h1;
        ICP_HIP_CHECK(hipEventCreate(&h0));
        ICP_HIP_CHECK(hipEventCreate(&h1));
        ICP_HIP_CHECK(hipEventRecord(h0, s));
        for (int r = 0; r < reps; ++r)
            if (int rc = grid_query(ctx, grid, q, n, nullptr, false, nullptr, idx2, nullptr, nullptr, idx, true)) return rc;
        ICP_HIP_CHECK(hipEventRecord(h1, s));
        ICP_HIP_CHECK(hipEventSynchronize(h1));
        float t = 0.f;
        ICP_HIP_CHECK(hipEventElapsedTime(&t, h0, h1));
        ms_h = t;
        (void)hipEventDestroy(h0);
        (void)hipEventDestroy(h1);
        for (int k = 0; k < 3; ++k) ICP_HIP_CHECK(hipEventCreate(&ev[k]));
        for (int r = 0; r < reps; ++r) {
            g_grid_events = ev;
            const int rc = grid_query(ctx, grid, q, n, nullptr, false, nullptr, idx2, nullptr, nullptr, idx, true);
            g_grid_events = nullptr;
            if (rc) return rc;
            ICP_HIP_CHECK(hipEventSynchronize(ev[2]));
            float a = 0.f, b2 = 0.f;
            ICP_HIP_CHECK(hipEventElapsedTime(&a, ev[0], ev[1]));
            ICP_HIP_CHECK(hipEventElapsedTime(&b2, ev[1], ev[2]));
            ms_hlane += a;
            ms_hdefer += b2;
        }
        for (int k = 0; k < 3; ++k) (void)hipEventDestroy(ev[k]);
    }
    const double avg[6] = {(double)ms / reps, ms_lane / reps, ms_defer / reps, ms_h / reps, ms_hlane / reps, ms_hdefer / reps};
    ICP_HIP_CHECK(hipMemcpyAsync(ms_out_dev, avg, sizeof(avg), hipMemcpyHostToDevice, s));
    ICP_HIP_CHECK(hipStreamSynchronize(s));
    return ICP_OK;
}
