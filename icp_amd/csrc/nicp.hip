// Non-rigid ICP (embedded deformation) on the GPU: KNN influence weights, Deform,
// residual, matrix-free Jacobian products and a Levenberg-Marquardt solver whose
// linear systems (J^T J + lambda D^2) delta = -J^T r are solved by preconditioned
// conjugate gradients.
//
// Parameter vector P (nicp_3d.py:154-156,171-175; 2-D: nicp_row.py:98,118-119):
//   per vertex j: A_j (DIMxDIM row-major), b_j (DIM)          -> S = DIM*DIM+DIM numbers
//   tail: Euler angles (3, or 1 in 2-D) then translation (DIM) -> NG numbers
// Residual rows (nicp_3d.py:239-267): [c_fit*(nn - Deform) (DIM*N) | c_rigid*e_rigid
// (NRIG*N) | c_smooth*e_smooth (2*DIM*E)].
//
// The reference's dense N x N weight matrix w (4 non-zeros per column) is stored as
// nbr[N,K] / wt[N,K]: w[nbr[i,k], i] = wt[i,k]; its dense (9N+6E) x (12N+6) Jacobian is
// never formed: J p and J^T u are gathered per vertex / per edge (reverse adjacency in
// CSR form, built once per mesh), so every product is deterministic (no atomics).
//
// NICP_REFERENCE_COMPAT reproduces the reference's latent quirks:
//   Q5 (3-D) rigid residual of vertex i written at slot 3i (the Jacobian rows stay at 6i),
//   Q6 global-rotation Jacobian columns use the already transformed points,
//   Q7 (2-D) b-columns of the fit block carry no weight factor.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

struct nicp_graph {
    int64_t n = 0, e = 0;
    int dim = 3, k = 4;
    int vb = 256;                // threads per block of the per-vertex kernels (64 on small meshes: more CUs busy)
    int32_t *nbr = nullptr;      // [n,k]
    double *wt = nullptr;        // [n,k]
    int32_t *edges = nullptr;    // [e,2]
    int32_t *rev_ptr = nullptr;  // [n+1]  reverse KNN adjacency: entries i*k+kk with nbr[i,kk]==j
    int32_t *rev_ent = nullptr;  // [n*k]
    int32_t *inc_ptr = nullptr;  // [n+1]  incident edges: entries e*2+end
    int32_t *inc_ent = nullptr;  // [2e]
    // the same adjacencies flattened for the J^T gather (one dependent load less per entry)
    int32_t *rev_i = nullptr;      // [n*k] row vertex i of entry rev_ent[.]
    double *rev_w = nullptr;       // [n*k] its weight wt[rev_ent[.]]
    int32_t *inc_other = nullptr;  // [2e]  the edge's other endpoint
    // optional row multipliers and landmark rows (nicp_set_weights / nicp_set_landmarks)
    double *fw = nullptr, *gw = nullptr, *ew = nullptr;  // [n] fit, [n] rigidity, [e] smoothness
    int64_t nl = 0;
    int32_t *lm_v = nullptr;     // [nl] landmark vertex
    double *lm_pos = nullptr;    // [nl,dim] landmark position
    int32_t *lm_of = nullptr;    // [n] landmark row of a vertex, or -1
    double c_land = 0.0;
    double *f2 = nullptr;        // [n] fit_w^2 + (c_land/c_fit)^2 [landmark vertices]: fit block of J^T W^2 J
    // solver workspace (grow-once)
    double *ws = nullptr;
    size_t ws_doubles = 0;
};

namespace icp {

constexpr int NB = 256;
constexpr int KNN_TILE = 512;
// CG iterations between two host reads of the stop flag (= one hipGraph replay).  Even: the fused CG kernels alternate
// a parity that must restart at 0 with every chunk.  20: in the reference's mode most LM trials are rejected and their
// CG solves take 30-100 iterations, so what is enqueued behind the converged iteration is mostly waste even though
// those kernels return at once (C3 reference mode: chunks of 10 / 20 / 30 / 50 / 100 -> 0.40 / 0.385 / 0.40 / 0.43 /
// 0.52 s; the long solves of the self-consistent mode do not care: 15.83 vs 15.84 s).  ICP_NICP_CHUNK overrides.
static int cg_chunk_size() {
    static const int v = [] {
        const char *e = getenv("ICP_NICP_CHUNK");
        const int c = e ? atoi(e) : 20;
        return c >= 2 ? (c & ~1) : 20;
    }();
    return v;
}

template <int DIM>
struct Lay {
    static constexpr int NA = DIM * DIM;
    static constexpr int S = DIM * DIM + DIM;
    static constexpr int NANG = DIM == 3 ? 3 : 1;
    static constexpr int NG = NANG + DIM;
    static constexpr int NRIG = DIM == 3 ? 6 : 3;
};

// rotation + derivatives + translation, rebuilt from the tail of P
struct Globals {
    double R[9];
    double dR[3][9];
    double T[3];
};

// Where element (vertex j, component e) of a per-vertex array lives.  The C-ABI hands arrays in the
// reference's layouts (P: 12 numbers per vertex in a row, points [n,dim], residual rows per vertex /
// per edge); with one lane per vertex those are 96-byte-strided accesses that thrash the L1, so the
// solver keeps its own copies lane-transposed (component-major: element e of vertex j at e*n + j) and
// the same kernels serve both through these strides.  The tail (global parameters) sits at S*n in both.
struct Strides {
    int64_t pj, pe;  // parameter-like vectors: P, p, A p, gradient, scaling ...
    int64_t vj, vc;  // per-vertex points: v, nn, Deform output
    int64_t fj, fo;  // fit rows        (dim per vertex)
    int64_t rj, rm;  // rigidity rows   (NRIG per vertex), relative to the start of the rigid block
    int64_t sj, so;  // smoothness rows (2*dim per edge), relative to the start of the smooth block
    int64_t lj, lo;  // landmark rows   (dim per landmark), relative to the start of the landmark block
};
// Optional per-row multipliers and landmark rows; every pointer may be null (= the reference's energy).
struct RowW {
    const double *fit = nullptr, *rig = nullptr, *edge = nullptr;  // linear multipliers [n], [n], [e]
    const double *f2 = nullptr;      // [n] squared fit multiplier incl. the landmark rows (see nicp_graph::f2)
    const int32_t *lm_of = nullptr;  // [n] landmark row of a vertex or -1
    const double *lm_pos = nullptr;  // [nl,dim]
    double c_land = 0.0;
    int64_t nl = 0;
};
template <int DIM>
static inline Strides strides_aos() {
    return {Lay<DIM>::S, 1, DIM, 1, DIM, 1, Lay<DIM>::NRIG, 1, 2 * DIM, 1, DIM, 1};
}
static inline Strides strides_soa(int64_t n, int64_t ne, int64_t nl) {
    return {1, n, 1, n, 1, n, 1, n, 1, ne > 0 ? ne : 1, 1, nl > 0 ? nl : 1};
}

// ------------------------------------------------------------------ KNN + weights
// Exact k-nearest by the key the reference sorts on, sqrt(d2) (nicp_3d.py:53-57), with
// ties resolved to the lower index (the C++ heap's rule, nicp3d.cpp:116-125, and what a
// stable sort gives).  sqrt is only evaluated for candidates that can enter the list:
// d2 > (largest d2 whose rounded root equals the current k-th root) is rejected first.
template <int DIM, int K>
__global__ __launch_bounds__(NB) void knn_rows_kernel(const double *__restrict__ v, int64_t n,
                                                     const double *__restrict__ q, int64_t nq,
                                                     int32_t *__restrict__ idx_out, double *__restrict__ dist_out) {
    __shared__ double tile[KNN_TILE * DIM];
    const int tid = threadIdx.x;
    const int64_t qi = (int64_t)blockIdx.x * NB + tid;
    const int64_t qq = qi < nq ? qi : nq - 1;
    double s[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) s[c] = q[qq * DIM + c];
    double key[K];
    int32_t id[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        key[k] = __builtin_inf();
        id[k] = 0x7fffffff;
    }
    double gate = __builtin_inf();  // reject d2 > gate without a sqrt
    for (int64_t base = 0; base < n; base += KNN_TILE) {
        const int cnt = (n - base) < KNN_TILE ? (int)(n - base) : KNN_TILE;
        __syncthreads();
        for (int e = tid; e < cnt * DIM; e += NB) tile[e] = v[base * DIM + e];
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            double d2 = 0.0;
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                const double d = tile[j * DIM + c] - s[c];
                d2 += d * d;
            }
            if (!(d2 <= gate)) continue;
            const double r = __dsqrt_rn(d2);
            const int32_t jj = (int32_t)(base + j);
            if (!(r < key[K - 1])) continue;  // ascending scan: an equal root with a higher index loses
            // insert (r, jj) keeping (key, id) sorted ascending, stable
            double ck = r;
            int32_t ci = jj;
            bool shifting = false;  // once placed, everything behind moves down one slot
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const bool sw = shifting || ck < key[k];
                shifting = sw;
                const double tk = sw ? key[k] : ck;
                const int32_t ti = sw ? id[k] : ci;
                key[k] = sw ? ck : key[k];
                id[k] = sw ? ci : id[k];
                ck = tk;
                ci = ti;
            }
            // new gate: sqrt_rn(fl(kr*kr)) == kr, and sqrt is monotone, so every d2 whose root is
            // below kr is below fl(kr*kr) (guarded against under/overflow of the square)
            const double kr = key[K - 1];
            gate = (kr > 1e-150 && kr < 1e150) ? kr * kr : __builtin_inf();
        }
    }
    if (qi < nq) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            idx_out[qi * K + k] = id[k] == 0x7fffffff ? -1 : id[k];  // unfilled slot (fewer than K comparable points)
            dist_out[qi * K + k] = key[k];
        }
    }
}

// weights from the K+2 nearest (nicp_3d.py:160-164 / nicp3d.cpp:208-214)
template <int NN>
__global__ void knn_weights_kernel(const int32_t *__restrict__ kidx, const double *__restrict__ kdist, int64_t n,
                                   int32_t *__restrict__ nbr, double *__restrict__ wt) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    constexpr int K = NN + 2;
    const double dmax = kdist[j * K + K - 1];
    double sum = 0.0;
    for (int i = 1; i <= NN; ++i) sum += kdist[j * K + i];
    const double denom = (double)NN - sum / dmax;
    for (int i = 1; i <= NN; ++i) {
        nbr[j * NN + i - 1] = kidx[j * K + i];
        wt[j * NN + i - 1] = (1.0 - kdist[j * K + i] / dmax) / denom;
    }
}

// ------------------------------------------------------------------ globals
template <int DIM>
__global__ void globals_kernel(const double *__restrict__ P, int64_t n, Globals *__restrict__ g) {
    if (threadIdx.x || blockIdx.x) return;
    const double *tail = P + n * Lay<DIM>::S;
    for (int k = 0; k < 9; ++k) {
        g->R[k] = 0;
        g->dR[0][k] = g->dR[1][k] = g->dR[2][k] = 0;
    }
    g->T[0] = g->T[1] = g->T[2] = 0;
    if (DIM == 3) {
        // nicp_3d.py:180-215 (rotation sequence X, Y, Z; row-vector convention)
        const double sx = sin(tail[0]), cx = cos(tail[0]), sy = sin(tail[1]), cy = cos(tail[1]);
        const double sz = sin(tail[2]), cz = cos(tail[2]);
        const double R[9] = {cy * cz, cy * sz, -sy, sx * sy * cz - cx * sz, sx * sy * sz + cx * cz, sx * cy,
                             cx * sy * cz + sx * sz, cx * sy * sz - sx * cz, cx * cy};
        const double dx[9] = {0, 0, 0, cx * sy * cz + sx * sz, cx * sy * sz - sx * cz, cx * cy,
                              -sx * sy * cz + cx * sz, -sx * sy * sz - cx * cz, -sx * cy};
        const double dy[9] = {-sy * cz, -sy * sz, -cy, sx * cy * cz, sx * cy * sz, -sx * sy,
                              cx * cy * cz, cx * cy * sz, -cx * sy};
        const double dz[9] = {-cy * sz, cy * cz, 0, -sx * sy * sz - cx * cz, sx * sy * cz - cx * sz, 0,
                              -cx * sy * sz + sx * cz, cx * sy * cz + sx * sz, 0};
        for (int k = 0; k < 9; ++k) {
            g->R[k] = R[k];
            g->dR[0][k] = dx[k];
            g->dR[1][k] = dy[k];
            g->dR[2][k] = dz[k];
        }
        for (int c = 0; c < 3; ++c) g->T[c] = tail[3 + c];
    } else {
        // nicp_row.py:124,175
        const double s = sin(tail[0]), c = cos(tail[0]);
        g->R[0] = c;
        g->R[1] = s;
        g->R[2] = -s;
        g->R[3] = c;
        g->dR[0][0] = -s;
        g->dR[0][1] = c;
        g->dR[0][2] = -c;
        g->dR[0][3] = -s;
        g->T[0] = tail[1];
        g->T[1] = tail[2];
    }
}

// ------------------------------------------------------------------ Deform
// pre_i = sum_k w_ik ((v_i - v_j) A_j + v_j + b_j), j = nbr[i,k];  dd_i = pre_i R + T
template <int DIM>
__global__ __launch_bounds__(NB) void deform_kernel(const double *__restrict__ P, const int32_t *__restrict__ nbr,
                                                   const double *__restrict__ wt, int K, const double *__restrict__ v,
                                                   int64_t n, const Globals *__restrict__ g, double *__restrict__ pre_out,
                                                   double *__restrict__ dd_out, Strides st) {
    const int64_t i = (int64_t)blockIdx.x * NB + threadIdx.x;
    if (i >= n) return;
    double vi[DIM], pre[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
        vi[c] = v[i * st.vj + c * st.vc];
        pre[c] = 0.0;
    }
    for (int k = 0; k < K; ++k) {
        const double w = wt[i * K + k];
        if (w == 0.0) continue;  // nicp_3d.py:227
        const int64_t j = nbr[i * K + k];
        const double *Pj = P + j * st.pj;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < DIM; ++r) a += (vi[r] - v[j * st.vj + r * st.vc]) * Pj[(r * DIM + c) * st.pe];
            pre[c] += w * (a + v[j * st.vj + c * st.vc] + Pj[(DIM * DIM + c) * st.pe]);
        }
    }
#pragma unroll
    for (int o = 0; o < DIM; ++o) {
        double a = 0.0;
#pragma unroll
        for (int c = 0; c < DIM; ++c) a += pre[c] * g->R[c * DIM + o];
        if (dd_out) dd_out[i * st.vj + o * st.vc] = a + g->T[o];
        if (pre_out) pre_out[i * st.vj + o * st.vc] = pre[o];
    }
}

// ------------------------------------------------------------------ residual
template <int DIM>
__device__ __forceinline__ void rigid_terms(const double *A, double *out) {
    if (DIM == 3) {
        // nicp_3d.py:248-253
        out[0] = A[0] * A[3] + A[1] * A[4] + A[2] * A[5];
        out[1] = A[3] * A[6] + A[4] * A[7] + A[5] * A[8];
        out[2] = A[6] * A[0] + A[7] * A[1] + A[8] * A[2];
        out[3] = 1 - A[0] * A[0] - A[1] * A[1] - A[2] * A[2];
        out[4] = 1 - A[3] * A[3] - A[4] * A[4] - A[5] * A[5];
        out[5] = 1 - A[6] * A[6] - A[7] * A[7] - A[8] * A[8];
    } else {
        // nicp_row.py:142-144
        out[0] = A[0] * A[2] + A[1] * A[3];
        out[1] = 1 - A[0] * A[0] - A[1] * A[1];
        out[2] = 1 - A[2] * A[2] - A[3] * A[3];
    }
}

template <int DIM, bool COMPAT>
__global__ __launch_bounds__(NB) void residual_kernel(const double *__restrict__ P, const double *__restrict__ v,
                                                     const double *__restrict__ nn, const double *__restrict__ dd,
                                                     const int32_t *__restrict__ edges, int64_t n, int64_t ne,
                                                     double c_fit, double c_rigid, double c_smooth,
                                                     double *__restrict__ r, Strides st, RowW rw) {
    constexpr int NRIG = Lay<DIM>::NRIG, NA = Lay<DIM>::NA;
    const int64_t t = (int64_t)blockIdx.x * NB + threadIdx.x;
    double *r_rig = r + n * DIM, *r_sm = r + n * DIM + n * NRIG, *r_land = r_sm + 2 * DIM * ne;
    if (t < n) {
        if (rw.fit) c_fit *= rw.fit[t];
        if (rw.rig) c_rigid *= rw.rig[t];
        const int32_t lk = rw.lm_of ? rw.lm_of[t] : -1;
        if (lk >= 0)
#pragma unroll
            for (int c = 0; c < DIM; ++c)
                r_land[lk * st.lj + c * st.lo] = rw.c_land * (rw.lm_pos[(int64_t)lk * DIM + c] - dd[t * st.vj + c * st.vc]);
    } else if (t < n + ne && rw.edge) {
        c_smooth *= rw.edge[t - n];
    }
    auto rig = [&](int64_t flat) { return (flat / NRIG) * st.rj + (flat % NRIG) * st.rm; };  // flat = reference slot
    if (t < n) {
#pragma unroll
        for (int c = 0; c < DIM; ++c)
            r[t * st.fj + c * st.fo] = c_fit * (nn[t * st.vj + c * st.vc] - dd[t * st.vj + c * st.vc]);
        double A[NA], e[NRIG];
#pragma unroll
        for (int k = 0; k < NA; ++k) A[k] = P[t * st.pj + k * st.pe];
        rigid_terms<DIM>(A, e);
        if (DIM == 3 && COMPAT) {
            // Q5: slot 3i..3i+5 written in ascending i: only the three orthogonality terms of
            // each vertex survive, except the last vertex which keeps all six
            for (int m = 0; m < 3; ++m) r_rig[rig(t * 3 + m)] = c_rigid * e[m];
            if (t == n - 1)
                for (int m = 3; m < 6; ++m) r_rig[rig(t * 3 + m)] = c_rigid * e[m];
            for (int64_t sl = 3 * n + 3 + t * 3; sl < 3 * n + 3 + t * 3 + 3 && sl < 6 * n; ++sl) r_rig[rig(sl)] = 0.0;
        } else {
            for (int m = 0; m < NRIG; ++m) r_rig[t * st.rj + m * st.rm] = c_rigid * e[m];
        }
    } else if (t < n + ne) {
        const int64_t ed = t - n;
        const int64_t p1 = edges[ed * 2], p2 = edges[ed * 2 + 1];
        const double *P1 = P + p1 * st.pj, *P2 = P + p2 * st.pj;
        // nicp_3d.py:261-262
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            double a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int rr = 0; rr < DIM; ++rr) {
                const double d = v[p2 * st.vj + rr * st.vc] - v[p1 * st.vj + rr * st.vc];
                a1 += d * P1[(rr * DIM + c) * st.pe];
                a2 += -d * P2[(rr * DIM + c) * st.pe];
            }
            const double x1 = v[p1 * st.vj + c * st.vc] + P1[(DIM * DIM + c) * st.pe];
            const double x2 = v[p2 * st.vj + c * st.vc] + P2[(DIM * DIM + c) * st.pe];
            r_sm[ed * st.sj + c * st.so] = c_smooth * (a1 + x1 - x2);
            r_sm[ed * st.sj + (DIM + c) * st.so] = c_smooth * (a2 + x2 - x1);
        }
    }
}

// XCD-aware block order of the per-vertex CG kernels.  Workgroups are dealt round-robin over the 8 XCDs, each with its own
// L2, and the vertices are in Morton order: with the plain order the blocks that share an L2 are 8 apart, so every
// neighbour gather that crosses a block boundary (p, u of adjacent vertices, edge rows) lands in ANOTHER XCD's L2 and is
// fetched again.  The launch is cut into super-chunks of 8 * R blocks in which XCD k takes the k-th run of R consecutive
// logical blocks.  Only which physical block does which logical block changes: rows, partial slots and all results are
// bit-identical.  R = 16 measured per CG iteration: 10 k vertices 43.4 -> 41.1 us, 65 k 82.3 -> 74.5, 200 k 225 -> 206,
// 640 k 675 -> 628 (R = 8 ... 32 within 1 %; contiguous EIGHTHS of the mesh per XCD were slower than the plain order,
// 229 -> 252 us at 200 k: the XCDs then stop sharing the Infinity Cache).  ICP_NICP_RUN=0 restores the plain order.
__device__ int g_nicp_run = 16;
__device__ __forceinline__ unsigned vblock() {
    const int R = g_nicp_run;
    if (R <= 0) return blockIdx.x;
    const unsigned per = 8u * (unsigned)R, sc = blockIdx.x / per;
    if ((sc + 1u) * per > gridDim.x) return blockIdx.x;   // ragged last super-chunk: plain order
    const unsigned r = blockIdx.x - sc * per;
    return sc * per + (r & 7u) * (unsigned)R + (r >> 3);
}

// ------------------------------------------------------------------ y = J p
template <int DIM, bool COMPAT>
__global__ __launch_bounds__(NB) void jp_kernel(const double *__restrict__ P, const double *__restrict__ p,
                                               const int32_t *__restrict__ nbr, const double *__restrict__ wt, int K,
                                               const double *__restrict__ v, const double *__restrict__ rs,
                                               const int32_t *__restrict__ edges, int64_t n, int64_t ne,
                                               const Globals *__restrict__ g, double c_fit, double c_rigid,
                                               double c_smooth, double *__restrict__ y, Strides st, RowW rw, int sq) {
    // sq = 0: y = W J p (rows as nicp_residual, landmark rows included); sq = 1: y = W^2 J p with the
    // landmark rows folded into the fit rows (rw.f2), which is what J^T needs next in the CG operator
    constexpr int S = Lay<DIM>::S, NRIG = Lay<DIM>::NRIG, NANG = Lay<DIM>::NANG, NA = Lay<DIM>::NA;
    const int64_t t = (int64_t)vblock() * blockDim.x + threadIdx.x;
    double *y_rig = y + n * DIM, *y_sm = y + n * DIM + n * NRIG, *y_land = y_sm + 2 * DIM * ne;
    const double *ptail = p + n * S;
    double c_land_row = 0.0;
    int32_t lk = -1;
    if (t < n) {
        if (sq) {
            if (rw.f2) c_fit *= rw.f2[t];
            if (rw.rig) c_rigid *= rw.rig[t] * rw.rig[t];
        } else {
            if (rw.fit) c_fit *= rw.fit[t];
            if (rw.rig) c_rigid *= rw.rig[t];
            lk = rw.lm_of ? rw.lm_of[t] : -1;
            c_land_row = rw.c_land;
        }
    } else if (t < n + ne && rw.edge) {
        const double e = rw.edge[t - n];
        c_smooth *= sq ? e * e : e;
    }
    if (t < n) {
        // fit rows: nicp_3d.py:286-306
        double vi[DIM], acc[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            vi[c] = v[t * st.vj + c * st.vc];
            acc[c] = 0.0;
        }
        for (int k0 = 0; k0 < K; k0 += 4) {  // four neighbours per batch: all gathers issued before use
            int32_t jj[4];
            double ww[4], pj[4][S], vn[4][DIM];
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
                const int kk = k0 + b4 < K ? k0 + b4 : K - 1;
                jj[b4] = nbr[t * K + kk];
                ww[b4] = k0 + b4 < K ? wt[t * K + kk] : 0.0;
            }
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
#pragma unroll
                for (int e = 0; e < S; ++e) pj[b4][e] = p[(int64_t)jj[b4] * st.pj + e * st.pe];
#pragma unroll
                for (int r = 0; r < DIM; ++r) vn[b4][r] = v[(int64_t)jj[b4] * st.vj + r * st.vc];
            }
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
                const double w = ww[b4];
                if (w == 0.0) continue;
                const double wb = (DIM == 2 && COMPAT) ? 1.0 : w;  // Q7
#pragma unroll
                for (int c = 0; c < DIM; ++c) {
                    double a = 0.0;
#pragma unroll
                    for (int r = 0; r < DIM; ++r) a += (vi[r] - vn[b4][r]) * pj[b4][r * DIM + c];
                    acc[c] += w * a + wb * pj[b4][DIM * DIM + c];
                }
            }
        }
#pragma unroll
        for (int o = 0; o < DIM; ++o) {
            double a = ptail[NANG + o];
#pragma unroll
            for (int c = 0; c < DIM; ++c) a += acc[c] * g->R[c * DIM + o];
            for (int an = 0; an < NANG; ++an) {
                double b = 0.0;
#pragma unroll
                for (int c = 0; c < DIM; ++c) b += rs[t * st.vj + c * st.vc] * g->dR[an][c * DIM + o];
                a += ptail[an] * b;
            }
            y[t * st.fj + o * st.fo] = -c_fit * a;
            if (lk >= 0) y_land[lk * st.lj + o * st.lo] = -c_land_row * a;
        }
        // rigid rows: nicp_3d.py:309-314 / nicp_row.py:193-204
        double A[NA], pa[NA];
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            A[k] = P[t * st.pj + k * st.pe];
            pa[k] = p[t * st.pj + k * st.pe];
        }
        if (DIM == 3) {
            double d01 = 0, d10 = 0, d12 = 0, d21 = 0, d20 = 0, d02 = 0, d00 = 0, d11 = 0, d22 = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                d10 += A[3 + c] * pa[c];       // a1 . pa0
                d01 += A[c] * pa[3 + c];       // a0 . pa1
                d21 += A[6 + c] * pa[3 + c];   // a2 . pa1
                d12 += A[3 + c] * pa[6 + c];   // a1 . pa2
                d20 += A[6 + c] * pa[c];       // a2 . pa0
                d02 += A[c] * pa[6 + c];       // a0 . pa2
                d00 += A[c] * pa[c];
                d11 += A[3 + c] * pa[3 + c];
                d22 += A[6 + c] * pa[6 + c];
            }
            y_rig[t * st.rj + 0 * st.rm] = c_rigid * (d10 + d01);
            y_rig[t * st.rj + 1 * st.rm] = c_rigid * (d21 + d12);
            y_rig[t * st.rj + 2 * st.rm] = c_rigid * (d20 + d02);
            y_rig[t * st.rj + 3 * st.rm] = c_rigid * (-2 * d00);
            y_rig[t * st.rj + 4 * st.rm] = c_rigid * (-2 * d11);
            y_rig[t * st.rj + 5 * st.rm] = c_rigid * (-2 * d22);
        } else {
            double d10 = 0, d01 = 0, d00 = 0, d11 = 0;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                d10 += A[2 + c] * pa[c];
                d01 += A[c] * pa[2 + c];
                d00 += A[c] * pa[c];
                d11 += A[2 + c] * pa[2 + c];
            }
            y_rig[t * st.rj + 0 * st.rm] = c_rigid * (d10 + d01);
            y_rig[t * st.rj + 1 * st.rm] = c_rigid * (-2 * d00);
            y_rig[t * st.rj + 2 * st.rm] = c_rigid * (-2 * d11);
        }
    } else if (t < n + ne) {
        // smooth rows: nicp_3d.py:316-333
        const int64_t ed = t - n;
        const int64_t p1 = edges[ed * 2], p2 = edges[ed * 2 + 1];
        const double *q1 = p + p1 * st.pj, *q2 = p + p2 * st.pj;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            double a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int r = 0; r < DIM; ++r) {
                const double d = v[p2 * st.vj + r * st.vc] - v[p1 * st.vj + r * st.vc];
                a1 += d * q1[(r * DIM + c) * st.pe];
                a2 += d * q2[(r * DIM + c) * st.pe];
            }
            const double b1 = q1[(DIM * DIM + c) * st.pe], b2 = q2[(DIM * DIM + c) * st.pe];
            y_sm[ed * st.sj + c * st.so] = c_smooth * (a1 + b1 - b2);
            y_sm[ed * st.sj + (DIM + c) * st.so] = c_smooth * (-a2 + b2 - b1);
        }
    }
}

// ------------------------------------------------------------------ out = J^T u (+ lam * d2 .* p)
// SQUARE = true computes the column sums of squares of J (diag of J^T J) instead.
template <int DIM, bool COMPAT, bool SQUARE>
__global__ __launch_bounds__(NB) void jtu_kernel(const double *__restrict__ P, const double *__restrict__ u,
                                                const int32_t *__restrict__ nbr, const double *__restrict__ wt, int K,
                                                const double *__restrict__ v, const double *__restrict__ rs,
                                                const int32_t *__restrict__ edges, const int32_t *__restrict__ rev_ptr,
                                                const int32_t *__restrict__ rev_i, const double *__restrict__ rev_w,
                                                const int32_t *__restrict__ inc_ptr, const int32_t *__restrict__ inc_ent,
                                                const int32_t *__restrict__ inc_other, int64_t n, int64_t ne,
                                                const Globals *__restrict__ g, double c_fit, double c_rigid,
                                                double c_smooth, const double *__restrict__ lam_d2,
                                                const double *__restrict__ p_in, double lam, double *__restrict__ out,
                                                double *__restrict__ partial, int64_t nown, Strides st, RowW rw) {
    // the row multipliers are already in u (jp_kernel with sq = 1, weight_rows_kernel); only the
    // column sums of squares (SQUARE) read them here
    constexpr int NRIG = Lay<DIM>::NRIG, NANG = Lay<DIM>::NANG, NG = Lay<DIM>::NG;
    const int64_t j = (int64_t)vblock() * blockDim.x + threadIdx.x;
    const double *u_rig = u + n * DIM, *u_sm = u + n * DIM + n * NRIG;
    double gacc[kSumSlots];
#pragma unroll
    for (int k = 0; k < kSumSlots; ++k) gacc[k] = 0.0;
    if (j < n) {
        double gA[DIM][DIM], gb[DIM], vj[DIM];
#pragma unroll
        for (int r = 0; r < DIM; ++r) {
            vj[r] = v[j * st.vj + r * st.vc];
            gb[r] = 0.0;
#pragma unroll
            for (int c = 0; c < DIM; ++c) gA[r][c] = 0.0;
        }
        // The gathers of this kernel form a chain  ptr -> entry -> u / v  per adjacency list.  To pay it
        // once, not once per entry, entries are taken in batches with every load of a batch issued
        // before its first use, and the FIRST batch of both lists (4 fit rows, 8 edges: all of them on
        // ordinary meshes) is fetched up front, side by side.  The accumulation order is unchanged
        // (fit rows ascending, rigid rows, edges ascending); padded slots add nothing.
        constexpr int RB = 4, IB = 8;
        double Rm[DIM][DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c)
#pragma unroll
            for (int o = 0; o < DIM; ++o) Rm[c][o] = g->R[c * DIM + o];
        const int32_t rv_beg = rev_ptr[j], rv_end = rev_ptr[j + 1], in_beg = inc_ptr[j], in_end = inc_ptr[j + 1];
        const int32_t rv_cl = rv_end > rv_beg ? rv_end - 1 : 0, in_cl = in_end > in_beg ? in_end - 1 : 0;

        auto fit_entries = [&](int32_t e0, int32_t (&ii)[RB], double (&ww)[RB]) {
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int32_t ee = e0 + q < rv_end ? e0 + q : rv_cl;
                ii[q] = rev_i[ee];
                ww[q] = e0 + q < rv_end ? rev_w[ee] : 0.0;
            }
        };
        auto fit_gather = [&](const int32_t (&ii)[RB], double (&uu)[RB][DIM], double (&vv)[RB][DIM]) {
#pragma unroll
            for (int q = 0; q < RB; ++q)
#pragma unroll
                for (int o = 0; o < DIM; ++o) {
                    uu[q][o] = SQUARE ? (rw.f2 ? rw.f2[ii[q]] : 1.0) : u[(int64_t)ii[q] * st.fj + o * st.fo];
                    vv[q][o] = v[(int64_t)ii[q] * st.vj + o * st.vc];
                }
        };
        auto fit_accumulate = [&](const double (&ww)[RB], const double (&uu)[RB][DIM], const double (&vv)[RB][DIM]) {
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const double w = ww[q];
                if (w == 0.0) continue;
                const double wb = (DIM == 2 && COMPAT) ? 1.0 : w;
                double yv[DIM];
#pragma unroll
                for (int c = 0; c < DIM; ++c) {
                    double a = 0.0;
#pragma unroll
                    for (int o = 0; o < DIM; ++o) a += SQUARE ? Rm[c][o] * Rm[c][o] : Rm[c][o] * uu[q][o];
                    yv[c] = SQUARE ? a * uu[q][0] : a;
                }
#pragma unroll
                for (int r = 0; r < DIM; ++r) {
                    const double d = vv[q][r] - vj[r];
#pragma unroll
                    for (int c = 0; c < DIM; ++c)
                        gA[r][c] += SQUARE ? (c_fit * w * d) * (c_fit * w * d) * yv[c] : -c_fit * w * d * yv[c];
                }
#pragma unroll
                for (int c = 0; c < DIM; ++c) gb[c] += SQUARE ? (c_fit * wb) * (c_fit * wb) * yv[c] : -c_fit * wb * yv[c];
            }
        };
        // with the other endpoint stored per entry, d = v[p2] - v[p1] is (v[other] - vj) at end 0 and (vj - v[other]) at end 1
        auto edge_entries = [&](int32_t e0, int32_t (&ent)[IB], int32_t (&oth)[IB]) {
#pragma unroll
            for (int q = 0; q < IB; ++q) {
                const int32_t ee = e0 + q < in_end ? e0 + q : in_cl;
                ent[q] = inc_ent[ee];
                oth[q] = inc_other[ee];
            }
        };
        auto edge_gather = [&](const int32_t (&ent)[IB], const int32_t (&oth)[IB], double (&us)[IB][2 * DIM], double (&vo)[IB][DIM]) {
#pragma unroll
            for (int q = 0; q < IB; ++q) {
#pragma unroll
                for (int o = 0; o < DIM; ++o) vo[q][o] = v[(int64_t)oth[q] * st.vj + o * st.vc];
#pragma unroll
                for (int o = 0; o < 2 * DIM; ++o)
                    us[q][o] = SQUARE ? (rw.edge ? rw.edge[ent[q] >> 1] * rw.edge[ent[q] >> 1] : 1.0)
                                      : u_sm[(int64_t)(ent[q] >> 1) * st.sj + o * st.so];
            }
        };
        auto edge_accumulate = [&](int32_t e0, const int32_t (&ent)[IB], const double (&us)[IB][2 * DIM], const double (&vo)[IB][DIM]) {
#pragma unroll
            for (int q = 0; q < IB; ++q) {
                if (e0 + q >= in_end) continue;
                const int end = ent[q] & 1;
#pragma unroll
                for (int c = 0; c < DIM; ++c) {
                    const double u0 = us[q][c], u1 = us[q][DIM + c];
#pragma unroll
                    for (int r = 0; r < DIM; ++r) {
                        const double d = end == 0 ? vo[q][r] - vj[r] : vj[r] - vo[q][r];
                        if (SQUARE) gA[r][c] += c_smooth * c_smooth * d * d * u0;
                        else gA[r][c] += end == 0 ? c_smooth * d * u0 : -c_smooth * d * u1;
                    }
                    if (SQUARE) gb[c] += 2 * c_smooth * c_smooth * u0;
                    else gb[c] += end == 0 ? c_smooth * (u0 - u1) : c_smooth * (u1 - u0);
                }
            }
        };

        // first batches of both lists + the vertex's own rows: entries, then every gather
        int32_t ii0[RB], ent0[IB], oth0[IB];
        double ww0[RB], uu0[RB][DIM], vv0[RB][DIM], us0[IB][2 * DIM], vo0[IB][DIM];
        fit_entries(rv_beg, ii0, ww0);
        edge_entries(in_beg, ent0, oth0);
        double Aown[Lay<DIM>::NA], urown[NRIG];
#pragma unroll
        for (int e = 0; e < Lay<DIM>::NA; ++e) Aown[e] = P[j * st.pj + e * st.pe];
#pragma unroll
        for (int e = 0; e < NRIG; ++e) urown[e] = SQUARE ? (rw.rig ? rw.rig[j] * rw.rig[j] : 1.0) : u_rig[j * st.rj + e * st.rm];
        fit_gather(ii0, uu0, vv0);
        edge_gather(ent0, oth0, us0, vo0);

        // fit rows that reference vertex j
        fit_accumulate(ww0, uu0, vv0);
        for (int32_t e0 = rv_beg + RB; e0 < rv_end; e0 += RB) {
            int32_t ii[RB];
            double ww[RB], uu[RB][DIM], vv[RB][DIM];
            fit_entries(e0, ii, ww);
            fit_gather(ii, uu, vv);
            fit_accumulate(ww, uu, vv);
        }
        // rigid rows of vertex j
        {
            const double *A = Aown, *ur = urown;
            if constexpr (DIM == 3) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double a0 = A[c], a1 = A[3 + c], a2 = A[6 + c];
                    if (SQUARE) {
                        gA[0][c] += c_rigid * c_rigid * ur[0] * (a1 * a1 + a2 * a2 + 4 * a0 * a0);
                        gA[1][c] += c_rigid * c_rigid * ur[0] * (a0 * a0 + a2 * a2 + 4 * a1 * a1);
                        gA[2][c] += c_rigid * c_rigid * ur[0] * (a1 * a1 + a0 * a0 + 4 * a2 * a2);
                    } else {
                        gA[0][c] += c_rigid * (ur[0] * a1 + ur[2] * a2 - 2 * ur[3] * a0);
                        gA[1][c] += c_rigid * (ur[0] * a0 + ur[1] * a2 - 2 * ur[4] * a1);
                        gA[2][c] += c_rigid * (ur[1] * a1 + ur[2] * a0 - 2 * ur[5] * a2);
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double a0 = A[c], a1 = A[2 + c];
                    if (SQUARE) {
                        gA[0][c] += c_rigid * c_rigid * ur[0] * (a1 * a1 + 4 * a0 * a0);
                        gA[1][c] += c_rigid * c_rigid * ur[0] * (a0 * a0 + 4 * a1 * a1);
                    } else {
                        gA[0][c] += c_rigid * (ur[0] * a1 - 2 * ur[1] * a0);
                        gA[1][c] += c_rigid * (ur[0] * a0 - 2 * ur[2] * a1);
                    }
                }
            }
        }
        // smooth rows of the edges incident to vertex j
        edge_accumulate(in_beg, ent0, us0, vo0);
        for (int32_t e0 = in_beg + IB; e0 < in_end; e0 += IB) {
            int32_t ent[IB], oth[IB];
            double us[IB][2 * DIM], vo[IB][DIM];
            edge_entries(e0, ent, oth);
            edge_gather(ent, oth, us, vo);
            edge_accumulate(e0, ent, us, vo);
        }
        double pap = 0.0;
#pragma unroll
        for (int r = 0; r < DIM; ++r)
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                const int64_t o = j * st.pj + (r * DIM + c) * st.pe;
                const double val = gA[r][c] + (lam_d2 ? lam * lam_d2[o] * p_in[o] : 0.0);
                out[o] = val;
                if (lam_d2) pap += val * p_in[o];
            }
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            const int64_t o = j * st.pj + (DIM * DIM + c) * st.pe;
            const double val = gb[c] + (lam_d2 ? lam * lam_d2[o] * p_in[o] : 0.0);
            out[o] = val;
            if (lam_d2) pap += val * p_in[o];
        }
        gacc[8] = pap;  // vertex part of p . (A p), summed by the tail kernels
        // global columns: vertex j as fit row j
        for (int an = 0; an < NANG; ++an) {
            double a = 0.0;
#pragma unroll
            for (int o = 0; o < DIM; ++o) {
                double b = 0.0;
#pragma unroll
                for (int c = 0; c < DIM; ++c) b += rs[j * st.vj + c * st.vc] * g->dR[an][c * DIM + o];
                a += SQUARE ? (c_fit * b) * (c_fit * b) : -c_fit * b * u[j * st.fj + o * st.fo];
            }
            gacc[an] = SQUARE && rw.f2 ? a * rw.f2[j] : a;
        }
#pragma unroll
        for (int o = 0; o < DIM; ++o)
            gacc[NANG + o] = SQUARE ? c_fit * c_fit * (rw.f2 ? rw.f2[j] : 1.0) : -c_fit * u[j * st.fj + o * st.fo];
    }
    // halo copies (j >= nown) belong to another rank's sums
    if (j >= nown) {
#pragma unroll
        for (int k = 0; k < 9; ++k) gacc[k] = 0.0;
    }
    // block partial of the NG global entries + p.Ap (summed by rows_sum_kernel)
    __shared__ double red[NB / 64][9];
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (k >= NG && k != 8) continue;
        const double s = wave_sum(gacc[k]);
        if ((tid & 63) == 0) red[tid >> 6][k] = s;
    }
    __syncthreads();
    if (tid < 9 && (tid < NG || tid == 8)) {
        double s = red[0][tid];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s += red[w][tid];
        partial[(j / blockDim.x) * kSumSlots + tid] = s;
    }
}

// ------------------------------------------------------------------ fused CG operator: out = J^T W^2 J p + lam_d2 .* p
// The CG operator never needs y = J p as a vector: jp_kernel wrote its 9N + 6E entries (27 doubles per vertex at
// E = 3N) and jtu_kernel read them back through gathers, about half of the memory-side traffic of the pair
// (profiles/r2_pmc_nicp200k_before.txt).  Only the FIT rows couple a vertex to the parameters of other vertices'
// neighbours (two hops), so only they keep an intermediate: fitrow_kernel writes dim doubles per vertex, and
// normal_kernel forms the rigidity rows (local) and the smoothness rows (from the two endpoint blocks of p) in
// registers.  Every row value and every sum is formed by the same operations in the same order as in
// jp_kernel(sq = 1) -> jtu_kernel, so the result is bit-identical to that pair (ICP_NICP_FUSEDOP=0 runs the pair;
// test_fused_operator_is_bit_identical).
template <int DIM, bool COMPAT>
__global__ __launch_bounds__(NB) void fitrow_kernel(const double *__restrict__ p, const int32_t *__restrict__ nbr,
                                                   const double *__restrict__ wt, int K, const double *__restrict__ v,
                                                   const double *__restrict__ rs, int64_t n, const Globals *__restrict__ g,
                                                   double c_fit, double *__restrict__ yfit, Strides st, RowW rw,
                                                   const int *__restrict__ stop) {
    constexpr int S = Lay<DIM>::S, NANG = Lay<DIM>::NANG;
    const int64_t t = (int64_t)vblock() * blockDim.x + threadIdx.x;
    if (t >= n) return;
    // CG iterations are enqueued in chunks (hipGraph replay): those behind the one that converged do nothing
    if (stop && *stop) return;
    const double *ptail = p + n * S;
    if (rw.f2) c_fit *= rw.f2[t];
    double vi[DIM], acc[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
        vi[c] = v[t * st.vj + c * st.vc];
        acc[c] = 0.0;
    }
    for (int k0 = 0; k0 < K; k0 += 4) {  // four neighbours per batch: all gathers issued before use
        int32_t jj[4];
        double ww[4], pj[4][S], vn[4][DIM];
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4) {
            const int kk = k0 + b4 < K ? k0 + b4 : K - 1;
            jj[b4] = nbr[t * K + kk];
            ww[b4] = k0 + b4 < K ? wt[t * K + kk] : 0.0;
        }
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4) {
#pragma unroll
            for (int e = 0; e < S; ++e) pj[b4][e] = p[(int64_t)jj[b4] * st.pj + e * st.pe];
#pragma unroll
            for (int r = 0; r < DIM; ++r) vn[b4][r] = v[(int64_t)jj[b4] * st.vj + r * st.vc];
        }
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4) {
            const double w = ww[b4];
            if (w == 0.0) continue;
            const double wb = (DIM == 2 && COMPAT) ? 1.0 : w;  // Q7
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                double a = 0.0;
#pragma unroll
                for (int r = 0; r < DIM; ++r) a += (vi[r] - vn[b4][r]) * pj[b4][r * DIM + c];
                acc[c] += w * a + wb * pj[b4][DIM * DIM + c];
            }
        }
    }
#pragma unroll
    for (int o = 0; o < DIM; ++o) {
        double a = ptail[NANG + o];
#pragma unroll
        for (int c = 0; c < DIM; ++c) a += acc[c] * g->R[c * DIM + o];
        for (int an = 0; an < NANG; ++an) {
            double b = 0.0;
#pragma unroll
            for (int c = 0; c < DIM; ++c) b += rs[t * st.vj + c * st.vc] * g->dR[an][c * DIM + o];
            a += ptail[an] * b;
        }
        yfit[t * st.fj + o * st.fo] = -c_fit * a;
    }
}

template <int DIM, bool COMPAT>
__global__ __launch_bounds__(NB) void normal_kernel(const double *__restrict__ P, const double *__restrict__ p,
                                                   const double *__restrict__ yfit, const double *__restrict__ v,
                                                   const double *__restrict__ rs, const int32_t *__restrict__ rev_ptr,
                                                   const int32_t *__restrict__ rev_i, const double *__restrict__ rev_w,
                                                   const int32_t *__restrict__ inc_ptr, const int32_t *__restrict__ inc_ent,
                                                   const int32_t *__restrict__ inc_other, int64_t n,
                                                   const Globals *__restrict__ g, double c_fit, double c_rigid, double c_smooth,
                                                   const double *__restrict__ lam_d2, double lam, double *__restrict__ out,
                                                   double *__restrict__ partial, int64_t nown, Strides st, RowW rw,
                                                   const int *__restrict__ stop) {
    constexpr int S = Lay<DIM>::S, NANG = Lay<DIM>::NANG, NG = Lay<DIM>::NG, NA = Lay<DIM>::NA;
    const int64_t j = (int64_t)vblock() * blockDim.x + threadIdx.x;
    if (stop && *stop) return;  // (see fitrow_kernel; nobody reads the partial rows of a stopped iteration)
    double gacc[kSumSlots];
#pragma unroll
    for (int k = 0; k < kSumSlots; ++k) gacc[k] = 0.0;
    if (j < n) {
        double gA[DIM][DIM], gb[DIM], vj[DIM], pown[S], Aown[NA], Rm[DIM][DIM];
#pragma unroll
        for (int r = 0; r < DIM; ++r) {
            vj[r] = v[j * st.vj + r * st.vc];
            gb[r] = 0.0;
#pragma unroll
            for (int c = 0; c < DIM; ++c) gA[r][c] = 0.0;
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c)
#pragma unroll
            for (int o = 0; o < DIM; ++o) Rm[c][o] = g->R[c * DIM + o];
        constexpr int RB = 4, IB = 3;  // fit rows / edges per batch (an edge brings a whole 12-double block of p)
        const int32_t rv_beg = rev_ptr[j], rv_end = rev_ptr[j + 1], in_beg = inc_ptr[j], in_end = inc_ptr[j + 1];
        const int32_t rv_cl = rv_end > rv_beg ? rv_end - 1 : 0, in_cl = in_end > in_beg ? in_end - 1 : 0;

        auto fit_entries = [&](int32_t e0, int32_t (&ii)[RB], double (&ww)[RB]) {
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int32_t ee = e0 + q < rv_end ? e0 + q : rv_cl;
                ii[q] = rev_i[ee];
                ww[q] = e0 + q < rv_end ? rev_w[ee] : 0.0;
            }
        };
        auto fit_gather = [&](const int32_t (&ii)[RB], double (&uu)[RB][DIM], double (&vv)[RB][DIM]) {
#pragma unroll
            for (int q = 0; q < RB; ++q)
#pragma unroll
                for (int o = 0; o < DIM; ++o) {
                    uu[q][o] = yfit[(int64_t)ii[q] * st.fj + o * st.fo];
                    vv[q][o] = v[(int64_t)ii[q] * st.vj + o * st.vc];
                }
        };
        auto fit_accumulate = [&](const double (&ww)[RB], const double (&uu)[RB][DIM], const double (&vv)[RB][DIM]) {
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const double w = ww[q];
                if (w == 0.0) continue;
                const double wb = (DIM == 2 && COMPAT) ? 1.0 : w;
                double yv[DIM];
#pragma unroll
                for (int c = 0; c < DIM; ++c) {
                    double a = 0.0;
#pragma unroll
                    for (int o = 0; o < DIM; ++o) a += Rm[c][o] * uu[q][o];
                    yv[c] = a;
                }
#pragma unroll
                for (int r = 0; r < DIM; ++r) {
                    const double d = vv[q][r] - vj[r];
#pragma unroll
                    for (int c = 0; c < DIM; ++c) gA[r][c] += -c_fit * w * d * yv[c];
                }
#pragma unroll
                for (int c = 0; c < DIM; ++c) gb[c] += -c_fit * wb * yv[c];
            }
        };
        auto edge_entries = [&](int32_t e0, int32_t (&ent)[IB], int32_t (&oth)[IB]) {
#pragma unroll
            for (int q = 0; q < IB; ++q) {
                const int32_t ee = e0 + q < in_end ? e0 + q : in_cl;
                ent[q] = inc_ent[ee];
                oth[q] = inc_other[ee];
            }
        };
        auto edge_gather = [&](const int32_t (&ent)[IB], const int32_t (&oth)[IB], double (&po)[IB][S], double (&vo)[IB][DIM],
                               double (&e2)[IB]) {
#pragma unroll
            for (int q = 0; q < IB; ++q) {
#pragma unroll
                for (int o = 0; o < DIM; ++o) vo[q][o] = v[(int64_t)oth[q] * st.vj + o * st.vc];
#pragma unroll
                for (int e = 0; e < S; ++e) po[q][e] = p[(int64_t)oth[q] * st.pj + e * st.pe];
                const double ew = rw.edge ? rw.edge[ent[q] >> 1] : 1.0;
                e2[q] = rw.edge ? ew * ew : 1.0;
            }
        };
        // smoothness rows of one incident edge (nicp_3d.py:316-333), as jp_kernel forms them with sq = 1, and their
        // J^T contributions to THIS endpoint, as jtu_kernel adds them
        auto edge_accumulate = [&](int32_t e0, const int32_t (&ent)[IB], const double (&po)[IB][S], const double (&vo)[IB][DIM],
                                   const double (&e2)[IB]) {
#pragma unroll
            for (int q = 0; q < IB; ++q) {
                if (e0 + q >= in_end) continue;
                const int end = ent[q] & 1;
                const double cs2 = rw.edge ? c_smooth * e2[q] : c_smooth;
                double d[DIM];
#pragma unroll
                for (int r = 0; r < DIM; ++r) d[r] = end == 0 ? vo[q][r] - vj[r] : vj[r] - vo[q][r];
#pragma unroll
                for (int c = 0; c < DIM; ++c) {
                    double a1 = 0.0, a2 = 0.0;  // d . (A-part of p of endpoint 1 / endpoint 2)[:, c]
#pragma unroll
                    for (int r = 0; r < DIM; ++r) {
                        const double q1 = end == 0 ? pown[r * DIM + c] : po[q][r * DIM + c];
                        const double q2 = end == 0 ? po[q][r * DIM + c] : pown[r * DIM + c];
                        a1 += d[r] * q1;
                        a2 += d[r] * q2;
                    }
                    const double b1 = end == 0 ? pown[DIM * DIM + c] : po[q][DIM * DIM + c];
                    const double b2 = end == 0 ? po[q][DIM * DIM + c] : pown[DIM * DIM + c];
                    const double u0 = cs2 * (a1 + b1 - b2), u1 = cs2 * (-a2 + b2 - b1);
#pragma unroll
                    for (int r = 0; r < DIM; ++r) gA[r][c] += end == 0 ? c_smooth * d[r] * u0 : -c_smooth * d[r] * u1;
                    gb[c] += end == 0 ? c_smooth * (u0 - u1) : c_smooth * (u1 - u0);
                }
            }
        };

        // first batches of both lists + the vertex's own blocks: entries, then every gather
        int32_t ii0[RB], ent0[IB], oth0[IB];
        double ww0[RB], uu0[RB][DIM], vv0[RB][DIM], po0[IB][S], vo0[IB][DIM], e20[IB];
        fit_entries(rv_beg, ii0, ww0);
        edge_entries(in_beg, ent0, oth0);
#pragma unroll
        for (int e = 0; e < NA; ++e) Aown[e] = P[j * st.pj + e * st.pe];
#pragma unroll
        for (int e = 0; e < S; ++e) pown[e] = p[j * st.pj + e * st.pe];
        fit_gather(ii0, uu0, vv0);
        edge_gather(ent0, oth0, po0, vo0, e20);

        fit_accumulate(ww0, uu0, vv0);
        for (int32_t e0 = rv_beg + RB; e0 < rv_end; e0 += RB) {
            int32_t ii[RB];
            double ww[RB], uu[RB][DIM], vv[RB][DIM];
            fit_entries(e0, ii, ww);
            fit_gather(ii, uu, vv);
            fit_accumulate(ww, uu, vv);
        }
        // rigidity rows of vertex j: y = W^2 G pa (jp_kernel), then G^T y (jtu_kernel)
        {
            double cr = c_rigid;
            if (rw.rig) cr *= rw.rig[j] * rw.rig[j];
            const double *A = Aown, *pa = pown;
            if constexpr (DIM == 3) {
                double d01 = 0, d10 = 0, d12 = 0, d21 = 0, d20 = 0, d02 = 0, d00 = 0, d11 = 0, d22 = 0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    d10 += A[3 + c] * pa[c];
                    d01 += A[c] * pa[3 + c];
                    d21 += A[6 + c] * pa[3 + c];
                    d12 += A[3 + c] * pa[6 + c];
                    d20 += A[6 + c] * pa[c];
                    d02 += A[c] * pa[6 + c];
                    d00 += A[c] * pa[c];
                    d11 += A[3 + c] * pa[3 + c];
                    d22 += A[6 + c] * pa[6 + c];
                }
                const double ur[6] = {cr * (d10 + d01), cr * (d21 + d12), cr * (d20 + d02), cr * (-2 * d00), cr * (-2 * d11), cr * (-2 * d22)};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double a0 = A[c], a1 = A[3 + c], a2 = A[6 + c];
                    gA[0][c] += c_rigid * (ur[0] * a1 + ur[2] * a2 - 2 * ur[3] * a0);
                    gA[1][c] += c_rigid * (ur[0] * a0 + ur[1] * a2 - 2 * ur[4] * a1);
                    gA[2][c] += c_rigid * (ur[1] * a1 + ur[2] * a0 - 2 * ur[5] * a2);
                }
            } else {
                double d10 = 0, d01 = 0, d00 = 0, d11 = 0;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    d10 += A[2 + c] * pa[c];
                    d01 += A[c] * pa[2 + c];
                    d00 += A[c] * pa[c];
                    d11 += A[2 + c] * pa[2 + c];
                }
                const double ur[3] = {cr * (d10 + d01), cr * (-2 * d00), cr * (-2 * d11)};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double a0 = A[c], a1 = A[2 + c];
                    gA[0][c] += c_rigid * (ur[0] * a1 - 2 * ur[1] * a0);
                    gA[1][c] += c_rigid * (ur[0] * a0 - 2 * ur[2] * a1);
                }
            }
        }
        // smoothness rows of the edges incident to vertex j
        edge_accumulate(in_beg, ent0, po0, vo0, e20);
        for (int32_t e0 = in_beg + IB; e0 < in_end; e0 += IB) {
            int32_t ent[IB], oth[IB];
            double po[IB][S], vo[IB][DIM], e2[IB];
            edge_entries(e0, ent, oth);
            edge_gather(ent, oth, po, vo, e2);
            edge_accumulate(e0, ent, po, vo, e2);
        }
        double pap = 0.0;
#pragma unroll
        for (int r = 0; r < DIM; ++r)
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                const int64_t o = j * st.pj + (r * DIM + c) * st.pe;
                const double val = gA[r][c] + lam * lam_d2[o] * pown[r * DIM + c];
                out[o] = val;
                pap += val * pown[r * DIM + c];
            }
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            const int64_t o = j * st.pj + (DIM * DIM + c) * st.pe;
            const double val = gb[c] + lam * lam_d2[o] * pown[DIM * DIM + c];
            out[o] = val;
            pap += val * pown[DIM * DIM + c];
        }
        gacc[8] = pap;  // vertex part of p . (A p), summed by the tail kernels
        // global columns: vertex j as fit row j
        double uown[DIM];
#pragma unroll
        for (int o = 0; o < DIM; ++o) uown[o] = yfit[j * st.fj + o * st.fo];
        for (int an = 0; an < NANG; ++an) {
            double a = 0.0;
#pragma unroll
            for (int o = 0; o < DIM; ++o) {
                double b = 0.0;
#pragma unroll
                for (int c = 0; c < DIM; ++c) b += rs[j * st.vj + c * st.vc] * g->dR[an][c * DIM + o];
                a += -c_fit * b * uown[o];
            }
            gacc[an] = a;
        }
#pragma unroll
        for (int o = 0; o < DIM; ++o) gacc[NANG + o] = -c_fit * uown[o];
    }
    // halo copies (j >= nown) belong to another rank's sums
    if (j >= nown) {
#pragma unroll
        for (int k = 0; k < 9; ++k) gacc[k] = 0.0;
    }
    __shared__ double red[NB / 64][9];
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (k >= NG && k != 8) continue;
        const double sv = wave_sum(gacc[k]);
        if ((tid & 63) == 0) red[tid >> 6][k] = sv;
    }
    __syncthreads();
    if (tid < 9 && (tid < NG || tid == 8)) {
        double sv = red[0][tid];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) sv += red[w][tid];
        partial[(j / blockDim.x) * kSumSlots + tid] = sv;
    }
}

// ------------------------------------------------------------------ vector kernels
// device scalars used by the CG / LM loops
struct Scal {
    double rz, rz_new, pAp, rr, rr0, alpha, beta, tmp[8], rz0;
    int32_t stop, iters;
};

__device__ __forceinline__ void block_sum_to(double vloc, double *partial) {
    __shared__ double red[NB / 64];
    const int tid = threadIdx.x;
    const double s = wave_sum(vloc);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        double t = red[0];
        for (int w = 1; w < NB / 64; ++w) t += red[w];
        partial[blockIdx.x] = t;
    }
}

// partial[b] = sum over the block's grid-stride elements of a[i]*b[i]*(w ? w[i] : 1)
__global__ __launch_bounds__(NB) void dot_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                 const double *__restrict__ w, int64_t len, double *__restrict__ partial) {
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * NB + threadIdx.x; i < len; i += (int64_t)gridDim.x * NB)
        s += a[i] * b[i] * (w ? w[i] : 1.0);
    block_sum_to(s, partial);
}

__global__ __launch_bounds__(NB) void final_sum_kernel(const double *__restrict__ partial, int nb, double *__restrict__ out) {
    __shared__ double red[NB];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int b = tid; b < nb; b += NB) s += partial[b];
    red[tid] = s;
    __syncthreads();
    for (int o = NB / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) *out = red[0];
}

// CG bookkeeping on the device (no host round trip inside the iteration)
// generic elementwise helpers
__global__ __launch_bounds__(NB) void axpby_kernel(double a, const double *__restrict__ x, double b,
                                                   const double *__restrict__ y, double *__restrict__ out, int64_t len) {
    for (int64_t i = (int64_t)blockIdx.x * NB + threadIdx.x; i < len; i += (int64_t)gridDim.x * NB)
        out[i] = a * x[i] + (y ? b * y[i] : 0.0);
}
// MINPACK-style scaling: d2 = max(d2_prev, colnorm2) (zero columns -> 1); lamd2 = lam d2 (what the CG
// operator adds: kept as a vector so that the kernel arguments of a CG iteration do not change with lam)
__global__ __launch_bounds__(NB) void scaling_kernel(const double *__restrict__ col2, double *__restrict__ d2,
                                                     double *__restrict__ lamd2, double lam, int first, int64_t len) {
    for (int64_t i = (int64_t)blockIdx.x * NB + threadIdx.x; i < len; i += (int64_t)gridDim.x * NB) {
        double c = col2[i];
        double d = first ? (c > 0 ? c : 1.0) : (d2[i] > c ? d2[i] : c);
        d2[i] = d;
        lamd2[i] = lam * d;
    }
}

// ------------------------------------------------------------------ two-level preconditioner
// M^-1 = blockdiag(J^T J + lam D^2)^-1  +  Z (Z^T (J^T J + lam D^2) Z)^-1 Z^T.
// The per-vertex S x S blocks remove the 1e6 : 1e4 : 1e-2 spread between the rigidity,
// smoothness and fit terms inside one vertex; the coarse space Z removes the modes the
// stiffness terms cannot see at all: a common rotation / translation of every vertex
// (A_j -> A_j Q, b_j -> (v_j + b_j) Q - v_j + t leaves E_rigid and E_smooth unchanged) together
// with the explicit global parameters that duplicate them (the gauge of this model).  Without
// it CG needs ~1e4 iterations per step at 10k vertices.
// The coarse space can be enriched: the NG zero-stiffness modes multiplied by M low-order polynomials of the
// (normalised) vertex position -- 1 | x y z | x^2 y^2 z^2 xy yz xz -- deflate the smooth low-frequency
// deformations on which block-Jacobi CG spends most of its iterations (about half as many iterations with the
// quadratics, tests/study_coarse_space.py).  Column a < NG*M of Z is phi_{a / NG}(v_j) z_{a % NG}(j); the last
// NG columns are the global parameters.  M = 1 is the plain 2*NG-column space.
constexpr int MAXPOLY = 10;
constexpr int NZMAX = 6 * MAXPOLY + 6;
constexpr int kMomSlots = 64;  // doubles per row of block moments Z_v^T r (NG*M <= 60)
struct PolyBasis {
    double c[3];  // centre (mean vertex)
    double inv;   // 1 / (2 rms radius)
    int M;        // number of polynomials in use: 1, 4, 10 (3-D) / 1, 3, 6 (2-D)
};
template <int DIM>
__device__ __forceinline__ void poly_eval(const double *vj, const PolyBasis &pb, double (&phi)[MAXPOLY]) {
    const double x = (vj[0] - pb.c[0]) * pb.inv, y = (vj[1] - pb.c[1]) * pb.inv;
    phi[0] = 1.0;
    if (DIM == 3) {
        const double z = (vj[2] - pb.c[2]) * pb.inv;
        phi[1] = x; phi[2] = y; phi[3] = z;
        phi[4] = x * x; phi[5] = y * y; phi[6] = z * z;
        phi[7] = x * y; phi[8] = y * z; phi[9] = x * z;
    } else {
        phi[1] = x; phi[2] = y;
        phi[3] = x * x; phi[4] = y * y; phi[5] = x * y;
        phi[6] = phi[7] = phi[8] = phi[9] = 0.0;
    }
}
struct Coarse {
    double Einv[NZMAX * NZMAX];
    double E[NZMAX * NZMAX];
    double ztr[NZMAX], y[NZMAX];
    double tail_minv[6];
    double tol2;   // square of the relative residual at which this CG solve stops (device-resident: may change per LM step)
    double tol2z;  // the same bound on the PRECONDITIONED residual r.z / r0.z0 (+inf: not tested); directly behind tol2
};

// the NG per-vertex zero-stiffness modes at vertex j: z[k][0..S)
template <int DIM>
__device__ __forceinline__ void zmodes(const double *__restrict__ Pj, const double *__restrict__ vj,
                                       double (&z)[Lay<DIM>::NG][Lay<DIM>::S]) {
    constexpr int S = Lay<DIM>::S, NG = Lay<DIM>::NG;
#pragma unroll
    for (int k = 0; k < NG; ++k)
#pragma unroll
        for (int e = 0; e < S; ++e) z[k][e] = 0.0;
    if (DIM == 3) {
        // x Omega_0 = (-x1, x0, 0); x Omega_1 = (0, -x2, x1); x Omega_2 = (x2, 0, -x0)
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // rows of A, then the row vector v_j + b_j
            double x0, x1, x2;
            if (r < 3) { x0 = Pj[r * 3]; x1 = Pj[r * 3 + 1]; x2 = Pj[r * 3 + 2]; }
            else { x0 = vj[0] + Pj[9]; x1 = vj[1] + Pj[10]; x2 = vj[2] + Pj[11]; }
            const int o = r * 3;
            z[0][o] = -x1; z[0][o + 1] = x0;
            z[1][o + 1] = -x2; z[1][o + 2] = x1;
            z[2][o] = x2; z[2][o + 2] = -x0;
        }
        z[3][9] = 1.0; z[4][10] = 1.0; z[5][11] = 1.0;
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double x0, x1;
            if (r < 2) { x0 = Pj[r * 2]; x1 = Pj[r * 2 + 1]; }
            else { x0 = vj[0] + Pj[4]; x1 = vj[1] + Pj[5]; }
            z[0][r * 2] = -x1; z[0][r * 2 + 1] = x0;
        }
        z[1][4] = 1.0; z[2][5] = 1.0;
    }
}

// S x S diagonal block of J^T J + lam D^2 for every vertex, inverted; packed lower triangle.
// One vertex per lane, everything in registers: only the lower triangle of the block is formed (78 doubles in 3-D,
// the Cholesky factorisation reads nothing else), the factor and then its inverse overwrite it in place, and every
// loop over matrix entries has constant bounds, so no index is a run-time value.  (The first form kept three full
// S x S arrays per lane with run-time indices -- 3.4 KB of scratch memory per lane: 1.15 ms per call at 200 k vertices,
// one call per LM trial.)  Same operations in the same order per entry: the factor is bit-identical to that form's.
__host__ __device__ constexpr int tri_index(int a, int b) { return a * (a + 1) / 2 + b; }  // a >= b

template <int DIM, bool COMPAT>
__global__ __launch_bounds__(64) void block_diag_kernel(const double *__restrict__ P, const double *__restrict__ wt, int K,
                                                       const double *__restrict__ v, const int32_t *__restrict__ edges,
                                                       const int32_t *__restrict__ rev_ptr, const int32_t *__restrict__ rev_ent,
                                                       const int32_t *__restrict__ inc_ptr, const int32_t *__restrict__ inc_ent,
                                                       int64_t n, const Globals *__restrict__ g, double c_fit, double c_rigid,
                                                       double c_smooth, const double *__restrict__ col2,
                                                       const double *__restrict__ d2, double lam,
                                                       double *__restrict__ binv, Coarse *__restrict__ coarse, Strides st,
                                                       RowW rw, int prec32) {
    // prec32 = 1: store the inverse Cholesky factor Li (B^-1 = Li^T Li) in fp32 instead of the packed fp64 inverse:
    // half the bytes of the largest array the CG update kernel streams, and Li^T (Li r) is symmetric positive
    // definite for ANY rounding of Li, which is all a preconditioner has to be
    constexpr int S = Lay<DIM>::S, NA = Lay<DIM>::NA, NG = Lay<DIM>::NG, TRI = S * (S + 1) / 2;
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (j == 0)
        for (int k = 0; k < NG; ++k) coarse->tail_minv[k] = 1.0 / (col2[n * S + k] + lam * d2[n * S + k]);
    if (j >= n) return;
    double B[TRI];  // lower triangle, row by row
#pragma unroll
    for (int q = 0; q < TRI; ++q) B[q] = 0.0;
    double M[DIM][DIM];  // R R^T
#pragma unroll
    for (int a = 0; a < DIM; ++a)
#pragma unroll
        for (int b = 0; b < DIM; ++b) {
            double m = 0.0;
#pragma unroll
            for (int o = 0; o < DIM; ++o) m += g->R[a * DIM + o] * g->R[b * DIM + o];
            M[a][b] = m;
        }
    double vj[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) vj[a] = v[j * st.vj + a * st.vc];
    // fit rows that reference vertex j: entries -c_fit w d[r] R[c][o] (A) and -c_fit wb R[c][o] (b)
    for (int32_t e = rev_ptr[j]; e < rev_ptr[j + 1]; ++e) {
        const int32_t ent = rev_ent[e];
        const int64_t i = ent / K;
        const double w = wt[ent];
        if (w == 0.0) continue;
        const double wb = (DIM == 2 && COMPAT) ? 1.0 : w;
        double d[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) d[a] = v[i * st.vj + a * st.vc] - vj[a];
        const double cf2 = c_fit * c_fit * (rw.f2 ? rw.f2[i] : 1.0);
#pragma unroll
        for (int r = 0; r < DIM; ++r)
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
#pragma unroll
                for (int r2 = 0; r2 < DIM; ++r2)
#pragma unroll
                    for (int c2 = 0; c2 < DIM; ++c2)
                        if (r * DIM + c >= r2 * DIM + c2) B[tri_index(r * DIM + c, r2 * DIM + c2)] += cf2 * w * w * d[r] * d[r2] * M[c][c2];
#pragma unroll
                for (int c2 = 0; c2 < DIM; ++c2) B[tri_index(NA + c2, r * DIM + c)] += cf2 * w * wb * d[r] * M[c][c2];
            }
#pragma unroll
        for (int c = 0; c < DIM; ++c)
#pragma unroll
            for (int c2 = 0; c2 <= c; ++c2) B[tri_index(NA + c, NA + c2)] += cf2 * wb * wb * M[c][c2];
    }
    // rigid rows: gradients of the NRIG terms w.r.t. the NA entries of A_j
    {
        double A[NA];
#pragma unroll
        for (int e = 0; e < NA; ++e) A[e] = P[j * st.pj + e * st.pe];
        const double cr2 = c_rigid * c_rigid * (rw.rig ? rw.rig[j] * rw.rig[j] : 1.0);
        constexpr int NRIG = Lay<DIM>::NRIG;
        double G[NRIG][NA];
#pragma unroll
        for (int m = 0; m < NRIG; ++m)
#pragma unroll
            for (int e = 0; e < NA; ++e) G[m][e] = 0.0;
        if (DIM == 3) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double a0 = A[c], a1 = A[3 + c], a2 = A[6 + c];
                G[0][c] = a1; G[0][3 + c] = a0;
                G[1][3 + c] = a2; G[1][6 + c] = a1;
                G[2][c] = a2; G[2][6 + c] = a0;
                G[3][c] = -2 * a0; G[4][3 + c] = -2 * a1; G[5][6 + c] = -2 * a2;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double a0 = A[c], a1 = A[2 + c];
                G[0][c] = a1; G[0][2 + c] = a0;
                G[1][c] = -2 * a0; G[2][2 + c] = -2 * a1;
            }
        }
#pragma unroll
        for (int m = 0; m < NRIG; ++m)
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) B[tri_index(a, b)] += cr2 * G[m][a] * G[m][b];
    }
    // smooth rows of the incident edges
    for (int32_t e = inc_ptr[j]; e < inc_ptr[j + 1]; ++e) {
        const int32_t ent = inc_ent[e];
        const int64_t ed = ent >> 1;
        const double sgn = (ent & 1) ? -1.0 : 1.0;
        const int64_t p1 = edges[ed * 2], p2 = edges[ed * 2 + 1];
        const double cs2 = c_smooth * c_smooth * (rw.edge ? rw.edge[ed] * rw.edge[ed] : 1.0);
        double d[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) d[a] = v[p2 * st.vj + a * st.vc] - v[p1 * st.vj + a * st.vc];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
#pragma unroll
            for (int r = 0; r < DIM; ++r) {
#pragma unroll
                for (int r2 = 0; r2 <= r; ++r2) B[tri_index(r * DIM + c, r2 * DIM + c)] += cs2 * d[r] * d[r2];
                B[tri_index(NA + c, r * DIM + c)] += sgn * cs2 * d[r];
            }
            B[tri_index(NA + c, NA + c)] += 2 * cs2;
        }
    }
    double diag[S];  // for the fall-back below (the factor overwrites B)
#pragma unroll
    for (int a = 0; a < S; ++a) {
        B[tri_index(a, a)] += lam * d2[j * st.pj + a * st.pe];
        diag[a] = B[tri_index(a, a)];
    }
    // Cholesky B = L L^T in place, then L^-1 in place; on breakdown fall back to the diagonal
    bool ok = true;
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            double sum = B[tri_index(a, b)];
#pragma unroll
            for (int k = 0; k < b; ++k) sum -= B[tri_index(a, k)] * B[tri_index(b, k)];
            if (a == b) {
                if (!(sum > 0.0)) ok = false;
                B[tri_index(a, a)] = ok ? sqrt(sum) : 1.0;
            } else {
                B[tri_index(a, b)] = sum / B[tri_index(b, b)];
            }
        }
    // packed lower triangle, entry q of vertex j at binv[q*n + j] (coalesced for the per-vertex lanes)
    double *out = binv + j;
    float *out32 = (float *)binv + j;
    if (!ok) {
#pragma unroll
        for (int a = 0; a < S; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                if (prec32) out32[(int64_t)tri_index(a, b) * n] = a == b ? (float)(1.0 / sqrt(diag[a])) : 0.0f;
                else out[(int64_t)tri_index(a, b) * n] = a == b ? 1.0 / diag[a] : 0.0;
            }
        return;
    }
    // column c of Li = L^-1 needs the columns >= c of L and the diagonal entries below: ascending c leaves them intact
#pragma unroll
    for (int c = 0; c < S; ++c) {
        double col[S];  // L[:, c] (the sums read it while Li[:, c] is being written)
#pragma unroll
        for (int a = c + 1; a < S; ++a) col[a] = B[tri_index(a, c)];
        B[tri_index(c, c)] = 1.0 / B[tri_index(c, c)];
#pragma unroll
        for (int a = c + 1; a < S; ++a) {
            double sum = 0.0;
#pragma unroll
            for (int k = c; k < a; ++k) sum -= (k == c ? col[a] : B[tri_index(a, k)]) * B[tri_index(k, c)];
            B[tri_index(a, c)] = sum / B[tri_index(a, a)];
        }
    }
    if (prec32) {
#pragma unroll
        for (int q = 0; q < TRI; ++q) out32[(int64_t)q * n] = (float)B[q];
        return;
    }
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            double sum = 0.0;
#pragma unroll
            for (int k = a; k < S; ++k) sum += B[tri_index(k, a)] * B[tri_index(k, b)];
            out[(int64_t)tri_index(a, b) * n] = sum;
        }
}

// column a of Z as a full parameter vector
template <int DIM>
__global__ __launch_bounds__(NB) void zcol_kernel(const double *__restrict__ P, const double *__restrict__ v, int64_t n,
                                                 int a, double *__restrict__ out, Strides st, PolyBasis pb) {
    constexpr int S = Lay<DIM>::S, NG = Lay<DIM>::NG;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nm = NG * pb.M, km = a % NG, mm = a / NG;
    if (j < n) {
        double z[NG][S];
        double Pj_[S], vj_[DIM], phi[MAXPOLY];
        for (int e = 0; e < S; ++e) Pj_[e] = P[j * st.pj + e * st.pe];
        for (int c = 0; c < DIM; ++c) vj_[c] = v[j * st.vj + c * st.vc];
        zmodes<DIM>(Pj_, vj_, z);
        poly_eval<DIM>(vj_, pb, phi);
        double f = 0.0;
        for (int m = 0; m < MAXPOLY; ++m) f = (a < nm && m == mm) ? phi[m] : f;
        for (int e = 0; e < S; ++e) {
            double val = 0.0;
            for (int k = 0; k < NG; ++k) val = (a < nm && k == km) ? z[k][e] : val;
            out[j * st.pj + e * st.pe] = pb.M == 1 ? val : f * val;
        }
    } else if (j < n + NG) {
        out[n * S + (j - n)] = (a - nm == (int)(j - n)) ? 1.0 : 0.0;
    }
}

// Block partial of the NG*M moments  sum_j phi_m(v_j) t_k(j)  (t_k(j) = z_k(j) . r_j, zero on idle lanes) into one
// wide row: slot m*NG + k.  One vertex per lane.  Sixty butterfly reductions would cost sixty shuffle chains, so
// every lane parks its 10 + NG numbers in an LDS row instead and lane q < NG*M of each wave adds up "its"
// product over the wave's 64 rows (rows are 128 B: the reads of one step fall in one row, conflict-free);
// the waves' results are then added in wave order.  Fixed order, hence deterministic.
template <int DIM>
__device__ __forceinline__ void block_moments(const double (&t)[Lay<DIM>::NG], const double (&phi)[MAXPOLY], int M,
                                              double *__restrict__ mom_row) {
    constexpr int NG = Lay<DIM>::NG, ROW = 16;
    static_assert(MAXPOLY + NG <= ROW, "LDS row layout");
    __shared__ double rows[NB][ROW];
    __shared__ double mred[NB / 64][MAXPOLY * NG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int m = 0; m < MAXPOLY; ++m) rows[tid][m] = phi[m];
#pragma unroll
    for (int k = 0; k < NG; ++k) rows[tid][MAXPOLY + k] = t[k];
    __syncthreads();
    const int nmom = M * NG;
    if (lane < nmom) {
        const int m = lane / NG, k = lane % NG;
        const double(*wr)[ROW] = rows + wave * 64;
        double sv = 0.0;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) sv += wr[i][m] * wr[i][MAXPOLY + k];
        mred[wave][lane] = sv;
    }
    __syncthreads();
    for (int q = tid; q < nmom; q += blockDim.x) {
        double sv = mred[0][q];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) sv += mred[w][q];
        mom_row[q] = sv;
    }
}

// partial[b][k] = sum over the block's vertices of z_k(j) . r_j  (k < NG)
template <int DIM>
__global__ __launch_bounds__(NB) void zt_kernel(const double *__restrict__ P, const double *__restrict__ v, int64_t n,
                                               const double *__restrict__ r, double *__restrict__ partial, Strides st,
                                               double *__restrict__ mom, PolyBasis pb) {  // n = owned vertices
    constexpr int S = Lay<DIM>::S, NG = Lay<DIM>::NG;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc[NG], phi[MAXPOLY];
    for (int k = 0; k < NG; ++k) acc[k] = 0.0;
    for (int m = 0; m < MAXPOLY; ++m) phi[m] = 0.0;
    if (j < n) {
        double z[NG][S];
        double Pj_[S], vj_[DIM];
        for (int e = 0; e < S; ++e) Pj_[e] = P[j * st.pj + e * st.pe];
        for (int c = 0; c < DIM; ++c) vj_[c] = v[j * st.vj + c * st.vc];
        zmodes<DIM>(Pj_, vj_, z);
        for (int k = 0; k < NG; ++k)
            for (int e = 0; e < S; ++e) acc[k] += z[k][e] * r[j * st.pj + e * st.pe];
        if (pb.M > 1) poly_eval<DIM>(vj_, pb, phi);
    }
    if (pb.M > 1) {  // enriched coarse space: all NG*M moments into the wide row
        block_moments<DIM>(acc, phi, pb.M, mom + (int64_t)blockIdx.x * kMomSlots);
        return;
    }
    __shared__ double red[NB / 64][NG];
    const int tid = threadIdx.x;
    for (int k = 0; k < NG; ++k) {
        const double s = wave_sum(acc[k]);
        if ((tid & 63) == 0) red[tid >> 6][k] = s;
    }
    __syncthreads();
    if (tid < NG) {
        double s = red[0][tid];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s += red[w][tid];
        partial[(int64_t)blockIdx.x * kSumSlots + tid] = s;
    }
}
// tot[off + k] = sum_b partial[b][k] for the slots in `mask`: one wave per slot (single block of 1024).
// Every reduction below is "block partials -> rows_sum -> [all-reduce over ranks] -> logic kernel", so the
// sharded solver only inserts the all-reduce; the summation tree is fixed, hence deterministic.
__device__ __forceinline__ void block_rows_sum(const double *__restrict__ partial, int nb, unsigned mask,
                                               double *__restrict__ tot, int off) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k < kSumSlots && ((mask >> k) & 1u)) {
        // the loads of 16 rows are issued before the first add: one memory latency per 1 024 block rows instead of
        // one per 64; the additions keep their order (ascending rows per lane), so the sums keep their bits
        constexpr int BATCH = 16;
        double a = 0.0;
        for (int b0 = lane; b0 < nb; b0 += 64 * BATCH) {
            double t[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int b = b0 + 64 * u;
                t[u] = b < nb ? partial[(int64_t)b * kSumSlots + k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
                if (b0 + 64 * u < nb) a += t[u];
        }
        a = wave_sum(a);
        if (lane == 0) tot[off + k] = a;
    }
}
// The same sums (same order, same bits) taken by EVERY block of a consumer kernel into its LDS.  Lets the
// single-GPU CG iteration drop its two single-block sum+logic launches.
template <unsigned MASK>
__device__ __forceinline__ void block_rows_sum_lds(const double *__restrict__ partial, int nb, double *out_lds) {
    // first wave only; four rows per lane in flight (every load issued before the first add: one memory latency for
    // up to 256 rows), accumulation per lane in ascending row order, then the wave sum: block_rows_sum's order
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    double acc[kSumSlots];
#pragma unroll
    for (int k = 0; k < kSumSlots; ++k) acc[k] = 0.0;
    for (int b0 = lane; b0 < nb; b0 += 256) {
        double val[4][kSumSlots];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b = b0 + 64 * q;
            const double *row = partial + (int64_t)(b < nb ? b : b0) * kSumSlots;
#pragma unroll
            for (int k = 0; k < kSumSlots; ++k)
                if ((MASK >> k) & 1u) val[q][k] = row[k];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < kSumSlots; ++k)
                if (((MASK >> k) & 1u) && b0 + 64 * q < nb) acc[k] += val[q][k];
    }
#pragma unroll
    for (int k = 0; k < kSumSlots; ++k)
        if ((MASK >> k) & 1u) {
            const double t = wave_sum(acc[k]);
            if (lane == 0) out_lds[k] = t;
        }
}
__global__ __launch_bounds__(1024) void rows_sum_kernel(const double *__restrict__ partial, int nb, unsigned mask,
                                                        double *__restrict__ tot, int off) {
    block_rows_sum(partial, nb, mask, tot, off);
}
// tot[off + k] = sum_b mom[b][k], k < nslots, over the wide moment rows: the waves of the block share the slots
// (1024 threads: thread = (slot, one of 16 row groups); the loads of a thread are independent and coalesced over
// the slots, the 16 group sums meet in LDS -- no shuffle chains.)  All threads of the block must call it.
__device__ __forceinline__ void block_rows_sum_wide(const double *__restrict__ mom, int nb, int nslots, double *__restrict__ tot,
                                                    int off) {
    __shared__ double wred[16][kMomSlots];
    const int slot = threadIdx.x & (kMomSlots - 1), grp = threadIdx.x / kMomSlots, ngrp = blockDim.x / kMomSlots;
    double a = 0.0;
    if (slot < nslots) {
        // eight rows in flight per thread (every load issued before the first add; the order of the additions is unchanged)
        constexpr int BATCH = 8;
        for (int b0 = grp; b0 < nb; b0 += ngrp * BATCH) {
            double t[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int b = b0 + u * ngrp;
                t[u] = b < nb ? mom[(int64_t)b * kMomSlots + slot] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
                if (b0 + u * ngrp < nb) a += t[u];
        }
    }
    wred[grp][slot] = a;
    __syncthreads();
    if (threadIdx.x < nslots) {
        double sv = wred[0][threadIdx.x];
        for (int q = 1; q < ngrp; ++q) sv += wred[q][threadIdx.x];
        tot[off + threadIdx.x] = sv;
    }
}
__global__ __launch_bounds__(1024) void rows_sum_wide_kernel(const double *__restrict__ mom, int nb, int nslots,
                                                             double *__restrict__ tot, int off) {
    block_rows_sum_wide(mom, nb, nslots, tot, off);
}

// out_tail[k] = tot[off + k] + lam d2_tail[k] p_tail[k]: tail of J^T u (the tail entries are replicated)
__global__ void tail_write_kernel(const double *__restrict__ tot, int off, int ng, const double *__restrict__ d2_tail,
                                  const double *__restrict__ p_tail, double lam, double *__restrict__ out_tail) {
    const int k = threadIdx.x;
    if (k < ng) out_tail[k] = tot[off + k] + (d2_tail ? lam * d2_tail[k] * p_tail[k] : 0.0);
}

// E[:, a]: tot[0..nm) = Z_v^T w (vertex modes), tot[nm+2..nm+2+ng) = tail of J^T J z_a  (nm = ng: tails at 8)
__global__ void ecol_logic_kernel(const double *__restrict__ tot, int ng, int nm, const double *__restrict__ d2_tail, double lam,
                                  int a, Coarse *__restrict__ coarse) {
    const int k = threadIdx.x, nz = nm + ng;
    if (k < nm) coarse->E[k * nz + a] = tot[k];
    if (k >= ng) return;
    const double ztail = (a - nm == k) ? 1.0 : 0.0;
    coarse->E[(nm + k) * nz + a] = tot[nm + 2 + k] + lam * d2_tail[k] * ztail;
}

// after J^T: tail of A p, p.Ap and alpha.  tot[0..ng) tail sums, tot[8] vertex part of p.Ap  (one thread)
__device__ __forceinline__ void alpha_logic(const double *tot, int ng, const double *d2_tail, const double *p_tail, double lam,
                                            double *Ap_tail, Scal *s) {
    if (s->stop) return;
    double pap = tot[8];
    for (int k = 0; k < ng; ++k) {
        const double val = tot[k] + lam * d2_tail[k] * p_tail[k];
        Ap_tail[k] = val;
        pap += val * p_tail[k];
    }
    s->pAp = pap;
    s->alpha = pap > 0 ? s->rz / pap : 0.0;
    if (!(pap > 0)) s->stop = 2;  // breakdown (non-positive curvature / NaN)
}
__global__ void cg_alpha_logic_kernel(const double *__restrict__ tot, int ng, const double *__restrict__ d2_tail,
                                      const double *__restrict__ p_tail, double lam, double *__restrict__ Ap_tail,
                                      Scal *__restrict__ s) {
    if (threadIdx.x == 0) alpha_logic(tot, ng, d2_tail, p_tail, lam, Ap_tail, s);
}
// single GPU: row sum and logic in one launch (nothing to all-reduce in between)
__global__ __launch_bounds__(1024) void cg_sum_alpha_kernel(const double *__restrict__ partial, int nb, unsigned mask,
                                                            double *__restrict__ tot, int ng, const double *__restrict__ d2_tail,
                                                            const double *__restrict__ p_tail, double lam,
                                                            double *__restrict__ Ap_tail, Scal *__restrict__ s) {
    // single block on the critical path of every iteration: thread 0 fetches the scalars of the alpha logic BEFORE the
    // row sums, so their latency overlaps the partial-row loads instead of following them
    __shared__ double tot_s[kSumSlots];
    (void)tot;
    int stop0 = 0;
    double rz0 = 0.0, dt0[6], pt0[6];
    if (threadIdx.x == 0) {
        stop0 = s->stop;
        rz0 = s->rz;
        for (int k = 0; k < ng; ++k) {
            dt0[k] = d2_tail[k];
            pt0[k] = p_tail[k];
        }
    }
    block_rows_sum(partial, nb, mask, tot_s, 0);
    __syncthreads();
    if (threadIdx.x == 0 && !stop0) {  // alpha_logic on the prefetched values (same operations, same order)
        double pap = tot_s[8];
        for (int k = 0; k < ng; ++k) {
            const double val = tot_s[k] + lam * dt0[k] * pt0[k];
            Ap_tail[k] = val;
            pap += val * pt0[k];
        }
        s->pAp = pap;
        s->alpha = pap > 0 ? rz0 / pap : 0.0;
        if (!(pap > 0)) s->stop = 2;  // breakdown (non-positive curvature / NaN)
    }
}

// tail update, coarse solve y = Einv Z^T r, r.z, beta, stop test.  tot[0] r.zb, tot[1] r.r, tot[2..2+ng) Z_v^T r
// tails_done: the (fused) update kernel has already advanced x / r / zb of the tail entries; only read them here
// Everything the coarse logic reads besides the sums, fetched into LDS BEFORE the sums are taken (the kernel is a single
// block on the critical path of every iteration: its dependent memory round trips are its run time)
struct CoarsePre {
    double rt[6], zt[6], pt[6], apt[6], xt[6], minv[6], alpha, rz, tol2, rr0, tol2z, rz0;
    int stop;
};
__device__ __forceinline__ void coarse_prefetch(CoarsePre &pre, int ng, const double *x_tail, const double *r_tail,
                                                const double *zb_tail, const double *p_tail, const double *Ap_tail,
                                                const Coarse *coarse, const Scal *s) {
    const int tid = threadIdx.x;
    if (tid < ng) {
        pre.rt[tid] = r_tail[tid];
        pre.zt[tid] = zb_tail[tid];
        pre.pt[tid] = p_tail[tid];
        pre.apt[tid] = Ap_tail[tid];
        pre.xt[tid] = x_tail[tid];
        pre.minv[tid] = coarse->tail_minv[tid];
    } else if (tid == 32) {
        pre.alpha = s->alpha;
        pre.rz = s->rz;
        pre.tol2 = coarse->tol2;
        pre.rr0 = s->rr0;
        pre.tol2z = coarse->tol2z;
        pre.rz0 = s->rz0;
        pre.stop = s->stop;
    }
}
// all threads of the block must call it after a __syncthreads() that follows coarse_prefetch; Einv may point to LDS
__device__ __forceinline__ void coarse_logic(const double *tot, int ng, int nm, double *x_tail, double *r_tail, double *zb_tail,
                                             Coarse *coarse, Scal *s, int init, int tails_done, const CoarsePre &pre,
                                             const double *Einv) {
    const int tid = threadIdx.x;
    const int stop = pre.stop;  // uniform
    if (stop) return;
    const int nz = nm + ng;  // tot[2..2+nm) Z_v^T r  (nm = ng for the plain coarse space)
    __shared__ double ztr_s[NZMAX], y_s[NZMAX], acc_s[2];
    if (tid == 0) {
        double rz = tot[0], rr = tot[1];
        const double alpha = pre.alpha;
        for (int k = 0; k < ng; ++k) {
            double rt, zt;
            if (tails_done) {
                rt = pre.rt[k];
                zt = pre.zt[k];
            } else {
                x_tail[k] = pre.xt[k] + alpha * pre.pt[k];
                rt = pre.rt[k] - alpha * pre.apt[k];
                r_tail[k] = rt;
                zt = rt * pre.minv[k];
                zb_tail[k] = zt;
            }
            rz += rt * zt;
            rr += rt * rt;
            ztr_s[nm + k] = rt;
        }
        acc_s[0] = rz;
        acc_s[1] = rr;
    }
    for (int k = tid; k < nm; k += blockDim.x) ztr_s[k] = tot[2 + k];
    __syncthreads();
    // y = Einv Z^T r.  Einv is exactly symmetric (the host symmetrises it): column `row` read as a row is coalesced
    // over the threads.
    if (nz > 16 && blockDim.x >= 256) {
        // enriched space in a large block: thread = (row, one of blockDim/128 column chunks), every load independent
        __shared__ double ypart[8][NZMAX];
        const int row = tid & 127, chunk = tid >> 7, nch = blockDim.x >> 7, per = (nz + nch - 1) / nch;
        if (row < nz) {
            double y = 0.0;
            const int b1 = (chunk + 1) * per < nz ? (chunk + 1) * per : nz;
            for (int b = chunk * per; b < b1; ++b) y += Einv[b * nz + row] * ztr_s[b];
            ypart[chunk][row] = y;
        }
        __syncthreads();
        if (tid < nz) {
            double y = ypart[0][tid];
            for (int q = 1; q < nch; ++q) y += ypart[q][tid];
            y_s[tid] = y;
            coarse->y[tid] = y;
        }
    } else {
        for (int row = tid; row < nz; row += blockDim.x) {  // one row per thread, columns in order
            double y = 0.0;
            for (int b = 0; b < nz; ++b) y += Einv[b * nz + row] * ztr_s[b];
            y_s[row] = y;
            coarse->y[row] = y;
        }
    }
    __syncthreads();
    double zy = 0.0;  // (Z^T r) . y, complete on thread 0
    if (nz > 16) {    // enriched space: the first wave adds the up to 128 products
        if (tid < 64) {
            zy = tid < nz ? ztr_s[tid] * y_s[tid] : 0.0;
            if (tid + 64 < nz) zy += ztr_s[tid + 64] * y_s[tid + 64];
            zy = wave_sum(zy);
        }
    }
    if (tid != 0) return;
    double rz = acc_s[0];
    const double rr = acc_s[1];
    if (nz > 16) rz += zy;
    else
        for (int a = 0; a < nz; ++a) rz += ztr_s[a] * y_s[a];  // the plain space keeps its historical order (= the fused kernels')
    s->rz_new = rz;
    s->rr = rr;
    if (init) {
        s->rz = rz;
        s->tmp[0] = rz;  // the fused kernels keep r.z in tmp[parity of the iteration] (see FusedCg)
        s->rr0 = rr;
        s->rz0 = rz;
        s->beta = 0.0;
    } else {
        s->beta = rz / pre.rz;
        s->rz = rz;
        s->iters += 1;
        if (!(rr > pre.tol2 * pre.rr0) && !(rz > pre.tol2z * pre.rz0)) s->stop = 1;
    }
}
__global__ __launch_bounds__(128) void cg_coarse_logic_kernel(const double *__restrict__ tot, int ng, double *__restrict__ x_tail,
                                                            double *__restrict__ r_tail, double *__restrict__ zb_tail,
                                                            const double *__restrict__ p_tail, const double *__restrict__ Ap_tail,
                                                            Coarse *__restrict__ coarse, Scal *__restrict__ s, int init, int nm) {
    __shared__ CoarsePre pre;
    coarse_prefetch(pre, ng, x_tail, r_tail, zb_tail, p_tail, Ap_tail, coarse, s);
    __syncthreads();
    coarse_logic(tot, ng, nm, x_tail, r_tail, zb_tail, coarse, s, init, 0, pre, coarse->Einv);
}
// mom != null (enriched coarse space): the moments come from the wide rows, slots 0/1 (r.zb, r.r) from `partial`
__global__ __launch_bounds__(1024) void cg_sum_coarse_kernel(const double *__restrict__ partial, int nb, unsigned mask,
                                                             double *__restrict__ tot, int ng, double *__restrict__ x_tail,
                                                             double *__restrict__ r_tail, double *__restrict__ zb_tail,
                                                             const double *__restrict__ p_tail, const double *__restrict__ Ap_tail,
                                                             Coarse *__restrict__ coarse, Scal *__restrict__ s, int init,
                                                             const double *__restrict__ mom, int nm, int tails_done) {
    // the sums stay in LDS (no round trip through `tot`, which nothing else reads on the single-GPU path)
    __shared__ double tot_s[2 + kMomSlots];
    __shared__ CoarsePre pre;
    __shared__ double einv_s[NZMAX * NZMAX];
    (void)tot;
    // E^-1 and the scalars first (coalesced, independent loads): in flight while the partial rows are summed
    const int nz = nm + ng;
    for (int i = threadIdx.x; i < nz * nz; i += blockDim.x) einv_s[i] = coarse->Einv[i];
    coarse_prefetch(pre, ng, x_tail, r_tail, zb_tail, p_tail, Ap_tail, coarse, s);
    block_rows_sum(partial, nb, mask, tot_s, 0);
    if (mom) block_rows_sum_wide(mom, nb, nm, tot_s, 2);
    __syncthreads();
    coarse_logic(tot_s, ng, nm, x_tail, r_tail, zb_tail, coarse, s, init, tails_done, pre, einv_s);
}

// tot[slot] += sum_k a[k] b[k] (w ? w[k] : 1) over the ng replicated tail entries (added on one rank only)
__global__ void add_tail_dot_kernel(double *__restrict__ tot, int slot, const double *__restrict__ a,
                                    const double *__restrict__ b, const double *__restrict__ w, int ng) {
    if (threadIdx.x) return;
    double s = 0.0;
    for (int k = 0; k < ng; ++k) s += a[k] * b[k] * (w ? w[k] : 1.0);
    tot[slot] += s;
}

// halo packing: buf[i*S + e] = vec(idx[i], e); unpacking: vec(first + h, e) = buf[h*S + e]
__global__ __launch_bounds__(NB) void pack_kernel(const double *__restrict__ vec, const int32_t *__restrict__ idx, int64_t cnt,
                                                 int S, double *__restrict__ buf, Strides st) {
    const int64_t t = (int64_t)blockIdx.x * NB + threadIdx.x;
    if (t < cnt * S) buf[t] = vec[(int64_t)idx[t / S] * st.pj + (t % S) * st.pe];
}
__global__ __launch_bounds__(NB) void unpack_kernel(double *__restrict__ vec, int64_t first, int64_t cnt, int S,
                                                   const double *__restrict__ buf, Strides st) {
    const int64_t t = (int64_t)blockIdx.x * NB + threadIdx.x;
    if (t < cnt * S) vec[(first + t / S) * st.pj + (t % S) * st.pe] = buf[t];
}

// dst(j, e) = src(j, e) for a [cnt, width] array between two layouts (row strides sj, element strides se)
__global__ __launch_bounds__(NB) void relayout_kernel(const double *__restrict__ src, int64_t sj, int64_t se,
                                                     double *__restrict__ dst, int64_t dj, int64_t de, int64_t cnt, int width) {
    const int64_t t = (int64_t)blockIdx.x * NB + threadIdx.x;
    if (t < cnt * width) {
        const int64_t j = t % cnt;
        const int e = (int)(t / cnt);
        dst[j * dj + e * de] = src[j * sj + e * se];
    }
}

// partial[b] = sum of a*b*(w?w:1) over the OWNED part of a parameter vector: vertices [0, nown), all S components
__global__ __launch_bounds__(NB) void dot_owned_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                      const double *__restrict__ w, int64_t nown, int S, Strides st,
                                                      double *__restrict__ partial) {
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * NB + threadIdx.x; i < nown * S; i += (int64_t)gridDim.x * NB) {
        const int64_t o = (i % nown) * st.pj + (i / nown) * st.pe;
        s += a[o] * b[o] * (w ? w[o] : 1.0);
    }
    block_sum_to(s, partial);
}

// x += alpha p; r -= alpha Ap; zb = Binv r (per vertex); partials: r.zb, r.r, Z^T r.
// Only the nown owned vertices are updated (halo copies [nown, n) are refreshed by the exchange).
// FUSED (single GPU): the two single-block "sum the block partials, then scalar logic" launches of an iteration
// are folded into their consumers -- every block of cg_update_prec_kernel sums the J^T kernel's partials and
// derives alpha itself, every block of cg_dir_prec_kernel sums the update kernel's partials and solves the
// coarse system itself; block 0 alone writes the shared state, and nothing a block reads is written by another
// block of the same launch: r.z alternates between Scal::tmp[0] and tmp[1] by iteration parity, the tail
// entries are advanced in the update kernel (which does not read them elsewhere) and only read in the
// direction kernel, and the two kernels use separate partial buffers.  A block that sees the stop flag set by
// block 0 of its own launch skips work nobody reads any more.  Same arithmetic in the same order as the
// unfused path: results are bit-identical.
struct FusedCg {
    const double *in_partial;  // block partials to sum: of the J^T kernel (update) / of the update kernel (direction)
    int nb;
    const double *lamd2_tail;
    double *x_tail, *r_tail, *zb_tail, *Ap_tail;
    const double *p_tail;
    int parity;  // which Scal::tmp slot holds r.z (see above); -1: the direction stage is NOT fused, r.z is Scal::rz
    const double *mom;  // direction kernel, enriched space fused (RICHF): the update kernel's block moment rows
};

template <int DIM, bool FUSED, bool RICH, bool P32>
__global__ __launch_bounds__(NB) void cg_update_prec_kernel(Scal *__restrict__ s, const double *__restrict__ P,
                                                           const double *__restrict__ v, int64_t n,
                                                           double *__restrict__ x, double *__restrict__ r,
                                                           double *__restrict__ zb, const double *__restrict__ p,
                                                           const double *__restrict__ Ap, const double *__restrict__ binv,
                                                           double *__restrict__ partial, int64_t nown, Strides st,
                                                           const Coarse *__restrict__ coarse, FusedCg f,
                                                           double *__restrict__ mom, PolyBasis pb) {
    constexpr int S = Lay<DIM>::S, NG = Lay<DIM>::NG, TRI = S * (S + 1) / 2;
    const int64_t j = (int64_t)vblock() * blockDim.x + threadIdx.x;
    double acc[2 + NG], phi[MAXPOLY];
    for (int k = 0; k < 2 + NG; ++k) acc[k] = 0.0;
    for (int m = 0; m < MAXPOLY; ++m) phi[m] = 0.0;
    // The vertex's own blocks do not depend on alpha: their loads are issued FIRST, so the memory latency of the body
    // overlaps the (fused) reduction prologue below instead of following it (10 k vertices: both are latency chains).
    // Every load first, every store last: with run-time strides the compiler cannot tell x[o_e] from x[o_e'] and would
    // otherwise wait for each store before the next load (12 round trips in a row).
    double xj[S], rj[S], pj[S], apj[S], zj[S], Pj_[S], vj_[DIM];
    float lrow[P32 ? TRI : 1];
    const bool mine = j < nown;
    auto load_own = [&]() {
#pragma unroll
        for (int e = 0; e < S; ++e) {
            const int64_t o = j * st.pj + e * st.pe;
            xj[e] = x[o];
            rj[e] = r[o];
            pj[e] = p[o];
            apj[e] = Ap[o];
            Pj_[e] = P[o];
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c) vj_[c] = v[j * st.vj + c * st.vc];
        if constexpr (P32) {
            const float *Lf = (const float *)binv + j;
#pragma unroll
            for (int q = 0; q < TRI; ++q) lrow[q] = Lf[(int64_t)q * n];
        }
    };
    // (only where there IS a prologue: on large meshes, whose sums have their own launch, holding ~200 registers
    // across nothing costs occupancy -- 200 k vertices: 36.7 us with the loads in the body, 48.6 us hoisted)
    if constexpr (FUSED)
        if (mine) load_own();
    double alpha_blk = 0.0;
    bool stop_blk;
    if constexpr (FUSED) {
        __shared__ double tot_s[kSumSlots];
        __shared__ double alpha_sh;
        __shared__ int stop_sh;
        // the scalars of the alpha logic, fetched before the sums they are combined with (thread 0 of every block)
        int stop0 = 0;
        double rz0 = 0.0, lt0[NG], pt0[NG];
        if (threadIdx.x == 0) {
            stop0 = s->stop;
            rz0 = f.parity >= 0 ? s->tmp[f.parity] : s->rz;
            for (int k = 0; k < NG; ++k) {
                lt0[k] = f.lamd2_tail[k];
                pt0[k] = f.p_tail[k];
            }
        }
        block_rows_sum_lds<((1u << NG) - 1u) | (1u << 8)>(f.in_partial, f.nb, tot_s);
        __syncthreads();
        if (threadIdx.x == 0) {  // alpha_logic, by every block for itself
            int stop = stop0;
            double alpha = 0.0;
            if (!stop) {
                double pap = tot_s[8], apt[NG];
                for (int k = 0; k < NG; ++k) {
                    const double val = tot_s[k] + lt0[k] * pt0[k];
                    apt[k] = val;
                    pap += val * pt0[k];
                }
                alpha = pap > 0 ? rz0 / pap : 0.0;
                if (!(pap > 0)) stop = 2;  // breakdown (non-positive curvature / NaN)
                if (blockIdx.x == 0) {
                    s->pAp = pap;
                    s->alpha = alpha;
                    if (stop) s->stop = 2;
                    for (int k = 0; k < NG; ++k) {
                        f.Ap_tail[k] = apt[k];
                        if (stop) continue;
                        f.x_tail[k] += alpha * pt0[k];  // the tail's share of the update (coarse_logic does it when unfused)
                        const double rt = f.r_tail[k] - alpha * apt[k];
                        f.r_tail[k] = rt;
                        f.zb_tail[k] = rt * coarse->tail_minv[k];
                    }
                }
            }
            alpha_sh = alpha;
            stop_sh = stop;
        }
        __syncthreads();
        alpha_blk = alpha_sh;
        stop_blk = stop_sh != 0;
    } else {
        stop_blk = s->stop != 0;
        alpha_blk = s->alpha;
    }
    if (mine && !stop_blk) {
        const double alpha = alpha_blk;
        if constexpr (!FUSED) load_own();
#pragma unroll
        for (int e = 0; e < S; ++e) {
            xj[e] += alpha * pj[e];
            rj[e] = rj[e] - alpha * apj[e];
            zj[e] = 0.0;
        }
        if constexpr (P32) {
            // zb = Li^T (Li r): one pass over the rows of Li, every entry loaded once (fp32) and used twice (fp64 arithmetic)
            int q = 0;
#pragma unroll
            for (int a = 0; a < S; ++a) {
                double ta = 0.0;
#pragma unroll
                for (int b = 0; b <= a; ++b) ta += (double)lrow[q + b] * rj[b];
#pragma unroll
                for (int b = 0; b <= a; ++b) zj[b] += (double)lrow[q + b] * ta;
                q += a + 1;
            }
        } else {
            const double *B = binv + j;
            int q = 0;
#pragma unroll
            for (int a = 0; a < S; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) {
                    const double m = B[(int64_t)(q++) * n];
                    zj[a] += m * rj[b];
                    if (a != b) zj[b] += m * rj[a];
                }
        }
        double z[NG][S];
        zmodes<DIM>(Pj_, vj_, z);
        if constexpr (RICH) poly_eval<DIM>(vj_, pb, phi);
#pragma unroll
        for (int e = 0; e < S; ++e) {
            acc[0] += rj[e] * zj[e];
            acc[1] += rj[e] * rj[e];
#pragma unroll
            for (int k = 0; k < NG; ++k) acc[2 + k] += z[k][e] * rj[e];
        }
#pragma unroll
        for (int e = 0; e < S; ++e) {
            const int64_t o = j * st.pj + e * st.pe;
            x[o] = xj[e];
            r[o] = rj[e];
            zb[o] = zj[e];
        }
    }
    __shared__ double red[NB / 64][2 + NG];
    const int tid = threadIdx.x;
    for (int k = 0; k < 2 + NG; ++k) {
        const double t = wave_sum(acc[k]);
        if ((tid & 63) == 0) red[tid >> 6][k] = t;
    }
    __syncthreads();
    if (tid < 2 + NG) {
        double t = red[0][tid];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t += red[w][tid];
        partial[(j / blockDim.x) * kSumSlots + tid] = t;
    }
    if constexpr (RICH) {  // enriched coarse space: the NG*M moments of the new residual
        double tk[NG];
#pragma unroll
        for (int k = 0; k < NG; ++k) tk[k] = acc[2 + k];
        block_moments<DIM>(tk, phi, pb.M, mom + (j / blockDim.x) * kMomSlots);
    }
}

// What the fit rows of the NEXT operator application need from this kernel when it also forms them (FIT, see below)
struct FitArgs {
    const int32_t *nbr;
    const double *wt;
    int K;
    const double *rs;
    const Globals *g;
    double c_fit;
    double *yfit;
    double *p_out;  // the new direction goes to a second buffer: neighbours still need the old one
    RowW rw;
};

// z = zb + Z y; p = z + beta p
// FIT (single GPU): the kernel also writes the fit rows of J p_new (what fitrow_kernel would compute in the next
// operator application), so that application is normal_kernel alone: one launch and one streamed pass over p less per
// iteration.  A fit row needs p_new of the vertex's 4 influence neighbours, which other threads of this launch are
// only just computing -- each thread recomputes them from the neighbours' zb, p_old, P, v by the same operations in
// the same order (so the rows are bit-identical to fitrow_kernel's), which is why p_new goes to a second buffer.
// RICHF (round 3): the FUSED form for the ENRICHED coarse space (meshes <= 65 536 vertices, one GPU): every block sums the
// update kernel's two scalar slots and its 60 moment columns and solves the 66 x 66 coarse system itself, in the order of
// cg_sum_coarse_kernel's 1024-thread block (16 row groups per moment column, 8 column chunks per row of E+) so that the
// iteration keeps its bits; that kernel -- a single block, 10 us of dependent latencies at 10 k vertices -- is not launched.
template <int DIM, bool FUSED, bool FIT, bool COMPAT, bool RICHF = false>
__global__ __launch_bounds__(NB) void cg_dir_prec_kernel(Scal *__restrict__ s, const Coarse *__restrict__ coarse,
                                                        const double *__restrict__ P, const double *__restrict__ v, int64_t n,
                                                        const double *__restrict__ zb, double *__restrict__ p, int64_t nown,
                                                        Strides st, FusedCg f, PolyBasis pb, FitArgs fa) {
    constexpr int S = Lay<DIM>::S, NG = Lay<DIM>::NG;
    const int nm = FUSED ? NG : NG * pb.M;  // vertex-mode columns of Z (FUSED serves the plain space, RICHF the enriched one)
    static_assert(!(FUSED && RICHF), "one fused form at a time");
    __shared__ double y_s[NZMAX];
    double beta;
    // the vertex's own blocks first: their latency overlaps the (fused) reduction prologue (see cg_update_prec_kernel)
    const int64_t j = (int64_t)vblock() * blockDim.x + threadIdx.x;
    double Pj_[S], vj_[DIM], zbj[S], pj[S];
    const bool mine = j < n && j < nown;
    if (mine) {
#pragma unroll
        for (int e = 0; e < S; ++e) {  // loads first, stores last (see cg_update_prec_kernel)
            const int64_t o = j * st.pj + e * st.pe;
            Pj_[e] = P[o];
            zbj[e] = zb[o];
            pj[e] = p[o];
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c) vj_[c] = v[j * st.vj + c * st.vc];
    }
    if constexpr (FUSED) {
        constexpr int nz = 2 * NG;
        __shared__ double tot_s[kSumSlots], ztr_s[NZMAX], acc_s[2];
        __shared__ double beta_sh;
        __shared__ int stop_sh;
        // scalars and the rows of E^-1, fetched before the sums they meet
        int stop0 = 0;
        double rt0[NG], zt0[NG], rzold0 = 0.0, tol20 = 0.0, rr00 = 0.0, tol2z0 = 0.0, rz00 = 0.0, erow[2 * NG];
        if (threadIdx.x == 0) {
            stop0 = s->stop;
            rzold0 = s->tmp[f.parity];
            tol20 = coarse->tol2;
            rr00 = s->rr0;
            tol2z0 = coarse->tol2z;
            rz00 = s->rz0;
            for (int k = 0; k < NG; ++k) {
                rt0[k] = f.r_tail[k];
                zt0[k] = f.zb_tail[k];
            }
        }
        if (threadIdx.x < nz)
            for (int b = 0; b < nz; ++b) erow[b] = coarse->Einv[threadIdx.x * nz + b];
        block_rows_sum_lds<(1u << (2 + NG)) - 1u>(f.in_partial, f.nb, tot_s);
        __syncthreads();
        if (threadIdx.x == 0) {  // coarse_logic, by every block for itself (the tails were advanced by the update kernel)
            const int stop = stop0;
            if (!stop) {
                double rz = tot_s[0], rr = tot_s[1];
                for (int k = 0; k < NG; ++k) {
                    const double rt = rt0[k], zt = zt0[k];
                    rz += rt * zt;
                    rr += rt * rt;
                    ztr_s[k] = tot_s[2 + k];
                    ztr_s[NG + k] = rt;
                }
                acc_s[0] = rz;
                acc_s[1] = rr;
            }
            stop_sh = stop;
        }
        __syncthreads();
        if (!stop_sh && threadIdx.x < nz) {
            double y = 0.0;
            for (int b = 0; b < nz; ++b) y += erow[b] * ztr_s[b];
            y_s[threadIdx.x] = y;
        }
        __syncthreads();
        if (threadIdx.x == 0 && !stop_sh) {
            double rz = acc_s[0];
            const double rr = acc_s[1];
            for (int a = 0; a < nz; ++a) rz += ztr_s[a] * y_s[a];
            const double bt = rz / rzold0;
            const bool done = !(rr > tol20 * rr00) && !(rz > tol2z0 * rz00);
            if (blockIdx.x == 0) {
                s->rz_new = rz;
                s->rr = rr;
                s->beta = bt;
                s->rz = rz;
                s->tmp[f.parity ^ 1] = rz;
                s->iters += 1;
                if (done) s->stop = 1;
            }
            beta_sh = bt;
            if (done) stop_sh = 1;
        }
        __syncthreads();
        if (stop_sh) return;  // uniform over the block
        beta = beta_sh;
    } else if constexpr (RICHF) {
        const int nz = nm + NG;
        __shared__ double tot_s[kSumSlots], ztr_s[NZMAX], acc_s[2];
        __shared__ double beta_sh;
        __shared__ int stop_sh;
        int stop0 = 0;
        double rt0[NG], zt0[NG], rzold0 = 0.0, tol20 = 0.0, rr00 = 0.0, tol2z0 = 0.0, rz00 = 0.0;
        if (threadIdx.x == 0) {   // scalars, fetched before the sums they meet
            stop0 = s->stop;
            rzold0 = s->tmp[f.parity];
            tol20 = coarse->tol2;
            rr00 = s->rr0;
            tol2z0 = coarse->tol2z;
            rz00 = s->rz0;
            for (int k = 0; k < NG; ++k) {
                rt0[k] = f.r_tail[k];
                zt0[k] = f.zb_tail[k];
            }
        }
        // moment columns: thread = column; block_rows_sum_wide's order (row group g = rows g, g + 16, ... ascending, then the
        // 16 group sums in order), 64 rows in flight
        if (threadIdx.x < kMomSlots && threadIdx.x < nm) {
            double a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = 0.0;
            for (int b0 = 0; b0 < f.nb; b0 += 64) {
                double t[64];
#pragma unroll
                for (int u = 0; u < 64; ++u) t[u] = b0 + u < f.nb ? f.mom[(int64_t)(b0 + u) * kMomSlots + threadIdx.x] : 0.0;
#pragma unroll
                for (int u = 0; u < 64; ++u)
                    if (b0 + u < f.nb) a[u & 15] += t[u];
            }
            double sv = a[0];
#pragma unroll
            for (int q = 1; q < 16; ++q) sv += a[q];
            ztr_s[threadIdx.x] = sv;
        }
        block_rows_sum_lds<0x3u>(f.in_partial, f.nb, tot_s);
        __syncthreads();
        if (threadIdx.x == 0) {  // coarse_logic (tails advanced by the update kernel), by every block for itself
            const int stop = stop0;
            if (!stop) {
                double rz = tot_s[0], rr = tot_s[1];
                for (int k = 0; k < NG; ++k) {
                    const double rt = rt0[k], zt = zt0[k];
                    rz += rt * zt;
                    rr += rt * rt;
                    ztr_s[nm + k] = rt;
                }
                acc_s[0] = rz;
                acc_s[1] = rr;
            }
            stop_sh = stop;
        }
        __syncthreads();
        if (!stop_sh) {
            // y = E+ Z^T r: a row's 8 column chunks in order, as the 1024-thread block adds them
            const int per = (nz + 7) / 8;
            for (int row = threadIdx.x; row < nz; row += blockDim.x) {
                double y = 0.0;
                for (int ch = 0; ch < 8; ++ch) {
                    const int b1 = (ch + 1) * per < nz ? (ch + 1) * per : nz;
                    double yp = 0.0;
                    for (int b = ch * per; b < b1; ++b) yp += coarse->Einv[b * nz + row] * ztr_s[b];
                    y = ch == 0 ? yp : y + yp;
                }
                y_s[row] = y;
            }
        }
        __syncthreads();
        double zy = 0.0;
        if (!stop_sh && threadIdx.x < 64) {
            zy = (int)threadIdx.x < nz ? ztr_s[threadIdx.x] * y_s[threadIdx.x] : 0.0;
            if ((int)threadIdx.x + 64 < nz) zy += ztr_s[threadIdx.x + 64] * y_s[threadIdx.x + 64];
            zy = wave_sum(zy);
        }
        if (threadIdx.x == 0 && !stop_sh) {
            const double rz = acc_s[0] + zy, rr = acc_s[1];
            const double bt = rz / rzold0;
            const bool done = !(rr > tol20 * rr00) && !(rz > tol2z0 * rz00);
            if (blockIdx.x == 0) {
                s->rz_new = rz;
                s->rr = rr;
                s->beta = bt;
                s->rz = rz;
                s->tmp[f.parity ^ 1] = rz;
                s->iters += 1;
                if (done) s->stop = 1;
            }
            beta_sh = bt;
            if (done) stop_sh = 1;
        }
        __syncthreads();
        if (stop_sh) return;  // uniform over the block
        beta = beta_sh;
    } else {
        if (s->stop) return;
        beta = s->beta;
        for (int i = threadIdx.x; i < nm + NG; i += blockDim.x) y_s[i] = coarse->y[i];
        __syncthreads();
    }
    // new direction of one vertex from its blocks (in place in pk)
    auto direction = [&](const double (&Pk)[S], const double (&vk)[DIM], const double (&zbk)[S], double (&pk)[S]) {
        double z[NG][S], yk[NG];
#pragma unroll
        for (int k = 0; k < NG; ++k) yk[k] = y_s[k];
        if (!FUSED && pb.M > 1) {  // y_k(j) = sum_m y[m*NG + k] phi_m(v_j)
            double phi[MAXPOLY];
            poly_eval<DIM>(vk, pb, phi);
#pragma unroll
            for (int m = 1; m < MAXPOLY; ++m) {
                if (m < pb.M) {  // uniform
#pragma unroll
                    for (int k = 0; k < NG; ++k) yk[k] += y_s[m * NG + k] * phi[m];
                }
            }
        }
        zmodes<DIM>(Pk, vk, z);
#pragma unroll
        for (int e = 0; e < S; ++e) {
            double val = zbk[e];
#pragma unroll
            for (int k = 0; k < NG; ++k) val += yk[k] * z[k][e];
            pk[e] = val + beta * pk[e];
        }
    };
    double *pw = FIT ? fa.p_out : p;
    if (j < n) {
        if (j >= nown) return;  // halo copy (never with FIT: single GPU)
        if constexpr (FIT) {
            // the influence neighbours' blocks, all gathers issued before the vertex's own direction is formed
            constexpr int NANG = Lay<DIM>::NANG;
            const int K = fa.K;
            int32_t jj[4];
            double ww[4], Pn[4][S], vn[4][DIM], zn[4][S], pn[4][S];
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
                const int kk = b4 < K ? b4 : K - 1;
                jj[b4] = fa.nbr[j * K + kk];
                ww[b4] = b4 < K ? fa.wt[j * K + kk] : 0.0;
            }
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
#pragma unroll
                for (int e = 0; e < S; ++e) {
                    const int64_t o = (int64_t)jj[b4] * st.pj + e * st.pe;
                    Pn[b4][e] = P[o];
                    zn[b4][e] = zb[o];
                    pn[b4][e] = p[o];
                }
#pragma unroll
                for (int r = 0; r < DIM; ++r) vn[b4][r] = v[(int64_t)jj[b4] * st.vj + r * st.vc];
            }
            double ptail[NG], rsj[DIM];
#pragma unroll
            for (int k = 0; k < NG; ++k) ptail[k] = zb[n * S + k] + y_s[nm + k] + beta * p[n * S + k];
#pragma unroll
            for (int c = 0; c < DIM; ++c) rsj[c] = fa.rs[j * st.vj + c * st.vc];
            double c_fit = fa.c_fit;
            if (fa.rw.f2) c_fit *= fa.rw.f2[j];
            direction(Pj_, vj_, zbj, pj);
#pragma unroll
            for (int e = 0; e < S; ++e) pw[j * st.pj + e * st.pe] = pj[e];
            double acc[DIM];
#pragma unroll
            for (int c = 0; c < DIM; ++c) acc[c] = 0.0;
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
                const double wgt = ww[b4];
                if (wgt == 0.0) continue;
                direction(Pn[b4], vn[b4], zn[b4], pn[b4]);
                const double wb = (DIM == 2 && COMPAT) ? 1.0 : wgt;  // Q7
#pragma unroll
                for (int c = 0; c < DIM; ++c) {
                    double a = 0.0;
#pragma unroll
                    for (int r = 0; r < DIM; ++r) a += (vj_[r] - vn[b4][r]) * pn[b4][r * DIM + c];
                    acc[c] += wgt * a + wb * pn[b4][DIM * DIM + c];
                }
            }
#pragma unroll
            for (int o = 0; o < DIM; ++o) {
                double a = ptail[NANG + o];
#pragma unroll
                for (int c = 0; c < DIM; ++c) a += acc[c] * fa.g->R[c * DIM + o];
                for (int an = 0; an < NANG; ++an) {
                    double b = 0.0;
#pragma unroll
                    for (int c = 0; c < DIM; ++c) b += rsj[c] * fa.g->dR[an][c * DIM + o];
                    a += ptail[an] * b;
                }
                fa.yfit[j * st.fj + o * st.fo] = -c_fit * a;
            }
        } else {
            direction(Pj_, vj_, zbj, pj);
#pragma unroll
            for (int e = 0; e < S; ++e) pw[j * st.pj + e * st.pe] = pj[e];
        }
    } else if (j < n + NG) {
        const int k = (int)(j - n);
        pw[n * S + k] = zb[n * S + k] + y_s[nm + k] + beta * p[n * S + k];
    }
}

// out = W^T-side weighting of a row vector u (rows as nicp_residual): fit rows take their multiplier and
// the landmark row of the same vertex (whose Jacobian row is c_land/c_fit times the fit row), so that
// J^T out = (W J)^T u with the unweighted J^T kernel.  out has no landmark block.
template <int DIM>
__global__ __launch_bounds__(NB) void weight_rows_kernel(const double *__restrict__ u, int64_t n, int64_t ne, double ratio,
                                                        double *__restrict__ out, Strides st, RowW rw) {
    constexpr int NRIG = Lay<DIM>::NRIG;
    const int64_t t = (int64_t)blockIdx.x * NB + threadIdx.x;
    const int64_t o_rig = n * DIM, o_sm = n * DIM + n * NRIG, o_land = o_sm + 2 * DIM * ne;
    if (t < n) {
        const double f = rw.fit ? rw.fit[t] : 1.0, gq = rw.rig ? rw.rig[t] : 1.0;
        const int32_t lk = rw.lm_of ? rw.lm_of[t] : -1;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            double val = f * u[t * st.fj + c * st.fo];
            if (lk >= 0) val += ratio * u[o_land + lk * st.lj + c * st.lo];
            out[t * st.fj + c * st.fo] = val;
        }
#pragma unroll
        for (int m = 0; m < NRIG; ++m) out[o_rig + t * st.rj + m * st.rm] = gq * u[o_rig + t * st.rj + m * st.rm];
    } else if (t < n + ne) {
        const int64_t ed = t - n;
        const double e = rw.edge ? rw.edge[ed] : 1.0;
#pragma unroll
        for (int o = 0; o < 2 * DIM; ++o) out[o_sm + ed * st.sj + o * st.so] = e * u[o_sm + ed * st.sj + o * st.so];
    }
}
// f2[i] = fit_w[i]^2 + ratio2 on landmark vertices
__global__ __launch_bounds__(NB) void f2_kernel(const double *__restrict__ fw, const int32_t *__restrict__ lm_of, double ratio2,
                                               int64_t n, double *__restrict__ f2) {
    const int64_t t = (int64_t)blockIdx.x * NB + threadIdx.x;
    if (t < n) {
        const double f = fw ? fw[t] : 1.0;
        f2[t] = f * f + (lm_of && lm_of[t] >= 0 ? ratio2 : 0.0);
    }
}
__global__ __launch_bounds__(NB) void fill_i32_kernel(int32_t *__restrict__ a, int32_t val, int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * NB + threadIdx.x;
    if (t < n) a[t] = val;
}
// lm_of[vertex[k]] = k; err[0] counts out-of-range vertices, err[1] duplicates
__global__ __launch_bounds__(NB) void landmark_map_kernel(const int32_t *__restrict__ vertex, int64_t nl, int64_t n,
                                                         int32_t *__restrict__ lm_of, int32_t *__restrict__ err) {
    const int64_t k = (int64_t)blockIdx.x * NB + threadIdx.x;
    if (k >= nl) return;
    const int32_t vtx = vertex[k];
    if (vtx < 0 || vtx >= n) {
        atomicAdd(&err[0], 1);
        return;
    }
    if (atomicCAS(&lm_of[vtx], -1, (int32_t)k) != -1) atomicAdd(&err[1], 1);
}

// partial[b][0..dim) = sum (v_j - c), [3] = sum |v_j - c|^2, [4] = count, over the owned vertices
__global__ __launch_bounds__(NB) void vstats_kernel(const double *__restrict__ v, int64_t nown, int dim, Strides st, double c0, double c1,
                                                   double c2, double *__restrict__ partial) {
    double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    const double c[3] = {c0, c1, c2};
    for (int64_t j = (int64_t)blockIdx.x * NB + threadIdx.x; j < nown; j += (int64_t)gridDim.x * NB) {
        for (int k = 0; k < dim; ++k) {
            const double x = v[j * st.vj + k * st.vc] - c[k];
            a[k] += x;
            a[3] += x * x;
        }
        a[4] += 1.0;
    }
    __shared__ double red[NB / 64][5];
    const int tid = threadIdx.x;
    for (int k = 0; k < 5; ++k) {
        const double sv = wave_sum(a[k]);
        if ((tid & 63) == 0) red[tid >> 6][k] = sv;
    }
    __syncthreads();
    if (tid < 5) {
        double sv = red[0][tid];
        for (int w = 1; w < NB / 64; ++w) sv += red[w][tid];
        partial[(int64_t)blockIdx.x * kSumSlots + tid] = sv;
    }
}

static inline int vgrid(const nicp_graph *g, int64_t len) { return (int)((len + g->vb - 1) / g->vb); }
static inline int grid_for(int64_t len, int cap = 1024) {
    int64_t nb = (len + NB - 1) / NB;
    return (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
}

// ------------------------------------------------------------------ host-side driver pieces
struct Dims {
    int dim, S, NG, NRIG;
    int64_t n, e, np, nr;  // np = parameters, nr = residual rows
};
static Dims dims_of(const nicp_graph *g) {
    Dims d;
    d.dim = g->dim;
    d.S = g->dim * g->dim + g->dim;
    d.NG = g->dim == 3 ? 6 : 3;
    d.NRIG = g->dim == 3 ? 6 : 3;
    d.n = g->n;
    d.e = g->e;
    d.np = g->n * d.S + d.NG;
    d.nr = g->n * g->dim + g->n * d.NRIG + 2 * g->dim * g->e + g->dim * g->nl;
    return d;
}

struct Work {
    Globals *glob;
    Scal *scal;
    double *partial, *partial2;            // npart * kSumSlots each (J^T kernel / CG update kernel)
    double *mom;                           // npart * kMomSlots: block moments Z_v^T r of the enriched coarse space
    PolyBasis pb;                          // coarse-space polynomials of this solve (M = 1: plain space)
    double *pre, *dd;                      // [n,dim]
    double *r, *r_trial, *y;               // nr
    double *grad, *col2, *d2, *lamd2;      // np
    double *delta, *rcg, *z, *pcg, *Ap;    // np
    double *pcg2;                          // np: second direction buffer (the fused direction + fit-row kernel ping-pongs)
    double *x_trial;                       // np
    double *binv;                          // n * S(S+1)/2: inverted diagonal blocks
    Coarse *coarse;
    double *tot;                           // 64 reduced sums (the all-reduce buffer when no comm is given)
    double *xs, *vs, *nns, *masks;         // solver-internal (lane-transposed) copies of P, v, nn and the row mask
    Strides st;                            // layout the helpers below index with (reference layout by default)
    RowW rw;                               // row multipliers / landmark rows of the graph (all null: none)
    bool weighted = false;
};

static int get_work(icp_ctx *ctx, nicp_graph *g, Work &w) {
    const Dims d = dims_of(g);
    constexpr size_t coarse_doubles = (sizeof(Coarse) + 7) / 8;
    const size_t small = 1024 + coarse_doubles;  // Globals | Scal | tot (128) | Coarse, in doubles
    const size_t tri = (size_t)d.S * (d.S + 1) / 2;
    // one row of block partials per block of the per-vertex kernels: sized by the mesh, not by a fixed cap
    const size_t npart = std::max<size_t>(kMaxPartials, (size_t)((d.n + d.NG + g->vb - 1) / g->vb) + 1);
    const size_t need = small + 2 * npart * kSumSlots + npart * kMomSlots + 4 * (size_t)d.n * d.dim + 4 * (size_t)d.nr + 12 * (size_t)d.np +
                        (size_t)d.n * tri + 64;
    if (g->ws_doubles < need) {
        if (g->ws) {
            ICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            ICP_HIP_CHECK(hipFree(g->ws));
            g->ws = nullptr;
        }
        if (hipMalloc((void **)&g->ws, need * sizeof(double)) != hipSuccess) {
            set_error("nicp workspace: hipMalloc(%zu) failed", need * sizeof(double));
            return ICP_ERR_NOMEM;
        }
        g->ws_doubles = need;
    }
    double *p = g->ws;
    w.glob = (Globals *)p;
    w.scal = (Scal *)(p + 256);
    w.tot = p + 512;
    w.coarse = (Coarse *)(p + 1024);
    static_assert(sizeof(Globals) <= 256 * 8 && sizeof(Scal) <= 256 * 8, "small area layout");
    p += small;
    w.partial = p; p += npart * kSumSlots;
    w.partial2 = p; p += npart * kSumSlots;
    w.mom = p; p += npart * kMomSlots;
    w.pre = p; p += (size_t)d.n * d.dim;
    w.dd = p; p += (size_t)d.n * d.dim;
    w.r = p; p += d.nr;
    w.r_trial = p; p += d.nr;
    w.y = p; p += d.nr;
    double **np_bufs[] = {&w.grad, &w.col2, &w.d2, &w.lamd2, &w.delta, &w.rcg, &w.z, &w.pcg, &w.Ap, &w.x_trial, &w.pcg2};
    for (double **b : np_bufs) {
        *b = p;
        p += d.np;
    }
    w.binv = p; p += (size_t)d.n * tri;
    w.xs = p; p += d.np;
    w.vs = p; p += (size_t)d.n * d.dim;
    w.nns = p; p += (size_t)d.n * d.dim;
    w.masks = p;
    w.st = d.dim == 3 ? strides_aos<3>() : strides_aos<2>();
    w.pb = PolyBasis{{0.0, 0.0, 0.0}, 1.0, 1};
    w.rw = RowW();
    w.rw.fit = g->fw;
    w.rw.rig = g->gw;
    w.rw.edge = g->ew;
    w.rw.f2 = (g->fw || g->nl) ? g->f2 : nullptr;
    w.rw.lm_of = g->nl ? g->lm_of : nullptr;
    w.rw.lm_pos = g->lm_pos;
    w.rw.c_land = g->c_land;
    w.rw.nl = g->nl;
    w.weighted = g->fw || g->gw || g->ew || g->nl;
    return ICP_OK;
}

#define NICP_DISPATCH(DIMV, COMPATV, CALL)           \
    do {                                             \
        if ((DIMV) == 3) {                           \
            if (COMPATV) { constexpr int D = 3; constexpr bool C = true; CALL; }   \
            else { constexpr int D = 3; constexpr bool C = false; CALL; }          \
        } else {                                     \
            if (COMPATV) { constexpr int D = 2; constexpr bool C = true; CALL; }   \
            else { constexpr int D = 2; constexpr bool C = false; CALL; }          \
        }                                            \
    } while (0)

// linearisation state at P: globals, pre/dd
static int linearise(icp_ctx *ctx, nicp_graph *g, Work &w, const double *P, const double *v) {
    hipStream_t st = ctx->stream;
    if (g->dim == 3) {
        hipLaunchKernelGGL(globals_kernel<3>, dim3(1), dim3(1), 0, st, P, g->n, w.glob);
        hipLaunchKernelGGL(deform_kernel<3>, dim3(grid_for(g->n, 1 << 30)), dim3(NB), 0, st, P, g->nbr, g->wt, g->k, v, g->n, w.glob, w.pre, w.dd, w.st);
    } else {
        hipLaunchKernelGGL(globals_kernel<2>, dim3(1), dim3(1), 0, st, P, g->n, w.glob);
        hipLaunchKernelGGL(deform_kernel<2>, dim3(grid_for(g->n, 1 << 30)), dim3(NB), 0, st, P, g->nbr, g->wt, g->k, v, g->n, w.glob, w.pre, w.dd, w.st);
    }
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

static int residual_at(icp_ctx *ctx, nicp_graph *g, Work &w, const double *P, const double *v, const double *nn,
                       const double *c, bool compat, double *r_out) {
    if (int rc = linearise(ctx, g, w, P, v)) return rc;
    const int nbk = grid_for(g->n + g->e, 1 << 30);
    NICP_DISPATCH(g->dim, compat,
                  hipLaunchKernelGGL((residual_kernel<D, C>), dim3(nbk), dim3(NB), 0, ctx->stream, P, v, nn, w.dd, g->edges,
                                     g->n, g->e, c[0], c[1], c[2], r_out, w.st, w.rw));
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

static int refuse_compat_weights(const Work &w, bool compat) {
    if (compat && w.weighted) {
        set_error("nicp: NICP_REFERENCE_COMPAT has no meaning on a graph with row weights or landmarks");
        return ICP_ERR_UNSUPPORTED;
    }
    return ICP_OK;
}

// (c_land / c_fit)^2 folded into the fit rows: refresh g->f2 for this call's c_fit
static int prepare_weights(icp_ctx *ctx, nicp_graph *g, Work &w, const double *c) {
    if (!w.rw.f2) return ICP_OK;
    if (g->nl && !(c[0] > 0.0)) {
        set_error("nicp: landmark rows need c_fit > 0 (they ride on the fit rows as c_land/c_fit)");
        return ICP_ERR_ARG;
    }
    const double ratio = g->nl ? g->c_land / c[0] : 0.0;
    hipLaunchKernelGGL(f2_kernel, dim3(grid_for(g->n, 1 << 30)), dim3(NB), 0, ctx->stream, (const double *)g->fw, w.rw.lm_of, ratio * ratio,
                       g->n, g->f2);
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

// out = W-weighted copy of the row vector u for the J^T kernel (see weight_rows_kernel); only on weighted graphs
static int weight_rows(icp_ctx *ctx, nicp_graph *g, Work &w, const double *c, const double *u, double *out) {
    const double ratio = g->nl ? g->c_land / c[0] : 0.0;
    const int nbk = grid_for(g->n + g->e, 1 << 30);
    if (g->dim == 3) hipLaunchKernelGGL(weight_rows_kernel<3>, dim3(nbk), dim3(NB), 0, ctx->stream, u, g->n, g->e, ratio, out, w.st, w.rw);
    else hipLaunchKernelGGL(weight_rows_kernel<2>, dim3(nbk), dim3(NB), 0, ctx->stream, u, g->n, g->e, ratio, out, w.st, w.rw);
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

// y = W J p (sq = 0) or W^2 J p (sq = 1) at the linearisation held in w (call linearise first)
static int apply_J(icp_ctx *ctx, nicp_graph *g, Work &w, const double *P, const double *v, const double *c, bool compat,
                   const double *p, double *y, int sq) {
    const int nbk = vgrid(g, g->n + g->e);
    const double *rs = compat ? w.dd : w.pre;  // Q6
    NICP_DISPATCH(g->dim, compat,
                  hipLaunchKernelGGL((jp_kernel<D, C>), dim3(nbk), dim3(g->vb), 0, ctx->stream, P, p, g->nbr, g->wt, g->k, v, rs,
                                     g->edges, g->n, g->e, w.glob, c[0], c[1], c[2], y, w.st, w.rw, sq));
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

// How the solver's sums and halo copies are completed: alone (comm == nullptr) or as one rank of a
// vertex-sharded solve (nicp_solve_dist).  Local vertices are [owned | halo]; tot is the 64-double
// buffer every reduction lands in before the (optional) all-reduce over ranks.
struct Par {
    const nicp_comm *comm = nullptr;
    icp_ctx *ctx = nullptr;
    int64_t nown = 0;
    double *tot = nullptr;
    const double *row_mask = nullptr;
    bool sharded() const { return comm && (comm->world > 1 || (comm->flags & ICP_COMM_FORCE)); }
    // collectives issued by the library itself on the context's stream (rccl.hip): nothing on the host per collective
    bool native() const { return sharded() && comm->rccl != nullptr; }
    int rank() const { return comm ? comm->rank : 0; }
};

static Par par_single(icp_ctx *ctx, const nicp_graph *g, Work &w) {
    Par p;
    p.ctx = ctx;
    p.nown = g->n;
    p.tot = w.tot;
    return p;
}

static int allreduce(const Par &par, int count) {
    if (!par.sharded()) return ICP_OK;
    if (par.native()) return rccl_allreduce_sum(par.comm->rccl, par.tot, count, par.ctx->stream);
    if (int rc = par.comm->allreduce_sum(par.comm->user, count)) {
        set_error("nicp_solve_dist: the all-reduce callback failed (%d)", rc);
        return ICP_ERR_COMM;
    }
    return ICP_OK;
}

// refresh the halo copies [nown, n) of a per-vertex vector from their owners
static int exchange(icp_ctx *ctx, const nicp_graph *g, const Work &w, const Par &par, double *vec) {
    if (!par.sharded()) return ICP_OK;
    const nicp_comm *cm = par.comm;
    const int S = g->dim * g->dim + g->dim;
    if (cm->n_send > 0)
        hipLaunchKernelGGL(pack_kernel, dim3(grid_for(cm->n_send * S, 1 << 30)), dim3(NB), 0, ctx->stream, (const double *)vec,
                           cm->send_idx_dev, cm->n_send, S, cm->send_buf_dev, w.st);
    ICP_HIP_CHECK(hipGetLastError());
    if (par.native()) {
        if (int rc = rccl_exchange(cm->rccl, cm->send_buf_dev, cm->send_counts_host, cm->recv_buf_dev, cm->recv_counts_host, S, ctx->stream))
            return rc;
    } else if (int rc = cm->exchange(cm->user)) {
        set_error("nicp_solve_dist: the halo-exchange callback failed (%d)", rc);
        return ICP_ERR_COMM;
    }
    const int64_t nh = g->n - par.nown;
    if (nh > 0)
        hipLaunchKernelGGL(unpack_kernel, dim3(grid_for(nh * S, 1 << 30)), dim3(NB), 0, ctx->stream, vec, par.nown, nh, S,
                           (const double *)cm->recv_buf_dev, w.st);
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

// out = J^T u (+ lam d2 .* p_in) on the owned vertices; square: column sums of squares of J.
// finish_tail: also complete the NG tail entries (sum over blocks, all-reduce over ranks);
// otherwise the block partials stay in w.partial (slots 0..NG tail, slot 8 p.Ap) for the caller.
static int apply_JT(icp_ctx *ctx, nicp_graph *g, Work &w, const Par &par, const double *P, const double *v, const double *c,
                    bool compat, const double *u, const double *lam_d2, const double *p_in, double lam, double *out,
                    bool square, bool finish_tail = true) {
    const Dims d = dims_of(g);
    const int nbk = vgrid(g, g->n);
    const double *rs = compat ? w.dd : w.pre;
    if (square) {
        NICP_DISPATCH(g->dim, compat,
                      hipLaunchKernelGGL((jtu_kernel<D, C, true>), dim3(nbk), dim3(g->vb), 0, ctx->stream, P, u, g->nbr, g->wt, g->k, v,
                                         rs, g->edges, g->rev_ptr, g->rev_i, g->rev_w, g->inc_ptr, g->inc_ent, g->inc_other, g->n, g->e, w.glob,
                                         c[0], c[1], c[2], lam_d2, p_in, lam, out, w.partial, par.nown, w.st, w.rw));
    } else {
        NICP_DISPATCH(g->dim, compat,
                      hipLaunchKernelGGL((jtu_kernel<D, C, false>), dim3(nbk), dim3(g->vb), 0, ctx->stream, P, u, g->nbr, g->wt, g->k, v,
                                         rs, g->edges, g->rev_ptr, g->rev_i, g->rev_w, g->inc_ptr, g->inc_ent, g->inc_other, g->n, g->e, w.glob,
                                         c[0], c[1], c[2], lam_d2, p_in, lam, out, w.partial, par.nown, w.st, w.rw));
    }
    ICP_HIP_CHECK(hipGetLastError());
    if (!finish_tail) return ICP_OK;
    hipLaunchKernelGGL(rows_sum_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double *)w.partial, nbk, (1u << d.NG) - 1u, par.tot, 0);
    if (int rc = allreduce(par, d.NG)) return rc;
    hipLaunchKernelGGL(tail_write_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double *)par.tot, 0, d.NG,
                       lam_d2 ? lam_d2 + d.n * d.S : nullptr, p_in ? p_in + d.n * d.S : nullptr, lam, out + d.n * d.S);
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

// out = J^T W^2 J p + lam lam_d2 .* p on the local vertices, block partials of the tail sums (slots 0..NG) and of
// p . out (slot 8) in w.partial -- what apply_J(sq = 1) followed by apply_JT(finish_tail = false) leaves, without
// the row vector in between (fitrow_kernel + normal_kernel; ICP_NICP_FUSEDOP=0: the two-kernel pair through w.y)
static int apply_A(icp_ctx *ctx, nicp_graph *g, Work &w, const Par &par, const double *P, const double *v, const double *c,
                   bool compat, const double *p, const double *lam_d2, double lam, double *out, hipEvent_t mid = nullptr,
                   bool fit_rows_ready = false, const int *stop = nullptr) {
    const char *env_op = getenv("ICP_NICP_FUSEDOP");
    const bool fused_op = fit_rows_ready || !(env_op && atoi(env_op) == 0);
    if (!fused_op) {
        if (int rc = apply_J(ctx, g, w, P, v, c, compat, p, w.y, 1)) return rc;
        if (mid) (void)hipEventRecord(mid, ctx->stream);
        return apply_JT(ctx, g, w, par, P, v, c, compat, w.y, lam_d2, p, lam, out, false, false);
    }
    const int nbk = vgrid(g, g->n);
    const double *rs = compat ? w.dd : w.pre;  // Q6
    if (!fit_rows_ready)  // (the fused direction kernel of the previous iteration has written them already)
        NICP_DISPATCH(g->dim, compat,
                      hipLaunchKernelGGL((fitrow_kernel<D, C>), dim3(nbk), dim3(g->vb), 0, ctx->stream, p, g->nbr, g->wt, g->k, v, rs, g->n,
                                         w.glob, c[0], w.y, w.st, w.rw, stop));
    if (mid) (void)hipEventRecord(mid, ctx->stream);  // nicp_profile_cg: boundary between the two kernels
    NICP_DISPATCH(g->dim, compat,
                  hipLaunchKernelGGL((normal_kernel<D, C>), dim3(nbk), dim3(g->vb), 0, ctx->stream, P, p, (const double *)w.y, v, rs,
                                     g->rev_ptr, g->rev_i, g->rev_w, g->inc_ptr, g->inc_ent, g->inc_other, g->n, w.glob, c[0], c[1],
                                     c[2], lam_d2, lam, out, w.partial, par.nown, w.st, w.rw, stop));
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

// tot[slot] = sum a.b.w over len entries (this rank's share; all-reduced later by the caller)
static int dot_slot(icp_ctx *ctx, Work &w, const Par &par, const double *a, const double *b, const double *wv, int64_t len, int slot) {
    const int nbk = grid_for(len);
    hipLaunchKernelGGL(dot_kernel, dim3(nbk), dim3(NB), 0, ctx->stream, a, b, wv, len, w.partial);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(NB), 0, ctx->stream, (const double *)w.partial, nbk, par.tot + slot);
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}
// parameter-vector dot: owned vertices + (on rank 0) the replicated tail
static int pdot_slot(icp_ctx *ctx, const nicp_graph *g, Work &w, const Par &par, const double *a, const double *b, const double *wv, int slot) {
    const Dims d = dims_of(g);
    const int nbk = grid_for(par.nown * d.S);
    hipLaunchKernelGGL(dot_owned_kernel, dim3(nbk), dim3(NB), 0, ctx->stream, a, b, wv, par.nown, d.S, w.st, w.partial);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(NB), 0, ctx->stream, (const double *)w.partial, nbk, par.tot + slot);
    if (par.rank() == 0)
        hipLaunchKernelGGL(add_tail_dot_kernel, dim3(1), dim3(64), 0, ctx->stream, par.tot, slot, a + d.n * d.S, b + d.n * d.S,
                           wv ? wv + d.n * d.S : nullptr, d.NG);
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}
static int read_tot(icp_ctx *ctx, const Par &par, int count, double *host) {
    ICP_HIP_CHECK(hipMemcpyAsync(host, par.tot, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

// inverse of a small symmetric positive (semi-)definite matrix by Gauss-Jordan with partial pivoting
static bool invert_spd(const double *E, int nz, double *out) {
    double a[NZMAX][2 * NZMAX];
    for (int i = 0; i < nz; ++i)
        for (int j = 0; j < nz; ++j) {
            const double v = 0.5 * (E[i * nz + j] + E[j * nz + i]);
            if (!(v == v) || std::isinf(v)) return false;
            a[i][j] = v;
            a[i][nz + j] = i == j ? 1.0 : 0.0;
        }
    double scale = 0.0;
    for (int i = 0; i < nz; ++i) scale = std::max(scale, std::fabs(a[i][i]));
    if (!(scale > 0.0)) return false;
    for (int c = 0; c < nz; ++c) {
        int piv = c;
        for (int r = c + 1; r < nz; ++r)
            if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (std::fabs(a[piv][c]) < 1e-14 * scale) return false;
        if (piv != c)
            for (int k = 0; k < 2 * nz; ++k) std::swap(a[piv][k], a[c][k]);
        const double inv = 1.0 / a[c][c];
        for (int k = 0; k < 2 * nz; ++k) a[c][k] *= inv;
        for (int r = 0; r < nz; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            if (f != 0.0)
                for (int k = 0; k < 2 * nz; ++k) a[r][k] -= f * a[c][k];
        }
    }
    for (int i = 0; i < nz; ++i)
        for (int j = 0; j < nz; ++j) out[i * nz + j] = 0.5 * (a[i][nz + j] + a[j][nz + i]);
    return true;
}

// Inverse of the coarse matrix on its numerically independent columns.  The enriched coarse space can be rank
// deficient -- on a sphere x^2 + y^2 + z^2 is constant, the global parameters duplicate the constant modes -- so
// columns are picked by pivoted Cholesky on the unit-diagonal scaling of E until the residual diagonal falls below
// `tol`, E is inverted on that set and the rows / columns of the dropped modes are zero: Z Einv Z^T stays symmetric
// positive semi-definite, which is all the preconditioner needs.  Returns the number of columns kept.
static int invert_spd_rank_revealing(const double *E, int nz, double *out, double tol = 1e-9) {
    std::vector<double> A((size_t)nz * nz), L((size_t)nz * nz, 0.0), dres(nz), sc(nz);
    for (int i = 0; i < nz; ++i) {
        const double dgi = E[i * nz + i];
        sc[i] = (dgi > 0.0 && std::isfinite(dgi)) ? 1.0 / std::sqrt(dgi) : 0.0;
    }
    for (int i = 0; i < nz; ++i)
        for (int j = 0; j < nz; ++j) {
            const double val = 0.5 * (E[i * nz + j] + E[j * nz + i]) * sc[i] * sc[j];
            if (!std::isfinite(val)) {
                std::fill(out, out + (size_t)nz * nz, 0.0);
                return 0;
            }
            A[(size_t)i * nz + j] = val;
        }
    for (int i = 0; i < nz; ++i) dres[i] = A[(size_t)i * nz + i];  // 1 (or 0 for a dead column)
    std::vector<int> kept;
    std::vector<char> used(nz, 0);
    for (int step = 0; step < nz; ++step) {
        int piv = -1;
        for (int i = 0; i < nz; ++i)
            if (!used[i] && (piv < 0 || dres[i] > dres[piv])) piv = i;
        if (piv < 0 || !(dres[piv] > tol)) break;
        used[piv] = 1;
        const int kcol = (int)kept.size();
        kept.push_back(piv);
        const double root = std::sqrt(dres[piv]);
        for (int i = 0; i < nz; ++i) {
            if (used[i] && i != piv) continue;
            double val = A[(size_t)i * nz + piv];
            for (int j = 0; j < kcol; ++j) val -= L[(size_t)i * nz + j] * L[(size_t)piv * nz + j];
            L[(size_t)i * nz + kcol] = val / root;
        }
        for (int i = 0; i < nz; ++i)
            if (!used[i]) dres[i] -= L[(size_t)i * nz + kcol] * L[(size_t)i * nz + kcol];
    }
    std::fill(out, out + (size_t)nz * nz, 0.0);
    const int nk = (int)kept.size();
    if (nk == 0) return 0;
    std::vector<double> Ek((size_t)nk * nk), Eki((size_t)nk * nk);
    for (int a = 0; a < nk; ++a)
        for (int b = 0; b < nk; ++b) Ek[(size_t)a * nk + b] = A[(size_t)kept[a] * nz + kept[b]];
    if (!invert_spd(Ek.data(), nk, Eki.data())) return 0;
    for (int a = 0; a < nk; ++a)
        for (int b = 0; b < nk; ++b) out[(size_t)kept[a] * nz + kept[b]] = Eki[(size_t)a * nk + b] * (sc[kept[a]] * sc[kept[b]]);  // exactly symmetric
    return nk;
}

static int read_scal(icp_ctx *ctx, const Scal *dev, Scal &host) {
    ICP_HIP_CHECK(hipMemcpyAsync(&host, dev, sizeof(Scal), hipMemcpyDeviceToHost, ctx->stream));
    ICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

}  // namespace icp

using namespace icp;

// ===================================================================== C-ABI
extern "C" int nicp_knn(icp_ctx *ctx, const double *v, int64_t n, int dim, const double *q, int64_t nq, int k,
                        int32_t *idx_out, double *dist_out) {
    ICP_REQUIRE(ctx && v && q && idx_out && dist_out, "null pointer");
    ICP_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
    ICP_REQUIRE(n > 0 && nq >= 0 && (k == 6 || k == 8), "k must be 6 or 8 (use 8 and truncate for other k <= 8)");
    // n < k: the n points come back in order and the remaining slots hold index -1 / distance +inf (the reference's
    // heap returns its size n and then pops past the empty heap, nicp3d.cpp:130-135; numpy's argsort returns n entries)
    if (nq == 0) return ICP_OK;
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    {   // large point sets: the exact grid search (same keys, same tie rule); ICP_KNN_GRID_MIN moves the switch
        const char *e = getenv("ICP_KNN_GRID_MIN");
        const long long gmin = e ? atoll(e) : 8192;
        if (n >= gmin && nq >= 256) {
            GridBuild gb;
            if (int rc = grid_build(ctx, v, n, dim, gb)) return rc;
            if (gb.usable) return grid_knn(ctx, gb, q, nq, k, idx_out, dist_out);  // non-finite points: full scan below
        }
    }
    const unsigned nb = (unsigned)((nq + NB - 1) / NB);
    hipStream_t st = ctx->stream;
    if (dim == 3 && k == 6) hipLaunchKernelGGL((knn_rows_kernel<3, 6>), dim3(nb), dim3(NB), 0, st, v, n, q, nq, idx_out, dist_out);
    else if (dim == 3) hipLaunchKernelGGL((knn_rows_kernel<3, 8>), dim3(nb), dim3(NB), 0, st, v, n, q, nq, idx_out, dist_out);
    else if (k == 6) hipLaunchKernelGGL((knn_rows_kernel<2, 6>), dim3(nb), dim3(NB), 0, st, v, n, q, nq, idx_out, dist_out);
    else hipLaunchKernelGGL((knn_rows_kernel<2, 8>), dim3(nb), dim3(NB), 0, st, v, n, q, nq, idx_out, dist_out);
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

extern "C" int nicp_knn_weights(icp_ctx *ctx, const double *v, int64_t n, int dim, int num_nn, int32_t *nbr_out,
                                double *wt_out) {
    ICP_REQUIRE(ctx && v && nbr_out && wt_out, "null pointer");
    ICP_REQUIRE(num_nn == 4, "num_nn must be 4 (nicp_3d.py:17, nicp3d.cpp:25)");
    ICP_REQUIRE(n >= num_nn + 2, "need at least num_nn+2 vertices");
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    const int K = num_nn + 2;
    if (int rc = ensure(ctx, ctx->work_a, (size_t)n * K * sizeof(int32_t))) return rc;
    if (int rc = ensure(ctx, ctx->work_b, (size_t)n * K * sizeof(double))) return rc;
    int32_t *kidx = (int32_t *)ctx->work_a.ptr;
    double *kdist = (double *)ctx->work_b.ptr;
    if (int rc = nicp_knn(ctx, v, n, dim, v, n, K, kidx, kdist)) return rc;
    hipLaunchKernelGGL(knn_weights_kernel<4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, kidx, kdist, n,
                       nbr_out, wt_out);
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

extern "C" int nicp_graph_destroy(icp_ctx *ctx, nicp_graph *g) {
    if (!g) return ICP_OK;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    void *ptrs[] = {g->nbr, g->wt, g->edges, g->rev_ptr, g->rev_ent, g->inc_ptr, g->inc_ent, g->rev_i, g->rev_w, g->inc_other, g->ws,
                    g->fw, g->gw, g->ew, g->lm_v, g->lm_pos, g->lm_of, g->f2};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete g;
    return ICP_OK;
}

extern "C" int nicp_graph_create(icp_ctx *ctx, const int32_t *nbr_dev, const double *wt_dev, int64_t n, int num_nn,
                                 const int32_t *edges_dev, int64_t e, int dim, nicp_graph **out) {
    ICP_REQUIRE(ctx && nbr_dev && wt_dev && out, "null pointer");
    ICP_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
    ICP_REQUIRE(n > 0 && num_nn > 0 && num_nn <= 8 && e >= 0 && (e == 0 || edges_dev), "bad sizes");
    ICP_REQUIRE(n * num_nn < 0x7fffffff && 2 * e < 0x7fffffff, "mesh too large for 32-bit adjacency entries");
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<int32_t> nbr((size_t)n * num_nn), edges((size_t)e * 2);
    std::vector<double> wth((size_t)n * num_nn);
    ICP_HIP_CHECK(hipMemcpyAsync(nbr.data(), nbr_dev, nbr.size() * 4, hipMemcpyDeviceToHost, st));
    ICP_HIP_CHECK(hipMemcpyAsync(wth.data(), wt_dev, wth.size() * 8, hipMemcpyDeviceToHost, st));
    if (e) ICP_HIP_CHECK(hipMemcpyAsync(edges.data(), edges_dev, edges.size() * 4, hipMemcpyDeviceToHost, st));
    ICP_HIP_CHECK(hipStreamSynchronize(st));
    for (int32_t x : nbr) ICP_REQUIRE(x >= 0 && x < n, "neighbour index out of range");
    for (int32_t x : edges) ICP_REQUIRE(x >= 0 && x < n, "edge vertex out of range");
    // reverse adjacency by counting sort (entries ascending within each row -> deterministic sums)
    std::vector<int32_t> rev_ptr(n + 1, 0), rev_ent(nbr.size()), inc_ptr(n + 1, 0), inc_ent(edges.size());
    for (int32_t x : nbr) rev_ptr[x + 1]++;
    for (int64_t i = 0; i < n; ++i) rev_ptr[i + 1] += rev_ptr[i];
    {
        std::vector<int32_t> cur(rev_ptr.begin(), rev_ptr.end() - 1);
        for (size_t t = 0; t < nbr.size(); ++t) rev_ent[cur[nbr[t]]++] = (int32_t)t;
    }
    for (int32_t x : edges) inc_ptr[x + 1]++;
    for (int64_t i = 0; i < n; ++i) inc_ptr[i + 1] += inc_ptr[i];
    {
        std::vector<int32_t> cur(inc_ptr.begin(), inc_ptr.end() - 1);
        for (size_t t = 0; t < edges.size(); ++t) inc_ent[cur[edges[t]]++] = (int32_t)t;  // t = e*2+end
    }
    std::vector<int32_t> rev_i(rev_ent.size()), inc_other(inc_ent.size());
    std::vector<double> rev_w(rev_ent.size());
    for (size_t t = 0; t < rev_ent.size(); ++t) {
        rev_i[t] = rev_ent[t] / num_nn;
        rev_w[t] = wth[rev_ent[t]];
    }
    for (size_t t = 0; t < inc_ent.size(); ++t) inc_other[t] = edges[(size_t)inc_ent[t] ^ 1];  // entry ed*2+end -> ed*2+(1-end)
    nicp_graph *g = new nicp_graph();
    g->n = n;
    g->e = e;
    g->dim = dim;
    g->k = num_nn;
    g->vb = n <= 65536 ? 64 : 256;
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        if (hipMalloc(dst, bytes ? bytes : 8) != hipSuccess) return ICP_ERR_NOMEM;
        if (bytes && hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return ICP_ERR_HIP;
        return ICP_OK;
    };
    int rc = ICP_OK;
    if (hipMalloc((void **)&g->nbr, nbr.size() * 4) != hipSuccess || hipMalloc((void **)&g->wt, nbr.size() * 8) != hipSuccess) rc = ICP_ERR_NOMEM;
    if (!rc && hipMemcpyAsync(g->nbr, nbr_dev, nbr.size() * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = ICP_ERR_HIP;
    if (!rc && hipMemcpyAsync(g->wt, wt_dev, nbr.size() * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = ICP_ERR_HIP;
    if (!rc) rc = up((void **)&g->edges, edges.data(), edges.size() * 4);
    if (!rc) rc = up((void **)&g->rev_ptr, rev_ptr.data(), rev_ptr.size() * 4);
    if (!rc) rc = up((void **)&g->rev_ent, rev_ent.data(), rev_ent.size() * 4);
    if (!rc) rc = up((void **)&g->inc_ptr, inc_ptr.data(), inc_ptr.size() * 4);
    if (!rc) rc = up((void **)&g->inc_ent, inc_ent.data(), inc_ent.size() * 4);
    if (!rc) rc = up((void **)&g->rev_i, rev_i.data(), rev_i.size() * 4);
    if (!rc) rc = up((void **)&g->rev_w, rev_w.data(), rev_w.size() * 8);
    if (!rc) rc = up((void **)&g->inc_other, inc_other.data(), inc_other.size() * 4);
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = ICP_ERR_HIP;
    if (rc) {
        set_error("nicp_graph_create: device allocation / upload failed");
        nicp_graph_destroy(ctx, g);
        return rc;
    }
    *out = g;
    return ICP_OK;
}

// ---- extended energy: row multipliers and landmark rows (no counterpart in the reference)
static int replace_copy(icp_ctx *ctx, double **slot, const double *src_dev, int64_t count) {
    if (*slot) {
        ICP_HIP_CHECK(hipFree(*slot));
        *slot = nullptr;
    }
    if (!src_dev || count <= 0) return ICP_OK;
    if (hipMalloc((void **)slot, count * sizeof(double)) != hipSuccess) {
        set_error("nicp weights: hipMalloc(%lld doubles) failed", (long long)count);
        return ICP_ERR_NOMEM;
    }
    ICP_HIP_CHECK(hipMemcpyAsync(*slot, src_dev, count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return ICP_OK;
}
static int ensure_f2(nicp_graph *g) {
    if (g->f2) return ICP_OK;
    if (hipMalloc((void **)&g->f2, g->n * sizeof(double)) != hipSuccess) {
        set_error("nicp weights: hipMalloc failed");
        return ICP_ERR_NOMEM;
    }
    return ICP_OK;
}

extern "C" int nicp_set_weights(icp_ctx *ctx, nicp_graph *g, const double *fit_w_dev, const double *rigid_w_dev,
                                const double *edge_w_dev) {
    ICP_REQUIRE(ctx && g, "null handle");
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    ICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // earlier work may still read the old arrays
    if (int rc = replace_copy(ctx, &g->fw, fit_w_dev, g->n)) return rc;
    if (int rc = replace_copy(ctx, &g->gw, rigid_w_dev, g->n)) return rc;
    if (int rc = replace_copy(ctx, &g->ew, edge_w_dev, g->e)) return rc;
    if (g->fw)
        if (int rc = ensure_f2(g)) return rc;
    ICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // the caller may free its arrays on return
    return ICP_OK;
}

extern "C" int nicp_set_landmarks(icp_ctx *ctx, nicp_graph *g, const int32_t *vertex_dev, const double *pos_dev, int64_t nl,
                                  double c_land) {
    ICP_REQUIRE(ctx && g, "null handle");
    ICP_REQUIRE(nl >= 0 && nl <= g->n && (nl == 0 || (vertex_dev && pos_dev)), "bad landmark arguments");
    ICP_REQUIRE(nl == 0 || (c_land >= 0.0 && c_land == c_land), "c_land must be a non-negative number");
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ICP_HIP_CHECK(hipStreamSynchronize(st));
    if (g->lm_v) ICP_HIP_CHECK(hipFree(g->lm_v));
    g->lm_v = nullptr;
    if (int rc = replace_copy(ctx, &g->lm_pos, nullptr, 0)) return rc;
    g->nl = 0;
    g->c_land = 0.0;
    if (nl == 0) return ICP_OK;
    if (!g->lm_of && hipMalloc((void **)&g->lm_of, (g->n + 2) * sizeof(int32_t)) != hipSuccess) {
        set_error("nicp landmarks: hipMalloc failed");
        return ICP_ERR_NOMEM;
    }
    if (int rc = ensure_f2(g)) return rc;
    if (hipMalloc((void **)&g->lm_v, nl * sizeof(int32_t)) != hipSuccess) {
        set_error("nicp landmarks: hipMalloc failed");
        return ICP_ERR_NOMEM;
    }
    ICP_HIP_CHECK(hipMemcpyAsync(g->lm_v, vertex_dev, nl * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (int rc = replace_copy(ctx, &g->lm_pos, pos_dev, nl * g->dim)) return rc;
    // lm_of[n], then two error counters behind it
    hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for(g->n, 1 << 30)), dim3(NB), 0, st, g->lm_of, -1, g->n);
    ICP_HIP_CHECK(hipMemsetAsync(g->lm_of + g->n, 0, 2 * sizeof(int32_t), st));
    hipLaunchKernelGGL(landmark_map_kernel, dim3(grid_for(nl, 1 << 30)), dim3(NB), 0, st, (const int32_t *)g->lm_v, nl, g->n, g->lm_of,
                       g->lm_of + g->n);
    ICP_HIP_CHECK(hipGetLastError());
    int32_t bad[2] = {0, 0};
    ICP_HIP_CHECK(hipMemcpyAsync(bad, g->lm_of + g->n, sizeof(bad), hipMemcpyDeviceToHost, st));
    ICP_HIP_CHECK(hipStreamSynchronize(st));
    if (bad[0] || bad[1]) {
        set_error("nicp_set_landmarks: %d vertex ids out of range, %d repeated", bad[0], bad[1]);
        return ICP_ERR_ARG;  // g->nl stays 0: the graph is left without landmarks
    }
    g->nl = nl;
    g->c_land = c_land;
    return ICP_OK;
}

extern "C" int nicp_rows(const nicp_graph *g, int64_t *rows_out) {
    ICP_REQUIRE(g && rows_out, "null pointer");
    *rows_out = dims_of(g).nr;
    return ICP_OK;
}

#define NICP_ARGS(ctx, g)                                     \
    ICP_REQUIRE(ctx != nullptr && g != nullptr, "null handle"); \
    ICP_HIP_CHECK(hipSetDevice(ctx->device));                 \
    Work w;                                                   \
    if (int rc_ = get_work(ctx, g, w)) return rc_

extern "C" int nicp_deform(icp_ctx *ctx, nicp_graph *g, const double *P, const double *v, double *out) {
    NICP_ARGS(ctx, g);
    ICP_REQUIRE(P && v && out, "null pointer");
    if (int rc = linearise(ctx, g, w, P, v)) return rc;
    ICP_HIP_CHECK(hipMemcpyAsync(out, w.dd, (size_t)g->n * g->dim * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return ICP_OK;
}

extern "C" int nicp_residual(icp_ctx *ctx, nicp_graph *g, const double *P, const double *v, const double *nn,
                             const double *c3_host, unsigned flags, double *r_out) {
    NICP_ARGS(ctx, g);
    ICP_REQUIRE(P && v && nn && c3_host && r_out, "null pointer");
    if (int rc = refuse_compat_weights(w, flags & 1u)) return rc;
    return residual_at(ctx, g, w, P, v, nn, c3_host, flags & 1u, r_out);
}

extern "C" int nicp_jvp(icp_ctx *ctx, nicp_graph *g, const double *P, const double *v, const double *c3_host,
                        unsigned flags, const double *p_in, double *y_out) {
    NICP_ARGS(ctx, g);
    ICP_REQUIRE(P && v && c3_host && p_in && y_out, "null pointer");
    if (int rc = refuse_compat_weights(w, flags & 1u)) return rc;
    if (int rc = linearise(ctx, g, w, P, v)) return rc;
    return apply_J(ctx, g, w, P, v, c3_host, flags & 1u, p_in, y_out, 0);
}

extern "C" int nicp_jtvp(icp_ctx *ctx, nicp_graph *g, const double *P, const double *v, const double *c3_host,
                         unsigned flags, const double *u_in, double *g_out) {
    NICP_ARGS(ctx, g);
    ICP_REQUIRE(P && v && c3_host && u_in && g_out, "null pointer");
    if (int rc = refuse_compat_weights(w, flags & 1u)) return rc;
    if (int rc = linearise(ctx, g, w, P, v)) return rc;
    if (w.weighted) {
        if (g->nl) ICP_REQUIRE(c3_host[0] > 0.0, "landmark rows need c_fit > 0");
        if (int rc = weight_rows(ctx, g, w, c3_host, u_in, w.y)) return rc;
        u_in = w.y;
    }
    return apply_JT(ctx, g, w, par_single(ctx, g, w), P, v, c3_host, flags & 1u, u_in, nullptr, nullptr, 0.0, g_out, false);
}

// Levenberg-Marquardt with two-level preconditioned CG; par says whether this is the whole mesh
// or one rank's shard.  Every host decision below is taken from all-reduced sums, so the ranks
// of a sharded solve follow the same control flow without further agreement.
static int relayout(icp_ctx *ctx, const double *src, int64_t sj, int64_t se, double *dst, int64_t dj, int64_t de, int64_t cnt,
                    int width) {
    if (cnt <= 0) return ICP_OK;
    hipLaunchKernelGGL(relayout_kernel, dim3(grid_for(cnt * width, 1 << 30)), dim3(NB), 0, ctx->stream, src, sj, se, dst, dj, de, cnt,
                       width);
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

static int solve_impl(icp_ctx *ctx, nicp_graph *g, Work &w, const Par &par_in, double *P_inout, const double *v_ext,
                      const double *nn_ext, const double *c3_host, unsigned flags, const nicp_solve_opts *opts_in,
                      nicp_solve_stats *stats_out, nicp_cg_profile *prof = nullptr) {
    {   // A/B switch for the block order of the CG kernels (vblock): written only when the variable is set
        static const int run_env = getenv("ICP_NICP_RUN") ? atoi(getenv("ICP_NICP_RUN")) : -1;
        if (run_env >= 0) ICP_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_nicp_run), &run_env, sizeof(int)));
    }
    const bool compat = flags & 1u;
    const Dims d = dims_of(g);
    nicp_solve_opts o = {1.49012e-8, 1.49012e-8, 0.0, 1e-3, 0.0, 0, 0, 0};
    if (opts_in) o = *opts_in;
    // leastsq's own limit (nicp_3d.py:351 passes no maxfev: MINPACK's default with an analytic Jacobian is
    // 100 * (n + 1) residual evaluations); max_lm_iters > 0 additionally caps the LM iterations
    const long long maxfev = 100ll * ((long long)d.np + 1);
    const int max_lm = o.max_lm_iters > 0 ? o.max_lm_iters : 0x7fffffff;
    // cg_tol <= 0: automatic.  The LM loop re-evaluates the gradient at every step, so the linear solves need no more
    // accuracy than the step's model deserves: ten times the square root of the tighter LM tolerance, at most 1e-3.
    // Measured on the whole C3 schedule (15 outer iterations, final energy against the one reached with 1e-4):
    //   1e-4: 423 k CG iterations, 16.1 s | 1e-3: 242 k, 9.1 s, energy equal to 2e-10, same number of outer iterations,
    //   1 746 LM steps against 1 717 | 3e-3: 194 k, 7.5 s, but the energy drifts by 6e-4 | 1e-2: 2 275 LM steps, 22 s.
    // (Stopping on the preconditioned residual instead needs 3e-3 for the same energy and is slower there: 11.6 s.)
    const bool cg_tol_auto = !(o.cg_tol > 0.0);
    if (cg_tol_auto) {
        const double lt = std::min(o.ftol > 0 ? o.ftol : 1.0, o.xtol > 0 ? o.xtol : 1.0);
        o.cg_tol = std::max(1e-12, std::min(1e-3, 10.0 * std::sqrt(lt)));
    }
    if (o.max_cg_iters <= 0) o.max_cg_iters = (int)std::min<int64_t>(20000, 4 * d.np + 100);
    hipStream_t st = ctx->stream;
    const double *c = c3_host;
    Scal hs;
    double ht[8];
    nicp_solve_stats stt = {0, 0, 0, 0, 0, 0, 0};
    const int nbp = grid_for(d.np);
    const int64_t nown = par_in.nown;
    int coarse_switch_to = 1;  // coarse-space size to move to once CG gets long (see below)
    int e_lag = 8, e_age = -1, e_M = 0;  // coarse-matrix reuse: rebuilt every 8th linear solve (ICP_NICP_ELAG overrides)
    if (const char *e = getenv("ICP_NICP_ELAG")) e_lag = std::max(1, atoi(e));
    const bool trace = getenv("ICP_NICP_TRACE") != nullptr && par_in.rank() == 0;  // one line per LM trial on stderr
    // ICP_NICP_TIMING: host wall time of the sections of a solve (synchronising after each: development only)
    const bool timing = getenv("ICP_NICP_TIMING") != nullptr && par_in.rank() == 0;
    double tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto tnow = [&]() -> double {
        if (!timing) return 0.0;
        (void)hipStreamSynchronize(ctx->stream);
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    double tlast = tnow();
    auto tick = [&](int k) {
        if (!timing) return;
        const double t = tnow();
        tsec[k] += t - tlast;
        tlast = t;
    };

    // ---- solver-internal, lane-transposed copies of the inputs (see Strides)
    if (int rc = refuse_compat_weights(w, compat)) return rc;
    if (int rc = prepare_weights(ctx, g, w, c)) return rc;
    // Accelerated Levenberg-Marquardt iteration (round 3; nicp_solve_opts::accel, ICP_NICP_ACCEL=0/1 overrides "automatic").
    // The energy's valley is curved -- the rigidity rows are quadratic in A with weight c_rigid^2 -- so a first-order step
    // is short: the self-consistent C3 schedule took 1 746 LM steps.  Geodesic acceleration (Transtrum & Sethna) adds the
    // second-order correction d2 = -1/2 (J^T J + lam D^2)^-1 J^T r''(d1, d1), with the directional second derivative taken
    // as the central difference r(x + d1) - 2 r(x) + r(x - d1) -- exact where the residual is quadratic in the
    // parameters, i.e. everywhere but the three global angles -- and is kept when |D d2| <= 0.75 |D d1|.  Measured on C3:
    // 464 LM steps instead of 1 746 for the same energies, and, what makes it pay, steps that tolerate INEXACT linear
    // solves: with the automatic CG tolerance of 1e-3 the CG iterations stay at 250 k (each solve is longer), with 0.3 they
    // fall to a third (without the correction a tolerance of 1e-2 already needs 2 275 LM steps, 22 s).  So with accel the
    // automatic CG tolerance starts at 0.3 (ICP_NICP_ETA), and a solve that meets its ftol / xtol test under it goes on with
    // a tight one (1e-2) until it meets the test again: the returned point has passed MINPACK's tests on accurate steps.
    // Automatic = on in the self-consistent mode (in the reference's mode the Jacobian is not the residual's derivative,
    // and the step sequence there is pinned to the plain iteration: DESIGN.md 'NICP solve parity').
    bool accel = o.accel > 0 || (o.accel == 0 && !compat);
    if (o.accel == 0)
        if (const char *e = getenv("ICP_NICP_ACCEL")) accel = atoi(e) != 0;
    if (prof) accel = false;
    static const double eta_env = getenv("ICP_NICP_ETA") ? atof(getenv("ICP_NICP_ETA")) : 0.3;
    // (the accurate steps of an accelerated solve: ten times the plain iteration's automatic tolerance -- measured on the
    //  first C3 solve against the minimiser: vertices within 5.5e-9 with 1e-2, 5.4e-9 with 1e-3, 1.6e-7 without the
    //  confirmation, 5.0e-7 for the plain iteration; benchmarks/accel_accuracy_probe.py)
    const double cg_tol_tight = (accel && cg_tol_auto) ? std::min(1e-2, 10.0 * o.cg_tol) : o.cg_tol;
    double cg_tol_cur = (accel && cg_tol_auto && eta_env > cg_tol_tight) ? eta_env : cg_tol_tight;
    const Strides ext = w.st, soa = strides_soa(d.n, d.e, g->nl);
    Par par = par_in;
    if (int rc = relayout(ctx, P_inout, ext.pj, ext.pe, w.xs, soa.pj, soa.pe, d.n, d.S)) return rc;
    ICP_HIP_CHECK(hipMemcpyAsync(w.xs + d.n * d.S, P_inout + d.n * d.S, d.NG * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (int rc = relayout(ctx, v_ext, ext.vj, ext.vc, w.vs, soa.vj, soa.vc, d.n, d.dim)) return rc;
    if (int rc = relayout(ctx, nn_ext, ext.vj, ext.vc, w.nns, soa.vj, soa.vc, d.n, d.dim)) return rc;
    if (par.row_mask) {
        const int64_t o_rig = d.n * d.dim, o_sm = d.n * d.dim + d.n * d.NRIG;
        if (int rc = relayout(ctx, par.row_mask, ext.fj, ext.fo, w.masks, soa.fj, soa.fo, d.n, d.dim)) return rc;
        if (int rc = relayout(ctx, par.row_mask + o_rig, ext.rj, ext.rm, w.masks + o_rig, soa.rj, soa.rm, d.n, d.NRIG)) return rc;
        if (int rc = relayout(ctx, par.row_mask + o_sm, ext.sj, ext.so, w.masks + o_sm, soa.sj, soa.so, d.e, 2 * d.dim)) return rc;
        const int64_t o_land = o_sm + 2 * d.dim * d.e;
        if (int rc = relayout(ctx, par.row_mask + o_land, ext.lj, ext.lo, w.masks + o_land, soa.lj, soa.lo, g->nl, d.dim)) return rc;
        par.row_mask = w.masks;
    }
    w.st = soa;
    double *x = w.xs;
    const double *v = w.vs, *nn = w.nns;
    // ---- coarse-space polynomials: centre and radius of the (whole) mesh from all-reduced sums, so every rank of
    // a sharded solve builds the same basis.  Default (meshes of >= 2048 vertices): start with the plain space and
    // switch to the quadratics for the rest of the solve once a linear solve has needed more than 400 CG iterations
    // (the 54 extra columns of E cost about 75 iterations' worth of launches per LM step and repay from ~200 on);
    // ICP_NICP_COARSE = 1 | 4 | 10 (3-D) / 1 | 3 | 6 (2-D) fixes the space instead.
    {
        const int nbs = grid_for(nown);
        double cen[3] = {0.0, 0.0, 0.0}, hv[5];
        for (int pass = 0; pass < 2; ++pass) {
            hipLaunchKernelGGL(vstats_kernel, dim3(nbs), dim3(NB), 0, st, v, nown, d.dim, w.st, cen[0], cen[1], cen[2], w.partial);
            hipLaunchKernelGGL(rows_sum_kernel, dim3(1), dim3(1024), 0, st, (const double *)w.partial, nbs, 0x1fu, par.tot, 0);
            if (int rc = allreduce(par, 5)) return rc;
            if (int rc = read_tot(ctx, par, 5, hv)) return rc;
            if (pass == 0 && hv[4] > 0)
                for (int k = 0; k < d.dim; ++k) cen[k] = hv[k] / hv[4];
        }
        const double rms = hv[4] > 0 ? std::sqrt(hv[3] / hv[4]) : 0.0;
        PolyBasis pb = {{cen[0], cen[1], cen[2]}, (rms > 0.0 && std::isfinite(rms)) ? 0.5 / rms : 1.0, 1};
        const int mmax = d.dim == 3 ? 10 : 6, mlin = d.dim == 3 ? 4 : 3;
        int M = 1;
        coarse_switch_to = hv[4] >= 2048.0 ? mmax : 1;
        if (const char *e = getenv("ICP_NICP_COARSE")) {
            const int want = atoi(e);
            M = want >= mmax ? mmax : (want >= mlin ? mlin : 1);
            coarse_switch_to = M;
        }
        if (!std::isfinite(cen[0] + cen[1] + cen[2])) M = coarse_switch_to = 1;
        pb.M = M;
        w.pb = pb;
    }
    if (int rc = residual_at(ctx, g, w, x, v, nn, c, compat, w.r)) return rc;
    if (int rc = dot_slot(ctx, w, par, w.r, w.r, par.row_mask, d.nr, 0)) return rc;
    if (int rc = allreduce(par, 1)) return rc;
    if (int rc = read_tot(ctx, par, 1, ht)) return rc;
    double f2 = ht[0];
    stt.err0 = f2;
    stt.nfev = 1;
    tick(0);  // set-up: layouts, mesh statistics, first residual
    double lam = o.lambda0, nu = 2.0;
    int info = 0;
    bool first_scaling = true;
    // hipGraph replay of the CG iterations: single GPU only (the sharded iteration calls back into the host).
    // Default: meshes of >= 1024 vertices; ICP_NICP_GRAPH=0 / 1 forces it off / on.
    struct CgGraph {
        hipGraphExec_t exec = nullptr;
        bool failed = false;
        int M = 0;
        ~CgGraph() {
            if (exec) (void)hipGraphExecDestroy(exec);
        }
    } cg;
    // fused sum+logic prologues pay on small meshes only (10 k vertices: +3 %; 200 k: -5 %, 784 blocks each re-reading the
    // partials); ICP_NICP_FUSED=0 / 1 forces the two single-block sum+logic launches / the fused kernels
    bool fused_enabled = g->vb == 64;
    if (const char *e = getenv("ICP_NICP_FUSED")) fused_enabled = atoi(e) != 0;
    // vertex blocks of the preconditioner as an fp32 inverse Cholesky factor (default) or the packed fp64 inverse
    bool prec32 = true;
    if (const char *e = getenv("ICP_NICP_PREC32")) prec32 = atoi(e) != 0;
    bool use_graph = !par.sharded() && d.n >= 1024;
    if (const char *e = getenv("ICP_NICP_GRAPH")) use_graph = !par.sharded() && atoi(e) != 0;
    // Sharded solve with library-side RCCL: nothing of an iteration runs on the host any more, so the chunk can be
    // captured like the single-GPU one.  Opt-in (ICP_NICP_GRAPH_DIST=1): capturing RCCL's launches has only been run on
    // a one-rank communicator (tests/test_gpu_nicp_dist.py), no multi-GPU node was available to this build.
    if (par.native())
        if (const char *e = getenv("ICP_NICP_GRAPH_DIST")) use_graph = atoi(e) != 0 && d.n >= 1024;
    if (use_graph) {
        if (!ctx->side && hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) != hipSuccess) use_graph = false;
        if (use_graph && !ctx->side_ev && hipEventCreateWithFlags(&ctx->side_ev, hipEventDisableTiming) != hipSuccess) use_graph = false;
        if (!use_graph) (void)hipGetLastError();
    }
    for (int it = 0; it < max_lm && !info && stt.nfev < maxfev; ++it) {
        stt.lm_iters = it + 1;
        if (!(f2 == f2) || std::isinf(f2)) {
            set_error("nicp_solve: residual is not finite");
            return ICP_ERR_NUMERIC;
        }
        // linearise at x (w.r holds r(x) and w.pre/dd/glob are those of x after residual_at)
        if (int rc = linearise(ctx, g, w, x, v)) return rc;
        const double *r_for_grad = w.r;  // gradient (W J)^T r = J^T (W r [+ landmark rows on their fit rows])
        if (w.weighted) {
            if (int rc = weight_rows(ctx, g, w, c, w.r, w.y)) return rc;
            r_for_grad = w.y;
        }
        if (int rc = apply_JT(ctx, g, w, par, x, v, c, compat, r_for_grad, nullptr, nullptr, 0.0, w.grad, false)) return rc;
        if (int rc = apply_JT(ctx, g, w, par, x, v, c, compat, w.r, nullptr, nullptr, 0.0, w.col2, true)) return rc;
        const double cg_tol_now = cg_tol_cur;
        tick(1);  // linearisation + gradient
        bool accepted = false;
        bool relinearise = false;
        while (!accepted && !info && stt.nfev < maxfev) {
            if (relinearise)  // a rejected trial left the linearisation state at x_trial
                if (int rc = linearise(ctx, g, w, x, v)) return rc;
            hipLaunchKernelGGL(scaling_kernel, dim3(nbp), dim3(NB), 0, st, w.col2, w.d2, w.lamd2, lam, first_scaling ? 1 : 0, d.np);
            first_scaling = false;
            // ---- two-level preconditioner for (J^T J + lam D^2): inverted vertex blocks + coarse space
            const int nm = d.NG * w.pb.M;  // vertex-mode columns of Z in this linear solve
            const int nbv = vgrid(g, nown), nbt = vgrid(g, d.n + d.NG), nz = nm + d.NG;
            const bool rich = w.pb.M > 1;
            const int nbj = vgrid(g, d.n);  // blocks of the J^T kernel (all local vertices)
            const unsigned ng_mask = (1u << d.NG) - 1u;
            double *x_tail = w.delta + d.n * d.S, *r_tail = w.rcg + d.n * d.S, *zb_tail = w.z + d.n * d.S;
            double *Ap_tail = w.Ap + d.n * d.S;
            // direction buffers: the fused direction + fit-row kernel (dirfit) writes the new direction into the other one
            double *pbuf[2] = {w.pcg, w.pcg2};
            int pcur = 0;
            const double *d2_tail = w.d2 + d.n * d.S;
            NICP_DISPATCH(g->dim, compat,
                          hipLaunchKernelGGL((block_diag_kernel<D, C>), dim3((unsigned)((d.n + 63) / 64)), dim3(64), 0, st, (const double *)x,
                                             (const double *)g->wt, g->k, v, (const int32_t *)g->edges, (const int32_t *)g->rev_ptr,
                                             (const int32_t *)g->rev_ent, (const int32_t *)g->inc_ptr, (const int32_t *)g->inc_ent, d.n,
                                             (const Globals *)w.glob, c[0], c[1], c[2], (const double *)w.col2, (const double *)w.d2, lam,
                                             w.binv, w.coarse, w.st, w.rw, prec32 ? 1 : 0));
            ICP_HIP_CHECK(hipGetLastError());
            // The coarse matrix may lag: any symmetric positive (semi-)definite E^+ keeps the preconditioner valid, and
            // between consecutive LM steps of a long solve E barely moves, so it is rebuilt every `e_lag`-th linear
            // solve only (and whenever the coarse space changes); the vertex blocks are always current.
            const bool rebuild_E = e_age < 0 || e_age >= e_lag || e_M != w.pb.M;
            if (rebuild_E) e_age = 0;
            ++e_age;
            e_M = w.pb.M;
            for (int a = 0; rebuild_E && a < nz; ++a) {  // E[:, a] = Z^T (J^T J + lam D^2) z_a
                if (g->dim == 3) hipLaunchKernelGGL(zcol_kernel<3>, dim3(nbt), dim3(g->vb), 0, st, (const double *)x, v, d.n, a, w.z, w.st, w.pb);
                else hipLaunchKernelGGL(zcol_kernel<2>, dim3(nbt), dim3(g->vb), 0, st, (const double *)x, v, d.n, a, w.z, w.st, w.pb);
                if (int rc = apply_A(ctx, g, w, par, x, v, c, compat, w.z, w.d2, lam, w.Ap)) return rc;
                hipLaunchKernelGGL(rows_sum_kernel, dim3(1), dim3(1024), 0, st, (const double *)w.partial, nbj, ng_mask, par.tot, nm + 2);
                if (g->dim == 3) hipLaunchKernelGGL(zt_kernel<3>, dim3(nbv), dim3(g->vb), 0, st, (const double *)x, v, nown, (const double *)w.Ap, w.partial, w.st, w.mom, w.pb);
                else hipLaunchKernelGGL(zt_kernel<2>, dim3(nbv), dim3(g->vb), 0, st, (const double *)x, v, nown, (const double *)w.Ap, w.partial, w.st, w.mom, w.pb);
                if (rich) hipLaunchKernelGGL(rows_sum_wide_kernel, dim3(1), dim3(1024), 0, st, (const double *)w.mom, nbv, nm, par.tot, 0);
                else hipLaunchKernelGGL(rows_sum_kernel, dim3(1), dim3(1024), 0, st, (const double *)w.partial, nbv, ng_mask, par.tot, 0);
                if (int rc = allreduce(par, std::max(16, nm + 2 + d.NG))) return rc;
                hipLaunchKernelGGL(ecol_logic_kernel, dim3(1), dim3(128), 0, st, (const double *)par.tot, d.NG, nm, d2_tail, lam, a, w.coarse);
            }
            ICP_HIP_CHECK(hipGetLastError());
            tick(1);
            if (rebuild_E) {
                double E[NZMAX * NZMAX], Einv[NZMAX * NZMAX];
                ICP_HIP_CHECK(hipMemcpyAsync(E, w.coarse->E, sizeof(double) * nz * nz, hipMemcpyDeviceToHost, st));
                ICP_HIP_CHECK(hipStreamSynchronize(st));
                if (rich) {
                    (void)invert_spd_rank_revealing(E, nz, Einv);  // dependent modes dropped; all dropped: block level alone
                } else if (!invert_spd(E, nz, Einv)) {
                    std::memset(Einv, 0, sizeof(Einv));  // fall back to the block level alone
                }
                ICP_HIP_CHECK(hipMemcpyAsync(w.coarse->Einv, Einv, sizeof(double) * nz * nz, hipMemcpyHostToDevice, st));
                ICP_HIP_CHECK(hipStreamSynchronize(st));  // the source is on this stack frame
            }
            {
                // [tol2, tol2z].  The stop test is on the residual |r| / |g|.  With the reference's constants (c_rigid 1000
                // against c_fit 0.1) the system is badly scaled, and a small |r| then says little about the soft (fit)
                // components of the step: tests/fuzz_nicp.py found LM steps 2e-6 away from the dense solve at cg_tol
                // 1e-13.  A caller who asks for a tight solve (cg_tol < 1e-6) therefore also gets the PRECONDITIONED
                // residual r.z / r0.z0 held to the same bound (the step then matches the refined dense solve to 1e-9 in all
                // 420 fuzz cases).  The default tolerance keeps the plain test: the LM loop corrects an inexact step
                // with the next one, and the stricter test costs 24-39 % more CG iterations for the same LM history (at cg_tol
                // 1e-4: C3 reference mode 9 232 -> 11 441, self-consistent 423 k -> 587 k).  ICP_NICP_STOP = 0 / 1 forces
                // either, 2 tests the preconditioned residual alone (measured: no better, see the automatic cg_tol above).
                static const int stop_mode = getenv("ICP_NICP_STOP") ? atoi(getenv("ICP_NICP_STOP")) : -1;
                const bool both = stop_mode < 0 ? cg_tol_now < 1e-6 : stop_mode != 0;
                double tol2[2] = {cg_tol_now * cg_tol_now, both ? cg_tol_now * cg_tol_now : HUGE_VAL};
                if (stop_mode == 2) tol2[0] = HUGE_VAL;
                ICP_HIP_CHECK(hipMemcpyAsync(&w.coarse->tol2, tol2, sizeof(tol2), hipMemcpyHostToDevice, st));
                ICP_HIP_CHECK(hipStreamSynchronize(st));  // both sources are on this stack frame
            }
            tick(2);  // vertex blocks + coarse matrix
            // ---- preconditioned CG on (J^T J + lam D^2) delta = -grad
            // dirfit (ICP_NICP_DIRFIT=1; single GPU, the reference's 4 influence neighbours): the direction kernel also forms
            // the next fit rows.  OFF by default -- measured per CG iteration: 10 k vertices 34.0 -> 33.6 us (the launch it
            // saves costs as much as the four neighbour blocks it recomputes), 200 k vertices 138 -> 166 us (312 B of
            // gathers per neighbour instead of 120: the direction kernel goes from 17 to 63 us).  Bit-identical either way.
            bool dirfit = false;
            if (const char *e = getenv("ICP_NICP_DIRFIT")) dirfit = !par.sharded() && g->k <= 4 && atoi(e) != 0;
            if (const char *e = getenv("ICP_NICP_FUSEDOP")) dirfit = dirfit && atoi(e) != 0;
            // start of a solve with right-hand side scale * vec: delta = 0, r = rhs
            auto cg_reset = [&](double scale, const double *vec) -> int {
                ICP_HIP_CHECK(hipMemsetAsync(w.delta, 0, d.np * sizeof(double), st));
                ICP_HIP_CHECK(hipMemsetAsync(w.pcg, 0, d.np * sizeof(double), st));
                ICP_HIP_CHECK(hipMemsetAsync(w.Ap, 0, d.np * sizeof(double), st));
                hipLaunchKernelGGL(axpby_kernel, dim3(nbp), dim3(NB), 0, st, scale, vec, 0.0, (const double *)nullptr, w.rcg, d.np);
                ICP_HIP_CHECK(hipMemsetAsync(w.scal, 0, sizeof(Scal), st));
                return ICP_OK;
            };
            if (int rc = cg_reset(-1.0, w.grad)) return rc;
            const double *lamd2_tail = w.lamd2 + d.n * d.S;
            const bool fused = !par.sharded() && fused_enabled && !rich;  // see FusedCg (plain coarse space only)
            const bool mixed = !par.sharded() && fused_enabled && rich;   // enriched space: alpha stage fused ...
            // (off by default: measured SLOWER -- whole C3 schedule 4.63 -> 5.53 s, same 110 991 iterations bit for bit: 157 blocks
            //  of 64 threads each re-summing 157 x 60 moments cost more than the one 1024-thread block they replace)
            static const int richf_env = getenv("ICP_NICP_FUSEDRICH") ? atoi(getenv("ICP_NICP_FUSEDRICH")) : 0;
            const bool richf = mixed && richf_env != 0 && !dirfit;        // ... and the coarse stage too (RICHF), or its own launch
            // slots of the update kernel's partial rows that the sums read: r.zb, r.r (+ the NG plain moments)
            const unsigned upd_mask = rich ? 0x3u : (1u << (2 + d.NG)) - 1u;
            hipEvent_t *pev = nullptr;  // nicp_profile_cg: events after every launch of an iteration
            auto mark = [&](int k) {
                if (pev) (void)hipEventRecord(pev[k], st);
            };
            auto precondition = [&](int init, int parity) -> int {
                const bool fz = (fused || richf) && !init;  // both stages fused
                const bool fzu = (fused || mixed) && !init;  // the update kernel derives alpha itself
                double *pk = pbuf[pcur], *p_tail = pk + d.n * d.S;
                FusedCg fa = {w.partial, nbj, lamd2_tail, x_tail, r_tail, zb_tail, Ap_tail, p_tail, (fused || richf) ? parity : -1, nullptr};
                FusedCg fb = {w.partial2, nbv, lamd2_tail, x_tail, r_tail, zb_tail, Ap_tail, p_tail, parity, w.mom};
                FitArgs fit = {g->nbr, g->wt, g->k, compat ? w.dd : w.pre, w.glob, c[0], w.y, pbuf[pcur ^ 1], w.rw};
                double *upd_partial = fzu ? w.partial2 : w.partial;
#define NICP_UPD(D, F) if (rich) { if (prec32) NICP_UPD2(D, F, true, true); else NICP_UPD2(D, F, true, false); } else { if (prec32) NICP_UPD2(D, F, false, true); else NICP_UPD2(D, F, false, false); }
#define NICP_UPD2(D, F, R, Q) hipLaunchKernelGGL((cg_update_prec_kernel<D, F, R, Q>), dim3(nbv), dim3(g->vb), 0, st, w.scal, (const double *)x, v, d.n, w.delta, w.rcg, w.z, (const double *)pk, (const double *)w.Ap, (const double *)w.binv, upd_partial, nown, w.st, (const Coarse *)w.coarse, fa, w.mom, w.pb)
#define NICP_DIR2(D, F, T, C) hipLaunchKernelGGL((cg_dir_prec_kernel<D, F, T, C>), dim3(nbt), dim3(g->vb), 0, st, w.scal, (const Coarse *)w.coarse, (const double *)x, v, d.n, (const double *)w.z, pk, nown, w.st, fb, w.pb, fit)
#define NICP_DIR(D, F) do { if (dirfit) { if (compat) NICP_DIR2(D, F, true, true); else NICP_DIR2(D, F, true, false); } else NICP_DIR2(D, F, false, false); } while (0)
#define NICP_DIRR(D) hipLaunchKernelGGL((cg_dir_prec_kernel<D, false, false, false, true>), dim3(nbt), dim3(g->vb), 0, st, w.scal, (const Coarse *)w.coarse, (const double *)x, v, d.n, (const double *)w.z, pk, nown, w.st, fb, w.pb, fit)
                if (g->dim == 3) { if (fzu) { NICP_UPD(3, true); } else { NICP_UPD(3, false); } }
                else { if (fzu) { NICP_UPD(2, true); } else { NICP_UPD(2, false); } }
                mark(4);
                if (fz) {
                    // sums + coarse solve happen inside the direction kernel
                } else if (!par.sharded()) {
                    hipLaunchKernelGGL(cg_sum_coarse_kernel, dim3(1), dim3(1024), 0, st, (const double *)upd_partial, nbv, upd_mask,
                                       par.tot, d.NG, x_tail, r_tail, zb_tail, (const double *)p_tail, (const double *)Ap_tail, w.coarse,
                                       w.scal, init, (const double *)(rich ? w.mom : nullptr), nm, fzu ? 1 : 0);
                } else {
                    hipLaunchKernelGGL(rows_sum_kernel, dim3(1), dim3(1024), 0, st, (const double *)w.partial, nbv, upd_mask, par.tot, 0);
                    if (rich) hipLaunchKernelGGL(rows_sum_wide_kernel, dim3(1), dim3(1024), 0, st, (const double *)w.mom, nbv, nm, par.tot, 2);
                    if (int rc = allreduce(par, 2 + nm)) return rc;
                    hipLaunchKernelGGL(cg_coarse_logic_kernel, dim3(1), dim3(128), 0, st, (const double *)par.tot, d.NG, x_tail, r_tail, zb_tail,
                                       (const double *)p_tail, (const double *)Ap_tail, w.coarse, w.scal, init, nm);
                }
                mark(5);
                if (fz && richf) { if (g->dim == 3) NICP_DIRR(3); else NICP_DIRR(2); }
                else if (g->dim == 3) { if (fz) NICP_DIR(3, true); else NICP_DIR(3, false); }
                else { if (fz) NICP_DIR(2, true); else NICP_DIR(2, false); }
                mark(6);
#undef NICP_UPD
#undef NICP_UPD2
#undef NICP_DIR
#undef NICP_DIRR
#undef NICP_DIR2
                if (dirfit) pcur ^= 1;  // the new direction (and its fit rows, in w.y) are in the other buffer
                return exchange(ctx, g, w, par, pbuf[pcur]);
            };
            if (int rc = precondition(1, 0)) return rc;  // alpha = 0: z = M^-1 r, rz, rr0, p = z
            ICP_HIP_CHECK(hipGetLastError());
            // `count` CG iterations.  The damping enters as the vector lam*D^2 (w.lamd2, multiplier 1.0: the same
            // bits as lam * d2 * p), so no kernel argument below changes during a solve and the single-GPU
            // path can replay cg_chunk_size() iterations as one hipGraph (6 launches each, launch-bound on small meshes).
            auto cg_iterations = [&](int count) -> int {
                for (int k = 0; k < count; ++k) {
                    mark(0);
                    if (int rc = apply_A(ctx, g, w, par, x, v, c, compat, pbuf[pcur], w.lamd2, 1.0, w.Ap, pev ? pev[1] : nullptr, dirfit, &w.scal->stop)) return rc;
                    mark(2);
                    const double *p_tail = pbuf[pcur] + d.n * d.S;
                    if (fused || mixed) {
                        // the update kernel sums the partials and derives alpha itself
                    } else if (!par.sharded()) {
                        hipLaunchKernelGGL(cg_sum_alpha_kernel, dim3(1), dim3(1024), 0, st, (const double *)w.partial, nbj, ng_mask | (1u << 8),
                                           par.tot, d.NG, lamd2_tail, (const double *)p_tail, 1.0, Ap_tail, w.scal);
                    } else {
                        hipLaunchKernelGGL(rows_sum_kernel, dim3(1), dim3(1024), 0, st, (const double *)w.partial, nbj, ng_mask | (1u << 8), par.tot, 0);
                        if (int rc = allreduce(par, 9)) return rc;
                        hipLaunchKernelGGL(cg_alpha_logic_kernel, dim3(1), dim3(64), 0, st, (const double *)par.tot, d.NG, lamd2_tail,
                                           (const double *)p_tail, 1.0, Ap_tail, w.scal);
                    }
                    mark(3);
                    if (int rc = precondition(0, k & 1)) return rc;
                    if (pev) {  // one iteration at a time: read its seven time stamps
                        ICP_HIP_CHECK(hipEventSynchronize(pev[6]));
                        for (int q = 0; q < 6; ++q) {
                            float ms = 0.f;
                            ICP_HIP_CHECK(hipEventElapsedTime(&ms, pev[q], pev[q + 1]));
                            prof->us_kernel[q] += 1e3 * ms;
                        }
                    }
                }
                ICP_HIP_CHECK(hipGetLastError());
                return ICP_OK;
            };
            if (cg.exec && cg.M != w.pb.M) {  // the coarse space changed: the captured launches no longer apply
                (void)hipGraphExecDestroy(cg.exec);
                cg.exec = nullptr;
            }
            cg.M = w.pb.M;
            if (use_graph && !cg.exec && !cg.failed) {
                // capture once per solve, on the context's side stream (st may be the legacy default stream)
                hipStream_t user = ctx->stream;
                ctx->stream = st = ctx->side;
                hipGraph_t captured = nullptr;
                int rc = ICP_OK;
                hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
                if (e == hipSuccess) {
                    rc = cg_iterations(cg_chunk_size());
                    e = hipStreamEndCapture(st, &captured);
                }
                ctx->stream = st = user;
                if (e == hipSuccess && !rc) e = hipGraphInstantiate(&cg.exec, captured, nullptr, nullptr, 0);
                if (captured) (void)hipGraphDestroy(captured);
                if (rc) return rc;
                if (e != hipSuccess) {  // no graphs on this runtime / stream: plain launches from here on
                    (void)hipGetLastError();
                    cg.exec = nullptr;
                    cg.failed = true;
                }
            }
            tick(3);  // CG start + graph capture / instantiation
            if (prof) {
                // Measurement only (nicp_profile_cg): the stop tolerance is zeroed so that no kernel takes its early
                // exit, then (1) `iterations` CG iterations as plain launches with an event after every launch and
                // (2) the same number the way the solver runs them (hipGraph replay where enabled) between two events.
                const int iters = std::max(2, prof->iterations) & ~1;  // even: the fused kernels alternate a parity
                const double zero = 0.0;
                ICP_HIP_CHECK(hipMemcpyAsync(&w.coarse->tol2, &zero, sizeof(double), hipMemcpyHostToDevice, st));
                ICP_HIP_CHECK(hipStreamSynchronize(st));
                hipEvent_t ev[9];
                for (auto &e : ev) ICP_HIP_CHECK(hipEventCreate(&e));
                for (double &u : prof->us_kernel) u = 0.0;
                if (int rc = cg_iterations(4)) return rc;  // warm
                pev = ev;
                int rc = cg_iterations(iters);
                pev = nullptr;
                if (rc) return rc;
                for (double &u : prof->us_kernel) u /= iters;
                int done_it = 0;
                ICP_HIP_CHECK(hipEventRecord(ev[7], st));
                while (done_it < iters) {
                    if (cg.exec && iters - done_it >= cg_chunk_size()) {
                        ICP_HIP_CHECK(hipEventRecord(ctx->side_ev, st));
                        ICP_HIP_CHECK(hipStreamWaitEvent(ctx->side, ctx->side_ev, 0));
                        ICP_HIP_CHECK(hipGraphLaunch(cg.exec, ctx->side));
                        ICP_HIP_CHECK(hipEventRecord(ctx->side_ev, ctx->side));
                        ICP_HIP_CHECK(hipStreamWaitEvent(st, ctx->side_ev, 0));
                        done_it += cg_chunk_size();
                        prof->graph_replay = 1;
                    } else {
                        const int cnt = std::min(cg_chunk_size(), iters - done_it);
                        if (int rc2 = cg_iterations(cnt)) return rc2;
                        done_it += cnt;
                    }
                }
                ICP_HIP_CHECK(hipEventRecord(ev[8], st));
                ICP_HIP_CHECK(hipEventSynchronize(ev[8]));
                float ms = 0.f;
                ICP_HIP_CHECK(hipEventElapsedTime(&ms, ev[7], ev[8]));
                prof->us_iteration = 1e3 * ms / done_it;
                prof->iterations = done_it;
                prof->launches_per_iteration = (fused ? 4 : (mixed ? 5 : 6));
                prof->vertices = d.n;
                prof->edges = d.e;
                prof->coarse_columns = nz;
                for (auto &e : ev) (void)hipEventDestroy(e);
                w.st = ext;
                return ICP_OK;
            }
            auto cg_loop = [&]() -> int {
                int cg_done = 0;
                for (int ci = 0; ci < o.max_cg_iters && !cg_done;) {
                    const int chunk = std::min(cg_chunk_size(), o.max_cg_iters - ci);
                    if (cg.exec && chunk == cg_chunk_size()) {
                        ICP_HIP_CHECK(hipEventRecord(ctx->side_ev, st));
                        ICP_HIP_CHECK(hipStreamWaitEvent(ctx->side, ctx->side_ev, 0));
                        ICP_HIP_CHECK(hipGraphLaunch(cg.exec, ctx->side));
                        ICP_HIP_CHECK(hipMemcpyAsync(&hs, w.scal, sizeof(Scal), hipMemcpyDeviceToHost, ctx->side));
                        ICP_HIP_CHECK(hipStreamSynchronize(ctx->side));  // also orders the side stream's work before anything queued on st next
                        stt.graph_launches++;
                    } else {
                        if (int rc = cg_iterations(chunk)) return rc;
                        if (int rc = read_scal(ctx, w.scal, hs)) return rc;
                    }
                    ci += chunk;
                    cg_done = hs.stop;
                }
                stt.cg_iters += hs.iters;
                return ICP_OK;
            };
            if (int rc = cg_loop()) return rc;
            tick(4);  // CG iterations
            const int cg_first = hs.iters;
            if (int rc = exchange(ctx, g, w, par, w.delta)) return rc;
            // ---- second-order correction of the step (accel, see the top of this function).  Buffers: d1 is kept in
            // w.x_trial once the two extra residuals are taken, J^T r'' goes through w.z (the next solve's start overwrites it)
            bool geo_used = false;
            const bool geo = accel;
            if (geo) {
                hipLaunchKernelGGL(axpby_kernel, dim3(nbp), dim3(NB), 0, st, 1.0, (const double *)x, 1.0, (const double *)w.delta, w.x_trial, d.np);
                if (int rc = residual_at(ctx, g, w, w.x_trial, v, nn, c, compat, w.r_trial)) return rc;
                hipLaunchKernelGGL(axpby_kernel, dim3(nbp), dim3(NB), 0, st, 1.0, (const double *)x, -1.0, (const double *)w.delta, w.x_trial, d.np);
                if (int rc = residual_at(ctx, g, w, w.x_trial, v, nn, c, compat, w.y)) return rc;
                ICP_HIP_CHECK(hipMemcpyAsync(w.x_trial, w.delta, d.np * sizeof(double), hipMemcpyDeviceToDevice, st));   // d1
                const int nbr_ = grid_for(d.nr);
                // r''(d1, d1) ~ r(x + d1) - 2 r(x) + r(x - d1)
                hipLaunchKernelGGL(axpby_kernel, dim3(nbr_), dim3(NB), 0, st, 1.0, (const double *)w.r_trial, 1.0, (const double *)w.y, w.y, d.nr);
                hipLaunchKernelGGL(axpby_kernel, dim3(nbr_), dim3(NB), 0, st, -2.0, (const double *)w.r, 1.0, (const double *)w.y, w.y, d.nr);
                stt.nfev += 2;
                if (int rc = linearise(ctx, g, w, x, v)) return rc;
                const double *rows2 = w.y;
                if (w.weighted) {   // (W J)^T r'' = J^T (W r'' [+ landmark rows on their fit rows]), as for the gradient
                    if (int rc = weight_rows(ctx, g, w, c, w.y, w.r_trial)) return rc;
                    rows2 = w.r_trial;
                }
                if (int rc = apply_JT(ctx, g, w, par, x, v, c, compat, rows2, nullptr, nullptr, 0.0, w.z, false)) return rc;
                if (int rc = cg_reset(-0.5, w.z)) return rc;
                pcur = 0;
                if (int rc = precondition(1, 0)) return rc;
                if (int rc = cg_loop()) return rc;
                if (int rc = exchange(ctx, g, w, par, w.delta)) return rc;
                if (int rc = pdot_slot(ctx, g, w, par, w.delta, w.delta, w.d2, 5)) return rc;
                if (int rc = pdot_slot(ctx, g, w, par, w.x_trial, w.x_trial, w.d2, 6)) return rc;
                if (int rc = allreduce(par, 7)) return rc;   // (slots 0..4 are formed afresh below)
                if (int rc = read_tot(ctx, par, 7, ht)) return rc;
                geo_used = ht[5] == ht[5] && ht[5] <= 0.5625 * ht[6];
                if (trace) std::fprintf(stderr, "   second-order correction: |D d2| / |D d1| = %.4f  cg %d  %s\n", std::sqrt(ht[5] / ht[6]), hs.iters, geo_used ? "used" : "dropped");
                // delta = d1 (+ d2)
                hipLaunchKernelGGL(axpby_kernel, dim3(nbp), dim3(NB), 0, st, 1.0, (const double *)w.x_trial, 1.0, geo_used ? (const double *)w.delta : (const double *)nullptr, w.delta, d.np);
                hs.iters += cg_first;
            }
            // a long first solve: richer coarse space from the next linear system on (only now -- the second-order solve
            // above still ran with the coarse matrix of this one).  Same decision on every rank (global counters)
            if (cg_first > 400 && w.pb.M < coarse_switch_to) w.pb.M = coarse_switch_to;
            // predicted reduction of |r|^2 : delta^T (lam D^2 delta - grad); with the second-order correction the model is
            // the first-order step's (Transtrum's gain ratio), which still sits in w.x_trial
            {
                const double *dmodel = geo ? w.x_trial : w.delta;
                if (int rc = pdot_slot(ctx, g, w, par, dmodel, dmodel, w.d2, 1)) return rc;   // |D delta|^2
                if (int rc = pdot_slot(ctx, g, w, par, dmodel, w.grad, nullptr, 2)) return rc;
            }
            // ---- trial point and gain ratio
            hipLaunchKernelGGL(axpby_kernel, dim3(nbp), dim3(NB), 0, st, 1.0, (const double *)x, 1.0, (const double *)w.delta, w.x_trial, d.np);
            if (int rc = pdot_slot(ctx, g, w, par, x, x, w.d2, 3)) return rc;               // |D x|^2
            if (int rc = pdot_slot(ctx, g, w, par, w.grad, w.grad, nullptr, 4)) return rc;
            if (int rc = residual_at(ctx, g, w, w.x_trial, v, nn, c, compat, w.r_trial)) return rc;
            if (int rc = dot_slot(ctx, w, par, w.r_trial, w.r_trial, par.row_mask, d.nr, 0)) return rc;
            if (int rc = allreduce(par, 5)) return rc;
            if (int rc = read_tot(ctx, par, 5, ht)) return rc;
            stt.nfev++;
            tick(5);  // trial point
            const double f2t = ht[0], dn2 = ht[1], dg = ht[2], xn2 = ht[3], gg = ht[4];
            const double pred = lam * dn2 - dg;  // = |r|^2 - |r + J delta|^2 - lam|D delta|^2 ... >= 0 for an exact solve
            const double actred = f2 - f2t;
            const double ratio = pred > 0 ? actred / pred : (actred > 0 ? 1.0 : -1.0);
            if (trace)
                std::fprintf(stderr, "nicp_solve lm %3d  f2 %.12e -> %.12e  lam %.3e  cg %5d (stop %d)  pred %.3e  actred %.3e  ratio %.4f  |g| %.3e  M %d\n",
                             it, f2, f2t, lam, hs.iters, hs.stop, pred, actred, ratio, std::sqrt(gg), w.pb.M);
            if (f2t == f2t && ratio > 1e-4) {
                // accept (lmder accepts at ratio >= 1e-4)
                ICP_HIP_CHECK(hipMemcpyAsync(x, w.x_trial, d.np * sizeof(double), hipMemcpyDeviceToDevice, st));
                std::swap(w.r, w.r_trial);
                accepted = true;
                const double t = 2 * ratio - 1;
                lam *= std::max(1.0 / 3.0, 1 - t * t * t);
                nu = 2.0;
                // MINPACK's tests (lmder.f): relative reductions and relative step
                const bool t_f = std::fabs(actred) <= o.ftol * f2 && pred <= o.ftol * f2 && 0.5 * ratio <= 1;
                const bool t_x = std::sqrt(dn2) <= o.xtol * std::sqrt(xn2);
                f2 = f2t;
                if ((t_f || t_x) && cg_tol_cur > cg_tol_tight) {
                    cg_tol_cur = cg_tol_tight;   // met under the loose tolerance: confirm on accurate steps
                } else {
                    if (t_f) info = 1;
                    if (t_x) info = info == 1 ? 3 : 2;
                }
            } else {
                relinearise = true;
                lam *= nu;
                nu *= 2.0;
                if (std::sqrt(dn2) <= o.xtol * std::sqrt(xn2) || lam > 1e30) {
                    if (cg_tol_cur > cg_tol_tight && lam <= 1e30) cg_tol_cur = cg_tol_tight;
                    else info = 2;  // step collapsed without progress
                }
            }
            if (!info && o.gtol > 0 && std::sqrt(gg) <= o.gtol * std::sqrt(f2)) info = 4;
        }
    }
    if (!info) info = 5;
    stt.info = info;
    stt.err = f2;
    if (stats_out) *stats_out = stt;
    // back to the caller's layout
    w.st = ext;
    if (int rc = relayout(ctx, x, soa.pj, soa.pe, P_inout, ext.pj, ext.pe, d.n, d.S)) return rc;
    ICP_HIP_CHECK(hipMemcpyAsync(P_inout + d.n * d.S, x + d.n * d.S, d.NG * sizeof(double), hipMemcpyDeviceToDevice, st));
    ICP_HIP_CHECK(hipStreamSynchronize(st));
    tick(6);
    if (timing)
        std::fprintf(stderr, "nicp_solve timing [ms]: setup %.2f  linearise+grad+blocks %.2f  coarse matrix %.2f  cg start+graph %.2f  cg %.2f  trial %.2f  finish %.2f  (lm %d, cg %d)\n",
                     1e3 * tsec[0], 1e3 * tsec[1], 1e3 * tsec[2], 1e3 * tsec[3], 1e3 * tsec[4], 1e3 * tsec[5], 1e3 * tsec[6], stt.lm_iters, stt.cg_iters);
    return ICP_OK;
}

extern "C" int nicp_solve(icp_ctx *ctx, nicp_graph *g, double *P_inout, const double *v, const double *nn,
                          const double *c3_host, unsigned flags, const nicp_solve_opts *opts_in,
                          nicp_solve_stats *stats_out) {
    NICP_ARGS(ctx, g);
    ICP_REQUIRE(P_inout && v && nn && c3_host, "null pointer");
    return solve_impl(ctx, g, w, par_single(ctx, g, w), P_inout, v, nn, c3_host, flags, opts_in, stats_out);
}

extern "C" int nicp_profile_cg(icp_ctx *ctx, nicp_graph *g, const double *P, const double *v, const double *nn,
                               const double *c3_host, unsigned flags, int iterations, nicp_cg_profile *out_host) {
    NICP_ARGS(ctx, g);
    ICP_REQUIRE(P && v && nn && c3_host && out_host && iterations > 0, "bad argument");
    // the solve works on a copy: P is left untouched
    const Dims d = dims_of(g);
    double *Pc = nullptr;
    ICP_HIP_CHECK(hipMalloc((void **)&Pc, d.np * sizeof(double)));
    int rc = hipMemcpyAsync(Pc, P, d.np * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess ? ICP_OK : ICP_ERR_HIP;
    std::memset(out_host, 0, sizeof(*out_host));
    out_host->iterations = iterations;
    if (!rc) rc = solve_impl(ctx, g, w, par_single(ctx, g, w), Pc, v, nn, c3_host, flags, nullptr, nullptr, out_host);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(Pc);
    return rc;
}

extern "C" int nicp_solve_dist(icp_ctx *ctx, nicp_graph *g, double *P_inout, const double *v, const double *nn,
                               const double *c3_host, unsigned flags, const nicp_solve_opts *opts_in,
                               nicp_solve_stats *stats_out, const nicp_comm *comm) {
    NICP_ARGS(ctx, g);
    ICP_REQUIRE(P_inout && v && nn && c3_host && comm, "null pointer");
    ICP_REQUIRE(comm->world >= 1 && comm->rank >= 0 && comm->rank < comm->world, "bad rank / world");
    ICP_REQUIRE(comm->n_owned >= 0 && comm->n_owned <= g->n, "n_owned exceeds the local mesh");
    ICP_REQUIRE(comm->n_send >= 0 && (comm->n_send == 0 || (comm->send_idx_dev && comm->send_buf_dev)), "send lists missing");
    ICP_REQUIRE(comm->n_owned == g->n || comm->recv_buf_dev, "recv_buf_dev missing");
    if (flags & 1u) {
        set_error("nicp_solve_dist: NICP_REFERENCE_COMPAT does not shard (quirk Q5 couples rigid-residual slots of unrelated vertices)");
        return ICP_ERR_UNSUPPORTED;
    }
    Par par;
    par.comm = comm;
    par.ctx = ctx;
    par.nown = comm->n_owned;
    par.tot = w.tot;
    if (par.sharded()) {
        ICP_REQUIRE(comm->reduce_buf_dev && comm->row_mask_dev, "comm buffers missing");
        ICP_REQUIRE(comm->rccl ? (comm->send_counts_host && comm->recv_counts_host) : (comm->allreduce_sum && comm->exchange),
                    "callbacks missing (or, with rccl, the per-rank send / receive counts)");
        par.tot = comm->reduce_buf_dev;
        par.row_mask = comm->row_mask_dev;
    }
    if (comm->world == 1) ICP_REQUIRE(comm->n_owned == g->n, "a single rank owns every local vertex");
    return solve_impl(ctx, g, w, par, P_inout, v, nn, c3_host, flags, opts_in, stats_out);
}

// NICP outer loop: nicp_3d.py:344-363 (Deform -> Match -> leastsq -> stiffness schedule).
extern "C" int nicp_run(icp_ctx *ctx, nicp_graph *g, double *P_inout, const double *v, const double *tgt, int64_t m,
                        double tol, int max_outer, double *c3_inout_host, unsigned flags,
                        const nicp_solve_opts *opts_host, double *deformed_out, double *err_hist_host,
                        int *outer_iters_host) {
    ICP_REQUIRE(ctx && g && P_inout && v && tgt && c3_inout_host, "null pointer");
    ICP_REQUIRE(m > 0 && m < 0x7fffffff && max_outer >= 0, "bad sizes");
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t n = g->n;
    const int dim = g->dim;
    if (int rc = ensure(ctx, ctx->grid_e, (size_t)n * dim * sizeof(double) * 2 + (size_t)n * sizeof(int32_t))) return rc;
    double *dd = (double *)ctx->grid_e.ptr, *nn = dd + n * dim;
    int32_t *idx = (int32_t *)(nn + n * dim);
    double err = 100000000.0;  // nicp_3d.py:346
    int it = 0;
    // the target does not change: one search index for every outer iteration, and from the second
    // iteration on the previous correspondences go in as starting bounds
    GridBuild grid;
    const bool indexed = grid_worthwhile(n, m, 0u);
    if (indexed)
        if (int rc = grid_build(ctx, tgt, m, dim, grid)) return rc;
    for (; it < max_outer; ++it) {
        const double e = err;
        if (int rc = nicp_deform(ctx, g, P_inout, v, dd)) return rc;
        if (int rc = closest_dispatch(ctx, dd, n, tgt, m, dim, it > 0 ? (unsigned)ICP_HINT_IDX : 0u, idx, nullptr, nullptr,
                                      indexed ? &grid : nullptr))
            return rc;
        if (int rc = icp_gather_points(ctx, tgt, idx, n, dim, nn)) return rc;
        nicp_solve_stats st;
        if (int rc = nicp_solve(ctx, g, P_inout, v, nn, c3_inout_host, flags, opts_host, &st)) return rc;
        err = st.err;
        if (err_hist_host) err_hist_host[it] = err;
        if ((e - err) < tol) {  // nicp_3d.py:355-359
            c3_inout_host[1] /= 2.0;
            c3_inout_host[2] /= 2.0;
            if (c3_inout_host[1] < 0.1 || c3_inout_host[2] < 0.1) {
                ++it;
                break;
            }
        }
    }
    if (outer_iters_host) *outer_iters_host = it;
    if (deformed_out)
        if (int rc = nicp_deform(ctx, g, P_inout, v, deformed_out)) return rc;
    return ICP_OK;
}
