/*
 * icp_hip.h -- C-ABI of libicp_hip.so: the MI355X (gfx950) implementation of the
 * drawpixels/icp hot path (closest point, Kabsch step, NICP solve).
 *
 * The reference has no FFI seam of its own: its hot path is a set of module-level
 * Python functions and private C++ methods.  Every entry
 * point below therefore cites the reference function(s) it replaces; the Python
 * mirror (the icp_amd Python modules) and the C++ nicp3d core (icp_amd/csrc/nicp3d_core.cpp)
 * bind exactly these symbols.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error (icp_last_error() gives the
 *    message of the calling thread's last failure); nothing throws across the ABI.
 *  - all array pointers are DEVICE pointers to caller-owned fp64 / int32 buffers,
 *    row-major [rows, dim] (AoS), dim in {2,3}; row-vector convention p' = p R + t.
 *  - work is enqueued on the context's stream (icp_ctx_set_stream) and is
 *    asynchronous unless stated; calls on one context must be serialised by the
 *    caller (one host thread at a time); contexts are independent.  A context's
 *    scratch state assumes one stream of execution: icp_ctx_set_stream with a
 *    different stream makes the new stream wait (by an event, without blocking the
 *    host) for everything the context queued on the previous one, so calls issued
 *    under different streams run in issue order.  Use one context per stream for
 *    work that should overlap.  The previous stream should still exist at that call;
 *    a host that destroys its streams detaches first with icp_ctx_set_stream(ctx, NULL)
 *    (the null stream).  If the previous stream is gone nevertheless, the call orders
 *    by a device-wide wait instead and still moves the context to the new stream.
 *  - the library frees nothing it did not allocate.
 */
#ifndef ICP_HIP_H
#define ICP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct icp_ctx icp_ctx;
typedef void *icp_stream; /* a hipStream_t */

enum {
    ICP_OK = 0,
    ICP_ERR_ARG = -1,         /* bad argument (null pointer, dim not 2/3, n<=0 ...) */
    ICP_ERR_HIP = -2,         /* a HIP runtime call failed */
    ICP_ERR_NOMEM = -3,       /* scratch allocation failed */
    ICP_ERR_UNSUPPORTED = -4, /* valid request this build cannot serve */
    ICP_ERR_NUMERIC = -5,     /* solver broke down (non-finite values) */
    ICP_ERR_COMM = -6         /* a communication callback of nicp_solve_dist reported failure */
};

/* flags for icp_closest / icp_rigid_run */
enum {
    ICP_COMPARE_SQRT = 1u, /* nicp3d.cpp:56-63,142-164: compare rounded sqrt(d2) (Q12) */
    ICP_FORCE_BRUTE = 2u,  /* never use the spatial-grid search, always scan all targets */
    ICP_FORCE_GRID = 4u,   /* use the grid search whatever the problem size (tests) */
    ICP_HINT_IDX = 8u,     /* idx_out holds, on entry, some target index per query (typically the previous
                              iteration's correspondence): the grid search evaluates it first as its starting
                              bound.  The result is the same exact argmin; out-of-range entries are ignored. */
    ICP_NO_FUSE = 16u,     /* icp_rigid_run[_dist]: keep the moved source as an array and run `apply` and the error
                              as kernels of their own instead of inside the grid search (same T and correspondences
                              bit for bit, errors equal up to summation order; for tests and measurements) */
    ICP_COHERENT = 32u     /* icp_rigid_run[_dist], fused form: every search also leaves, per source row, a lower bound on
                              the distance to every target OTHER than its match; in the following iterations a row whose
                              match is still provably the nearest target (triangle inequality on how far the row has moved
                              since, with slack for every rounding) keeps it without a search.  Same correspondences, T
                              and errors bit for bit (tests/test_gpu_grid.py::test_coherent_loop_equals_plain_loop,
                              tests/test_gpu_rigid.py, benchmarks/fuzz_grid.py --coherent); an iteration in
                              which more than 12 % of the rows cannot be proven searches every row as usual.  Pays from
                              about 1e5 rows on (2x at 2^20, 3x at 2^24).  No counterpart in the reference, whose
                              Match (icp_3d.py:48-53) scans every target for every row in every iteration. */
};

const char *icp_last_error(void);
int icp_version(void);

/* context: owns scratch buffers and the stream for one device */
int icp_ctx_create(int device, icp_ctx **out);
int icp_ctx_destroy(icp_ctx *ctx);
int icp_ctx_set_stream(icp_ctx *ctx, icp_stream stream);
int icp_sync(icp_ctx *ctx);

/* device-memory helpers for hosts that have no tensor library (the C++ nicp3d
 * core); replaces TNT::Array1D/Array2D storage, tnt_array1d.h:38-85,
 * tnt_array2d.h:36-79.  h2d/d2h are stream-ordered and synchronise before return. */
int icp_malloc(icp_ctx *ctx, size_t bytes, void **dptr);
int icp_free(icp_ctx *ctx, void *dptr);
int icp_memcpy_h2d(icp_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int icp_memcpy_d2h(icp_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* ---------------------------------------------------------------- rigid ICP */

/* Closest / Match: icp_3d.py:39-42,48-53 (= icp_row.py:50-64, nicp_3d.py:62-77,
 * nicp_row.py:75-89, nicp_row_rt.py:55-69, icp.py:18-21) and nicp3d.cpp:142-164,
 * 259-272 with ICP_COMPARE_SQRT.  idx_out[i] = argmin_j ((dx*dx)+(dy*dy))+(dz*dz),
 * fp64, no FMA, lowest index on ties, first NaN wins (np.argmin).  d2_out may be
 * NULL.  Large problems (n*m >= 2^30) are served by an exact multi-resolution grid search
 * over the target that returns the identical indices (grid.hip); ICP_FORCE_BRUTE keeps the
 * full scan.  The grid build synchronises the stream once. */
int icp_closest(icp_ctx *ctx, const double *src, int64_t n, const double *tgt, int64_t m, int dim,
                unsigned flags, int32_t *idx_out, double *d2_out);

/* Persistent search index over one target cloud, for callers that run many Match passes
 * against the same target (every ICP iteration does: the `target` argument of icp_3d.py:102
 * never changes inside the loop).  create builds the grid once (synchronises);
 * icp_closest_target then never rebuilds and never synchronises.  The target buffer must stay
 * alive and unchanged while the handle exists.  Results are identical to icp_closest. */
typedef struct icp_target icp_target;
int icp_target_create(icp_ctx *ctx, const double *tgt, int64_t m, int dim, icp_target **out);
int icp_target_destroy(icp_ctx *ctx, icp_target *t);
int icp_closest_target(icp_ctx *ctx, const icp_target *t, const double *src, int64_t n, unsigned flags,
                       int32_t *idx_out, double *d2_out);

/* Morton (Z-curve) ordering of a point set: sorted_out[i] = pts[order_out[i]].  Used to put a
 * source shard into a cache-friendly order once per run (a rigid motion keeps neighbours
 * neighbours).  No counterpart in the reference.  Synchronises. */
int icp_morton_sort(icp_ctx *ctx, const double *pts, int64_t n, int dim, double *sorted_out, int32_t *order_out);

/* nn[i] = tgt[idx[i]]: the gather inside Match, icp_3d.py:52, nicp3d.cpp:266 */
int icp_gather_points(icp_ctx *ctx, const double *tgt, const int32_t *idx, int64_t n, int dim,
                      double *nn_out);

/* Distance: icp_3d.py:59-62 -- sum_i ||a_i - b_i||_2 -> out_dev[0] */
int icp_distance_sum(icp_ctx *ctx, const double *a, const double *b, int64_t n, int dim,
                     double *out_dev);

/* FindTransform: icp_3d.py:68-90 (icp_row.py:79-107 packs the same numbers into T;
 * icp.py:59-74 is the transpose).  Kabsch, no scale, no reflection fix (Q2).
 * R_out_dev[dim*dim] row-major, t_out_dev[dim]. */
int icp_find_transform(icp_ctx *ctx, const double *src, const double *nn, int64_t n, int dim,
                       double *R_out_dev, double *t_out_dev);

/* FindTransform for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same arguments, same kernels and the
 * same bits as icp_find_transform, but the block-partial rows and the small state live in a file-scope
 * device array of 66 048 bytes that every GPU holds from library load to process end: this call performs no
 * hipMalloc / hipFree and no allocating library routine.  Limit: n <= ICP_NOALLOC_MAX_ROWS, larger inputs
 * are refused with ICP_ERR_UNSUPPORTED.  All calls on one GPU (from any context) share that array: calls
 * that overlap on one GPU (different streams or threads) race; serialise them. */
#define ICP_NOALLOC_MAX_ROWS ((int64_t)1 << 27)
int icp_find_transform_noalloc(icp_ctx *ctx, const double *src, const double *nn, int64_t n, int dim,
                               double *R_out_dev, double *t_out_dev);

/* ss = src R + t: icp_3d.py:113 */
int icp_apply_transform(icp_ctx *ctx, const double *src, int64_t n, int dim, const double *R_dev,
                        const double *t_dev, double *out);

/* ICP: icp_3d.py:102-121 (use_tol=0: icp_row.py:112-143, Q4).  Runs up to max_iter
 * iterations entirely on the device (no host round trip per iteration).
 * T_out_dev[(dim+1)^2] row-vector homogeneous matrix; err_hist_dev[max_iter] (may be
 * NULL) receives Distance(nn,ss) per executed iteration; iters_out_dev[0] = number of
 * executed iterations; idx_out (may be NULL) = correspondences of the last executed
 * iteration.  max_iter = 0 is an error like the reference's NameError (Q3). */
int icp_rigid_run(icp_ctx *ctx, const double *src, int64_t n, const double *tgt, int64_t m, int dim,
                  double tol, int max_iter, int use_tol, unsigned flags, double *T_out_dev,
                  double *err_hist_dev, int32_t *iters_out_dev, int32_t *idx_out);

/* Building blocks of one ICP iteration for source-sharded multi-GPU runs
 * (SURVEY.md section 8e): every rank calls these on its shard and all-reduces (sum) the
 * small fp64 vectors in between.
 *   icp_shard_sums:  sums_out_dev[16] = [ sum(nn-ct) (3) | sum(s-cs) (3) |
 *                    sum (nn-ct)^T (s-cs) (9, [a*dim+b], a indexes nn) | count ]
 *                    with nn = tgt[idx]; cs_dev/ct_dev are dim-vectors of shifts
 *                    (any value; the global source mean / target mean are good).
 *                    For dim=2 the unused slots are 0.
 *   icp_kabsch_solve: from (all-reduced) sums and the shifts -> R_out_dev, t_out_dev.
 *   icp_shard_apply_err: ss = src R + t, err_out_dev[0] = sum_i ||tgt[idx_i]-ss_i||.
 * Together they compute what FindTransform + icp_3d.py:113-114 compute. */
int icp_shard_sums(icp_ctx *ctx, const double *src, const double *tgt, const int32_t *idx, int64_t n,
                   int dim, const double *cs_dev, const double *ct_dev, double *sums_out_dev);
int icp_kabsch_solve(icp_ctx *ctx, const double *sums_dev, int dim, const double *cs_dev,
                     const double *ct_dev, double *R_out_dev, double *t_out_dev);
int icp_shard_apply_err(icp_ctx *ctx, const double *src, const double *tgt, const int32_t *idx,
                        int64_t n, int dim, const double *R_dev, const double *t_dev, double *ss_out,
                        double *err_out_dev);
/* column sums of a [n,dim] array -> out_dev[dim] (for the shifts above) */
int icp_column_sums(icp_ctx *ctx, const double *a, int64_t n, int dim, double *out_dev);

/* The same loop as icp_rigid_run with the SOURCE ROWS sharded over ranks (one process per GPU) and the target
 * replicated: icp_3d.py:102-121 needs one exchange per iteration, the sums behind
 * FindTransform.  The library never talks to the network itself: it calls `allreduce_sum`, which the host implements
 * over RCCL / MPI / torch.distributed; the callback must be ordered after the work already queued on the context's
 * stream and must have completed, or be ordered on that stream, when it returns (0 = success).  Per call every rank
 * issues exactly max_iter + 2 collectives (4 doubles once, 17 per iteration, 1 at the end) whatever the stop test
 * decides -- decisions are taken on the device from all-reduced values, so the host never reads anything back inside
 * the loop.  src is the rank's shard [n_local, dim]; T / err_hist / iters are identical on all ranks; idx_out
 * [n_local] are the shard's correspondences of the last executed iteration.  world = 1 runs without callbacks and
 * returns what icp_rigid_run returns, bit for bit.  No counterpart in the reference (single process).
 *
 * Library-side RCCL (icp_comm.rccl / nicp_comm.rccl).  The north_star's "RCCL over xGMI all-reduces the covariance /
 * CG partial sums": rank 0 calls icp_rccl_unique_id, the host distributes the ICP_RCCL_ID_BYTES bytes to every rank by
 * whatever means it has (torch.distributed broadcast, MPI_Bcast, a file), every rank calls icp_rccl_create with the
 * same id -- collectively, one process per GPU, with its context's device current -- and hands the handle over in the
 * comm structs.  The library then enqueues ncclAllReduce / grouped ncclSend + ncclRecv on the context's stream, ordered
 * like kernel launches: zero host callbacks per iteration.  librccl is opened at run time (no link-time dependency);
 * without it icp_rccl_* return ICP_ERR_UNSUPPORTED and the callbacks remain.  Destroy the handle (collectively) after the
 * last call that used it has completed. */
#define ICP_RCCL_ID_BYTES 128
typedef struct icp_rccl icp_rccl; /* the library's own RCCL communicator */
int icp_rccl_unique_id(void *id_out_host);
int icp_rccl_create(icp_ctx *ctx, const void *id_host, int rank, int world, icp_rccl **out);
int icp_rccl_destroy(icp_rccl *r);

enum {
    ICP_COMM_FORCE = 1u /* issue every collective even when world == 1 (rehearsals and tests on one GPU: the whole
                           sharded code path runs, the sums over "ranks" are sums over one rank) */
};
typedef struct icp_comm {
    int rank, world;
    double *reduce_buf_dev; /* >= 32 doubles (device); allreduce_sum sums its first `count` entries over ranks in place */
    void *user;
    int (*allreduce_sum)(void *user, int count);
    icp_rccl *rccl;         /* non-NULL: the library issues ncclAllReduce on the context's stream itself and never calls
                               allreduce_sum (which may then be NULL): no host code runs per collective */
    unsigned flags;         /* ICP_COMM_FORCE */
} icp_comm;
int icp_rigid_run_dist(icp_ctx *ctx, const double *src_local, int64_t n_local, const double *tgt, int64_t m, int dim,
                       double tol, int max_iter, int use_tol, unsigned flags, double *T_out_dev, double *err_hist_dev,
                       int32_t *iters_out_dev, int32_t *idx_out, const icp_comm *comm);

/* Measurement hook (no counterpart in the reference): builds the target grid and the Morton
 * order of the queries exactly as icp_rigid_run does, then times `reps` grid searches with HIP
 * events on the context's stream.  ms_out_dev[6] = average milliseconds of {one whole search,
 * its first kernel (grid_pool_kernel), its second kernel for the deferred queries
 * (grid_defer_kernel)} without hints -- the first iteration of a loop -- and of the same three
 * with the previous answer as the hint, as every later iteration runs it.  Synchronises. */
int icp_bench_grid_query(icp_ctx *ctx, const double *src, int64_t n, const double *tgt, int64_t m, int dim,
                         int reps, double *ms_out_dev);

/* ------------------------------------------------------------ non-rigid ICP */

typedef struct nicp_graph nicp_graph;

/* flags for the nicp_* operators */
enum {
    NICP_REFERENCE_COMPAT = 1u /* reproduce the reference's quirks Q5/Q6/Q7; 0 =
                                  self-consistent residual/Jacobian pair */
};

/* KNN: nicp_3d.py:53-57 / nicp3d.cpp:100-137.  For every row of q[nq,dim] the k smallest
 * Euclidean distances (fp64, with sqrt) to the rows of v[n,dim], ascending, ties to the lower
 * index; k must be 6 or 8.  idx_out[nq,k] int32, dist_out[nq,k].  With fewer than k comparable points
 * (n < k, or NaN coordinates) the unfilled slots hold index -1 and distance +inf. */
int nicp_knn(icp_ctx *ctx, const double *v, int64_t n, int dim, const double *q, int64_t nq, int k,
             int32_t *idx_out, double *dist_out);

/* The weight loop of Initialise: nicp_3d.py:158-164 / nicp_row.py:100-106 /
 * nicp3d.cpp:196-215.  Sparse form of the reference's dense w[N,N]:
 * w[nbr_out[j,i], j] = wt_out[j,i], i < num_nn (= 4). */
int nicp_knn_weights(icp_ctx *ctx, const double *v, int64_t n, int dim, int num_nn, int32_t *nbr_out,
                     double *wt_out);

/* Mesh connectivity handle: copies nbr/wt/edges[e,2] and builds the reverse adjacency used by
 * the J^T products (synchronises; once per mesh).  Replaces the dense `weights` member
 * (nicp3d.cpp:39,199) and the `w`, `edges` arguments of nicp_3d.py:220,239,274,344. */
int nicp_graph_create(icp_ctx *ctx, const int32_t *nbr_dev, const double *wt_dev, int64_t n, int num_nn,
                      const int32_t *edges_dev, int64_t e, int dim, nicp_graph **out);
int nicp_graph_destroy(icp_ctx *ctx, nicp_graph *g);

/* Extended energy.  No counterpart in the reference, whose three
 * stiffness constants are module globals (nicp_3d.py:17-20) applied uniformly: these attach
 * per-row multipliers and landmark rows to a graph, and every operator below (nicp_residual,
 * nicp_jvp, nicp_jtvp, nicp_solve, nicp_solve_dist, nicp_run) then works on
 *   [c_fit fit_w_i (nn_i - Deform_i) | c_rigid rigid_w_i e_rigid_i | c_smooth edge_w_e e_smooth_e |
 *    c_land (L_k - Deform_{l_k})].
 * nicp_set_weights: fit_w_dev[n], rigid_w_dev[n], edge_w_dev[e] are DEVICE arrays of non-negative
 * multipliers, copied; any of them may be NULL (= 1 everywhere); all NULL removes the weights.
 * nicp_set_landmarks: nl landmark rows, vertex_dev[nl] (distinct vertex ids) pulled towards
 * pos_dev[nl,dim] with the constant c_land; the rows are appended after the smoothness rows, dim per
 * landmark, in the order given; nl = 0 removes them.  nicp_rows returns the length of the residual
 * vector of the graph as it is now.  Weighted graphs refuse NICP_REFERENCE_COMPAT. */
int nicp_set_weights(icp_ctx *ctx, nicp_graph *g, const double *fit_w_dev, const double *rigid_w_dev,
                     const double *edge_w_dev);
int nicp_set_landmarks(icp_ctx *ctx, nicp_graph *g, const int32_t *vertex_dev, const double *pos_dev,
                       int64_t nl, double c_land);
int nicp_rows(const nicp_graph *g, int64_t *rows_out);

/* Deform: nicp_3d.py:220-233 / nicp_row.py:112-126 / nicp3d.cpp:277-371.  P has
 * (dim*dim+dim)*n + (dim==3 ? 6 : 3) entries (PtoAB layout, nicp_3d.py:171-175). */
int nicp_deform(icp_ctx *ctx, nicp_graph *g, const double *P, const double *v, double *out);

/* MinFunc: nicp_3d.py:239-267 / nicp_row.py:132-160.  c3_host = {c_fit, c_rigid, c_smooth}
 * (HOST pointer).  r_out has dim*n + (dim==3 ? 6 : 3)*n + 2*dim*e entries. */
int nicp_residual(icp_ctx *ctx, nicp_graph *g, const double *P, const double *v, const double *nn,
                  const double *c3_host, unsigned flags, double *r_out);

/* Products with the Jacobian D_MinFunc builds densely (nicp_3d.py:274-338 / nicp_row.py:167-230)
 * without forming it: y = J p, g = J^T u. */
int nicp_jvp(icp_ctx *ctx, nicp_graph *g, const double *P, const double *v, const double *c3_host,
             unsigned flags, const double *p_in, double *y_out);
int nicp_jtvp(icp_ctx *ctx, nicp_graph *g, const double *P, const double *v, const double *c3_host,
              unsigned flags, const double *u_in, double *g_out);

/* options / statistics of nicp_solve (plain old data in HOST memory) */
typedef struct nicp_solve_opts {
    double ftol, xtol, gtol; /* MINPACK-style relative tolerances; leastsq defaults 1.49012e-8, 1.49012e-8, 0 */
    double lambda0;          /* initial Levenberg-Marquardt damping (1e-3) */
    double cg_tol;           /* relative residual |r| / |g| of the inner conjugate-gradient solve; <= 0 (default): automatic,
                                min(1e-3, 10 sqrt(min(ftol, xtol))).  Below 1e-6 the preconditioned residual r.z / r0.z0 is
                                held to the same bound as well: on the badly scaled systems of the reference's constants
                                a small |r| alone leaves the step up to 2e-6 from the exact one */
    int max_lm_iters;        /* cap on the LM iterations; <= 0 (default): none beyond leastsq's own limit of
                                100 * (n + 1) residual evaluations (MINPACK's maxfev default, nicp_3d.py:351), which
                                always applies; stats.info = 5 says that a limit ended the solve */
    int max_cg_iters;        /* per linear solve (0 = automatic) */
    int accel;               /* accelerated LM iteration: every step gets the second-order (geodesic acceleration) correction
                                -1/2 (J^T J + lambda D^2)^-1 J^T r''(d, d), r'' by central differences of the residual, and an
                                automatic cg_tol starts at 0.3 and returns to its tight value before the stopping tests are
                                accepted.  0 (default): automatic = on in the self-consistent mode, off under
                                NICP_REFERENCE_COMPAT; 1: on; -1: off (the plain iteration, step for step
                                what the oracle's lm_dense / lm_sparse restate).  Same minimiser either way. */
} nicp_solve_opts;
typedef struct nicp_solve_stats {
    double err;  /* sum of squared residuals at the returned P (nicp_3d.py:353) */
    double err0; /* ... at the initial P */
    int lm_iters, cg_iters, nfev;
    int info;    /* 1 ftol, 2 xtol, 3 both, 4 gtol, 5 iteration / evaluation limit */
    int graph_launches; /* 20-iteration CG chunks replayed as one hipGraph (0: plain launches; see ICP_NICP_GRAPH) */
} nicp_solve_stats;

/* The optimize.leastsq call of nicp_3d.py:351 / nicp_row.py:246: minimise |MinFunc(P)|^2 from
 * P_inout with correspondences nn fixed.  Levenberg-Marquardt; each step solves
 * (J^T J + lambda D^2) delta = -J^T r by conjugate gradients on the matrix-free operators above,
 * preconditioned by the inverted per-vertex blocks plus a coarse space of the modes the stiffness
 * terms cannot see: 12 columns (6 in 2-D), enriched to 66 (21) -- those modes times the quadratic
 * polynomials of the vertex position -- on meshes of >= 2048 vertices once a linear solve needs more
 * than 400 iterations (environment ICP_NICP_COARSE = 1 | 4 | 10 fixes the space).  On one GPU the CG iterations are replayed 20 at a time
 * as a hipGraph captured once per solve (meshes of >= 1024 vertices; environment ICP_NICP_GRAPH=0/1
 * forces it off/on; same kernels, same bits).  opts_host may be NULL; stats_host may be NULL.  Synchronises. */
int nicp_solve(icp_ctx *ctx, nicp_graph *g, double *P_inout, const double *v, const double *nn,
               const double *c3_host, unsigned flags, const nicp_solve_opts *opts_host,
               nicp_solve_stats *stats_host);

/* Measurement hook (no counterpart in the reference): sets up the first linear solve of nicp_solve at P exactly as
 * the solver does (linearisation, column scaling, vertex blocks, coarse matrix) and times its conjugate-gradient
 * iteration with HIP events on the context's stream: every launch of an iteration separately (plain launches, an
 * event after each), and the iteration as the solver really runs it (hipGraph replay where enabled).  P is not
 * modified.  us_kernel = {fitrow_kernel, normal_kernel, cg_sum_alpha_kernel, cg_update_prec_kernel,
 * cg_sum_coarse_kernel, cg_dir_prec_kernel}; a stage folded into its consumer (small meshes) reads ~0.  Synchronises. */
typedef struct nicp_cg_profile {
    int iterations;             /* in: CG iterations to time; out: iterations timed for us_iteration */
    int launches_per_iteration; /* 4, 5 or 6 */
    int graph_replay;           /* 1: us_iteration was measured on the hipGraph replay */
    int coarse_columns;         /* columns of the coarse space in use */
    int64_t vertices, edges;
    double us_iteration;        /* microseconds per CG iteration, the solver's own launch path */
    double us_kernel[6];        /* microseconds between the events around each launch (kernel + launch gap) */
} nicp_cg_profile;
int nicp_profile_cg(icp_ctx *ctx, nicp_graph *g, const double *P, const double *v, const double *nn,
                    const double *c3_host, unsigned flags, int iterations, nicp_cg_profile *out_host);

/* One rank's view of a vertex-sharded solve.  The mesh is cut
 * into vertex ranges; a rank's local mesh is [owned vertices | halo copies] with the fit rows of
 * every vertex whose influence set touches an owned vertex and the edges with an owned endpoint
 * (icp_amd/nicp_dist.py builds it).  The library never talks to the network itself: it calls the
 * two callbacks below, which the host implements over RCCL / MPI / torch.distributed.  Both must be
 * ordered after the work already queued on the context's stream and must have completed, or be
 * ordered on that stream, when they return; they return 0 on success.  All pointers except `user`
 * are DEVICE pointers. */
typedef struct nicp_comm {
    int rank, world;
    int64_t n_owned;             /* the first n_owned local vertices are owned, the rest are halo copies */
    double *reduce_buf_dev;      /* >= 128 doubles; allreduce_sum sums its first `count` entries over ranks in place */
    const int32_t *send_idx_dev; /* [n_send] local ids of owned vertices that peers hold as halo, grouped by destination rank */
    int64_t n_send;
    double *send_buf_dev;        /* [n_send, S] packed by the library before every exchange, S = dim*dim+dim */
    double *recv_buf_dev;        /* [n_halo, S] filled by exchange() in local halo order */
    const double *row_mask_dev;  /* [rows] 1.0 for the residual rows this rank owns, else 0.0 (rows as in nicp_residual) */
    void *user;
    int (*allreduce_sum)(void *user, int count);
    int (*exchange)(void *user);
    icp_rccl *rccl;                 /* non-NULL: both collectives are issued by the library over RCCL (see icp_rccl_create);
                                       the callbacks are not called and may be NULL */
    const int64_t *send_counts_host; /* [world] with rccl: vertices sent to every rank (send_idx_dev is grouped that way) */
    const int64_t *recv_counts_host; /* [world] with rccl: halo vertices received from every rank (local halo order is by owner rank) */
    unsigned flags;                 /* ICP_COMM_FORCE */
} nicp_comm;

/* nicp_solve on one rank of a sharded mesh: same algorithm, every sum completed by allreduce_sum
 * (<= 68 doubles each) and the search direction's halo refreshed once per CG iteration.  g is the
 * LOCAL graph, P_inout the local parameter vector [owned | halo | tail] (tail replicated), nn the
 * matched points of the local vertices.  stats are global.  flags must be 0: quirk Q5 of the
 * compat mode couples rigid-residual slots of unrelated vertices and does not shard.  Every rank
 * must call it collectively.  No counterpart in the reference (single process, dense Jacobian). */
int nicp_solve_dist(icp_ctx *ctx, nicp_graph *g, double *P_inout, const double *v, const double *nn,
                    const double *c3_host, unsigned flags, const nicp_solve_opts *opts_host,
                    nicp_solve_stats *stats_host, const nicp_comm *comm);

/* NICP: nicp_3d.py:344-363 (interactive twin nicp_row.py:243-264).  Outer loop
 * Deform -> Match -> nicp_solve; when the energy improves by less than tol, c_rigid and
 * c_smooth are halved and the loop ends once either drops below 0.1.  c3_inout_host is
 * updated in place (the reference mutates its module globals, nicp_3d.py:345,356-357).
 * tgt[m,dim] is a device pointer; err_hist_host[max_outer], outer_iters_host and
 * deformed_out[n,dim] (device) may be NULL.  Synchronises. */
int nicp_run(icp_ctx *ctx, nicp_graph *g, double *P_inout, const double *v, const double *tgt, int64_t m,
             double tol, int max_outer, double *c3_inout_host, unsigned flags,
             const nicp_solve_opts *opts_host, double *deformed_out, double *err_hist_host,
             int *outer_iters_host);

#ifdef __cplusplus
}
#endif
#endif /* ICP_HIP_H */
