/*
 * rigid_c_abi.c -- a plain C99 host of libicp_hip.so: no C++, no HIP headers, no tensor
 * library.  It builds the clouds of BASELINE config C1 at a size given on the command line from a small LCG (so
 * that a test can rebuild the very same doubles), runs ICP (icp_3d.py:102-121) through icp_rigid_run, then
 * FindTransform (icp_3d.py:68-90) on the final correspondences through both icp_find_transform and the entry point
 * that allocates nothing, and prints everything as text.
 *
 *   gcc -std=c99 -Wall -Wextra -pedantic -Iinclude examples/rigid_c_abi.c -Licp_amd -licp_hip -lm -o icp_amd/rigid_c_abi
 *   icp_amd/rigid_c_abi [N [MAX_ITER]]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "icp_hip.h"

#define CHECK(call)                                                                         \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_ != ICP_OK) {                                                                \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, icp_last_error());          \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

static uint32_t lcg_state = 12345u;
static double lcg_uniform(void) { /* in [0, 1): numerical recipes' 32-bit LCG, top 32 bits as a fraction */
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (double)lcg_state / 4294967296.0;
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000;
    const int max_iter = argc > 2 ? atoi(argv[2]) : 50;
    if (n < 4 || max_iter < 1) {
        fprintf(stderr, "usage: %s [N >= 4 [MAX_ITER >= 1]]\n", argv[0]);
        return 2;
    }
    double *tgt = (double *)malloc((size_t)n * 3 * sizeof(double)), *src = (double *)malloc((size_t)n * 3 * sizeof(double));
    if (!tgt || !src) return 2;
    /* target: uniform in the unit cube; source: the target moved by a small rigid motion, rows reversed */
    const double ang = 0.05, c = cos(ang), s = sin(ang), shift[3] = {0.01, 0.005, -0.002};
    for (int64_t i = 0; i < n * 3; ++i) tgt[i] = lcg_uniform();
    for (int64_t i = 0; i < n; ++i) {
        const double *p = tgt + 3 * (n - 1 - i);
        const double x = p[0] - shift[0], y = p[1] - shift[1], z = p[2] - shift[2];
        src[3 * i] = c * x - s * y;
        src[3 * i + 1] = s * x + c * y;
        src[3 * i + 2] = z;
    }

    icp_ctx *ctx = NULL;
    CHECK(icp_ctx_create(0, &ctx));
    void *d_src, *d_tgt, *d_T, *d_hist, *d_iters, *d_idx, *d_nn, *d_Rt, *d_Rt2;
    CHECK(icp_malloc(ctx, (size_t)n * 3 * sizeof(double), &d_src));
    CHECK(icp_malloc(ctx, (size_t)n * 3 * sizeof(double), &d_tgt));
    CHECK(icp_malloc(ctx, 16 * sizeof(double), &d_T));
    CHECK(icp_malloc(ctx, (size_t)max_iter * sizeof(double), &d_hist));
    CHECK(icp_malloc(ctx, sizeof(int32_t), &d_iters));
    CHECK(icp_malloc(ctx, (size_t)n * sizeof(int32_t), &d_idx));
    CHECK(icp_malloc(ctx, (size_t)n * 3 * sizeof(double), &d_nn));
    CHECK(icp_malloc(ctx, 12 * sizeof(double), &d_Rt));
    CHECK(icp_malloc(ctx, 12 * sizeof(double), &d_Rt2));
    CHECK(icp_memcpy_h2d(ctx, d_src, src, (size_t)n * 3 * sizeof(double)));
    CHECK(icp_memcpy_h2d(ctx, d_tgt, tgt, (size_t)n * 3 * sizeof(double)));

    /* Trans = ICP(srcPts, tgtPts, 0.10, 50) of icp_3d.py:151, with the tolerance scaled to this cloud */
    CHECK(icp_rigid_run(ctx, (const double *)d_src, n, (const double *)d_tgt, n, 3, 1e-9, max_iter, 1, 0u, (double *)d_T,
                        (double *)d_hist, (int32_t *)d_iters, (int32_t *)d_idx));
    double T[16], *hist = (double *)malloc((size_t)max_iter * sizeof(double));
    int32_t iters = 0;
    CHECK(icp_memcpy_d2h(ctx, T, d_T, sizeof(T)));
    CHECK(icp_memcpy_d2h(ctx, &iters, d_iters, sizeof(iters)));
    CHECK(icp_memcpy_d2h(ctx, hist, d_hist, (size_t)max_iter * sizeof(double)));
    printf("iters %d\n", (int)iters);
    for (int i = 0; i < iters; ++i) printf("err %d %.17g\n", i, hist[i]);
    for (int r = 0; r < 4; ++r) printf("T %.17g %.17g %.17g %.17g\n", T[4 * r], T[4 * r + 1], T[4 * r + 2], T[4 * r + 3]);

    /* FindTransform(source, Match(...)) once more from the returned correspondences, both entry points */
    CHECK(icp_gather_points(ctx, (const double *)d_tgt, (const int32_t *)d_idx, n, 3, (double *)d_nn));
    CHECK(icp_find_transform(ctx, (const double *)d_src, (const double *)d_nn, n, 3, (double *)d_Rt, (double *)d_Rt + 9));
    CHECK(icp_find_transform_noalloc(ctx, (const double *)d_src, (const double *)d_nn, n, 3, (double *)d_Rt2, (double *)d_Rt2 + 9));
    double Rt[12], Rt2[12];
    CHECK(icp_memcpy_d2h(ctx, Rt, d_Rt, sizeof(Rt)));
    CHECK(icp_memcpy_d2h(ctx, Rt2, d_Rt2, sizeof(Rt2)));
    int same = 1;
    for (int k = 0; k < 12; ++k) same = same && Rt[k] == Rt2[k];
    printf("find_transform_noalloc_identical %d\n", same);
    for (int r = 0; r < 3; ++r) printf("R %.17g %.17g %.17g\n", Rt[3 * r], Rt[3 * r + 1], Rt[3 * r + 2]);
    printf("t %.17g %.17g %.17g\n", Rt[9], Rt[10], Rt[11]);
    int32_t *idx = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    CHECK(icp_memcpy_d2h(ctx, idx, d_idx, (size_t)n * sizeof(int32_t)));
    int64_t reversed = 0;
    for (int64_t i = 0; i < n; ++i) reversed += idx[i] == (int32_t)(n - 1 - i);
    printf("correspondences_recovered %lld of %lld\n", (long long)reversed, (long long)n);

    /* an error is a return code and a message, never an exception or an abort */
    const int rc = icp_rigid_run(ctx, (const double *)d_src, n, (const double *)d_tgt, n, 5, 0.1, max_iter, 1, 0u, (double *)d_T,
                                 NULL, (int32_t *)d_iters, NULL);
    printf("bad_dim_rc %d\n", rc);

    void *bufs[] = {d_src, d_tgt, d_T, d_hist, d_iters, d_idx, d_nn, d_Rt, d_Rt2};
    for (size_t k = 0; k < sizeof(bufs) / sizeof(bufs[0]); ++k) CHECK(icp_free(ctx, bufs[k]));
    CHECK(icp_ctx_destroy(ctx));
    free(tgt); free(src); free(hist); free(idx);
    return 0;
}
