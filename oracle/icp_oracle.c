/* CPU oracle in plain C -- TEST INFRASTRUCTURE ONLY (see oracle/icp_oracle.py).
 *
 * Restates the scalar semantics of the reference's closest-point search so that
 * (a) large parity cases finish in seconds (OpenMP over source points) and
 * (b) bench.py has a compiled CPU baseline to time on the GPU box's host cores.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Build: gcc -O3 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile); contraction
 * is disabled so every operation rounds like the reference's numpy / x86-64 code.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* d2 = ((dx*dx)+(dy*dy))+(dz*dz): icp_3d.py:40-41; nicp3d.cpp:56-63 accumulates
 * ((0+dx*dx)+dy*dy)+dz*dz, which is the same sequence of roundings. */
static inline double sqdist(const double *t, const double *s, int dim) {
    double acc = 0.0;
    for (int c = 0; c < dim; ++c) {
        double d = t[c] - s[c];
        acc += d * d;
    }
    return acc;
}

/* np.argmin semantics (icp_3d.py:42): lowest index on ties, first NaN wins.
 * compare_sqrt: nicp3d.cpp:142-164 -- strict '<' on sqrt(d2), starting from index 0
 * (a NaN at index 0 is never replaced, later NaNs never win). */
void oracle_match(const double *src, int64_t n, const double *tgt, int64_t m, int dim, int compare_sqrt,
                  int32_t *idx_out, double *d2_out, int threads) {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < n; ++i) {
        const double *s = src + i * dim;
        int64_t best_i = 0;
        double best_d2 = sqdist(tgt, s, dim);
        double best = compare_sqrt ? sqrt(best_d2) : best_d2;
        for (int64_t j = 1; j < m; ++j) {
            double d2 = sqdist(tgt + j * dim, s, dim);
            double key = compare_sqrt ? sqrt(d2) : d2;
            int take = compare_sqrt ? (key < best) : ((key < best) || (key != key && best == best));
            if (take) {
                best = key;
                best_d2 = d2;
                best_i = j;
            }
        }
        idx_out[i] = (int32_t)best_i;
        if (d2_out) d2_out[i] = best_d2;
    }
}

/* Distance, icp_3d.py:59-62 (sequential sum; the reference's pairwise sum differs in
 * the last bits only) */
double oracle_distance(const double *a, const double *b, int64_t n, int dim) {
    double tot = 0.0;
    for (int64_t i = 0; i < n; ++i) tot += sqrt(sqdist(a + i * dim, b + i * dim, dim));
    return tot;
}

/* the sums of FindTransform, icp_3d.py:69-78: means by sequential row-order sums / N,
 * W[a][b] += t1[a]*s1[b] point by point.  out: s_mean[dim], t_mean[dim], W[dim*dim] */
void oracle_kabsch_sums(const double *src, const double *nn, int64_t n, int dim, double *s_mean, double *t_mean,
                        double *W) {
    for (int c = 0; c < dim; ++c) s_mean[c] = t_mean[c] = 0.0;
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < dim; ++c) {
            s_mean[c] += src[i * dim + c];
            t_mean[c] += nn[i * dim + c];
        }
    for (int c = 0; c < dim; ++c) {
        s_mean[c] /= (double)n;
        t_mean[c] /= (double)n;
    }
    for (int k = 0; k < dim * dim; ++k) W[k] = 0.0;
    for (int64_t i = 0; i < n; ++i)
        for (int a = 0; a < dim; ++a)
            for (int b = 0; b < dim; ++b)
                W[a * dim + b] += (nn[i * dim + a] - t_mean[a]) * (src[i * dim + b] - s_mean[b]);
}

/* KNN with the C++ plug-in's tie rule (nicp3d.cpp:100-137): keep the k smallest
 * sqrt-distances, replace the current maximum only if it is strictly greater, i.e.
 * earlier indices win ties; output ascending by (distance, then heap order ~ index). */
void oracle_knn_rows(const double *v, int64_t n, int dim, int k, int32_t *idx_out, double *dist_out, int threads) {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t q = 0; q < n; ++q) {
        double bd[16];
        int32_t bi[16];
        int cnt = 0;
        for (int64_t j = 0; j < n; ++j) {
            double d = sqrt(sqdist(v + j * dim, v + q * dim, dim));
            if (cnt < k) {
                int p = cnt++;
                while (p > 0 && bd[p - 1] > d) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
                bd[p] = d; bi[p] = (int32_t)j;
            } else if (bd[k - 1] > d) {
                int p = k - 1;
                while (p > 0 && bd[p - 1] > d) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
                bd[p] = d; bi[p] = (int32_t)j;
            }
        }
        for (int p = 0; p < k; ++p) { idx_out[q * k + p] = bi[p]; dist_out[q * k + p] = bd[p]; }
    }
}
