// What would a persistent CG iteration cost at 10 000 vertices?  ("one persistent cooperative kernel
// per CG iteration chunk with bounded-spin grid barriers".)  A skeleton with the iteration's synchronisation structure and
// none of its arithmetic: B one-wave blocks, all resident, per "iteration"
//   1. every lane writes 12 doubles (its vertex's direction), grid barrier, gathers 4 neighbours' 12 doubles
//   2. every lane writes 3 doubles (its fit row),             grid barrier, gathers 4 neighbours' 3 doubles
//   3. every block writes one partial (p.Ap),                 grid barrier, every block sums the B partials
//   4. every block writes 14 partials (moments, r.z),         grid barrier, every block sums the B x 14 partials
// Barriers: one device-scope atomic add per block on a monotone counter, then polling loads, spin BOUNDED (a block that
// gives up raises a flag that every other block sees and leaves too).  Build: hipcc --offload-arch=gfx950 -O3 -o
// benchmarks/grid_barrier_probe benchmarks/grid_barrier_probe.hip ; run: benchmarks/grid_barrier_probe [blocks] [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Sync { unsigned int count; unsigned int abort; unsigned int pad0[30]; unsigned int flag; unsigned int pad1[31]; unsigned int sub[8][32]; };

// two levels: blocks of one XCD (blockIdx % 8) count on their own line; the last of them counts on the global line; the last
// of those publishes the epoch in a line of its own, which is the only one polled
__device__ __forceinline__ bool tree_barrier(Sync *sy, unsigned int &epoch, unsigned int nblocks) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        epoch += 1;
        const unsigned int x = blockIdx.x & 7u, members = (nblocks - x + 7u) / 8u;
        const unsigned int nx = nblocks < 8u ? nblocks : 8u;
        if (__hip_atomic_fetch_add(&sy->sub[x][0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == epoch * members)
            if (__hip_atomic_fetch_add(&sy->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == epoch * nx)
                __hip_atomic_store(&sy->flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int spins = 0;
        while (__hip_atomic_load(&sy->flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            if (++spins > (1u << 16) || __hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&sy->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
    }
    ok = __syncthreads_and(ok);
    return ok;
}

__device__ __forceinline__ bool grid_barrier(Sync *sy, unsigned int &epoch, unsigned int nblocks) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        epoch += nblocks;
        __hip_atomic_fetch_add(&sy->count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int spins = 0;
        while (__hip_atomic_load(&sy->count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            if (++spins > (1u << 16) || __hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&sy->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
    }
    ok = __syncthreads_and(ok);
    return ok;
}

// MODE 0: barriers only; 1: + the two block-sum exchanges; 2: + the two neighbour gathers
template <int MODE, bool TREE>
__global__ __launch_bounds__(64) void skeleton(Sync *sy, double *dirs, double *rows, double *part, int iters, int nv, double *out) {
    const unsigned int nb = gridDim.x;
    unsigned int epoch = 0;
    const int v = blockIdx.x * 64 + threadIdx.x;
    const int nbr[4] = {(v + 1) % nv, (v + nv - 1) % nv, (v + 100) % nv, (v + nv - 100) % nv};
    double acc = 0.0, p[12];
    for (int k = 0; k < 12; ++k) p[k] = v + k;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 2)
            for (int k = 0; k < 12; ++k) __hip_atomic_store(&dirs[(size_t)v * 12 + k], p[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(TREE ? tree_barrier(sy, epoch, nb) : grid_barrier(sy, epoch, nb))) return;
        double f[3] = {0, 0, 0};
        if (MODE >= 2) {
            for (int j = 0; j < 4; ++j)
                for (int k = 0; k < 12; ++k) f[k % 3] += __hip_atomic_load(&dirs[(size_t)nbr[j] * 12 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int k = 0; k < 3; ++k) __hip_atomic_store(&rows[(size_t)v * 3 + k], f[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!(TREE ? tree_barrier(sy, epoch, nb) : grid_barrier(sy, epoch, nb))) return;
        double g = 0.0;
        if (MODE >= 2)
            for (int j = 0; j < 4; ++j)
                for (int k = 0; k < 3; ++k) g += __hip_atomic_load(&rows[(size_t)nbr[j] * 3 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE >= 1 && threadIdx.x == 0) __hip_atomic_store(&part[(size_t)(it & 1) * nb * 16 + blockIdx.x], g + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(TREE ? tree_barrier(sy, epoch, nb) : grid_barrier(sy, epoch, nb))) return;
        double pap = 0.0;
        if (MODE >= 1) {
            for (unsigned int b = threadIdx.x; b < nb; b += 64) pap += __hip_atomic_load(&part[(size_t)(it & 1) * nb * 16 + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int o = 32; o > 0; o >>= 1) pap += __shfl_xor(pap, o, 64);
            if (threadIdx.x < 14) __hip_atomic_store(&part[(size_t)(2 + (it & 1)) * nb * 16 + blockIdx.x * 16 + threadIdx.x], pap + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!(TREE ? tree_barrier(sy, epoch, nb) : grid_barrier(sy, epoch, nb))) return;
        if (MODE >= 1) {
            double m = 0.0;
            for (unsigned int e = threadIdx.x; e < nb * 16; e += 64) m += (e & 15u) < 14u ? __hip_atomic_load(&part[(size_t)(2 + (it & 1)) * nb * 16 + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o, 64);
            for (int k = 0; k < 12; ++k) p[k] = p[k] * 0.5 + m * 1e-9;
            acc += m;
        }
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc + p[0];
}

template <int MODE, bool TREE>
static int run(int blocks, int iters, Sync *sy, double *dirs, double *rows, double *part, double *out, hipStream_t st) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemsetAsync(sy, 0, sizeof(Sync), st));
        CHECK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((skeleton<MODE, TREE>), dim3(blocks), dim3(64), 0, st, sy, dirs, rows, part, iters, blocks * 64, out);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    Sync h;
    CHECK(hipMemcpy(&h, sy, sizeof(h), hipMemcpyDeviceToHost));
    std::printf("%s mode %d: %d blocks, %d iterations of 4 barriers: %.2f us per iteration (%.2f us per barrier)%s\n", TREE ? "two-level barrier," : "one counter,", MODE, blocks, iters,
                best * 1e3 / iters, best * 1e3 / iters / 4, h.abort ? "  ** ABORTED (a block gave up spinning) **" : "");
    std::fflush(stdout);
    return h.abort ? 2 : 0;
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 157, iters = argc > 2 ? atoi(argv[2]) : 200;
    if (blocks < 1 || blocks > 1024 || iters < 1 || iters > 10000) return 1;
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    Sync *sy;
    double *dirs, *rows, *part, *out;
    CHECK(hipMalloc(&sy, sizeof(Sync)));
    CHECK(hipMalloc(&dirs, sizeof(double) * blocks * 64 * 12));
    CHECK(hipMalloc(&rows, sizeof(double) * blocks * 64 * 3));
    CHECK(hipMalloc(&part, sizeof(double) * 4 * blocks * 16));
    CHECK(hipMalloc(&out, sizeof(double) * blocks));
    CHECK(hipMemset(dirs, 0, sizeof(double) * blocks * 64 * 12));
    CHECK(hipMemset(rows, 0, sizeof(double) * blocks * 64 * 3));
    CHECK(hipMemset(part, 0, sizeof(double) * 4 * blocks * 16));
    if (int rc = run<0, false>(blocks, iters, sy, dirs, rows, part, out, st)) return rc;
    if (int rc = run<1, false>(blocks, iters, sy, dirs, rows, part, out, st)) return rc;
    if (int rc = run<2, false>(blocks, iters, sy, dirs, rows, part, out, st)) return rc;
    if (int rc = run<0, true>(blocks, iters, sy, dirs, rows, part, out, st)) return rc;
    if (int rc = run<1, true>(blocks, iters, sy, dirs, rows, part, out, st)) return rc;
    if (int rc = run<2, true>(blocks, iters, sy, dirs, rows, part, out, st)) return rc;
    return 0;
}
