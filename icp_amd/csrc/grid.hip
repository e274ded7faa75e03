// Exact nearest neighbour through a multi-resolution uniform grid.
//
// Same result as the brute-force scan of closest.hip, bit for bit: every candidate is
// evaluated with the reference formula d2 = ((dx*dx)+(dy*dy))+(dz*dz) in fp64 without
// FMA, candidates are compared by the total order (d2, index) -- lowest index on ties --
// and a cell is skipped / the search stopped only when a lower bound that is provably
// <= every computed d2 in the unvisited region is STRICTLY greater than the current best
// (a tie would have to be visited, because it may carry a lower index).
//
// Why the bounds are safe in floating point: rounding is monotone.  If a target point t
// lies in a region whose distance from the query along axis a is at least g_a (a double
// computed with downward slack), then |fl(t_a - s_a)| >= g_a, fl(dx*dx) >= fl(g_a*g_a),
// and adding / rounding non-negative terms keeps the inequality, so the computed d2 is
// >= ((gx*gx)+(gy*gy))+(gz*gz) evaluated the same way.
//
// Structure: targets are counting-sorted by the Morton code of their finest-level cell
// (2^b cells per axis).  A level-l cell (side h*2^l) is then the contiguous range
// start[code << DIM*l] .. start[(code+1) << DIM*l] of the sorted array, so ONE start
// array serves every level.  A query visits the 3^DIM neighbourhood of its cell at level
// 0, 1, 2, ... until the best distance is below the distance to the neighbourhood's
// boundary; sparse regions climb levels, dense regions stop at level 0.
//
// Targets containing NaN/inf are refused by the builder (flag) and the caller falls back
// to the brute-force kernels, whose "first NaN wins" rule needs a full scan anyway.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

#include "common.h"

namespace icp {

constexpr int GB = 256;

// device-side bbox record: min[3], max[3], nonfinite count
struct BBox {
    double mn[3], mx[3];
    unsigned long long bad;
    double qlo[3], qhi[3];  // per-axis 1 % / 99 % order statistics of a fixed-stride sample (quantile_kernel)
    // ordered sample pairs (i, j), self-pairs included, whose codes first share a cell at level l of the B-bit hierarchy
    // (occupancy_kernel), spread over kPairSub sub-counters so that the blocks' atomics do not queue on one address
    unsigned int pairs_u[14][8];
};
constexpr int kPairSub = 8;

// Its first 4096 / GB blocks also gather the fixed-stride sample of the statistics kernels into a compact per-axis array
// (samp[a * 4096 + i], +inf beyond ns; null: no sample wanted): 384 + 1024 blocks read it, and read straight from the
// cloud every one of them touched 4096 cache lines for 4096 numbers.
template <int DIM>
__global__ __launch_bounds__(GB) void bbox_kernel(const double *__restrict__ p, int64_t n, double *__restrict__ partial,
                                                  int64_t stride, int ns, double *__restrict__ samp) {
    if (samp && blockIdx.x < 4096 / GB) {
        const int i = blockIdx.x * GB + threadIdx.x;
#pragma unroll
        for (int a = 0; a < DIM; ++a) samp[a * 4096 + i] = i < ns ? p[(int64_t)i * stride * DIM + a] : __builtin_inf();
    }
    double mn[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()};
    double mx[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
    double bad = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * GB + threadIdx.x; i < n; i += (int64_t)gridDim.x * GB) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            const double v = p[i * DIM + c];
            if (!(fabs(v) < __builtin_inf())) bad += 1.0;  // NaN or inf
            else {
                mn[c] = v < mn[c] ? v : mn[c];
                mx[c] = v > mx[c] ? v : mx[c];
            }
        }
    }
    __shared__ double red[GB / 64][7];
    const int tid = threadIdx.x;
    double vals[7] = {mn[0], mn[1], mn[2], mx[0], mx[1], mx[2], bad};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        double v = vals[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double w = shfl_down_f64(v, o);
            v = k < 3 ? (w < v ? w : v) : (k < 6 ? (w > v ? w : v) : v + w);
        }
        if ((tid & 63) == 0) red[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid < 7) {
        double v = red[0][tid];
        for (int w = 1; w < GB / 64; ++w) {
            const double x = red[w][tid];
            v = tid < 3 ? (x < v ? x : v) : (tid < 6 ? (x > v ? x : v) : v + x);
        }
        partial[(int64_t)blockIdx.x * 8 + tid] = v;
    }
}

// one wave per quantity (7 waves): strided pass over the block partials, then a shuffle tree
__global__ void bbox_final_kernel(const double *__restrict__ partial, int nb, BBox *__restrict__ out) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k >= 7) return;
    if (threadIdx.x < 14 * kPairSub) (&out->pairs_u[0][0])[threadIdx.x] = 0u;   // occupancy_kernel adds into them
    double v = k < 3 ? __builtin_inf() : (k < 6 ? -__builtin_inf() : 0.0);
    for (int b = lane; b < nb; b += 64) {
        const double x = partial[(int64_t)b * 8 + k];
        v = k < 3 ? (x < v ? x : v) : (k < 6 ? (x > v ? x : v) : v + x);
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double x = shfl_down_f64(v, o);
        v = k < 3 ? (x < v ? x : v) : (k < 6 ? (x > v ? x : v) : v + x);
    }
    if (lane == 0) {
        if (k < 3) out->mn[k] = v;
        else if (k < 6) out->mx[k - 3] = v;
        else out->bad = (unsigned long long)v;
    }
}

__device__ __forceinline__ uint32_t spread3(uint32_t x) {  // 10 bits -> every third bit
    x &= 0x3ff;
    x = (x | (x << 16)) & 0x030000ff;
    x = (x | (x << 8)) & 0x0300f00f;
    x = (x | (x << 4)) & 0x030c30c3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
__device__ __forceinline__ uint32_t spread2(uint32_t x) {  // 16 bits -> every second bit
    x &= 0xffff;
    x = (x | (x << 8)) & 0x00ff00ff;
    x = (x | (x << 4)) & 0x0f0f0f0f;
    x = (x | (x << 2)) & 0x33333333;
    x = (x | (x << 1)) & 0x55555555;
    return x;
}
template <int DIM>
__device__ __forceinline__ uint32_t morton(const int *c) {
    if (DIM == 3) return spread3(c[0]) | (spread3(c[1]) << 1) | (spread3(c[2]) << 2);
    return spread2(c[0]) | (spread2(c[1]) << 1);
}

template <int DIM>
__device__ __forceinline__ void cell_of(const double *p, const GridParams &gp, int *c) {
    const int top = (1 << gp.bits) - 1;
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
        const double f = (p[a] - gp.mn[a]) * gp.inv_h;
        int ci = f >= 0.0 ? (f < (double)top ? (int)f : top) : 0;  // NaN -> 0
        c[a] = ci;
    }
}

// Robust extents (grid_build_into): the grid may cover less than the bounding box; points beyond it are clamped into
// the first / last column by cell_of, so those columns are UNBOUNDED outwards and the gap from a query to such a column
// must not count its outer face.  g / gh = signed gaps to the low / high face of column x of axis a (ncl columns).
__device__ __forceinline__ void open_sides(const GridParams &gp, int a, int x, int ncl, double &g, double &gh) {
    if (x <= 0 && ((gp.open >> (2 * a)) & 1)) g = -__builtin_inf();
    if (x >= ncl - 1 && ((gp.open >> (2 * a + 1)) & 1)) gh = -__builtin_inf();
}

// inclusive prefix sum over the 64 lanes of a wave by DPP moves (rows of 16: shifts by 1, 2, 4, 8; then the last lane of a
// row, and lane 31, broadcast to the rows behind them); six dependent vector instructions, no LDS round trips
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// LDS written by some lanes of a wave and read by others of the same wave: the hardware keeps a wave's LDS operations
// in order, this keeps the compiler from moving them across each other
__device__ __forceinline__ void lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sample statistics of a target build (robust extents, resolution from the data).  Rounds 1-2 sorted the sample twice
// with a bitonic network in ONE block each (quantile_kernel 37 us, occupancy_kernel 81 us: both on the critical path of
// a call's set-up, on one compute unit of 256).  Neither statistic needs an order: an order statistic is "the sample
// that exactly k others precede", and the pair counts are sums over all pairs -- 4096^2 comparisons each, which the
// whole chip does in a few microseconds.
constexpr int kQuantSamples = 4096;
constexpr int kQuantCand = 32;   // candidates per block of quantile_kernel (grid: dim x kQuantSamples / kQuantCand)

// k-th smallest and k-th largest coordinate of the sample, per axis: the quantiles that set the extents of the grid.
// Block (a, y): the 32 candidates 32y .. 32y+31 of axis a; thread (candidate c, part q) counts, over an eighth of the
// sample, the values below v[c] and the values not above it; v[c] IS the k-th order statistic iff lt <= k < le.  Tied
// candidates all qualify and all write the same value, so the result is what a sort would give, deterministically.
__global__ __launch_bounds__(256) void quantile_kernel(const double *__restrict__ samp, int ns, int k, BBox *__restrict__ out) {
    __shared__ double v[kQuantSamples];
    __shared__ int part[8][kQuantCand][2];
    const int a = blockIdx.x;
    for (int i = threadIdx.x; i < kQuantSamples; i += 256) v[i] = samp[a * kQuantSamples + i];   // (+inf beyond ns; targets are finite, or the build is refused)
    __syncthreads();
    const int cl = threadIdx.x & (kQuantCand - 1);
    const int c = blockIdx.y * kQuantCand + cl;
    const int q = threadIdx.x / kQuantCand;   // 8 parts of 512 samples
    const double vc = v[c];
    int lt = 0, le = 0;
    const int j0 = q * (kQuantSamples / 8);
#pragma unroll 8
    for (int j = j0; j < j0 + kQuantSamples / 8; ++j) {
        const double x = v[j];
        lt += x < vc ? 1 : 0;
        le += x <= vc ? 1 : 0;
    }
    part[q][cl][0] = lt;
    part[q][cl][1] = le;
    __syncthreads();
    if (threadIdx.x < kQuantCand && c < ns) {
        lt = 0;
        le = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            lt += part[w][cl][0];
            le += part[w][cl][1];
        }
        if (lt <= k && k < le) out->qlo[a] = vc;
        if (lt <= ns - 1 - k && ns - 1 - k < le) out->qhi[a] = vc;
    }
}

// How fine should the grid be?  What a query pays for is the population of the cells it visits, and that depends on how
// the points fill space, not only on how many there are: a volume-filling cloud, a Gaussian and a scanned surface of
// the same size want cells of different sizes (measured: 2^20 points, best finest level 6 / 7 / 8+ bits per axis).  The
// density-weighted mean population of a cell is 1 + N * P(two random points share a cell), and that probability is
// estimated here at EVERY level at once from the pairs of the sample: with the samples' Morton codes at the finest
// conceivable level (B bits per axis), two samples share a level-l cell iff their codes agree above dim*l bits, i.e.
// from level ceil(bitlength(a ^ b) / dim) upwards.  Block (x, y): rows 64x .. 64x+63 against the columns of sixteenth y,
// each of its four waves 64 of them (1024 blocks: four waves per SIMD hide the latency of the dependent chain per pair); pairs_u[l] += ordered pairs (self-pairs included: the host takes them out) whose
// FIRST shared level is l.  (Round 2 sorted the codes and let every sample count at most 32 predecessors: the same
// numbers wherever a cell holds at most 33 samples, i.e. at every level near the population the host looks for.)
constexpr int kOccB3 = 9, kOccB2 = 13;
__global__ __launch_bounds__(256) void occupancy_kernel(const double *__restrict__ samp, int ns, int dim, double clip,
                                                        BBox *__restrict__ io) {
    __shared__ uint32_t rowc[64];
    __shared__ uint32_t colc[kQuantSamples / 16];
    __shared__ unsigned int tot[14];
    const int B = dim == 3 ? kOccB3 : kOccB2;
    double mn[3] = {0, 0, 0}, ext = 0.0;
    for (int a = 0; a < dim; ++a) {  // the extents grid_build_finish will choose (same arithmetic)
        double lo = io->mn[a], hi = io->mx[a];
        const double wdt = io->qhi[a] - io->qlo[a];
        if (wdt > 0.0 && wdt < __builtin_inf()) {
            const double l2 = io->qlo[a] - clip * wdt, h2 = io->qhi[a] + clip * wdt;
            lo = l2 > lo ? l2 : lo;
            hi = h2 < hi ? h2 : hi;
        }
        mn[a] = lo;
        ext = hi - lo > ext ? hi - lo : ext;
    }
    if (!(ext > 0.0)) ext = 1.0;
    const double inv_h = 1.0 / (ext / (double)(1 << B) * (1.0 + 0x1p-30));
    const int top = (1 << B) - 1;
    auto code_of = [&](int i) -> uint32_t {
        int cell[3] = {0, 0, 0};
        for (int a = 0; a < dim; ++a) {
            const double f = (samp[a * kQuantSamples + i] - mn[a]) * inv_h;
            cell[a] = f >= 0.0 ? (f < (double)top ? (int)f : top) : 0;
        }
        return dim == 3 ? (spread3(cell[0]) | (spread3(cell[1]) << 1) | (spread3(cell[2]) << 2)) : (spread2(cell[0]) | (spread2(cell[1]) << 1));
    };
    constexpr int NC = kQuantSamples / 16;   // columns of a block
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * NC;
    if (threadIdx.x < 14) tot[threadIdx.x] = 0u;
    if (threadIdx.x < 64) rowc[threadIdx.x] = r0 + threadIdx.x < ns ? code_of(r0 + threadIdx.x) : 0u;
    for (int j = threadIdx.x; j < NC; j += 256) colc[j] = c0 + j < ns ? code_of(c0 + j) : 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t ci = rowc[lane];
    const bool row_ok = r0 + lane < ns;
    const int jb = wv * (NC / 4);
    int jn = ns - (c0 + jb);               // columns of this wave that are real samples (wave-uniform)
    jn = jn < 0 ? 0 : (jn > NC / 4 ? NC / 4 : jn);
    // 64 pairs per lane: the 8-bit fields of h0 / h1 (one per first-shared level) cannot overflow
    unsigned long long h0 = 0ull, h1 = 0ull;
    auto pair = [&](int j) {
        const uint32_t x = ci ^ colc[jb + j];   // (every lane reads the same word: a broadcast)
        const int nb = x ? 32 - __clz((int)x) : 0;
        const int lmin = dim == 3 ? (nb + 2) / 3 : (nb + 1) >> 1;
        const unsigned long long one = 1ull << (8 * (lmin & 7));
        h0 += lmin < 8 ? one : 0ull;
        h1 += lmin < 8 ? 0ull : one;
    };
    static_assert(NC / 4 == 64, "a wave takes 64 columns");
    if (jn == 64) {
#pragma unroll 16
        for (int j = 0; j < 64; ++j) pair(j);
    } else {
        for (int j = 0; j < jn; ++j) pair(j);
    }
    unsigned int cnt[14];
#pragma unroll
    for (int l = 0; l < 14; ++l) cnt[l] = (unsigned int)(((l < 8 ? h0 : h1) >> (8 * (l & 7))) & 0xffull);
#pragma unroll
    for (int l = 0; l < 14; ++l) {
        unsigned int v = row_ok ? cnt[l] : 0u;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0 && v) atomicAdd(&tot[l], v);
    }
    __syncthreads();
    if (threadIdx.x <= B && threadIdx.x < 14 && tot[threadIdx.x])
        atomicAdd(&io->pairs_u[threadIdx.x][(blockIdx.x + blockIdx.y) & (kPairSub - 1)], tot[threadIdx.x]);
}

template <int DIM>
__global__ __launch_bounds__(GB) void cell_count_kernel(const double *__restrict__ tgt, int64_t m, GridParams gp,
                                                       uint32_t *__restrict__ code_out, uint32_t *__restrict__ count,
                                                       uint32_t *__restrict__ rank_out) {
    const int64_t i = (int64_t)blockIdx.x * GB + threadIdx.x;
    if (i >= m) return;
    double p[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) p[a] = tgt[i * DIM + a];
    int c[DIM];
    cell_of<DIM>(p, gp, c);
    const uint32_t code = morton<DIM>(c);
    code_out[i] = code;
    // rank_out (unordered fill): the point's arrival number in its cell is its slot there -- the fill then needs no second
    // round of atomics
    if (rank_out) rank_out[i] = atomicAdd(&count[code], 1u);
    else atomicAdd(&count[code], 1u);
}

// the cell codes alone (a build that only wants the Morton ORDER of its points: no cell ranges, no counters)
template <int DIM>
__global__ __launch_bounds__(GB) void cell_code_kernel(const double *__restrict__ pts, int64_t m, GridParams gp,
                                                      uint32_t *__restrict__ code_out) {
    const int64_t i = (int64_t)blockIdx.x * GB + threadIdx.x;
    if (i >= m) return;
    double p[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) p[a] = pts[i * DIM + a];
    int c[DIM];
    cell_of<DIM>(p, gp, c);
    code_out[i] = morton<DIM>(c);
}

// exclusive scan of `count[0..len)` in three passes; block sums in `bsum`
constexpr int SCAN_ITEMS = 8;  // per thread
__global__ __launch_bounds__(GB) void scan_block_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                       uint32_t *__restrict__ bsum, int64_t len) {
    __shared__ uint32_t wsum[GB / 64];
    const int tid = threadIdx.x;
    const int64_t base = ((int64_t)blockIdx.x * GB + tid) * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], tot = 0;
    const bool whole = base + SCAN_ITEMS <= len;   // (in / out are 256-byte aligned scratch: two 16-byte accesses per thread)
    if (whole) {
        const uint4 a = ((const uint4 *)(in + base))[0], b = ((const uint4 *)(in + base))[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) v[k] = base + k < len ? in[base + k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) tot += v[k];
    // inclusive wave scan of tot
    uint32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if ((tid & 63) >= o) inc += t;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < (tid >> 6); ++w) woff += wsum[w];
    uint32_t run = woff + inc - tot;
    if (whole) {
        uint32_t o[SCAN_ITEMS];
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            o[k] = run;
            run += v[k];
        }
        ((uint4 *)(out + base))[0] = make_uint4(o[0], o[1], o[2], o[3]);
        ((uint4 *)(out + base))[1] = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            if (base + k < len) out[base + k] = run;
            run += v[k];
        }
    }
    if (tid == GB - 1) bsum[blockIdx.x] = woff + inc;
}
__global__ void scan_sums_kernel(uint32_t *__restrict__ bsum, int nb) {  // single thread-block, sequential chunks
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        uint32_t v = i < nb ? bsum[i] : 0u;
        // block-wide exclusive scan over 1024 threads via LDS
        __shared__ uint32_t tmp[1024];
        tmp[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            uint32_t t = threadIdx.x >= o ? tmp[threadIdx.x - o] : 0u;
            __syncthreads();
            tmp[threadIdx.x] += t;
            __syncthreads();
        }
        const uint32_t incl = tmp[threadIdx.x];
        if (i < nb) bsum[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
}
__global__ __launch_bounds__(GB) void scan_add_kernel(uint32_t *__restrict__ out, const uint32_t *__restrict__ bsum,
                                                     int64_t len, uint32_t total_at_end) {
    const int64_t base = ((int64_t)blockIdx.x * GB + threadIdx.x) * SCAN_ITEMS;
    const uint32_t off = bsum[blockIdx.x];
    if (base + SCAN_ITEMS <= len) {
        uint4 a = ((uint4 *)(out + base))[0], b = ((uint4 *)(out + base))[1];
        a.x += off; a.y += off; a.z += off; a.w += off; b.x += off; b.y += off; b.z += off; b.w += off;
        ((uint4 *)(out + base))[0] = a;
        ((uint4 *)(out + base))[1] = b;
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k)
            if (base + k < len) out[base + k] += off;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[len] = total_at_end;
}

// Gather into Morton order: sorted position `pos` holds point sidx[pos], the permutation of the stable code sort
// (stablesort.hip: cells in code order, ascending index inside a cell -- the same order on every run).
// Records readable past the last one (copies of it).  The per-lane search reads records in groups of UNR without
// clamping to the end of a cell's range: what lies beyond is the next cell's records -- real targets, which any exact
// search may look at -- and, at the very end of the array, these copies (same index, same distance: never preferred).
constexpr int kRecPad = 7;

// fp32 PRE-FILTER of the per-lane search.  Beside its 32-byte record every target has a 16-byte one: its
// coordinates in cell units, u = (x - mn) * inv_h, rounded to float.  The record walk forms the squared distance of those
// to the query's in fp32 (half the bytes through the L1 path, a third of the vector-ALU cycles) and looks at the fp64 record
// only when that cannot prove the fp64 d2 larger than what it is tested against.  The proof: for |u| <= 1.5 * 2^bits
// (kFilterLo / kFilterHi below; anything else is stored as NaN, which never proves anything) each float coordinate is
// within e1 = 2^-24 * 1.5 * 2^bits of the exact one, so with N the norm of the fp32 differences the true distance in cell
// units is at least N (1 - 2^-24) - 2 sqrt(3) e1, the computed fp32 sum of squares is at most N^2 (1 + 2^-22), and the
// computed fp64 d2 is at least the true one times (1 - 1e-15): a candidate with
//     d2_f32 > [(sqrt(P) * inv_h + 2 sqrt(3) e1 * 1.001) * (1 + 1e-6)]^2      (rounded up to float)
// has a computed fp64 d2 > P and can neither win nor tie against P (the best d2, or the second best in the BND forms).
// Records in the padding are copies of the last one, as in the fp64 array.
constexpr double kFilterLo = -0.5, kFilterHi = 1.5;   // in units of 2^bits cells
template <int DIM>
__device__ __forceinline__ float4 filter_record(double x, double y, double z, const GridParams &gp) {
    const double nc = (double)(1 << gp.bits), lo = kFilterLo * nc, hi = kFilterHi * nc;
    const double u0 = (x - gp.mn[0]) * gp.inv_h, u1 = (y - gp.mn[1]) * gp.inv_h, u2 = DIM == 3 ? (z - gp.mn[2]) * gp.inv_h : 0.0;
    const bool safe = u0 >= lo && u0 <= hi && u1 >= lo && u1 <= hi && u2 >= lo && u2 <= hi;   // (NaN / inf: false)
    return make_float4(safe ? (float)u0 : __builtin_nanf(""), (float)u1, (float)u2, 0.0f);
}
__device__ __forceinline__ float4 *filter_records(double4 *recs, int64_t m) { return (float4 *)(recs + m + kRecPad); }
__device__ __forceinline__ const float4 *filter_records(const double4 *recs, int64_t m) { return (const float4 *)(recs + m + kRecPad); }

template <int DIM>
__global__ __launch_bounds__(GB) void scatter_kernel(const double *__restrict__ tgt, uint32_t *code,
                                                    int64_t m, double *__restrict__ spts, const int32_t *__restrict__ sidx,
                                                    double4 *__restrict__ recs, GridParams gp) {
    const int64_t pos = (int64_t)blockIdx.x * GB + threadIdx.x;
    if (pos >= m) return;
    const int64_t i = sidx[pos];
    // one 32-byte record per target: coordinates + the original index in the 4th slot, so a
    // candidate costs two 16-byte loads from a single cache line
    double4 rec;
    rec.x = tgt[i * DIM];
    rec.y = tgt[i * DIM + 1];
    rec.z = DIM == 3 ? tgt[i * DIM + 2] : 0.0;
    rec.w = __longlong_as_double((long long)i);
    recs[pos] = rec;
    const float4 frec = filter_record<DIM>(rec.x, rec.y, rec.z, gp);
    float4 *recs32 = filter_records(recs, m);
    recs32[pos] = frec;
    if (pos == m - 1)   // kRecPad copies of the last record: the searches read whole groups of records past a range's end
        for (int k = 1; k <= kRecPad; ++k) {
            recs[pos + k] = rec;
            recs32[pos + k] = frec;
        }
#pragma unroll
    for (int a = 0; a < DIM; ++a) spts[(int64_t)pos * DIM + a] = tgt[i * DIM + a];
    code[i] = (uint32_t)pos;  // the cell code is spent: the array becomes the inverse permutation (GridBuild::inv)
}

// order-only build: the points in Morton order, nothing else
template <int DIM>
__global__ __launch_bounds__(GB) void gather_order_kernel(const double *__restrict__ pts, int64_t m, double *__restrict__ spts,
                                                         const int32_t *__restrict__ sidx) {
    const int64_t pos = (int64_t)blockIdx.x * GB + threadIdx.x;
    if (pos >= m) return;
    const int64_t i = sidx[pos];
#pragma unroll
    for (int a = 0; a < DIM; ++a) spts[pos * DIM + a] = pts[i * DIM + a];
}

// The same scatter without the sort, for grids that are only SEARCHED (targets): a point takes the next free slot of
// its cell (the counts of cell_count_kernel are counted down), so the order inside a cell varies from run to run.
// Nothing a search returns depends on it -- every comparison is on (distance, original index), positions are
// translated back through sidx -- but sums taken over the sorted order would, so the Morton order of a SOURCE
// (icp_rigid_run's reductions, icp_morton_sort) comes from the stable sort above.  1 M points: 45 us against
// 160 (sort) + 43 (gather).
template <int DIM>
__global__ __launch_bounds__(GB) void scatter_atomic_kernel(const double *__restrict__ tgt, uint32_t *code, const uint32_t *__restrict__ rank,
                                                           const uint32_t *__restrict__ start, int64_t m,
                                                           double *__restrict__ spts, int32_t *__restrict__ sidx,
                                                           double4 *__restrict__ recs, GridParams gp) {
    const int64_t i = (int64_t)blockIdx.x * GB + threadIdx.x;
    if (i >= m) return;
    const uint32_t c = code[i];
    const uint32_t pos = start[c] + rank[i];
    double4 rec;
    rec.x = tgt[i * DIM];
    rec.y = tgt[i * DIM + 1];
    rec.z = DIM == 3 ? tgt[i * DIM + 2] : 0.0;
    rec.w = __longlong_as_double((long long)i);
    recs[pos] = rec;
    const float4 frec = filter_record<DIM>(rec.x, rec.y, rec.z, gp);
    float4 *recs32 = filter_records(recs, m);
    recs32[pos] = frec;
    if ((int64_t)pos == m - 1)
        for (int k = 1; k <= kRecPad; ++k) {
            recs[pos + k] = rec;
            recs32[pos + k] = frec;
        }
    spts[(int64_t)pos * DIM] = rec.x;
    spts[(int64_t)pos * DIM + 1] = rec.y;
    if (DIM == 3) spts[(int64_t)pos * DIM + 2] = rec.z;
    sidx[pos] = (int32_t)i;
    code[i] = pos;  // (GridBuild::inv, as in scatter_kernel)
}

__device__ __forceinline__ bool better2(double da, int32_t ia, double db, int32_t ib) {
    const bool an = da != da, bn = db != db;
    if (an || bn) return (an && bn) ? (ia < ib) : an;
    return da < db || (da == db && ia < ib);
}


// LE = false: argmin (d2, index).  LE = true: lowest index with d2 <= thr[q].
// A query the per-lane pass gave up on after `cap` levels: state to resume from
struct alignas(16) Deferred {   // 64 bytes: four 16-byte accesses
    int32_t q, level, bi;
    uint32_t bp;  // sorted position of the best record so far
    double best;
    double s[3];  // the query as the per-lane pass searched it (moved, in the fused loop): saves the second pass a dependent load
    double second;  // BND searches: least d2 among the OTHER records evaluated so far
};

// FUSED (the rigid loop from its second iteration on, kabsch.hip): the query is src[q] moved by the transform the last
// solve left in device memory (the arithmetic of apply_err_kernel, so the same bits as a materialised copy), and the
// distance to its hint -- last iteration's correspondence -- is that iteration's error term (icp_3d.py:113-114):
// the block's sum of them goes to err_part[block], which saves the loop a streaming kernel and a reduction launch.
//
// BND (the coherent rigid loop, kabsch.hip): a TWO-nearest search.  Cells are pruned and the search is
// stopped against the second-best d2 instead of the best (the returned argmin is the same: less is pruned), so that
// `second` ends as the least computed d2 over ALL targets other than the one returned -- every target is evaluated, or
// lies in a cell pruned on bound > second, or beyond a box left on second < gap^2 -- and it is stored per query (lb2)
// together with the query's position (sref).
// COH = 1 (FUSED && BND; the test pass of a coherent iteration): form the moved query and the error term as FUSED does,
// then try to PROVE the hinted record still nearest: with delta = |s - sref|, every other target is at least
// sqrt(lb2) - delta away (triangle inequality, with slack for every rounding involved), so if the hinted record is nearer
// than that no other target's computed d2 can reach, let alone tie, its own: the full scan would return the same index.
// Such a query keeps its record; the others go straight to the deferred pass (one half-wave per query: a few thousand
// scattered queries are a latency problem -- run through the per-lane kernel they took 70 us, the longest lane of the few
// resident waves), as Deferred records seeded with the hinted record and with an upper bound on the new second-best d2
// (the two targets nearest at sref are both within sqrt(lb2) + delta now).  Block b compacts its records into ITS OWN
// region of the list (QB slots) and writes their number to coh_count[b]: no atomics (16 000 per-wave atomics on one
// counter took 170 us, 4 096 per-block ones on sixteen counters still 40 us of a 57 us pass: now 17 us).
template <int DIM, bool LE, int QB, int WPE, bool FUSED = false, int LISTN = 4, int UNRN = 3, bool BND = false, int COH = 0>
__global__ __launch_bounds__(QB) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void grid_query_kernel(const double *__restrict__ src, int64_t n,
                                                       const int32_t *__restrict__ qorder,
                                                       const double4 *__restrict__ recs,
                                                       const uint32_t *__restrict__ start, GridParams gp,
                                                       const double *__restrict__ thr_in, int32_t *idx_out,
                                                       double *__restrict__ d2_out, const int32_t *__restrict__ done,
                                                       long long *__restrict__ dbg, int cap,
                                                       Deferred *__restrict__ deferred, uint32_t *__restrict__ n_deferred,
                                                       const int32_t *hint, const uint32_t *__restrict__ inv, uint32_t m,
                                                       int pos_mode, int xcd_chunk, const double *__restrict__ Rm = nullptr,
                                                       const double *__restrict__ tv = nullptr,
                                                       double *__restrict__ err_part = nullptr,
                                                       double *__restrict__ lb2_io = nullptr, double *__restrict__ sref_io = nullptr,
                                                       uint32_t *__restrict__ coh_count = nullptr,
                                                       const uint32_t *__restrict__ gate = nullptr, uint32_t defer_max = 0xffffffffu) {
    static_assert(!BND || !LE, "bounds belong to the argmin search");
    if (gate && *gate == 0u) return;   // (coherent iterations: the whole-cloud search runs only when too few rows were proven)
    static_assert(COH == 0 || (BND && FUSED), "the coherent passes are forms of the fused, bounded search");
    if (done && *done) return;
    // XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one), each
    // with its own 4 MiB L2, and the queries arrive in Morton order.  The launch is cut into super-chunks of
    // 8 * xcd_chunk consecutive blocks; inside one, XCD k takes the k-th run of xcd_chunk consecutive blocks, so the
    // blocks resident in one XCD at a time are neighbours in space and share target records in its L2 (host side,
    // grid_query(): how xcd_chunk is chosen, with the measurements).
    int64_t blk = blockIdx.x;
    if (xcd_chunk > 0) {
        const uint32_t per = 8u * (uint32_t)xcd_chunk, sc = blockIdx.x / per, r = blockIdx.x - sc * per;
        blk = (int64_t)sc * per + (int64_t)(r & 7u) * xcd_chunk + (r >> 3);
    }
    const int64_t t = blk * QB + threadIdx.x;
    if (!FUSED && t >= n) return;
    const bool in_range = t < n;     // FUSED: every thread stays until the block's error sum is formed
    const long long t_begin = dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
    int dbg_levels = 0, dbg_cand = 0;
    const int64_t q = in_range ? (qorder ? (int64_t)qorder[t] : t) : 0;
    double s[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) s[a] = src[q * DIM + a];
    if (FUSED) {   // ss = src R + t, operation for operation as apply_err_kernel forms it
        double o[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            double v = s[0] * Rm[c];
#pragma unroll
            for (int k = 1; k < DIM; ++k) v += s[k] * Rm[k * DIM + c];
            o[c] = v + tv[c];
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c) s[c] = o[c];
    }
    int c0[DIM];
    cell_of<DIM>(s, gp, c0);
    double slack_abs[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) slack_abs[a] = (fabs(gp.mn[a]) + fabs(gp.mx[a])) * 0x1p-48;

    double best = __builtin_inf();
    int32_t bi = 0x7fffffff;
    const double thr = LE ? thr_in[q] : 0.0;
    // Targets are finite (the builder refuses others).  A query with a NaN / inf coordinate
    // therefore sees d2 = NaN (or +inf) against EVERY target, and np.argmin of a constant
    // NaN / inf array is index 0.  For finite queries no d2 can be NaN, so the plain
    // (d2, index) comparison below is the whole of np.argmin's order.
    {
        bool finite = true, isnan = false;
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
            finite = finite && fabs(s[a]) < __builtin_inf();
            isnan = isnan || s[a] != s[a];
        }
        if (FUSED) {
            // error term of the previous iteration: ||nn - ss|| with nn the hinted record (d2 summed as apply_err_kernel
            // sums it: ((0 + dx^2) + dy^2) + dz^2); a non-finite query contributes its NaN / inf like there
            double e = 0.0;
            double hd2 = 0.0;        // COH: the computed d2 to the hinted record (eval()'s arithmetic), whether the bound
            bool keep = false;       // taken at the last search still proves it the nearest, and, where it does not, an
            double sec0 = __builtin_inf();   // upper bound on the second-best d2 the search will find
            int32_t hid = 0x7fffffff;
            uint32_t hpos = 0;
            if (in_range) {
                const uint32_t h = (uint32_t)hint[q];
                double lb2 = 0.0, sr[DIM];
                if (COH == 1) {      // (issued with the record's load)
                    lb2 = lb2_io[q];
#pragma unroll
                    for (int a = 0; a < DIM; ++a) sr[a] = sref_io[q * DIM + a];
                }
                const double4 rc = recs[h < m ? h : 0u];   // (always in range inside the loop; never fault on a bad one)
                const double dx = rc.x - s[0], dy = rc.y - s[1];
                double d = dx * dx + dy * dy;
                if (DIM == 3) {
                    const double dz = rc.z - s[2];
                    d = d + dz * dz;
                }
                e = sqrt(d);
                if (COH == 1) {
                    // every other target t had computed d2(s_ref, t) >= lb2, so its true distance from s_ref is at least
                    // sqrt(lb2) (1 - 3e-16), from s at least that minus |s - s_ref|, and its computed d2 from s at least the
                    // square of that times (1 - 6e-16): a relative slack of 1e-12 on each factor covers all of it
                    // (and 1e-280 keeps every quantity far above the underflow range, where those relative errors hold)
                    hd2 = d;
                    double m2 = 0.0;
#pragma unroll
                    for (int a = 0; a < DIM; ++a) m2 += (s[a] - sr[a]) * (s[a] - sr[a]);
                    const double rl = __dsqrt_rn(lb2), dl = __dsqrt_rn(m2);
                    const double L = rl * (1.0 - 1e-12) - dl * (1.0 + 1e-12);
                    keep = h < m && L > 0.0 && lb2 > 1e-280 && L * L * (1.0 - 1e-12) > d && L * L > 1e-280;
                    // the two targets nearest at sref lie within rl (1 + 3e-16) of it, hence within that + |s - sref| of s:
                    // the search will find a second-best d2 no larger than this (NaN / inf: nothing is pruned on it)
                    const double U = (rl + dl) * (1.0 + 1e-12);
                    sec0 = U * U * (1.0 + 1e-12);
                    if (h < m) {
                        hid = (int32_t)__double_as_longlong(rc.w);
                        hpos = h;
                    } else {
                        hd2 = __builtin_inf();
                        sec0 = __builtin_inf();
                    }
                }
            }
            __shared__ double ered[QB / 64];
            const double w = wave_sum(e);
            if ((threadIdx.x & 63) == 0) ered[threadIdx.x >> 6] = w;
            __syncthreads();
            if (threadIdx.x == 0) {
                double v = ered[0];
#pragma unroll
                for (int k = 1; k < QB / 64; ++k) v += ered[k];
                err_part[blk] = v;
            }
            if (COH == 1) {
                bool finite = true;
#pragma unroll
                for (int a = 0; a < DIM; ++a) finite = finite && fabs(s[a]) < __builtin_inf();
                if (in_range && !finite) {   // (np.argmin of a constant NaN / inf row: index 0; never kept)
                    idx_out[q] = (int32_t)inv[0];
                    lb2_io[q] = 0.0;
                }
                if (in_range && finite && keep) idx_out[q] = hint[q];   // fused loop: hints and results are sorted positions
                // the others: compacted over the block into the block's own region of the list
                const bool go = in_range && finite && !keep;
                __shared__ uint32_t wcnt[QB / 64];
                const unsigned long long gm = __builtin_amdgcn_ballot_w64(go);
                const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
                if (ln == 0) wcnt[wv] = (uint32_t)__builtin_popcountll(gm);
                __syncthreads();
                uint32_t off = (uint32_t)__builtin_popcountll(gm & ((1ull << ln) - 1ull)), tot = 0;
#pragma unroll
                for (int k = 0; k < QB / 64; ++k) {
                    off += k < wv ? wcnt[k] : 0u;
                    tot += wcnt[k];
                }
                if (threadIdx.x == 0) coh_count[blk] = tot;
                if (go) {
                    Deferred dq;
                    dq.q = (int32_t)q;
                    dq.level = 0;
                    dq.bi = hid;
                    dq.bp = hpos;
                    dq.best = hd2;
#pragma unroll
                    for (int a = 0; a < 3; ++a) dq.s[a] = a < DIM ? s[a] : 0.0;
                    dq.second = sec0;
                    deferred[(size_t)blk * QB + off] = dq;
                }
                return;
            }
            if (!in_range) return;
            (void)hd2;
            (void)sec0;
            (void)hid;
            (void)hpos;
        }
        if (!finite) {
            if (BND) lb2_io[q] = 0.0;   // never kept
            idx_out[q] = pos_mode ? (int32_t)inv[0] : 0;
            if (d2_out) d2_out[q] = LE ? __builtin_inf() : (isnan ? __builtin_nan("") : __builtin_inf());
            return;
        }
    }
    constexpr int NBR = DIM == 3 ? 27 : 9;
    // Per-lane work list of candidate ranges in LDS ([slot][lane] layout: conflict-free).  Lanes of
    // a wave need different neighbour cells with different populations; walking the 27 cells in
    // lockstep would cost (cells x longest range) trips per wave.  Instead each lane first lists
    // the ranges that survive its bound test, then all lanes walk their OWN concatenated list, so
    // a wave's trip count is the longest TOTAL, not the sum of per-cell maxima.
    constexpr int LIST = LISTN;
    constexpr int SCANN = UNRN;   // records per trip, own-cell scan and list walk alike
    __shared__ uint2 wl[LIST][QB];
    __shared__ float wlb[LIST][QB];  // lower bound of each listed cell (rounded down)
    const int lane_slot = threadIdx.x;

    uint32_t bp = 0;  // sorted position of the best record (what pos_mode returns instead of its index)
    double second = __builtin_inf();   // BND: least d2 among the evaluated records other than the best one
    // what a cell's bound / the gap is tested against: the best d2, or (BND) the second best
#define ICP_PRUNE_AT (BND ? second : best)
    auto eval = [&](const double4 &rc, uint32_t pos) {
        const double dx = rc.x - s[0], dy = rc.y - s[1];
        double d = dx * dx + dy * dy;
        if (DIM == 3) {
            const double dz = rc.z - s[2];
            d = d + dz * dz;
        }
        const int32_t id = (int32_t)__double_as_longlong(rc.w);
        if (BND) {   // (a record may be met more than once: as the hint, past the end of a range, again one level up)
            const bool better = d < best || (d == best && id < bi);
            second = better ? best : ((id != bi && d < second) ? d : second);
        }
        if (LE) {
            if (d <= thr && id < bi) {
                bi = id;
                bp = pos;
                best = 0.0;
            }
        } else if (d < best || (d == best && id < bi)) {
            best = d;
            bi = id;
            bp = pos;
        }
    };
    // fp32 pre-filter (filter_record above): the query in cell units, the threshold the fp32 d2 is tested against, and the
    // position of the best record so far (meeting it again changes nothing: skipped without its fp64 record)
    constexpr bool FILT = !LE;
    const float4 *recs32 = filter_records(recs, (int64_t)m);
    float uq[3] = {0.0f, 0.0f, 0.0f};
    bool f_ok = FILT && gp.h > 1e-100 && gp.h < 1e100;   // (far from the under- and overflow range: the relative error bounds hold)
    if (FILT) {
        const double nc = (double)(1 << gp.bits);
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
            const double u = (s[a] - gp.mn[a]) * gp.inv_h;
            f_ok = f_ok && u >= kFilterLo * nc && u <= kFilterHi * nc;
            uq[a] = (float)u;
        }
    }
    const double f_eps = 3.4641016151377544 * 1.001 * (0x1p-24 * kFilterHi * (1.0 + 1e-7)) * (double)(1 << gp.bits);
    float f_thr = __builtin_inff();
    uint32_t f_skip = 0xffffffffu;
    auto refresh = [&]() {   // after every change of best / second
        const double r = (__dsqrt_rn(ICP_PRUNE_AT) * gp.inv_h + f_eps) * (1.0 + 1e-6);
        f_thr = f_ok ? __double2float_ru(r * r) : __builtin_inff();
        f_skip = bi == 0x7fffffff ? 0xffffffffu : bp;
    };
    // SCANN records from position p on: the fp32 test for all of them, the fp64 records of those it cannot reject
    auto walk = [&](uint32_t p) {
        if (FILT) {
            float4 rf[SCANN];
            const float4 *b32 = recs32 + p;   // no clamp at the range's end: see kRecPad
#pragma unroll
            for (int u = 0; u < SCANN; ++u) rf[u] = b32[u];
            bool pass[SCANN], any = false;
#pragma unroll
            for (int u = 0; u < SCANN; ++u) {
                const float dx = rf[u].x - uq[0], dy = rf[u].y - uq[1];
                float d = __builtin_fmaf(dy, dy, dx * dx);   // (fused: fewer roundings than the bound allows for)
                if (DIM == 3) {
                    const float dz = rf[u].z - uq[2];
                    d = __builtin_fmaf(dz, dz, d);
                }
                pass[u] = !(d > f_thr) && p + (uint32_t)u != f_skip;   // (NaN: not rejected)
                any = any || pass[u];
            }
            if (any) {
#pragma unroll
                for (int u = 0; u < SCANN; ++u)
                    if (pass[u]) eval(recs[p + u], p + u);
                refresh();
            }
        } else {
            double4 rc[SCANN];
            const double4 *base = recs + p;   // no clamp at the range's end: see kRecPad
#pragma unroll
            for (int u = 0; u < SCANN; ++u) rc[u] = base[u];
#pragma unroll
            for (int u = 0; u < SCANN; ++u) eval(rc[u], p + u);
        }
    };
    auto scan = [&](uint32_t p0, uint32_t p1) {
        for (uint32_t p = p0; p < p1; p += SCANN) walk(p);
    };

    // ICP_HINT_IDX: a target the caller already knows (last iteration's match) is an ordinary candidate,
    // evaluated first.  Its distance prunes most neighbour cells at once; the answer is unchanged because
    // every cell that could hold a better or tying point is still visited.  (hint may alias idx_out:
    // each lane reads its entry here and writes it at the very end.)
    if (hint) {
        const uint32_t h = (uint32_t)hint[q];
        if (h < m) {
            const uint32_t hp = pos_mode ? h : inv[h];   // pos_mode: hints are sorted positions too
            eval(recs[hp], hp);
        }
    }
    if (FILT) refresh();
    bool may_defer = true;
    for (int l = 0; l <= gp.bits; ++l) {
        if (l >= cap && may_defer) {
            // A few queries (sparse regions, far outliers) need many levels and would hold their whole
            // wave -- and, at the end of the launch, the whole GPU -- for hundreds of microseconds.
            // They are handed to grid_defer_kernel, which gives each of them a wave of its own.
            // (the list holds a quarter of the queries: a lane that finds it full keeps its query and climbs on by itself --
            //  slower for its wave, exact either way; grid_defer_kernel reads min(counter, capacity))
            const uint32_t slot = atomicAdd(n_deferred, 1u);
            may_defer = slot < defer_max;
            if (may_defer) {
            Deferred dq;
            dq.q = (int32_t)q;
            dq.level = l;
            dq.bi = bi;
            dq.bp = bp;
            dq.best = best;
#pragma unroll
            for (int a = 0; a < 3; ++a) dq.s[a] = a < DIM ? s[a] : 0.0;
            dq.second = second;
            deferred[slot] = dq;
            return;
            }
        }
        ++dbg_levels;
        const double hl = gp.h * (double)(1 << l);
        const int ncl = 1 << (gp.bits - l);
        const double slack_h = hl * 0x1p-20;
        int cl[DIM];
        // per axis and column offset (-1, 0, +1): squared gap (monotone-safe lower bound), validity
        // and the column's Morton bits -- everything a neighbour cell needs is then a static pick
        double g2[DIM][3];
        bool okc[DIM][3];
        uint32_t sp[DIM][3];
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
            cl[a] = c0[a] >> l;
            const double sl = slack_h + slack_abs[a];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const int x = cl[a] + o - 1;
                okc[a][o] = x >= 0 && x < ncl;
                sp[a][o] = okc[a][o] ? (DIM == 3 ? spread3((uint32_t)x) : spread2((uint32_t)x)) << a : 0u;
                const double lo = gp.mn[a] + (double)x * hl, hi = gp.mn[a] + (double)(x + 1) * hl;
                double g = lo - s[a] - sl;
                double gh = s[a] - hi - sl;
                open_sides(gp, a, x, ncl, g, gh);
                g = gh > g ? gh : g;
                g = g > 0.0 ? g : 0.0;
                g2[a][o] = g * g;
            }
        }
        // phase A: the query's own cell (every lane has one: coherent)
        {
            uint32_t code = sp[0][1] | sp[1][1];
            if (DIM == 3) code |= sp[2][1];
            scan(start[(uint64_t)code << (DIM * l)], start[((uint64_t)code + 1) << (DIM * l)]);
        }
        // phase B: which neighbour cells can still hold a better (or tying) point?
        uint32_t mask = 0;
#pragma unroll
        for (int nb = 0; nb < NBR; ++nb) {
            if (nb == NBR / 2) continue;
            const int ox = nb % 3, oy = (nb / 3) % 3, oz = nb / 9;
            bool inside = okc[0][ox] && okc[1][oy];
            double bound = g2[0][ox] + g2[1][oy];
            if (DIM == 3) {
                inside = inside && okc[2][oz];
                bound = bound + g2[2][oz];
            }
            if (inside && !(LE ? (bound > thr) : (bound > ICP_PRUNE_AT))) mask |= 1u << nb;
        }
        // phase C: fetch the ranges of up to LIST surviving cells in one batch, then every lane walks
        // its own concatenated list.  Cells are taken nearest class first (faces, edges, corners) and
        // their bound is tested again when they are drawn and again when they are walked: the best
        // distance shrinks as cells are scanned, and the lane that holds its wave back is typically one
        // whose own cell gave a poor first bound.
        while (__builtin_amdgcn_ballot_w64(mask != 0) != 0) {
            uint32_t r0[LIST], r1[LIST];
            float fb[LIST];
#pragma unroll
            for (int k = 0; k < LIST; ++k) {
                bool have = mask != 0;
                uint32_t m = mask;
                if (DIM == 3) {
                    constexpr uint32_t FACE = (1u << 4) | (1u << 10) | (1u << 12) | (1u << 14) | (1u << 16) | (1u << 22);
                    constexpr uint32_t CORNER = (1u << 0) | (1u << 2) | (1u << 6) | (1u << 8) | (1u << 18) | (1u << 20) | (1u << 24) | (1u << 26);
                    m = (mask & FACE) ? (mask & FACE) : ((mask & ~CORNER) ? (mask & ~CORNER) : mask);
                } else {
                    constexpr uint32_t FACE = (1u << 1) | (1u << 3) | (1u << 5) | (1u << 7);
                    m = (mask & FACE) ? (mask & FACE) : mask;
                }
                const int nb = have ? __builtin_ctz(m) : 0;
                mask &= ~(have ? (1u << nb) : 0u);
                const int ox = nb % 3, oy = (nb / 3) % 3, oz = nb / 9;
                uint32_t code = (ox == 0 ? sp[0][0] : (ox == 1 ? sp[0][1] : sp[0][2])) |
                                (oy == 0 ? sp[1][0] : (oy == 1 ? sp[1][1] : sp[1][2]));
                double bound = (ox == 0 ? g2[0][0] : (ox == 1 ? g2[0][1] : g2[0][2])) +
                               (oy == 0 ? g2[1][0] : (oy == 1 ? g2[1][1] : g2[1][2]));
                if (DIM == 3) {
                    code |= (oz == 0 ? sp[2][0] : (oz == 1 ? sp[2][1] : sp[2][2]));
                    bound = bound + (oz == 0 ? g2[2][0] : (oz == 1 ? g2[2][1] : g2[2][2]));
                }
                if (!LE) have = have && !(bound > ICP_PRUNE_AT);  // best may have improved since the mask was built
                code = have ? code : 0u;  // exhausted mask: columns -1 may be outside the grid
                r0[k] = start[(uint64_t)code << (DIM * l)];
                r1[k] = have ? start[((uint64_t)code + 1) << (DIM * l)] : r0[k];
                fb[k] = __double2float_rd(bound);  // <= bound: skipping on (double)fb > best stays exact
            }
#pragma unroll
            for (int k = 0; k < LIST; ++k) {
                wl[k][lane_slot] = make_uint2(r0[k], r1[k]);
                wlb[k][lane_slot] = fb[k];
            }
            constexpr int UNR = UNRN;
            int cur = 0;
            uint32_t p = 0, pe = 0;
            while (true) {
                while (p >= pe && cur < LIST) {
                    const uint2 r = wl[cur][lane_slot];
                    const float f = wlb[cur][lane_slot];
                    ++cur;
                    if (!LE && (double)f > ICP_PRUNE_AT) continue;  // a tie (bound == best) is still visited
                    p = r.x;
                    pe = r.y;
                }
                const bool work = p < pe;
                if (__builtin_amdgcn_ballot_w64(work) == 0) break;
                if (work) {
                    static_assert(UNR - 1 <= kRecPad, "records read past the end of the array");
                    walk(p);
                    p += UNR;
                    dbg_cand += UNR;
                }
            }
        }
        if (ncl == 1) break;  // the neighbourhood was the whole grid
        // distance from the query to the boundary of the visited box [cl-1, cl+1]; sides that
        // coincide with the grid boundary have nothing beyond them
        double mgap = __builtin_inf();
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
            const double sl = slack_h + slack_abs[a];
            if (cl[a] - 1 > 0) {
                const double g = s[a] - (gp.mn[a] + (double)(cl[a] - 1) * hl) - sl;
                mgap = g < mgap ? g : mgap;
            }
            if (cl[a] + 2 < ncl) {
                const double g = (gp.mn[a] + (double)(cl[a] + 2) * hl) - s[a] - sl;
                mgap = g < mgap ? g : mgap;
            }
        }
        if (mgap == __builtin_inf()) break;  // box covers the grid
        if (mgap > 0.0) {
            const double m2 = mgap * mgap;
            if (LE ? (thr < m2) : (ICP_PRUNE_AT < m2)) break;
        }
    }
#undef ICP_PRUNE_AT
    if (BND) {
        lb2_io[q] = second;
#pragma unroll
        for (int a = 0; a < DIM; ++a) sref_io[q * DIM + a] = s[a];
    }
    if (dbg) {
        const long long dt = (long long)__builtin_amdgcn_s_memtime() - t_begin;
        dbg[t * 3] = dt;
        dbg[t * 3 + 1] = dbg_levels;
        dbg[t * 3 + 2] = dbg_cand;
    }
    if (bi == 0x7fffffff) {  // nothing comparable found (e.g. every d2 = +inf): np.argmin -> 0
        bi = 0;
        bp = inv[0];
    }
    idx_out[q] = pos_mode ? (int32_t)bp : bi;
    if (d2_out) d2_out[q] = best;
}

// The per-lane pass with the neighbour cells handled by the WAVE (argmin searches).  In
// grid_query_kernel every lane looks up and walks its own surviving neighbour cells, so a wave takes as long as its
// unluckiest lane -- the one with the most cells or records (2^20 x 2^20: 68 records for the longest lane of a wave
// against 21 on average) -- and five sixths of the kernel's vector instructions were that phase.  Here the surviving cells of
// all 64 queries go into ONE list per wave (a static pass over the 26 neighbours writes (cell code, owner lane) pairs at
// slots from a wave-wide scan); all ranges of the list are looked up at once, their lengths are scanned into the offsets of
// one concatenation of all the records the wave has to test, and every lane tests an equal share of it, consecutive records,
// against the OWNER's query, which it reads from LDS (cell units, fp32: the pre-filter above).
// A record the filter cannot reject is evaluated in fp64 by the lane that found it, with the owner's coordinates and
// eval()'s arithmetic, and merged into the owner's slot by LDS atomics in three steps: least d2 (as bits: d2 >= 0), least
// index among the records with that d2, position of that record -- the (d2, index) order of the full scan whatever the
// order of arrival (the candidates of a walk are queued and merged 64 at a time when it is over).  The owner then renews its
// threshold.  Lanes whose own query is finished, deferred or out of range keep serving the others.  Own cell, hint, levels, gap test and hand-over to the deferred pass are those of grid_query_kernel.
#ifndef ICP_POOL_CAP
#define ICP_POOL_CAP 384
#endif
constexpr int kPoolCap = ICP_POOL_CAP;   // list entries per wave and compaction pass (more: another pass; 256 / 384 / 512: 177 / 169 / 170 us)
constexpr int kPoolQ = 384;              // candidates queued per wave before they are evaluated (a trip adds at most 64 x 3)
#ifndef ICP_POOL_UNR
#define ICP_POOL_UNR 3                   // records per lane and trip of the wave-level walk (2 / 3 / 4 / 6: 169 / 169 / 169 / 185 us)
#endif

template <int DIM, bool FUSED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void grid_pool_kernel(
    const double *__restrict__ src, int64_t n, const int32_t *__restrict__ qorder, const double4 *__restrict__ recs,
    const uint32_t *__restrict__ start, GridParams gp, int32_t *idx_out, double *__restrict__ d2_out,
    const int32_t *__restrict__ done, int cap, Deferred *__restrict__ deferred, uint32_t *__restrict__ n_deferred,
    const int32_t *hint, const uint32_t *__restrict__ inv, uint32_t m, int pos_mode, int xcd_chunk,
    const double *__restrict__ Rm, const double *__restrict__ tv, double *__restrict__ err_part, uint32_t defer_max) {
    if (done && *done) return;
    constexpr int QB = 256, NW = QB / 64, UNR = 3, PUNR = ICP_POOL_UNR;
    int64_t blk = blockIdx.x;
    if (xcd_chunk > 0) {   // XCD-aware block order: see grid_query_kernel
        const uint32_t per = 8u * (uint32_t)xcd_chunk, sc = blockIdx.x / per, r = blockIdx.x - sc * per;
        blk = (int64_t)sc * per + (int64_t)(r & 7u) * xcd_chunk + (r >> 3);
    }
    const int64_t t = blk * QB + threadIdx.x;
    const bool in_range = t < n;     // every lane stays to the end: it serves the other queries of its wave
    const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t q = in_range ? (qorder ? (int64_t)qorder[t] : t) : 0;
    double s[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) s[a] = src[q * DIM + a];
    if (FUSED) {   // ss = src R + t, operation for operation as apply_err_kernel forms it
        double o[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            double v = s[0] * Rm[c];
#pragma unroll
            for (int k = 1; k < DIM; ++k) v += s[k] * Rm[k * DIM + c];
            o[c] = v + tv[c];
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c) s[c] = o[c];
    }
    int c0[DIM];
    cell_of<DIM>(s, gp, c0);
    double slack_abs[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) slack_abs[a] = (fabs(gp.mn[a]) + fabs(gp.mx[a])) * 0x1p-48;
    double best = __builtin_inf();
    int32_t bi = 0x7fffffff;
    uint32_t bp = 0;
    bool finite = true, isnan = false;
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
        finite = finite && fabs(s[a]) < __builtin_inf();
        isnan = isnan || s[a] != s[a];
    }
    if (FUSED) {   // error term of the previous iteration and the block's sum of them: see grid_query_kernel
        double e = 0.0;
        if (in_range) {
            const uint32_t h = (uint32_t)hint[q];
            const double4 rc = recs[h < m ? h : 0u];
            const double dx = rc.x - s[0], dy = rc.y - s[1];
            double d = dx * dx + dy * dy;
            if (DIM == 3) {
                const double dz = rc.z - s[2];
                d = d + dz * dz;
            }
            e = sqrt(d);
            // the hinted record is the first candidate as well (eval()'s arithmetic and test; the fused loop's hints are
            // sorted positions): no second load of it behind the barrier
            const int32_t id = (int32_t)__double_as_longlong(rc.w);
            if (h < m && (d < best || (d == best && id < bi))) {
                best = d;
                bi = id;
                bp = h;
            }
        }
        __shared__ double ered[NW];
        const double w = wave_sum(e);
        if (ln == 0) ered[wv] = w;
        __syncthreads();
        if (threadIdx.x == 0) {
            double v = ered[0];
#pragma unroll
            for (int k = 1; k < NW; ++k) v += ered[k];
            err_part[blk] = v;
        }
    }
    constexpr int NBR = DIM == 3 ? 27 : 9;
    __shared__ uint2 plist[NW][kPoolCap];              // (cell code, owner lane), then (first record, owner lane)
    __shared__ uint32_t ptab[NW][kPoolCap + 4];        // offset of the range's records in the wave's concatenation (+ sentinels)
    __shared__ float4 pq[NW][64];                      // owner's query in cell units and its threshold
    __shared__ uint32_t pskip[NW][64];                 // position of the owner's best record
    __shared__ double psrc[NW][DIM][64];               // owner's query
    __shared__ uint2 pqueue[NW][kPoolQ];               // candidates waiting for their fp64 evaluation: (owner lane, position)
    __shared__ unsigned long long pbest[NW][64];       // owner's best d2 (bits)
    __shared__ uint32_t pid[NW][64], ppos[NW][64];     // its index and position
#pragma unroll
    for (int a = 0; a < DIM; ++a) psrc[wv][a][ln] = s[a];
    lds_wave_sync();
    bool active = in_range && finite, deferred_q = false;
    if (in_range && !finite) {   // np.argmin of a constant NaN / inf row: index 0
        idx_out[q] = pos_mode ? (int32_t)inv[0] : 0;
        if (d2_out) d2_out[q] = isnan ? __builtin_nan("") : __builtin_inf();
    }
    // fp32 pre-filter (filter_record): the query in cell units, its threshold, the position of its best record
    const float4 *recs32 = filter_records(recs, (int64_t)m);
    float uq[3] = {0.0f, 0.0f, 0.0f};
    bool f_ok = gp.h > 1e-100 && gp.h < 1e100;
    {
        const double nc = (double)(1 << gp.bits);
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
            const double u = (s[a] - gp.mn[a]) * gp.inv_h;
            f_ok = f_ok && u >= kFilterLo * nc && u <= kFilterHi * nc;
            uq[a] = (float)u;
        }
    }
    const double f_eps = 3.4641016151377544 * 1.001 * (0x1p-24 * kFilterHi * (1.0 + 1e-7)) * (double)(1 << gp.bits);
    float f_thr = __builtin_inff();
    uint32_t f_skip = 0xffffffffu;
    auto refresh = [&]() {
        const double r = (__dsqrt_rn(best) * gp.inv_h + f_eps) * (1.0 + 1e-6);
        f_thr = f_ok ? __double2float_ru(r * r) : __builtin_inff();
        f_skip = bi == 0x7fffffff ? 0xffffffffu : bp;
    };
    auto eval = [&](const double4 &rc, uint32_t pos) {
        const double dx = rc.x - psrc[wv][0][ln], dy = rc.y - psrc[wv][1][ln];
        double d = dx * dx + dy * dy;
        if (DIM == 3) {
            const double dz = rc.z - psrc[wv][2][ln];
            d = d + dz * dz;
        }
        const int32_t id = (int32_t)__double_as_longlong(rc.w);
        if (d < best || (d == best && id < bi)) {
            best = d;
            bi = id;
            bp = pos;
        }
    };
    auto own_scan = [&](uint32_t p0, uint32_t p1) {   // the query's own cell: per lane, as in grid_query_kernel
        for (uint32_t p = p0; p < p1; p += UNR) {
            float4 rf[UNR];
            const float4 *b32 = recs32 + p;
#pragma unroll
            for (int u = 0; u < UNR; ++u) rf[u] = b32[u];
            bool pass[UNR], any = false;
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const float dx = rf[u].x - uq[0], dy = rf[u].y - uq[1];
                float d = __builtin_fmaf(dy, dy, dx * dx);
                if (DIM == 3) {
                    const float dz = rf[u].z - uq[2];
                    d = __builtin_fmaf(dz, dz, d);
                }
                pass[u] = !(d > f_thr) && p + (uint32_t)u != f_skip;
                any = any || pass[u];
            }
            if (any) {
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    if (pass[u]) eval(recs[p + u], p + u);
                refresh();
            }
        }
    };
    if (!FUSED && active && hint) {
        const uint32_t h = (uint32_t)hint[q];
        if (h < m) {
            const uint32_t hp = pos_mode ? h : inv[h];
            eval(recs[hp], hp);
        }
    }
    refresh();

    bool may_defer = true;
    for (int l = 0; l <= gp.bits; ++l) {
        if (__builtin_amdgcn_ballot_w64(active) == 0) break;
        if (active && l >= cap && may_defer) {   // hand-over to grid_defer_kernel: see grid_query_kernel
            const uint32_t slot = atomicAdd(n_deferred, 1u);
            may_defer = slot < defer_max;
            if (may_defer) {
                Deferred dq;
                dq.q = (int32_t)q;
                dq.level = l;
                dq.bi = bi;
                dq.bp = bp;
                dq.best = best;
#pragma unroll
                for (int a = 0; a < 3; ++a) dq.s[a] = a < DIM ? psrc[wv][a < DIM ? a : 0][ln] : 0.0;
                dq.second = __builtin_inf();
                deferred[slot] = dq;
                active = false;
                deferred_q = true;
            }
        }
        if (l >= cap && __builtin_amdgcn_ballot_w64(active) == 0) break;   // (everything left was handed over: no set-up for nobody)
        const double hl = gp.h * (double)(1 << l);
        const int ncl = 1 << (gp.bits - l);
        const double slack_h = hl * 0x1p-20;
        int cl[DIM];
        float g2[DIM][3];   // squared gaps as floats scaled by (1 - 2^-21): their rounded sum stays below the fp64 bound
        bool okc[DIM][3];
        uint32_t sp[DIM][3];
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
            cl[a] = c0[a] >> l;
            const double sl = slack_h + slack_abs[a];
            const double sa = psrc[wv][a][ln];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const int x = cl[a] + o - 1;
                okc[a][o] = x >= 0 && x < ncl;
                sp[a][o] = okc[a][o] ? (DIM == 3 ? spread3((uint32_t)x) : spread2((uint32_t)x)) << a : 0u;
                const double lo = gp.mn[a] + (double)x * hl, hi = gp.mn[a] + (double)(x + 1) * hl;
                double g = lo - sa - sl;
                double gh = sa - hi - sl;
                open_sides(gp, a, x, ncl, g, gh);
                g = gh > g ? gh : g;
                g = g > 0.0 ? g : 0.0;
                g2[a][o] = __double2float_rd(g * g) * (1.0f - 0x1p-21f);
            }
        }
        if (active) {   // phase A: the query's own cell, per lane (listed with the neighbours it gains 8 % where the hints are exact and nothing in
                        // the ICP loop; an unhinted query would list all 27 cells and queue every record of them: 0.58 against 0.13 ms)
            uint32_t code = sp[0][1] | sp[1][1];
            if (DIM == 3) code |= sp[2][1];
            own_scan(start[(uint64_t)code << (DIM * l)], start[((uint64_t)code + 1) << (DIM * l)]);
        }
        // phase B: the neighbour cells that can still hold a better (or tying) point
        uint32_t mask = 0;
        if (active) {
            const float bestf = __double2float_ru(best);
#pragma unroll
            for (int nb = 0; nb < NBR; ++nb) {
                if (nb == NBR / 2) continue;
                const int ox = nb % 3, oy = (nb / 3) % 3, oz = nb / 9;
                bool inside = okc[0][ox] && okc[1][oy];
                float bound = g2[0][ox] + g2[1][oy];
                if (DIM == 3) {
                    inside = inside && okc[2][oz];
                    bound = bound + g2[2][oz];
                }
                if (inside && !(bound > bestf)) mask |= 1u << nb;
            }
        }
        // what the other lanes need of this query
        unsigned long long mybits = (unsigned long long)__double_as_longlong(best);
        pq[wv][ln] = make_float4(uq[0], uq[1], uq[2], f_thr);
        pskip[wv][ln] = f_skip;
        pbest[wv][ln] = mybits;
        pid[wv][ln] = (uint32_t)bi;
        ppos[wv][ln] = bp;
        // fp64 evaluation of the queued candidates by whichever lane takes them (the owner's coordinates from LDS, eval()'s
        // arithmetic: the same bits), 64 per round; one that cannot reach its owner's present best is dropped, the others are
        // merged into the owners' slots: least d2, then least index among the records with that d2, then that record's position
        uint32_t qcount = 0;   // (wave-uniform)
        auto flush = [&]() {
            lds_wave_sync();
            for (uint32_t base = 0; base < qcount; base += 64) {
                const uint32_t qi = base + (uint32_t)ln;
                const bool have = qi < qcount;
                const uint2 qe = pqueue[wv][have ? qi : 0u];
                const int o = have ? (int)qe.x : 0;
                unsigned long long db = ~0ull;
                uint32_t id = 0x7fffffffu;
                bool cand = false;
                if (have) {
                    const double4 rc = recs[qe.y];
                    const double dx = rc.x - psrc[wv][0][o], dy = rc.y - psrc[wv][1][o];
                    double d = dx * dx + dy * dy;
                    if (DIM == 3) {
                        const double dz = rc.z - psrc[wv][2][o];
                        d = d + dz * dz;
                    }
                    id = (uint32_t)(int32_t)__double_as_longlong(rc.w);
                    db = (unsigned long long)__double_as_longlong(d);
                    cand = db <= pbest[wv][o];
                }
                if (__builtin_amdgcn_ballot_w64(cand) == 0) continue;
                if (cand) atomicMin(&pbest[wv][o], db);
                lds_wave_sync();
                const unsigned long long nbits = pbest[wv][ln];
                const bool chg = nbits != mybits;
                if (chg) {   // a better d2 for this lane's own query: its old index no longer counts
                    pid[wv][ln] = 0x7fffffffu;
                    mybits = nbits;
                }
                lds_wave_sync();
                const bool top = cand && db == pbest[wv][o];
                if (top) atomicMin(&pid[wv][o], id);
                lds_wave_sync();
                if (top && id == pid[wv][o]) ppos[wv][o] = qe.y;
                lds_wave_sync();
                const uint32_t nid = pid[wv][ln];
                if (chg || nid != (uint32_t)bi) {
                    best = __longlong_as_double((long long)mybits);
                    bi = (int32_t)nid;
                    bp = ppos[wv][ln];
                    refresh();
                    pq[wv][ln].w = f_thr;
                    pskip[wv][ln] = f_skip;
                }
                lds_wave_sync();
            }
            qcount = 0;
        };
        // phase C: the wave's list of (cell, owner) pairs, worked off 64 at a time
        while (true) {
            const uint32_t cnt = (uint32_t)__builtin_popcount(mask);
            const uint32_t inc = wave_scan_u32(cnt);
            const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            if (T == 0) break;
            lds_wave_sync();   // (the previous pass's list has been read)
            uint32_t slot = inc - cnt;
#pragma unroll
            for (int nb = 0; nb < NBR; ++nb) {
                const int ox = nb % 3, oy = (nb / 3) % 3, oz = nb / 9;
                if (mask & (1u << nb)) {
                    if (slot < (uint32_t)kPoolCap) {
                        uint32_t code = sp[0][ox] | sp[1][oy];
                        if (DIM == 3) code |= sp[2][oz];
                        plist[wv][slot] = make_uint2(code, (uint32_t)ln);
                        mask &= ~(1u << nb);
                    }
                    ++slot;
                }
            }
            lds_wave_sync();
            const uint32_t Tc = T < (uint32_t)kPoolCap ? T : (uint32_t)kPoolCap;
            // every range of the list looked up at once (entries [ln K', ln K' + K') by lane ln), their lengths scanned into
            // the offsets of ONE concatenation of all the records the wave has to test
            constexpr int KMAX = kPoolCap / 64;
            const uint32_t Kp = (Tc + 63u) / 64u;
            uint32_t er0[KMAX], elen[KMAX], eown[KMAX], lsum = 0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const uint32_t idx = (uint32_t)ln * Kp + (uint32_t)k;
                const bool have = (uint32_t)k < Kp && idx < Tc;
                const uint2 e = plist[wv][have ? idx : 0u];
                uint32_t a0 = 0, a1 = 0;
                if (have) {
                    a0 = start[(uint64_t)e.x << (DIM * l)];
                    a1 = start[((uint64_t)e.x + 1) << (DIM * l)];
                }
                er0[k] = a0;
                elen[k] = a1 - a0;
                eown[k] = e.y;
            }
#pragma unroll
            for (int k = 0; k < KMAX; ++k) lsum += elen[k];
            const uint32_t linc = wave_scan_u32(lsum);
            const uint32_t R = (uint32_t)__builtin_amdgcn_readlane((int)linc, 63);
            // the table keeps the NON-EMPTY ranges only (first record, owner; offset), so that PUNR consecutive records of the
            // concatenation lie in at most PUNR consecutive entries and a trip finds them without a loop
            uint32_t ne = 0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) ne += elen[k] ? 1u : 0u;
            const uint32_t ninc = wave_scan_u32(ne);
            const uint32_t Tn = (uint32_t)__builtin_amdgcn_readlane((int)ninc, 63);
            lds_wave_sync();   // (every lane has read its entries of the list: it is overwritten in place)
            {
                uint32_t pre = linc - lsum, slot = ninc - ne;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    if (elen[k]) {
                        plist[wv][slot] = make_uint2(er0[k], eown[k]);
                        ptab[wv][slot] = pre;
                        ++slot;
                    }
                    pre += elen[k];
                }
                if (ln < 4) ptab[wv][Tn + (uint32_t)ln] = R;   // sentinels: no record index reaches them
            }
            lds_wave_sync();
            if (R != 0) {
            // lane ln tests records [ln C, ln C + C) of the concatenation, PUNR consecutive ones per trip
            static_assert(PUNR == 3, "the trip's entry arithmetic below is written for three records");
            const uint32_t C = (R + 63u) / 64u;
            uint32_t i0 = (uint32_t)ln * C;
            const uint32_t iend = i0 + C < R ? i0 + C : R;
            uint32_t cur = 0;
            {
                uint32_t hi = Tn;   // last entry whose offset is <= i0
                while (hi - cur > 1) {
                    const uint32_t mid = (cur + hi) >> 1;
                    if (ptab[wv][mid] <= i0) cur = mid;
                    else hi = mid;
                }
            }
            uint32_t b0 = ptab[wv][cur];
            for (; __builtin_amdgcn_ballot_w64(i0 < iend) != 0; i0 += PUNR) {
                // the three records of this trip lie in entries cur .. cur + 2, the first of the next trip in cur .. cur + 3
                const uint32_t b1 = ptab[wv][cur + 1], b2 = ptab[wv][cur + 2], b3 = ptab[wv][cur + 3];
                bool pass[PUNR];
                uint32_t pos[PUNR], own[PUNR];
                uint2 ent[PUNR];
                uint32_t epre[PUNR];
#pragma unroll
                for (int u = 0; u < PUNR; ++u) {
                    const uint32_t ii = i0 + (uint32_t)u;
                    const uint32_t kk = (ii >= b1 ? 1u : 0u) + (ii >= b2 ? 1u : 0u);
                    ent[u] = plist[wv][cur + kk];
                    epre[u] = kk == 0u ? b0 : (kk == 1u ? b1 : b2);
                }
#pragma unroll
                for (int u = 0; u < PUNR; ++u) {
                    const uint32_t ii = i0 + (uint32_t)u;
                    pass[u] = ii < iend;
                    pos[u] = pass[u] ? ent[u].x + (ii - epre[u]) : 0u;
                    own[u] = pass[u] ? ent[u].y : 0u;
                }
                {
                    const uint32_t nx = i0 + (uint32_t)PUNR;
                    const uint32_t kk = (nx >= b1 ? 1u : 0u) + (nx >= b2 ? 1u : 0u) + (nx >= b3 ? 1u : 0u);
                    b0 = kk == 0u ? b0 : (kk == 1u ? b1 : (kk == 2u ? b2 : b3));
                    cur = cur + kk < Tn ? cur + kk : Tn - 1u;   // (a lane past its share: nothing of it is used)
                }
                float4 rf[PUNR], oq[PUNR];
                uint32_t osk[PUNR];
#pragma unroll
                for (int u = 0; u < PUNR; ++u) {
                    rf[u] = recs32[pos[u]];
                    oq[u] = pq[wv][own[u]];
                    osk[u] = pskip[wv][own[u]];
                }
#pragma unroll
                for (int u = 0; u < PUNR; ++u) {
                    const float dx = rf[u].x - oq[u].x, dy = rf[u].y - oq[u].y;
                    float d = __builtin_fmaf(dy, dy, dx * dx);
                    if (DIM == 3) {
                        const float dz = rf[u].z - oq[u].z;
                        d = __builtin_fmaf(dz, dz, d);
                    }
                    pass[u] = (int)pass[u] & (int)!(d > oq[u].w) & (int)(pos[u] != osk[u]);
                }
                // candidates the filter let through: queued (owner, position) -- slots from the ballots, no atomics -- and
                // evaluated 64 at a time when the walk is over (or the queue nearly full)
#pragma unroll
                for (int u = 0; u < PUNR; ++u) {
                    const unsigned long long bm = __builtin_amdgcn_ballot_w64(pass[u]);
                    if (bm == 0) continue;
                    if (pass[u]) pqueue[wv][qcount + (uint32_t)__popcll(bm & ((1ull << ln) - 1ull))] = make_uint2(own[u], pos[u]);
                    qcount += (uint32_t)__popcll(bm);
                }
                if (qcount > (uint32_t)(kPoolQ - 64 * PUNR)) flush();
            }
            }
            if (qcount) flush();
            if (T <= (uint32_t)kPoolCap) break;   // (every mask is empty: no second scan to find that out)
        }
        lds_wave_sync();   // (before the slots are rewritten at the next level)
        if (active) {
            if (ncl == 1) {
                active = false;  // the neighbourhood was the whole grid
            } else {
                double mgap = __builtin_inf();
#pragma unroll
                for (int a = 0; a < DIM; ++a) {
                    const double sl = slack_h + slack_abs[a];
                    const double sa = psrc[wv][a][ln];
                    if (cl[a] - 1 > 0) {
                        const double g = sa - (gp.mn[a] + (double)(cl[a] - 1) * hl) - sl;
                        mgap = g < mgap ? g : mgap;
                    }
                    if (cl[a] + 2 < ncl) {
                        const double g = (gp.mn[a] + (double)(cl[a] + 2) * hl) - sa - sl;
                        mgap = g < mgap ? g : mgap;
                    }
                }
                if (mgap == __builtin_inf()) active = false;  // box covers the grid
                else if (mgap > 0.0 && best < mgap * mgap) active = false;
            }
        }
    }
    if (!in_range || !finite || deferred_q) return;
    if (bi == 0x7fffffff) {  // nothing comparable found (e.g. every d2 = +inf): np.argmin -> 0
        bi = 0;
        bp = inv[0];
    }
    idx_out[q] = pos_mode ? (int32_t)bp : bi;
    if (d2_out) d2_out[q] = best;
}

// Second pass: HALF a wave per deferred query.  Same search (3^DIM neighbourhood per level, bound test, gap test), but
// the 27 range lookups are done by 27 lanes at once and the surviving ranges are scanned 32 records per step, so a
// query that needs many levels costs a few microseconds.  The pass is a chain of dependent latencies per level (ranges ->
// records -> reduction), so two queries share a wave (32-lane groups; everything a group exchanges goes through
// width-32 shuffles, no LDS and no barrier: the two halves of a wave may be at different levels or out of work):
// 2^20 x 2^20, 30 k deferred queries: 40 -> 34 us.
// cells on the deferred pass's stack: coordinates at their level (3-D: 9 bits each, 2-D: 14) and the level above them
template <int DIM>
__device__ __forceinline__ uint32_t pack_cell(const int *x, int lvl) {
    if (DIM == 3) return ((uint32_t)lvl << 27) | ((uint32_t)x[2] << 18) | ((uint32_t)x[1] << 9) | (uint32_t)x[0];
    return ((uint32_t)lvl << 28) | ((uint32_t)x[1] << 14) | (uint32_t)x[0];
}
template <int DIM>
__device__ __forceinline__ void unpack_cell(uint32_t v, int *x, int &lvl) {
    if (DIM == 3) {
        lvl = (int)(v >> 27);
        x[0] = (int)(v & 511u);
        x[1] = (int)((v >> 9) & 511u);
        x[2] = (int)((v >> 18) & 511u);
    } else {
        lvl = (int)(v >> 28);
        x[0] = (int)(v & 16383u);
        x[1] = (int)((v >> 14) & 16383u);
    }
}
#ifndef ICP_DEFER_G
#define ICP_DEFER_G 32
#endif
constexpr int kSplit = 32;    // a cell above the finest level with more records than this is replaced by its children
constexpr int kStack = 384;   // cells per group's stack (a cell whose children would not fit is scanned whole)

template <int DIM, bool LE, bool BND = false>
__global__ __launch_bounds__(64) void grid_defer_kernel(const double4 *__restrict__ recs,
                                                        const uint32_t *__restrict__ start, GridParams gp,
                                                        const double *__restrict__ thr_in, int32_t *__restrict__ idx_out,
                                                        double *__restrict__ d2_out, const int32_t *__restrict__ done,
                                                        const Deferred *__restrict__ deferred,
                                                        const uint32_t *__restrict__ n_deferred,
                                                        uint32_t *__restrict__ next_counter,
                                                        const uint32_t *__restrict__ inv, int pos_mode, int jump, int split_min,
                                                        double *__restrict__ lb2_io = nullptr, double *__restrict__ sref_io = nullptr,
                                                        uint32_t defer_max = 0xffffffffu,
                                                        const uint32_t *__restrict__ mode_counts = nullptr, uint32_t mode_nb = 0,
                                                        uint32_t mode_limit = 0, uint32_t *__restrict__ gate_out = nullptr) {
    // Coherent iterations: how many rows could this iteration's test pass not prove?  Beyond a share of them the eight-lane
    // pass costs more than searching the whole cloud (it spends ~10 us of latency per row where the per-lane pass overlaps
    // thousands: break-even near 16 % at 2^20 x 2^20), and the NEXT iteration takes the ordinary fused search instead
    // (*gate_out = 1; a lag of one iteration: the share changes slowly, and either form returns the same results).  The
    // last block sums the counts -- it has no query of its own unless the list is nearly full.
    if (mode_counts && blockIdx.x == gridDim.x - 1) {
        uint32_t v = 0;
        for (uint32_t b = threadIdx.x; b < mode_nb; b += 64) v += mode_counts[b];
        for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_down((int)v, o, 64);
        if (threadIdx.x == 0) *gate_out = v > mode_limit ? 1u : 0u;
    }
    // the two counters alternate between searches: nobody touches the other one during this search, so it is
    // cleared here for the next search (before the `done` test, so that it is clear whatever that says)
    if (blockIdx.x == 0 && threadIdx.x == 0) *next_counter = 0u;
    if (done && *done) return;
    constexpr int NBR = DIM == 3 ? 27 : 9;
    constexpr int G = ICP_DEFER_G;  // lanes per query
    constexpr uint32_t GMASK = (uint32_t)((1ull << G) - 1ull);
    __shared__ uint32_t stk[64 / G][kStack];
    const int lane = threadIdx.x & (G - 1), grp = threadIdx.x / G;
    const uint32_t nd = *n_deferred < defer_max ? *n_deferred : defer_max;   // (lanes that found the list full kept their queries)
    const uint32_t ngroups = gridDim.x * (64 / G);
    for (uint32_t it = blockIdx.x * (64 / G) + grp; it < nd; it += ngroups) {
        const Deferred dq = deferred[it];
        const int64_t q = dq.q;
        double s[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) s[a] = dq.s[a];
        int c0[DIM];
        cell_of<DIM>(s, gp, c0);
        double best = dq.best;     // group-uniform between levels
        int32_t bi = dq.bi;
        uint32_t bp = dq.bp;
        // BND (see grid_query_kernel): `sec` = least d2 among the evaluated records other than the best one, group-uniform
        // between rounds, and what cells and gaps are tested against instead of `best`
        double sec = BND ? dq.second : 0.0;
        const double thr = LE ? thr_in[q] : 0.0;
        int l0 = dq.level;
        if (!LE && jump && (BND ? sec : best) < __builtin_inf()) {
            // The bound the query arrives with (its hint: last iteration's match) already says how far the search has
            // to reach: the first level whose 3^DIM box passes the gap test with it.  The best distance only shrinks,
            // so that level's scan is the last one, and the levels below it -- subsets of its box -- add nothing:
            // start there instead of climbing (one chain of dependent latencies instead of one per level).
            for (; l0 < gp.bits; ++l0) {
                const double hl = gp.h * (double)(1 << l0);
                const int ncl = 1 << (gp.bits - l0);
                const double slack_h = hl * 0x1p-20;
                double mgap = __builtin_inf();
#pragma unroll
                for (int a = 0; a < DIM; ++a) {
                    const int ca = c0[a] >> l0;
                    const double sl = slack_h + (fabs(gp.mn[a]) + fabs(gp.mx[a])) * 0x1p-48;
                    if (ca - 1 > 0) {
                        const double g = s[a] - (gp.mn[a] + (double)(ca - 1) * hl) - sl;
                        mgap = g < mgap ? g : mgap;
                    }
                    if (ca + 2 < ncl) {
                        const double g = (gp.mn[a] + (double)(ca + 2) * hl) - s[a] - sl;
                        mgap = g < mgap ? g : mgap;
                    }
                }
                if (mgap == __builtin_inf()) break;                 // the box covers the grid
                if (mgap > 0.0 && (BND ? sec : best) < mgap * mgap) break;        // the same test as at the end of a level below
            }
        }
        for (int l = l0; l <= gp.bits; ++l) {
            const double hl = gp.h * (double)(1 << l);
            const int ncl = 1 << (gp.bits - l);
            const double slack_h = hl * 0x1p-20;
            int cl[DIM];
#pragma unroll
            for (int a = 0; a < DIM; ++a) cl[a] = c0[a] >> l;
            // The 3^DIM cells of the query's level-l neighbourhood seed a stack of cells (packed coordinates + level,
            // per group, in LDS).  Each round the group takes up to G cells off the top, one per lane: bound test, range
            // lookup; a cell that holds more than kSplit records and is not at the finest level is replaced by its 2^DIM
            // children (which are tested in their turn), the others are scanned together, G records per step.  A coarse
            // cell that merely touches the query's ball therefore costs a descent along the ball's surface instead of
            // a scan of everything in it (2^18-point Gaussian: a few rim queries used to scan tens of thousands of
            // records each and held the whole pass for 0.1 ms).
            int sp = 0;  // group-uniform
#pragma unroll
            for (int nb0 = 0; nb0 < NBR; nb0 += G) {
                const int nb = nb0 + lane;
                bool inside = nb < NBR;
                int x[DIM];
                const int off[3] = {nb % 3 - 1, (nb / 3) % 3 - 1, nb / 9 - 1};
#pragma unroll
                for (int a = 0; a < DIM; ++a) {
                    x[a] = cl[a] + off[a];
                    inside = inside && x[a] >= 0 && x[a] < ncl;
                }
                const uint32_t gm = (uint32_t)(__ballot(inside) >> (grp * G)) & GMASK;
                if (inside) stk[grp][sp + __popc(gm & ((1u << lane) - 1u))] = pack_cell<DIM>(x, l);
                sp += __popc(gm);
            }
            while (sp > 0) {
                lds_wave_sync();
                const int take = sp < G ? sp : G;
                sp -= take;
                uint32_t r0 = 0, r1 = 0;
                int lvl = 0, x[DIM];
#pragma unroll
                for (int a = 0; a < DIM; ++a) x[a] = 0;
                if (lane < take) {
                    unpack_cell<DIM>(stk[grp][sp + lane], x, lvl);
                    const double hv = gp.h * (double)(1 << lvl);
                    const int nv = 1 << (gp.bits - lvl);
                    const double slv = hv * 0x1p-20;
                    double bound = 0.0;
#pragma unroll
                    for (int a = 0; a < DIM; ++a) {
                        const double sl = slv + (fabs(gp.mn[a]) + fabs(gp.mx[a])) * 0x1p-48;
                        const double lo = gp.mn[a] + (double)x[a] * hv, hi = gp.mn[a] + (double)(x[a] + 1) * hv;
                        double g = lo - s[a] - sl;
                        double gh = s[a] - hi - sl;
                        open_sides(gp, a, x[a], nv, g, gh);
                        g = gh > g ? gh : g;
                        g = g > 0.0 ? g : 0.0;
                        bound = bound + g * g;   // same association as the per-lane pass: ((gx + gy) + gz)
                    }
                    if (!(LE ? (bound > thr) : (bound > (BND ? sec : best)))) {
                        const uint32_t code = morton<DIM>(x);
                        r0 = start[(uint64_t)code << (DIM * lvl)];
                        r1 = start[((uint64_t)code + 1) << (DIM * lvl)];
                    }
                }
                lds_wave_sync();   // every lane has read its cell before children are written over the same slots
                uint32_t len = r1 - r0;
                // split: children pushed in lane order; a cell whose children would not fit is scanned whole
                constexpr int NCH = 1 << DIM;
                bool split = lvl > 0 && len > (uint32_t)split_min;
                {
                    const uint32_t sm = (uint32_t)(__ballot(split) >> (grp * G)) & GMASK;
                    const int rank = __popc(sm & ((1u << lane) - 1u));
                    if (split && sp + (rank + 1) * NCH > kStack) split = false;
                }
                const uint32_t sm = (uint32_t)(__ballot(split) >> (grp * G)) & GMASK;
                if (split) {
                    const int at = sp + __popc(sm & ((1u << lane) - 1u)) * NCH;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        int y[DIM];
#pragma unroll
                        for (int a = 0; a < DIM; ++a) y[a] = 2 * x[a] + ((c >> a) & 1);
                        stk[grp][at + c] = pack_cell<DIM>(y, lvl - 1);
                    }
                    len = 0;
                }
                sp += __popc(sm) * NCH;
                // prefix of the range lengths over the group's lanes
                uint32_t incl = len;
#pragma unroll
                for (int o = 1; o < G; o <<= 1) {
                    const uint32_t v = __shfl_up(incl, o, G);
                    if (lane >= o) incl += v;
                }
                const uint32_t excl = incl - len;                      // records before this lane's cell
                const uint32_t total = __shfl(incl, G - 1, G);
                double lb = best;
                int32_t li = bi;
                uint32_t lp = bp;
                double ls = sec;   // BND: this lane's second best
                for (uint32_t base = 0; base < total; base += G) {      // group-uniform trip count: the shuffles need every lane
                    const uint32_t tpos = base + lane;
                    int c = 0;  // last lane with excl[c] <= tpos
#pragma unroll
                    for (int stp = G / 2; stp > 0; stp >>= 1) {
                        const uint32_t e = __shfl(excl, c + stp, G);    // (c + stp < G always: c <= G - 2 * stp before the step)
                        if (e <= tpos) c += stp;
                    }
                    const uint32_t rc0 = __shfl(r0, c, G), ex = __shfl(excl, c, G);
                    if (tpos < total) {
                        const uint32_t rpos = rc0 + (tpos - ex);
                        const double4 rc = recs[rpos];
                        const double dx = rc.x - s[0], dy = rc.y - s[1];
                        double d = dx * dx + dy * dy;
                        if (DIM == 3) {
                            const double dz = rc.z - s[2];
                            d = d + dz * dz;
                        }
                        const int32_t id = (int32_t)__double_as_longlong(rc.w);
                        if (BND) {
                            const bool better = d < lb || (d == lb && id < li);
                            ls = better ? lb : ((id != li && d < ls) ? d : ls);
                        }
                        if (LE) {
                            if (d <= thr && id < li) {
                                li = id;
                                lp = rpos;
                                lb = 0.0;
                            }
                        } else if (d < lb || (d == lb && id < li)) {
                            lb = d;
                            li = id;
                            lp = rpos;
                        }
                    }
                }
                if (total > 0) {
                    const double lb_own = lb;
                    const int32_t li_own = li;
                    // group argmin of (d2, index); the position rides along
#pragma unroll
                    for (int o = G / 2; o > 0; o >>= 1) {
                        const double ob = shfl_xor_f64(lb, o);
                        const int32_t oi = __shfl_xor(li, o, 64);
                        const uint32_t op = (uint32_t)__shfl_xor((int)lp, o, 64);
                        if (ob < lb || (ob == lb && oi < li)) {
                            lb = ob;
                            li = oi;
                            lp = op;
                        }
                    }
                    best = lb;
                    bi = li;
                    bp = lp;
                    if (BND) {
                        // the group's second best: a lane whose own best is not the group's contributes that best (its
                        // own second is no smaller), the others their second
                        double c = li_own == li ? ls : lb_own;
#pragma unroll
                        for (int o = G / 2; o > 0; o >>= 1) {
                            const double oc = shfl_xor_f64(c, o);
                            c = oc < c ? oc : c;
                        }
                        sec = c;
                    }
                }
            }
            if (ncl == 1) break;
            double mgap = __builtin_inf();
#pragma unroll
            for (int a = 0; a < DIM; ++a) {
                const double sl = slack_h + (fabs(gp.mn[a]) + fabs(gp.mx[a])) * 0x1p-48;
                if (cl[a] - 1 > 0) {
                    const double g = s[a] - (gp.mn[a] + (double)(cl[a] - 1) * hl) - sl;
                    mgap = g < mgap ? g : mgap;
                }
                if (cl[a] + 2 < ncl) {
                    const double g = (gp.mn[a] + (double)(cl[a] + 2) * hl) - s[a] - sl;
                    mgap = g < mgap ? g : mgap;
                }
            }
            if (mgap == __builtin_inf()) break;
            if (mgap > 0.0) {
                const double m2 = mgap * mgap;
                if (LE ? (thr < m2) : ((BND ? sec : best) < m2)) break;
            }
        }
        if (lane == 0) {
            if (bi == 0x7fffffff) {
                bi = 0;
                bp = inv[0];
            }
            idx_out[q] = pos_mode ? (int32_t)bp : bi;
            if (d2_out) d2_out[q] = best;
            if (BND) {
                lb2_io[q] = sec;
#pragma unroll
                for (int a = 0; a < DIM; ++a) sref_io[q * DIM + a] = s[a];
            }
        }
    }
}

// Coherent iterations: the queries the test pass could not prove (grid_query_kernel<COH = 1> left them as Deferred
// records in one list per block), EIGHT LANES PER QUERY.  Such a query arrives with its old match as the best so far and
// an upper bound sec0 on its second-best d2, which says at once how far the search must reach: the level whose 3^DIM box
// passes the gap test with sec0 (level 0 for 97 % of them at 2^20 x 2^20).  The generic deferred pass spends about twelve
// dependent memory round trips on such a query (27 cells in rounds of G, then their records in steps of G); here the
// group tests all 3^DIM cells at once (up to four per lane), issues every range lookup together, concatenates the
// surviving ranges through LDS and reads the records 4 per lane per step: record, ranges, records -- three round trips.
// Same candidates, same (d2, index) order, cells skipped only on bound > sec0 (>= the best so far), so the argmin is the
// full scan's; `second` ends as min(second-best d2 found, sec0), a lower bound on every other target's d2 whatever the
// quality of sec0.  Levels above 1 (sparse rims: coarse cells hold thousands of records and want the descent of
// grid_defer_kernel) are forwarded to that kernel's list.
template <int DIM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void coh_small_kernel(const double4 *__restrict__ recs, const uint32_t *__restrict__ start,
                                                       GridParams gp, int32_t *__restrict__ idx_out,
                                                       const int32_t *__restrict__ done, const Deferred *__restrict__ lists,
                                                       const uint32_t *__restrict__ counts, uint32_t nsub, uint32_t sub_cap,
                                                       Deferred *__restrict__ fwd, uint32_t *__restrict__ n_fwd,
                                                       double *__restrict__ lb2_io, double *__restrict__ sref_io,
                                                       const uint32_t *__restrict__ gate) {
    if (done && *done) return;
    if (*gate != 0u) return;   // too few rows were proven: the whole-cloud search takes this iteration
    constexpr int G = 8, NGW = 64 / G, NBR = DIM == 3 ? 27 : 9, CPL = (NBR + G - 1) / G, U = 3;
    __shared__ uint32_t cr0[NGW][NBR + 1], cpre[NGW][NBR + 1];
    const int lane = threadIdx.x & (G - 1), grp = threadIdx.x / G;
    const uint32_t NG = gridDim.x * NGW, gid = blockIdx.x * NGW + grp;
    const uint32_t gps = NG >= nsub ? NG / nsub : 1u;
    for (uint32_t sub = gid / gps; sub < nsub; sub += NG / gps) {
        // (the first record is fetched together with the list's length: the slot exists whatever the length says)
        uint32_t it = gid % gps;
        Deferred dq = lists[(size_t)sub * sub_cap + (it < sub_cap ? it : 0u)];
        const uint32_t nd = counts[sub];
        for (; it < nd; it += gps, dq = lists[(size_t)sub * sub_cap + (it < nd ? it : 0u)]) {
            double s[DIM];
#pragma unroll
            for (int a = 0; a < DIM; ++a) s[a] = dq.s[a];
            const double sec0 = dq.second;
            int c0[DIM];
            cell_of<DIM>(s, gp, c0);
            // the level the search has to reach (the deferred pass's jump, against sec0)
            int l = 0;
            bool found = false;
            for (; l <= 1 && l <= gp.bits; ++l) {
                const double hl = gp.h * (double)(1 << l);
                const int ncl = 1 << (gp.bits - l);
                const double slack_h = hl * 0x1p-20;
                double mgap = __builtin_inf();
#pragma unroll
                for (int a = 0; a < DIM; ++a) {
                    const int ca = c0[a] >> l;
                    const double sl = slack_h + (fabs(gp.mn[a]) + fabs(gp.mx[a])) * 0x1p-48;
                    if (ca - 1 > 0) {
                        const double g = s[a] - (gp.mn[a] + (double)(ca - 1) * hl) - sl;
                        mgap = g < mgap ? g : mgap;
                    }
                    if (ca + 2 < ncl) {
                        const double g = (gp.mn[a] + (double)(ca + 2) * hl) - s[a] - sl;
                        mgap = g < mgap ? g : mgap;
                    }
                }
                if (mgap == __builtin_inf() || (mgap > 0.0 && sec0 < mgap * mgap)) {   // box covers the grid | everything within sec0 is inside it
                    found = true;
                    break;
                }
            }
            if (!found) {   // (group-uniform) a wide ball, or no usable sec0: the generic pass
                if (lane == 0) fwd[atomicAdd(n_fwd, 1u)] = dq;
                continue;
            }
            const double hv = gp.h * (double)(1 << l);
            const int nv = 1 << (gp.bits - l);
            const double slv = hv * 0x1p-20;
            uint32_t r0[CPL], len[CPL];
            uint64_t cd[CPL];
            bool alive[CPL];
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int nb = lane + k * G;
                bool inside = nb < NBR;
                int x[DIM];
                const int off[3] = {nb % 3 - 1, (nb / 3) % 3 - 1, nb / 9 - 1};
                double bound = 0.0;
#pragma unroll
                for (int a = 0; a < DIM; ++a) {
                    x[a] = (c0[a] >> l) + off[a];
                    inside = inside && x[a] >= 0 && x[a] < nv;
                    const double sl = slv + (fabs(gp.mn[a]) + fabs(gp.mx[a])) * 0x1p-48;
                    const double lo = gp.mn[a] + (double)x[a] * hv, hi = gp.mn[a] + (double)(x[a] + 1) * hv;
                    double g = lo - s[a] - sl;
                    double gh = s[a] - hi - sl;
                    open_sides(gp, a, x[a], nv, g, gh);
                    g = gh > g ? gh : g;
                    g = g > 0.0 ? g : 0.0;
                    bound = bound + g * g;   // the per-lane pass's bound, same association
                }
                alive[k] = inside && !(bound > sec0);
                if (!inside)
#pragma unroll
                    for (int a = 0; a < DIM; ++a) x[a] = 0;
                cd[k] = (uint64_t)morton<DIM>(x);
            }
#pragma unroll
            for (int k = 0; k < CPL; ++k) {   // every lookup in flight before the first use
                r0[k] = alive[k] ? start[cd[k] << (DIM * l)] : 0u;
                len[k] = alive[k] ? start[(cd[k] + 1) << (DIM * l)] : 0u;
            }
            uint32_t mycells = 0, myrecs = 0;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                len[k] = alive[k] ? len[k] - r0[k] : 0u;
                mycells += len[k] ? 1u : 0u;
                myrecs += len[k];
            }
            uint32_t ic = mycells, ir = myrecs;   // inclusive scans over the group
#pragma unroll
            for (int o = 1; o < G; o <<= 1) {
                const uint32_t vc = (uint32_t)__shfl_up((int)ic, o, G), vr = (uint32_t)__shfl_up((int)ir, o, G);
                if (lane >= o) {
                    ic += vc;
                    ir += vr;
                }
            }
            const uint32_t nslot = (uint32_t)__shfl((int)ic, G - 1, G), total = (uint32_t)__shfl((int)ir, G - 1, G);
            lds_wave_sync();   // the previous query's table is dead
            {
                uint32_t slot = ic - mycells, pre = ir - myrecs;
#pragma unroll
                for (int k = 0; k < CPL; ++k)
                    if (len[k]) {
                        cr0[grp][slot] = r0[k];
                        cpre[grp][slot] = pre;
                        ++slot;
                        pre += len[k];
                    }
                if (lane == 0) cpre[grp][nslot] = total;
            }
            lds_wave_sync();
            double lb = dq.best, ls = sec0;
            int32_t li = dq.bi;
            uint32_t lp = dq.bp;
            for (uint32_t base = 0; base < total; base += G * U) {
                // a lane takes U CONSECUTIVE records of the concatenation (one 128-byte line when they lie in one cell): one
                // search of the table for the first, a walk along it for the others
                uint32_t rp[U];
                bool ok[U];
                const uint32_t i0 = base + (uint32_t)(lane * U);
                uint32_t lo = 0, hi = nslot;   // last slot with cpre <= i0
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (cpre[grp][mid] <= i0) lo = mid;
                    else hi = mid;
                }
                uint32_t cbeg = cpre[grp][lo], cend = cpre[grp][lo + 1], cr = cr0[grp][lo];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t i = i0 + (uint32_t)u;
                    ok[u] = i < total;
                    while (ok[u] && i >= cend) {   // (slots hold at least one record: at most U - 1 steps in all)
                        ++lo;
                        cbeg = cend;
                        cend = cpre[grp][lo + 1];
                        cr = cr0[grp][lo];
                    }
                    rp[u] = ok[u] ? cr + (i - cbeg) : 0u;
                }
                double4 rc[U];
#pragma unroll
                for (int u = 0; u < U; ++u) rc[u] = recs[rp[u]];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!ok[u]) continue;
                    const double dx = rc[u].x - s[0], dy = rc[u].y - s[1];
                    double d = dx * dx + dy * dy;
                    if (DIM == 3) {
                        const double dz = rc[u].z - s[2];
                        d = d + dz * dz;
                    }
                    const int32_t id = (int32_t)__double_as_longlong(rc[u].w);
                    const bool better = d < lb || (d == lb && id < li);
                    ls = better ? lb : ((id != li && d < ls) ? d : ls);
                    if (better) {
                        lb = d;
                        li = id;
                        lp = rp[u];
                    }
                }
            }
            const double lb_own = lb;
            const int32_t li_own = li;
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) {
                const double ob = shfl_xor_f64(lb, o);
                const int32_t oi = __shfl_xor(li, o, 64);
                const uint32_t op = (uint32_t)__shfl_xor((int)lp, o, 64);
                if (ob < lb || (ob == lb && oi < li)) {
                    lb = ob;
                    li = oi;
                    lp = op;
                }
            }
            double c = li_own == li ? ls : lb_own;   // (see grid_defer_kernel)
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) {
                const double oc = shfl_xor_f64(c, o);
                c = oc < c ? oc : c;
            }
            if (lane == 0) {
                const int64_t q = dq.q;
                idx_out[q] = (int32_t)lp;   // coherent loop: sorted positions
                lb2_io[q] = c;
#pragma unroll
                for (int a = 0; a < DIM; ++a) sref_io[q * DIM + a] = s[a];
            }
        }
    }
}

// ------------------------------------------------------------------ exact k nearest through the grid
// KNN of nicp_3d.py:53-57 / nicp3d.cpp:100-137: the k smallest keys fl(sqrt(d2)), ties to the lower
// index.  One query per lane, a sorted list of K (key, index) pairs in registers.  Level by level the
// 3^DIM neighbourhood of the query's cell is scanned from scratch (a restart, so no record is ever
// inserted twice); a cell is skipped when fl(sqrt(bound)) > k-th key and the search ends when the
// k-th key is strictly below fl(sqrt(gap^2)): sqrt and its rounding are monotone, so every unvisited
// record has a key >= that, and an equal key would still have to be looked at for its index.
constexpr int kKnnSplit = 64;   // records: a cell above the finest level with more is descended into, not scanned
constexpr int kKnnStack = 64;   // cells on a lane's stack

template <int DIM, int K>
__global__ __launch_bounds__(GB) void grid_knn_kernel(const double *__restrict__ qpts, int64_t nq,
                                                     const double4 *__restrict__ recs, const uint32_t *__restrict__ start,
                                                     GridParams gp, int32_t *__restrict__ idx_out, double *__restrict__ dist_out,
                                                     int defer_from, int32_t *__restrict__ dlist, uint32_t *__restrict__ ndef) {
    const int64_t t = (int64_t)blockIdx.x * GB + threadIdx.x;
    if (t >= nq) return;
    double s[DIM];
    bool finite = true;
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
        s[a] = qpts[t * DIM + a];
        finite = finite && fabs(s[a]) < __builtin_inf();
    }
    double key[K];
    int32_t id[K];
    auto reset = [&]() {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            key[k] = __builtin_inf();
            id[k] = 0x7fffffff;
        }
    };
    reset();
    if (finite) {
        int c0[DIM];
        cell_of<DIM>(s, gp, c0);
        constexpr int NBR = DIM == 3 ? 27 : 9;
        for (int l = 0; l <= gp.bits; ++l) {
            if (l >= defer_from) {
                // A query whose k nearest are not within the 3^DIM boxes of the first levels (a stray vertex far from the
                // mesh, a hole) would climb on alone while its wave waits: it goes to the wave-per-query pass instead
                // (grid_knn_defer_kernel), which starts again from the root.
                dlist[atomicAdd(ndef, 1u)] = (int32_t)t;
                return;
            }
            reset();
            double gate = __builtin_inf();
            const double hl = gp.h * (double)(1 << l);
            const int ncl = 1 << (gp.bits - l);
            const double slack_h = hl * 0x1p-20;
            int cl[DIM];
#pragma unroll
            for (int a = 0; a < DIM; ++a) cl[a] = c0[a] >> l;
            // lower bound of the squared distance to cell y of level lv (the arithmetic of the searches above)
            auto cell_bound = [&](const int *y, int lv) {
                const double hv = gp.h * (double)(1 << lv);
                const int nv = 1 << (gp.bits - lv);
                const double slv = hv * 0x1p-20;
                double bound = 0.0;
#pragma unroll
                for (int a = 0; a < DIM; ++a) {
                    const double sl = slv + (fabs(gp.mn[a]) + fabs(gp.mx[a])) * 0x1p-48;
                    const double lo = gp.mn[a] + (double)y[a] * hv, hi = gp.mn[a] + (double)(y[a] + 1) * hv;
                    double g = lo - s[a] - sl;
                    double gh = s[a] - hi - sl;
                    open_sides(gp, a, y[a], nv, g, gh);
                    g = gh > g ? gh : g;
                    g = g > 0.0 ? g : 0.0;
                    bound = bound + g * g;
                }
                return bound;
            };
            auto consider = [&](const double4 &rc) {
                {
                    double d2 = 0.0;
                    {
                        const double dx = rc.x - s[0], dy = rc.y - s[1];
                        d2 += dx * dx;
                        d2 += dy * dy;
                        if (DIM == 3) {
                            const double dz = rc.z - s[DIM - 1];
                            d2 += dz * dz;
                        }
                    }
                    if (!(d2 <= gate)) return;
                    const double r = __dsqrt_rn(d2);
                    const int32_t jj = (int32_t)__double_as_longlong(rc.w);
                    if (!(r < key[K - 1] || (r == key[K - 1] && jj < id[K - 1]))) return;
                    double ck = r;
                    int32_t ci = jj;
                    bool shifting = false;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const bool sw = shifting || ck < key[k] || (ck == key[k] && ci < id[k]);
                        shifting = sw;
                        const double tk = sw ? key[k] : ck;
                        const int32_t ti = sw ? id[k] : ci;
                        key[k] = sw ? ck : key[k];
                        id[k] = sw ? ci : id[k];
                        ck = tk;
                        ci = ti;
                    }
                    const double kr = key[K - 1];
                    // gate >= every d2 whose rounded root is <= kr.  Records arrive in no particular index
                    // order here, so an EQUAL root must get through too (it may carry a lower index): the
                    // class of d2 values that round to kr ends below kr^2 (1 + 2^-52), hence the margin.
                    gate = (kr > 1e-150 && kr < 1e150) ? kr * kr * (1.0 + 0x1p-50) : __builtin_inf();
                }
            };
            // records of a range four at a time: the loads of a group are in flight together (one lane walking a long
            // range paid a full memory latency per record); the index is clamped, not the group read past the range's
            // end as in the searches above -- a record of the next cell would be met again there and entered twice
            auto scan_range = [&](uint32_t p0, uint32_t p1) {
                for (uint32_t p = p0; p < p1; p += 4) {
                    double4 rc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) rc[u] = recs[p + u < p1 ? p + u : p1 - 1];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (p + u < p1) consider(rc[u]);
                }
            };
            for (int step = 0; step < NBR; ++step) {
                // own cell first: it fills the list and makes the bound test bite
                const int nb = step == 0 ? NBR / 2 : (step <= NBR / 2 ? step - 1 : step);
                const int off[3] = {nb % 3 - 1, (nb / 3) % 3 - 1, nb / 9 - 1};
                bool inside = true;
                int x[DIM];
#pragma unroll
                for (int a = 0; a < DIM; ++a) {
                    x[a] = cl[a] + off[a];
                    inside = inside && x[a] >= 0 && x[a] < ncl;
                }
                if (!inside) continue;
                if (__dsqrt_rn(cell_bound(x, l)) > key[K - 1]) continue;
                const uint32_t code = morton<DIM>(x);
                const uint32_t p0 = start[(uint64_t)code << (DIM * l)], p1 = start[((uint64_t)code + 1) << (DIM * l)];
                if (l == 0 || p1 - p0 <= (uint32_t)kKnnSplit) {
                    scan_range(p0, p1);
                    continue;
                }
                // A coarse cell with many records (a stray vertex far from the mesh climbs to levels whose cells hold
                // most of it: 10 strays in 200 000 vertices took the whole launch from 2 to 140 ms): descend instead of
                // scanning it whole -- a stack of cells, children pushed farthest first so that the nearest is looked at
                // first and the k-th key prunes the rest.  A cell whose children would not fit is scanned whole.
                constexpr int NCH = 1 << DIM;
                uint32_t stk[kKnnStack];
                int sp = 0;
                stk[sp++] = pack_cell<DIM>(x, l);
                while (sp > 0) {
                    int y[DIM], lv;
                    unpack_cell<DIM>(stk[--sp], y, lv);
                    if (__dsqrt_rn(cell_bound(y, lv)) > key[K - 1]) continue;
                    const uint32_t cc = morton<DIM>(y);
                    const uint32_t q0 = start[(uint64_t)cc << (DIM * lv)], q1 = start[((uint64_t)cc + 1) << (DIM * lv)];
                    if (q1 == q0) continue;
                    if (lv == 0 || q1 - q0 <= (uint32_t)kKnnSplit || sp + NCH > kKnnStack) {
                        scan_range(q0, q1);
                        continue;
                    }
                    double cb[NCH];
                    uint32_t cn[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        int z[DIM];
#pragma unroll
                        for (int a = 0; a < DIM; ++a) z[a] = 2 * y[a] + ((c >> a) & 1);
                        cb[c] = cell_bound(z, lv - 1);
                        cn[c] = pack_cell<DIM>(z, lv - 1);
                    }
                    for (int i = 0; i < NCH; ++i) {   // farthest first
                        int mx = -1;
                        for (int c = 0; c < NCH; ++c)
                            if (cn[c] != 0xffffffffu && (mx < 0 || cb[c] > cb[mx])) mx = c;
                        stk[sp++] = cn[mx];
                        cn[mx] = 0xffffffffu;
                    }
                }
            }
            if (ncl == 1) break;
            double mgap = __builtin_inf();
#pragma unroll
            for (int a = 0; a < DIM; ++a) {
                const double sl = slack_h + (fabs(gp.mn[a]) + fabs(gp.mx[a])) * 0x1p-48;
                if (cl[a] - 1 > 0) {
                    const double g = s[a] - (gp.mn[a] + (double)(cl[a] - 1) * hl) - sl;
                    mgap = g < mgap ? g : mgap;
                }
                if (cl[a] + 2 < ncl) {
                    const double g = (gp.mn[a] + (double)(cl[a] + 2) * hl) - s[a] - sl;
                    mgap = g < mgap ? g : mgap;
                }
            }
            if (mgap == __builtin_inf()) break;
            if (mgap > 0.0 && key[K - 1] < __dsqrt_rn(mgap * mgap)) break;
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        idx_out[t * K + k] = id[k] == 0x7fffffff ? -1 : id[k];
        dist_out[t * K + k] = key[k];
    }
}


// Second pass of the grid KNN: ONE WAVE PER QUERY for the queries the per-lane pass handed over.  A descent from the
// root cell over a stack of cells in LDS.  Each round takes up to 64 cells off the top, ONE PER LANE (a single cell while
// the list of k is not full yet: plain depth-first to the nearest leaf, so the k-th key exists before anything wide
// happens): bound test -- a cell is dropped when fl(sqrt(bound)) > k-th key; a tie must be visited, it may carry a
// lower index -- and range lookup run in parallel, so a round costs the two dependent memory latencies once, not once
// per cell (one wave walking cell by cell took 2.5 ms for a vertex 50 sigma from a 200 000-point cloud: the ball of
// its k-th key cuts through hundreds of sparsely filled cells).  A cell above the finest level with more than
// kKnnWaveScan records is replaced by its 2^DIM children (farthest first within the cell, so the nearest ends up on
// top), the others' ranges are concatenated and scanned by all 64 lanes, 64 records per step, each record re-tested
// against its cell's bound as the k-th key tightens.  Children partition their parent, so no record is met twice.
// The k best (key, index) pairs are held identically by every lane; after a step the lanes whose record can enter are
// taken in (key, index) order by wave-wide minima.  Same keys, same tie rule, same bounds as the per-lane pass: the
// result is the same list bit for bit (tests/test_gpu_nicp.py::test_knn_far_strays_and_duplicates).
constexpr int kKnnWaveScan = 128;
constexpr int kKnnWaveStack = 1024;

template <int DIM, int K>
__global__ __launch_bounds__(64) void grid_knn_defer_kernel(const double *__restrict__ qpts, const double4 *__restrict__ recs,
                                                           const uint32_t *__restrict__ start, GridParams gp,
                                                           const int32_t *__restrict__ dlist, const uint32_t *__restrict__ ndef,
                                                           int32_t *__restrict__ idx_out, double *__restrict__ dist_out) {
    __shared__ uint32_t stk[kKnnWaveStack];
    __shared__ uint32_t pre[65], rq0[64];
    __shared__ double rbound[64];
    constexpr int NCH = 1 << DIM;
    const int lane = threadIdx.x;
    const uint32_t nd = *ndef;
    for (uint32_t qi = blockIdx.x; qi < nd; qi += gridDim.x) {
        const int64_t t = dlist[qi];
        double s[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) s[a] = qpts[t * DIM + a];
        double key[K];
        int32_t id[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            key[k] = __builtin_inf();
            id[k] = 0x7fffffff;
        }
        double gate = __builtin_inf();
        auto cell_bound = [&](const int *y, int lv) {   // the per-lane pass's bound, operation for operation
            const double hv = gp.h * (double)(1 << lv);
            const int nv = 1 << (gp.bits - lv);
            const double slv = hv * 0x1p-20;
            double bound = 0.0;
#pragma unroll
            for (int a = 0; a < DIM; ++a) {
                const double sl = slv + (fabs(gp.mn[a]) + fabs(gp.mx[a])) * 0x1p-48;
                const double lo = gp.mn[a] + (double)y[a] * hv, hi = gp.mn[a] + (double)(y[a] + 1) * hv;
                double g = lo - s[a] - sl;
                double gh = s[a] - hi - sl;
                open_sides(gp, a, y[a], nv, g, gh);
                g = gh > g ? gh : g;
                g = g > 0.0 ? g : 0.0;
                bound = bound + g * g;
            }
            return bound;
        };
        __syncthreads();   // (block = one wave) the previous query's stack is dead
        if (lane == 0) {
            int root[DIM];
#pragma unroll
            for (int a = 0; a < DIM; ++a) root[a] = 0;
            stk[0] = pack_cell<DIM>(root, gp.bits);
        }
        int sp = 1;   // uniform
        __syncthreads();
        while (sp > 0) {
            const int take = key[K - 1] == __builtin_inf() ? 1 : (sp < 64 ? sp : 64);
            const int base = sp - take;
            // lane i < take: cell base + take - 1 - i (lane 0 holds the top of the stack = the nearest)
            int y[DIM], lv = 0;
            bool alive = lane < take;
            double bnd = 0.0;
            uint32_t q0 = 0, cnt = 0;
            if (alive) {
                unpack_cell<DIM>(stk[base + take - 1 - lane], y, lv);
                bnd = __dsqrt_rn(cell_bound(y, lv));
                alive = !(bnd > key[K - 1]);
            }
            if (alive) {
                const uint32_t cc = morton<DIM>(y);
                q0 = start[(uint64_t)cc << (DIM * lv)];
                cnt = start[((uint64_t)cc + 1) << (DIM * lv)] - q0;
                alive = cnt > 0;
            }
            bool split = alive && lv > 0 && cnt > (uint32_t)kKnnWaveScan;
            const unsigned long long sm = __ballot(split);
            const int srank = __popcll(sm & ((1ull << lane) - 1ull));
            if (split && base + (srank + 1) * NCH > kKnnWaveStack) split = false;   // no room for its children: scanned whole
            const int nsplit = __popcll(__ballot(split));
            __syncthreads();   // every lane has read its entry of the stack
            if (split) {
                // children farthest first: the last split lane's nearest child ends on top
                double cb[NCH];
                uint32_t cn[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    int z[DIM];
#pragma unroll
                    for (int a = 0; a < DIM; ++a) z[a] = 2 * y[a] + ((c >> a) & 1);
                    cb[c] = cell_bound(z, lv - 1);
                    cn[c] = pack_cell<DIM>(z, lv - 1);
                }
                // lane 0 is the nearest cell of the batch: its children must come last (top), so split lane r writes block nsplit-1-r
                uint32_t *dst = stk + base + (nsplit - 1 - srank) * NCH;
                for (int i = 0; i < NCH; ++i) {
                    int mx = -1;
                    for (int c = 0; c < NCH; ++c)
                        if (cn[c] != 0xffffffffu && (mx < 0 || cb[c] > cb[mx])) mx = c;
                    dst[i] = cn[mx];
                    cn[mx] = 0xffffffffu;
                }
            }
            sp = base + nsplit * NCH;
            // the other live cells: concatenate their ranges and scan them with all lanes
            const bool scan = alive && !split;
            uint32_t incl = scan ? cnt : 0u;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
                if (lane >= o) incl += up;
            }
            const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
            pre[lane] = incl - (scan ? cnt : 0u);
            rq0[lane] = q0;
            rbound[lane] = bnd;
            if (lane == 63) pre[64] = total;
            __syncthreads();
            for (uint32_t off = 0; off < total; off += 64) {
                const uint32_t i = off + (uint32_t)lane;
                bool cand = false;
                double r = __builtin_inf();
                int32_t jj = 0x7fffffff;
                if (i < total) {
                    int lo = 0, hi = 63;
#pragma unroll
                    for (int it = 0; it < 6; ++it) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (pre[mid] <= i) lo = mid;
                        else hi = mid - 1;
                    }
                    if (!(rbound[lo] > key[K - 1])) {   // the cell may have fallen behind the k-th key since it was listed
                        const double4 rc = recs[rq0[lo] + (i - pre[lo])];
                        double d2 = 0.0;
                        const double dx = rc.x - s[0], dy = rc.y - s[1];
                        d2 += dx * dx;
                        d2 += dy * dy;
                        if (DIM == 3) {
                            const double dz = rc.z - s[DIM - 1];
                            d2 += dz * dz;
                        }
                        if (d2 <= gate) {
                            r = __dsqrt_rn(d2);
                            jj = (int32_t)__double_as_longlong(rc.w);
                            cand = r < key[K - 1] || (r == key[K - 1] && jj < id[K - 1]);
                        }
                    }
                }
                while (__ballot(cand)) {   // uniform
                    double mr = cand ? r : __builtin_inf();
                    int32_t mi = cand ? jj : 0x7fffffff;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const double orr = shfl_xor_f64(mr, o);
                        const int32_t oi = __shfl_xor(mi, o, 64);
                        if (orr < mr || (orr == mr && oi < mi)) {
                            mr = orr;
                            mi = oi;
                        }
                    }
                    // (mr, mi) is the best pending candidate, and it beats the k-th entry: insert it (every lane alike)
                    double ck = mr;
                    int32_t ci = mi;
                    bool shifting = false;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const bool sw = shifting || ck < key[k] || (ck == key[k] && ci < id[k]);
                        shifting = sw;
                        const double tk = sw ? key[k] : ck;
                        const int32_t ti = sw ? id[k] : ci;
                        key[k] = sw ? ck : key[k];
                        id[k] = sw ? ci : id[k];
                        ck = tk;
                        ci = ti;
                    }
                    if (cand && jj == mi) cand = false;   // indices are unique: that lane's record is in
                    cand = cand && (r < key[K - 1] || (r == key[K - 1] && jj < id[K - 1]));
                }
                const double kr = key[K - 1];
                gate = (kr > 1e-150 && kr < 1e150) ? kr * kr * (1.0 + 0x1p-50) : __builtin_inf();
            }
            __syncthreads();   // pre / rq0 / rbound are rewritten by the next round
        }
        if (lane < K) {
            // every lane holds the list; lane k writes entry k
            double kk = key[0];
            int32_t ii = id[0];
#pragma unroll
            for (int k = 1; k < K; ++k)
                if (lane == k) {
                    kk = key[k];
                    ii = id[k];
                }
            idx_out[t * K + lane] = ii == 0x7fffffff ? -1 : ii;
            dist_out[t * K + lane] = kk;
        }
    }
}

int grid_knn(icp_ctx *ctx, const GridBuild &gb, const double *q, int64_t nq, int k, int32_t *idx_out, double *dist_out) {
    const unsigned nb = (unsigned)((nq + GB - 1) / GB);
    hipStream_t st = ctx->stream;
    const double4 *recs = (const double4 *)gb.recs;
    const uint32_t *start = (const uint32_t *)gb.start;
    // queries still open after the levels below `defer_from` go to the wave-per-query pass (ICP_KNN_DEFER: that level;
    // a value above every level keeps the per-lane pass alone, round 2's behaviour)
    static const int defer_env = getenv("ICP_KNN_DEFER") ? atoi(getenv("ICP_KNN_DEFER")) : 2;
    const int defer_from = defer_env > 0 ? defer_env : 1000;
    // the list lives in the scratch of the closest-point search's deferred pass (never in use at the same time); its
    // counter is word 2 of the header, the search's own two counters are left alone
    const size_t cap0 = ctx->defer.cap;
    if (int rc = ensure(ctx, ctx->defer, (size_t)nq * sizeof(int32_t) + 256)) return rc;
    if (ctx->defer.cap != cap0) ctx->defer_cleared = nullptr;
    uint32_t *ndef = (uint32_t *)ctx->defer.ptr + 2;
    int32_t *dlist = (int32_t *)((char *)ctx->defer.ptr + 256);
    ICP_HIP_CHECK(hipMemsetAsync(ndef, 0, sizeof(uint32_t), st));
    const unsigned nb2 = (unsigned)(nq < 4096 ? nq : 4096);
#define ICP_KNN(D, KK) \
    do { \
        hipLaunchKernelGGL((grid_knn_kernel<D, KK>), dim3(nb), dim3(GB), 0, st, q, nq, recs, start, gb.gp, idx_out, dist_out, defer_from, dlist, ndef); \
        if (defer_from < 1000) \
            hipLaunchKernelGGL((grid_knn_defer_kernel<D, KK>), dim3(nb2), dim3(64), 0, st, q, recs, start, gb.gp, (const int32_t *)dlist, \
                               (const uint32_t *)ndef, idx_out, dist_out); \
    } while (0)
    if (gb.gp.dim == 3 && k == 6) ICP_KNN(3, 6);
    else if (gb.gp.dim == 3) ICP_KNN(3, 8);
    else if (k == 6) ICP_KNN(2, 6);
    else ICP_KNN(2, 8);
#undef ICP_KNN
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

// ------------------------------------------------------------------ host side
static int env_bits() {
    const char *e = getenv("ICP_GRID_BITS");
    return e ? atoi(e) : 0;
}

// Builds the grid for tgt[m,dim] in ctx scratch (grid_a..grid_d).  One host sync (bbox).
// slot 1: the Morton ORDER of a point set is what the caller wants (sorted source): reproducible, stable sort.
// slot 0: a search structure over a target: order inside a cell free (see scatter_atomic_kernel).
int grid_build(icp_ctx *ctx, const double *tgt, int64_t m, int dim, GridBuild &gb, int slot) {
    if (slot) return grid_build_into(ctx, tgt, m, dim, gb, ctx->sort_a, ctx->sort_b, ctx->sort_c, ctx->sort_d, true);
    return grid_build_into(ctx, tgt, m, dim, gb, ctx->grid_a, ctx->grid_b, ctx->grid_c, ctx->grid_d, false);
}

static double grid_clip() {
    static const double clip = getenv("ICP_GRID_CLIP") ? atof(getenv("ICP_GRID_CLIP")) : 0.3;
    return clip;
}
static double grid_occ_target() {
    static const double occ_target = getenv("ICP_GRID_OCC") ? atof(getenv("ICP_GRID_OCC")) : 5.0;
    return occ_target;
}

int grid_build_into(icp_ctx *ctx, const double *tgt, int64_t m, int dim, GridBuild &gb, Scratch &sa, Scratch &sb,
                    Scratch &sc, Scratch &sd, bool stable) {
    GridPending pend;
    if (int rc = grid_build_begin(ctx, tgt, m, dim, gb, sa, sb, sc, sd, stable, pend)) return rc;
    return grid_build_finish(ctx, pend, gb);
}

// First half of a build: the statistics kernels (bounding box; for a target also the sample quantiles and cell
// populations) and their copy to pinned host memory, all asynchronous on ctx->stream.  grid_build_finish() waits for
// that copy and enqueues the rest.  Split so that icp_rigid_run can start the statistics of the target and of the
// source on two streams before it waits for either (build_grids in kabsch.hip); a sorted source (stable) and a target
// use disjoint scratch for this half.
int grid_build_begin(icp_ctx *ctx, const double *tgt, int64_t m, int dim, GridBuild &gb, Scratch &sa, Scratch &sb,
                     Scratch &sc, Scratch &sd, bool stable, GridPending &pend) {
    hipStream_t st = ctx->stream;
    gb.usable = false;
    if (int rc = ensure(ctx, ctx->partials, sizeof(double) * kMaxPartials * kSumSlots)) return rc;
    if (int rc = ensure(ctx, ctx->small, kSmallBytes)) return rc;
    if (!ctx->pinned) ICP_HIP_CHECK(hipHostMalloc(&ctx->pinned, 4096, hipHostMallocDefault));
    static_assert(sizeof(BBox) <= 2048 && 768 * 8 + sizeof(BBox) <= kSmallBytes && 1024 * 8 * 2 <= kMaxPartials * kSumSlots,
                  "scratch of the two concurrent builds overlaps");
    double *part = (double *)ctx->partials.ptr + (stable ? 1024 * 8 : 0);
    BBox *dbox = (BBox *)((double *)ctx->small.ptr + (stable ? 768 : 512));
    int nb = (int)((m + GB - 1) / GB);
    if (nb > 1024) nb = 1024;
    ICP_HIP_CHECK(hipGetLastError());
    // Robust extents.  A uniform grid over the bounding box degenerates when a few points lie far out (one outlier at
    // 1e6 puts a whole unit-size cloud into ONE cell: the search is then a full scan per query) and wastes resolution on
    // thin tails (a Gaussian's bounding box is +-5 sigma, 99.9 % of it lies within +-3.3).  So the grid spans, per axis,
    // the 1 % .. 99 % order statistics of a fixed-stride sample of 4096 points widened by `clip` of their distance on each side
    // (and never more than the bounding box); points beyond are clamped into the border columns, which the searches
    // treat as unbounded outwards (open_sides).  Exactness does not depend on where the extents are.
    const double clip = grid_clip();
    const int ns = (int)std::min<int64_t>(kQuantSamples, m);
    const bool robust = clip >= 0.0 && ns >= 64 && !stable;  // (a sorted SOURCE only needs an order: any extents do)
    // the sample of the statistics kernels, compact and per axis, behind the partial rows of both builds and of the means
    static_assert(1024 * 8 * 2 + 3 * kQuantSamples <= kMaxPartials * kSumSlots, "no room for the sample");
    double *samp = robust ? (double *)ctx->partials.ptr + 1024 * 8 * 2 : nullptr;
    if (samp && nb < kQuantSamples / GB) nb = kQuantSamples / GB;   // (the sample is gathered by the first 16 blocks)
    if (dim == 3) hipLaunchKernelGGL(bbox_kernel<3>, dim3(nb), dim3(GB), 0, st, tgt, m, part, m / ns, ns, samp);
    else hipLaunchKernelGGL(bbox_kernel<2>, dim3(nb), dim3(GB), 0, st, tgt, m, part, m / ns, ns, samp);
    hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(448), 0, st, (const double *)part, nb, dbox);
    if (robust)  // (same stream, after bbox_final_kernel: it adds its six numbers to the same record)
        hipLaunchKernelGGL(quantile_kernel, dim3(dim, kQuantSamples / kQuantCand), dim3(256), 0, st, (const double *)samp, ns, std::max(1, ns / 100), dbox);
    const bool adaptive = robust && env_bits() <= 0 && grid_occ_target() > 0.0;
    if (adaptive) hipLaunchKernelGGL(occupancy_kernel, dim3(kQuantSamples / 64, 16), dim3(256), 0, st, (const double *)samp, ns, dim, clip, dbox);
    ICP_HIP_CHECK(hipGetLastError());
    void *host = (char *)ctx->pinned + (stable ? 2048 : 0);
    ICP_HIP_CHECK(hipMemcpyAsync(host, dbox, sizeof(BBox), hipMemcpyDeviceToHost, st));
    hipEvent_t &ev = ctx->stat_ev[stable ? 1 : 0];
    if (!ev) ICP_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    ICP_HIP_CHECK(hipEventRecord(ev, st));   // (the caller may queue unrelated work behind the copy: build_grids in kabsch.hip)
    pend.ev = ev;
    pend.pts = tgt;
    pend.m = m;
    pend.dim = dim;
    pend.stable = stable;
    pend.robust = robust;
    pend.adaptive = adaptive;
    pend.ns = ns;
    pend.host = host;
    pend.sa = &sa;
    pend.sb = &sb;
    pend.sc = &sc;
    pend.sd = &sd;
    return ICP_OK;
}

// Second half: waits for the statistics (ctx->stream must be the stream grid_build_begin ran on), fixes the grid's
// extents and resolution on the host and enqueues the counting sort.
int grid_build_finish(icp_ctx *ctx, const GridPending &pend, GridBuild &gb) {
    hipStream_t st = ctx->stream;
    const double *tgt = pend.pts;
    const int64_t m = pend.m;
    const int dim = pend.dim, ns = pend.ns;
    const bool stable = pend.stable, robust = pend.robust, adaptive = pend.adaptive;
    Scratch &sa = *pend.sa, &sb = *pend.sb, &sc = *pend.sc, &sd = *pend.sd;
    const double clip = grid_clip(), occ_target = grid_occ_target();
    ICP_HIP_CHECK(hipEventSynchronize(pend.ev));
    BBox hb;
    std::memcpy(&hb, pend.host, sizeof(BBox));
    if (hb.bad) return ICP_OK;  // NaN / inf in the target: caller uses the brute-force scan
    int open = 0;
    for (int a = 0; robust && a < dim; ++a) {
        const double wdt = hb.qhi[a] - hb.qlo[a];
        if (!(wdt > 0.0) || !(wdt < __builtin_inf())) continue;  // (nearly) constant or non-finite sample: keep the bounding box
        const double lo = hb.qlo[a] - clip * wdt, hi = hb.qhi[a] + clip * wdt;
        if (lo > hb.mn[a]) {
            hb.mn[a] = lo;
            open |= 1 << (2 * a);
        }
        if (hi < hb.mx[a]) {
            hb.mx[a] = hi;
            open |= 1 << (2 * a + 1);
        }
    }
    double ext = 0.0;
    for (int a = 0; a < dim; ++a) {
        const double e = hb.mx[a] - hb.mn[a];
        if (!(e < __builtin_inf())) return ICP_OK;  // extent overflows
        ext = e > ext ? e : ext;
    }
    if (!(ext > 0.0)) ext = 1.0;  // all points identical
    // finest level: about one cell per target point, capped by the start-array size
    int bits = env_bits();
    if (bits <= 0) {
        bits = 1;
        const int maxbits = dim == 3 ? 8 : 12;
        while (bits < maxbits && (double)(1ull << (dim * (bits + 1))) <= 4.0 * (double)m) ++bits;
    }
    if (adaptive) {
        // the level whose estimated density-weighted cell population, 1 + N * pairs / (S choose 2), is closest to the
        // target (5: measured best on volume-filling, Gaussian and surface clouds of 2^18 .. 2^22 points)
        const int B = dim == 3 ? kOccB3 : kOccB2, maxbits = dim == 3 ? 8 : 12;
        const double denom = 0.5 * (double)ns * (double)(ns - 1);
        double best_d = __builtin_inf();
        for (int b = 1; b <= maxbits; ++b) {
            // unordered pairs sharing a cell at hierarchy level B - b = (ordered pairs whose first shared level is <= that,
            // minus the ns self-pairs, which share every level) / 2
            double ordered = 0.0;
            for (int l = 0; l <= B - b; ++l)
                for (int u = 0; u < kPairSub; ++u) ordered += (double)hb.pairs_u[l][u];
            const double occ = 1.0 + (double)m * (0.5 * (ordered - (double)ns)) / denom;
            const double d = std::fabs(std::log(occ / occ_target));
            if (d <= best_d) {   // ties: the finer level
                best_d = d;
                bits = b;
            }
        }
    }
    if (bits > (dim == 3 ? 9 : 14)) bits = dim == 3 ? 9 : 14;
    if (getenv("ICP_GRID_TRACE"))
        std::fprintf(stderr, "grid_build: m %lld dim %d bits %d (%s) open %d ext %.6g\n", (long long)m, dim, bits,
                     adaptive ? "from the sample's cell populations" : "size rule / ICP_GRID_BITS", open, ext);
    GridParams gp;
    for (int a = 0; a < 3; ++a) {
        gp.mn[a] = a < dim ? hb.mn[a] : 0.0;
        gp.mx[a] = a < dim ? hb.mx[a] : 0.0;
    }
    gp.bits = bits;
    gp.dim = dim;
    gp.open = open;
    gp.h = ext / (double)(1 << bits);
    gp.h *= 1.0 + 0x1p-30;  // the far face of the last cell lies beyond the largest coordinate
    gp.inv_h = 1.0 / gp.h;
    const int64_t ncell = 1ll << (dim * bits);
    const int64_t nsb = (ncell + (int64_t)GB * SCAN_ITEMS - 1) / ((int64_t)GB * SCAN_ITEMS);
    if (stable) {
        // A build that only wants the ORDER of its points (a sorted source, icp_morton_sort): codes -> stable sort ->
        // gather.  No cell counters (a million device-wide atomics: 53 us), no scan, no records: set-up of a rigid run
        // 0.54 -> see DESIGN.md 3.1b.  start / recs / inv stay null: such a build is never searched.
        if (int rc = ensure(ctx, sa, (size_t)m * dim * sizeof(double) + 64)) return rc;
        if (int rc = ensure(ctx, sb, (size_t)m * sizeof(int32_t) * 4)) return rc;     // sorted idx | codes | sorted codes | 0..m-1
        gb.spts = (double *)sa.ptr;
        gb.sidx = (int32_t *)sb.ptr;
        gb.recs = nullptr;
        gb.start = nullptr;
        gb.inv = nullptr;
        uint32_t *oc = (uint32_t *)sb.ptr + m;
        const unsigned ob = (unsigned)((m + GB - 1) / GB);
        if (dim == 3) hipLaunchKernelGGL(cell_code_kernel<3>, dim3(ob), dim3(GB), 0, st, tgt, m, gp, oc);
        else hipLaunchKernelGGL(cell_code_kernel<2>, dim3(ob), dim3(GB), 0, st, tgt, m, gp, oc);
        if (int rc = stable_sort_codes(ctx, ctx->sort_tmp, oc, oc + m, (int32_t *)(oc + 2 * m), gb.sidx, m, dim * bits)) return rc;
        if (dim == 3) hipLaunchKernelGGL(gather_order_kernel<3>, dim3(ob), dim3(GB), 0, st, tgt, m, gb.spts, (const int32_t *)gb.sidx);
        else hipLaunchKernelGGL(gather_order_kernel<2>, dim3(ob), dim3(GB), 0, st, tgt, m, gb.spts, (const int32_t *)gb.sidx);
        ICP_HIP_CHECK(hipGetLastError());
        gb.gp = gp;
        gb.m = m;
        gb.usable = true;
        return ICP_OK;
    }
    if (int rc = ensure(ctx, sa, (size_t)m * (dim + 4) * sizeof(double) + 64 + kRecPad * 32 + (size_t)(m + kRecPad) * 16)) return rc;   // sorted points | 32-byte records (+ pad) | 16-byte filter records (+ pad)
    if (int rc = ensure(ctx, sb, (size_t)m * sizeof(int32_t) * 4)) return rc;     // sorted idx | codes | sorted codes | 0..m-1
    if (int rc = ensure(ctx, sc, (size_t)(ncell + 64) * sizeof(uint32_t) * 2)) return rc;  // start | count/fill
    if (int rc = ensure(ctx, sd, (size_t)(nsb + 1) * sizeof(uint32_t))) return rc;
    gb.spts = (double *)sa.ptr;
    gb.recs = (void *)(((uintptr_t)((double *)sa.ptr + (size_t)m * dim) + 31) & ~(uintptr_t)31);
    gb.sidx = (int32_t *)sb.ptr;
    uint32_t *codes = (uint32_t *)sb.ptr + m;
    gb.start = (uint32_t *)sc.ptr;
    uint32_t *count = gb.start + (ncell + 64);   // (256-byte aligned: the memset below is then one fill kernel, not three)
    uint32_t *bsum = (uint32_t *)sd.ptr;
    const unsigned mb = (unsigned)((m + GB - 1) / GB);
    static const bool force_stable = getenv("ICP_GRID_STABLE") && atoi(getenv("ICP_GRID_STABLE")) != 0;
    const bool sorted_fill = force_stable || m > ((int64_t)1 << 22);
    uint32_t *rank = sorted_fill ? nullptr : codes + m;   // (the sort's scratch, unused by the unordered fill)
    ICP_HIP_CHECK(hipMemsetAsync(count, 0, (size_t)ncell * sizeof(uint32_t), st));
    if (dim == 3) hipLaunchKernelGGL(cell_count_kernel<3>, dim3(mb), dim3(GB), 0, st, tgt, m, gp, codes, count, rank);
    else hipLaunchKernelGGL(cell_count_kernel<2>, dim3(mb), dim3(GB), 0, st, tgt, m, gp, codes, count, rank);
    hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nsb), dim3(GB), 0, st, (const uint32_t *)count, gb.start, bsum, ncell);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, bsum, (int)nsb);
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nsb), dim3(GB), 0, st, gb.start, (const uint32_t *)bsum, ncell, (uint32_t)m);
    // (beyond a few million points the scattered 56-byte writes of the unordered fill cost more than sort + gather:
    //  2^24 points 1.76 ms against 0.25 + 0.83; 2^20 points 68 us against 160 + 43)
    if (sorted_fill) {
        // (a sorted source and a large target are built on two streams at once by icp_rigid_run: separate sort storage)
        if (int rc = stable_sort_codes(ctx, ctx->sort_tmp_tgt, codes, codes + m, (int32_t *)(codes + 2 * m),
                                       gb.sidx, m, dim * bits))
            return rc;
        if (dim == 3) hipLaunchKernelGGL(scatter_kernel<3>, dim3(mb), dim3(GB), 0, st, tgt, codes, m, gb.spts, (const int32_t *)gb.sidx, (double4 *)gb.recs, gp);
        else hipLaunchKernelGGL(scatter_kernel<2>, dim3(mb), dim3(GB), 0, st, tgt, codes, m, gb.spts, (const int32_t *)gb.sidx, (double4 *)gb.recs, gp);
    } else {
        if (dim == 3) hipLaunchKernelGGL(scatter_atomic_kernel<3>, dim3(mb), dim3(GB), 0, st, tgt, codes, (const uint32_t *)rank, (const uint32_t *)gb.start, m, gb.spts, gb.sidx, (double4 *)gb.recs, gp);
        else hipLaunchKernelGGL(scatter_atomic_kernel<2>, dim3(mb), dim3(GB), 0, st, tgt, codes, (const uint32_t *)rank, (const uint32_t *)gb.start, m, gb.spts, gb.sidx, (double4 *)gb.recs, gp);
    }
    ICP_HIP_CHECK(hipGetLastError());
    gb.gp = gp;
    gb.inv = codes;
    gb.m = m;
    gb.usable = true;
    return ICP_OK;
}

long long *g_grid_dbg = nullptr;    // per-lane instrumentation (ICP_GRID_DEBUG in icp_bench_grid_query)
hipEvent_t *g_grid_events = nullptr;  // [3]: before the per-lane pass, between the passes, after the deferred pass

// levels the per-lane pass handles itself (ICP_GRID_CAP=0: all of them, single pass)
static int grid_cap() {
    static const int cap_env = getenv("ICP_GRID_CAP") ? atoi(getenv("ICP_GRID_CAP")) : 1;
    return g_grid_dbg ? 1000 : (cap_env > 0 ? cap_env : 1000);
}
// The fused rigid loop needs the two-pass search: under the single-pass switches (ICP_GRID_CAP=0, ICP_GRID_DEBUG) the
// callers keep the moved source as an array instead (the ICP_NO_FUSE form), they do not fail.
bool grid_fusable() { return grid_cap() < 1000; }

int grid_query(icp_ctx *ctx, const GridBuild &gb, const double *src, int64_t n, const int32_t *qorder, bool le,
               const double *thr, int32_t *idx_out, double *d2_out, const int32_t *done, const int32_t *hint, bool pos_mode,
               const FusedApply *fa, const Coherence *coh) {
    hipStream_t st = ctx->stream;
    const int cap = grid_cap();
    if (coh && (le || qorder || d2_out || thr || cap >= 1000 || !pos_mode)) {
        set_error("grid_query: bounds come with the two-pass argmin search over sorted positions");
        return ICP_ERR_ARG;
    }
    double *clb = coh ? coh->lb2 : nullptr, *csr = coh ? coh->sref : nullptr;
    const size_t defer_cap = ctx->defer.cap;
    // The list holds a quarter of the queries (1.4 % are deferred at 2^20 x 2^20, a few tens of per cent on sheets and rims; a
    // lane that finds it full keeps its query): 16 bytes of scratch per query instead of 64.  Coherent iterations: one region
    // of 256 records per block of the test pass (padded block count) and, behind them, a full-size list for the forwarded
    // queries and the whole-cloud search.
    const size_t defer_slots = coh ? (size_t)n : (size_t)n / 4 + 4096;
    const uint32_t defer_max = (uint32_t)std::min<size_t>(defer_slots, 0xfffffff0u);
    if (int rc = ensure(ctx, ctx->defer, (defer_slots + (coh ? (size_t)n + 16 * 256 : 0)) * sizeof(Deferred) + 256)) return rc;
    if (ctx->defer.cap != defer_cap) ctx->defer_cleared = nullptr;   // reallocated (possibly at the same address)
    // two counters in the header, used alternately: the deferred pass of one search clears the counter of the
    // next, which saves a 4 us memset launch per search (1 % of an iteration at 2^20 points)
    uint32_t *counters = (uint32_t *)ctx->defer.ptr;
    Deferred *dlist = (Deferred *)((char *)ctx->defer.ptr + 256);
    if (ctx->defer_cleared != counters || cap >= 1000) {
        ICP_HIP_CHECK(hipMemsetAsync(counters, 0, 2 * sizeof(uint32_t), st));
        ctx->defer_cleared = counters;
        ctx->defer_parity = 0;
    }
    uint32_t *ndef = counters + ctx->defer_parity, *ndef_next = counters + (ctx->defer_parity ^ 1);
    if (cap < 1000) ctx->defer_parity ^= 1;
    static const int xcd_env = getenv("ICP_GRID_XCD") ? atoi(getenv("ICP_GRID_XCD")) : 1;
    unsigned nb = (unsigned)((n + 255) / 256);
    // xcd_chunk = consecutive blocks (of consecutive Morton-ordered queries) that one XCD takes; 8 * xcd_chunk blocks
    // form a super-chunk.  While an eighth of the 32-byte records is of the order of one L2 (<= 1.5 M targets) one
    // super-chunk spans the whole search: XCD k works on the k-th eighth of the queries (2^20 x 2^20: -6 % time, HBM
    // traffic 210 -> 109 MB).  Beyond that size that form LOSES (2^21: +4 %, 2^24: +3 %: the XCDs drift apart and stop
    // sharing the Infinity Cache) and short runs of 32 blocks win instead: the blocks that are resident in one XCD at
    // the same time are then neighbours in query order and share their neighbourhood's records in L2, while all XCDs
    // stay in one region of the target (2^21: -5 %, 2^22: -4 %, 2^24: -5 %; 8 / 16 / 32 within 1 % of each other; at
    // 2^20 runs of 32 are as fast as the eighths but fetch 126 MB).  ICP_GRID_XCD=0: plain order; N > 1: runs of N.
    int xcd_chunk = 0;
    if (xcd_env && nb >= 64) {
        const unsigned eighth = (nb + 7) / 8;
        unsigned c = gb.m * 32 <= (int64_t)48 << 20 ? eighth : 32u;
        if (xcd_env > 1) c = (unsigned)xcd_env;
        xcd_chunk = (int)(c < eighth ? c : eighth);
        const unsigned per = 8u * (unsigned)xcd_chunk;
        nb = (nb + per - 1) / per * per;            // blocks past the end return at once
    }
    unsigned nb2 = n < 8192 ? (unsigned)n : 8192u;   // one wave per deferred query, at most
    const double *fR = nullptr, *ft = nullptr;
    if (fa) {   // fused rigid loop (FusedApply in common.h): needs positions as hints, the plain argmin and both passes
        if (le || !hint || !pos_mode || qorder || d2_out || thr || cap >= 1000) {
            set_error("grid_query: the fused form needs hints as sorted positions and the two-pass argmin search");
            return ICP_ERR_ARG;
        }
        if (int rc = ensure(ctx, ctx->err_part, (size_t)nb * sizeof(double))) return rc;
        fR = fa->R;
        ft = fa->t;
        *fa->err_part = (double *)ctx->err_part.ptr;
        *fa->n_err = (int)nb;
    }
#define ICP_GQ3(D, L, F, LN, UN, B) \
    do { \
        if (g_grid_events) (void)hipEventRecord(g_grid_events[0], st); \
        hipLaunchKernelGGL((grid_query_kernel<D, L, 256, 4, F, LN, UN, B>), dim3(nb), dim3(256), 0, st, src, n, qorder, (const double4 *)gb.recs, \
                           (const uint32_t *)gb.start, gb.gp, thr, idx_out, d2_out, done, g_grid_dbg, cap, dlist, ndef, hint, \
                           (const uint32_t *)gb.inv, (uint32_t)gb.m, pos_mode ? 1 : 0, xcd_chunk, fR, ft, \
                           fa ? (double *)ctx->err_part.ptr : (double *)nullptr, clb, csr, (uint32_t *)nullptr, (const uint32_t *)nullptr, defer_max); \
        if (g_grid_events) (void)hipEventRecord(g_grid_events[1], st); \
        if (cap < 1000) \
            hipLaunchKernelGGL((grid_defer_kernel<D, L, B>), dim3(nb2), dim3(64), 0, st, (const double4 *)gb.recs, \
                               (const uint32_t *)gb.start, gb.gp, thr, idx_out, d2_out, done, (const Deferred *)dlist, \
                               (const uint32_t *)ndef, ndef_next, (const uint32_t *)gb.inv, pos_mode ? 1 : 0, \
                               jump_env >= 0 ? jump_env : ((hint && pos_mode) ? 1 : 0), split_env, clb, csr, defer_max); \
        if (g_grid_events) (void)hipEventRecord(g_grid_events[2], st); \
    } while (0)
#define ICP_GQ2(D, L, F, LN, UN) ICP_GQ3(D, L, F, LN, UN, false)
    // a coherent iteration: test pass over all queries (error sum; proof, or a record for the deferred pass) -> deferred pass
#define ICP_GQC(D) \
    do { \
        if (g_grid_events) (void)hipEventRecord(g_grid_events[0], st); \
        hipLaunchKernelGGL((grid_query_kernel<D, false, 256, 8, true, 4, 3, true, 1>), dim3(nb), dim3(256), 0, st, src, n, qorder, (const double4 *)gb.recs, \
                           (const uint32_t *)gb.start, gb.gp, thr, idx_out, d2_out, done, (long long *)nullptr, cap, dlist, ndef, hint, \
                           (const uint32_t *)gb.inv, (uint32_t)gb.m, 1, xcd_chunk, fR, ft, (double *)ctx->err_part.ptr, clb, csr, coh->count); \
        if (coh_trace) { \
            std::vector<uint32_t> hc(nb); \
            uint32_t tot = 0; \
            (void)hipMemcpyAsync(hc.data(), coh->count, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, st); \
            (void)hipStreamSynchronize(st); \
            for (unsigned y = 0; y < nb; ++y) tot += hc[y]; \
            std::fprintf(stderr, "coherent iteration: %u of %lld queries searched\n", tot, (long long)n); \
        } \
        if (g_grid_events) (void)hipEventRecord(g_grid_events[1], st); \
        /* few unproven rows: eight lanes each (the block lists occupy the deferred list's storage from slot 0, forwarded    \
           records go behind them); many: the ordinary fused search of every row.  One of the two returns at once. */ \
        hipLaunchKernelGGL((coh_small_kernel<D>), dim3(coh_blocks), dim3(64), 0, st, (const double4 *)gb.recs, (const uint32_t *)gb.start, \
                           gb.gp, idx_out, done, (const Deferred *)dlist, (const uint32_t *)coh->count, (uint32_t)nb, 256u, \
                           dlist + (size_t)nb * 256, ndef, clb, csr, (const uint32_t *)coh->gate); \
        hipLaunchKernelGGL((grid_query_kernel<D, false, 256, 4, true, 4, 3, true, 0>), dim3(nb), dim3(256), 0, st, src, n, qorder, (const double4 *)gb.recs, \
                           (const uint32_t *)gb.start, gb.gp, thr, idx_out, d2_out, done, (long long *)nullptr, cap, dlist + (size_t)nb * 256, ndef, hint, \
                           (const uint32_t *)gb.inv, (uint32_t)gb.m, 1, xcd_chunk, fR, ft, (double *)ctx->err_part.ptr, clb, csr, (uint32_t *)nullptr, \
                           (const uint32_t *)coh->gate); \
        hipLaunchKernelGGL((grid_defer_kernel<D, false, true>), dim3(nb2), dim3(64), 0, st, (const double4 *)gb.recs, \
                           (const uint32_t *)gb.start, gb.gp, thr, idx_out, d2_out, done, (const Deferred *)(dlist + (size_t)nb * 256), \
                           (const uint32_t *)ndef, ndef_next, (const uint32_t *)gb.inv, 1, \
                           jump_env >= 0 ? jump_env : 1, split_env, clb, csr, 0xffffffffu, (const uint32_t *)coh->count, (uint32_t)nb, coh_limit, coh->gate); \
        if (g_grid_events) (void)hipEventRecord(g_grid_events[2], st); \
    } while (0)
    static const int list8 = getenv("ICP_GRID_LIST8") ? atoi(getenv("ICP_GRID_LIST8")) : 0;
    // default: inside the rigid loops, whose hints are last iteration's matches (positions); a caller's ICP_HINT_IDX
    // may be anything, and a poor bound would make the jump scan far more than the climb
    static const int jump_env = getenv("ICP_GRID_JUMP") ? atoi(getenv("ICP_GRID_JUMP")) : -1;
    static const int split_env = getenv("ICP_GRID_SPLIT") ? atoi(getenv("ICP_GRID_SPLIT")) : kSplit;
#define ICP_GQL(D, L, F) \
    do { \
        if (list8) ICP_GQ2(D, L, F, 8, 4); \
        else ICP_GQ2(D, L, F, 4, 3); \
    } while (0)
    // argmin searches: the neighbour cells worked off by the wave (grid_pool_kernel), then the same deferred pass
#define ICP_GQP(D, F) \
    do { \
        if (g_grid_events) (void)hipEventRecord(g_grid_events[0], st); \
        hipLaunchKernelGGL((grid_pool_kernel<D, F>), dim3(nb), dim3(256), 0, st, src, n, qorder, (const double4 *)gb.recs, \
                           (const uint32_t *)gb.start, gb.gp, idx_out, d2_out, done, cap, dlist, ndef, hint, \
                           (const uint32_t *)gb.inv, (uint32_t)gb.m, pos_mode ? 1 : 0, xcd_chunk, fR, ft, \
                           fa ? (double *)ctx->err_part.ptr : (double *)nullptr, defer_max); \
        if (g_grid_events) (void)hipEventRecord(g_grid_events[1], st); \
        if (cap < 1000) \
            hipLaunchKernelGGL((grid_defer_kernel<D, false, false>), dim3(nb2), dim3(64), 0, st, (const double4 *)gb.recs, \
                               (const uint32_t *)gb.start, gb.gp, thr, idx_out, d2_out, done, (const Deferred *)dlist, \
                               (const uint32_t *)ndef, ndef_next, (const uint32_t *)gb.inv, pos_mode ? 1 : 0, \
                               jump_env >= 0 ? jump_env : ((hint && pos_mode) ? 1 : 0), split_env, clb, csr, defer_max); \
        if (g_grid_events) (void)hipEventRecord(g_grid_events[2], st); \
    } while (0)
    static const int pool_env = getenv("ICP_GRID_POOL") ? atoi(getenv("ICP_GRID_POOL")) : 1;   // (0: per-lane kernel everywhere; 2: the pool kernel for hinted searches only)
    const bool pool = pool_env != 0 && (hint || pool_env != 2) && !le && !coh && !g_grid_dbg && !list8;
    static const bool coh_trace = getenv("ICP_COHERENT_TRACE") != nullptr;
    // share of unproven rows beyond which an iteration takes the whole-cloud search (ICP_COHERENT_LIMIT, default 0.12)
    static const double coh_frac = getenv("ICP_COHERENT_LIMIT") ? atof(getenv("ICP_COHERENT_LIMIT")) : 0.12;
    const uint32_t coh_limit = (uint32_t)std::min<double>(4e9, coh_frac * (double)n);
    static const unsigned coh_blocks = getenv("ICP_COHERENT_BLOCKS") ? (unsigned)atoi(getenv("ICP_COHERENT_BLOCKS")) : 10240u;   // (5120 / 8192 / 10240: 34.8 / 32.7 / 30.2 us)
    if (coh && fa) {
        if (gb.gp.dim == 3) ICP_GQC(3);
        else ICP_GQC(2);
    } else if (coh) {   // first iteration: the bounded search for every query (the list-of-8 probe has no bounded variant)
        if (gb.gp.dim == 3) ICP_GQ3(3, false, false, 4, 3, true);
        else ICP_GQ3(2, false, false, 4, 3, true);
    } else if (pool) {
        if (gb.gp.dim == 3) {
            if (fa) ICP_GQP(3, true);
            else ICP_GQP(3, false);
        } else {
            if (fa) ICP_GQP(2, true);
            else ICP_GQP(2, false);
        }
    } else if (gb.gp.dim == 3) {
        if (le) ICP_GQL(3, true, false);
        else if (fa) ICP_GQL(3, false, true);
        else ICP_GQL(3, false, false);
    } else {
        if (le) ICP_GQL(2, true, false);
        else if (fa) ICP_GQL(2, false, true);
        else ICP_GQL(2, false, false);
    }
#undef ICP_GQL
#undef ICP_GQP
#undef ICP_GQ2
#undef ICP_GQ3
#undef ICP_GQC
    ICP_HIP_CHECK(hipGetLastError());
    return ICP_OK;
}

}  // namespace icp
