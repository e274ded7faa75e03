// Stable sort of (cell code, point index) pairs for the Morton ordering of grid.hip.
//
// The counting sort of grid.hip used to fill its cells through atomic cursors, so the order of the points INSIDE a
// cell changed from run to run: harmless for the search (an exact argmin does not depend on the visiting order) but
// the Kabsch sums of icp_rigid_run run over the Morton-sorted source, and a different order of the same summands moved
// the transform in its last bits.  A stable LSD radix sort of the codes, with the point indices 0..m-1 as payload,
// leaves every cell in ascending index order: the ordering, hence every sum over it, is reproducible.
// The sort itself is rocPRIM's (a plain library operation, once per sorted SOURCE; grids that are only searched fill
// their cells without it, grid.hip scatter_atomic_kernel); only the significant code bits are compared.  At 2^20 pairs
// rocPRIM's default takes its block-sort + merge path (about 20 small kernels, 160 us); the configuration below
// sends everything above 64 k items down its Onesweep radix path instead.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace icp {

__global__ void iota_kernel(int32_t *__restrict__ a, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (int32_t)i;
}

// idx_out = the permutation that sorts codes_in stably; codes_out = the sorted codes.  idx_tmp: m ints of scratch.
// tmp: rocPRIM's temporary storage (grow-only; sorts that may run at the same time on two streams need one each).
int stable_sort_codes(icp_ctx *ctx, Scratch &tmp, const uint32_t *codes_in, uint32_t *codes_out, int32_t *idx_tmp, int32_t *idx_out,
                      int64_t m, int code_bits) {
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, idx_tmp, m);
    ICP_HIP_CHECK(hipGetLastError());
    // rocPRIM's default takes its block-sort + merge path up to 2^20 items (21 launches, 0.17 ms at 2^20 pairs: the
    // critical path of icp_rigid_run's set-up); above 64 k items the Onesweep radix path -- stable as
    // well, 2-3 passes over the 21-27 significant bits -- is the shorter one here
    using sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 65536>;
    size_t bytes = 0;
    ICP_HIP_CHECK(rocprim::radix_sort_pairs<sort_config>(nullptr, bytes, codes_in, codes_out, (const int32_t *)idx_tmp, idx_out,
                                                         (size_t)m, 0u, (unsigned)code_bits, st));
    if (int rc = ensure(ctx, tmp, bytes + 256)) return rc;
    ICP_HIP_CHECK(rocprim::radix_sort_pairs<sort_config>(tmp.ptr, bytes, codes_in, codes_out, (const int32_t *)idx_tmp, idx_out,
                                                         (size_t)m, 0u, (unsigned)code_bits, st));
    return ICP_OK;
}

}  // namespace icp
