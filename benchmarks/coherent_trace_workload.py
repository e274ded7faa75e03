"""Workload for a rocprofv3 --kernel-trace of the coherent loop: three 30-iteration icp_rigid_run calls with ICP_COHERENT at
2^20 x 2^20 (profiles/r3_kernel_stats_coherent.txt)."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from bench import make_cloud
from icp_amd import rigid
from icp_amd._lib import get_context
ctx = get_context(0)
src, tgt = make_cloud(1 << 20, seed=1)
s, t = ctx.to_dev(src), ctx.to_dev(tgt)
for _ in range(3):
    rigid.rigid_icp(s, t, 0.0, 30, use_tol=False, want_idx=True, ctx=ctx, coherent=True)
    torch.cuda.synchronize()
