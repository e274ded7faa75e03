"""Workload for a rocprofv3 --kernel-trace timeline of ONE icp_rigid_run call (set-up + k iterations).
Run:  rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python3 benchmarks/setup_timeline.py [k]
then: python3 profiles/timeline_rocpd.py gpurun_out/tl/*/tl_results.db  (prints the last call's kernels in start order)."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from bench import make_cloud
from icp_amd import rigid
from icp_amd._lib import get_context
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = get_context(0)
src, tgt = make_cloud(1 << 20, seed=1)
s, t = ctx.to_dev(src), ctx.to_dev(tgt)
for _ in range(4):
    torch.cuda.synchronize()
    r = rigid.rigid_icp(s, t, 0.0, k, use_tol=False, want_idx=True, ctx=ctx)
    torch.cuda.synchronize()
