"""Why is the first K-step call of a process slower than the following ones?  (bench.py's `value` is that call.)
Sequences of icp_rigid_run calls at 2^20 x 2^20 after different warm-ups; prints ms per step of each call."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from bench import make_cloud
from icp_amd import rigid
from icp_amd._lib import get_context
ctx = get_context(0)
src, tgt = make_cloud(1 << 20, seed=1)
s, t = ctx.to_dev(src), ctx.to_dev(tgt)
def call(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rigid.rigid_icp(s, t, 0.0, k, use_tol=False, want_idx=True, ctx=ctx)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / k
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
if mode == "a":
    call(1)
elif mode == "b":
    call(30)
elif mode == "c":
    call(1); time.sleep(0.5)
elif mode == "d":
    for _ in range(20): call(1)
print(mode, " ".join("%.4f" % call(30) for _ in range(6)))
