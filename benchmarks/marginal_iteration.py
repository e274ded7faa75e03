"""Marginal ICP iteration of icp_rigid_run at 2^20 x 2^20 from 1-, 30- and 60-step calls (A/B of two library builds on one box: swap icp_amd/libicp_hip.so between runs)."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import icp_oracle as orc
from icp_amd import rigid
src, tgt = orc.make_rigid_cloud(1 << 20, seed=1)
sd, td = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
def t(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rigid.rigid_icp(sd, td, 0.0, k, use_tol=False, want_idx=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e6
t(2); t(30)
a = min(t(1) for _ in range(5)); b = min(t(30) for _ in range(5)); c = min(t(60) for _ in range(3))
print("1-step %.0f us  30-step %.0f us  60-step %.0f  marginal %.1f / %.1f us" % (a, b, c, (b - a) / 29, (c - b) / 30))
