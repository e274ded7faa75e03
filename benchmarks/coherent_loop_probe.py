"""Coherent loop (ICP_COHERENT) against the plain fused loop at 2^20 x 2^20: bit-equality of T / correspondences / errors for 2, 3, 5 and 30 iterations, then the wall time of 30- and 60-step calls either way."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import icp_oracle as orc
from icp_amd import rigid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
src, tgt = orc.make_rigid_cloud(n, seed=1)
sd, td = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
for k in (2, 3, 5, 30):
    a = rigid.rigid_icp(sd, td, 0.0, k, use_tol=False, want_idx=True)
    b = rigid.rigid_icp(sd, td, 0.0, k, use_tol=False, want_idx=True, coherent=True)
    torch.cuda.synchronize()
    print(k, "T", bool(torch.equal(a["T"], b["T"])), "idx", bool(torch.equal(a["idx"], b["idx"])), int((a["idx"] != b["idx"]).sum()),
          "err", bool(torch.equal(a["err_hist"], b["err_hist"])), float((a["err_hist"] - b["err_hist"]).abs().max()))
for coh in (False, True):
    for k in (30, 30, 30, 60):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = rigid.rigid_icp(sd, td, 0.0, k, use_tol=False, want_idx=True, coherent=coh)
        torch.cuda.synchronize(); print("coherent", coh, k, "%.0f us" % ((time.perf_counter() - t0) * 1e6))
