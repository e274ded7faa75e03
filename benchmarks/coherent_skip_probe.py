"""How many C2 queries could a second-nearest-neighbour bound prove unchanged from one ICP iteration to the next?"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import icp_oracle as orc
from icp_amd import rigid, nicp
src, tgt = orc.make_rigid_cloud(1 << 20, seed=1)
sd, td = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
Ts = {}
for k in (1, 2, 3, 4, 5, 6, 10, 11, 20, 21, 29, 30):
    Ts[k] = rigid.rigid_icp(sd, td, 0.0, k, use_tol=False)["T"]
def moved(k):   # the queries of iteration k+1 (after k solves)
    T = Ts[k]
    return rigid.apply_transform(sd, T[:3, :3].contiguous(), T[3, :3].contiguous())
for k in (1, 2, 3, 4, 5, 10, 20, 29):
    a, b = moved(k), moved(k + 1)
    idx, dist = nicp.knn(td, a, 6)
    d1, d2 = dist[:, 0], dist[:, 1]
    delta = (b - a).norm(dim=1)
    t1 = td[idx[:, 0].long()]
    d1n = (b - t1).norm(dim=1)
    ok = d1n * (1 + 1e-12) < (d2 - delta) * (1 - 1e-12)
    print("iteration %2d -> %2d: max move %.3e  mean d1 %.3e  mean d2 %.3e  provably unchanged %.4f" % (k + 1, k + 2, delta.max().item(), d1.mean().item(), d2.mean().item(), ok.double().mean().item()))
