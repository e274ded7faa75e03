import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import icp_oracle as orc
from icp_amd import rigid
src, tgt = orc.make_rigid_cloud(1 << 20, seed=1)
sd, td = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
for _ in range(3):
    r = rigid.rigid_icp(sd, td, 0.0, 30, use_tol=False, want_idx=True, coherent=True)
    torch.cuda.synchronize()
