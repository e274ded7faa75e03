import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import icp_oracle as orc
from icp_amd import rigid
src, tgt = orc.make_rigid_cloud(1 << 20, seed=1)
sd, td = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
r3 = [rigid.rigid_icp(sd, td, 0.0, 3, use_tol=False, want_idx=True) for _ in range(3)]
r2 = [rigid.rigid_icp(sd, td, 0.0, 2, use_tol=False, want_idx=True) for _ in range(3)]
r1 = [rigid.rigid_icp(sd, td, 0.0, 1, use_tol=False, want_idx=True) for _ in range(3)]
for name, rs in (("r3", r3), ("r2", r2), ("r1", r1)):
    print(name, [bool(torch.equal(rs[0]["T"], r["T"])) for r in rs], [bool(torch.equal(rs[0]["idx"], r["idx"])) for r in rs],
          [r["err_hist"].cpu().numpy().tolist() for r in rs][:1])
for r in r2:
    prev = r["T"]
    ss = rigid.apply_transform(sd, prev[:3, :3].contiguous(), prev[3, :3].contiguous())
    bi = rigid.closest_indices(ss, td, force_brute=True)
    print("mismatches vs r3 idx:", int((bi != r3[0]["idx"]).sum()))
prev = r1[0]["T"]
ss = rigid.apply_transform(sd, prev[:3, :3].contiguous(), prev[3, :3].contiguous())
bi = rigid.closest_indices(ss, td, force_brute=True)
print("r1->r2 mismatches:", int((bi != r2[0]["idx"]).sum()))
print((r2[0]["T"] - r3[0]["T"]).abs().max().item())
