import sys, time
sys.path.insert(0,'/root/repo')
import torch
from bench import make_cloud
from icp_amd import rigid
from icp_amd._lib import get_context
ctx=get_context(0)
src,tgt=make_cloud(1<<20,seed=1)
s,t=ctx.to_dev(src),ctx.to_dev(tgt)
for k in (2,1,1,1,30,30):
    torch.cuda.synchronize(); t0=time.perf_counter()
    r=rigid.rigid_icp(s,t,0.0,k,use_tol=False,want_idx=True,ctx=ctx)
    torch.cuda.synchronize(); print(k,'python wall %.0f us'%((time.perf_counter()-t0)*1e6),flush=True)
# do the first calls of a process run slower?  (host clock inside the library: entry -> GPU done)
for k in [20] * 10:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = rigid.rigid_icp(s, t, 0.0, k, use_tol=False, want_idx=True, ctx=ctx)
    torch.cuda.synchronize(); print(k, 'python wall %.0f us' % ((time.perf_counter() - t0) * 1e6), flush=True)
