"""diagnostic: icp_rigid_run / icp_closest issued under different torch streams"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import icp_oracle as orc
from icp_amd import rigid

src, tgt = orc.make_rigid_cloud(200000, seed=9)
rng = np.random.default_rng(10)
src2, tgt2 = rng.standard_normal((150000, 3)), rng.standard_normal((180000, 3))
sd, td, s2d, t2d = (torch.from_numpy(a).cuda() for a in (src, tgt, src2, tgt2))

def run():
    return rigid.rigid_icp(sd, td, -np.inf, 6, want_idx=True)

def cmp(tag, r, ref):
    torch.cuda.synchronize()
    dT = float((r["T"] - ref["T"]).abs().max())
    nidx = int((r["idx"] != ref["idx"]).sum())
    dh = float((r["err_hist"] - ref["err_hist"]).abs().max())
    print(tag, "dT", dT, "idx mismatches", nidx, "d err_hist", dh, flush=True)

ref = run(); torch.cuda.synchronize()
ref = {k: v.clone() for k, v in ref.items()}
print("T[0]", ref["T"][0].tolist(), "err_hist", ref["err_hist"].tolist())
cmp("repeat default", run(), ref)
c_ref = rigid.closest_indices(s2d, t2d, want_d2=True); torch.cuda.synchronize()
c_ref = (c_ref[0].clone(), c_ref[1].clone())
cmp("after closest, default", run(), ref)
a, b = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
with torch.cuda.stream(a):
    r = run()
cmp("stream a only", r, ref)
with torch.cuda.stream(a):
    r = run()
torch.cuda.synchronize()
with torch.cuda.stream(b):
    c = rigid.closest_indices(s2d, t2d, want_d2=True)
torch.cuda.synchronize()
print("closest on b after sync equal:", torch.equal(c[0], c_ref[0]), torch.equal(c[1], c_ref[1]))
cmp("a then sync then b", r, ref)
for rep in range(3):
    with torch.cuda.stream(a):
        r = run()
    with torch.cuda.stream(b):
        c = rigid.closest_indices(s2d, t2d, want_d2=True)
    with torch.cuda.stream(a):
        r2 = run()
    torch.cuda.synchronize()
    cmp("rep %d r " % rep, r, ref)
    cmp("rep %d r2" % rep, r2, ref)
    print("   closest equal:", torch.equal(c[0], c_ref[0]), torch.equal(c[1], c_ref[1]), "mismatch", int((c[0] != c_ref[0]).sum()))
