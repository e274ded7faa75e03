"""Library-side RCCL (include/icp_hip.h icp_rccl_*): the collectives of the two sharded entry
points issued by the library itself on the context's stream.  The test box has ONE GPU and RCCL refuses two ranks on
one device, so what can be run here is a one-rank communicator with ICP_COMM_FORCE: the whole sharded code path
(pack -> ncclAllReduce -> logic kernels, halo pack / unpack, row masks) runs and every collective is really issued.
Checked: no host callback is made, and the results equal the callback path's bit for bit (same kernels, same sums)
and the single-GPU entry points'.  Multi-rank RCCL parity is unverified on hardware."""
import numpy as np
import pytest

import icp_oracle as orc
import c_oracle

pytestmark = pytest.mark.gpu
C = (0.1, 1000.0, 100.0)


@pytest.fixture(scope="module")
def rccl():
    import torch
    assert torch.cuda.is_available()
    from icp_amd import _lib
    r = _lib.LibraryRccl(world=1, rank=0)
    yield r
    r.close()


@pytest.mark.parametrize("n,no_fuse", [(3000, False), (200000, False), (200000, True)])
def test_rigid_loop_over_library_rccl(rccl, n, no_fuse):
    from icp_amd import rigid
    from icp_amd.dist import RigidComm, rigid_icp_dist
    src, tgt = orc.make_rigid_cloud(n, seed=4)
    run = 6
    ref = rigid.rigid_icp(src, tgt, 1e-3, run, want_idx=True, no_fuse=no_fuse)
    cb = RigidComm("cuda:0", world=1, force=True)                       # callbacks, every collective "issued" (one rank: identity)
    a = rigid_icp_dist(src, tgt, 1e-3, run, want_idx=True, comm=cb, no_fuse=no_fuse)
    iters = int(a["iters"].cpu()[0])
    assert cb.calls == [4] + [17] * run + [1]
    nat = RigidComm("cuda:0", world=1, rccl=rccl, force=True)           # the library's own ncclAllReduce
    b = rigid_icp_dist(src, tgt, 1e-3, run, want_idx=True, comm=nat, no_fuse=no_fuse)
    assert nat.calls == []                                              # zero host callbacks
    for k in ("T", "err_hist", "iters", "idx"):
        x, y = a[k].cpu().numpy(), b[k].cpu().numpy()
        if k == "err_hist":
            x, y = x[:iters], y[:iters]
        assert np.array_equal(x, y), k
    assert int(ref["iters"].cpu()[0]) == iters
    assert np.array_equal(ref["idx"].cpu().numpy(), b["idx"].cpu().numpy())
    np.testing.assert_allclose(b["T"].cpu().numpy(), ref["T"].cpu().numpy(), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(b["err_hist"].cpu().numpy()[:iters], ref["err_hist"].cpu().numpy()[:iters], rtol=1e-12)


@pytest.mark.parametrize("graph", [False, True])
def test_nicp_solve_over_library_rccl(rccl, monkeypatch, graph):
    from icp_amd import nicp, nicp_dist as nd
    v, edges, tgt = orc.make_grid_mesh(40, 36, seed=11)                 # 1 440 vertices
    nbr, wt = orc.knn_weights(v)
    nn = tgt[c_oracle.match_idx(v, tgt)[0]]
    opts = dict(max_lm_iters=3, cg_tol=1e-10)
    w = nicp.SparseWeights(nbr, wt, 3)
    P_ref, st_ref = nicp.solve(nicp.initial_params(len(v), 3), w, v, edges, nn, C, compat=False, opts=opts)
    monkeypatch.delenv("ICP_NICP_GRAPH_DIST", raising=False)
    cb = nd.ShardedNICP(v, nbr, wt, edges, tgt, world=1, rank=0, force=True)
    st_cb = cb.solve(cb.ctx.to_dev(nn), C, opts)
    assert cb.comm.calls["allreduce"] > 3 * st_cb["cg_iters"] // 2 and cb.comm.calls["exchange"] >= st_cb["cg_iters"]
    if graph:
        monkeypatch.setenv("ICP_NICP_GRAPH_DIST", "1")                  # the sharded CG chunk as a hipGraph (RCCL launches captured)
    nat = nd.ShardedNICP(v, nbr, wt, edges, tgt, world=1, rank=0, rccl=rccl, force=True)
    st = nat.solve(nat.ctx.to_dev(nn), C, opts)
    assert nat.comm.calls == {"allreduce": 0, "exchange": 0}            # zero host callbacks
    assert (st["graph_launches"] > 0) == graph
    for k in ("err", "err0", "lm_iters", "cg_iters", "nfev", "info"):
        assert st[k] == st_cb[k], k
    assert np.array_equal(nat.P.cpu().numpy(), cb.P.cpu().numpy())      # same kernels, same sums: same bits
    assert st["cg_iters"] >= 50
    np.testing.assert_allclose(st["err"], st_ref["err"], rtol=1e-9)
    np.testing.assert_allclose(nat.deform().cpu().numpy(), nicp.deform(P_ref, w, v).cpu().numpy(), rtol=1e-6, atol=1e-9)


def test_rccl_handle_errors():
    import ctypes
    from icp_amd import _lib
    lib = _lib.load_library()
    assert lib.icp_rccl_unique_id(None) == -1
    h = ctypes.c_void_p()
    ident = (ctypes.c_char * 128)()
    ctx = _lib.get_context()
    assert lib.icp_rccl_create(ctx.handle, ident, 3, 2, ctypes.byref(h)) == -1      # rank outside the world
    assert lib.icp_rccl_destroy(None) == 0
