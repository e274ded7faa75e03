"""GPU tests of the vertex-sharded NICP solve (nicp_solve_dist, icp_amd/nicp_dist.py).

The sharded solve must reach the same minimiser as the single-GPU nicp_solve (which the tests
in test_gpu_nicp.py pin to scipy's leastsq through the oracle): 1e-6 relative on parameters /
deformed vertices when both are driven to tight convergence.  Two ranks share the one GPU of
the test box, so the collectives go over gloo (host-staged); the RCCL path differs only in the
transport of the same buffers."""
import os

import numpy as np
import pytest

import icp_oracle as orc
import c_oracle

pytestmark = pytest.mark.gpu
C = (0.1, 1000.0, 100.0)
TIGHT = dict(ftol=1e-15, xtol=1e-15, cg_tol=1e-13)


def _problem(nx, ny, seed=5):
    v, edges, tgt = orc.make_grid_mesh(nx, ny, seed=seed)
    nbr, wt = orc.knn_weights(v)
    nn = tgt[c_oracle.match_idx(v, tgt)[0]]
    return v, edges, tgt, nbr, wt, nn


def _extension(v, edges):
    """Row multipliers + landmarks of the extended energy (nicp_set_weights / nicp_set_landmarks), global arrays."""
    rng = np.random.default_rng(17)
    n, ne = v.shape[0], len(edges)
    lm = rng.permutation(n)[:n // 6].astype(np.int32)
    return dict(fit_w=rng.uniform(0.2, 2.0, n), rigid_w=rng.uniform(0.5, 1.5, n), edge_w=rng.uniform(0.3, 3.0, ne)), \
        dict(vertex=lm, pos=v[lm] + 0.03 * rng.standard_normal((len(lm), 3)), c_land=5.0)


def _single(v, edges, nbr, wt, nn, opts, ext=False):
    from icp_amd import nicp
    w = nicp.SparseWeights(nbr, wt, 3)
    if ext:
        wts, lms = _extension(v, edges)
        w.set_weights(edges, **wts)
        w.set_landmarks(edges, **lms)
    P0 = nicp.initial_params(v.shape[0], 3)
    P, st = nicp.solve(P0, w, v, edges, nn, C, compat=False, opts=opts)
    return P.cpu().numpy(), st


def test_world1_is_nicp_solve():
    """A single-rank comm takes exactly the nicp_solve path (same kernels, no callbacks)."""
    import torch
    from icp_amd import nicp_dist as nd
    v, edges, tgt, nbr, wt, nn = _problem(12, 10)
    P_ref, st_ref = _single(v, edges, nbr, wt, nn, TIGHT)
    sh = nd.ShardedNICP(v, nbr, wt, edges, tgt, world=1, rank=0)
    st = sh.solve(sh.ctx.to_dev(nn), C, TIGHT)
    assert sh.comm.calls == {"allreduce": 0, "exchange": 0}
    assert st == st_ref
    assert np.array_equal(sh.P.cpu().numpy(), P_ref)


def test_compat_flag_is_refused():
    import ctypes
    from icp_amd import nicp_dist as nd, _lib
    v, edges, tgt, nbr, wt, nn = _problem(6, 5)
    sh = nd.ShardedNICP(v, nbr, wt, edges, tgt, world=1, rank=0)
    st = _lib.SolveStats()
    c3 = (ctypes.c_double * 3)(*C)
    with pytest.raises(_lib.IcpHipError, match="does not shard"):
        sh.ctx.call("nicp_solve_dist", sh.graph, _lib.ptr(sh.P), _lib.ptr(sh.v), _lib.ptr(sh.ctx.to_dev(nn)), c3,
                    _lib.NICP_REFERENCE_COMPAT, None, ctypes.byref(st), ctypes.byref(sh.comm.struct))


def _worker(rank, world, port, q, nx, ny, mode):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from icp_amd import nicp_dist as nd
        v, edges, tgt, nbr, wt, nn = _problem(nx, ny)
        sh = nd.ShardedNICP(v, nbr, wt, edges, tgt)
        if mode == "solve_rich":
            os.environ["ICP_NICP_COARSE"] = "10"          # enriched coarse space on a small mesh: 62-double all-reduces
        if mode == "solve_ext":
            wts, lms = _extension(v, edges)
            sh.set_weights(**wts)
            sh.set_landmarks(**lms)
        if mode in ("solve", "solve_ext", "solve_rich"):
            st = sh.solve(sh.ctx.to_dev(sh.shard.localize(nn, 3)), C, TIGHT)
            out = dict(st=st)
        else:
            hist = []
            c = sh.run(0.01, 4, C, opts=TIGHT, history=hist)
            out = dict(c=c, errs=[h[0] for h in hist])
        ids, Pv, tail = sh.owned_params()
        _, dd = sh.owned_deformed()
        out.update(ids=ids, Pv=Pv, tail=tail, dd=dd, calls=dict(sh.comm.calls), n_halo=sh.shard.n_halo)
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_world(world, nx, ny, mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, nx, ny, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return [o for _, o in sorted(res, key=lambda x: x[0])]


def _assemble(res, n):
    P = np.zeros((n, 12))
    dd = np.zeros((n, 3))
    for o in res:
        P[o["ids"]] = o["Pv"]
        dd[o["ids"]] = o["dd"]
    return P, dd


@pytest.mark.parametrize("world", [2, 4])
def test_world2_solve_matches_single(world):
    """(world 4: every rank has halos from several others; all ranks share the test box's one GPU)"""
    nx, ny = 14, 12
    v, edges, tgt, nbr, wt, nn = _problem(nx, ny)
    n = v.shape[0]
    P_ref, st_ref = _single(v, edges, nbr, wt, nn, TIGHT)
    res = _run_world(world, nx, ny, "solve")
    P, dd = _assemble(res, n)
    for o in res:
        assert o["n_halo"] > 0 and o["calls"]["exchange"] > 0 and o["calls"]["allreduce"] > 0
        np.testing.assert_array_equal(o["tail"], res[0]["tail"])            # replicated globals stay bit-identical
        assert o["st"] == res[0]["st"]                                      # every rank took the same decisions
    st = res[0]["st"]
    print("single", st_ref, "sharded", st)
    np.testing.assert_allclose(st["err0"], st_ref["err0"], rtol=1e-12)
    np.testing.assert_allclose(st["err"], st_ref["err"], rtol=1e-9)
    np.testing.assert_allclose(P.ravel(), P_ref[:n * 12], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(res[0]["tail"], P_ref[n * 12:], rtol=1e-6, atol=1e-8)
    dd_ref = orc.deform(P_ref, nbr, wt, v)
    np.testing.assert_allclose(dd, dd_ref, rtol=1e-6, atol=1e-8)


def test_world2_extended_energy_matches_single():
    """Row multipliers and landmark rows through the sharded solve: every landmark row is counted by
    exactly one rank, halo copies carry the multipliers of their fit rows."""
    nx, ny = 13, 11
    v, edges, tgt, nbr, wt, nn = _problem(nx, ny)
    n = v.shape[0]
    P_ref, st_ref = _single(v, edges, nbr, wt, nn, TIGHT, ext=True)
    P_plain, st_plain = _single(v, edges, nbr, wt, nn, TIGHT)
    assert abs(st_ref["err0"] - st_plain["err0"]) > 1e-3 * st_plain["err0"]      # the extension really changes the energy
    res = _run_world(2, nx, ny, "solve_ext")
    P, dd = _assemble(res, n)
    st = res[0]["st"]
    assert res[1]["st"] == st
    print("single", st_ref, "sharded", st)
    np.testing.assert_allclose(st["err0"], st_ref["err0"], rtol=1e-12)
    np.testing.assert_allclose(st["err"], st_ref["err"], rtol=1e-9)
    np.testing.assert_allclose(dd, orc.deform(P_ref, nbr, wt, v), rtol=1e-6, atol=1e-8)


def test_world2_enriched_coarse_space_matches_single(monkeypatch):
    """The polynomial-enriched coarse space through the sharded solve: the 60 moments Z^T r are block sums over
    owned vertices, all-reduced with r.z and r.r; the basis (centre, radius) comes from all-reduced sums, so both
    ranks build the same one and take the same decisions."""
    nx, ny = 14, 12
    v, edges, tgt, nbr, wt, nn = _problem(nx, ny)
    n = v.shape[0]
    monkeypatch.setenv("ICP_NICP_COARSE", "10")
    P_ref, st_ref = _single(v, edges, nbr, wt, nn, TIGHT)
    monkeypatch.setenv("ICP_NICP_COARSE", "1")
    P_plain, st_plain = _single(v, edges, nbr, wt, nn, TIGHT)
    monkeypatch.delenv("ICP_NICP_COARSE")
    assert st_ref["cg_iters"] != st_plain["cg_iters"]                      # the other preconditioner really ran
    res = _run_world(2, nx, ny, "solve_rich")
    P, dd = _assemble(res, n)
    st = res[0]["st"]
    assert res[1]["st"] == st
    print("single", st_ref, "sharded", st)
    np.testing.assert_allclose(st["err"], st_ref["err"], rtol=1e-9)
    np.testing.assert_allclose(dd, orc.deform(P_ref, nbr, wt, v), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(dd, orc.deform(P_plain, nbr, wt, v), rtol=1e-6, atol=1e-8)


def test_world2_outer_loop_matches_single():
    from icp_amd import nicp
    nx, ny = 10, 9
    v, edges, tgt, nbr, wt, nn = _problem(nx, ny)
    n = v.shape[0]
    hist = []
    w = nicp.SparseWeights(nbr, wt, 3)
    dd_ref, P_ref, c_ref = nicp.nicp(nicp.initial_params(n, 3), w, v, edges, tgt, 0.01, 4, C, compat=False, opts=TIGHT,
                                     history=hist)
    res = _run_world(2, nx, ny, "run")
    P, dd = _assemble(res, n)
    assert res[0]["c"] == res[1]["c"] == c_ref
    np.testing.assert_allclose(res[0]["errs"], [h[0] for h in hist], rtol=1e-6)
    np.testing.assert_allclose(dd, dd_ref.cpu().numpy(), rtol=1e-6, atol=1e-8)


def test_c5_size_properties():
    """BASELINE config C5 size on one GPU (448 x 448 = 200 704 vertices): size-independent
    properties -- weights sum to 1, Deform(identity) is the source, <Jp, Jq> == <p, J^T J q>,
    a bounded solve lowers the energy and is deterministic, and the single-rank sharded entry
    point reproduces it bit for bit."""
    import torch
    from icp_amd import nicp, nicp_dist as nd
    v, edges, tgt = orc.make_grid_mesh(448, 448, seed=7)
    n = v.shape[0]
    w = nicp.knn_weights(v)
    np.testing.assert_allclose(w.sum(0), 1.0, atol=1e-13)
    P0 = nicp.initial_params(n, 3)
    np.testing.assert_allclose(nicp.deform(P0, w, v).cpu().numpy(), v, rtol=0, atol=1e-13)
    rng = np.random.default_rng(0)
    P = P0 + 0.01 * rng.standard_normal(P0.shape)
    p, q = rng.standard_normal(P0.shape), rng.standard_normal(P0.shape)
    Jp, Jq = nicp.jvp(P, w, v, edges, C, p, compat=False), nicp.jvp(P, w, v, edges, C, q, compat=False)
    lhs = float((Jp * Jq).sum())
    rhs = float((torch.as_tensor(p, device=Jp.device) * nicp.jtvp(P, w, v, edges, C, Jq, compat=False)).sum())
    assert abs(lhs - rhs) <= 1e-9 * abs(lhs)
    from icp_amd import rigid
    nn = rigid.gather_points(w.ctx.to_dev(tgt), rigid.closest_indices(v, tgt))
    opts = dict(max_lm_iters=2, max_cg_iters=150, cg_tol=1e-6)
    P1, st1 = nicp.solve(P0, w, v, edges, nn, C, compat=False, opts=opts)
    P2, st2 = nicp.solve(P0, w, v, edges, nn, C, compat=False, opts=opts)
    assert st1["err"] < st1["err0"] and st1 == st2 and torch.equal(P1, P2)
    sh = nd.ShardedNICP(v, w.nbr.cpu().numpy(), w.wt.cpu().numpy(), edges, tgt, world=1, rank=0)
    st3 = sh.solve(nn, C, opts)
    assert st3 == st1 and torch.equal(sh.P, P1)
