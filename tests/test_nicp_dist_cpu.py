"""CPU tests of the vertex-sharded NICP host logic (icp_amd/nicp_dist.py).

The partition is checked against the oracle's DENSE Jacobian (self-consistent mode): the
per-rank products the kernels compute -- J_loc p on the local rows, J_loc^T on the owned
vertices, tail sums over the owned rows only -- must assemble to the global J^T J p, and the
masked residual norms must add up to the global energy.  A world_size-2 gloo test drives
TorchComm's all-reduce and halo exchange on CPU tensors."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import icp_oracle as orc
from icp_amd import nicp_dist as nd

C = (1.0, 50.0, 5.0)


def _mesh(nx=9, ny=8, seed=3):
    v, edges, tgt = orc.make_grid_mesh(nx, ny, seed=seed)
    nbr, wt = orc.knn_weights(v)
    rng = np.random.default_rng(seed)
    n = v.shape[0]
    blk = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=np.float64)
    P = np.append(np.tile(blk, n), np.zeros(6)) + 0.05 * rng.standard_normal(12 * n + 6)
    nn = tgt[orc.match_idx(orc.deform(P, nbr, wt, v), tgt)[0]]
    return v, edges, tgt, nbr, wt, P, nn


@pytest.mark.parametrize("world,nx,ny", [(2, 9, 8), (3, 9, 8), (5, 13, 11), (8, 14, 12)])
def test_partition_covers_and_orders(world, nx, ny):
    v, edges, tgt, nbr, wt, P, nn = _mesh(nx, ny)
    shards, owner = nd.make_shards(v, nbr, wt, edges, world)
    n = v.shape[0]
    assert sorted(np.concatenate([s.owned for s in shards]).tolist()) == list(range(n))
    sizes = [s.n_owned for s in shards]
    assert max(sizes) - min(sizes) <= 1
    e_owned = 0
    for s in shards:
        assert np.array_equal(s.local_ids[:s.n_owned], s.owned)
        assert (owner[s.halo] != s.rank).all()
        assert np.array_equal(owner[s.halo], np.sort(owner[s.halo]))           # grouped by source rank
        assert s.recv_counts.sum() == s.n_halo and s.send_counts[s.rank] == 0
        for q in range(world):                                                  # what s sends to q is what q expects from s
            assert s.send_counts[q] == shards[q].recv_counts[s.rank] or q == s.rank
        e_owned += int(s.row_mask[9 * len(s.local_ids):].sum()) // 6
    assert e_owned == len(edges)                                                # every edge row owned exactly once


@pytest.mark.parametrize("world,nx,ny", [(2, 9, 8), (3, 9, 8), (8, 14, 12)])
def test_sharded_operator_assembles_to_global(world, nx, ny):
    """(8 ranks = the C5 configuration's rank count: shards with several neighbours each, halos from up to 7 ranks)"""
    v, edges, tgt, nbr, wt, P, nn = _mesh(nx, ny)
    n, S, NG = v.shape[0], 12, 6
    J = orc.d_min_func(P, nbr, wt, v, edges, nn, C, compat=False)
    r = orc.min_func(P, nbr, wt, v, edges, nn, C, compat=False)
    rng = np.random.default_rng(0)
    p = rng.standard_normal(len(P))
    want = J.T @ (J @ p)
    want_grad = J.T @ r
    shards, owner = nd.make_shards(v, nbr, wt, edges, world)
    got_v, got_tail, grad_v, grad_tail, energy = [], np.zeros(NG), [], np.zeros(NG), 0.0
    for s in shards:
        nl = len(s.local_ids)
        Pl, pl = s.local_params(P), s.local_params(p)
        vl, nnl = s.localize(v, 3), s.localize(nn, 3)
        Jl = orc.d_min_func(Pl, s.nbr, s.wt, vl, s.edges, nnl, C, compat=False)
        rl = orc.min_func(Pl, s.nbr, s.wt, vl, s.edges, nnl, C, compat=False)
        assert len(rl) == len(s.row_mask)
        energy += float(np.sum(s.row_mask * rl * rl))
        yl = Jl @ pl
        full = Jl.T @ yl
        got_v.append(full[:nl * S].reshape(nl, S))
        got_tail += Jl[:, nl * S:].T @ (s.row_mask * yl)
        gl = Jl.T @ rl
        grad_v.append(gl[:nl * S].reshape(nl, S))
        grad_tail += Jl[:, nl * S:].T @ (s.row_mask * rl)
    np.testing.assert_allclose(energy, r @ r, rtol=1e-12)
    got = np.concatenate([nd.assemble(shards, got_v, S).ravel(), got_tail])
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-9 * np.abs(want).max())
    gg = np.concatenate([nd.assemble(shards, grad_v, S).ravel(), grad_tail])
    np.testing.assert_allclose(gg, want_grad, rtol=1e-10, atol=1e-9 * np.abs(want_grad).max())


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_extended_energy_assembles_to_global(world):
    """Row multipliers and landmark rows (nicp_set_weights / nicp_set_landmarks) cut per rank by
    Shard.local_weights / local_landmarks: energy, J^T J p and the gradient still assemble."""
    v, edges, tgt, nbr, wt, P, nn = _mesh()
    n, S, NG = v.shape[0], 12, 6
    rng = np.random.default_rng(9)
    fw, gw, ew = rng.uniform(0, 2, n), rng.uniform(0.5, 1.5, n), rng.uniform(0.2, 3, len(edges))
    lm_v = rng.permutation(n)[:n // 5]
    lm_p = v[lm_v] + 0.05 * rng.standard_normal((len(lm_v), 3))
    c_land = 3.0
    ext = dict(fit_w=fw, rigid_w=gw, edge_w=ew, lm_vertex=lm_v, lm_pos=lm_p, c_land=c_land)
    J = orc.d_min_func_ext(P, nbr, wt, v, edges, nn, C, **ext)
    r = orc.min_func_ext(P, nbr, wt, v, edges, nn, C, **ext)
    p = rng.standard_normal(len(P))
    want, want_grad = J.T @ (J @ p), J.T @ r
    shards, owner = nd.make_shards(v, nbr, wt, edges, world)
    got_v, got_tail, grad_v, grad_tail, energy, owned_lm = [], np.zeros(NG), [], np.zeros(NG), 0.0, 0
    for s in shards:
        nl = len(s.local_ids)
        Pl, pl = s.local_params(P), s.local_params(p)
        vl, nnl = s.localize(v, 3), s.localize(nn, 3)
        f, g, e = s.local_weights(fw, gw, ew)
        sel, lv, mask = s.local_landmarks(lm_v)
        owned_lm += int(mask[len(s.row_mask):].sum()) // 3
        extl = dict(fit_w=f, rigid_w=g, edge_w=e, lm_vertex=lv, lm_pos=lm_p[sel], c_land=c_land)
        Jl = orc.d_min_func_ext(Pl, s.nbr, s.wt, vl, s.edges, nnl, C, **extl)
        rl = orc.min_func_ext(Pl, s.nbr, s.wt, vl, s.edges, nnl, C, **extl)
        assert len(rl) == len(mask)
        energy += float(np.sum(mask * rl * rl))
        yl = Jl @ pl
        got_v.append((Jl.T @ yl)[:nl * S].reshape(nl, S))
        got_tail += Jl[:, nl * S:].T @ (mask * yl)
        grad_v.append((Jl.T @ rl)[:nl * S].reshape(nl, S))
        grad_tail += Jl[:, nl * S:].T @ (mask * rl)
    assert owned_lm == len(lm_v)                                                # every landmark row owned exactly once
    np.testing.assert_allclose(energy, r @ r, rtol=1e-12)
    got = np.concatenate([nd.assemble(shards, got_v, S).ravel(), got_tail])
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-9 * np.abs(want).max())
    gg = np.concatenate([nd.assemble(shards, grad_v, S).ravel(), grad_tail])
    np.testing.assert_allclose(gg, want_grad, rtol=1e-10, atol=1e-9 * np.abs(want_grad).max())


def test_single_rank_shard_is_the_mesh():
    v, edges, tgt, nbr, wt, P, nn = _mesh(6, 5)
    shards, owner = nd.make_shards(v, nbr, wt, edges, 1)
    s = shards[0]
    assert s.n_halo == 0 and s.n_owned == v.shape[0] and len(s.send_idx) == 0
    assert np.array_equal(s.nbr, nbr) and np.array_equal(s.wt, wt) and np.array_equal(s.edges, edges)
    assert s.row_mask.all()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    v, edges, tgt, nbr, wt, P, nn = _mesh()
    shards, owner = nd.make_shards(v, nbr, wt, edges, world)
    s = shards[rank]
    comm = nd.TorchComm(s, "cpu")
    S = 12
    # a per-vertex vector whose value encodes (global id, component)
    f = lambda ids: (ids[:, None] * 100.0 + np.arange(S)[None, :])
    vec = torch.zeros((len(s.local_ids), S), dtype=torch.float64)
    vec[:s.n_owned] = torch.from_numpy(f(s.owned))
    ok = True
    for _ in range(2):   # twice: buffers are reused
        comm.send_buf[:len(s.send_idx) * S] = vec[comm.send_idx.long()].reshape(-1)     # what pack_kernel does
        assert comm._exchange(None) == 0
        vec[s.n_owned:] = comm.recv_buf[:s.n_halo * S].reshape(-1, S)
        ok &= bool(np.array_equal(vec.numpy(), f(s.local_ids)))
    comm.reduce_buf[:5] = torch.arange(5, dtype=torch.float64) + rank
    assert comm._allreduce(None, 3) == 0
    red = comm.reduce_buf[:5].numpy().copy()
    q.put((rank, ok, red, dict(comm.calls)))
    dist.barrier()
    dist.destroy_process_group()


def test_torchcomm_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, red, calls in res:
        assert ok, "halo exchange delivered wrong rows on rank %d" % rank
        np.testing.assert_array_equal(red[:3], [1.0, 3.0, 5.0])          # summed over 2 ranks
        np.testing.assert_array_equal(red[3:], [3.0 + rank, 4.0 + rank])  # beyond count: untouched
        assert calls == {"allreduce": 1, "exchange": 2}
