"""world_size-2 gloo test of the sharded-ICP host logic (icp_amd/dist.py) on CPU.
The device steps are replaced by an oracle-backed stand-in with the same 16-slot sums
layout as icp_shard_sums, so what is tested is the sharding, the reduction layout, the
redundant solve and the loop exit -- not the kernels (those are tested with -m gpu)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import icp_oracle as orc
from icp_amd.dist import Comm, shard_bounds, sharded_icp


class OracleOps:
    """numpy/torch-CPU stand-in for GpuOps (same call signatures)."""

    def to_dev(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64))

    def zeros(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def column_sums(self, a):
        out = torch.zeros(3, dtype=torch.float64)
        out[:a.shape[1]] = a.sum(0)
        return out

    def new_idx(self, n):
        return torch.zeros(n, dtype=torch.int32)

    def closest(self, ss, tgt, idx):
        i, _ = orc.match_idx(ss.numpy(), tgt.numpy())
        idx.copy_(torch.from_numpy(i))

    def shard_sums(self, src, tgt, idx, cs, ct, out):
        dim = src.shape[1]
        s = src.numpy() - cs.numpy()[:dim]
        t = tgt.numpy()[idx.numpy()] - ct.numpy()[:dim]
        v = np.zeros(16)
        v[0:dim] = t.sum(0)
        v[3:3 + dim] = s.sum(0)
        v[6:6 + dim * dim] = (t.T @ s).reshape(-1)
        v[15] = src.shape[0]
        out.copy_(torch.from_numpy(v))

    def kabsch_solve(self, sums, dim, cs, ct, R, t):
        v = sums.numpy()
        N = v[15]
        sn, ss = v[0:dim], v[3:3 + dim]
        W = v[6:6 + dim * dim].reshape(dim, dim) - np.outer(sn, ss) / N
        U, S, VT = np.linalg.svd(W)
        Rm = (U @ VT).T
        tbar = ct.numpy()[:dim] + sn / N
        sbar = cs.numpy()[:dim] + ss / N
        R.copy_(torch.from_numpy(Rm.reshape(-1).copy()))
        t.copy_(torch.from_numpy(tbar - sbar @ Rm))

    def apply_err(self, src, tgt, idx, R, t, ss, err):
        dim = src.shape[1]
        moved = src.numpy() @ R.numpy().reshape(dim, dim) + t.numpy()
        ss.copy_(torch.from_numpy(moved))
        err[0] = orc.distance(tgt.numpy()[idx.numpy()], moved)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    src, tgt = orc.make_rigid_cloud(601, seed=9)
    lo, hi = shard_bounds(src.shape[0], world, rank)
    T, hist = sharded_icp(OracleOps(), Comm(), src[lo:hi], tgt, 0.10, 50)
    q.put((rank, T, hist))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    for n in (0, 1, 7, 8, 1000):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_sharded_icp_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    src, tgt = orc.make_rigid_cloud(601, seed=9)
    hist = []
    T = orc.icp(src, tgt, 0.10, 50, history=hist)
    for rank, Tr, hr in res:
        np.testing.assert_allclose(Tr, T, rtol=1e-9, atol=1e-12)
        assert len(hr) == len(hist)
        np.testing.assert_allclose(hr, [h[1] for h in hist], rtol=1e-9, atol=1e-9)
    assert np.array_equal(res[0][1], res[1][1])      # ranks agree bit for bit


def test_sharded_icp_single_process_matches():
    src, tgt = orc.make_rigid_cloud(300, seed=4)
    T, hist = sharded_icp(OracleOps(), Comm(enabled=False), src, tgt, 0.10, 50)
    np.testing.assert_allclose(T, orc.icp(src, tgt, 0.10, 50), rtol=1e-9, atol=1e-12)
