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
from icp_amd.dist import Comm, RigidComm, shard_bounds, sharded_icp


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


# ---- the protocol of icp_rigid_run_dist (include/icp_hip.h), restated on the CPU around the REAL callback object
def rigid_run_dist_protocol(ops, comm, src_shard, tgt, tol, run, use_tol=True):
    """Line-for-line host mirror of icp_rigid_run_dist's loop (icp_amd/csrc/kabsch.hip): 4 doubles once, 17 per
    iteration (16 sums of this iteration + the error of the previous one), 1 at the end; the stop test of iteration
    k - 1 is evaluated after the all-reduce of iteration k; after `done` the collectives go on with zeros.  The
    device steps are the oracle stand-ins above; the all-reduce is RigidComm's (gloo here, RCCL on GPUs)."""
    src, tgt = ops.to_dev(src_shard), ops.to_dev(tgt)
    dim, n = src.shape[1], src.shape[0]
    buf = comm.reduce_buf
    buf[:3] = ops.column_sums(src)
    buf[3] = float(n)
    comm.allreduce(4)
    cs = torch.zeros(3, dtype=torch.float64)
    cs[:dim] = buf[:dim] / buf[3]
    ct = ops.column_sums(tgt) / float(tgt.shape[0])
    ss = src.clone()
    ib = [ops.new_idx(n), ops.new_idx(n)]
    R, t = ops.zeros(dim * dim), ops.zeros(dim)
    sums, err_loc = ops.zeros(16), ops.zeros(1)
    done, iters, err_prev, hist = False, 0, 100000000.0, []
    for it in range(run):
        idx = ib[it & 1]
        if not done:
            ops.closest(ss, tgt, idx)
            ops.shard_sums(src, tgt, idx, cs, ct, sums)
        buf[:16] = 0.0 if done else sums
        buf[16] = 0.0 if (done or it == 0) else err_loc[0]
        comm.allreduce(17)
        if not done and it > 0:
            e, err = err_prev, float(buf[16])
            err_prev = err
            hist.append(err)
            iters += 1
            if use_tol and (e - err) < tol:
                done = True
        if not done:
            ops.kabsch_solve(buf[:16].clone(), dim, cs, ct, R, t)
            ops.apply_err(src, tgt, idx, R, t, ss, err_loc)
    buf[0] = 0.0 if done else err_loc[0]
    comm.allreduce(1)
    if not done:
        hist.append(float(buf[0]))
        iters += 1
    T = np.identity(dim + 1)
    T[:dim, :dim] = R.numpy().reshape(dim, dim)
    T[dim, :dim] = t.numpy()
    return T, hist, iters, ib[(iters - 1) & 1].numpy().copy()


def _protocol_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    src, tgt = orc.make_rigid_cloud(601, seed=9)
    lo, hi = shard_bounds(src.shape[0], world, rank)
    out = []
    for tol, run in ((0.10, 50), (-np.inf, 5)):
        cm = RigidComm("cpu")
        T, hist, iters, idx = rigid_run_dist_protocol(OracleOps(), cm, src[lo:hi], tgt, tol, run)
        out.append((T, hist, iters, idx, list(cm.calls)))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_rigid_run_dist_protocol_gloo_world2():
    """world-2 gloo run of the sharded loop's collective protocol through the product's callback object
    (icp_amd.dist.RigidComm): both ranks issue run + 2 collectives of 4 / 17 / 1 doubles, agree bit for bit, stop
    together, and reproduce the oracle's ICP (transform, error history, last correspondences)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_protocol_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [o for _, o in sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    src, tgt = orc.make_rigid_cloud(601, seed=9)
    for case, (tol, run) in enumerate(((0.10, 50), (-np.inf, 5))):
        hist = []
        T = orc.icp(src, tgt, tol, run, history=hist)
        r0, r1 = res[0][case], res[1][case]
        assert np.array_equal(r0[0], r1[0]) and r0[1] == r1[1] and r0[2] == r1[2]
        np.testing.assert_allclose(r0[0], T, rtol=1e-9, atol=1e-12)
        assert r0[2] == len(hist)
        np.testing.assert_allclose(r0[1], [h[1] for h in hist], rtol=1e-9, atol=1e-9)
        assert r0[4] == r1[4] == [4] + [17] * run + [1]
        idx_last = np.concatenate([r0[3], r1[3]])
        assert np.array_equal(idx_last, hist[-1][0])


def test_rigid_comm_single_rank_is_a_noop():
    cm = RigidComm("cpu", rank=0, world=1)
    cm.reduce_buf[:3] = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)
    assert cm._allreduce(None, 3) == 0 and cm.calls == [3]
    assert cm.reduce_buf[:3].tolist() == [1.0, 2.0, 3.0] and cm.struct.world == 1
