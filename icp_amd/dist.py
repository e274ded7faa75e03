"""Source-sharded multi-GPU rigid ICP.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Source
rows are split contiguously, the target is replicated.  Per iteration every rank
runs the closest-point search on its shard (no communication), then ONE all-reduce
of 17 doubles (16 shifted sums for the centroid + cross-covariance, plus the error of
the previous iteration) and an identical redundant 3x3 SVD on every rank
(bit-identical inputs -> identical R,t, no broadcast).  The payload is 136 B, i.e.
latency-bound; nothing here depends on xGMI bandwidth.

The loop is written against a small `ops` interface so the host logic can be
exercised on CPU (gloo) in tests with a stand-in; the product `GpuOps` calls the
C-ABI and has no CPU fallback.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import ptr


def shard_bounds(n, world, rank):
    """Contiguous split of n rows over `world` ranks (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GpuOps:
    """The four device steps of one sharded ICP iteration, over include/icp_hip.h."""

    def __init__(self, ctx=None, compare_sqrt=False, force_brute=False):
        self.ctx = ctx or _lib.get_context()
        self.torch = self.ctx.torch
        self.flags = (_lib.ICP_COMPARE_SQRT if compare_sqrt else 0) | (_lib.ICP_FORCE_BRUTE if force_brute else 0)

    def to_dev(self, a):
        return self.ctx.to_dev(a)

    def begin(self):
        """Once per iteration: hand torch's current stream to the context (the per-call form
        costs two extra host calls per kernel, which shows at ~100 us per iteration)."""
        self.ctx.use_current_stream()

    def zeros(self, n):
        return self.torch.zeros(n, dtype=self.torch.float64, device="cuda:%d" % self.ctx.device)

    def column_sums(self, a):
        out = self.zeros(3)
        self.ctx.call("icp_column_sums", ptr(a), a.shape[0], a.shape[1], ptr(out))
        return out

    def prepare(self, src, tgt):
        """Once per run: search index over the (replicated, fixed) target and a Morton-ordered
        copy of the local source shard (sums are order-independent; a rigid motion keeps the
        order spatially coherent).  Returns the source array the iterations should use."""
        import ctypes
        self._tgt_keepalive = tgt
        h = ctypes.c_void_p()
        self.ctx.call("icp_target_create", ptr(tgt), tgt.shape[0], tgt.shape[1], ctypes.byref(h))
        self._target = h
        if src.shape[0] >= 32768 and bool(self.torch.isfinite(src).all()):
            out = self.torch.empty_like(src)
            order = self.new_idx(src.shape[0])
            self.ctx.call("icp_morton_sort", ptr(src), src.shape[0], src.shape[1], ptr(out), ptr(order))
            return out
        return src

    def closest(self, ss, tgt, idx):
        # from the second call on, idx still holds the previous correspondences: hand them in as hints
        flags = self.flags | (_lib.ICP_HINT_IDX if getattr(self, "_warm", None) is idx else 0)
        self._warm = idx
        if getattr(self, "_target", None) is not None:
            self.ctx.call_fast("icp_closest_target", self._target, ptr(ss), ss.shape[0], flags, ptr(idx), ptr(None))
        else:
            self.ctx.call_fast("icp_closest", ptr(ss), ss.shape[0], ptr(tgt), tgt.shape[0], ss.shape[1], self.flags,
                          ptr(idx), ptr(None))

    def __del__(self):
        try:
            if getattr(self, "_target", None) is not None:
                self.ctx.lib.icp_target_destroy(self.ctx.handle, self._target)
                self._target = None
        except Exception:
            pass

    def new_idx(self, n):
        return self.torch.empty(n, dtype=self.torch.int32, device="cuda:%d" % self.ctx.device)

    def shard_sums(self, src, tgt, idx, cs, ct, out):
        self.ctx.call_fast("icp_shard_sums", ptr(src), ptr(tgt), ptr(idx), src.shape[0], src.shape[1], ptr(cs), ptr(ct),
                      ptr(out))

    def kabsch_solve(self, sums, dim, cs, ct, R, t):
        self.ctx.call_fast("icp_kabsch_solve", ptr(sums), dim, ptr(cs), ptr(ct), ptr(R), ptr(t))

    def apply_err(self, src, tgt, idx, R, t, ss, err):
        self.ctx.call_fast("icp_shard_apply_err", ptr(src), ptr(tgt), ptr(idx), src.shape[0], src.shape[1], ptr(R), ptr(t),
                      ptr(ss), ptr(err))


class Comm:
    """Sum all-reduce over torch.distributed (or a no-op for a single process)."""

    def __init__(self, group=None, enabled=True):
        self.enabled = enabled
        self.group = group
        if enabled:
            import torch.distributed as dist
            self.dist = dist

    def all_reduce_sum(self, t):
        if self.enabled:
            if getattr(t, "is_cuda", False) and self.dist.get_backend(self.group) != "nccl":
                # gloo (ranks sharing a GPU, rehearsals): stage through host memory
                h = t.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t


class ShardedICP:
    """State of a sharded run: call `setup` once, then `step()` per iteration."""

    def __init__(self, ops, comm, src_shard, tgt):
        self.ops, self.comm = ops, comm
        self.src = ops.to_dev(src_shard)
        self.tgt = ops.to_dev(tgt)
        if hasattr(ops, "prepare"):
            self.src = ops.prepare(self.src, self.tgt)
        self.dim = self.src.shape[1]
        n_loc = self.src.shape[0]
        # iteration-invariant shifts: global mean of the ORIGINAL source (Q1) and target mean
        cs = ops.column_sums(self.src)            # [sx, sy, sz] local column sums
        cnt = ops.zeros(1) + float(n_loc)
        packed = ops.zeros(4)
        packed[:3] = cs
        packed[3:] = cnt
        comm.all_reduce_sum(packed)
        self.n_total = float(packed[3].item()) if hasattr(packed[3], "item") else float(packed[3])
        self.cs = packed[:3] / packed[3]
        self.cs = self.cs.contiguous() if hasattr(self.cs, "contiguous") else self.cs
        ct = ops.column_sums(self.tgt)
        self.ct = ct / float(self.tgt.shape[0])
        self.ss = self.src.clone() if hasattr(self.src, "clone") else self.src.copy()
        self.idx = ops.new_idx(n_loc)
        # one buffer for the per-iteration all-reduce: 16 sums of THIS iteration + the error of the
        # PREVIOUS one (a single latency-bound collective per iteration instead of two)
        self.buf = ops.zeros(17)
        self.sums = self.buf[:16]
        self.err = self.buf[16:17]
        self.R = ops.zeros(self.dim * self.dim)
        self.t = ops.zeros(self.dim)
        self.pending_err = False      # buf[16] holds a local (not yet reduced) error

    def begin_step(self):
        """Match + sums + the all-reduce.  Afterwards `self.err` is the GLOBAL error of the
        previous iteration (if there was one) and `self.sums` the global sums of this one."""
        o = self.ops
        if hasattr(o, "begin"):
            o.begin()
        o.closest(self.ss, self.tgt, self.idx)
        o.shard_sums(self.src, self.tgt, self.idx, self.cs, self.ct, self.sums)
        self.comm.all_reduce_sum(self.buf)
        self.pending_err = False

    def finish_step(self):
        """Kabsch solve (redundant, identical on every rank) + apply + local error."""
        o = self.ops
        o.kabsch_solve(self.sums, self.dim, self.cs, self.ct, self.R, self.t)
        o.apply_err(self.src, self.tgt, self.idx, self.R, self.t, self.ss, self.err)
        self.pending_err = True

    def flush_err(self):
        """All-reduce the error of the last finished iteration (end of a run)."""
        if self.pending_err:
            self.comm.all_reduce_sum(self.err)
            self.pending_err = False
        return self.err

    def step(self):
        """One ICP iteration (icp_3d.py:108-114) across all ranks.  The error it leaves in
        `self.err` is local until the next begin_step() / flush_err() reduces it."""
        self.begin_step()
        self.finish_step()
        return self.err

    def T(self):
        dim = self.dim
        R = np.asarray(self.R.cpu() if hasattr(self.R, "cpu") else self.R).reshape(dim, dim)
        t = np.asarray(self.t.cpu() if hasattr(self.t, "cpu") else self.t)
        T = np.identity(dim + 1)
        T[:dim, :dim] = R
        T[dim, :dim] = t
        return T


def sharded_icp(ops, comm, src_shard, tgt, tol, run, use_tol=True):
    """ICP (icp_3d.py:102-121) over a sharded source; every rank returns the same
    (T, err_history).  The stop test reads the all-reduced error, which is identical on
    all ranks, so ranks leave the loop together."""
    if run <= 0:
        raise NameError("name 'R' is not defined")
    st = ShardedICP(ops, comm, src_shard, tgt)
    err = 100000000
    hist = []
    done = 0
    while True:
        # the all-reduce of iteration k also delivers the error of iteration k-1, so the stop test
        # of icp_3d.py:115-116 is evaluated one Match late and that Match is discarded
        if done < run:
            st.begin_step()
        else:
            st.flush_err()
        if done > 0:
            e, err = err, float(st.err[0])
            hist.append(err)
            if (use_tol and (e - err) < tol) or done >= run:
                break
        st.finish_step()
        done += 1
    return st.T(), hist


# ---------------------------------------------------------------------------------------------
# The same loop below Python: icp_rigid_run_dist (include/icp_hip.h) enqueues every iteration itself and calls
# back only for the all-reduce, so a rank issues no host-language call per kernel and never reads the device
# inside the loop.  This is what bench.py --gpus N runs; ShardedICP above is the step-by-step form.
class IcpComm(ctypes.Structure):
    """icp_comm of include/icp_hip.h"""
    _fields_ = [("rank", ctypes.c_int), ("world", ctypes.c_int), ("reduce_buf_dev", ctypes.c_void_p),
                ("user", ctypes.c_void_p), ("allreduce_sum", ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int)),
                ("rccl", ctypes.c_void_p), ("flags", ctypes.c_uint)]


class RigidComm:
    """The all-reduce callback of icp_comm over torch.distributed.  backend "nccl" (= RCCL): the collective runs on
    the device buffer, ordered on torch's current stream = the stream the context launches on, no host
    synchronisation.  Any other backend (gloo: ranks sharing one GPU in rehearsals and tests) stages through the host.
    `device="cpu"` exercises the protocol without a GPU (tests/test_dist_cpu.py).
    `rccl`: a _lib.LibraryRccl -- the library then issues the collectives itself on the context's stream and this
    object's callback is never called (`calls` stays empty).  `force`: ICP_COMM_FORCE, every collective is issued even
    with one rank."""

    def __init__(self, device, group=None, rank=None, world=None, rccl=None, force=False):
        import torch
        self.torch, self.group = torch, group
        self.dist = None
        if world is None or world > 1:
            import torch.distributed as dist
            self.dist = dist
            rank = dist.get_rank(group) if rank is None else rank
            world = dist.get_world_size(group) if world is None else world
        self.rank, self.world = int(rank or 0), int(world)
        self.reduce_buf = torch.zeros(32, dtype=torch.float64, device=device)
        self.native = self.world > 1 and self.reduce_buf.is_cuda and self.dist.get_backend(group) == "nccl"
        self.calls = []          # doubles per collective, in call order
        self.error = None
        self._cb = IcpComm._fields_[4][1](self._allreduce)
        self.rccl = rccl
        self.struct = IcpComm(self.rank, self.world, ctypes.c_void_p(self.reduce_buf.data_ptr()), None, self._cb,
                              rccl.handle if rccl is not None else None, _lib.ICP_COMM_FORCE if force else 0)

    def allreduce(self, count):
        self.calls.append(count)
        if self.world == 1:
            return
        buf = self.reduce_buf[:count]
        if self.native:
            self.dist.all_reduce(buf, group=self.group)
        else:
            h = buf.cpu()
            self.dist.all_reduce(h, group=self.group)
            buf.copy_(h)

    def _allreduce(self, user, count):
        try:
            self.allreduce(count)
            return 0
        except Exception as e:          # never let an exception cross the C frames
            self.error = e
            return 1


def rigid_icp_dist(src_shard, target, tol, run, use_tol=True, compare_sqrt=False, force_brute=False, want_idx=False,
                   comm=None, ctx=None, no_fuse=False, force_grid=False, coherent=False):
    """Source-sharded ICP through icp_rigid_run_dist.  Every rank passes its rows of the source and the whole target;
    returns dict(T, err_hist, iters[, idx]) of device tensors, T / err_hist / iters identical on all ranks, idx for the
    rank's own rows."""
    ctx = ctx or _lib.get_context()
    torch = ctx.torch
    src, tgt = ctx.to_dev(src_shard), ctx.to_dev(target)
    if src.dim() != 2 or src.shape[1] not in (2, 3) or tgt.dim() != 2 or tgt.shape[1] != src.shape[1]:
        raise ValueError("points must be [N, dim] with dim 2 or 3 on both sides")
    dim, run = src.shape[1], int(run)
    if run <= 0:
        raise NameError("name 'R' is not defined")          # icp_3d.py:118-120
    comm = comm or RigidComm("cuda:%d" % ctx.device)
    T = ctx.empty((dim + 1, dim + 1))
    hist = ctx.empty((run,))
    iters = ctx.empty((1,), torch.int32)
    idx = ctx.empty((src.shape[0],), torch.int32) if want_idx else None
    flags = ((_lib.ICP_COMPARE_SQRT if compare_sqrt else 0) | (_lib.ICP_FORCE_BRUTE if force_brute else 0) |
             (_lib.ICP_NO_FUSE if no_fuse else 0) | (_lib.ICP_FORCE_GRID if force_grid else 0) |
             (_lib.ICP_COHERENT if coherent else 0))
    comm.error = None
    try:
        ctx.call("icp_rigid_run_dist", ptr(src), src.shape[0], ptr(tgt), tgt.shape[0], dim, ctypes.c_double(float(tol)), run,
                 1 if use_tol else 0, flags, ptr(T), ptr(hist), ptr(iters), ptr(idx), ctypes.byref(comm.struct))
    except _lib.IcpHipError:
        if comm.error is not None:
            raise comm.error
        raise
    out = dict(T=T, err_hist=hist, iters=iters)
    if want_idx:
        out["idx"] = idx
    return out
