"""Vertex-sharded multi-GPU NICP.

One process per GPU.  The mesh is cut into `world` vertex ranges along a Morton curve;
rank r OWNS the parameters (A_j, b_j) of its range and holds read-only HALO copies of
the foreign vertices its rows touch.  A rank's local problem has

  * the fit rows of every vertex i whose influence set nbr[i,:] contains an owned
    vertex (set F; computed redundantly on every rank that needs them),
  * the rigidity rows of its owned vertices,
  * the smoothness rows of every edge with an owned endpoint,

so J^T(J p) is complete on the owned vertices after ONE halo refresh of p per CG
iteration; the 6 global parameters are replicated and their sums all-reduced together
with the CG scalars (<= 16 doubles per all-reduce).  The target cloud is replicated:
every rank matches its own F rows, no correspondence is exchanged.

The partition below is plain numpy (tested on CPU); the solve itself is
`nicp_solve_dist` of include/icp_hip.h, which calls back into `TorchComm` for the two
collectives.  There is no CPU fallback for the solve.
"""
import ctypes

import numpy as np

from . import _lib

NUM_NN = 4


# --------------------------------------------------------------------------- partition
def morton_order(v):
    """Permutation that sorts the rows of v[n,dim] along a 21-bit-per-axis Morton curve."""
    v = np.asarray(v, dtype=np.float64)
    n, dim = v.shape
    lo, hi = v.min(0), v.max(0)
    span = np.where(hi > lo, hi - lo, 1.0)
    q = np.minimum(((v - lo) / span * (1 << 21)).astype(np.uint64), (1 << 21) - 1)
    code = np.zeros(n, np.uint64)
    for b in range(21):
        for a in range(dim):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(b * dim + a)
    return np.argsort(code, kind="stable")


def owners_by_morton(v, world):
    """owner[n]: rank that owns every vertex (contiguous Morton ranges of near-equal size)."""
    order = morton_order(v)
    n = len(order)
    owner = np.empty(n, np.int32)
    base, rem = divmod(n, world)
    pos = 0
    for r in range(world):
        cnt = base + (1 if r < rem else 0)
        owner[order[pos:pos + cnt]] = r
        pos += cnt
    return owner


class Shard:
    """Rank `rank`'s local mesh.  Local vertex ids: [owned (ascending global id) | halo
    (by owner rank, then global id)]."""

    def __init__(self, rank, world, owner, nbr, wt, edges, dim):
        nbr = np.asarray(nbr)
        wt = np.asarray(wt, dtype=np.float64)
        edges = np.asarray(edges).astype(np.int64).reshape(-1, 2)
        n = nbr.shape[0]
        self.rank, self.world, self.dim = rank, world, dim
        mine = owner == rank
        owned = np.nonzero(mine)[0]
        # F: rows whose influence set touches an owned vertex (plus the owned rows themselves)
        frow = mine[nbr].any(1) | mine
        f_ids = np.nonzero(frow)[0]
        e_sel = mine[edges[:, 0]] | mine[edges[:, 1]] if len(edges) else np.zeros(0, bool)
        e_loc = edges[e_sel]
        need = np.zeros(n, bool)
        need[f_ids] = True
        need[nbr[f_ids].ravel()] = True
        need[e_loc.ravel()] = True
        need[owned] = False
        halo = np.nonzero(need)[0]
        halo = halo[np.lexsort((halo, owner[halo]))]
        self.owned, self.halo = owned, halo
        self.n_owned, self.n_halo = len(owned), len(halo)
        self.local_ids = np.concatenate([owned, halo])            # local -> global
        g2l = np.full(n, -1, np.int64)
        g2l[self.local_ids] = np.arange(len(self.local_ids))
        self.g2l = g2l
        nl = len(self.local_ids)
        # local influence lists: rows outside F get weight 0 on themselves (their fit row is inert)
        self.nbr = np.tile(np.arange(nl, dtype=np.int32)[:, None], (1, nbr.shape[1]))
        self.wt = np.zeros((nl, nbr.shape[1]))
        fl = g2l[f_ids]
        self.nbr[fl] = g2l[nbr[f_ids]].astype(np.int32)
        self.wt[fl] = wt[f_ids]
        self.fit_rows = np.zeros(nl, bool)
        self.fit_rows[fl] = True
        self.edges = g2l[e_loc].astype(np.int32).reshape(-1, 2)
        self.edge_ids = np.nonzero(e_sel)[0]                      # local edge -> global edge
        assert (self.nbr >= 0).all() and (self.edges >= 0).all()
        # rows this rank owns (layout of nicp_residual: fit | rigid | smooth)
        nrig = 6 if dim == 3 else 3
        own_v = np.arange(nl) < self.n_owned
        own_e = owner[e_loc[:, 0]] == rank if len(e_loc) else np.zeros(0, bool)
        self.row_mask = np.concatenate([np.repeat(own_v, dim), np.repeat(own_v, nrig),
                                        np.repeat(own_e, 2 * dim)]).astype(np.float64)
        self.recv_counts = np.bincount(owner[halo], minlength=world).astype(np.int64)
        self.send_counts = None
        self.send_idx = None

    def localize(self, arr, width):
        """Rows of a per-vertex global array [n, width] (or flat) in local order."""
        return np.ascontiguousarray(np.asarray(arr).reshape(-1, width)[self.local_ids])

    def local_weights(self, fit_w=None, rigid_w=None, edge_w=None):
        """Global row multipliers (nicp_set_weights) cut to this rank's local vertices / edges."""
        cut = lambda a, ids: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64)[ids])
        return cut(fit_w, self.local_ids), cut(rigid_w, self.local_ids), cut(edge_w, self.edge_ids)

    def local_landmarks(self, vertex):
        """Landmarks this rank evaluates: those on a local vertex whose fit row it holds (set F).
        Returns (positions in the global landmark list, local vertex ids, row mask with the landmark
        rows appended: a landmark row belongs to the owner of its vertex)."""
        vertex = np.asarray(vertex).astype(np.int64).reshape(-1)
        loc = self.g2l[vertex]
        sel = np.nonzero((loc >= 0) & self.fit_rows[np.maximum(loc, 0)])[0]
        lv = loc[sel]
        mask = np.concatenate([self.row_mask, np.repeat((lv < self.n_owned).astype(np.float64), self.dim)])
        return sel, lv.astype(np.int32), mask

    def local_params(self, P):
        """Local parameter vector [owned | halo | tail] from the global one."""
        S = self.dim * self.dim + self.dim
        P = np.asarray(P, dtype=np.float64)
        n = (len(P) - (6 if self.dim == 3 else 3)) // S
        return np.concatenate([P[:n * S].reshape(n, S)[self.local_ids].ravel(), P[n * S:]])


def make_shards(v, nbr, wt, edges, world, owner=None):
    """All ranks' shards (deterministic; every rank can build the full list from the
    replicated mesh, which is how the send lists are agreed without communication)."""
    v = np.asarray(v)
    dim = v.shape[1]
    owner = owners_by_morton(v, world) if owner is None else np.asarray(owner, np.int32)
    shards = [Shard(r, world, owner, nbr, wt, edges, dim) for r in range(world)]
    for s in shards:
        # what s must send to q = the part of q's halo that s owns, in q's halo order
        lists = [shards[q].halo[owner[shards[q].halo] == s.rank] if q != s.rank else np.zeros(0, np.int64)
                 for q in range(world)]
        s.send_counts = np.array([len(x) for x in lists], np.int64)
        s.send_idx = s.g2l[np.concatenate(lists)].astype(np.int32) if world > 1 else np.zeros(0, np.int32)
        assert (s.send_idx >= 0).all() and (s.send_idx < s.n_owned).all()
    return shards, owner


def assemble(shards, locals_, width):
    """Global per-vertex array from every rank's owned rows."""
    n = sum(s.n_owned for s in shards)
    out = np.zeros((n, width))
    for s, a in zip(shards, locals_):
        out[s.owned] = np.asarray(a).reshape(-1, width)[:s.n_owned]
    return out


# --------------------------------------------------------------------------- collectives
class NicpComm(ctypes.Structure):
    """nicp_comm of include/icp_hip.h"""
    _fields_ = [("rank", ctypes.c_int), ("world", ctypes.c_int), ("n_owned", ctypes.c_int64),
                ("reduce_buf_dev", ctypes.c_void_p), ("send_idx_dev", ctypes.c_void_p), ("n_send", ctypes.c_int64),
                ("send_buf_dev", ctypes.c_void_p), ("recv_buf_dev", ctypes.c_void_p), ("row_mask_dev", ctypes.c_void_p),
                ("user", ctypes.c_void_p),
                ("allreduce_sum", ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int)),
                ("exchange", ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)),
                ("rccl", ctypes.c_void_p), ("send_counts_host", ctypes.c_void_p), ("recv_counts_host", ctypes.c_void_p),
                ("flags", ctypes.c_uint)]


class TorchComm:
    """The two callbacks of nicp_comm over torch.distributed.

    backend "nccl" (= RCCL): the collectives run on the device buffers and are ordered
    after the work queued on torch's current stream, which is the stream the context
    launches on -- no host synchronisation.  Any other backend (gloo: CPU tests, or
    several ranks sharing one GPU) stages through host memory.
    `device` may be "cpu" to exercise the exchange logic without a GPU.
    `rccl`: a _lib.LibraryRccl -- the library issues both collectives itself over RCCL on the context's stream and the
    callbacks are never called; `force`: ICP_COMM_FORCE (every collective issued even with one rank)."""

    def __init__(self, shard, device, group=None, rccl=None, force=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.shard = shard
        self.rank, self.world = shard.rank, shard.world
        S = shard.dim * shard.dim + shard.dim
        self.S = S
        f64 = torch.float64
        self.reduce_buf = torch.zeros(128, dtype=f64, device=device)
        self.send_idx = torch.from_numpy(shard.send_idx.astype(np.int32)).to(device)
        self.send_buf = torch.zeros(max(len(shard.send_idx), 1) * S, dtype=f64, device=device)
        self.recv_buf = torch.zeros(max(shard.n_halo, 1) * S, dtype=f64, device=device)
        self.row_mask = torch.from_numpy(shard.row_mask).to(device)
        self.in_splits = [int(c) * S for c in shard.send_counts]
        self.out_splits = [int(c) * S for c in shard.recv_counts]
        self.native = self.world > 1 and dist.is_initialized() and dist.get_backend(group) == "nccl" and self.reduce_buf.is_cuda
        self.calls = {"allreduce": 0, "exchange": 0}
        self.error = None
        self._cb_ar = NicpComm._fields_[10][1](self._allreduce)
        self._cb_ex = NicpComm._fields_[11][1](self._exchange)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        self.rccl = rccl
        self._send_counts = np.ascontiguousarray(shard.send_counts, dtype=np.int64)
        self._recv_counts = np.ascontiguousarray(shard.recv_counts, dtype=np.int64)
        self.struct = NicpComm(self.rank, self.world, shard.n_owned, p(self.reduce_buf), p(self.send_idx),
                               len(shard.send_idx), p(self.send_buf), p(self.recv_buf), p(self.row_mask), None,
                               self._cb_ar, self._cb_ex, rccl.handle if rccl is not None else None,
                               self._send_counts.ctypes.data, self._recv_counts.ctypes.data,
                               _lib.ICP_COMM_FORCE if force else 0)

    # plain methods (also used directly by the CPU tests)
    def allreduce(self, count):
        self.calls["allreduce"] += 1
        if self.world == 1:
            return
        buf = self.reduce_buf[:count]
        if self.native:
            self.dist.all_reduce(buf, group=self.group)
        else:
            h = buf.cpu()
            self.dist.all_reduce(h, group=self.group)
            buf.copy_(h)

    def exchange(self):
        self.calls["exchange"] += 1
        if self.world == 1:
            return
        ns, nr = sum(self.in_splits), sum(self.out_splits)
        send, recv = self.send_buf[:ns], self.recv_buf[:nr]
        if self.native:
            self.dist.all_to_all_single(recv, send, self.out_splits, self.in_splits, group=self.group)
            return
        hs = send.cpu()
        hr = self.torch.empty(nr, dtype=hs.dtype)
        ops, so, ro = [], 0, 0
        for q in range(self.world):
            if self.in_splits[q]:
                ops.append(self.dist.P2POp(self.dist.isend, hs[so:so + self.in_splits[q]], q, group=self.group))
            if self.out_splits[q]:
                ops.append(self.dist.P2POp(self.dist.irecv, hr[ro:ro + self.out_splits[q]], q, group=self.group))
            so += self.in_splits[q]
            ro += self.out_splits[q]
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        recv.copy_(hr)

    def _allreduce(self, user, count):
        try:
            self.allreduce(count)
            return 0
        except Exception as exc:      # an exception must not unwind through the C caller
            self.error = exc
            return 1

    def _exchange(self, user):
        try:
            self.exchange()
            return 0
        except Exception as exc:
            self.error = exc
            return 1


# --------------------------------------------------------------------------- sharded solve / outer loop
class ShardedNICP:
    """Rank-local state of a sharded NICP run over the C-ABI."""

    def __init__(self, source, nbr, wt, edges, target, P0=None, world=None, rank=None, owner=None, ctx=None, group=None,
                 rccl=None, force=False):
        import torch.distributed as dist
        from . import nicp as _nicp
        self.ctx = ctx or _lib.get_context()
        ctx = self.ctx
        torch = ctx.torch
        world = dist.get_world_size(group) if world is None else world
        rank = dist.get_rank(group) if rank is None else rank
        source = np.asarray(source, dtype=np.float64)
        self.dim = source.shape[1]
        self.n_global = source.shape[0]
        shards, self.owner = make_shards(source, nbr, wt, edges, world, owner)
        self.shard = sh = shards[rank]
        self.shards_meta = [(s.n_owned, s.n_halo) for s in shards]
        dev = "cuda:%d" % ctx.device
        self.comm = TorchComm(sh, dev, group, rccl=rccl, force=force)
        self.v = ctx.to_dev(sh.localize(source, self.dim))
        self.weights = _nicp.SparseWeights(sh.nbr, sh.wt, self.dim, ctx)
        self.graph, self.ne = self.weights.graph(sh.edges if len(sh.edges) else None)
        self.tgt = ctx.to_dev(target)
        if P0 is None:
            P0 = _nicp.initial_params(self.n_global, self.dim)
        self.P = ctx.to_dev(sh.local_params(P0))
        self._torch = torch

    # extended energy (include/icp_hip.h nicp_set_weights / nicp_set_landmarks), global arrays in
    def set_weights(self, fit_w=None, rigid_w=None, edge_w=None):
        f, g, e = self.shard.local_weights(fit_w, rigid_w, edge_w)
        self.weights.set_weights(self.shard.edges if len(self.shard.edges) else None, f, g, e)

    def set_landmarks(self, vertex=None, pos=None, c_land=0.0):
        sh, comm = self.shard, self.comm
        edges = sh.edges if len(sh.edges) else None
        if vertex is None or len(vertex) == 0:
            self.weights.set_landmarks(edges)
            mask = sh.row_mask
        else:
            sel, lv, mask = sh.local_landmarks(vertex)
            self.weights.set_landmarks(edges, lv, np.asarray(pos, dtype=np.float64).reshape(-1, self.dim)[sel], c_land)
        comm.row_mask = self._torch.from_numpy(np.ascontiguousarray(mask)).to(comm.reduce_buf.device)
        comm.struct.row_mask_dev = ctypes.c_void_p(comm.row_mask.data_ptr())

    def deform(self):
        out = self.ctx.empty(tuple(self.v.shape))
        self.ctx.call("nicp_deform", self.graph, _lib.ptr(self.P), _lib.ptr(self.v), _lib.ptr(out))
        return out

    def solve(self, nn, c, opts=None):
        o = None
        if opts:
            o = _lib.SolveOpts(1.49012e-8, 1.49012e-8, 0.0, 1e-3, 0.0, 0, 0, 0)
            for k, val in opts.items():
                setattr(o, k, val)
        st = _lib.SolveStats()
        c3 = (ctypes.c_double * 3)(*[float(x) for x in c])
        try:
            self.ctx.call("nicp_solve_dist", self.graph, _lib.ptr(self.P), _lib.ptr(self.v), _lib.ptr(nn), c3, 0,
                          ctypes.byref(o) if o is not None else None, ctypes.byref(st), ctypes.byref(self.comm.struct))
        except _lib.IcpHipError:
            if self.comm.error is not None:
                raise self.comm.error
            raise
        return dict(err=st.err, err0=st.err0, lm_iters=st.lm_iters, cg_iters=st.cg_iters, nfev=st.nfev, info=st.info,
                    graph_launches=st.graph_launches)

    def match(self):
        """Deform -> Match for the local rows against the replicated target."""
        from . import rigid
        dd = self.deform()
        idx = rigid.closest_indices(dd, self.tgt, ctx=self.ctx)
        return rigid.gather_points(self.tgt, idx, ctx=self.ctx)

    def run(self, tol, run, c, opts=None, history=None):
        """Outer loop of nicp_3d.py:344-363 on the sharded mesh (every rank takes the same
        branches: err is the all-reduced energy).  Returns the final (c_fit, c_rigid, c_smooth)."""
        c = [float(x) for x in c]
        err = 100000000
        for _ in range(run):
            e = err
            nn = self.match()
            st = self.solve(nn, c, opts)
            err = st["err"]
            if history is not None:
                history.append((err, c[1], c[2], st))
            if (e - err) < tol:
                c[1] /= 2.0
                c[2] /= 2.0
                if c[1] < 0.1 or c[2] < 0.1:
                    break
        return tuple(c)

    def owned_params(self):
        """(global ids, [n_owned, S] parameters, tail) of this rank, on the host."""
        S = self.dim * self.dim + self.dim
        P = self.P.cpu().numpy()
        nl = self.shard.n_owned + self.shard.n_halo
        return self.shard.owned, P[:self.shard.n_owned * S].reshape(-1, S), P[nl * S:]

    def owned_deformed(self):
        return self.shard.owned, self.deform()[:self.shard.n_owned].cpu().numpy()
