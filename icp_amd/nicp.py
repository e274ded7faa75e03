"""Device-resident NICP operators over the C-ABI (include/icp_hip.h, nicp_* entry points).

`SparseWeights` stands in for the reference's dense N x N influence matrix `w`
(nicp_3d.py:158: exactly 4 non-zeros per column): it keeps nbr[N,4] / wt[N,4] on the
device, still answers `w[j, i]`, `.sum(0)` and `.toarray()` for small N, and owns the
mesh-graph handle the Jacobian products need.
"""
import ctypes
import hashlib

import numpy as np

from . import _lib, rigid
from ._lib import get_context, ptr

NUM_NN = 4


class SparseWeights:
    def __init__(self, nbr, wt, dim, ctx=None):
        self.ctx = ctx or get_context()
        self.nbr = self.ctx.to_dev(nbr, self.ctx.torch.int32)
        self.wt = self.ctx.to_dev(wt)
        self.dim = dim
        self.shape = (self.nbr.shape[0], self.nbr.shape[0])
        self._graphs = {}

    @classmethod
    def from_dense(cls, w, dim, ctx=None):
        """Adopt a dense reference-style w (4 non-zeros per column)."""
        w = np.asarray(w)
        n = w.shape[0]
        nbr = np.zeros((n, NUM_NN), np.int32)
        wt = np.zeros((n, NUM_NN))
        for j in range(n):
            nz = np.nonzero(w[:, j])[0]
            if len(nz) > NUM_NN:
                raise ValueError("column %d of w has %d non-zeros (> %d)" % (j, len(nz), NUM_NN))
            nbr[j, :len(nz)] = nz
            wt[j, :len(nz)] = w[nz, j]
        return cls(nbr, wt, dim, ctx)

    # -- dense-matrix conveniences (small N only) -------------------------
    def toarray(self):
        n = self.shape[0]
        if n > 8192:
            raise MemoryError("dense w for N=%d would need %.1f GB" % (n, n * n * 8 / 1e9))
        nbr, wt = self.nbr.cpu().numpy(), self.wt.cpu().numpy()
        w = np.zeros((n, n))
        for k in range(nbr.shape[1]):
            np.add.at(w, (nbr[:, k], np.arange(n)), wt[:, k])
        return w

    def __getitem__(self, ji):
        j, i = ji
        nbr, wt = self.nbr[i].cpu().numpy(), self.wt[i].cpu().numpy()
        return float(wt[nbr == j].sum())

    def sum(self, axis=0):
        if axis == 0:
            return self.wt.sum(1).cpu().numpy()
        return self.toarray().sum(axis)

    # -- graph handle -----------------------------------------------------
    def graph(self, edges=None):
        """nicp_graph for these weights + edge list.  numpy edge lists are cached by content.  A device address
        alone is no identity (the caching allocator hands it to the next tensor; tensors are edited in place), so
        tensors are cached by (address, shape, version counter) and the entry HOLDS the tensor: its storage cannot
        be recycled while the entry lives, and an in-place edit bumps the version and builds a new graph."""
        ctx = self.ctx
        keep = None
        if edges is None:
            e_dev, key = None, None
        elif hasattr(edges, "is_cuda"):
            keep = edges
            key = ("tensor", edges.data_ptr(), tuple(edges.shape), str(edges.dtype), str(edges.device), edges._version)
            e_dev = None
        else:
            edges = np.ascontiguousarray(np.asarray(edges).astype(np.int32))   # Q11: the reference holds floats
            key = ("numpy", edges.shape, hashlib.sha1(edges.tobytes()).digest())
            e_dev = None
        if key not in self._graphs:
            if edges is not None:
                e_dev = ctx.to_dev(edges, ctx.torch.int32)
            h = ctypes.c_void_p()
            ne = 0 if e_dev is None else e_dev.shape[0]
            ctx.call("nicp_graph_create", ptr(self.nbr), ptr(self.wt), self.nbr.shape[0], self.nbr.shape[1],
                     ptr(e_dev), ne, self.dim, ctypes.byref(h))
            self._graphs[key] = (h, ne, keep)
        return self._graphs[key][:2]

    # -- extended energy (not in the reference) -----------
    def set_weights(self, edges, fit_w=None, rigid_w=None, edge_w=None):
        """Per-vertex multipliers of c_fit / c_rigid and per-edge multipliers of c_smooth ("per-edge
        stiffness") on the graph of these weights + `edges`; None = 1 everywhere."""
        ctx = self.ctx
        g, ne = self.graph(edges)
        n = self.shape[0]
        arrs = []
        for a, cnt in ((fit_w, n), (rigid_w, n), (edge_w, ne)):
            if a is None:
                arrs.append(None)
                continue
            t = ctx.to_dev(a)
            if tuple(t.shape) != (cnt,):
                raise ValueError("weight array of shape %s, expected (%d,)" % (tuple(t.shape), cnt))
            if bool((t < 0).any()) or not bool(ctx.torch.isfinite(t).all()):
                raise ValueError("row multipliers must be finite and non-negative")
            arrs.append(t)
        ctx.call("nicp_set_weights", g, ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]))

    def set_landmarks(self, edges, vertex=None, pos=None, c_land=0.0):
        """Landmark rows c_land * (pos[k] - Deform(P)[vertex[k]]) appended to the residual; vertex=None removes them."""
        ctx = self.ctx
        g, _ = self.graph(edges)
        if vertex is None or len(vertex) == 0:
            ctx.call("nicp_set_landmarks", g, None, None, 0, 0.0)
            return
        vt = ctx.to_dev(vertex, ctx.torch.int32)
        ps = ctx.to_dev(pos)
        if tuple(ps.shape) != (vt.shape[0], self.dim):
            raise ValueError("landmark positions must be [%d,%d]" % (vt.shape[0], self.dim))
        ctx.call("nicp_set_landmarks", g, ptr(vt), ptr(ps), vt.shape[0], float(c_land))

    def rows(self, edges):
        """length of the residual vector of the graph (landmark rows included)"""
        g, _ = self.graph(edges)
        out = ctypes.c_int64()
        _lib.check(self.ctx.lib.nicp_rows(g, ctypes.byref(out)))
        return out.value

    def __del__(self):
        try:
            for h, _, _ in self._graphs.values():
                self.ctx.lib.nicp_graph_destroy(self.ctx.handle, h)
            self._graphs = {}
        except Exception:
            pass


def knn(dset, queries, k, ctx=None):
    """(idx[nq,k] int32, dist[nq,k]) device tensors; k <= 8."""
    ctx = ctx or get_context()
    v = ctx.to_dev(dset)
    q = ctx.to_dev(queries)
    if k > 8:
        raise ValueError("KNN on the GPU supports k <= 8 (the reference uses num_nn+2 = 6)")
    kk = 6 if k <= 6 else 8
    if v.shape[0] < 1:
        raise ValueError("KNN over an empty point set")
    idx = ctx.empty((q.shape[0], kk), ctx.torch.int32)
    dist = ctx.empty((q.shape[0], kk))
    ctx.call("nicp_knn", ptr(v), v.shape[0], v.shape[1], ptr(q), q.shape[0], kk, ptr(idx), ptr(dist))
    k = min(k, v.shape[0])      # fewer points than k: np.argsort(...)[0:k] returns them all (nicp_3d.py:56)
    return idx[:, :k].contiguous(), dist[:, :k].contiguous()


def knn_weights(source, ctx=None):
    ctx = ctx or get_context()
    v = ctx.to_dev(source)
    n, dim = v.shape
    nbr = ctx.empty((n, NUM_NN), ctx.torch.int32)
    wt = ctx.empty((n, NUM_NN))
    ctx.call("nicp_knn_weights", ptr(v), n, dim, NUM_NN, ptr(nbr), ptr(wt))
    return SparseWeights(nbr, wt, dim, ctx)


def initial_params(n, dim, ctx=None):
    """nicp_3d.py:154-156 / nicp_row.py:98 / nicp3d.cpp:186-192 (without the debug Tz, Q9)."""
    if dim == 3:
        blk = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=np.float64)
        return np.append(np.tile(blk, n), np.zeros(6))
    return np.append(np.tile(np.array([1, 0, 0, 1, 0, 0.0]), n), np.zeros(3))


def _c3(c):
    return (ctypes.c_double * 3)(*[float(x) for x in c])


def _as_weights(w, dim, ctx):
    return w if isinstance(w, SparseWeights) else SparseWeights.from_dense(w, dim, ctx)


def deform(P, w, source, ctx=None):
    ctx = ctx or get_context()
    v = ctx.to_dev(source)
    w = _as_weights(w, v.shape[1], ctx)
    g, _ = w.graph(None)
    Pd = ctx.to_dev(P)
    out = ctx.empty(tuple(v.shape))
    ctx.call("nicp_deform", g, ptr(Pd), ptr(v), ptr(out))
    return out


def n_rows(n, ne, dim):
    return dim * n + (6 if dim == 3 else 3) * n + 2 * dim * ne


def residual(P, w, source, edges, nn, c, compat=True, ctx=None):
    ctx = ctx or get_context()
    v = ctx.to_dev(source)
    w = _as_weights(w, v.shape[1], ctx)
    g, ne = w.graph(edges)
    Pd, nnd = ctx.to_dev(P), ctx.to_dev(nn)
    r = ctx.empty((w.rows(edges),))
    ctx.call("nicp_residual", g, ptr(Pd), ptr(v), ptr(nnd), _c3(c), 1 if compat else 0, ptr(r))
    return r


def jvp(P, w, source, edges, c, p, compat=True, ctx=None):
    ctx = ctx or get_context()
    v = ctx.to_dev(source)
    w = _as_weights(w, v.shape[1], ctx)
    g, ne = w.graph(edges)
    Pd, pd = ctx.to_dev(P), ctx.to_dev(p)
    y = ctx.empty((w.rows(edges),))
    ctx.call("nicp_jvp", g, ptr(Pd), ptr(v), _c3(c), 1 if compat else 0, ptr(pd), ptr(y))
    return y


def jtvp(P, w, source, edges, c, u, compat=True, ctx=None):
    ctx = ctx or get_context()
    v = ctx.to_dev(source)
    w = _as_weights(w, v.shape[1], ctx)
    g, ne = w.graph(edges)
    Pd, ud = ctx.to_dev(P), ctx.to_dev(u)
    out = ctx.empty((Pd.shape[0],))
    ctx.call("nicp_jtvp", g, ptr(Pd), ptr(v), _c3(c), 1 if compat else 0, ptr(ud), ptr(out))
    return out


def dense_jacobian(P, w, source, edges, c, compat=True, ctx=None):
    """D_MinFunc's dense matrix, column by column through J e_k (small meshes only)."""
    ctx = ctx or get_context()
    v = ctx.to_dev(source)
    Pd = ctx.to_dev(P)
    npar = Pd.shape[0]
    if npar > 20000:
        raise MemoryError("dense Jacobian with %d columns refused; use jvp/jtvp" % npar)
    w = _as_weights(w, v.shape[1], ctx)
    cols = []
    e = ctx.torch.zeros(npar, dtype=ctx.torch.float64, device=Pd.device)
    for k in range(npar):
        e[k] = 1.0
        cols.append(jvp(Pd, w, v, edges, c, e, compat, ctx))
        e[k] = 0.0
    return ctx.torch.stack(cols, 1)


def solve(P, w, source, edges, nn, c, compat=True, opts=None, ctx=None):
    """The leastsq call of nicp_3d.py:351.  Returns (P_new device tensor, stats dict)."""
    ctx = ctx or get_context()
    v = ctx.to_dev(source)
    w = _as_weights(w, v.shape[1], ctx)
    g, ne = w.graph(edges)
    Pd = ctx.to_dev(P).clone()
    nnd = ctx.to_dev(nn)
    o = None
    if opts:
        o = _lib.SolveOpts(1.49012e-8, 1.49012e-8, 0.0, 1e-3, 0.0, 0, 0, 0)
        for k, val in opts.items():
            setattr(o, k, val)
    st = _lib.SolveStats()
    ctx.call("nicp_solve", g, ptr(Pd), ptr(v), ptr(nnd), _c3(c), 1 if compat else 0,
             ctypes.byref(o) if o is not None else None, ctypes.byref(st))
    return Pd, dict(err=st.err, err0=st.err0, lm_iters=st.lm_iters, cg_iters=st.cg_iters, nfev=st.nfev, info=st.info,
                    graph_launches=st.graph_launches)


CG_KERNELS = ("fitrow_kernel", "normal_kernel", "cg_sum_alpha_kernel", "cg_update_prec_kernel", "cg_sum_coarse_kernel",
              "cg_dir_prec_kernel")


def profile_cg(P, w, source, edges, nn, c, compat=False, iterations=100, ctx=None):
    """HIP-event timing of the CG iteration of the first linear solve at P (nicp_profile_cg).  Returns a dict with
    us_iteration (the solver's own launch path) and us_kernel {name: microseconds}."""
    ctx = ctx or get_context()
    v = ctx.to_dev(source)
    w = _as_weights(w, v.shape[1], ctx)
    g, ne = w.graph(edges)
    Pd, nnd = ctx.to_dev(P), ctx.to_dev(nn)
    pr = _lib.CgProfile()
    ctx.call("nicp_profile_cg", g, ptr(Pd), ptr(v), ptr(nnd), _c3(c), 1 if compat else 0, int(iterations), ctypes.byref(pr))
    return dict(iterations=pr.iterations, launches_per_iteration=pr.launches_per_iteration, graph_replay=bool(pr.graph_replay),
                coarse_columns=pr.coarse_columns, vertices=pr.vertices, edges=pr.edges, us_iteration=pr.us_iteration,
                us_kernel={k: pr.us_kernel[i] for i, k in enumerate(CG_KERNELS)})


def nicp(P, w, source, edges, target, tol, run, c, compat=True, opts=None, history=None, ctx=None):
    """Outer loop of nicp_3d.py:344-363 with the stiffness-halving schedule.
    Returns (deformed, P, (c_fit, c_rigid, c_smooth))."""
    ctx = ctx or get_context()
    v = ctx.to_dev(source)
    tgt = ctx.to_dev(target)
    w = _as_weights(w, v.shape[1], ctx)
    c = [float(x) for x in c]
    Pd = ctx.to_dev(P)
    err = 100000000
    for i in range(run):
        e = err
        dd = deform(Pd, w, v, ctx)
        idx = rigid.closest_indices(dd, tgt, ctx=ctx)
        nn = rigid.gather_points(tgt, idx, ctx=ctx)
        Pd, st = solve(Pd, w, v, edges, nn, c, compat, opts, ctx)
        err = st["err"]
        if history is not None:
            history.append((err, c[1], c[2], st))
        if (e - err) < tol:
            c[1] /= 2.0
            c[2] /= 2.0
            if c[1] < 0.1 or c[2] < 0.1:
                break
    return deform(Pd, w, v, ctx), Pd, tuple(c)
