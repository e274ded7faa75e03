"""Sparse / vectorised twin of the NICP part of icp_oracle.py  --  TEST INFRASTRUCTURE ONLY.

icp_oracle.py restates MinFunc / D_MinFunc (nicp_3d.py:239-338, nicp_row.py:132-230) the way the
reference writes them: Python loops and a DENSE Jacobian, (9N+6E) x (12N+6) doubles -- 260 GB at the
10 000-vertex mesh of BASELINE config C3, so no oracle number existed for the
solver path that C3 / C5 actually run.  This file holds the same arithmetic as numpy index expressions
and a scipy.sparse CSR Jacobian, so the checker reaches those sizes:

  * deform_v / min_func_v / d_min_func_sparse  -- same values as orc.deform / min_func / d_min_func
    (tests/test_oracle_sparse_cpu.py holds them to the loop versions and to the reference-generated
    fixtures at 1e-12, both modes, 3-D and 2-D);
  * lm_sparse -- orc.lm_dense with a sparse LU of (J^T J + lam D^2): the dense-quality Levenberg-
    Marquardt step the GPU's CG solve is compared with at 2 500 and 10 000 vertices.

Only tests/ may import this module; the product (icp_amd) never does.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import icp_oracle as orc


def _split(P, n, dim):
    s = dim * dim + dim
    Pm = P[:n * s].reshape(n, s)
    return Pm[:, :dim * dim].reshape(n, dim, dim), Pm[:, dim * dim:]


def deform_local_v(P, nbr, wt, source):
    """orc.deform_local (nicp_3d.py:224-228): sum_j w[j,i] ((v_i - v_j) A_j + v_j + b_j), neighbours in
    ascending j like the reference's inner loop."""
    source = np.asarray(source, dtype=np.float64)
    n, dim = source.shape
    A, b = _split(P, n, dim)
    order = np.argsort(nbr, axis=1, kind="stable")
    rows = np.arange(n)
    out = np.zeros_like(source)
    for kk in range(nbr.shape[1]):
        k = order[:, kk]
        j = nbr[rows, k]
        w = wt[rows, k]
        term = np.einsum("nr,nrc->nc", source - source[j], A[j]) + source[j] + b[j]
        out += w[:, None] * term
    return out


def deform_v(P, nbr, wt, source):
    """orc.deform (nicp_3d.py:220-233)."""
    source = np.asarray(source, dtype=np.float64)
    R, T = orc._globals(P, source.shape[1])
    return deform_local_v(P, nbr, wt, source) @ R + T


def min_func_v(P, nbr, wt, srcdata, edges, tgtdata, c=(orc.C_FIT, orc.C_RIGID, orc.C_SMOOTH), compat=True):
    """orc.min_func (nicp_3d.py:239-267 / nicp_row.py:132-160), incl. the slot layout of quirk Q5."""
    srcdata = np.asarray(srcdata, dtype=np.float64)
    n, dim = srcdata.shape
    A, b = _split(P, n, dim)
    e_fit = tgtdata - deform_v(P, nbr, wt, srcdata)
    if dim == 3:
        a0, a1, a2 = A[:, 0], A[:, 1], A[:, 2]
        six = np.stack([(a0 * a1).sum(1), (a1 * a2).sum(1), (a2 * a0).sum(1),
                        1 - (a0 * a0).sum(1), 1 - (a1 * a1).sum(1), 1 - (a2 * a2).sum(1)], 1)
        e_rigid = np.zeros(n * 6)
        if compat:
            # vertex i writes slots 3i..3i+5; vertex i+1 then overwrites 3i+3..3i+5 (nicp_3d.py:245-253)
            e_rigid[:3 * n] = six[:, :3].reshape(-1)
            e_rigid[3 * n:3 * n + 3] = six[n - 1, 3:]
        else:
            e_rigid[:] = six.reshape(-1)
    else:
        a0, a1 = A[:, 0], A[:, 1]
        e_rigid = np.stack([(a0 * a1).sum(1), 1 - (a0 * a0).sum(1), 1 - (a1 * a1).sum(1)], 1).reshape(-1)
    p1, p2 = edges[:, 0].astype(np.int64), edges[:, 1].astype(np.int64)
    d12 = srcdata[p2] - srcdata[p1]
    s1 = np.einsum("nr,nrc->nc", d12, A[p1]) + srcdata[p1] + b[p1] - (srcdata[p2] + b[p2])
    s2 = np.einsum("nr,nrc->nc", -d12, A[p2]) + srcdata[p2] + b[p2] - (srcdata[p1] + b[p1])
    e_smooth = np.stack([s1, s2], 1).reshape(-1)
    return np.concatenate([(e_fit * c[0]).reshape(-1), e_rigid * c[1], e_smooth * c[2]])


def d_min_func_sparse(P, nbr, wt, srcdata, edges, tgtdata=None, c=(orc.C_FIT, orc.C_RIGID, orc.C_SMOOTH), compat=True):
    """orc.d_min_func (nicp_3d.py:274-338 / nicp_row.py:167-230) as a scipy.sparse CSR matrix: same rows,
    same columns, same values, quirks Q6 / Q7 in compat mode."""
    srcdata = np.asarray(srcdata, dtype=np.float64)
    n, dim = srcdata.shape
    ne = edges.shape[0]
    s = dim * dim + dim
    nP = P.shape[0]
    nrig = 6 if dim == 3 else 3
    K = nbr.shape[1]
    A, _ = _split(P, n, dim)
    R, T = orc._globals(P, dim)
    pre = deform_local_v(P, nbr, wt, srcdata)
    rot_src = pre @ R + T if compat else pre
    rows, cols, vals = [], [], []

    def add(r, cidx, v):
        r, cidx, v = np.broadcast_arrays(r, cidx, v)
        rows.append(r.reshape(-1))
        cols.append(cidx.reshape(-1))
        vals.append(v.reshape(-1).astype(np.float64))

    ii = np.arange(n)
    # ---- fit rows (nicp_3d.py:286-306): row i*dim + c
    for k in range(K):
        j = nbr[:, k].astype(np.int64)
        w = wt[:, k]
        keep = w != 0
        disp = srcdata - srcdata[j]
        for r in range(dim):
            for cc in range(dim):
                for co in range(dim):
                    add((ii * dim + co)[keep], (j * s + r * dim + cc)[keep], (-w * disp[:, r] * R[cc, co] * c[0])[keep])
        for cc in range(dim):
            wb = np.ones(n) if (dim == 2 and compat) else w          # Q7
            for co in range(dim):
                add((ii * dim + co)[keep], (j * s + dim * dim + cc)[keep], (-wb * R[cc, co] * c[0])[keep])
    if dim == 3:
        for a, dR in enumerate(orc.d_rot_matrix(P[-6], P[-5], P[-4])):
            g = -(rot_src @ dR) * c[0]
            for co in range(3):
                add(ii * 3 + co, nP - 6 + a, g[:, co])
    else:
        g = -(rot_src @ orc.d_rot2(P[-3])) * c[0]
        for co in range(2):
            add(ii * 2 + co, nP - 3, g[:, co])
    for cc in range(dim):
        add(ii * dim + cc, nP - dim + cc, np.full(n, -1.0 * c[0]))
    # ---- rigidity rows (nicp_3d.py:309-314): row n*dim + i*nrig + q, always at offset nrig*i
    r0 = n * dim + ii * nrig
    o = ii * s
    if dim == 3:
        a0, a1, a2 = A[:, 0], A[:, 1], A[:, 2]
        blocks = [(0, [a1, a0, None]), (1, [None, a2, a1]), (2, [a2, None, a0]),
                  (3, [-2 * a0, None, None]), (4, [None, -2 * a1, None]), (5, [None, None, -2 * a2])]
        for q, parts in blocks:
            for blk, arr in enumerate(parts):
                if arr is None:
                    continue
                for t in range(3):
                    add(r0 + q, o + blk * 3 + t, arr[:, t] * c[1])
    else:
        add(r0, o + 0, A[:, 1, 0] * c[1]); add(r0, o + 1, A[:, 1, 1] * c[1])
        add(r0, o + 2, A[:, 0, 0] * c[1]); add(r0, o + 3, A[:, 0, 1] * c[1])
        add(r0 + 1, o + 0, -2 * A[:, 0, 0] * c[1]); add(r0 + 1, o + 1, -2 * A[:, 0, 1] * c[1])
        add(r0 + 2, o + 2, -2 * A[:, 1, 0] * c[1]); add(r0 + 2, o + 3, -2 * A[:, 1, 1] * c[1])
    # ---- smoothness rows (nicp_3d.py:316-333)
    p1, p2 = edges[:, 0].astype(np.int64), edges[:, 1].astype(np.int64)
    disp = srcdata[p2] - srcdata[p1]
    e0 = n * dim + n * nrig + np.arange(ne) * 2 * dim
    one = np.full(ne, c[2])
    for cc in range(dim):
        for r in range(dim):
            add(e0 + cc, p1 * s + r * dim + cc, disp[:, r] * c[2])
            add(e0 + dim + cc, p2 * s + r * dim + cc, -disp[:, r] * c[2])
        add(e0 + cc, p1 * s + dim * dim + cc, one)
        add(e0 + cc, p2 * s + dim * dim + cc, -one)
        add(e0 + dim + cc, p2 * s + dim * dim + cc, one)
        add(e0 + dim + cc, p1 * s + dim * dim + cc, -one)
    nrows = n * dim + n * nrig + ne * 2 * dim
    J = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nrows, nP))
    return J.tocsr()


def nd_order(adj, pts, leaf=48):
    """Nested-dissection ordering of a vertex graph by recursive coordinate bisection: [left | right | separator],
    the separator being the vertices of the left half that touch the right half.  Keeps the fill of the sparse
    factorisation near that of a planar mesh (SuperLU's own orderings choke on the 12 x 12 vertex blocks)."""
    n = adj.shape[0]
    adj = adj.tocsr()
    mark = np.zeros(n)
    out = []

    def rec(verts):
        if len(verts) <= leaf:
            out.append(verts)
            return
        ext = pts[verts].max(0) - pts[verts].min(0)
        a = int(np.argmax(ext))
        lm = pts[verts, a] <= np.median(pts[verts, a])
        left, right = verts[lm], verts[~lm]
        if len(left) == 0 or len(right) == 0:
            out.append(verts)
            return
        mark[right] = 1.0
        touch = adj[left] @ mark > 0
        mark[right] = 0.0
        rec(left[~touch])
        rec(right)
        out.append(left[touch])

    rec(np.arange(n))
    return np.concatenate(out)


class NormalSolver:
    """Direct solves of (J^T J + lam diag(d2)) x = rhs for one Jacobian J and several lam.

    The columns of the global parameters (the last 6, or 3 in 2-D) are dense, so they are taken out by a Schur
    complement; the vertex part is factorised by SuperLU in a nested-dissection order of the mesh (pts = vertex
    coordinates); three rounds of iterative refinement form the residual through J itself (J^T (J x)), i.e. with
    the operator the GPU applies, not with the rounded product matrix."""

    def __init__(self, J, pts):
        n, dim = pts.shape
        self.s = s = dim * dim + dim
        self.nv = nv = n * s
        self.J = J
        self.Jv = J[:, :nv].tocsc()
        self.Jg = J[:, nv:].toarray()
        self.Hvv = (self.Jv.T @ self.Jv).tocsr()
        self.Hvg = self.Jv.T @ self.Jg
        self.Hgg = self.Jg.T @ self.Jg
        coo = self.Hvv.tocoo()
        adj = sp.coo_matrix((np.ones(coo.nnz), (coo.row // s, coo.col // s)), shape=(n, n)).tocsr()
        order = nd_order(adj, np.asarray(pts, dtype=np.float64))
        self.perm = (order[:, None] * s + np.arange(s)[None, :]).reshape(-1)

    def solve(self, lam, d2, rhs):
        nv, perm = self.nv, self.perm
        key = (float(lam), hash(d2.tobytes()))
        if getattr(self, "_key", None) != key:      # (a second right-hand side under the same damping reuses the factors)
            M = (self.Hvv + sp.diags(lam * d2[:nv])).tocsc()[perm][:, perm]
            lu = spla.splu(M.tocsc(), permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
            Y = np.empty((nv, self.Jg.shape[1]))
            Y[perm] = lu.solve(self.Hvg[perm])
            self._key, self._lu, self._Y = key, lu, Y
        lu, Y = self._lu, self._Y
        S = self.Hgg + np.diag(lam * d2[nv:]) - self.Hvg.T @ Y

        def once(b):
            y = np.empty(nv)
            y[perm] = lu.solve(b[:nv][perm])
            xg = np.linalg.solve(S, b[nv:] - self.Hvg.T @ y)
            return np.concatenate([y - Y @ xg, xg])

        x = once(rhs)
        for _ in range(3):
            x = x + once(rhs - (self.J.T @ (self.J @ x) + lam * d2 * x))
        self.last_residual = float(np.linalg.norm(rhs - (self.J.T @ (self.J @ x) + lam * d2 * x)) / np.linalg.norm(rhs))
        return x


def lm_sparse(P0, nbr, wt, source, edges, nn, c=(orc.C_FIT, orc.C_RIGID, orc.C_SMOOTH), compat=True, steps=1,
              lambda0=1e-3, ftol=1.49012e-8, xtol=1.49012e-8, trace=None, accel=False):
    """orc.lm_dense -- the dense twin of the PRODUCT's Levenberg-Marquardt loop on the oracle's pinned
    (MinFunc, D_MinFunc) pair -- with the sparse Jacobian and a sparse LU in place of numpy.linalg.solve.  Same
    damping rules, acceptance test and stopping tests, statement for statement."""
    x = np.array(P0, dtype=np.float64)
    f = lambda p: min_func_v(p, nbr, wt, source, edges, nn, c, compat)
    Jf = lambda p: d_min_func_sparse(p, nbr, wt, source, edges, nn, c, compat)
    r = f(x)
    f2 = float(r @ r)
    out = dict(err0=f2, nfev=1, lm_iters=0, info=0, lams=[])
    lam, nu = lambda0, 2.0
    d2 = None
    for it in range(steps):
        out["lm_iters"] = it + 1
        J = Jf(x)
        grad = J.T @ r
        col2 = np.asarray(J.multiply(J).sum(0)).reshape(-1)
        d2 = np.where(col2 > 0, col2, 1.0) if d2 is None else np.maximum(d2, col2)
        solver = NormalSolver(J, np.asarray(source, dtype=np.float64))
        accepted = False
        while not accepted and not out["info"]:
            delta = solver.solve(lam, d2, -grad)
            dmodel = delta
            if accel:   # the product's accelerated iteration, as in orc.lm_dense
                fvv = f(x + delta) - 2.0 * r + f(x - delta)
                out["nfev"] += 2
                delta2 = solver.solve(lam, d2, -0.5 * (J.T @ fvv))
                if float((delta2 * delta2 * d2).sum()) <= 0.5625 * float((delta * delta * d2).sum()):
                    delta = delta + delta2
            xt = x + delta
            rt = f(xt)
            f2t = float(rt @ rt)
            out["nfev"] += 1
            dn2, dg, xn2 = float((dmodel * dmodel * d2).sum()), float(dmodel @ grad), float((x * x * d2).sum())
            pred, actred = lam * dn2 - dg, f2 - f2t
            ratio = actred / pred if pred > 0 else (1.0 if actred > 0 else -1.0)
            if trace is not None:
                trace.append(dict(it=it, lam=lam, f2=f2, f2t=f2t, ratio=ratio, delta=delta.copy(), x=x.copy(),
                                  solve_residual=solver.last_residual))
            if np.isfinite(f2t) and ratio > 1e-4:
                out["lams"].append(lam)
                x, r, accepted = xt, rt, True
                t = 2 * ratio - 1
                lam *= max(1.0 / 3.0, 1 - t * t * t)
                nu = 2.0
                t_f = abs(actred) <= ftol * f2 and pred <= ftol * f2 and 0.5 * ratio <= 1
                t_x = np.sqrt(dn2) <= xtol * np.sqrt(xn2)
                f2 = f2t
                if t_f:
                    out["info"] = 1
                if t_x:
                    out["info"] = 3 if out["info"] == 1 else 2
            else:
                lam *= nu
                nu *= 2.0
                if np.sqrt(dn2) <= xtol * np.sqrt(xn2) or lam > 1e30:
                    out["info"] = 2
        if out["info"]:
            break
    if not out["info"]:
        out["info"] = 5
    out["err"] = f2
    return x, out
