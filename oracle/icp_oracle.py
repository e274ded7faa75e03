"""CPU oracle for the ICP / NICP hot path  --  TEST INFRASTRUCTURE ONLY.

This module restates, in Python 3 / numpy, the arithmetic of the reference
(drawpixels/icp) hot-path functions.  It is the *checker* for the HIP kernels:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product (``icp_amd``) never does.

Parity status: the reference ships no tests / golden vectors,
so this oracle is pinned against the reference's own function bodies executed in
the build container (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``); see
``tests/test_oracle_golden.py``.

Each function cites the reference file:line it follows (paths relative to
/root/reference).  Where the reference is scalar-looped Python the restatement is
vectorised but keeps the same floating-point evaluation order wherever the result
is required to be bit-exact (squared distances, argmin tie-break).
"""
import numpy as np

# nicp_3d.py:17-20 / nicp_row.py:26-29 / nicp3d.cpp:25-28
NUM_NN = 4
C_FIT = 0.1
C_RIGID = 1000.0
C_SMOOTH = 100.0


# --------------------------------------------------------------------------
# rigid ICP
# --------------------------------------------------------------------------
def sqdist(dset, pt):
    """d2[m] = ((dx*dx)+(dy*dy))+(dz*dz), dx = dset[m]-pt; every op rounded to fp64.

    icp_3d.py:40-41 (``disp = dset - pt; (disp*disp).sum(1)``).  numpy reduces the
    short last axis left to right, which is the order written here explicitly.
    """
    dset = np.asarray(dset, dtype=np.float64)
    pt = np.asarray(pt, dtype=np.float64)
    d = dset - pt
    q = d * d
    acc = q[:, 0].copy()
    for k in range(1, q.shape[1]):
        acc = acc + q[:, k]
    return acc


def closest(dset, pt, compare_sqrt=False):
    """Index of the nearest row of dset; lowest index on ties.

    icp_3d.py:39-42 (np.argmin of squared distances).  ``compare_sqrt`` gives the
    C++ twin's semantics, nicp3d.cpp:56-63,142-164: compare the *rounded square
    root* with strict ``<`` scanning from index 0.
    """
    d2 = sqdist(dset, pt)
    if compare_sqrt:
        d2 = np.sqrt(d2)
    return int(np.argmin(d2))


def match_idx(source, target, compare_sqrt=False):
    """Correspondence indices + squared distances for every source row.

    icp_3d.py:48-53 keeps only ``target[c]``; the indices are what the GPU path
    must reproduce bit-exactly, so the oracle exposes them.
    """
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    n = source.shape[0]
    idx = np.empty(n, dtype=np.int32)
    d2o = np.empty(n, dtype=np.float64)
    for i in range(n):
        d2 = sqdist(target, source[i])
        key = np.sqrt(d2) if compare_sqrt else d2
        c = int(np.argmin(key))
        idx[i] = c
        d2o[i] = d2[c]
    return idx, d2o


def match(source, target, compare_sqrt=False):
    """icp_3d.py:48-53: the matched target *points*."""
    idx, _ = match_idx(source, target, compare_sqrt)
    return np.asarray(target, dtype=np.float64)[idx]


def distance(s1, s2):
    """Sum of Euclidean norms of row differences.  icp_3d.py:59-62."""
    d = np.asarray(s1, dtype=np.float64) - np.asarray(s2, dtype=np.float64)
    return np.sqrt((d * d).sum(1)).sum(0)


def find_transform(source, target):
    """Kabsch without scale and without reflection fix.  icp_3d.py:68-90.

    W[a,b] = sum_i t1[i,a]*s1[i,b] accumulated in row order (the reference adds one
    outer product per point, icp_3d.py:77-78); R = (U VT)^T; t = t_mean - s_mean R.
    """
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    s_mean = source.mean(0)
    t_mean = target.mean(0)
    s1 = source - s_mean
    t1 = target - t_mean
    dim = source.shape[1]
    W = np.zeros((dim, dim))
    # sequential accumulation keeps the reference's rounding order
    for a in range(dim):
        for b in range(dim):
            W[a, b] = np.cumsum(t1[:, a] * s1[:, b])[-1] if len(s1) else 0.0
    U, S, VT = np.linalg.svd(W)
    R = (U @ VT).T
    t = t_mean - s_mean @ R
    return R, t


def find_transform_T(source, target):
    """icp_row.py:79-107: same numbers packed into a (dim+1)x(dim+1) row-vector T."""
    R, t = find_transform(source, target)
    dim = R.shape[0]
    T = np.identity(dim + 1)
    T[:dim, :dim] = R
    T[dim, :dim] = t
    return T


def find_transform_col(source, target):
    """icp.py:59-74 column-vector twin: points are columns ([dim,N]); R = U VT,
    t = t_mean - R s_mean (t is a [dim,1] column)."""
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    t_mean = target.mean(1)[:, None]
    s_mean = source.mean(1)[:, None]
    t1 = target - t_mean
    s1 = source - s_mean
    dim = source.shape[0]
    W = np.zeros((dim, dim))
    for a in range(dim):
        for b in range(dim):
            W[a, b] = np.cumsum(t1[a] * s1[b])[-1]
    U, S, VT = np.linalg.svd(W)
    R = U @ VT
    t = t_mean - R @ s_mean
    return R, t


def icp(source, target, tol, run, history=None, use_tol=True):
    """icp_3d.py:102-121.  Absolute transform from the ORIGINAL source each
    iteration (Q1); stop when (e-err) < tol (signed, Q3).  ``use_tol=False`` is the
    icp_row.py:112-143 behaviour (tol ignored, Q4).  ``history`` (a list) receives
    (idx, err) per iteration.
    """
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    dim = source.shape[1]
    ss = source.copy()
    err = 100000000
    R = t = None
    for i in range(run):
        e = err
        idx, _ = match_idx(ss, target)
        nn = target[idx]
        R, t = find_transform(source, nn)
        ss = source @ R + t
        err = distance(nn, ss)
        if history is not None:
            history.append((idx.copy(), float(err)))
        if use_tol and (e - err) < tol:
            break
    if R is None:
        raise NameError("ICP with run=0 leaves R undefined (icp_3d.py:118-120)")
    T = np.identity(dim + 1)
    T[:dim, :dim] = R
    T[dim, :dim] = t
    return T


def icp_col_step(source, target):
    """One round of the icp.py:52-98 loop (column-major, incremental): returns
    (nn, R, t, moved_source); the caller overwrites source with moved_source."""
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    idx, _ = match_idx(source.T.copy(), target.T.copy())
    nn = target[:, idx]
    R, t = find_transform_col(source, nn)
    return nn, R, t, R @ source + t


# --------------------------------------------------------------------------
# NICP: KNN weights, deformation, residual, Jacobian
# --------------------------------------------------------------------------
def knn(dset, pt, k, cxx_order=False):
    """k smallest Euclidean distances (with sqrt) and their indices, ascending.

    nicp_3d.py:53-57: argsort for indices, separate sort for distances.
    ``cxx_order=True`` follows nicp3d.cpp:100-137: bounded max-heap that replaces
    the top only when ``top > d`` -- i.e. among equal distances the EARLIER index
    is kept; emitted ascending.  (Without ties the two agree.)
    """
    dist = np.sqrt(sqdist(dset, pt))
    if cxx_order:
        order = np.argsort(dist, kind="stable")[:k]
        return order.astype(np.int64), dist[order]
    sidx = np.argsort(dist)
    return sidx[:k], np.sort(dist)[:k]


def knn_weights(source, num_nn=NUM_NN, cxx_order=False):
    """Sparse form of the influence weights of nicp_3d.py:158-164 /
    nicp3d.cpp:200-215.

    For vertex j: idx,dist = KNN(source, source[j], num_nn+2);
    denom = num_nn - sum(dist[1:-1])/dist[-1]; w[idx[i], j] = (1-dist[i]/dist[-1])/denom
    for i=1..num_nn.  Returned as nbr[j,i-1] = idx[i], wt[j,i-1] = that weight, i.e.
    column j of the reference's dense w.
    """
    source = np.asarray(source, dtype=np.float64)
    n = source.shape[0]
    nbr = np.empty((n, num_nn), dtype=np.int32)
    wt = np.empty((n, num_nn), dtype=np.float64)
    for j in range(n):
        idx, dist = knn(source, source[j], num_nn + 2, cxx_order)
        if cxx_order:
            # nicp3d.cpp:208-214: distSum accumulated in a scalar loop
            s = 0.0
            for i in range(1, num_nn + 1):
                s += dist[i]
            denom = float(num_nn) - s / dist[num_nn + 1]
        else:
            denom = num_nn - dist[1:-1].sum() / dist[-1]
        for i in range(1, num_nn + 1):
            nbr[j, i - 1] = idx[i]
            wt[j, i - 1] = (1 - dist[i] / dist[-1]) / denom
    return nbr, wt


def dense_weights(nbr, wt, n=None):
    """Expand (nbr, wt) into the reference's dense w[N,N] (w[nbr[j,k], j] = wt[j,k])."""
    n = nbr.shape[0] if n is None else n
    w = np.zeros((n, n))
    for j in range(nbr.shape[0]):
        for k in range(nbr.shape[1]):
            w[nbr[j, k], j] = wt[j, k]
    return w


def initialise(source, target=None, dim=None):
    """nicp_3d.py:150-166 (dim=3, 12 params/vertex + 6) and nicp_row.py:94-107
    (dim=2, 6 params/vertex + 3).  Returns (P, nbr, wt)."""
    source = np.asarray(source, dtype=np.float64)
    n, d = source.shape
    if d == 3:
        P = np.append(np.tile(np.append(np.identity(3).flatten(), np.zeros(3)), n), np.zeros(6))
    else:
        P = np.append(np.tile(np.array([1, 0, 0, 1, 0, 0.0]), n), np.zeros(3))
    nbr, wt = knn_weights(source, NUM_NN)
    return P, nbr, wt


def initialise_cxx(vert):
    """nicp3d.cpp:182-224: identity A, b=0, global Tz = 2.0 (debug value, Q9);
    heap-ordered KNN."""
    vert = np.asarray(vert, dtype=np.float64)
    n = vert.shape[0]
    P = np.zeros(12 * n + 6)
    P[0:12 * n:12] = 1.0
    P[4:12 * n:12] = 1.0
    P[8:12 * n:12] = 1.0
    P[-1] = 2.0
    nbr, wt = knn_weights(vert, NUM_NN, cxx_order=True)
    return P, nbr, wt


def p_to_ab(P, v, dim=3):
    """nicp_3d.py:171-175 / nicp_row.py:118-119."""
    s = dim * dim + dim
    A = P[v * s:v * s + dim * dim].reshape(dim, dim)
    b = P[v * s + dim * dim:v * s + s]
    return A, b


def rot_matrix(x, y, z):
    """nicp_3d.py:180-191 (row-vector rotation, sequence X,Y,Z)."""
    sx, cx = np.sin(x), np.cos(x)
    sy, cy = np.sin(y), np.cos(y)
    sz, cz = np.sin(z), np.cos(z)
    return np.array([[cy * cz, cy * sz, -sy],
                     [sx * sy * cz - cx * sz, sx * sy * sz + cx * cz, sx * cy],
                     [cx * sy * cz + sx * sz, cx * sy * sz - sx * cz, cx * cy]])


def d_rot_matrix(x, y, z):
    """nicp_3d.py:196-215."""
    sx, cx = np.sin(x), np.cos(x)
    sy, cy = np.sin(y), np.cos(y)
    sz, cz = np.sin(z), np.cos(z)
    dx = np.array([[0, 0, 0],
                   [cx * sy * cz + sx * sz, cx * sy * sz - sx * cz, cx * cy],
                   [-sx * sy * cz + cx * sz, -sx * sy * sz - cx * cz, -sx * cy]])
    dy = np.array([[-sy * cz, -sy * sz, -cy],
                   [sx * cy * cz, sx * cy * sz, -sx * sy],
                   [cx * cy * cz, cx * cy * sz, -cx * sy]])
    dz = np.array([[-cy * sz, cy * cz, 0],
                   [-sx * sy * sz - cx * cz, sx * sy * cz - cx * sz, 0],
                   [-cx * sy * sz + sx * cz, cx * sy * cz + sx * sz, 0]])
    return dx, dy, dz


def rot2(r):
    """nicp_row.py:124 / nicp_row_rt.py:76."""
    return np.array([[np.cos(r), np.sin(r)], [-np.sin(r), np.cos(r)]])


def d_rot2(r):
    """nicp_row.py:175."""
    return np.array([[-np.sin(r), np.cos(r)], [-np.cos(r), -np.sin(r)]])


def _globals(P, dim):
    if dim == 3:
        return rot_matrix(P[-6], P[-5], P[-4]), np.array([P[-3], P[-2], P[-1]])
    return rot2(P[-3]), np.array([P[-2], P[-1]])


def deform_local(P, nbr, wt, source):
    """The pre-rotation sum of nicp_3d.py:224-228: for vertex i,
    sum_j w[j,i]*((v_i - v_j) A_j + v_j + b_j), neighbours visited in ascending j
    like the reference's inner loop."""
    source = np.asarray(source, dtype=np.float64)
    n, dim = source.shape
    out = np.zeros_like(source)
    for i in range(n):
        order = np.argsort(nbr[i], kind="stable")
        for k in order:
            j = int(nbr[i, k])
            if wt[i, k] == 0:
                continue
            A, b = p_to_ab(P, j, dim)
            out[i] += wt[i, k] * ((source[i] - source[j]) @ A + source[j] + b)
    return out


def deform(P, nbr, wt, source):
    """nicp_3d.py:220-233 / nicp_row.py:112-126 / nicp3d.cpp:277-371 (live lines)."""
    source = np.asarray(source, dtype=np.float64)
    R, T = _globals(P, source.shape[1])
    return deform_local(P, nbr, wt, source) @ R + T


def deform_rt(dset, rot, tx, ty):
    """nicp_row_rt.py:74-80."""
    return np.asarray(dset, dtype=np.float64) @ rot2(rot) + np.array([tx, ty])


def disp_rt(p, srcdata, tgtdata):
    """nicp_row_rt.py:86-88."""
    return (tgtdata - deform_rt(srcdata, p[0], p[1], p[2])).flatten()


def d_disp_rt(p, srcdata, tgtdata):
    """nicp_row_rt.py:94-112 (live lines)."""
    srcdata = np.asarray(srcdata, dtype=np.float64)
    d_r = -1 * (srcdata @ d_rot2(p[0]))
    J = np.zeros((srcdata.size, 3))
    J[:, 0] = d_r.flatten()
    J[:, 1] = np.tile(np.array([-1.0, 0.0]), srcdata.shape[0])
    J[:, 2] = np.tile(np.array([0.0, -1.0]), srcdata.shape[0])
    return J


def min_func(P, nbr, wt, srcdata, edges, tgtdata, c=(C_FIT, C_RIGID, C_SMOOTH), compat=True):
    """Residual vector.  3-D: nicp_3d.py:239-267; 2-D: nicp_row.py:132-160.

    compat=True reproduces Q5 for dim=3: e_rigid has 6N slots but vertex i writes
    slots 3i..3i+5, so vertex i+1 overwrites the three unit-length terms of vertex i.
    compat=False writes at 6i (the layout D_MinFunc assumes).
    """
    srcdata = np.asarray(srcdata, dtype=np.float64)
    n, dim = srcdata.shape
    ne = edges.shape[0]
    e_fit = tgtdata - deform(P, nbr, wt, srcdata)
    if dim == 3:
        e_rigid = np.zeros(n * 6)
        stride = 3 if compat else 6
        for i in range(n):
            A, _ = p_to_ab(P, i, 3)
            o = i * stride
            e_rigid[o] = A[0, 0] * A[1, 0] + A[0, 1] * A[1, 1] + A[0, 2] * A[1, 2]
            e_rigid[o + 1] = A[1, 0] * A[2, 0] + A[1, 1] * A[2, 1] + A[1, 2] * A[2, 2]
            e_rigid[o + 2] = A[2, 0] * A[0, 0] + A[2, 1] * A[0, 1] + A[2, 2] * A[0, 2]
            e_rigid[o + 3] = 1 - A[0, 0] * A[0, 0] - A[0, 1] * A[0, 1] - A[0, 2] * A[0, 2]
            e_rigid[o + 4] = 1 - A[1, 0] * A[1, 0] - A[1, 1] * A[1, 1] - A[1, 2] * A[1, 2]
            e_rigid[o + 5] = 1 - A[2, 0] * A[2, 0] - A[2, 1] * A[2, 1] - A[2, 2] * A[2, 2]
    else:
        e_rigid = np.zeros(n * 3)
        for i in range(n):
            A, _ = p_to_ab(P, i, 2)
            e_rigid[i * 3] = A[0, 0] * A[1, 0] + A[0, 1] * A[1, 1]
            e_rigid[i * 3 + 1] = 1 - A[0, 0] * A[0, 0] - A[0, 1] * A[0, 1]
            e_rigid[i * 3 + 2] = 1 - A[1, 0] * A[1, 0] - A[1, 1] * A[1, 1]
    e_smooth = np.zeros((ne * 2, dim))
    for i in range(ne):
        p1, p2 = int(edges[i, 0]), int(edges[i, 1])
        A1, b1 = p_to_ab(P, p1, dim)
        A2, b2 = p_to_ab(P, p2, dim)
        e_smooth[i * 2] = (srcdata[p2] - srcdata[p1]) @ A1 + srcdata[p1] + b1 - (srcdata[p2] + b2)
        e_smooth[i * 2 + 1] = (srcdata[p1] - srcdata[p2]) @ A2 + srcdata[p2] + b2 - (srcdata[p1] + b1)
    return np.concatenate([(e_fit * c[0]).flatten(), e_rigid * c[1], (e_smooth * c[2]).flatten()])


def d_min_func(P, nbr, wt, srcdata, edges, tgtdata, c=(C_FIT, C_RIGID, C_SMOOTH), compat=True):
    """Dense Jacobian as coded by the reference (NOT the derivative of min_func in
    compat mode).  3-D: nicp_3d.py:274-338; 2-D: nicp_row.py:167-230.

    compat quirks kept: Q6 (global-rotation columns use the already transformed
    points), Q7 (2-D: b-columns of the fit block carry no weight factor).
    compat=False gives the true derivative of min_func(compat=False).
    """
    srcdata = np.asarray(srcdata, dtype=np.float64)
    n, dim = srcdata.shape
    ne = edges.shape[0]
    s = dim * dim + dim
    nP = P.shape[0]
    ng = 6 if dim == 3 else 3
    R, T = _globals(P, dim)
    pre = deform_local(P, nbr, wt, srcdata)
    dd = pre @ R + T
    rot_src = dd if compat else pre
    J_fit = np.zeros((n * dim, nP))
    for i in range(n):
        for k in range(nbr.shape[1]):
            j = int(nbr[i, k])
            w = wt[i, k]
            if w == 0:
                continue
            disp = srcdata[i] - srcdata[j]
            for r in range(dim):
                for cc in range(dim):
                    # d/dA_j[r,cc]: disp[r] * R[cc,:]
                    J_fit[i * dim:i * dim + dim, j * s + r * dim + cc] = -w * disp[r] * R[cc]
            for cc in range(dim):
                wb = 1.0 if (dim == 2 and compat) else w
                J_fit[i * dim:i * dim + dim, j * s + dim * dim + cc] = -wb * R[cc]
        if dim == 3:
            dRx, dRy, dRz = d_rot_matrix(P[-6], P[-5], P[-4])
            J_fit[i * 3:i * 3 + 3, -6] = -rot_src[i] @ dRx
            J_fit[i * 3:i * 3 + 3, -5] = -rot_src[i] @ dRy
            J_fit[i * 3:i * 3 + 3, -4] = -rot_src[i] @ dRz
        else:
            J_fit[i * 2:i * 2 + 2, -3] = -rot_src[i] @ d_rot2(P[-3])
        for cc in range(dim):
            col = np.zeros(dim)
            col[cc] = -1
            J_fit[i * dim:i * dim + dim, nP - dim + cc] = col
    if dim == 3:
        J_rig = np.zeros((n * 6, nP))
        for i in range(n):
            A, _ = p_to_ab(P, i, 3)
            a0, a1, a2 = A[0], A[1], A[2]
            z = np.zeros(3)
            o = i * 12
            J_rig[i * 6, o:o + 9] = np.concatenate([a1, a0, z])
            J_rig[i * 6 + 1, o:o + 9] = np.concatenate([z, a2, a1])
            J_rig[i * 6 + 2, o:o + 9] = np.concatenate([a2, z, a0])
            J_rig[i * 6 + 3, o:o + 9] = np.concatenate([-2 * a0, z, z])
            J_rig[i * 6 + 4, o:o + 9] = np.concatenate([z, -2 * a1, z])
            J_rig[i * 6 + 5, o:o + 9] = np.concatenate([z, z, -2 * a2])
    else:
        J_rig = np.zeros((n * 3, nP))
        for i in range(n):
            A, _ = p_to_ab(P, i, 2)
            o = i * 6
            J_rig[i * 3, o:o + 4] = [A[1, 0], A[1, 1], A[0, 0], A[0, 1]]
            J_rig[i * 3 + 1, o:o + 4] = [-2 * A[0, 0], -2 * A[0, 1], 0, 0]
            J_rig[i * 3 + 2, o:o + 4] = [0, 0, -2 * A[1, 0], -2 * A[1, 1]]
    J_sm = np.zeros((ne * 2 * dim, nP))
    for i in range(ne):
        p1, p2 = int(edges[i, 0]), int(edges[i, 1])
        disp = srcdata[p2] - srcdata[p1]
        r0 = i * 2 * dim
        for cc in range(dim):
            for r in range(dim):
                J_sm[r0 + cc, p1 * s + r * dim + cc] = disp[r]
                J_sm[r0 + dim + cc, p2 * s + r * dim + cc] = -disp[r]
            J_sm[r0 + cc, p1 * s + dim * dim + cc] = 1
            J_sm[r0 + cc, p2 * s + dim * dim + cc] = -1
            J_sm[r0 + dim + cc, p2 * s + dim * dim + cc] = 1
            J_sm[r0 + dim + cc, p1 * s + dim * dim + cc] = -1
    return np.concatenate([J_fit * c[0], J_rig * c[1], J_sm * c[2]], axis=0)


def row_multipliers(n, ne, dim, fit_w=None, rigid_w=None, edge_w=None):
    """Per-row multipliers of the extended energy (include/icp_hip.h nicp_set_weights; no
    counterpart in the reference), in the row order of min_func(compat=False)."""
    nrig = 6 if dim == 3 else 3
    f = np.ones(n) if fit_w is None else np.asarray(fit_w, dtype=np.float64)
    g = np.ones(n) if rigid_w is None else np.asarray(rigid_w, dtype=np.float64)
    e = np.ones(ne) if edge_w is None else np.asarray(edge_w, dtype=np.float64)
    return np.concatenate([np.repeat(f, dim), np.repeat(g, nrig), np.repeat(e, 2 * dim)])


def min_func_ext(P, nbr, wt, srcdata, edges, tgtdata, c, fit_w=None, rigid_w=None, edge_w=None,
                 lm_vertex=None, lm_pos=None, c_land=0.0):
    """min_func(compat=False) with row multipliers, followed by the landmark rows
    c_land * (L_k - Deform(P)[l_k]) (nicp_set_landmarks)."""
    srcdata = np.asarray(srcdata, dtype=np.float64)
    n, dim = srcdata.shape
    r = min_func(P, nbr, wt, srcdata, edges, tgtdata, c, compat=False)
    r = r * row_multipliers(n, edges.shape[0], dim, fit_w, rigid_w, edge_w)
    if lm_vertex is not None and len(lm_vertex):
        dd = deform(P, nbr, wt, srcdata)
        r = np.concatenate([r, (c_land * (np.asarray(lm_pos) - dd[np.asarray(lm_vertex)])).flatten()])
    return r


def d_min_func_ext(P, nbr, wt, srcdata, edges, tgtdata, c, fit_w=None, rigid_w=None, edge_w=None,
                   lm_vertex=None, lm_pos=None, c_land=0.0):
    """Jacobian of min_func_ext: weighted rows of d_min_func(compat=False); a landmark row is the fit
    row of its vertex with c_land in place of c_fit."""
    srcdata = np.asarray(srcdata, dtype=np.float64)
    n, dim = srcdata.shape
    J = d_min_func(P, nbr, wt, srcdata, edges, tgtdata, c, compat=False)
    Jw = J * row_multipliers(n, edges.shape[0], dim, fit_w, rigid_w, edge_w)[:, None]
    if lm_vertex is not None and len(lm_vertex):
        J1 = d_min_func(P, nbr, wt, srcdata, edges, tgtdata, (1.0, c[1], c[2]), compat=False)
        rows = (np.asarray(lm_vertex)[:, None] * dim + np.arange(dim)[None, :]).flatten()
        Jw = np.concatenate([Jw, c_land * J1[rows]], axis=0)
    return Jw


def nicp(P, nbr, wt, source, edges, target, tol, run, c=(C_FIT, C_RIGID, C_SMOOTH), history=None,
         compat=True, **lsq_kw):
    """nicp_3d.py:344-363: outer loop with scipy.optimize.leastsq (MINPACK lmder)
    and the stiffness-halving schedule.  Returns (deformed, P, (c_fit,c_rigid,c_smooth)).
    lsq_kw (e.g. ftol=, xtol=) tightens MINPACK's stopping tests for solver-parity tests; the
    reference itself uses the defaults."""
    from scipy import optimize
    c = list(c)
    source = np.asarray(source, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    err = 100000000
    for i in range(run):
        e = err
        dd = deform(P, nbr, wt, source)
        nn = match(dd, target)
        cc = tuple(c)
        lsq = optimize.leastsq(lambda p: min_func(p, nbr, wt, source, edges, nn, cc, compat), P,
                               Dfun=lambda p: d_min_func(p, nbr, wt, source, edges, nn, cc, compat),
                               full_output=1, **lsq_kw)
        P = lsq[0]
        err = (lsq[2]['fvec'] * lsq[2]['fvec']).sum()
        if history is not None:
            history.append((float(err), c[1], c[2], lsq[2]['nfev'], lsq[4]))
        if (e - err) < tol:
            c[1] /= 2.0
            c[2] /= 2.0
            if c[1] < 0.1 or c[2] < 0.1:
                break
    return deform(P, nbr, wt, source), P, tuple(c)


def lm_dense(P0, nbr, wt, source, edges, nn, c=(C_FIT, C_RIGID, C_SMOOTH), compat=True, steps=1,
             lambda0=1e-3, ftol=1.49012e-8, xtol=1.49012e-8, trace=None, accel=False):
    """Dense twin of the PRODUCT's Levenberg-Marquardt loop (icp_amd/csrc/nicp.hip, solve_impl), on this
    file's min_func / d_min_func -- NOT a restatement of MINPACK.  Test infrastructure: what the reference
    determines in compat mode is the pair (MinFunc, D_MinFunc), pinned to it at 1e-12 by the golden
    fixtures; given that pair, every LM step `(J^T J + lam D^2) delta = -J^T r` with MINPACK's column
    scaling D = max-so-far column norm of J (lmder.f, mode 1) is a determinate quantity, which the
    leastsq END POINT in compat mode is not (oracle/study_compat_sensitivity.py).  The GPU solve with a
    tight CG tolerance must reproduce these steps to 1e-6 (tests/test_gpu_nicp.py).

    Runs `steps` accepted steps (or stops on lmder's ftol / xtol tests) and returns
    (P, info dict(err0, err, nfev, lm_iters, info, lams)).  Acceptance: gain ratio > 1e-4 (lmder.f);
    damping: Nielsen's update on acceptance, lam *= nu, nu *= 2 on rejection.
    accel=True: the product's accelerated iteration (nicp_solve_opts::accel): every step gets the second-order
    correction d2 = -1/2 (J^T J + lam D^2)^-1 J^T (r(x+d1) - 2 r(x) + r(x-d1)), kept when |D d2| <= 0.75 |D d1|; the
    gain ratio uses the first-order step's model."""
    x = np.array(P0, dtype=np.float64)
    f = lambda p: min_func(p, nbr, wt, source, edges, nn, c, compat)
    Jf = lambda p: d_min_func(p, nbr, wt, source, edges, nn, c, compat)
    r = f(x)
    f2 = float(r @ r)
    out = dict(err0=f2, nfev=1, lm_iters=0, info=0, lams=[])
    lam, nu = lambda0, 2.0
    d2 = None
    for it in range(steps):
        out["lm_iters"] = it + 1
        J = Jf(x)
        grad = J.T @ r
        col2 = (J * J).sum(0)
        d2 = np.where(col2 > 0, col2, 1.0) if d2 is None else np.maximum(d2, col2)
        H = J.T @ J
        accepted = False
        while not accepted and not out["info"]:
            delta = np.linalg.solve(H + lam * np.diag(d2), -grad)
            dmodel = delta
            if accel:
                fvv = f(x + delta) - 2.0 * r + f(x - delta)
                out["nfev"] += 2
                delta2 = np.linalg.solve(H + lam * np.diag(d2), -0.5 * (J.T @ fvv))
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
                trace.append(dict(it=it, lam=lam, f2=f2, f2t=f2t, ratio=ratio, delta=delta.copy()))
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


# --------------------------------------------------------------------------
# seeded synthetic workloads -- shared by tests and bench
# --------------------------------------------------------------------------
def rz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def make_rigid_cloud(n, seed=0, angle=0.3, shift=(0.1, 0.05, -0.02)):
    """C1/C2 recipe: tgt = N(0,1)^3; src = ((tgt - shift) Rz(angle)^T)[perm]."""
    rng = np.random.default_rng(seed)
    tgt = rng.standard_normal((n, 3))
    src = ((tgt - np.array(shift)) @ rz(angle).T)[rng.permutation(n)]
    return np.ascontiguousarray(src), np.ascontiguousarray(tgt)


def make_grid_mesh(nx, ny, seed=0):
    """C3-style mesh: nx*ny grid on a sphere patch, edges from the grid (E ~ 3N
    with diagonals); target = source displaced along the normal by a smooth field
    plus a small rigid motion."""
    u = np.linspace(-0.6, 0.6, nx)
    v = np.linspace(-0.6, 0.6, ny)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    rng = np.random.default_rng(seed)
    uu = uu + rng.uniform(-1, 1, uu.shape) * (0.15 * 1.2 / nx)
    vv = vv + rng.uniform(-1, 1, vv.shape) * (0.15 * 1.2 / ny)
    x = np.sin(uu) * np.cos(vv)
    y = np.sin(vv)
    z = np.cos(uu) * np.cos(vv)
    src = np.stack([x, y, z], -1).reshape(-1, 3)
    nrm = src / np.linalg.norm(src, axis=1, keepdims=True)
    idx = np.arange(nx * ny).reshape(nx, ny)
    e = [np.stack([idx[:-1, :].ravel(), idx[1:, :].ravel()], 1),
         np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()], 1),
         np.stack([idx[:-1, :-1].ravel(), idx[1:, 1:].ravel()], 1)]
    edges = np.concatenate(e, 0).astype(np.int32)
    tgt = src + 0.05 * np.sin(2 * np.pi * src[:, :1]) * nrm
    tgt = tgt @ rot_matrix(0.02, -0.03, 0.05) + np.array([0.01, -0.02, 0.015])
    return np.ascontiguousarray(src), edges, np.ascontiguousarray(tgt)
