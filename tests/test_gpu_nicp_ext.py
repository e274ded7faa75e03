"""Extended NICP energy (include/icp_hip.h nicp_set_weights / nicp_set_landmarks):
per-vertex fit / rigidity multipliers, per-edge stiffness and landmark rows.  The reference has no
such terms, so parity is against the oracle's extension of its corrected residual / Jacobian
(oracle/icp_oracle.py min_func_ext, d_min_func_ext) and scipy's leastsq on them: parity unpinned
by the reference, 1e-6 relative like the rest of the NICP path."""
import numpy as np
import pytest

import icp_oracle as orc
from conftest import golden

pytestmark = pytest.mark.gpu
RTOL = 1e-6
C = (0.1, 1000.0, 100.0)
TIGHT = 1e-13   # both solvers run to the minimiser itself: with zero fit weights the default 1.5e-8 stops ~1e-6 short


@pytest.fixture(scope="module")
def nic():
    import torch
    assert torch.cuda.is_available()
    from icp_amd import nicp
    return nicp


def sparse_from_dense(w, k=4):
    n = w.shape[0]
    nbr = np.zeros((n, k), np.int32)
    wt = np.zeros((n, k))
    for j in range(n):
        nz = np.nonzero(w[:, j])[0]
        nbr[j] = nz
        wt[j] = w[nz, j]
    return nbr, wt


def extension(n, ne, dim, v, seed):
    rng = np.random.default_rng(seed)
    fit_w = rng.uniform(0.0, 2.0, n)
    fit_w[rng.permutation(n)[:n // 5]] = 0.0          # vertices without a fit term
    rigid_w = rng.uniform(0.5, 1.5, n)
    edge_w = rng.uniform(0.0, 3.0, ne)
    lm_vertex = rng.permutation(n)[:max(3, n // 8)].astype(np.int32)
    lm_vertex[0] = np.nonzero(fit_w == 0.0)[0][0]     # a landmark on a vertex whose fit weight is 0
    lm_vertex = np.unique(lm_vertex).astype(np.int32)[::-1].copy()   # not ascending on purpose
    lm_pos = v[lm_vertex] + rng.standard_normal((len(lm_vertex), dim)) * 0.05
    return dict(fit_w=fit_w, rigid_w=rigid_w, edge_w=edge_w, lm_vertex=lm_vertex, lm_pos=lm_pos, c_land=7.5)


def attach(w, edges, ext):
    w.set_weights(edges, ext["fit_w"], ext["rigid_w"], ext["edge_w"])
    w.set_landmarks(edges, ext["lm_vertex"], ext["lm_pos"], ext["c_land"])


@pytest.mark.parametrize("name", ["nicp_step_24", "nicp_step_60", "nicp_row_20"])
def test_weighted_operators_vs_oracle(nic, name):
    g = golden(name)
    v, edges, nn, P = g["v"], g["edges"], g["nn"], g["P_rand"]
    n, dim = v.shape
    nbr, wt = sparse_from_dense(g["w"])
    w = nic.SparseWeights(nbr, wt, dim)
    base_rows = w.rows(edges)
    assert base_rows == nic.n_rows(n, len(edges), dim)
    ext = extension(n, len(edges), dim, v, 3)
    attach(w, edges, ext)
    assert w.rows(edges) == base_rows + dim * len(ext["lm_vertex"])
    r = nic.residual(P, w, v, edges, nn, C, compat=False).cpu().numpy()
    np.testing.assert_allclose(r, orc.min_func_ext(P, nbr, wt, v, edges, nn, C, **ext), rtol=RTOL, atol=1e-9)
    Jo = orc.d_min_func_ext(P, nbr, wt, v, edges, nn, C, **ext)
    rng = np.random.default_rng(4)
    pv, pu = rng.standard_normal(len(P)), rng.standard_normal(Jo.shape[0])
    np.testing.assert_allclose(nic.jvp(P, w, v, edges, C, pv, compat=False).cpu().numpy(), Jo @ pv, rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(nic.jtvp(P, w, v, edges, C, pu, compat=False).cpu().numpy(), Jo.T @ pu, rtol=RTOL, atol=1e-6)
    # the Jacobian of the oracle extension really is the derivative of its residual (central differences)
    h = 1e-6
    k = 5
    e = np.zeros(len(P))
    e[k] = h
    fd = (orc.min_func_ext(P + e, nbr, wt, v, edges, nn, C, **ext) - orc.min_func_ext(P - e, nbr, wt, v, edges, nn, C, **ext)) / (2 * h)
    np.testing.assert_allclose(Jo[:, k], fd, rtol=1e-5, atol=1e-5)
    # the reference's quirk mode has no meaning here
    with pytest.raises(Exception, match="COMPAT"):
        nic.residual(P, w, v, edges, nn, C, compat=True)
    # removing the extension gives the plain operators back, bit for bit
    plain = nic.SparseWeights(nbr, wt, dim)
    w.set_weights(edges)
    w.set_landmarks(edges)
    assert w.rows(edges) == base_rows
    r0 = nic.residual(P, w, v, edges, nn, C, compat=False).cpu().numpy()
    assert np.array_equal(r0, nic.residual(P, plain, v, edges, nn, C, compat=False).cpu().numpy())


def test_unit_weights_change_nothing(nic):
    g = golden("nicp_step_60")
    v, edges, nn, P0 = g["v"], g["edges"], g["nn"], g["P0"]
    nbr, wt = sparse_from_dense(g["w"])
    a, b = nic.SparseWeights(nbr, wt, 3), nic.SparseWeights(nbr, wt, 3)
    b.set_weights(edges, np.ones(len(v)), np.ones(len(v)), np.ones(len(edges)))
    Pa, sa = nic.solve(P0, a, v, edges, nn, C, compat=False)
    Pb, sb = nic.solve(P0, b, v, edges, nn, C, compat=False)
    assert sa["lm_iters"] == sb["lm_iters"] and sa["cg_iters"] == sb["cg_iters"]
    assert np.array_equal(Pa.cpu().numpy(), Pb.cpu().numpy())


@pytest.mark.parametrize("name", ["nicp_step_24", "nicp_step_60", "nicp_row_20"])
def test_weighted_solve_vs_leastsq(nic, name):
    from scipy import optimize
    g = golden(name)
    v, edges, nn, P0 = g["v"], g["edges"], g["nn"], g["P0"]
    n, dim = v.shape
    nbr, wt = sparse_from_dense(g["w"])
    ext = extension(n, len(edges), dim, v, 8)
    lsq = optimize.leastsq(lambda p: orc.min_func_ext(p, nbr, wt, v, edges, nn, C, **ext), P0,
                           Dfun=lambda p: orc.d_min_func_ext(p, nbr, wt, v, edges, nn, C, **ext), full_output=1,
                           ftol=TIGHT, xtol=TIGHT)
    dd_ref = orc.deform(lsq[0], nbr, wt, v)
    e_ref = (lsq[2]["fvec"] ** 2).sum()
    w = nic.SparseWeights(nbr, wt, dim)
    attach(w, edges, ext)
    P, st = nic.solve(P0, w, v, edges, nn, C, compat=False, opts=dict(ftol=TIGHT, xtol=TIGHT))
    dd = nic.deform(P, w, v).cpu().numpy()
    print(name, "extended energy: leastsq", e_ref, "gpu", st)
    np.testing.assert_allclose(st["err"], e_ref, rtol=1e-6)
    # 1e-6 relative with an absolute floor of 1e-7 of the O(1) mesh size: vertices without a fit term leave nearly flat
    # directions, along which scipy's own result moves by several 1e-8 from run to run (threaded BLAS in the oracle)
    np.testing.assert_allclose(dd, dd_ref, rtol=RTOL, atol=1e-7)
    # the landmarks pull: with a large c_land the landmark vertices land on their positions
    w.set_landmarks(edges, ext["lm_vertex"], ext["lm_pos"], 1e4)
    P2, st2 = nic.solve(P0, w, v, edges, nn, C, compat=False)
    dd2 = nic.deform(P2, w, v).cpu().numpy()
    assert np.abs(dd2[ext["lm_vertex"]] - ext["lm_pos"]).max() < 1e-3
    assert np.abs(dd[ext["lm_vertex"]] - ext["lm_pos"]).max() > np.abs(dd2[ext["lm_vertex"]] - ext["lm_pos"]).max()


def test_landmark_argument_errors(nic):
    g = golden("nicp_step_24")
    v, edges = g["v"], g["edges"]
    nbr, wt = sparse_from_dense(g["w"])
    w = nic.SparseWeights(nbr, wt, 3)
    with pytest.raises(Exception, match="repeated"):
        w.set_landmarks(edges, np.array([1, 2, 1], np.int32), np.zeros((3, 3)), 1.0)
    with pytest.raises(Exception, match="out of range"):
        w.set_landmarks(edges, np.array([1, len(v)], np.int32), np.zeros((2, 3)), 1.0)
    assert w.rows(edges) == nic.n_rows(len(v), len(edges), 3)        # a refused call leaves no landmarks behind
    with pytest.raises(ValueError):
        w.set_weights(edges, fit_w=-np.ones(len(v)))
    with pytest.raises(ValueError):
        w.set_weights(edges, edge_w=np.ones(len(edges) + 1))
