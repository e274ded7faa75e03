"""The sparse / vectorised oracle (oracle/icp_oracle_sparse.py) against the loop oracle and the fixtures the
reference's own code produced (tests/golden/nicp_step_*.npz, nicp_row_20.npz): same residual, same Jacobian
entry for entry, same Levenberg-Marquardt steps as the dense twin.  CPU only."""
import numpy as np
import pytest

import icp_oracle as orc
import icp_oracle_sparse as ors
from conftest import golden

C = (0.1, 1000.0, 100.0)


def sparse_from_dense(w, k=4):
    n = w.shape[0]
    nbr = np.zeros((n, k), np.int32)
    wt = np.zeros((n, k))
    for j in range(n):
        nz = np.nonzero(w[:, j])[0]
        nbr[j] = nz
        wt[j] = w[nz, j]
    return nbr, wt


@pytest.mark.parametrize("name", ["nicp_step_24", "nicp_step_60", "nicp_row_20"])
def test_sparse_operators_match_reference_fixtures(name):
    g = golden(name)
    v, edges, nn, P = g["v"], g["edges"], g["nn"], g["P_rand"]
    nbr, wt = sparse_from_dense(g["w"])
    if "deform_rand" in g.files:
        np.testing.assert_allclose(ors.deform_v(P, nbr, wt, v), g["deform_rand"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(ors.min_func_v(P, nbr, wt, v, edges, nn, C, True), g["r"], rtol=1e-12, atol=1e-12)
    J = ors.d_min_func_sparse(P, nbr, wt, v, edges, nn, C, True)
    if "J" in g.files:                                        # the reference's own dense D_MinFunc output
        np.testing.assert_allclose(J.toarray(), g["J"], rtol=1e-12, atol=1e-12)
    if "pv" in g.files:
        np.testing.assert_allclose(J @ g["pv"], g["Jv"], rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(J.T @ g["pu"], g["JTu"], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("name", ["nicp_init_60", "nicp_init_200"])
def test_vectorised_deform_matches_reference_fixtures(name):
    g = golden(name)
    nbr, wt = sparse_from_dense(g["w"])
    np.testing.assert_allclose(ors.deform_v(g["P0"], nbr, wt, g["v"]), g["deform_id"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(ors.deform_v(g["P_rand"], nbr, wt, g["v"]), g["deform_rand"], rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("name", ["nicp_step_24", "nicp_step_60", "nicp_row_20"])
@pytest.mark.parametrize("compat", [True, False])
def test_sparse_matches_loop_oracle(name, compat):
    g = golden(name)
    v, edges, nn = g["v"], g["edges"], g["nn"]
    nbr, wt = sparse_from_dense(g["w"])
    for P in (g["P0"], g["P_rand"]):
        np.testing.assert_allclose(ors.min_func_v(P, nbr, wt, v, edges, nn, C, compat),
                                   orc.min_func(P, nbr, wt, v, edges, nn, C, compat), rtol=1e-12, atol=1e-12)
        Jd = orc.d_min_func(P, nbr, wt, v, edges, nn, C, compat)
        Js = ors.d_min_func_sparse(P, nbr, wt, v, edges, nn, C, compat)
        np.testing.assert_allclose(Js.toarray(), Jd, rtol=1e-12, atol=1e-12)
        assert Js.nnz <= np.count_nonzero(Jd) + Jd.shape[0] * 9      # sparse: nothing stored densely


@pytest.mark.parametrize("name", ["nicp_step_60", "nicp_row_20"])
@pytest.mark.parametrize("compat", [True, False])
def test_lm_sparse_equals_lm_dense(name, compat):
    g = golden(name)
    v, edges, nn = g["v"], g["edges"], g["nn"]
    nbr, wt = sparse_from_dense(g["w"])
    for start in ("P0", "P_rand"):
        td, ts = [], []
        orc.lm_dense(g[start], nbr, wt, v, edges, nn, C, compat, steps=3, ftol=0.0, xtol=0.0, trace=td)
        ors.lm_sparse(g[start], nbr, wt, v, edges, nn, C, compat, steps=3, ftol=0.0, xtol=0.0, trace=ts)
        # trial by trial: same damping, same trial energy, same step.  Trials whose damping has run away (compat mode
        # from a perturbed P: the step collapses and accept / reject is decided by the rounding noise of the energy)
        # are not a determinate quantity in either twin and are left out.
        live = [(a, b) for a, b in zip(td, ts) if a["lam"] < 1e8]
        assert len(live) >= 3
        for a, b in live:
            assert a["it"] == b["it"] and abs(a["lam"] - b["lam"]) <= 1e-9 * a["lam"]
            np.testing.assert_allclose(b["f2t"], a["f2t"], rtol=1e-9)
            assert np.abs(a["delta"] - b["delta"]).max() <= 1e-7 * np.abs(a["delta"]).max()


def test_sparse_oracle_scales_to_a_2500_vertex_mesh():
    """What the GPU tests use it for: a mesh the dense oracle cannot hold (J would be 24 000 x 30 006 here,
    90 000 x 120 006 at C3)."""
    v, edges, tgt = orc.make_grid_mesh(50, 50, seed=7)
    import c_oracle
    nbr, wt = orc.knn_weights(v)
    nn = tgt[c_oracle.match_idx(v, tgt)[0]]
    n = len(v)
    P0 = np.append(np.tile(np.append(np.identity(3).flatten(), np.zeros(3)), n), np.zeros(6))
    P, info = ors.lm_sparse(P0, nbr, wt, v, edges, nn, C, False, steps=1, ftol=0.0, xtol=0.0)
    assert info["err"] < info["err0"]
    J = ors.d_min_func_sparse(P, nbr, wt, v, edges, nn, C, False)
    # true derivative in corrected mode: central difference along a random direction
    rng = np.random.default_rng(0)
    d = rng.standard_normal(P.shape)
    h = 1e-6
    fd = (ors.min_func_v(P + h * d, nbr, wt, v, edges, nn, C, False) - ors.min_func_v(P - h * d, nbr, wt, v, edges, nn, C, False)) / (2 * h)
    np.testing.assert_allclose(J @ d, fd, rtol=1e-5, atol=1e-4)
