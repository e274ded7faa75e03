"""The oracle's extended NICP energy (row multipliers + landmark rows) is not in the
reference, so it is pinned by construction: unit multipliers and no landmarks reproduce the corrected
reference restatement exactly, and its Jacobian is the derivative of its residual."""
import numpy as np
import pytest

import icp_oracle as orc
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


@pytest.mark.parametrize("name", ["nicp_step_24", "nicp_row_20"])
def test_extension_is_consistent(name):
    g = golden(name)
    v, edges, nn, P = g["v"], g["edges"], g["nn"], g["P_rand"]
    n, dim = v.shape
    nbr, wt = sparse_from_dense(g["w"])
    assert np.array_equal(orc.min_func_ext(P, nbr, wt, v, edges, nn, C), orc.min_func(P, nbr, wt, v, edges, nn, C, False))
    assert np.array_equal(orc.d_min_func_ext(P, nbr, wt, v, edges, nn, C), orc.d_min_func(P, nbr, wt, v, edges, nn, C, False))
    rng = np.random.default_rng(2)
    ext = dict(fit_w=rng.uniform(0, 2, n), rigid_w=rng.uniform(0.5, 1.5, n), edge_w=rng.uniform(0, 3, len(edges)),
               lm_vertex=np.array([3, 0, n - 1]), lm_pos=rng.standard_normal((3, dim)), c_land=4.0)
    r = orc.min_func_ext(P, nbr, wt, v, edges, nn, C, **ext)
    J = orc.d_min_func_ext(P, nbr, wt, v, edges, nn, C, **ext)
    assert r.shape[0] == J.shape[0] == orc.min_func(P, nbr, wt, v, edges, nn, C, False).shape[0] + 3 * dim
    h = 1e-6
    for k in rng.permutation(len(P))[:40]:
        e = np.zeros(len(P))
        e[k] = h
        fd = (orc.min_func_ext(P + e, nbr, wt, v, edges, nn, C, **ext) - orc.min_func_ext(P - e, nbr, wt, v, edges, nn, C, **ext)) / (2 * h)
        np.testing.assert_allclose(J[:, k], fd, rtol=2e-5, atol=2e-5)
