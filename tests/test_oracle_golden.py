"""Pin oracle/icp_oracle.py against fixtures produced by the reference's own
function bodies (oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest

import icp_oracle as orc
from conftest import golden


def sparse_from_dense(w, k=4):
    n = w.shape[0]
    nbr = np.zeros((n, k), np.int32)
    wt = np.zeros((n, k))
    for j in range(n):
        nz = np.nonzero(w[:, j])[0]
        assert len(nz) == k
        nbr[j] = nz
        wt[j] = w[nz, j]
    return nbr, wt


@pytest.mark.parametrize("name", ["closest_1k_normal3", "closest_20_50_dim2", "closest_ties3", "closest_nearties3"])
def test_match_idx(name):
    g = golden(name)
    idx, d2 = orc.match_idx(g["src"], g["tgt"])
    assert np.array_equal(idx, g["idx"])


def test_match_4k_50k():
    g = golden("closest_4k_50k_uniform3")
    rng = np.random.default_rng(int(g["seed"]))
    src = rng.uniform(-1, 1, (4096, 3))
    tgt = rng.uniform(-1, 1, (50000, 3))
    idx, _ = orc.match_idx(src[:512], tgt)
    assert np.array_equal(idx, g["idx"][:512])


def test_closest_col():
    g = golden("closest_col2")
    idx, _ = orc.match_idx(g["src"].T.copy(), g["tgt"].T.copy())
    assert np.array_equal(idx, g["idx"])


def test_kabsch():
    g = golden("kabsch_3d")
    R, t = orc.find_transform(g["src"], g["nn"])
    assert np.array_equal(R, g["R"]) and np.array_equal(t, g["t"])
    assert orc.distance(g["src"], g["nn"]) == float(g["dist"])
    g = golden("kabsch_2d_T")
    assert np.array_equal(orc.find_transform_T(g["src"], g["nn"]), g["T"])
    g = golden("kabsch_3d_reflect")
    R, t = orc.find_transform(g["src"], g["nn"])
    assert np.array_equal(R, g["R"]) and np.linalg.det(R) < 0   # Q2: no reflection fix


def test_icp_c1():
    g = golden("icp_c1")
    src, tgt = orc.make_rigid_cloud(int(g["n"]), seed=int(g["seed"]))
    hist = []
    T = orc.icp(src, tgt, float(g["tol"]), int(g["run"]), history=hist)
    assert np.array_equal(T, g["T"])
    assert len(hist) == g["idx_hist"].shape[0]
    for (idx, err), gi, ge in zip(hist, g["idx_hist"], g["err_hist"]):
        assert np.array_equal(idx, gi) and err == ge
    # recovered motion: R = Rz(0.3), t = (0.1, 0.05, -0.02)
    assert np.allclose(T[:3, :3], orc.rz(0.3), atol=1e-12)
    assert np.allclose(T[3, :3], [0.1, 0.05, -0.02], atol=1e-12)


def test_icp_forced_and_row():
    g = golden("icp_forced30")
    assert np.array_equal(orc.icp(g["src"], g["tgt"], -np.inf, 30), g["T"])
    g = golden("icp_row_demo")
    hist = []
    T = orc.icp(g["src"], g["tgt"], 0.1, 20, history=hist, use_tol=False)
    assert np.array_equal(T, g["T"])
    assert np.array_equal(np.array([h[1] for h in hist]), g["err_hist"])


def test_icp_run0():
    with pytest.raises(NameError):
        orc.icp(np.zeros((3, 3)), np.zeros((3, 3)), 0.1, 0)


@pytest.mark.parametrize("name", ["nicp_init_60", "nicp_init_200"])
def test_nicp_init(name):
    g = golden(name)
    v = g["v"]
    P, nbr, wt = orc.initialise(v)
    assert np.array_equal(P, g["P0"])
    w = orc.dense_weights(nbr, wt)
    assert np.array_equal(w, g["w"])
    assert np.allclose(w.sum(0), 1.0, atol=1e-14)
    for row, j in enumerate(range(0, v.shape[0], 7)):
        idx, dist = orc.knn(v, v[j], 6)
        assert np.array_equal(idx, g["knn_idx"][row]) and np.array_equal(dist, g["knn_dist"][row])
        idx2, dist2 = orc.knn(v, v[j], 6, cxx_order=True)
        assert np.array_equal(idx2, idx) and np.array_equal(dist2, dist)
    # reference Deform visits neighbours in ascending j: same order here -> bitwise
    nbr_s, wt_s = sparse_from_dense(g["w"])
    assert np.array_equal(orc.deform(g["P0"], nbr_s, wt_s, v), g["deform_id"])
    np.testing.assert_allclose(orc.deform(g["P_rand"], nbr, wt, v), g["deform_rand"], rtol=1e-13, atol=1e-15)
    assert np.allclose(g["deform_id"], v, atol=1e-14)
    assert np.array_equal(orc.rot_matrix(0.3, -0.4, 0.5), g["R"])
    assert np.array_equal(np.array(orc.d_rot_matrix(0.3, -0.4, 0.5)), g["dR"])


@pytest.mark.parametrize("name", ["nicp_step_24", "nicp_step_60"])
def test_nicp_step(name):
    g = golden(name)
    v, edges = g["v"], g["edges"]
    nbr, wt = sparse_from_dense(g["w"])
    r = orc.min_func(g["P_rand"], nbr, wt, v, edges, g["nn"])
    np.testing.assert_allclose(r, g["r"], rtol=1e-12, atol=1e-13)
    J = orc.d_min_func(g["P_rand"], nbr, wt, v, edges, g["nn"])
    if "J" in g.files:
        np.testing.assert_allclose(J, g["J"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(J @ g["pv"], g["Jv"], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(J.T @ g["pu"], g["JTu"], rtol=1e-10, atol=1e-9)
    # Q5: compat residual is NOT what the Jacobian differentiates; the non-compat pair is consistent
    n = v.shape[0]
    assert np.count_nonzero(r[3 * n:9 * n]) <= 3 * n + 3
    P = g["P_rand"]
    Jc = orc.d_min_func(P, nbr, wt, v, edges, g["nn"], compat=False)
    rng = np.random.default_rng(0)
    dp = rng.standard_normal(P.shape) * 1e-6
    fd = (orc.min_func(P + dp, nbr, wt, v, edges, g["nn"], compat=False) -
          orc.min_func(P - dp, nbr, wt, v, edges, g["nn"], compat=False)) / 2
    np.testing.assert_allclose(Jc @ dp, fd, rtol=1e-5, atol=1e-7)


def test_nicp_leastsq_and_outer():
    g = golden("nicp_step_24")
    v, edges = g["v"], g["edges"]
    nbr, wt = sparse_from_dense(g["w"])
    from scipy import optimize
    lsq = optimize.leastsq(lambda p: orc.min_func(p, nbr, wt, v, edges, g["nn"]), g["P0"],
                           Dfun=lambda p: orc.d_min_func(p, nbr, wt, v, edges, g["nn"]), full_output=1)
    np.testing.assert_allclose((lsq[2]["fvec"] ** 2).sum(), float(g["lsq_err"]), rtol=1e-8)
    np.testing.assert_allclose(orc.deform(lsq[0], nbr, wt, v), g["lsq_deformed"], rtol=1e-6, atol=1e-8)
    dd, P, c = orc.nicp(g["P0"].copy(), nbr, wt, v, edges, g["tgt"], 0.01, int(g["nicp_run"]))
    np.testing.assert_allclose(dd, g["nicp_dd"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(np.array(c), g["nicp_c"])


def test_nicp_row():
    g = golden("nicp_row_20")
    v, edges = g["v"], g["edges"]
    P0, nbr, wt = orc.initialise(v)
    assert np.array_equal(P0, g["P0"])
    assert np.array_equal(orc.dense_weights(nbr, wt), g["w"])
    np.testing.assert_allclose(orc.deform(g["P_rand"], nbr, wt, v), g["deform_rand"], rtol=1e-13)
    np.testing.assert_allclose(orc.min_func(g["P_rand"], nbr, wt, v, edges, g["nn"]), g["r"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(orc.d_min_func(g["P_rand"], nbr, wt, v, edges, g["nn"]), g["J"], rtol=1e-12, atol=1e-13)


def test_nicp_row_rt():
    g = golden("nicp_row_rt")
    p = g["p"]
    assert np.array_equal(orc.deform_rt(g["src"], *p), g["deform"])
    assert np.array_equal(orc.disp_rt(p, g["src"], g["nn"]), g["disp"])
    assert np.array_equal(orc.d_disp_rt(p, g["src"], g["nn"]), g["d_disp"])
    assert np.array_equal(orc.match(orc.deform_rt(g["src"], 0, 0, 0), g["tgt"]), g["nn"])
