"""GPU parity tests for the NICP path (KNN weights, Deform, residual, Jacobian products,
LM-CG solve, outer loop) against the golden fixtures / the oracle.  Tolerance for fp64
values: 1e-6 relative (north_star); KNN indices are exact."""
import numpy as np
import pytest

import icp_oracle as orc
import c_oracle
from conftest import golden

pytestmark = pytest.mark.gpu
RTOL = 1e-6
C = (0.1, 1000.0, 100.0)


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available()
    from icp_amd import nicp, nicp_3d
    return dict(nicp=nicp, nicp_3d=nicp_3d, torch=torch)


def sparse_from_dense(w, k=4):
    n = w.shape[0]
    nbr = np.zeros((n, k), np.int32)
    wt = np.zeros((n, k))
    for j in range(n):
        nz = np.nonzero(w[:, j])[0]
        nbr[j] = nz
        wt[j] = w[nz, j]
    return nbr, wt


@pytest.mark.parametrize("name", ["nicp_init_60", "nicp_init_200"])
def test_initialise_golden(mods, name):
    g = golden(name)
    v = g["v"]
    P, w = mods["nicp_3d"].Initialise(v, g["tgt"])
    assert np.array_equal(P, g["P0"])
    np.testing.assert_allclose(w.toarray(), g["w"], rtol=1e-12, atol=1e-15)
    assert np.array_equal(w.toarray() != 0, g["w"] != 0)          # same sparsity = same neighbours
    np.testing.assert_allclose(w.sum(0), 1.0, atol=1e-14)
    for row, j in enumerate(range(0, v.shape[0], 7)):
        idx, dist = mods["nicp_3d"].KNN(v, v[j], 6)
        assert np.array_equal(idx, g["knn_idx"][row])
        assert np.array_equal(dist, g["knn_dist"][row])            # sqrt is correctly rounded
    np.testing.assert_allclose(mods["nicp_3d"].Deform(g["P0"], w, v), g["deform_id"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(mods["nicp_3d"].Deform(g["P_rand"], w, v), g["deform_rand"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(mods["nicp_3d"].Deform(g["P_rand"], g["w"], v), g["deform_rand"], rtol=RTOL, atol=1e-12)
    assert np.array_equal(mods["nicp_3d"].RotMatrix(0.3, -0.4, 0.5), g["R"])


def test_knn_vs_c_oracle_ties(mods):
    # lattice with exact distance ties + duplicates: order must be (distance, index)
    gpts = np.stack(np.meshgrid(np.arange(8.), np.arange(8.), np.arange(4.), indexing="ij"), -1).reshape(-1, 3)
    v = np.concatenate([gpts, gpts[:40]])
    idx, dist = mods["nicp"].knn(v, v, 6)
    oi, od = c_oracle.knn_rows(v, 6)
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(dist.cpu().numpy(), od)
    rng = np.random.default_rng(3)
    v = rng.standard_normal((5000, 3))
    idx, dist = mods["nicp"].knn(v, v, 6)
    oi, od = c_oracle.knn_rows(v, 6)
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(dist.cpu().numpy(), od)
    idx8, dist8 = mods["nicp"].knn(v, v[:100], 8)
    oi8, od8 = c_oracle.knn_rows(v, 8)
    assert np.array_equal(idx8.cpu().numpy(), oi8[:100])
    v2 = rng.standard_normal((700, 2))
    idx, dist = mods["nicp"].knn(v2, v2, 6)
    oi, od = c_oracle.knn_rows(v2, 6)
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(dist.cpu().numpy(), od)


@pytest.mark.parametrize("name", ["nicp_step_24", "nicp_step_60"])
def test_residual_and_jacobian_golden(mods, name):
    g = golden(name)
    n3 = mods["nicp_3d"]
    v, edges, nn, P = g["v"], g["edges"], g["nn"], g["P_rand"]
    r = n3.MinFunc(P, g["w"], v, edges, nn)
    np.testing.assert_allclose(r, g["r"], rtol=RTOL, atol=1e-9)
    nzr = g["r"][3 * len(v):9 * len(v)] != 0                        # Q5 layout: same slots populated
    assert np.array_equal(r[3 * len(v):9 * len(v)] != 0, nzr)
    Jv = mods["nicp"].jvp(P, g["w"], v, edges, C, g["pv"]).cpu().numpy()
    np.testing.assert_allclose(Jv, g["Jv"], rtol=RTOL, atol=1e-7)
    JTu = mods["nicp"].jtvp(P, g["w"], v, edges, C, g["pu"]).cpu().numpy()
    np.testing.assert_allclose(JTu, g["JTu"], rtol=RTOL, atol=1e-7)
    if "J" in g.files:
        J = n3.D_MinFunc(P, g["w"], v, edges, nn)
        np.testing.assert_allclose(J, g["J"], rtol=RTOL, atol=1e-9)


def test_corrected_mode_operators(mods):
    g = golden("nicp_step_24")
    v, edges, nn, P = g["v"], g["edges"], g["nn"], g["P_rand"]
    nbr, wt = sparse_from_dense(g["w"])
    r = mods["nicp"].residual(P, g["w"], v, edges, nn, C, compat=False).cpu().numpy()
    np.testing.assert_allclose(r, orc.min_func(P, nbr, wt, v, edges, nn, compat=False), rtol=RTOL, atol=1e-9)
    Jo = orc.d_min_func(P, nbr, wt, v, edges, nn, compat=False)
    Jv = mods["nicp"].jvp(P, g["w"], v, edges, C, g["pv"], compat=False).cpu().numpy()
    np.testing.assert_allclose(Jv, Jo @ g["pv"], rtol=RTOL, atol=1e-7)
    JTu = mods["nicp"].jtvp(P, g["w"], v, edges, C, g["pu"], compat=False).cpu().numpy()
    np.testing.assert_allclose(JTu, Jo.T @ g["pu"], rtol=RTOL, atol=1e-7)


def test_row2d_operators(mods):
    g = golden("nicp_row_20")
    v, edges, nn, P = g["v"], g["edges"], g["nn"], g["P_rand"]
    nic = mods["nicp"]
    w = nic.knn_weights(v)
    np.testing.assert_allclose(w.toarray(), g["w"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(nic.deform(P, w, v).cpu().numpy(), g["deform_rand"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(nic.residual(P, w, v, edges, nn, (0.1, 1000, 100)).cpu().numpy(), g["r"], rtol=RTOL, atol=1e-9)
    J = nic.dense_jacobian(P, w, v, edges, (0.1, 1000, 100)).cpu().numpy()
    np.testing.assert_allclose(J, g["J"], rtol=RTOL, atol=1e-9)                # incl. Q7
    rng = np.random.default_rng(1)
    u = rng.standard_normal(J.shape[0])
    np.testing.assert_allclose(nic.jtvp(P, w, v, edges, (0.1, 1000, 100), u).cpu().numpy(), g["J"].T @ u, rtol=RTOL, atol=1e-7)


@pytest.mark.parametrize("name", ["nicp_step_24", "nicp_step_60"])
def test_solve_corrected_vs_leastsq(mods, name):
    """Self-consistent residual/Jacobian: the minimiser is well defined, so the LM-CG solve
    must land on scipy.optimize.leastsq's deformed vertices to 1e-6."""
    from scipy import optimize
    g = golden(name)
    v, edges, nn, P0 = g["v"], g["edges"], g["nn"], g["P0"]
    nbr, wt = sparse_from_dense(g["w"])
    lsq = optimize.leastsq(lambda p: orc.min_func(p, nbr, wt, v, edges, nn, C, False), P0,
                           Dfun=lambda p: orc.d_min_func(p, nbr, wt, v, edges, nn, C, False), full_output=1)
    dd_ref = orc.deform(lsq[0], nbr, wt, v)
    e_ref = (lsq[2]["fvec"] ** 2).sum()
    P, st = mods["nicp"].solve(P0, g["w"], v, edges, nn, C, compat=False)
    dd = mods["nicp"].deform(P, g["w"], v).cpu().numpy()
    print(name, "corrected: leastsq err", e_ref, "gpu", st)
    np.testing.assert_allclose(st["err"], e_ref, rtol=1e-6)
    np.testing.assert_allclose(dd, dd_ref, rtol=RTOL, atol=1e-8)


@pytest.mark.parametrize("name", ["nicp_step_24", "nicp_step_60"])
def test_solve_compat_vs_golden(mods, name):
    """Reference-as-coded mode: the Jacobian is not the derivative of the residual (Q5/Q6), so
    MINPACK stops where its trust region collapses (ier=2) -- a path-dependent point no other
    solver can hit exactly (DESIGN.md 'NICP solve parity').  Measured floor: perturbing P0 by 1e-15 moves
    scipy's OWN end point on these fixtures by 2e-5 ... 1.7e-3 (relative, deformed vertices) and its
    energy by 4 % (oracle/study_compat_sensitivity.py -> profiles/r2_compat_sensitivity.txt), so 1e-6 of
    that end point is not a meaningful bar.  What is determinate -- every LM step's linear solve -- is held
    to 1e-6 by test_lm_steps_match_dense_solves.  Pinned here: the GPU solve reaches an energy no worse
    than the reference's and deformed vertices within 1e-4 of it."""
    g = golden(name)
    v, edges, nn, P0 = g["v"], g["edges"], g["nn"], g["P0"]
    P, st = mods["nicp"].solve(P0, g["w"], v, edges, nn, C, compat=True)
    dd = mods["nicp"].deform(P, g["w"], v).cpu().numpy()
    print(name, "compat: leastsq err", float(g["lsq_err"]), "gpu", st, "max |dd-ref|", np.abs(dd - g["lsq_deformed"]).max())
    assert st["err"] <= float(g["lsq_err"]) * (1 + 1e-6)
    np.testing.assert_allclose(dd, g["lsq_deformed"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["nicp_step_24", "nicp_step_60", "nicp_row_20"])
@pytest.mark.parametrize("compat", [True, False, "accel"])
@pytest.mark.parametrize("start", ["P0", "P_rand"])
def test_lm_steps_match_dense_solves(mods, name, compat, start):
    """What IS determinate in the reference-as-coded mode: given the pinned pair (MinFunc, D_MinFunc), each
    Levenberg-Marquardt step  (J^T J + lam D^2) delta = -J^T r.  The GPU solve (matrix-free J, preconditioned
    CG driven to 1e-13) must reproduce the dense numpy solves built from the oracle's J and r -- whose
    operators are pinned to the reference at 1e-12 -- to 1e-6, in compat AND corrected mode, from the
    reference's start P0 (R = I) and from a perturbed P (R != I: exercises Q6, rejected trials and the
    damping updates).  A sign or slot error in a compat-only branch of jp / jtu / block_diag that still
    lowered the energy would fail here."""
    # "accel": the accelerated iteration (nicp_solve_opts::accel = 1: second-order correction of every step), which is the
    # DEFAULT of the self-consistent mode, against the oracle's twin of it; True / False: the plain iteration (accel = -1)
    accel = compat == "accel"
    compat = compat is True
    g = golden(name)
    v, edges, nn, Ps = g["v"], g["edges"], g["nn"], g[start]
    nbr, wt = sparse_from_dense(g["w"])
    nic = mods["nicp"]
    w = nic.SparseWeights(nbr, wt, v.shape[1])
    for steps in (1, 4):
        Pref, info = orc.lm_dense(Ps, nbr, wt, v, edges, nn, C, compat, steps=steps, ftol=0.0, xtol=0.0, accel=accel)
        P, st = nic.solve(Ps, w, v, edges, nn, C, compat=compat,
                          opts=dict(max_lm_iters=steps, cg_tol=1e-13, ftol=0.0, xtol=0.0, accel=1 if accel else -1))
        d_ref, d_gpu = Pref - Ps, P.cpu().numpy() - Ps
        print(name, compat, start, steps, "ref", {k: info[k] for k in ("err0", "err", "nfev")}, "gpu", st,
              "max|d-dref|/max|dref|", np.abs(d_gpu - d_ref).max() / np.abs(d_ref).max())
        assert st["nfev"] == info["nfev"] and st["lm_iters"] == info["lm_iters"]      # same accept / reject sequence
        np.testing.assert_allclose(st["err0"], info["err0"], rtol=1e-9)
        np.testing.assert_allclose(st["err"], info["err"], rtol=RTOL)
        assert np.abs(d_gpu - d_ref).max() <= RTOL * np.abs(d_ref).max()
        np.testing.assert_allclose(nic.deform(P, w, v).cpu().numpy(), orc.deform(Pref, nbr, wt, v), rtol=RTOL, atol=1e-9)


def test_nicp_outer_loop(mods):
    g = golden("nicp_step_24")
    v, edges, tgt, P0 = g["v"], g["edges"], g["tgt"], g["P0"]
    nbr, wt = sparse_from_dense(g["w"])
    run = int(g["nicp_run"])
    # corrected mode vs the oracle's outer loop in the same mode.  With both solvers driven to
    # tight convergence the minimiser is unique -> 1e-6; at leastsq's default stopping tests
    # (ftol = xtol = 1.49e-8 on the ENERGY) each side is itself only determined to ~1e-5.
    tight = dict(ftol=1e-15, xtol=1e-15)
    dd_o, P_o, c_o = orc.nicp(P0.copy(), nbr, wt, v, edges, tgt, 0.01, run, compat=False, **tight)
    hist = []
    dd, P, c = mods["nicp"].nicp(P0, g["w"], v, edges, tgt, 0.01, run, C, compat=False, history=hist,
                                 opts=dict(ftol=1e-15, xtol=1e-15, cg_tol=1e-13))
    assert c == c_o
    np.testing.assert_allclose(dd.cpu().numpy(), dd_o, rtol=RTOL, atol=1e-8)
    dd_d, _, c_d = orc.nicp(P0.copy(), nbr, wt, v, edges, tgt, 0.01, run, compat=False)
    dd_g, _, c_g = mods["nicp"].nicp(P0, g["w"], v, edges, tgt, 0.01, run, C, compat=False)
    assert c_d == c_g
    np.testing.assert_allclose(dd_g.cpu().numpy(), dd_d, rtol=1e-4, atol=1e-5)
    # reference mode vs the reference's own run (golden): schedule identical, vertices close
    n3 = mods["nicp_3d"]
    n3.c_fit, n3.c_rigid, n3.c_smooth = C
    dd2, P2 = n3.NICP(P0, g["w"], v, edges, tgt, 0.01, run)
    np.testing.assert_allclose([n3.c_fit, n3.c_rigid, n3.c_smooth], g["nicp_c"])
    np.testing.assert_allclose(dd2, g["nicp_dd"], rtol=1e-4, atol=1e-4)
    n3.c_fit, n3.c_rigid, n3.c_smooth = C


def test_solve_10k_mesh_properties(mods):
    """BASELINE config C3 size (10k-vertex mesh): size-independent properties -- weights sum
    to 1, Deform(identity) == source, J^T(J p) is symmetric (<Jp,Jq> == <p,J^T J q>), the solve
    decreases the energy and is deterministic run to run."""
    nic, torch = mods["nicp"], mods["torch"]
    v, edges, tgt = orc.make_grid_mesh(100, 100, seed=7)
    n = v.shape[0]
    w = nic.knn_weights(v)
    np.testing.assert_allclose(w.sum(0), 1.0, atol=1e-13)
    P0 = nic.initial_params(n, 3)
    np.testing.assert_allclose(nic.deform(P0, w, v).cpu().numpy(), v, rtol=0, atol=1e-13)
    rng = np.random.default_rng(0)
    P = P0 + 0.01 * rng.standard_normal(P0.shape)
    p, q = rng.standard_normal(P0.shape), rng.standard_normal(P0.shape)
    Jp, Jq = nic.jvp(P, w, v, edges, C, p), nic.jvp(P, w, v, edges, C, q)
    lhs = float((Jp * Jq).sum())
    rhs = float((nic.ctx_to(p) * nic.jtvp(P, w, v, edges, C, Jq)).sum()) if hasattr(nic, "ctx_to") else \
        float((torch.as_tensor(p, device=Jp.device) * nic.jtvp(P, w, v, edges, C, Jq)).sum())
    assert abs(lhs - rhs) <= 1e-9 * abs(lhs)
    idx = c_oracle.match_idx(v, tgt)[0]
    nn = tgt[idx]
    opts = dict(max_lm_iters=3, cg_tol=1e-6)
    P1, st1 = nic.solve(P0, w, v, edges, nn, C, compat=False, opts=opts)
    P2, st2 = nic.solve(P0, w, v, edges, nn, C, compat=False, opts=opts)
    assert st1["err"] < st1["err0"]
    assert torch.equal(P1, P2) and st1 == st2


@pytest.mark.parametrize("kind", ["mesh", "clusters_dups", "plane2d", "line"])
def test_knn_grid_equals_full_scan(mods, kind, monkeypatch):
    """The grid-accelerated KNN (large point sets) returns the same indices and the same fp64 keys
    as the full-scan kernel: meshes, clustered clouds with exact duplicates (ties -> lower index),
    2-D points and a degenerate line."""
    nic = mods["nicp"]
    rng = np.random.default_rng(11)
    if kind == "mesh":
        v = orc.make_grid_mesh(150, 120, seed=3)[0]
    elif kind == "clusters_dups":
        base = np.concatenate([rng.standard_normal((3000, 3)) * 0.02 + c for c in rng.uniform(-3, 3, (5, 3))])
        v = np.concatenate([base, base[rng.permutation(len(base))[:4000]], np.round(base[:3000], 1)])
    elif kind == "plane2d":
        v = rng.uniform(-1, 1, (20000, 2))
    else:
        v = np.zeros((12000, 3))
        v[:, 0] = np.round(rng.uniform(0, 1, 12000), 3)      # many coincident points
    q = np.concatenate([v[::7], v[:50] + 10.0, rng.uniform(v.min(), v.max(), (500, v.shape[1]))])
    for k in (6, 8):
        monkeypatch.setenv("ICP_KNN_GRID_MIN", "1")
        gi, gd = nic.knn(v, q, k)
        monkeypatch.setenv("ICP_KNN_GRID_MIN", "1000000000")
        bi, bd = nic.knn(v, q, k)
        assert np.array_equal(gi.cpu().numpy(), bi.cpu().numpy()), (kind, k)
        assert np.array_equal(gd.cpu().numpy(), bd.cpu().numpy()), (kind, k)
    monkeypatch.setenv("ICP_KNN_GRID_MIN", "1")
    si, sd = nic.knn(v, v, 6)                       # the Initialise case: every point against the set itself
    oi, od = c_oracle.knn_rows(v, 6)
    assert np.array_equal(si.cpu().numpy(), oi) and np.array_equal(sd.cpu().numpy(), od)


@pytest.mark.parametrize("kind", ["strays", "strays_dups", "two_islands", "plane2d_strays"])
def test_knn_far_strays_and_duplicates(mods, kind, monkeypatch):
    """vertices far from the mesh (and exact duplicates of them, and of mesh vertices) leave the
    per-lane pass of the grid KNN after two levels and are finished by the wave-per-query pass (grid_knn_defer_kernel):
    indices AND keys must equal the full scan's bit for bit, k = 6 and 8, as queries against the set itself (the
    Initialise case) and as foreign queries."""
    nic = mods["nicp"]
    rng = np.random.default_rng(23)
    if kind == "plane2d_strays":
        v = np.concatenate([rng.uniform(-1, 1, (30000, 2)), rng.uniform(-1, 1, (12, 2)) * 40.0, np.full((7, 2), 55.0)])
    else:
        v = orc.make_grid_mesh(200, 150, seed=3)[0]
        strays = rng.standard_normal((10, 3)) * 8.0 + 12.0
        if kind == "strays":
            v = np.concatenate([v, strays])
        elif kind == "strays_dups":
            v = np.concatenate([v, strays, strays[:4], strays[:4], v[:500], v[1000:1200]])   # exact ties far out and inside
        else:
            v = np.concatenate([v, v[:4000] * 0.05 + np.array([30.0, -20.0, 10.0])])             # a second, tiny island far away
    q = np.concatenate([v[::11], v[-40:], rng.uniform(v.min(), v.max(), (300, v.shape[1])), v[:20] * 3.0 + 100.0])
    out = {}
    for mode, env in (("grid", "1"), ("scan", "1000000000")):
        monkeypatch.setenv("ICP_KNN_GRID_MIN", env)
        out[mode] = [nic.knn(v, v, 6), nic.knn(v, q, 6), nic.knn(v, q, 8)]
    for (gi, gd), (bi, bd) in zip(out["grid"], out["scan"]):
        assert np.array_equal(gi.cpu().numpy(), bi.cpu().numpy()), kind
        assert np.array_equal(gd.cpu().numpy(), bd.cpu().numpy()), kind
    if kind == "strays":
        oi, od = c_oracle.knn_rows(v, 6)
        assert np.array_equal(out["grid"][0][0].cpu().numpy(), oi) and np.array_equal(out["grid"][0][1].cpu().numpy(), od)


def test_cg_graph_replay_is_bit_identical(mods, monkeypatch):
    """The hipGraph replay of the CG iterations (single GPU) launches the same kernels with the same
    arguments: parameters, energies and iteration counts must not change by a bit."""
    v, edges, tgt = orc.make_grid_mesh(40, 36, seed=11)       # 1440 vertices: above the default threshold
    nic = mods["nicp"]
    w = nic.knn_weights(v)
    nn = tgt[c_oracle.match_idx(v, tgt)[0]]
    P0 = nic.initial_params(len(v), 3)
    opts = dict(max_lm_iters=3, cg_tol=1e-8)
    out = {}
    for mode in ("0", "1", None):
        if mode is None:
            monkeypatch.delenv("ICP_NICP_GRAPH", raising=False)
        else:
            monkeypatch.setenv("ICP_NICP_GRAPH", mode)
        P, st = nic.solve(P0, w, v, edges, nn, C, compat=False, opts=opts)
        out[mode] = (P.cpu().numpy(), st)
    assert out["0"][1]["graph_launches"] == 0
    assert out["1"][1]["graph_launches"] > 0 and out[None][1]["graph_launches"] == out["1"][1]["graph_launches"]
    assert out["1"][1]["cg_iters"] >= 50
    for k in ("err", "err0", "lm_iters", "cg_iters", "nfev", "info"):
        assert out["0"][1][k] == out["1"][1][k] == out[None][1][k]
    assert np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][0], out[None][0])
    # the fused update / direction kernels (default) against the separate sum+logic launches: same bits
    monkeypatch.setenv("ICP_NICP_FUSED", "0")
    for mode in ("0", "1"):
        monkeypatch.setenv("ICP_NICP_GRAPH", mode)
        P, st = nic.solve(P0, w, v, edges, nn, C, compat=False, opts=opts)
        assert np.array_equal(P.cpu().numpy(), out["0"][0])
        assert all(st[k] == out["0"][1][k] for k in ("err", "err0", "lm_iters", "cg_iters", "nfev", "info"))
    monkeypatch.delenv("ICP_NICP_FUSED")
    # small meshes stay on plain launches by default, and can be forced onto the graph
    g = golden("nicp_step_60")
    monkeypatch.delenv("ICP_NICP_GRAPH", raising=False)
    Pa, sa = nic.solve(g["P0"], g["w"], g["v"], g["edges"], g["nn"], C, compat=False)
    monkeypatch.setenv("ICP_NICP_GRAPH", "1")
    Pb, sb = nic.solve(g["P0"], g["w"], g["v"], g["edges"], g["nn"], C, compat=False)
    assert sa["graph_launches"] == 0 and sb["graph_launches"] > 0
    assert np.array_equal(Pa.cpu().numpy(), Pb.cpu().numpy())


@pytest.mark.parametrize("case", ["corrected", "compat", "weighted", "row2d", "row2d_compat"])
def test_fused_operator_is_bit_identical(mods, monkeypatch, case):
    """The fused CG operator (fitrow_kernel + normal_kernel: no J p row vector in memory) forms every row value and
    every sum by the same operations in the same order as jp_kernel -> jtu_kernel (ICP_NICP_FUSEDOP=0): parameters,
    energies and iteration counts must agree bit for bit -- 3-D and 2-D, compat quirks, row weights + landmarks."""
    nic = mods["nicp"]
    rng = np.random.default_rng(5)
    if case.startswith("row2d"):
        x = np.linspace(0, 2 * np.pi, 1500)
        v = np.stack([x, np.sin(x) * (0.5 + x / 6)], 1) @ orc.rot2(0.3) + np.array([2.0, 3.5])
        edges = np.stack([np.arange(1499), np.arange(1, 1500)], 1).astype(np.int32)
        tgt = v + 0.05 * rng.standard_normal(v.shape)
    else:
        v, edges, tgt = orc.make_grid_mesh(40, 36, seed=11)
    w = nic.knn_weights(v)
    nn = tgt[c_oracle.match_idx(v, tgt)[0]]
    P0 = nic.initial_params(len(v), v.shape[1])
    P0 = P0 + 0.01 * rng.standard_normal(P0.shape)          # R != I, A != I: every term of the operator is live
    if case == "weighted":
        w.set_weights(edges, rng.uniform(0.2, 2.0, len(v)), rng.uniform(0.5, 1.5, len(v)), rng.uniform(0.3, 3.0, len(edges)))
        lm = rng.permutation(len(v))[:50].astype(np.int32)
        w.set_landmarks(edges, lm, v[lm] + 0.02, 5.0)
    compat = case.endswith("compat")
    opts = dict(max_lm_iters=3, cg_tol=1e-8, max_cg_iters=400)
    out = {}
    # (J p rows through memory) | (fused operator, direction and fit rows as two kernels) | (default: the direction
    # kernel also forms the next fit rows, recomputing the neighbours' new directions)
    for mode, (fusedop, dirfit) in {"pair": ("0", "0"), "fused": ("1", "0"), "dirfit": ("1", "1")}.items():
        monkeypatch.setenv("ICP_NICP_FUSEDOP", fusedop)
        monkeypatch.setenv("ICP_NICP_DIRFIT", dirfit)
        for coarse in ("1", "10" if v.shape[1] == 3 else "6"):
            monkeypatch.setenv("ICP_NICP_COARSE", coarse)
            for graph in ("0", "1"):
                monkeypatch.setenv("ICP_NICP_GRAPH", graph)
                P, st = nic.solve(P0, w, v, edges, nn, C, compat=compat, opts=opts)
                st.pop("graph_launches")
                out[mode, coarse, graph] = (P.cpu().numpy(), st)
    for coarse in ("1", "10" if v.shape[1] == 3 else "6"):
        a = out["pair", coarse, "0"]
        assert a[1]["cg_iters"] >= 50
        for mode in ("pair", "fused", "dirfit"):
            for graph in ("0", "1"):
                b = out[mode, coarse, graph]
                assert a[1] == b[1], (mode, coarse, graph)
                assert np.array_equal(a[0], b[0]), (mode, coarse, graph)


def test_mesh_beyond_the_old_partial_cap(mods):
    """640 000 vertices (the block-partial buffers used to cap meshes at 2048 blocks = 524 288 vertices):
    <Jp, Jq> == <p, J^T J q>, and a bounded solve lowers the energy, deterministically."""
    torch = mods["torch"]
    nic = mods["nicp"]
    v, edges, tgt = orc.make_grid_mesh(800, 800, seed=3)
    n = len(v)
    w = nic.knn_weights(v)
    P0 = nic.initial_params(n, 3)
    rng = np.random.default_rng(1)
    P = P0 + 0.01 * rng.standard_normal(P0.shape)
    p, q = rng.standard_normal(P0.shape), rng.standard_normal(P0.shape)
    Jp, Jq = nic.jvp(P, w, v, edges, C, p, compat=False), nic.jvp(P, w, v, edges, C, q, compat=False)
    lhs = float((Jp * Jq).sum())
    rhs = float((torch.as_tensor(p, device=Jp.device) * nic.jtvp(P, w, v, edges, C, Jq, compat=False)).sum())
    assert abs(lhs - rhs) <= 1e-9 * abs(lhs)
    from icp_amd import rigid
    nn = rigid.gather_points(w.ctx.to_dev(tgt), rigid.closest_indices(v, tgt))
    opts = dict(max_lm_iters=1, max_cg_iters=100, cg_tol=1e-6)
    P1, st1 = nic.solve(P0, w, v, edges, nn, C, compat=False, opts=opts)
    P2, st2 = nic.solve(P0, w, v, edges, nn, C, compat=False, opts=opts)
    assert st1["err"] < st1["err0"] and st1 == st2 and torch.equal(P1, P2)


@pytest.mark.parametrize("name,coarse", [("nicp_step_60", "10"), ("nicp_step_60", "4"), ("nicp_step_24", "10"), ("nicp_row_20", "6"),
                                         ("nicp_row_20", "3")])
def test_enriched_coarse_space_same_minimiser(mods, monkeypatch, name, coarse):
    """The polynomial-enriched coarse space only changes the preconditioner: the solve must land on the same
    minimiser (scipy's leastsq on the oracle's consistent pair) to 1e-6, here forced onto small meshes."""
    from scipy import optimize
    g = golden(name)
    v, edges, nn, P0 = g["v"], g["edges"], g["nn"], g["P0"]
    dim = v.shape[1]
    nbr, wt = sparse_from_dense(g["w"])
    lsq = optimize.leastsq(lambda p: orc.min_func(p, nbr, wt, v, edges, nn, C, False), P0,
                           Dfun=lambda p: orc.d_min_func(p, nbr, wt, v, edges, nn, C, False), full_output=1,
                           ftol=1e-13, xtol=1e-13)          # all three solvers run to the minimiser itself
    dd_ref = orc.deform(lsq[0], nbr, wt, v)
    nic = mods["nicp"]
    w = nic.SparseWeights(nbr, wt, dim)
    tight = dict(ftol=1e-13, xtol=1e-13)
    monkeypatch.setenv("ICP_NICP_COARSE", "1")
    P1, st1 = nic.solve(P0, w, v, edges, nn, C, compat=False, opts=tight)
    monkeypatch.setenv("ICP_NICP_COARSE", coarse)
    P2, st2 = nic.solve(P0, w, v, edges, nn, C, compat=False, opts=tight)
    print(name, coarse, "plain", st1, "enriched", st2)
    np.testing.assert_allclose(st2["err"], (lsq[2]["fvec"] ** 2).sum(), rtol=1e-6)
    # 1e-6 relative; the absolute floor is 1e-7 of the O(1) mesh size (the 2-D chain has near-zero coordinates and a
    # flat gauge direction along which independent solvers stop a few 1e-8 apart)
    np.testing.assert_allclose(nic.deform(P2, w, v).cpu().numpy(), dd_ref, rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(nic.deform(P2, w, v).cpu().numpy(), nic.deform(P1, w, v).cpu().numpy(), rtol=RTOL, atol=1e-7)


def test_enriched_fused_alpha_stage_is_bit_identical(mods, monkeypatch):
    """Enriched coarse space on a small mesh: the alpha stage folded into the update kernel (default) against the
    six-launch form (ICP_NICP_FUSED=0), with and without the graph replay -- same kernels' arithmetic, same bits."""
    v, edges, tgt = orc.make_grid_mesh(40, 36, seed=11)
    nic = mods["nicp"]
    w = nic.knn_weights(v)
    nn = tgt[c_oracle.match_idx(v, tgt)[0]]
    P0 = nic.initial_params(len(v), 3)
    opts = dict(max_lm_iters=3, cg_tol=1e-8)
    monkeypatch.setenv("ICP_NICP_COARSE", "10")
    out = []
    for fused, graph in (("1", "0"), ("0", "0"), ("1", "1"), ("0", "1")):
        monkeypatch.setenv("ICP_NICP_FUSED", fused)
        monkeypatch.setenv("ICP_NICP_GRAPH", graph)
        P, st = nic.solve(P0, w, v, edges, nn, C, compat=False, opts=opts)
        out.append((P.cpu().numpy(), st))
    for P, st in out[1:]:
        assert np.array_equal(P, out[0][0])
        assert all(st[k] == out[0][1][k] for k in ("err", "err0", "lm_iters", "cg_iters", "nfev", "info"))
    assert out[0][1]["cg_iters"] >= 50
