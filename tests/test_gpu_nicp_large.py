"""The solver path that BASELINE configs C3 / C5 really run -- hipGraph replay of the CG chunks, the adaptive
(polynomial-enriched) coarse space, the lagged coarse matrix, the XCD block map, the automatic switches by mesh
size -- against the oracle, in the DEFAULT environment.

The oracle side is oracle/icp_oracle_sparse.lm_sparse (sparse Jacobian of D_MinFunc, nicp_3d.py:274-338, pinned to
the reference-generated fixtures at 1e-12 by tests/test_oracle_sparse_cpu.py; every Levenberg-Marquardt system
solved by a sparse LU and refined through J).  One factorisation takes 7 s at 2 500 vertices and 50 s at 10 000,
so its results are committed as fixtures by oracle/gen_large_mesh.py (tests/golden/nicp_large_*.npz); the CPU suite
re-derives part of them so they cannot drift.  Bar: 1e-6 relative (north_star) on the step and on the deformed
vertices, same accept / reject sequence, same energies."""
import os
import sys

import numpy as np
import pytest

from conftest import golden, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_large_mesh as glm  # noqa: E402  (inputs() only: the shared recipe of the cases)

pytestmark = pytest.mark.gpu
RTOL = 1e-6
C = (0.1, 1000.0, 100.0)
SWITCHES = ("ICP_NICP_GRAPH", "ICP_NICP_COARSE", "ICP_NICP_FUSED", "ICP_NICP_FUSEDOP", "ICP_NICP_ELAG", "ICP_NICP_RUN",
            "ICP_NICP_CHUNK", "ICP_NICP_PREC32", "ICP_NICP_STOP", "ICP_NICP_DIRFIT")


@pytest.fixture(scope="module")
def nic():
    import torch
    assert torch.cuda.is_available()
    from icp_amd import nicp
    return nicp


_inputs = {}


def case_inputs(nx):
    if nx not in _inputs:
        _inputs[nx] = glm.inputs(nx)
    return _inputs[nx]


@pytest.mark.parametrize("nx", [50, 100])
@pytest.mark.parametrize("mode", ["compat", "corrected", "accel"])
@pytest.mark.parametrize("start", ["P0", "Ppert"])
def test_lm_steps_on_large_meshes_match_sparse_direct_solves(nic, monkeypatch, nx, mode, start):
    for k in SWITCHES:                                      # the DEFAULT path: no switch may be set
        monkeypatch.delenv(k, raising=False)
    d = case_inputs(nx)
    n = len(d["v"])
    g = golden("nicp_large_%d" % n)
    assert str(g["inputs_sha1"]) == glm.digest(d["v"], d["edges"], d["nn"], d["nbr"]), "fixture made from other inputs"
    key = "%s_%s" % (mode, start)
    w = nic.knn_weights(d["v"])
    assert np.array_equal(w.nbr.cpu().numpy(), d["nbr"])   # same neighbours as the oracle's Initialise
    Ps = d[start]
    nP = len(Ps)
    sel = np.concatenate([np.arange(0, nP - 6, int(g["stride"])), np.arange(nP - 6, nP)])
    proj = glm.projections(nP)
    checked = 0
    for steps in (1, 3):
        ck = "%s_s%d" % (key, steps)
        if ck + "_err" not in g.files:                      # the oracle's own run did not get that far
            continue
        # a step accepted only after the damping ran away (compat mode from a perturbed P: the trial step collapses and
        # accept / reject is decided by the rounding noise of the energy) is not a determinate quantity -- DESIGN.md
        # 'NICP solve parity'; such checkpoints are left out
        # (modes: compat / corrected = the plain LM iteration, accel = the accelerated one that the self-consistent mode
        # runs by default, each against lm_sparse in the same form)
        upto = np.nonzero(g[key + "_accepted"])[0][steps - 1]
        if g[key + "_lam"][:upto + 1].max() > 1e8:
            continue
        P, st = nic.solve(Ps, w, d["v"], d["edges"], d["nn"], C, compat=(mode == "compat"),
                          opts=dict(max_lm_iters=steps, cg_tol=1e-13, ftol=0.0, xtol=0.0, accel=1 if mode == "accel" else -1))
        if n >= 1024:
            assert st["graph_launches"] > 0                 # it really was the graph-replay path
        delta = P.cpu().numpy() - Ps
        dmax = float(g[ck + "_delta_max"])
        e_sel = np.abs(delta[sel] - g[ck + "_delta_sel"]).max() / dmax
        e_proj = np.abs(proj @ delta - g[ck + "_delta_proj"]).max() / (float(g[ck + "_delta_norm"]) * np.sqrt(nP))
        dd = nic.deform(P, w, d["v"]).cpu().numpy()
        e_dd = np.abs(dd - g[ck + "_deformed"]).max() / np.abs(g[ck + "_deformed"]).max()
        print(n, key, steps, "gpu", st, "oracle err", float(g[ck + "_err"]), "nfev", int(g[ck + "_nfev"]),
              "| step %.1e proj %.1e deformed %.1e" % (e_sel, e_proj, e_dd))
        assert st["lm_iters"] == steps and st["nfev"] == int(g[ck + "_nfev"])       # same accept / reject sequence
        np.testing.assert_allclose(st["err0"], float(g[key + "_err0"]), rtol=1e-9)
        np.testing.assert_allclose(st["err"], float(g[ck + "_err"]), rtol=RTOL)
        assert abs(np.abs(delta).max() - dmax) <= RTOL * dmax
        assert e_sel <= RTOL and e_proj <= RTOL and e_dd <= RTOL
        checked += 1
    assert checked >= 1


def test_whole_solve_2500_lands_on_the_minimiser(nic, monkeypatch):
    """One whole self-consistent solve on the default path against the minimiser found by an unrelated loop
    (Marquardt's x10 / :10 rule on sparse direct solves, run until the step vanishes): deformed vertices to 1e-6."""
    for k in SWITCHES:
        monkeypatch.delenv(k, raising=False)
    d = case_inputs(50)
    g = golden("nicp_large_solve_2500")
    assert str(g["inputs_sha1"]) == glm.digest(d["v"], d["edges"], d["nn"], d["nbr"])
    w = nic.knn_weights(d["v"])
    P, st = nic.solve(d["P0"], w, d["v"], d["edges"], d["nn"], C, compat=False, opts=dict(ftol=1e-15, xtol=1e-15))
    dd = nic.deform(P, w, d["v"]).cpu().numpy()
    print("whole solve 2500:", st, "oracle err", float(g["err"]), "max |dd - ref|", np.abs(dd - g["deformed"]).max())
    assert st["graph_launches"] > 0
    np.testing.assert_allclose(st["err"], float(g["err"]), rtol=RTOL)
    np.testing.assert_allclose(dd, g["deformed"], rtol=RTOL, atol=1e-7)
