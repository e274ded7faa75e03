"""Fixtures for the solver path that BASELINE configs C3 / C5 really run  --  TEST INFRASTRUCTURE ONLY.

The dense oracle (icp_oracle.d_min_func / lm_dense) stops at a few hundred vertices.  This script runs the sparse
twin (icp_oracle_sparse.lm_sparse: the same Levenberg-Marquardt loop, every linear system solved by a sparse LU
with iterative refinement through J) on the 2 500- and 10 000-vertex meshes (the latter is the C3 mesh of bench.py:
make_grid_mesh(100, 100, seed=7)) and writes what the GPU tests compare with:

    tests/golden/nicp_large_<n>.npz    per case <mode>_<start>:  the trial history (damping, trial energy,
        accepted?), and after 1 and after 3 accepted steps: the deformed vertices (all), the step delta = P - P_start
        on every 16th unknown + the global parameters, its maximum, and 8 seeded random projections of it.
    tests/golden/nicp_large_solve_2500.npz   the minimiser of the self-consistent energy on the 2 500-vertex mesh
        (a plain Marquardt loop -- damping x10 / :10, nothing shared with the product's rules -- run until the
        gradient vanishes to rounding), as deformed vertices + energy.

One sparse factorisation takes 7 s (2 500 vertices) / 50 s (10 000) on one core, which is why these are fixtures and
not computed inside the GPU tests; tests/test_oracle_sparse_cpu.py re-derives part of the 2 500-vertex file on every
CPU run so that the fixtures cannot drift from the oracle.  Usage:  python oracle/gen_large_mesh.py [2500|10000 [case]|solve|merge 10000]
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import c_oracle  # noqa: E402
import icp_oracle as orc  # noqa: E402
import icp_oracle_sparse as ors  # noqa: E402

C = (0.1, 1000.0, 100.0)
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
STRIDE = 16


def inputs(nx):
    """Mesh, weights, fixed correspondences and the two starting points of a case (shared with the tests)."""
    v, edges, tgt = orc.make_grid_mesh(nx, nx, seed=7)
    n = len(v)
    ki, kd = c_oracle.knn_rows(v, 6)                      # the KNN of Initialise, nicp_3d.py:158-164 (pinned to orc.knn)
    nbr = ki[:, 1:5].astype(np.int32)
    denom = 4 - kd[:, 1:5].sum(1) / kd[:, 5]
    wt = (1 - kd[:, 1:5] / kd[:, 5:6]) / denom[:, None]
    nn = tgt[c_oracle.match_idx(v, tgt)[0]]
    P0 = np.append(np.tile(np.append(np.identity(3).flatten(), np.zeros(3)), n), np.zeros(6))
    Ppert = P0 + 0.01 * np.random.default_rng(0).standard_normal(P0.shape)
    return dict(v=v, edges=edges, tgt=tgt, nbr=nbr, wt=wt, nn=nn, P0=P0, Ppert=Ppert)


def digest(*arrays):
    h = hashlib.sha1()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def projections(nP):
    return np.random.default_rng(12345).standard_normal((8, nP))


def checkpoint(out, key, P, Ps, d):
    delta = P - Ps
    sel = np.concatenate([np.arange(0, len(P) - 6, STRIDE), np.arange(len(P) - 6, len(P))])
    out[key + "_deformed"] = ors.deform_v(P, d["nbr"], d["wt"], d["v"])
    out[key + "_delta_sel"] = delta[sel]
    out[key + "_delta_max"] = np.abs(delta).max()
    out[key + "_delta_norm"] = np.linalg.norm(delta)
    out[key + "_delta_proj"] = projections(len(P)) @ delta


def lm_cases(nx, only=None):
    """only = "<mode>_<start>": that case alone into a part file (the four cases of the 10 000-vertex mesh are run as four
    processes, 3 - 17 minutes each); `merge` joins the parts."""
    d = inputs(nx)
    n = len(d["v"])
    out = dict(nx=nx, n=n, stride=STRIDE, inputs_sha1=digest(d["v"], d["edges"], d["nn"], d["nbr"]),
               meta_numpy=np.__version__)
    # modes: the reference's (compat), the self-consistent one with the plain LM iteration (corrected) and with the
    # accelerated iteration that is its default (accel: second-order correction of every step, nicp_solve_opts::accel)
    for mode, compat, accel in (("compat", True, False), ("corrected", False, False), ("accel", False, True)):
        for start in ("P0", "Ppert"):
            key = "%s_%s" % (mode, start)
            if only and key != only:
                continue
            t0 = time.time()
            tr = []
            P, info = ors.lm_sparse(d[start], d["nbr"], d["wt"], d["v"], d["edges"], d["nn"], C, compat, steps=3,
                                    ftol=0.0, xtol=0.0, trace=tr, accel=accel)
            acc = np.array([np.isfinite(t["f2t"]) and t["ratio"] > 1e-4 for t in tr])
            out[key + "_lam"] = np.array([t["lam"] for t in tr])
            out[key + "_f2t"] = np.array([t["f2t"] for t in tr])
            out[key + "_ratio"] = np.array([t["ratio"] for t in tr])
            out[key + "_accepted"] = acc
            out[key + "_err0"] = info["err0"]
            out[key + "_solve_residual"] = np.array([t["solve_residual"] for t in tr])
            k = 0
            for t, a in zip(tr, acc):
                if a:
                    k += 1
                    if k in (1, 3):
                        checkpoint(out, "%s_s%d" % (key, k), t["x"] + t["delta"], d[start], d)
                        out["%s_s%d_err" % (key, k)] = t["f2t"]
                        out["%s_s%d_nfev" % (key, k)] = 1 + (int(np.nonzero(acc)[0][k - 1]) + 1) * (3 if accel else 1)
            print(key, "trials", len(tr), "accepted", int(acc.sum()), "lam", out[key + "_lam"], "f2t", out[key + "_f2t"],
                  "max solve residual %.1e" % out[key + "_solve_residual"].max(), "%.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(GOLDEN, "nicp_large_%d%s.npz" % (n, "_part_" + only if only else "")), **out)


def merge(n):
    """join the part files (and what nicp_large_<n>.npz already holds) into nicp_large_<n>.npz"""
    out = {}
    whole = os.path.join(GOLDEN, "nicp_large_%d.npz" % n)
    if os.path.exists(whole):
        old = np.load(whole)
        out.update({k: old[k] for k in old.files})
    for key in ("compat_P0", "compat_Ppert", "corrected_P0", "corrected_Ppert", "accel_P0", "accel_Ppert"):
        path = os.path.join(GOLDEN, "nicp_large_%d_part_%s.npz" % (n, key))
        if not os.path.exists(path):
            continue
        part = np.load(path)
        for k in part.files:
            if k in out and not k.startswith(key.split("_")[0]):
                assert np.array_equal(out[k], part[k]), k
            out[k] = part[k]
        os.remove(path)
    np.savez_compressed(os.path.join(GOLDEN, "nicp_large_%d.npz" % n), **out)


def whole_solve(nx=50):
    """Minimiser of the self-consistent energy: Marquardt's original rule (lam x10 on failure, :10 on success) on
    the sparse normal equations, until the scaled gradient is at rounding level."""
    d = inputs(nx)
    n = len(d["v"])
    a = (d["nbr"], d["wt"], d["v"], d["edges"], d["nn"], C, False)
    x = d["P0"].copy()
    r = ors.min_func_v(x, *a)
    f2 = float(r @ r)
    lam = 1e-3
    hist = []
    for it in range(400):
        J = ors.d_min_func_sparse(x, *a)
        g = J.T @ r
        d2 = np.asarray(J.multiply(J).sum(0)).reshape(-1)
        solver = ors.NormalSolver(J, d["v"])
        while True:
            delta = solver.solve(lam, d2, -g)
            rt = ors.min_func_v(x + delta, *a)
            f2t = float(rt @ rt)
            if f2t < f2:
                break
            lam *= 10.0
            if lam > 1e12:
                break
        rel = (f2 - f2t) / f2
        step = np.abs(delta).max()
        hist.append((f2t, lam, step))
        print(it, "f2 %.15e lam %.1e rel %.2e step %.2e |g| %.2e" % (f2t, lam, rel, step, np.abs(g).max()), flush=True)
        if f2t < f2:
            x, r, f2 = x + delta, rt, f2t
            lam = max(lam / 10.0, 1e-12)
        if step < 1e-11 or lam > 1e12:
            break
    np.savez_compressed(os.path.join(GOLDEN, "nicp_large_solve_%d.npz" % n), nx=nx, n=n, err=f2,
                        deformed=ors.deform_v(x, d["nbr"], d["wt"], d["v"]), hist=np.array(hist),
                        inputs_sha1=digest(d["v"], d["edges"], d["nn"], d["nbr"]), meta_numpy=np.__version__)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "2500"
    if what == "solve":
        whole_solve()
    elif what == "merge":
        merge(int(sys.argv[2]))
    else:
        lm_cases({"2500": 50, "10000": 100}[what], sys.argv[2] if len(sys.argv) > 2 else None)
