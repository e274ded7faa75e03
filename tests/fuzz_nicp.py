#!/usr/bin/env python3
"""Randomised parity sweep of the NICP operators on the GPU against the CPU oracle (test infrastructure, hence under
tests/; not collected by pytest -- run by name on a GPU box):  python tests/fuzz_nicp.py [--cases 200] [--seed 0]
Random small meshes (3-D and 2-D), random parameter vectors, random stiffness constants, both modes: KNN weights,
Deform, residual, J v, J^T u and one Levenberg-Marquardt step against the dense solve, each to 1e-6 relative."""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import icp_oracle as orc  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--ext", action="store_true", help="the extended energy (per-vertex / per-edge multipliers, landmarks): residual, "
                                                      "J v, J^T u and one LM step against the oracle's extension")
    ap.add_argument("--solve", action="store_true", help="whole solves in the self-consistent mode against scipy's leastsq on the "
                                                        "oracle (energy and deformed vertices to 1e-6) instead of the operators")
    args = ap.parse_args()
    import torch
    from icp_amd import nicp
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        dim = int(rng.choice([2, 3]))
        n = int(rng.integers(8, 140))
        compat = bool(rng.integers(0, 2))
        # a bumpy sheet (3-D) / a wiggly band (2-D): distinct points, no exact KNN ties
        if dim == 3:
            uv = rng.uniform(-1, 1, (n, 2))
            v = np.stack([uv[:, 0], uv[:, 1], 0.3 * np.sin(2 * uv[:, 0]) * np.cos(uv[:, 1])], 1)
        else:
            t = np.sort(rng.uniform(-2, 2, n))
            v = np.stack([t, 0.4 * np.sin(2 * t) + 0.05 * rng.standard_normal(n)], 1)
        v = v * float(rng.choice([0.1, 1.0, 7.0]))
        # edges: every vertex to its 2-3 nearest others, deduplicated
        d = ((v[:, None, :] - v[None, :, :]) ** 2).sum(-1)
        np.fill_diagonal(d, np.inf)
        nb = np.argsort(d, 1)[:, : int(rng.integers(2, 4))]
        edges = np.unique(np.sort(np.stack([np.repeat(np.arange(n), nb.shape[1]), nb.ravel()], 1), 1), axis=0).astype(np.int32)
        S = dim * dim + dim
        ng = 3 if dim == 2 else 6
        P = np.append(np.tile(np.append(np.eye(dim).ravel(), np.zeros(dim)), n), np.zeros(ng))
        P = P + rng.standard_normal(P.shape) * float(rng.choice([0.0, 0.01, 0.2]))
        nn = v + rng.standard_normal(v.shape) * 0.05 * np.abs(v).max()
        c = (float(rng.choice([0.1, 1.0])), float(rng.choice([1000.0, 30.0, 1.0])), float(rng.choice([100.0, 3.0, 0.1])))
        nbr, wt = orc.knn_weights(v)
        w = nicp.knn_weights(v)
        errs = {}
        if args.ext:
            fit_w = rng.uniform(0.2, 3.0, n) if rng.integers(0, 2) else None
            rigid_w = rng.uniform(0.2, 3.0, n) if rng.integers(0, 2) else None
            edge_w = rng.uniform(0.2, 3.0, len(edges)) if rng.integers(0, 2) else None
            nl = int(rng.integers(0, max(2, n // 4)))
            lm_v = rng.choice(n, nl, replace=False).astype(np.int32) if nl else None
            lm_p = (v[lm_v] + rng.standard_normal((nl, dim)) * 0.05 * np.abs(v).max()) if nl else None
            c_land = float(rng.choice([0.5, 5.0, 50.0]))
            w.set_weights(edges, fit_w, rigid_w, edge_w)
            w.set_landmarks(edges, lm_v, lm_p, c_land)
            kw = dict(fit_w=fit_w, rigid_w=rigid_w, edge_w=edge_w, lm_vertex=lm_v, lm_pos=lm_p, c_land=c_land)
            r_o = orc.min_func_ext(P, nbr, wt, v, edges, nn, c, **kw)
            J_o = orc.d_min_func_ext(P, nbr, wt, v, edges, nn, c, **kw)
            errs["r"] = rel(nicp.residual(P, w, v, edges, nn, c, compat=False).cpu().numpy(), r_o)
            pv, pu = rng.standard_normal(P.shape), rng.standard_normal(r_o.shape)
            errs["Jv"] = rel(nicp.jvp(P, w, v, edges, c, pv, compat=False).cpu().numpy(), J_o @ pv)
            errs["JTu"] = rel(nicp.jtvp(P, w, v, edges, c, pu, compat=False).cpu().numpy(), J_o.T @ pu)
            # one LM step (first trial, lam = 1e-3, D = column norms) against the refined dense solve
            P1, st = nicp.solve(P, w, v, edges, nn, c, compat=False, opts=dict(max_lm_iters=1, cg_tol=1e-13, accel=-1))
            col2 = (J_o * J_o).sum(0)
            H = J_o.T @ J_o + 1e-3 * np.diag(np.where(col2 > 0, col2, 1.0))
            g_o = J_o.T @ r_o
            x = np.linalg.solve(H, -g_o)
            for _ in range(4):
                x = x + np.linalg.solve(H, np.asarray((-g_o).astype(np.longdouble) - H.astype(np.longdouble) @ x.astype(np.longdouble),
                                                      dtype=np.float64))
            r1 = orc.min_func_ext(P + x, nbr, wt, v, edges, nn, c, **kw)
            first_trial_accepted = st["nfev"] == 2
            errs["lm_step"] = rel(P1.cpu().numpy() - P, x) if first_trial_accepted and float(r1 @ r1) < float(r_o @ r_o) else 0.0
            ok = all(e < 1e-6 for e in errs.values())
            bad += 0 if ok else 1
            print("%4d %-5s ext dim %d n %3d edges %3d landmarks %2d c %s  %s" % (case, "ok" if ok else "FAIL", dim, n, len(edges), nl, c,
                  " ".join("%s %.1e" % kv for kv in errs.items())), flush=True)
            continue
        if args.solve:
            from scipy import optimize
            if n > 40 or c[1] < 30 or c[2] < 3:      # weakly regularised fits have flat minima: nothing to compare at 1e-6
                continue
            P0 = np.append(np.tile(np.append(np.eye(dim).ravel(), np.zeros(dim)), n), np.zeros(ng))
            lsq = optimize.leastsq(lambda q: orc.min_func(q, nbr, wt, v, edges, nn, c, False), P0,
                                   Dfun=lambda q: orc.d_min_func(q, nbr, wt, v, edges, nn, c, False), full_output=1,
                                   ftol=1e-13, xtol=1e-13, maxfev=3000)
            e_ref = float((lsq[2]["fvec"] ** 2).sum())
            dd_ref = orc.deform(lsq[0], nbr, wt, v)
            Pg, st = nicp.solve(P0, w, v, edges, nn, c, compat=False, opts=dict(ftol=1e-13, xtol=1e-13, max_lm_iters=3000))
            errs["energy"] = abs(st["err"] - e_ref) / (e_ref + 1e-300)
            errs["deformed"] = float(np.abs(nicp.deform(Pg, w, v).cpu().numpy() - dd_ref).max() / np.abs(v).max())
            limit = lsq[4] == 5 or st["info"] == 5      # an evaluation limit on either side: no minimiser to compare
            ok = limit or (errs["energy"] < 1e-6 and errs["deformed"] < 1e-6)
            bad += 0 if ok else 1
            print("%4d %-5s solve dim %d n %3d edges %3d c %s  leastsq ier %d nfev %d | gpu info %d lm %d cg %d  %s" %
                  (case, ("skip" if limit else "ok") if ok else "FAIL", dim, n, len(edges), c, lsq[4], lsq[2]["nfev"], st["info"], st["lm_iters"], st["cg_iters"],
                   " ".join("%s %.1e" % kv for kv in errs.items())), flush=True)
            continue
        errs["w"] = rel(w.toarray(), orc.dense_weights(nbr, wt, n))
        errs["deform"] = rel(nicp.deform(P, w, v).cpu().numpy(), orc.deform(P, nbr, wt, v))
        r_o = orc.min_func(P, nbr, wt, v, edges, nn, c, compat)
        J_o = orc.d_min_func(P, nbr, wt, v, edges, nn, c, compat)
        errs["r"] = rel(nicp.residual(P, w, v, edges, nn, c, compat=compat).cpu().numpy(), r_o)
        pv, pu = rng.standard_normal(P.shape), rng.standard_normal(r_o.shape)
        errs["Jv"] = rel(nicp.jvp(P, w, v, edges, c, pv, compat=compat).cpu().numpy(), J_o @ pv)
        errs["JTu"] = rel(nicp.jtvp(P, w, v, edges, c, pu, compat=compat).cpu().numpy(), J_o.T @ pu)
        # one LM step of the product against the dense twin of its loop
        P1, st = nicp.solve(P, w, v, edges, nn, c, compat=compat, opts=dict(max_lm_iters=1, cg_tol=1e-13, accel=-1))
        tr = []
        P1o = orc.lm_dense(P, nbr, wt, v, edges, nn, c, compat, steps=1, trace=tr)
        P1o = P1o[0] if isinstance(P1o, tuple) else P1o
        errs["lm_step"] = rel(P1.cpu().numpy() - P, np.asarray(P1o) - P) if np.abs(np.asarray(P1o) - P).max() > 0 else 0.0
        note = ""
        if errs["lm_step"] >= 1e-7:
            # the dense solve is itself only good to cond * eps: price the comparison against the system's condition
            col2 = (J_o * J_o).sum(0)
            d2 = np.where(col2 > 0, col2, 1.0)
            H = J_o.T @ J_o + tr[-1]["lam"] * np.diag(d2)
            dsc = 1.0 / np.sqrt(np.diag(H))
            cond = np.linalg.cond(H * dsc[:, None] * dsc[None, :])
            note = " cond(scaled H) %.1e" % cond
            # who is right?  refine the dense solve (residuals in extended precision) and compare both against that
            g_o = J_o.T @ r_o
            Hl, bl = H.astype(np.longdouble), (-g_o).astype(np.longdouble)
            x = np.linalg.solve(H, -g_o)
            for _ in range(5):
                res = np.asarray(bl - Hl @ x.astype(np.longdouble), dtype=np.float64)
                x = x + np.linalg.solve(H, res)
            e_gpu, e_dense = rel(P1.cpu().numpy() - P, x), rel(tr[-1]["delta"], x)
            note += " vs refined solve: gpu %.1e, dense %.1e" % (e_gpu, e_dense)
            if len(tr) == 1 and e_gpu < 1e-6:
                errs["lm_step"] = e_gpu
        ok = all(e < 1e-6 for e in errs.values())
        bad += 0 if ok else 1
        print("%4d %-5s dim %d n %3d edges %3d compat %d c %s  %s" % (case, "ok" if ok else "FAIL", dim, n, len(edges), compat, c,
              " ".join("%s %.1e" % kv for kv in errs.items()) + note), flush=True)
    print("cases %d, failures %d" % (args.cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
