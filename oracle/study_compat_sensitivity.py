#!/usr/bin/env python3
"""How determinate is the END POINT of the reference's `optimize.leastsq` call (nicp_3d.py:351,
nicp_row.py:246) in the reference-as-coded ("compat") mode?  Test infrastructure / evidence only.

In compat mode D_MinFunc is not the derivative of MinFunc (quirks Q5/Q6/Q7) and J has a
6-dimensional (3 in 2-D) gauge null space, so MINPACK's lmder stops where its trust region collapses
(ier = 2).  This script runs scipy's leastsq on the oracle's (MinFunc, D_MinFunc) pair -- pinned to the
reference at 1e-12 by tests/golden -- from the fixture's P0 and from P0 + eps * N(0,1) for eps down to
1e-15 (one ulp of the unit entries of P0) and prints how far the deformed vertices and the energy move.
If an input perturbation of 1e-15 moves the reference's own answer by more than 1e-6, no independent
solver can be held to 1e-6 of that answer: the tests then pin what IS determinate (operators, every LM
step's linear solve: tests/test_gpu_nicp.py::test_lm_steps_match_dense_solves) at 1e-6 and the end
point at the measured floor.

Usage: python oracle/study_compat_sensitivity.py > profiles/r2_compat_sensitivity.txt
"""
import os
import sys

import numpy as np
import scipy
from scipy import optimize

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import icp_oracle as orc  # noqa: E402

C = (0.1, 1000.0, 100.0)


def sparse(w):
    n = w.shape[0]
    nbr = np.zeros((n, 4), np.int64)
    wt = np.zeros((n, 4))
    for i in range(n):
        js = np.nonzero(w[:, i])[0]
        nbr[i] = js
        wt[i] = w[js, i]
    return nbr, wt


def main():
    print("numpy", np.__version__, "scipy", scipy.__version__)
    print("columns: fixture mode eps | max rel. move of the deformed vertices | max rel. move of the energy (4 draws)")
    for name in ("nicp_step_24", "nicp_step_60", "nicp_row_20"):
        g = np.load(os.path.join(HERE, "..", "tests", "golden", name + ".npz"))
        v, edges, nn, P0 = g["v"], g["edges"].astype(np.int64), g["nn"], g["P0"]
        nbr, wt = sparse(g["w"])
        for compat in (True, False):
            f = lambda p: orc.min_func(p, nbr, wt, v, edges, nn, C, compat)
            J = lambda p: orc.d_min_func(p, nbr, wt, v, edges, nn, C, compat)
            base = optimize.leastsq(f, P0, Dfun=J, full_output=1)
            d0, e0 = orc.deform(base[0], nbr, wt, v), (base[2]["fvec"] ** 2).sum()
            mode = "compat" if compat else "corrected"
            if compat:
                print("%-13s %-9s oracle-vs-reference(golden): deformed %.2e (rel), energy %.2e (rel), ier %d, nfev %d"
                      % (name, mode, np.abs(d0 - g["lsq_deformed"]).max() / np.abs(d0).max(),
                         abs(e0 - float(g["lsq_err"])) / e0, base[4], base[2]["nfev"]))
            rng = np.random.default_rng(0)
            for eps in (1e-15, 1e-13, 1e-11):
                dv, de = [], []
                for _ in range(4):
                    l = optimize.leastsq(f, P0 + eps * rng.standard_normal(P0.shape), Dfun=J, full_output=1)
                    dv.append(np.abs(orc.deform(l[0], nbr, wt, v) - d0).max() / np.abs(d0).max())
                    de.append(abs((l[2]["fvec"] ** 2).sum() - e0) / e0)
                print("%-13s %-9s eps %.0e | %.2e | %.2e" % (name, mode, eps, max(dv), max(de)))


if __name__ == "__main__":
    main()
