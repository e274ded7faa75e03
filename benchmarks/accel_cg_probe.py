#!/usr/bin/env python3
"""One accelerated LM step on the C3 mesh from P0 at several CG tolerances: energy after the step, CG iterations
(oracle, sparse direct solves: tests/golden/nicp_large_10000.npz accel_P0_s1_err)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_large_mesh as glm
from icp_amd import nicp
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 100
d = glm.inputs(nx)
g = np.load(os.path.join(ROOT, "tests", "golden", "nicp_large_%d.npz" % len(d["v"])))
print("oracle err after the step: %.12e" % float(g["accel_P0_s1_err"]))
w = nicp.knn_weights(d["v"])
for tol in (1e-2, 1e-4, 1e-6, 1e-8, 1e-10, 1e-13):
    P, st = nicp.solve(d["P0"], w, d["v"], d["edges"], d["nn"], (0.1, 1000.0, 100.0), compat=False,
                       opts=dict(max_lm_iters=1, cg_tol=tol, ftol=0.0, xtol=0.0, accel=1, max_cg_iters=60000))
    print("cg_tol %.0e  err %.12e  cg %d" % (tol, st["err"], st["cg_iters"]), flush=True)
