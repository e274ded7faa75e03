#!/usr/bin/env python3
"""How accurately does one nicp_solve (fixed correspondences, C3 mesh, MINPACK's default ftol / xtol) land on the
minimiser, per LM variant?  Reference = the same solve driven to ftol = xtol = 1e-14 with exact steps.
  ICP_NICP_ETA=0.3 ICP_NICP_CONFIRM=1e-2 python benchmarks/accel_accuracy_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from nicp_bench import make_grid_mesh
import torch
from icp_amd import nicp, rigid
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 100
v, edges, tgt = make_grid_mesh(nx, nx, seed=7)
w = nicp.knn_weights(v)
nn = rigid.gather_points(tgt, rigid.closest_indices(v, tgt))
P0 = nicp.initial_params(len(v), 3)
C = (0.1, 1000.0, 100.0)
Pr, sr = nicp.solve(P0, w, v, edges, nn, C, compat=False, opts=dict(ftol=1e-14, xtol=1e-14, cg_tol=1e-10, accel=1))
ddr = nicp.deform(Pr, w, v)
print("reference: err %.12e lm %d cg %d" % (sr["err"], sr["lm_iters"], sr["cg_iters"]))
for name, opts in (("plain LM, default tolerances", dict(accel=-1)), ("accelerated, default", dict(accel=1))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    P, st = nicp.solve(P0, w, v, edges, nn, C, compat=False, opts=opts)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    dd = nicp.deform(P, w, v)
    print("%-30s err %.12e (rel. excess %.2e)  max vertex deviation %.2e  lm %d cg %d  %.2f s" % (
        name, st["err"], (st["err"] - sr["err"]) / sr["err"], float((dd - ddr).abs().max()), st["lm_iters"], st["cg_iters"], dt))
