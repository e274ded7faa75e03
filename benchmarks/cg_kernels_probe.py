#!/usr/bin/env python3
"""Per-launch times of the CG iteration (nicp_profile_cg) at a few mesh sizes: python benchmarks/cg_kernels_probe.py [448 100]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from nicp_bench import make_grid_mesh
from icp_amd import nicp, rigid
for nx in [int(a) for a in (sys.argv[1:] or ["448", "100"])]:
    v, edges, tgt = make_grid_mesh(nx, nx, seed=7)
    w = nicp.knn_weights(v)
    nn = rigid.gather_points(tgt, rigid.closest_indices(v, tgt))
    P0 = nicp.initial_params(len(v), 3)
    pr = nicp.profile_cg(P0, w, v, edges, nn, (0.1, 1000.0, 100.0), compat=False, iterations=100)
    print(nx, "us/iteration %.1f" % pr["us_iteration"], {k: round(x, 1) for k, x in pr["us_kernel"].items()}, flush=True)
