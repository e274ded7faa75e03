#!/usr/bin/env python3
"""Experiment: CG iteration time on the same mesh with its vertices in the order given (row-major grid) and in Morton
order (permuted on the host before the library sees it).  python benchmarks/morton_probe.py [448] [100]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from nicp_bench import make_grid_mesh
from icp_amd import nicp, rigid, nicp_dist

for nx in [int(a) for a in (sys.argv[1:] or ["448", "100"])]:
    v, edges, tgt = make_grid_mesh(nx, nx, seed=7)
    for order in ("given", "morton"):
        if order == "morton":
            perm = nicp_dist.morton_order(v)
            inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
            v2, e2 = np.ascontiguousarray(v[perm]), inv[edges].astype(np.int32)
        else:
            v2, e2 = v, edges
        w = nicp.knn_weights(v2)
        nn = rigid.gather_points(tgt, rigid.closest_indices(v2, tgt))
        P0 = nicp.initial_params(len(v2), 3)
        pr = nicp.profile_cg(P0, w, v2, e2, nn, (0.1, 1000.0, 100.0), compat=False, iterations=100)
        print(nx, order, "us/iteration %.1f" % pr["us_iteration"], {k: round(x, 1) for k, x in pr["us_kernel"].items()}, flush=True)
