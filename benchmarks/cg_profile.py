#!/usr/bin/env python3
"""Live HIP-event timing of the NICP CG iteration (nicp_profile_cg) at C3 size (10k vertices) and C5 size (200k):
per-launch microseconds, the iteration as the solver runs it, and roofline fractions.
  python benchmarks/cg_profile.py [--sizes 100,448] [--iters 100]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK = 8.0e12


def kernel_bytes(n, e):
    """bytes each kernel of the iteration must move once (3-D), per launch"""
    return {"fitrow_kernel": n * 216.0, "normal_kernel": n * 488.0 + e * 16.0, "cg_update_prec_kernel": n * 1104.0,
            "cg_dir_prec_kernel": n * 408.0}


def profile(nx, ny, iters, coarse=None, ctx=None):
    import torch
    from nicp_bench import make_grid_mesh
    from icp_amd import nicp, rigid
    v, edges, tgt = make_grid_mesh(nx, ny, seed=7)
    w = nicp.knn_weights(v, ctx)
    P0 = nicp.initial_params(len(v), 3)
    nn = rigid.gather_points(tgt, rigid.closest_indices(v, tgt, ctx=ctx), ctx=ctx)
    if coarse:
        os.environ["ICP_NICP_COARSE"] = str(coarse)
    try:
        pr = nicp.profile_cg(P0, w, v, edges, nn, (0.1, 1000.0, 100.0), compat=False, iterations=iters, ctx=ctx)
    finally:
        os.environ.pop("ICP_NICP_COARSE", None)
    n, e = pr["vertices"], pr["edges"]
    kb = kernel_bytes(n, e)
    alg = 860.0 * n                               # 0.86 kB per vertex per CG iteration
    t = pr["us_iteration"] * 1e-6
    pr["roofline"] = {"bound": "hbm", "algorithmic_bytes_per_iteration": alg, "achieved_GBs": alg / t / 1e9, "peak_GBs": HBM_PEAK / 1e9,
                      "frac": alg / t / HBM_PEAK,
                      "kernels": {k: {"us": pr["us_kernel"][k], "minimal_bytes": b, "GBs": b / (pr["us_kernel"][k] * 1e-6) / 1e9,
                                      "frac": b / (pr["us_kernel"][k] * 1e-6) / HBM_PEAK} for k, b in kb.items() if pr["us_kernel"][k] > 0}}
    return pr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="100,448")
    ap.add_argument("--iters", type=int, default=100)
    args = ap.parse_args()
    for nx in [int(x) for x in args.sizes.split(",")]:
        for coarse in (1, 10):
            print(json.dumps(dict(mesh="%dx%d" % (nx, nx), coarse=coarse, **profile(nx, nx, args.iters, coarse))))


if __name__ == "__main__":
    main()
