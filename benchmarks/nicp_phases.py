#!/usr/bin/env python3
"""Where the wall time of the NICP outer loop goes: wraps the four calls of icp_amd.nicp.nicp (Deform, Match, gather,
solve) with synchronised timers.  python benchmarks/nicp_phases.py [--nx 448 --ny 448] [--fixed]"""
import argparse
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=100)
    ap.add_argument("--ny", type=int, default=100)
    ap.add_argument("--fixed", action="store_true")
    args = ap.parse_args()
    import torch
    from icp_amd import nicp, rigid
    from nicp_bench import make_grid_mesh
    v, edges, tgt = make_grid_mesh(args.nx, args.ny, seed=7)
    w = nicp.knn_weights(v)
    P0 = nicp.initial_params(v.shape[0], 3)
    acc = collections.defaultdict(float)

    def timed(mod, name):
        f = getattr(mod, name)

        def g(*a, **k):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = f(*a, **k)
            torch.cuda.synchronize()
            acc[name] += time.perf_counter() - t
            return r
        setattr(mod, name, g)
    timed(nicp, "deform")
    timed(rigid, "closest_indices")
    timed(rigid, "gather_points")
    timed(nicp, "solve")
    hist = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nicp.nicp(P0, w, v, edges, tgt, 0.01, 50, (0.1, 1000.0, 100.0), compat=not args.fixed, history=hist)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"seconds": dt, "outer": len(hist), "cg": sum(h[3]["cg_iters"] for h in hist),
                      "lm": sum(h[3]["lm_iters"] for h in hist), "nfev": sum(h[3]["nfev"] for h in hist),
                      "phases_s": dict(acc)}))


if __name__ == "__main__":
    main()
