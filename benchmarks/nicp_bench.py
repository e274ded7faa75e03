#!/usr/bin/env python3
"""NICP timing on the C3-style synthetic mesh (BASELINE.json config 3: 10k-vertex mesh with
the decreasing stiffness schedule).  Not the driver's bench.py; prints one JSON line with
outer-iterations/s and CG-iterations/s.

  python benchmarks/nicp_bench.py [--nx 100 --ny 100] [--run 50] [--fixed] [--lm 0] [--cgtol 0]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_grid_mesh(nx, ny, seed=0):
    """Same recipe as the oracle's generator (kept separate: the bench does not import the oracle)."""
    u = np.linspace(-0.6, 0.6, nx)
    v = np.linspace(-0.6, 0.6, ny)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    rng = np.random.default_rng(seed)
    uu = uu + rng.uniform(-1, 1, uu.shape) * (0.15 * 1.2 / nx)
    vv = vv + rng.uniform(-1, 1, vv.shape) * (0.15 * 1.2 / ny)
    src = np.stack([np.sin(uu) * np.cos(vv), np.sin(vv), np.cos(uu) * np.cos(vv)], -1).reshape(-1, 3)
    nrm = src / np.linalg.norm(src, axis=1, keepdims=True)
    idx = np.arange(nx * ny).reshape(nx, ny)
    edges = np.concatenate([np.stack([idx[:-1, :].ravel(), idx[1:, :].ravel()], 1),
                            np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()], 1),
                            np.stack([idx[:-1, :-1].ravel(), idx[1:, 1:].ravel()], 1)], 0).astype(np.int32)
    tgt = src + 0.05 * np.sin(2 * np.pi * src[:, :1]) * nrm
    sx, cx, sy, cy, sz, cz = np.sin(0.02), np.cos(0.02), np.sin(-0.03), np.cos(-0.03), np.sin(0.05), np.cos(0.05)
    R = np.array([[cy * cz, cy * sz, -sy], [sx * sy * cz - cx * sz, sx * sy * sz + cx * cz, sx * cy],
                  [cx * sy * cz + sx * sz, cx * sy * sz - sx * cz, cx * cy]])
    tgt = tgt @ R + np.array([0.01, -0.02, 0.015])
    return np.ascontiguousarray(src), edges, np.ascontiguousarray(tgt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=100)
    ap.add_argument("--ny", type=int, default=100)
    ap.add_argument("--run", type=int, default=50)
    ap.add_argument("--tol", type=float, default=0.01)
    ap.add_argument("--fixed", action="store_true", help="self-consistent mode instead of reference_compat")
    ap.add_argument("--lm", type=int, default=0, help="cap on LM iterations per solve; 0 = leastsq's own limit only (100*(n+1) evaluations)")
    ap.add_argument("--cgtol", type=float, default=0.0, help="0 = the library default (automatic)")
    args = ap.parse_args()
    import torch
    from icp_amd import nicp
    v, edges, tgt = make_grid_mesh(args.nx, args.ny, seed=7)
    n = v.shape[0]
    t0 = time.perf_counter()
    w = nicp.knn_weights(v)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0
    P0 = nicp.initial_params(n, 3)
    class Progress(list):           # one line per outer iteration so a long run never looks hung
        def append(self, h):
            super().append(h)
            print("outer %d err %.6g c_rigid %.4g lm %d cg %d info %d  t=%.1fs" % (
                len(self) - 1, h[0], h[1], h[3]["lm_iters"], h[3]["cg_iters"], h[3]["info"], time.perf_counter() - t0),
                file=sys.stderr, flush=True)
    hist = Progress()
    opts = dict(max_lm_iters=args.lm, cg_tol=args.cgtol)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dd, P, c = nicp.nicp(P0, w, v, edges, tgt, args.tol, args.run, (0.1, 1000.0, 100.0), compat=not args.fixed,
                         opts=opts, history=hist)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cg = sum(h[3]["cg_iters"] for h in hist)
    lm = sum(h[3]["lm_iters"] for h in hist)
    res = np.abs(dd.cpu().numpy() - tgt).max()
    print(json.dumps({"workload": "C3 NICP %dx%d grid mesh (%d vertices, %d edges), tol %g, run %d, %s" %
                      (args.nx, args.ny, n, len(edges), args.tol, args.run, "self-consistent" if args.fixed else "reference_compat"),
                      "outer_iters": len(hist), "seconds": dt, "outer_iters_per_s": len(hist) / dt, "lm_iters": lm,
                      "cg_iters": cg, "cg_iters_per_s": cg / dt, "init_knn_weights_s": t_init,
                      "err_first": hist[0][0], "err_last": hist[-1][0], "final_c": list(c),
                      "max_abs_dev_from_target": float(res),
                      "lm_cap": args.lm if args.lm > 0 else "none (leastsq's maxfev = 100*(n+1) only)",
                      "stopped_by": {name: sum(1 for h in hist if h[3]["info"] == code) for code, name in
                                     ((1, "ftol"), (2, "xtol"), (3, "ftol+xtol"), (4, "gtol"), (5, "limit"))},
                      "per_outer": [dict(err=h[0], c_rigid=h[1], lm=h[3]["lm_iters"], cg=h[3]["cg_iters"], nfev=h[3]["nfev"],
                                         info=h[3]["info"]) for h in hist]}))


if __name__ == "__main__":
    main()
