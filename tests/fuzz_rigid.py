#!/usr/bin/env python3
"""Randomised parity sweep of the rigid path on the GPU against the CPU oracle (test infrastructure; not collected by
pytest -- run by name on a GPU box):  python tests/fuzz_rigid.py [--cases 300] [--seed 0]
Random small clouds (2-D / 3-D, scales 1e-6 .. 1e4, offsets up to 1e7, noise, partial overlap): Match indices exact,
FindTransform and Distance to 1e-6 (transforms compared through what they do to the cloud), and whole ICP runs
(iteration count, error history, final transform, last correspondences) against oracle.icp."""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import icp_oracle as orc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import torch
    from icp_amd import rigid
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        dim = int(rng.choice([2, 3]))
        m = int(rng.integers(4, 1500))
        n = int(rng.integers(4, 1500))
        kind = str(rng.choice(["normal", "uniform", "shell"]))
        tgt = rng.standard_normal((m, dim)) if kind != "uniform" else rng.uniform(-1, 1, (m, dim))
        if kind == "shell":
            tgt /= np.linalg.norm(tgt, axis=1, keepdims=True)
        a = rng.uniform(-0.3, 0.3)
        R = np.eye(dim)
        R[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        src = (tgt[rng.integers(0, m, n)] + rng.standard_normal((n, dim)) * float(rng.choice([0.0, 1e-3, 0.05]))) @ R.T \
            + rng.uniform(-0.2, 0.2, dim)
        scale, off = float(rng.choice([1e-6, 1.0, 1.0, 1e4])), float(rng.choice([0.0, 0.0, 1e3, 1e7]))
        tgt, src = tgt * scale + off * scale, src * scale + off * scale
        span = np.abs(tgt - tgt.mean(0)).max()
        errs = {}
        # Match
        io = orc.match_idx(src, tgt)[0]
        ig = rigid.closest_indices(src, tgt).cpu().numpy()
        errs["idx"] = float((io != ig).sum())
        nn = tgt[io]
        # FindTransform / Distance
        Ro, to = orc.find_transform(src, nn)
        Rg, tg = rigid.find_transform(src, nn)
        Rg, tg = Rg.cpu().numpy(), tg.cpu().numpy()
        errs["kabsch"] = float(np.abs((src @ Rg + tg) - (src @ Ro + to)).max() / span)
        do = orc.distance(src, nn)
        dg = float(rigid.distance_sum(src, nn).cpu()[0])
        errs["dist"] = abs(dg - do) / (abs(do) + 1e-300) if do > 0 else abs(dg)
        # whole ICP (signed stop test, Q3)
        run = int(rng.integers(1, 12))
        tol = float(rng.choice([1e-9, 1e-3, 1e-1])) * span * n
        ho = []
        To = orc.icp(src, tgt, tol, run, history=ho)
        out = rigid.rigid_icp(src, tgt, tol, run, want_idx=True)
        k = int(out["iters"].cpu()[0])
        Tg = out["T"].cpu().numpy()
        errs["iters"] = float(abs(k - len(ho)))
        kk = min(k, len(ho))
        hg = out["err_hist"].cpu().numpy()[:kk]
        hov = np.array([h[1] for h in ho][:kk], dtype=float)
        # Distance is a sum of n norms of differences of points of magnitude |c|, formed through t = tbar - sbar R: numpy
        # knows each term to about 10 eps |c| (the shifted sums of the GPU path sit 10-100x lower, its residual errors
        # are the smaller ones in every case looked at), so that much absolute slack comes before the relative 1e-6
        slack = 100 * n * 2.2e-16 * np.abs(tgt).max()
        errs["hist"] = float(np.maximum(np.abs(hg - hov) - slack, 0.0).max() / (np.abs(hov).max() + 1e-300)) if kk else 0.0
        errs["last_idx"] = float((ho[-1][0] != out["idx"].cpu().numpy()).sum()) if k == len(ho) else 0.0
        errs["T"] = float(np.abs((src @ Tg[:dim, :dim] + Tg[dim, :dim]) - (src @ To[:dim, :dim] + To[dim, :dim])).max() / span)
        tolv = dict(idx=0.5, kabsch=1e-6, dist=1e-9, iters=0.5, hist=1e-6, T=1e-6, last_idx=0.5)
        ok = all(errs[q] < tolv[q] for q in errs)
        bad += 0 if ok else 1
        print("%4d %-5s dim %d n %4d m %4d %-7s scale %g off %g run %d/%d  %s" %
              (case, "ok" if ok else "FAIL", dim, n, m, kind, scale, off, k, run, " ".join("%s %.1e" % kv for kv in errs.items())), flush=True)
        if not ok:
            print("       hist gpu", hg, "oracle", hov, "tol", tolv["hist"], flush=True)
    print("cases %d, failures %d" % (args.cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
