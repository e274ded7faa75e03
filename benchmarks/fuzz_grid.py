#!/usr/bin/env python3
"""Randomised parity sweep of the exact grid search against the full scan (both on the GPU, bit-exact indices AND d2),
for Match (icp_closest) and KNN (nicp_knn): sizes, dimensions, distributions, scales, offsets, ties, duplicates,
hints, the rounded-sqrt comparison.  Not part of the test suite (minutes); prints one line per case and a summary.
  python benchmarks/fuzz_grid.py [--cases 200] [--seed 0] [--max-n 200000]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cloud(rng, n, dim, kind):
    if kind == "uniform":
        p = rng.uniform(-1, 1, (n, dim))
    elif kind == "normal":
        p = rng.standard_normal((n, dim))
    elif kind == "clusters":
        k = int(rng.integers(1, 12))
        c = rng.uniform(-5, 5, (k, dim))
        p = c[rng.integers(0, k, n)] + rng.standard_normal((n, dim)) * rng.choice([1e-3, 0.05, 0.5])
    elif kind == "lattice":      # exact ties by construction
        s = int(max(2, round(n ** (1.0 / dim))))
        p = rng.integers(0, s, (n, dim)).astype(np.float64) * 0.5
    elif kind == "line":
        t = rng.uniform(-3, 3, (n, 1))
        d = rng.standard_normal((1, dim))
        p = t * d + rng.standard_normal((n, dim)) * rng.choice([0.0, 1e-9, 1e-3])
    elif kind == "plane":
        p = rng.uniform(-1, 1, (n, dim))
        p[:, -1] = rng.choice([0.0, 0.25])
    elif kind == "dups":
        base = rng.standard_normal((max(1, n // 7), dim))
        p = base[rng.integers(0, len(base), n)]
    elif kind == "outliers":
        p = rng.standard_normal((n, dim))
        k = max(1, n // 50)
        p[rng.integers(0, n, k)] *= rng.choice([30.0, 1e3, 1e6])
    else:
        raise ValueError(kind)
    return p


KINDS = ["uniform", "normal", "clusters", "lattice", "line", "plane", "dups", "outliers"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-n", type=int, default=200000)
    ap.add_argument("--loop", action="store_true", help="fuzz the fused ICP loop (icp_rigid_run: grid, sorted source, positions, "
                                                       "hints) against the same loop on the full scan instead")
    ap.add_argument("--coherent", action="store_true", help="fuzz the coherent loop (ICP_COHERENT: rows whose match is provably "
                                                           "unchanged skip the search) against the plain fused loop: T, "
                                                           "correspondences and errors bit for bit, ties and duplicates included")
    args = ap.parse_args()
    import torch
    from icp_amd import rigid, nicp
    rng = np.random.default_rng(args.seed)
    bad = 0
    t0 = time.perf_counter()
    for case in range(args.cases if args.coherent else 0):
        dim = int(rng.choice([2, 3]))
        m = int(np.exp(rng.uniform(np.log(8), np.log(args.max_n))))
        n = int(np.exp(rng.uniform(np.log(8), np.log(args.max_n))))
        kt = str(rng.choice(KINDS))
        tgt = cloud(rng, m, dim, kt)
        a = rng.uniform(-0.2, 0.2)
        R = np.eye(dim)
        R[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        src = (tgt[rng.integers(0, m, n)] + rng.standard_normal((n, dim)) * rng.choice([0.0, 1e-3, 0.05])) @ R.T + rng.uniform(-0.1, 0.1, dim)
        if rng.integers(0, 6) == 0:
            src[rng.integers(0, n, max(1, n // 50))] = rng.choice([np.nan, np.inf, 1e9])   # rows no bound can hold
        run = int(rng.integers(2, 41))
        tol = float(rng.choice([0.0, 1e-6, 1e-3]))
        use_tol = bool(rng.integers(0, 2))
        A = rigid.rigid_icp(src, tgt, tol, run, use_tol=use_tol, force_grid=True, want_idx=True)
        B = rigid.rigid_icp(src, tgt, tol, run, use_tol=use_tol, force_grid=True, want_idx=True, coherent=True)
        k = int(A["iters"].cpu()[0])
        bits = lambda t: t.contiguous().view(torch.int64)   # (NaN rows make T and the errors NaN: compare the bit patterns)
        same = (k == int(B["iters"].cpu()[0]) and torch.equal(bits(A["T"]), bits(B["T"])) and torch.equal(A["idx"], B["idx"])
                and torch.equal(bits(A["err_hist"][:k]), bits(B["err_hist"][:k])))
        bad += 0 if same else 1
        print("%4d %-5s coherent dim %d n %7d m %7d tgt %-8s run %2d (%2d done) tol %g/%d  idx_diff %d" %
              (case, "ok" if same else "FAIL", dim, n, m, kt, run, k, tol, use_tol, int((A["idx"] != B["idx"]).sum())), flush=True)
    for case in range(args.cases if (args.loop and not args.coherent) else 0):
        # tie-free distributions only: the two loops sum in different orders, so their transforms differ in the last
        # bits, and an exact tie may then legitimately resolve differently
        dim = int(rng.choice([2, 3]))
        m = int(np.exp(rng.uniform(np.log(8), np.log(args.max_n))))
        n = int(np.exp(rng.uniform(np.log(8), np.log(args.max_n))))
        kt = str(rng.choice(["uniform", "normal", "clusters", "outliers"]))
        tgt = cloud(rng, m, dim, kt)
        a = rng.uniform(-0.2, 0.2)
        R = np.eye(dim)
        R[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        src = (tgt[rng.integers(0, m, n)] + rng.standard_normal((n, dim)) * rng.choice([0.0, 1e-3, 0.05])) @ R.T + rng.uniform(-0.1, 0.1, dim)
        run = int(rng.integers(1, 9))
        # (the same loop on the full scan sums in another order, and on ill-conditioned clouds -- a target that is
        # one tight cluster -- its transforms then drift apart for reasons that have nothing to do with the search;
        # so the check is path-consistent instead, like bench.py's config.check: the correspondences of iteration K
        # of the fused loop must be the full scan's answer for the positions its own iteration K-1 produced)
        sq = bool(rng.integers(0, 4) == 0)     # the C++ command's comparison of rounded roots (two-pass search)
        B = rigid.rigid_icp(src, tgt, 0.0, run, use_tol=False, force_grid=True, want_idx=True, compare_sqrt=sq)
        if run > 1:
            prev = rigid.rigid_icp(src, tgt, 0.0, run - 1, use_tol=False, force_grid=True, compare_sqrt=sq)["T"]
            ss = rigid.apply_transform(src, prev[:dim, :dim].contiguous(), prev[dim, :dim].contiguous())
        else:
            ss = src
        ref = rigid.closest_indices(ss, tgt, force_brute=True, compare_sqrt=sq)
        nidx = int((ref != B["idx"]).sum())
        again = rigid.rigid_icp(src, tgt, 0.0, run, use_tol=False, force_grid=True, want_idx=True, compare_sqrt=sq)
        same = torch.equal(again["T"], B["T"]) and torch.equal(again["idx"], B["idx"]) and torch.equal(again["err_hist"], B["err_hist"])
        ok = nidx == 0 and same
        bad += 0 if ok else 1
        print("%4d %-5s loop dim %d n %7d m %7d tgt %-8s run %d sqrt %d  idx_diff_vs_scan %d reproducible %d" %
              (case, "ok" if ok else "FAIL", dim, n, m, kt, run, sq, nidx, same), flush=True)
    for case in range(0 if (args.loop or args.coherent) else args.cases):
        dim = int(rng.choice([2, 3]))
        n = int(np.exp(rng.uniform(0, np.log(args.max_n))))
        m = int(np.exp(rng.uniform(0, np.log(args.max_n))))
        kt, ks = str(rng.choice(KINDS)), str(rng.choice(KINDS + ["moved"]))
        tgt = cloud(rng, m, dim, kt)
        if ks == "moved":   # the ICP situation: the source is the target, moved a little, subsampled
            src = tgt[rng.integers(0, m, n)] + rng.standard_normal((n, dim)) * rng.choice([0.0, 1e-12, 1e-3, 0.1])
        else:
            src = cloud(rng, n, dim, ks)
        scale, off = float(rng.choice([1e-6, 1.0, 1.0, 1e4])), float(rng.choice([0.0, 0.0, 1e3, -1e7]))
        tgt, src = tgt * scale + off, src * scale + off
        sq = bool(rng.integers(0, 4) == 0)
        what = "match"
        try:
            if rng.integers(0, 5) == 0 and m >= 8:
                what = "knn"
                k = int(rng.choice([6, 8]))
                q = src[: min(n, 20000)]
                os.environ["ICP_KNN_GRID_MIN"] = "1000000000"    # the full scan
                ia, da = nicp.knn(tgt, q, k)
                os.environ["ICP_KNN_GRID_MIN"] = "1"             # the grid search
                ib, db = nicp.knn(tgt, q, k)
                os.environ.pop("ICP_KNN_GRID_MIN")
                ok = torch.equal(ia, ib) and torch.equal(da, db)
            else:
                ia, da = rigid.closest_indices(src, tgt, compare_sqrt=sq, want_d2=True, force_brute=True)
                hint = None
                hk = int(rng.integers(0, 3))
                if hk == 1:
                    hint = rng.integers(0, m, n).astype(np.int32)
                elif hk == 2:
                    hint = ia.cpu().numpy().copy()
                    hint[rng.integers(0, n, max(1, n // 10))] = -5
                ib, db = rigid.closest_indices(src, tgt, compare_sqrt=sq, want_d2=True, force_grid=True, hint=hint)
                ok = torch.equal(ia, ib) and torch.equal(da, db)
                what += " sqrt" if sq else ""
                what += " hint%d" % hk
        except Exception as e:   # a refusal is fine only if both paths refuse; report it
            ok = False
            what += " EXC %r" % (e,)
        bad += 0 if ok else 1
        print("%4d %-5s dim %d n %7d m %7d tgt %-8s src %-8s scale %g off %g  %s" %
              (case, "ok" if ok else "FAIL", dim, n, m, kt, ks, scale, off, what), flush=True)
    print("cases %d, failures %d, %.0f s" % (args.cases, bad, time.perf_counter() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
