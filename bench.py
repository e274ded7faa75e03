#!/usr/bin/env python3
"""Benchmark of the rigid-ICP hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (config C2): 2^20 source x 2^20 target fp64 points,
tgt = N(0,1)^3 with seed 1, src = ((tgt - shift) Rz(0.3)^T)[perm].  One "step" is one
full ICP iteration: Match (closest point for every source point) + FindTransform
(Kabsch) + apply + Distance, inputs resident in HBM.  N>1 shards the SOURCE rows over
the ranks (target replicated), all-reduces 17 doubles per iteration over RCCL (one
collective) and reports strong scaling (total work fixed).

Prints ONE JSON line (rank 0).  `value` = closest-point queries/s over whole ICP
iterations (= N_src * K / time); `icp_iters_per_s` is the same run in iterations.
`roofline` is for the dominant kernel (the exact grid closest-point search; --brute: the
full-scan kernel, bound by fp64 VALU) -- see DESIGN.md section 3.  `cpu_baseline` times the C oracle (test infrastructure, used
here only as the thing compared against) on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X spec, FMA-rated (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)
HBM_PEAK_GBS = 8000.0


def make_cloud(n, seed=1, angle=0.3, shift=(0.1, 0.05, -0.02)):
    rng = np.random.default_rng(seed)
    tgt = rng.standard_normal((n, 3))
    c, s = np.cos(angle), np.sin(angle)
    rz = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    src = ((tgt - np.array(shift)) @ rz.T)[rng.permutation(n)]
    return np.ascontiguousarray(src), np.ascontiguousarray(tgt)


def cpu_baseline(src, tgt, sample):
    """C oracle (OpenMP) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    threads = min(os.cpu_count() or 1, 16)
    q = src[:sample]
    c_oracle.match_idx(q[:256], tgt, threads=threads)        # warm up / page in
    t0 = time.perf_counter()
    c_oracle.match_idx(q, tgt, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": sample / dt, "unit": "queries/s", "cores": threads, "kind": "port",
            "sample": "%d of the %d source points matched against the full %d-point target by the C oracle "
                      "(gcc -O3 -ffp-contract=off -fopenmp, %d threads); %.1f s" % (sample, src.shape[0], tgt.shape[0], threads, dt),
            "pair_evals_per_s": sample * tgt.shape[0] / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)   # C2: 30 iterations
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=1 << 20, help="source AND target points (default 2^20)")
    ap.add_argument("--cpu-sample", type=int, default=262144, help="queries the CPU baseline matches (about 15-20 s on 16 threads)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--brute", action="store_true", help="force the brute-force closest-point kernel")
    ap.add_argument("--roofline-brute", action="store_true", help="also time one brute-force launch (adds ~1 s)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses GPU 0 and the all-reduce goes over gloo")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(60, exit=True)   # a rehearsal that hangs says where
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            # RCCL builds its communicator at the first collective (seconds): do that here, never inside the
            # timed region, also when the caller asks for --warmup 0
            dist.all_reduce(torch.zeros(17, dtype=torch.float64, device="cuda"))
            torch.cuda.synchronize()

    from icp_amd import rigid, _lib
    from icp_amd._lib import get_context, ptr
    from icp_amd.dist import Comm, GpuOps, ShardedICP, shard_bounds

    ctx = get_context(local_rank)
    n = args.n
    src, tgt = make_cloud(n, seed=1)
    lo, hi = shard_bounds(n, world, rank)
    flags = _lib.ICP_FORCE_BRUTE if args.brute else 0
    K, W = args.steps, args.warmup

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    if not distributed:
        src_d, tgt_d = ctx.to_dev(src), ctx.to_dev(tgt)
        if W > 0:
            rigid.rigid_icp(src_d, tgt_d, 0.0, W, use_tol=False, force_brute=args.brute, ctx=ctx)
        barrier()
        t0 = time.perf_counter()
        res = rigid.rigid_icp(src_d, tgt_d, 0.0, K, use_tol=False, force_brute=args.brute, ctx=ctx)
        barrier()
        dt = time.perf_counter() - t0
        assert int(res["iters"].cpu()[0]) == K
        err_last = float(res["err_hist"].cpu()[K - 1])
    else:
        ops = GpuOps(ctx, force_brute=args.brute)
        st = ShardedICP(ops, Comm(), src[lo:hi], tgt)
        for _ in range(W):
            st.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(K):
            st.step()
        err = st.flush_err()          # the last iteration's error (earlier ones ride on the next step's all-reduce)
        barrier()
        dt = time.perf_counter() - t0
        err_last = float(err.cpu()[0])
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.share_gpu else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.cpu()[0])
        src_d, tgt_d = st.src, st.tgt

    # ---- dominant kernel, timed live with HIP events on the launch stream ----
    # Default path: the exact grid search (grid_query_kernel; the target grid is built once and the
    # queries are in Morton order, as inside icp_rigid_run).  --brute: the full-scan kernel.
    n_loc = src_d.shape[0]
    idx = ctx.empty((n_loc,), torch.int32)

    def time_closest(fl, reps):
        ctx.call("icp_closest", ptr(src_d), n_loc, ptr(tgt_d), n, 3, fl, ptr(idx), ptr(None))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            ctx.call("icp_closest", ptr(src_d), n_loc, ptr(tgt_d), n, 3, fl, ptr(idx), ptr(None))
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) * 1e-3 / reps

    pairs = float(n_loc) * n
    compulsory = 24.0 * n_loc + 24.0 * n + 4.0 * n_loc
    t_brute = time_closest(_lib.ICP_FORCE_BRUTE, 2) if (args.brute or args.roofline_brute) else None
    # per-launch time of the kernel the timed loop spent its time in
    if args.brute:
        t_kernel = t_brute
    else:
        t_kernel, t_lane, t_defer = rigid.time_grid_query(src_d, tgt_d, reps=max(3, min(K, 10)), ctx=ctx, parts=True)
    if args.brute:
        achieved = 8.0 * pairs / t_kernel / 1e12
        roof = {"bound": "valu", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_VECTOR_PEAK_TFLOPS, "traffic": None,
                "kernel": "closest_tiles<3,4,false>", "kernel_ms": t_kernel * 1e3,
                "note": "8 fp64 flop per source-target pair x %.3g pairs per launch; peak = MI355X fp64 VECTOR peak "
                        "(FMA-rated; the reference arithmetic forbids FMA so <= 39.3, and 9 instructions per 8 flop => "
                        "ceiling ~35 TFLOP/s); VALU-bound, compulsory HBM bytes %.1f MB" % (pairs, compulsory / 1e6)}
    else:
        gbytes = 24.0 * n_loc + 4.0 * n_loc + 32.0 * n + 4.0 * (1 << 21)   # queries + idx out + 32-B target records + cell-start array
        achieved = gbytes / t_kernel / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": "grid search = grid_query_kernel<3,false,256,4> + grid_defer_kernel<3,false>", "kernel_ms": t_kernel * 1e3,
                "kernel_ms_parts": {"grid_query_kernel": t_lane * 1e3, "grid_defer_kernel": t_defer * 1e3},
                "note": "exact grid search (two launches: per-lane pass, then one wave per deferred query): algorithmic "
                        "bytes per search = 24 B/query + 4 B/index out + 32 B/target record (Morton-sorted, read at least "
                        "once) + the 2^21-entry cell-start array = %.1f MB; the search is bound by the L1 address/data "
                        "path (TA busy 66-82%%) and fp64 VALU (~50%%) on gathers from the L2/MALL-resident sorted target, "
                        "not by HBM, so the HBM fraction is low by construction; equivalent brute-force rate = %.3g "
                        "pair-evaluations/s" % (gbytes / 1e6, pairs / t_kernel)}
        try:   # HBM-side bytes per launch from the committed PMC passes (profiles/r1_pmc_grid.txt); valid for the default size only
            tr = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
            if n == (1 << 20) and world == 1:
                roof["traffic"] = tr["traffic_bytes"]
                roof["traffic_source"] = tr["source"]
        except Exception:
            pass
        if t_brute is not None:
            ab = 8.0 * pairs / t_brute / 1e12
            roof["bruteforce_kernel"] = {"bound": "valu", "achieved": ab, "peak": FP64_VECTOR_PEAK_TFLOPS,
                                         "unit": "TFLOP/s", "frac": ab / FP64_VECTOR_PEAK_TFLOPS,
                                         "kernel": "closest_tiles<3,4,false>", "kernel_ms": t_brute * 1e3}

    if rank == 0:
        out = {
            "metric": "closest-point queries/s over full ICP iterations (Match+FindTransform+apply+Distance), %s pts" % (
                "1Mx1M" if n == (1 << 20) else "%dx%d" % (n, n)),
            "value": n * K / dt,
            "unit": "queries/s",
            "icp_iters_per_s": K / dt,
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: rigid ICP, %d source x %d target fp64 points (N(0,1)^3, seed 1, Rz(0.3)+shift, permuted), "
                                   "source rows sharded over %d GPU(s), target replicated" % (
                                       "C2" if n == (1 << 20) else ("C4-size" if n == (1 << 24) else "custom"), n, n, world),
                       "n_source": n, "n_target": n, "dim": 3, "search": "brute-force scan" if args.brute else "exact multi-resolution grid (bit-identical to the scan)",
                       "last_err": err_last},
            "roofline": roof,
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(src, tgt, min(args.cpu_sample, n))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
