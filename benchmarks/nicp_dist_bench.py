#!/usr/bin/env python3
"""Sharded NICP timing (BASELINE.json config C5: 200k-vertex mesh, vertex ranges after a Morton
reorder).  One rank per GPU under torchrun (backend nccl = RCCL); with --share-gpu every rank
uses GPU 0 and the collectives go over gloo (rehearsal on a one-GPU box).

  python benchmarks/nicp_dist_bench.py --nx 448 --ny 448 --lm 2 --maxcg 300
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 benchmarks/nicp_dist_bench.py ...

A bounded piece of the first outer iteration is timed (--lm LM steps of at most --maxcg CG
iterations each): CG iterations/s is the figure that scales; whole schedules take minutes.
Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=448)
    ap.add_argument("--ny", type=int, default=448)
    ap.add_argument("--lm", type=int, default=2)
    ap.add_argument("--maxcg", type=int, default=300)
    ap.add_argument("--cgtol", type=float, default=1e-6)
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--rccl", action="store_true", help="library-side RCCL (icp_rccl_create): the collectives are issued by the library, no callbacks")
    ap.add_argument("--force", action="store_true", help="ICP_COMM_FORCE: one rank runs the whole sharded code path, every collective issued")
    ap.add_argument("--single", action="store_true", help="one rank through nicp_solve (the single-GPU entry point) for comparison")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from nicp_bench import make_grid_mesh
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    from icp_amd import nicp, nicp_dist, _lib
    ctx = _lib.get_context(local)
    v, edges, tgt = make_grid_mesh(args.nx, args.ny, seed=7)
    n = v.shape[0]
    t0 = time.perf_counter()
    w = nicp.knn_weights(v, ctx)            # replicated: embarrassingly parallel, O(N^2 / rank) if split -- here every rank does all
    nbr, wt = w.nbr.cpu().numpy(), w.wt.cpu().numpy()
    t_init = time.perf_counter() - t0
    t0 = time.perf_counter()
    lib_rccl = _lib.LibraryRccl(ctx, rank=rank, world=world) if args.rccl else None
    sh = nicp_dist.ShardedNICP(v, nbr, wt, edges, tgt, world=world, rank=rank, ctx=ctx, rccl=lib_rccl, force=args.force)
    if args.single:
        class Single:                                   # same timing harness over nicp_solve
            P = ctx.to_dev(nicp.initial_params(n, 3))
            shard = sh.shard
            comm = sh.comm
            match = sh.match
            def solve(self, nn, c, opts):
                Pn, st = nicp.solve(self.P, w, v, edges, nn, c, compat=False, opts=opts, ctx=ctx)
                self.P.copy_(Pn)
                return st
        sh = Single()
    t_part = time.perf_counter() - t0
    nn = sh.match()
    opts = dict(max_lm_iters=args.lm, max_cg_iters=args.maxcg, cg_tol=args.cgtol)
    c = (0.1, 1000.0, 100.0)
    P_keep = sh.P.clone()
    sh.solve(nn, c, dict(max_lm_iters=1, max_cg_iters=20, cg_tol=args.cgtol))      # warm-up (workspace, clocks)
    sh.P.copy_(P_keep)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = sh.solve(nn, c, opts)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tm = torch.tensor([dt], dtype=torch.float64)
        if not args.share_gpu:
            tm = tm.cuda()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt = float(tm.cpu()[0])
    if rank == 0:
        print(json.dumps({"workload": "C5-style NICP %dx%d grid mesh (%d vertices, %d edges), %d rank(s)%s, first outer iteration, "
                                      "%d LM step(s) of <= %d CG iterations" % (args.nx, args.ny, n, len(edges), world,
                                                                               " sharing one GPU over gloo" if args.share_gpu else "",
                                                                               args.lm, args.maxcg),
                          "n_ranks": world, "seconds": dt, "lm_iters": st["lm_iters"], "cg_iters": st["cg_iters"],
                          "cg_iters_per_s": st["cg_iters"] / dt, "us_per_cg_iter": dt / max(st["cg_iters"], 1) * 1e6,
                          "err0": st["err0"], "err": st["err"], "owned": sh.shard.n_owned, "halo": sh.shard.n_halo,
                          "comm_calls": sh.comm.calls, "knn_weights_s": t_init, "partition_s": t_part,
                          "path": "nicp_solve" if args.single else ("nicp_solve_dist, library RCCL" if args.rccl else "nicp_solve_dist, callbacks"),
                          "graph_launches": st["graph_launches"], "forced": bool(args.force)}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
