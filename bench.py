#!/usr/bin/env python3
"""Benchmark of the rigid-ICP hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (config C2): 2^20 source x 2^20 target fp64 points,
tgt = N(0,1)^3 with seed 1, src = ((tgt - shift) Rz(0.3)^T)[perm].  One "step" is one
full ICP iteration: Match (closest point for every source point) + FindTransform
(Kabsch) + apply + Distance, inputs resident in HBM.  N>1 shards the SOURCE rows over
the ranks (target replicated), all-reduces 17 doubles per iteration over RCCL (one
collective, issued from inside icp_rigid_run_dist) and reports strong scaling (total work
fixed).  On every N the timed region is ONE library call that runs K iterations and
includes the one-time set-up (target grid build + Morton sort of the source shard).

Prints ONE JSON line (rank 0).  `value` = closest-point queries/s over whole ICP
iterations (= N_src * K / time) of the first timed call; `extra.repeats` holds further
identical calls (median / min / max), `extra.setup_ms` the one-time part of a call.
`config.check` ties the timed run to the full scan: the correspondences the timed call
returned are compared with an ICP_FORCE_BRUTE Match of the same final source positions.
`roofline` is for the dominant kernel (the exact grid closest-point search; --brute: the
full-scan kernel, bound by fp64 VALU; its roofline is always given as
roofline.bruteforce_kernel) -- see DESIGN.md section 3.  `extra.nicp_c3` runs config C3
(10k-vertex NICP with the stiffness schedule) in the reference's mode and in the
self-consistent one.  `cpu_baseline` times the C oracle (test infrastructure, used here
only as the thing compared against) on the host cores, and the literal numpy restatement
of the reference's per-point Closest.
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


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(src, tgt, sample, numpy_queries):
    """C oracle (OpenMP) on a bounded sample of the same workload + the literal numpy restatement of the
    reference's Closest (icp_3d.py:39-53: one Python iteration per source point) on a few queries and on C1."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    import icp_oracle as orc
    nproc = os.cpu_count() or 1
    threads = min(nproc, 16)
    q = src[:sample]
    c_oracle.match_idx(q[:256], tgt, threads=threads)        # warm up / page in
    t0 = time.perf_counter()
    c_oracle.match_idx(q, tgt, threads=threads)
    dt = time.perf_counter() - t0
    out = {"value": sample / dt, "unit": "queries/s", "cores": threads, "kind": "port",
           "sample": "%d of the %d source points matched against the full %d-point target by the C oracle "
                     "(gcc -O3 -ffp-contract=off -fopenmp, %d threads); %.1f s" % (sample, src.shape[0], tgt.shape[0], threads, dt),
           "pair_evals_per_s": sample * tgt.shape[0] / dt, "nproc": nproc, "cpu_model": cpu_model()}
    if numpy_queries > 0:
        # (i) of SURVEY.md 8d: the numpy restatement, single-threaded by nature
        t0 = time.perf_counter()
        for i in range(numpy_queries):
            orc.closest(tgt, src[i])
        dq = (time.perf_counter() - t0) / numpy_queries
        s1, t1 = orc.make_rigid_cloud(1000, seed=0)
        t0 = time.perf_counter()
        orc.icp(s1, t1, -np.inf, 30)
        d1 = time.perf_counter() - t0
        out["numpy_reference"] = {
            "kind": "port", "cores": 1, "nproc": nproc, "cpu_model": cpu_model(),
            "closest_queries_per_s": 1.0 / dq,
            "closest_sample": "%d queries of the numpy Closest (icp_3d.py:39-42 restated: dset - pt, (disp*disp).sum(1), argmin) "
                              "against the full %d-point target, %.1f ms per query; a whole Match would take %.1f h (extrapolated)"
                              % (numpy_queries, tgt.shape[0], dq * 1e3, dq * src.shape[0] / 3600.0),
            "c1_icp_iters_per_s": 30.0 / d1,
            "c1_sample": "config C1 in full: ICP(1000 x 1000 points, 30 forced iterations) with the numpy restatement, %.2f s" % d1}
    if numpy_queries > 0:
        # (iii) of SURVEY.md 8d: the reference's NICP formulation (dense D_MinFunc + scipy leastsq = MINPACK lmder) is
        # only runnable on tiny meshes (the dense J of C3 would be 260 GB): 60 vertices, 3 outer iterations
        v, edges, t60 = orc.make_grid_mesh(10, 6, seed=3)
        nbr, wt = orc.knn_weights(v)
        P0 = orc.initialise(v)[0]
        hist = []
        t0 = time.perf_counter()
        orc.nicp(P0, nbr, wt, v, edges, t60, 0.01, 3, history=hist)
        d60 = time.perf_counter() - t0
        out["nicp_reference_formulation"] = {
            "kind": "port", "cores": 1, "vertices": int(v.shape[0]), "edges": int(len(edges)), "outer_iters": len(hist),
            "seconds": d60, "outer_iters_per_s": len(hist) / d60,
            "sample": "nicp_3d.py:344-363 restated (dense Jacobian of 9N+6E x 12N+6 entries + scipy.optimize.leastsq) on a "
                      "%d-vertex mesh, %d outer iterations, %.1f s; C3 in this formulation needs a 260 GB Jacobian" %
                      (v.shape[0], len(hist), d60)}
    return out


def nicp_c3(ctx, compat):
    """BASELINE config C3: nicp_3d.py's outer loop (Deform -> Match -> leastsq -> stiffness halving) on the 10k-vertex
    synthetic mesh, tol 0.01, up to 50 outer iterations; MINPACK's own stopping rule (no LM cap)."""
    import torch
    from icp_amd import nicp
    sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
    from nicp_bench import make_grid_mesh
    v, edges, tgt = make_grid_mesh(100, 100, seed=7)
    w = nicp.knn_weights(v, ctx)
    P0 = nicp.initial_params(v.shape[0], 3)
    hist = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dd, P, c = nicp.nicp(P0, w, v, edges, tgt, 0.01, 50, (0.1, 1000.0, 100.0), compat=compat, history=hist, ctx=ctx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cg = sum(h[3]["cg_iters"] for h in hist)
    return {"mode": "reference_compat" if compat else "self-consistent", "vertices": int(v.shape[0]), "edges": int(len(edges)),
            "seconds": dt, "outer_iters": len(hist), "outer_iters_per_s": len(hist) / dt,
            "lm_iters": sum(h[3]["lm_iters"] for h in hist), "cg_iters": cg, "cg_iters_per_s": cg / dt,
            "us_per_cg_iter_wall": dt / max(cg, 1) * 1e6, "err_first": hist[0][0], "err_last": hist[-1][0],
            "final_c_rigid": c[1],
            "stopped_by": {name: sum(1 for h in hist if h[3]["info"] == code) for code, name in
                           ((1, "ftol"), (2, "xtol"), (3, "ftol+xtol"), (4, "gtol"), (5, "limit"))}}


def _max_over_ranks(dt, share_gpu):
    import torch
    import torch.distributed as dist
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.cpu()[0])


def leg_c4(args, ctx, comm, lib_rccl, world, rank, barrier):
    """BASELINE config C4: rigid ICP 2^24 x 2^24, source rows sharded, target replicated, 10 iterations through
    icp_rigid_run_dist; every rank checks a sample of its shard's correspondences against the full scan."""
    import torch
    import torch.distributed as dist
    from icp_amd import rigid
    from icp_amd.dist import RigidComm, rigid_icp_dist, shard_bounds
    n4, K4 = args.c4_n, 10
    src4, tgt4 = make_cloud(n4, seed=2)
    lo, hi = shard_bounds(n4, world, rank)
    s_d, t_d = ctx.to_dev(src4[lo:hi]), ctx.to_dev(tgt4)
    del src4, tgt4
    cm = RigidComm("cuda:%d" % ctx.device, rccl=lib_rccl)
    run4 = lambda k, want: rigid_icp_dist(s_d, t_d, 0.0, k, use_tol=False, want_idx=want, comm=cm, ctx=ctx)
    run4(2, False)                                         # warm (scratch buffers, communicator)
    barrier()
    t0 = time.perf_counter()
    res = run4(K4, True)
    barrier()
    dt = _max_over_ranks(time.perf_counter() - t0, args.share_gpu)
    # correspondences of the last iteration on a sample of this rank's rows vs the brute-force scan of the same positions
    prev = run4(K4 - 1, False)["T"]
    n_loc = s_d.shape[0]
    take = torch.from_numpy(np.random.default_rng(100 + rank).choice(n_loc, size=min(n_loc, 1 << 15), replace=False)).to(s_d.device)
    ss = rigid.apply_transform(s_d[take].contiguous(), prev[:3, :3].contiguous(), prev[3, :3].contiguous(), ctx=ctx)
    ok = torch.equal(rigid.closest_indices(ss, t_d, force_brute=True, ctx=ctx), res["idx"][take])
    flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cpu" if args.share_gpu else "cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    # the same call with ICP_COHERENT (rows whose match is provably unchanged skip the search): beside the
    # plain figure, never instead of it; must return the plain run's bits on every rank
    coh = None
    try:
        run4c = lambda k: rigid_icp_dist(s_d, t_d, 0.0, k, use_tol=False, want_idx=True, comm=cm, ctx=ctx, coherent=True)
        run4c(2)
        barrier()
        t0 = time.perf_counter()
        resc = run4c(K4)
        barrier()
        dtc = _max_over_ranks(time.perf_counter() - t0, args.share_gpu)
        same = torch.equal(resc["T"], res["T"]) and torch.equal(resc["idx"], res["idx"]) and torch.equal(resc["err_hist"], res["err_hist"])
        f2 = torch.tensor([1.0 if same else 0.0], dtype=torch.float64, device="cpu" if args.share_gpu else "cuda")
        dist.all_reduce(f2, op=dist.ReduceOp.MIN)
        coh = {"seconds": dtc, "ms_per_step": dtc / K4 * 1e3, "queries_per_s": n4 * K4 / dtc,
               "equals_plain_run_bit_for_bit": bool(float(f2.cpu()[0]) == 1.0)}
    except Exception as e:  # pragma: no cover
        coh = {"error": repr(e)}
    return {"coherent_loop": coh,
            "workload": "C4: rigid ICP %d x %d fp64 points (seed 2), source rows sharded over %d ranks, target replicated, "
                        "%d iterations in one icp_rigid_run_dist call incl. grid build + Morton sort" % (n4, n4, world, K4),
            "seconds": dt, "ms_per_step": dt / K4 * 1e3, "queries_per_s": n4 * K4 / dt, "icp_iters_per_s": K4 / dt,
            "rows_per_rank": int(n_loc), "collectives_per_iteration": 1, "doubles_per_collective": 17,
            "host_callbacks": len(cm.calls), "last_err": float(res["err_hist"].cpu()[K4 - 1]),
            "check_idx_equal_bruteforce": bool(float(flag.cpu()[0]) == 1.0), "n_checked_per_rank": int(take.numel()),
            "check": "sample of each rank's last-iteration correspondences vs an ICP_FORCE_BRUTE Match of the same moved rows "
                     "against all %d targets" % n4}


def leg_c5(args, ctx, comm, lib_rccl, world, rank, barrier):
    """BASELINE config C5: NICP on the 200 704-vertex mesh, vertices sharded along the Morton curve with halos, through
    nicp_solve_dist.  A bounded piece of the first outer iteration is timed (whole self-consistent schedules take
    minutes): Deform -> Match -> 2 LM steps of at most 300 CG iterations."""
    import torch
    from icp_amd import nicp, nicp_dist
    sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
    from nicp_bench import make_grid_mesh
    v, edges, tgt = make_grid_mesh(args.c5_nx, args.c5_nx, seed=7)
    w = nicp.knn_weights(v, ctx)
    nbr, wt = w.nbr.cpu().numpy(), w.wt.cpu().numpy()
    sh = nicp_dist.ShardedNICP(v, nbr, wt, edges, tgt, world=world, rank=rank, ctx=ctx, rccl=lib_rccl)
    c = (0.1, 1000.0, 100.0)
    opts = dict(max_lm_iters=2, max_cg_iters=300, cg_tol=1e-6)
    nn = sh.match()
    keep = sh.P.clone()
    sh.solve(nn, c, dict(max_lm_iters=1, max_cg_iters=20, cg_tol=1e-6))      # warm (workspace, communicator)
    sh.P.copy_(keep)
    calls0 = dict(sh.comm.calls)
    barrier()
    t0 = time.perf_counter()
    nn = sh.match()
    torch.cuda.synchronize()
    t_match = time.perf_counter() - t0
    st = sh.solve(nn, c, opts)
    barrier()
    dt = _max_over_ranks(time.perf_counter() - t0, args.share_gpu)
    cg = max(st["cg_iters"], 1)
    cb = sum(sh.comm.calls.values()) - sum(calls0.values())
    return {"workload": "C5: NICP %dx%d grid mesh (%d vertices, %d edges), vertices sharded over %d ranks along a Morton curve, "
                        "halo = KNN lists + edges; first outer iteration bounded to 2 LM steps of <= 300 CG iterations, "
                        "self-consistent mode (the reference's quirk Q5 does not shard)" % (args.c5_nx, args.c5_nx, len(v), len(edges), world),
            "seconds": dt, "match_seconds": t_match, "lm_iters": st["lm_iters"], "cg_iters": st["cg_iters"],
            "cg_iters_per_s": st["cg_iters"] / dt, "us_per_cg_iter": (dt - t_match) / cg * 1e6,
            "outer_iters_per_s_bounded": 1.0 / dt, "err0": st["err0"], "err": st["err"],
            "owned_vertices": int(sh.shard.n_owned), "halo_vertices": int(sh.shard.n_halo),
            "halo_fraction": sh.shard.n_halo / max(sh.shard.n_owned, 1),
            "collectives_per_cg_iteration": 3, "collectives": "2 all-reduces of <= 68 doubles + 1 halo exchange of the search direction",
            "host_callbacks_per_cg_iteration": cb / cg, "graph_launches": st["graph_launches"]}


def _time_call(barrier, fn):
    barrier()
    t0 = time.perf_counter()
    fn()
    barrier()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)   # C2: 30 iterations
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--points", dest="n", type=int, default=1 << 20,
                    help="source AND target points (default 2^20); under torch.distributed.run spell it --points (its own parser trips over --n)")
    ap.add_argument("--repeats", type=int, default=5, help="further timed calls of K steps after the first (extra.repeats)")
    ap.add_argument("--cpu-sample", type=int, default=262144, help="queries the CPU baseline matches (about 15-20 s on 16 threads)")
    ap.add_argument("--numpy-queries", type=int, default=64, help="queries of the numpy Closest restatement (0: skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-nicp", action="store_true", help="skip extra.nicp_c3 (about 20 s)")
    ap.add_argument("--no-check", action="store_true", help="skip the brute-force check of the timed run's correspondences")
    ap.add_argument("--brute", action="store_true", help="force the brute-force closest-point kernel")
    ap.add_argument("--no-fuse", action="store_true", help="ICP_NO_FUSE: moved source as an array, apply / error as kernels of their own")
    ap.add_argument("--rehearse-comm", action="store_true",
                    help="one process, one GPU: run the SHARDED entry point with a one-rank RCCL group and make the library "
                         "call back every collective (measures the host cost of the multi-GPU path on one GPU)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses GPU 0 and the all-reduce goes over gloo")
    ap.add_argument("--comm", choices=("rccl", "callback"), default="rccl",
                    help="N > 1 (and --rehearse-comm): who issues the collectives -- the library itself over its own RCCL "
                         "communicator on the context's stream (default; include/icp_hip.h icp_rccl_create), or the host "
                         "callback over torch.distributed.  --share-gpu always uses the callback (RCCL refuses two ranks on one GPU)")
    ap.add_argument("--no-multi-legs", action="store_true", help="N > 1: skip extra.c4 / extra.c5")
    ap.add_argument("--c4-n", type=int, default=1 << 24, help="N > 1: points per cloud of the extra.c4 leg (BASELINE config C4: 2^24)")
    ap.add_argument("--c5-nx", type=int, default=448, help="N > 1: grid side of the extra.c5 mesh (448 x 448 = 200 704 vertices)")
    ap.add_argument("--legs-deadline", type=float, default=300.0,
                    help="N > 1: seconds the extra.c4 / extra.c5 legs may take before the line is printed without them")
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
        faulthandler.dump_traceback_later(120, exit=True)   # a rehearsal that hangs says where
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
    from icp_amd.dist import RigidComm, rigid_icp_dist, shard_bounds

    ctx = get_context(local_rank)
    n = args.n
    src, tgt = make_cloud(n, seed=1)
    lo, hi = shard_bounds(n, world, rank)
    K, W = args.steps, args.warmup
    src_d, tgt_d = ctx.to_dev(src[lo:hi]), ctx.to_dev(tgt)
    n_loc = src_d.shape[0]
    # (ICP_BENCH_FORCE_LIBRCCL=1: try the library communicator even with ranks sharing a GPU -- RCCL refuses that, which is how
    #  the fallback below is rehearsed on a one-GPU box)
    use_lib_rccl = args.comm == "rccl" and (not args.share_gpu or os.environ.get("ICP_BENCH_FORCE_LIBRCCL") == "1")
    lib_rccl = None
    comm = None
    rccl_note = None
    if distributed:
        if use_lib_rccl:
            # the library's own communicator (id broadcast over torch).  It has never run on more than one rank (no
            # multi-GPU node in any round): if creating it, or a first tiny sharded run over it, fails on ANY rank, every
            # rank falls back to the torch.distributed callbacks -- the headline must not depend on an unrehearsed path
            ok, why = 1.0, ""
            try:
                lib_rccl = _lib.LibraryRccl(ctx, rank=rank, world=world)
                tiny = ctx.to_dev(src[lo:hi][:4096])
                rigid_icp_dist(tiny, ctx.to_dev(tgt[:8192]), 0.0, 2, use_tol=False, comm=RigidComm("cuda:%d" % local_rank, rccl=lib_rccl), ctx=ctx)
                torch.cuda.synchronize()
            except Exception as e:  # pragma: no cover (needs a multi-GPU node)
                ok, why = 0.0, repr(e)
            flag = torch.tensor([ok], dtype=torch.float64, device="cpu" if args.share_gpu else "cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag.cpu()[0]) != 1.0:
                rccl_note = "library RCCL unavailable on this node (%s): torch.distributed callbacks instead" % (why or "another rank failed")
                try:
                    if lib_rccl is not None:
                        lib_rccl.close()
                except Exception:
                    pass
                lib_rccl = None
        comm = RigidComm("cuda:%d" % local_rank, rccl=lib_rccl)
    if args.rehearse_comm and not distributed:
        if use_lib_rccl:
            lib_rccl = _lib.LibraryRccl(ctx, rank=0, world=1)              # one-rank communicator, every collective really issued
            comm = RigidComm("cuda:%d" % local_rank, world=1, rccl=lib_rccl, force=True)
        else:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
            dist.all_reduce(torch.zeros(17, dtype=torch.float64, device="cuda"))
            comm = RigidComm("cuda:%d" % local_rank)
            comm.world, comm.native = 2, True        # Python side: really all-reduce (over the one-rank RCCL group)
            comm.struct.world = 2                    # library side: call back for every collective

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def run(steps, want_idx=True):
        """ONE library call = `steps` whole ICP iterations, device-resident, set-up included."""
        if distributed or comm is not None:
            return rigid_icp_dist(src_d, tgt_d, 0.0, steps, use_tol=False, force_brute=args.brute, want_idx=want_idx, comm=comm, ctx=ctx,
                                  no_fuse=args.no_fuse)
        return rigid.rigid_icp(src_d, tgt_d, 0.0, steps, use_tol=False, force_brute=args.brute, want_idx=want_idx, ctx=ctx,
                               no_fuse=args.no_fuse)

    def timed(steps):
        barrier()
        t0 = time.perf_counter()
        res = run(steps)
        barrier()
        dt = time.perf_counter() - t0
        if distributed:      # MAX over ranks
            tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.share_gpu else "cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.cpu()[0])
        return dt, res

    if W > 0:
        run(W)
    dt, res = timed(K)                                     # the contract's timed region: exactly K steps
    assert int(res["iters"].cpu()[0]) == K
    err_last = float(res["err_hist"].cpu()[K - 1])
    idx_timed = res["idx"].clone()
    reps = [timed(K)[0] for _ in range(max(args.repeats, 0))] if not args.brute else []
    # one-time part of a call (target grid + Morton sort + means): a 1-step call minus one marginal step
    t1 = min(timed(1)[0] for _ in range(3)) if not args.brute else None
    all_runs = [dt] + reps
    best = min(all_runs)
    marginal = (best - t1) / (K - 1) if (t1 is not None and K > 1) else None

    # ---- tie the timed run to the full scan: Match of the final source positions by the brute-force kernel
    check = None
    if not args.no_check and not args.brute and float(n_loc) * n <= 2.0 ** 42:   # the full scan of a C4-size shard would take minutes
        if K > 1:
            prev = run(K - 1, want_idx=False)["T"]         # deterministic: the same bits as inside the timed call
            ss = rigid.apply_transform(src_d, prev[:3, :3].contiguous(), prev[3, :3].contiguous(), ctx=ctx)
        else:
            ss = src_d
        t0 = time.perf_counter()
        idx_brute = rigid.closest_indices(ss, tgt_d, force_brute=True, ctx=ctx)
        torch.cuda.synchronize()
        ok = torch.tensor([1.0 if torch.equal(idx_brute, idx_timed) else 0.0], dtype=torch.float64,
                          device="cpu" if (args.share_gpu or not distributed) else "cuda")
        if distributed:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        check = {"idx_equal_bruteforce": bool(float(ok.cpu()[0]) == 1.0), "n_checked": n,
                 "what": "correspondences returned by the timed call (last of its %d iterations) vs an ICP_FORCE_BRUTE Match of "
                         "src . R_%d + t_%d over all %d targets" % (K, K - 1, K - 1, n),
                 "brute_match_s": time.perf_counter() - t0}

    # ---- dominant kernel, timed live with HIP events on the launch stream ----
    # Default path: the exact grid search (grid_query_kernel; the target grid is built once and the
    # queries are in Morton order, as inside icp_rigid_run).  --brute: the full-scan kernel.
    idx = ctx.empty((n_loc,), torch.int32)

    def time_closest(fl, reps_):
        ctx.call("icp_closest", ptr(src_d), n_loc, ptr(tgt_d), n, 3, fl, ptr(idx), ptr(None))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps_):
            ctx.call("icp_closest", ptr(src_d), n_loc, ptr(tgt_d), n, 3, fl, ptr(idx), ptr(None))
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) * 1e-3 / reps_

    pairs = float(n_loc) * n
    compulsory = 24.0 * n_loc + 24.0 * n + 4.0 * n_loc
    t_brute = time_closest(_lib.ICP_FORCE_BRUTE, 2) if pairs <= 2.0 ** 42 else None    # ~0.3 s per launch at 2^20 x 2^20
    if args.brute:
        t_kernel = t_brute
    else:
        # the search as iterations 2..K of the loop run it (previous answer as the hint), and the first iteration's (no hint);
        # both through grid_pool_kernel
        t_kernel, t_lane, t_defer = rigid.time_grid_query(src_d, tgt_d, reps=max(3, min(K, 10)), ctx=ctx, parts=True, hinted=True)
        t_first, t_first_lane, t_first_defer = rigid.time_grid_query(src_d, tgt_d, reps=max(3, min(K, 10)), ctx=ctx, parts=True)
    brute_roof = None
    if t_brute is not None:
        ab = 8.0 * pairs / t_brute / 1e12
        brute_roof = {"bound": "valu", "achieved": ab, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": ab / FP64_VECTOR_PEAK_TFLOPS, "traffic": None, "kernel": "closest_tiles<3,4,false>",
                      "kernel_ms": t_brute * 1e3,
                      "note": "8 fp64 flop per source-target pair x %.3g pairs per launch; peak = MI355X fp64 VECTOR peak "
                              "(FMA-rated; the reference arithmetic forbids FMA so <= 39.3, and 9 instructions per 8 flop => "
                              "ceiling ~35 TFLOP/s); VALU-bound, compulsory HBM bytes %.1f MB" % (pairs, compulsory / 1e6)}
    if args.brute:
        roof = brute_roof
    else:
        gbytes = 24.0 * n_loc + 4.0 * n_loc + 32.0 * n + 4.0 * (1 << 21)   # queries + idx out + 32-B target records + cell-start array
        achieved = gbytes / t_kernel / 1e9
        roof = {"bound": "latency: vector-instruction issue + L1 address path (not hbm, not mfma; counters below)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "frac_compulsory": compulsory / t_kernel / 1e9 / HBM_PEAK_GBS,
                "compulsory_bytes": compulsory,
                "kernel": "grid search as iterations 2..K run it (previous match as the hint) = grid_pool_kernel<3> + "
                          "grid_defer_kernel<3,false,false>", "kernel_ms": t_kernel * 1e3,
                "kernel_ms_parts": {"grid_pool_kernel": t_lane * 1e3, "grid_defer_kernel": t_defer * 1e3},
                "first_iteration": {"kernel": "grid_pool_kernel<3> + grid_defer_kernel (no hint)",
                                    "kernel_ms": t_first * 1e3,
                                    "kernel_ms_parts": {"grid_pool_kernel": t_first_lane * 1e3, "grid_defer_kernel": t_first_defer * 1e3},
                                    "achieved": gbytes / t_first / 1e9, "frac": gbytes / t_first / 1e9 / HBM_PEAK_GBS},
                "note": "exact grid search (two launches: the wave-cooperative pass, then one half-wave per deferred query): "
                        "algorithmic bytes per search = 24 B/query + 4 B/index out + 32 B/target record (Morton-sorted, read at "
                        "least once) + the 2^21-entry cell-start array = %.1f MB (the figure of rounds 1-2, kept: the pass now "
                        "reads 16-byte fp32 records and the 32-byte one only for the candidates the fp32 test cannot reject); the "
                        "search is bound by vector-instruction issue and dependent-access latency on gathers from the "
                        "L2/MALL-resident sorted target, not by HBM, so the HBM fraction is low by construction; equivalent "
                        "brute-force rate = %.3g pair-evaluations/s" % (gbytes / 1e6, pairs / t_kernel)}
        try:   # HBM-side bytes per launch and the busy fractions from the committed PMC passes; valid for the default size only
            tr = None
            for name in ("r3_traffic.json", "r2_traffic.json"):
                path = os.path.join(ROOT, "profiles", name)
                if os.path.exists(path):
                    tr = json.load(open(path))
                    break
            if tr is not None and n == (1 << 20) and world == 1:
                roof["traffic"] = tr["traffic_bytes"]
                roof["traffic_source"] = tr["source"]
                for k in ("valu_busy", "ta_busy", "wait_frac", "l2_hit"):
                    if k in tr:
                        roof[k] = tr[k]
        except Exception:
            pass
        if brute_roof is not None:
            roof["bruteforce_kernel"] = brute_roof

    extra = {"timed_calls_ms_per_step": [t / K * 1e3 for t in all_runs],
             "repeats": {"calls": len(all_runs), "median_ms_per_step": float(np.median(all_runs)) / K * 1e3,
                         "min_ms_per_step": min(all_runs) / K * 1e3, "max_ms_per_step": max(all_runs) / K * 1e3},
             "setup_ms": None if marginal is None else (t1 - marginal) * 1e3,
             "ms_per_step_without_setup": None if marginal is None else marginal * 1e3,
             "setup_note": "every timed call includes the one-time target-grid build, the Morton sort of the source shard and "
                           "the two means (setup_ms, from 1-step calls); ms_per_step_without_setup is the marginal iteration"}
    if world == 1 and comm is None and not args.brute and not args.no_fuse:
        # ---- the coherent loop (ICP_COHERENT, opt-in): NOT the headline.  Rows whose match is provably unchanged (second-nearest
        # bound + triangle inequality) skip the search; the run must return the plain loop's bits.
        try:
            def run_c(steps):
                return rigid.rigid_icp(src_d, tgt_d, 0.0, steps, use_tol=False, want_idx=True, ctx=ctx, coherent=True)
            run_c(max(W, 1))
            tc = []
            for _ in range(3):
                barrier()
                t0 = time.perf_counter()
                rc_ = run_c(K)
                barrier()
                tc.append(time.perf_counter() - t0)
            t1c = min(_time_call(barrier, lambda: run_c(1)) for _ in range(3))
            same = (torch.equal(rc_["T"], res["T"]) and torch.equal(rc_["idx"], idx_timed) and torch.equal(rc_["err_hist"], res["err_hist"]))
            extra["coherent_loop"] = {
                "flag": "ICP_COHERENT", "ms_per_step": min(tc) / K * 1e3, "calls_ms_per_step": [t / K * 1e3 for t in tc],
                "queries_per_s": n * K / min(tc),
                "ms_per_step_without_setup": (min(tc) - t1c) / (K - 1) * 1e3 if K > 1 else None,
                "equals_plain_loop_bit_for_bit": bool(same),
                "speedup_vs_plain_call": best / min(tc),
                "note": "same K-step call with ICP_COHERENT: every search also returns the second-best squared distance; later "
                        "iterations prove most matches unchanged (triangle inequality on the row's motion since its last search) "
                        "and search only the rest; T, correspondences and error history compared with the timed plain call's. "
                        "Not `value`: the headline stays the loop that searches every row in every iteration"}
        except Exception as e:  # pragma: no cover
            extra["coherent_loop"] = {"error": repr(e)}
    if world == 1 and not args.no_nicp and not args.brute:
        try:
            extra["nicp_c3"] = {"compat": nicp_c3(ctx, True), "corrected": nicp_c3(ctx, False),
                                "note": "config C3, whole outer loop per mode; NICP CG kernels: see DESIGN.md 3.3 and profiles/"}
            # the CG iteration timed live with HIP events on the launch stream (nicp_profile_cg): C3 size with the plain and the
            # enriched coarse space, and C5 size (200k vertices) on this one GPU; roofline = SURVEY 8(d)'s 0.86 kB per vertex per
            # iteration over the iteration as the solver runs it, and each kernel's own minimal bytes over its launch
            sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
            import cg_profile
            extra["nicp_cg"] = {"c3_10k_plain_coarse": cg_profile.profile(100, 100, 100, 1, ctx),
                                "c3_10k_enriched_coarse": cg_profile.profile(100, 100, 100, 10, ctx),
                                "c5_size_200k_one_gpu": cg_profile.profile(448, 448, 100, 1, ctx),
                                "note": "us_iteration: hipGraph replay, as nicp_solve runs it; us_kernel: plain launches with an event "
                                        "after each (kernel + launch gap, an upper bound on the kernel)"}
        except Exception as e:      # the rigid line must survive an NICP failure
            extra["nicp_c3"] = {"error": repr(e)}

    def build_line():
        cfg = {"workload": "%s: rigid ICP, %d source x %d target fp64 points (N(0,1)^3, seed 1, Rz(0.3)+shift, permuted), "
                           "source rows sharded over %d GPU(s), target replicated" % (
                               "C2" if n == (1 << 20) else ("C4-size" if n == (1 << 24) else "custom"), n, n, world),
               "n_source": n, "n_target": n, "dim": 3,
               "search": "brute-force scan" if args.brute else "exact multi-resolution grid (bit-identical to the scan)",
               "timed_region": "one icp_rigid_run%s call of K iterations incl. grid build + Morton sort" % (
                   "_dist" if (distributed or comm is not None) else "") + (
                   " (REHEARSAL: one rank, every collective really issued on a one-rank RCCL communicator)" if args.rehearse_comm else ""),
               "last_err": err_last, "check": check,
               # flat copies: the driver's parser drops nested objects
               "check_idx_equal_bruteforce": None if check is None else check["idx_equal_bruteforce"],
               "n_checked": None if check is None else check["n_checked"],
               "collectives": ("library RCCL (icp_rccl_create): zero host callbacks" if lib_rccl is not None else
                               ((rccl_note or "host callback over torch.distributed") if comm is not None else "none (one rank)"))}
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
            "config": cfg,
            "roofline": roof,
            "extra": extra,
        }
        return out

    # ---- N > 1: BASELINE configs C4 and C5 through the sharded entry points (extra.c4 / extra.c5).  The line above is
    # complete without them: a watchdog prints it and ends the process if a leg overruns its deadline (a hung collective
    # on a node nobody could rehearse on must not cost the headline measurement).
    if distributed and not args.no_multi_legs and not args.brute:
        import threading
        state = {"printed": False}

        def bail():
            if rank == 0 and not state["printed"]:
                state["printed"] = True
                extra["multi_gpu_legs"] = "deadline of %.0f s exceeded: line printed without the unfinished legs" % args.legs_deadline
                line = build_line()
                line["cpu_baseline"] = None
                print(json.dumps(line), flush=True)
            os._exit(0)
        guard = threading.Timer(args.legs_deadline + (0.0 if rank == 0 else 10.0), bail)
        guard.daemon = True
        guard.start()
        for name, leg in (("c4", leg_c4), ("c5", leg_c5)):
            try:
                extra[name] = leg(args, ctx, comm, lib_rccl, world, rank, barrier)
            except Exception as e:
                extra[name] = {"error": repr(e)}
        guard.cancel()

    if rank == 0:
        out = build_line()
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(src, tgt, min(args.cpu_sample, n), args.numpy_queries)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
    if lib_rccl is not None:
        lib_rccl.close()
    if distributed or (args.rehearse_comm and lib_rccl is None):
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
