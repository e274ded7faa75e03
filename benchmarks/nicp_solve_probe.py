import sys, time, json
sys.path.insert(0, '.')
sys.path.insert(0, 'benchmarks')
import numpy as np, torch
from nicp_bench import make_grid_mesh
from icp_amd import nicp, rigid
for nx in (32, 100):
    v, edges, tgt = make_grid_mesh(nx, nx, seed=7)
    w = nicp.knn_weights(v)
    P0 = nicp.initial_params(len(v), 3)
    idx = rigid.closest_indices(v, tgt)
    nn = rigid.gather_points(tgt, idx)
    for lm in (1, 3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        P, st = nicp.solve(P0, w, v, edges, nn, (0.1, 1000.0, 100.0), compat=False, opts=dict(max_lm_iters=lm, cg_tol=1e-8, max_cg_iters=3000))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(nx * nx, 'lm', lm, st, 'sec', round(dt, 3), 'us/cg', round(dt / max(st['cg_iters'], 1) * 1e6, 1), flush=True)
