"""nicp_knn through the grid on a vertex set with a few far-away strays (does the level climb scan coarse cells whole?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icp_amd import nicp
from icp_amd._lib import get_context
ctx = get_context(0)
rng = np.random.default_rng(5)
n = 200000
v = rng.standard_normal((n, 3)) * [1.0, 1.0, 0.05]
for strays in (0, 10, 1000):
    w = v.copy()
    if strays:
        w[:strays] = rng.standard_normal((strays, 3)) * 50.0
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = nicp.knn_weights(w, ctx)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("strays", strays, "knn_weights %.2f ms" % (dt * 1e3), flush=True)
