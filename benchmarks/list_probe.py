"""Per-search time of the grid search on differently distributed clouds, without hints (the first iteration of a loop) and with
the answer as the hint (the later ones).  ICP_GRID_LIST8=1: work lists of 8 ranges; ICP_GRID_POOL=0: per-lane kernel for hinted searches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icp_amd import rigid
from icp_amd._lib import get_context
ctx = get_context(0)
rng = np.random.default_rng(3)
def cases():
    n = 1 << 20
    g = rng.standard_normal((n, 3)); yield "gauss3 2^20", g + 0.01 * rng.standard_normal((n, 3)), g
    u = rng.uniform(0, 1, (n, 3)); yield "uniform3 2^20", u + 0.002 * rng.standard_normal((n, 3)), u
    m = 1 << 18
    sh = rng.uniform(0, 1, (m, 3)) * [1.0, 1.0, 0.0]; sh[:, 2] = 0.1 * np.sin(6 * sh[:, 0]) * np.cos(5 * sh[:, 1])
    yield "sheet 2^18", sh + 0.001 * rng.standard_normal((m, 3)), sh
    g2 = rng.standard_normal((n, 2)); yield "gauss2 2^20", g2 + 0.005 * rng.standard_normal((n, 2)), g2
    g4 = rng.standard_normal((1 << 22, 3)); yield "gauss3 2^22", g4[: 1 << 21] + 0.005, g4
for name, s, t in cases():
    sd, td = ctx.to_dev(s), ctx.to_dev(t)
    r = rigid.time_grid_query(sd, td, reps=5, ctx=ctx, parts=True)
    h = rigid.time_grid_query(sd, td, reps=5, ctx=ctx, parts=True, hinted=True)
    print("list8", os.environ.get("ICP_GRID_LIST8", "0"), "pool", os.environ.get("ICP_GRID_POOL", "1"), name,
          "no hint [whole, pass, deferred] ms", ["%.4f" % (x * 1e3) for x in r], "hinted", ["%.4f" % (x * 1e3) for x in h], flush=True)
