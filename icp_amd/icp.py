"""Drop-in for the numeric core of the reference's icp.py (oldest 2-D demo):
COLUMN-major data ([dim, N] arrays), R = U VT, t = t_mean - R s_mean, and an
incremental update (the caller overwrites `source` with the moved points).
"""
import numpy as np

from . import rigid
from ._lib import get_context


def closest(set, pt):
    """icp.py:18-21: `set` is [dim, M], `pt` is [dim, 1]; returns the column index."""
    ctx = get_context()
    tgt = ctx.to_dev(np.ascontiguousarray(np.asarray(set).T))
    q = ctx.to_dev(np.ascontiguousarray(np.asarray(pt).reshape(1, -1)))
    return int(rigid.closest_indices(q, tgt, ctx=ctx).cpu()[0])


def icp_step(source, target):
    """One pass of the icp.py:52-98 loop body: returns (nn, R, t, moved) with
    nn/moved as [dim, N], R = U VT (column-vector convention), t as [dim, 1]."""
    ctx = get_context()
    s = ctx.to_dev(np.ascontiguousarray(np.asarray(source).T))
    tg = ctx.to_dev(np.ascontiguousarray(np.asarray(target).T))
    idx = rigid.closest_indices(s, tg, ctx=ctx)
    nn = rigid.gather_points(tg, idx, ctx=ctx)
    R, t = rigid.find_transform(s, nn, ctx=ctx)     # row-vector R_row; R_col = R_row^T
    moved = rigid.apply_transform(s, R, t, ctx=ctx)
    return (nn.cpu().numpy().T.copy(), R.cpu().numpy().T.copy(), t.cpu().numpy()[:, None].copy(),
            moved.cpu().numpy().T.copy())
