"""Device-resident rigid-ICP operators over the C-ABI (include/icp_hip.h).

These take/return torch tensors that already live in HBM (numpy arrays are uploaded);
the reference-named shims in icp_3d.py / icp_row.py / icp.py are thin wrappers.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import get_context, ptr


def _dims(ctx, src, tgt=None):
    src = ctx.to_dev(src)
    if src.dim() != 2 or src.shape[1] not in (2, 3):
        raise ValueError("points must be [N, 2] or [N, 3], got %s" % (tuple(src.shape),))
    if tgt is None:
        return src, None, src.shape[1]
    tgt = ctx.to_dev(tgt)
    if tgt.dim() != 2 or tgt.shape[1] != src.shape[1]:
        raise ValueError("shape mismatch: %s vs %s" % (tuple(src.shape), tuple(tgt.shape)))
    return src, tgt, src.shape[1]


def closest_indices(source, target, compare_sqrt=False, want_d2=False, force_brute=False, force_grid=False,
                    ctx=None, hint=None):
    """idx[i] = argmin_j |target[j]-source[i]|^2 (bit-exact reference arithmetic,
    lowest index on ties).  Returns int32 tensor (and the fp64 d2 tensor).
    hint: optional int32 indices (e.g. the previous iteration's result) used as the grid
    search's starting bound (ICP_HINT_IDX); the result does not depend on them."""
    ctx = ctx or get_context()
    torch = ctx.torch
    src, tgt, dim = _dims(ctx, source, target)
    n, m = src.shape[0], tgt.shape[0]
    if m == 0:
        raise ValueError("attempt to get argmin of an empty sequence")  # numpy's message
    idx = ctx.empty((n,), torch.int32) if hint is None else ctx.to_dev(hint, torch.int32).clone()
    d2 = ctx.empty((n,), torch.float64) if want_d2 else None
    if n == 0:      # Match over an empty source: nothing to launch (empty tensors have no storage)
        return (idx, d2) if want_d2 else idx
    flags = ((_lib.ICP_COMPARE_SQRT if compare_sqrt else 0) | (_lib.ICP_FORCE_BRUTE if force_brute else 0) |
             (_lib.ICP_FORCE_GRID if force_grid else 0) | (_lib.ICP_HINT_IDX if hint is not None else 0))
    ctx.call("icp_closest", ptr(src), n, ptr(tgt), m, dim, flags, ptr(idx), ptr(d2))
    return (idx, d2) if want_d2 else idx


def gather_points(target, idx, ctx=None):
    ctx = ctx or get_context()
    tgt = ctx.to_dev(target)
    idx = ctx.to_dev(idx, ctx.torch.int32)
    out = ctx.empty((idx.shape[0], tgt.shape[1]))
    ctx.call("icp_gather_points", ptr(tgt), ptr(idx), idx.shape[0], tgt.shape[1], ptr(out))
    return out


def distance_sum(s1, s2, ctx=None):
    ctx = ctx or get_context()
    a, b, dim = _dims(ctx, s1, s2)
    if a.shape != b.shape:
        raise ValueError("operands could not be broadcast together")
    out = ctx.empty((1,))
    ctx.call("icp_distance_sum", ptr(a), ptr(b), a.shape[0], dim, ptr(out))
    return out


def find_transform(source, target, ctx=None):
    """(R[dim,dim], t[dim]) device tensors; icp_3d.py:68-90."""
    ctx = ctx or get_context()
    s, t, dim = _dims(ctx, source, target)
    if s.shape != t.shape:
        raise ValueError("source and target must have the same shape")
    R = ctx.empty((dim, dim))
    tv = ctx.empty((dim,))
    ctx.call("icp_find_transform", ptr(s), ptr(t), s.shape[0], dim, ptr(R), ptr(tv))
    return R, tv


def apply_transform(source, R, t, ctx=None):
    ctx = ctx or get_context()
    s, _, dim = _dims(ctx, source)
    R = ctx.to_dev(R)
    t = ctx.to_dev(t)
    out = ctx.empty(tuple(s.shape))
    ctx.call("icp_apply_transform", ptr(s), s.shape[0], dim, ptr(R), ptr(t), ptr(out))
    return out


# from this many source rows on the hosts that mirror the reference's ICP() turn ICP_COHERENT on by themselves: the same
# results bit for bit, no gain below ~1e5 rows; rigid_icp itself keeps the flag explicit
COHERENT_MIN_ROWS = 131072


def rigid_icp(source, target, tol, run, use_tol=True, compare_sqrt=False, force_brute=False, want_idx=False,
              force_grid=False, ctx=None, no_fuse=False, coherent=False):
    """Fused ICP loop on the device.  Returns dict(T, err_hist, iters[, idx]) of device tensors.
    coherent: ICP_COHERENT (icp_hip.h) -- rows whose match is provably unchanged skip the search; same results."""
    ctx = ctx or get_context()
    torch = ctx.torch
    src, tgt, dim = _dims(ctx, source, target)
    run = int(run)
    if run <= 0:
        # icp_3d.py:118-120: R is never bound when the loop body does not execute
        raise NameError("name 'R' is not defined")
    T = ctx.empty((dim + 1, dim + 1))
    hist = ctx.empty((run,))
    iters = ctx.empty((1,), torch.int32)
    idx = ctx.empty((src.shape[0],), torch.int32) if want_idx else None
    flags = ((_lib.ICP_COMPARE_SQRT if compare_sqrt else 0) | (_lib.ICP_FORCE_BRUTE if force_brute else 0) |
             (_lib.ICP_FORCE_GRID if force_grid else 0) | (_lib.ICP_NO_FUSE if no_fuse else 0) |
             (_lib.ICP_COHERENT if coherent else 0))
    tolv = float(tol)
    ctx.call("icp_rigid_run", ptr(src), src.shape[0], ptr(tgt), tgt.shape[0], dim, ctypes.c_double(tolv), run,
             1 if use_tol else 0, flags, ptr(T), ptr(hist), ptr(iters), ptr(idx))
    out = dict(T=T, err_hist=hist, iters=iters)
    if want_idx:
        out["idx"] = idx
    return out


def time_grid_query(source, target, reps=5, ctx=None, parts=False, hinted=False):
    """Average duration (s) of one grid search as it runs inside icp_rigid_run: target grid
    prebuilt, queries in Morton order.  Measured with HIP events on the launch stream
    (bench.py's roofline leg).  parts=True: (whole search, per-lane kernel, grid_defer_kernel).
    hinted=True: the search with the previous answer as its hint (iterations 2.. of a loop);
    False: without (the first iteration).  Both run grid_pool_kernel."""
    ctx = ctx or get_context()
    torch = ctx.torch
    src, tgt, dim = _dims(ctx, source, target)
    ms = ctx.empty((6,))
    ctx.call("icp_bench_grid_query", ptr(src), src.shape[0], ptr(tgt), tgt.shape[0], dim, int(reps), ptr(ms))
    torch.cuda.synchronize()
    t = [float(x) * 1e-3 for x in ms.cpu()]
    t = t[3:] if hinted else t[:3]
    return tuple(t) if parts else t[0]
