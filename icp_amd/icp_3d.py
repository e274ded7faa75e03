"""Drop-in for the hot-path functions of the reference's icp_3d.py (3-D rigid
point-to-point ICP): same names, positional arguments and return shapes, computed
by libicp_hip.so on an MI355X.

numpy in -> numpy out (like the reference).  torch CUDA tensors in -> CUDA tensors
out, with no host round trip (use that form to keep clouds resident in HBM).
The Maya glue of the reference (GetModel, GetVertices, ArrayToMMatrix, the `icp`
command class; icp_3d.py:16-34,126-207) is host-application code and not part of
this package.
"""
import numpy as np

from . import rigid
from ._lib import get_context


def _is_np(*xs):
    return not any(hasattr(x, "is_cuda") for x in xs)


def _out(t, as_np):
    return t.cpu().numpy() if as_np else t


def Closest(dset, pt):
    """Index of the row of dset nearest to pt (icp_3d.py:39-42)."""
    ctx = get_context()
    as_np = _is_np(dset, pt)
    q = ctx.to_dev(pt).reshape(1, -1)
    idx = rigid.closest_indices(q, dset, ctx=ctx)
    return int(idx.cpu()[0]) if as_np else idx[0]


def Match(source, target):
    """Closest target point for every source point (icp_3d.py:48-53)."""
    as_np = _is_np(source, target)
    ctx = get_context()
    tgt = ctx.to_dev(target)
    idx = rigid.closest_indices(source, tgt, ctx=ctx)
    return _out(rigid.gather_points(tgt, idx, ctx=ctx), as_np)


def MatchIndices(source, target):
    """The correspondence indices Match() discards (extension; int32)."""
    as_np = _is_np(source, target)
    return _out(rigid.closest_indices(source, target), as_np)


def Distance(s1, s2):
    """Sum of Euclidean distances between corresponding rows (icp_3d.py:59-62)."""
    as_np = _is_np(s1, s2)
    d = rigid.distance_sum(s1, s2)
    return float(d.cpu()[0]) if as_np else d[0]


def FindTransform(source, target):
    """Kabsch step without reflection fix; returns (R, t) (icp_3d.py:68-90)."""
    as_np = _is_np(source, target)
    R, t = rigid.find_transform(source, target)
    return _out(R, as_np), _out(t, as_np)


def ICP(source, target, tol, run):
    """Rigid ICP; returns the (dim+1)x(dim+1) row-vector matrix T (icp_3d.py:102-121).
    Prints the reference's per-iteration line `err = %f (%2d)` (icp_3d.py:115)."""
    as_np = _is_np(source, target)
    r = rigid.rigid_icp(source, target, tol, run)
    n = int(r["iters"].cpu()[0])
    for i, e in enumerate(r["err_hist"].cpu().numpy()[:n]):
        print("err = %f (%2d)" % (e, i))
    return _out(r["T"], as_np)
