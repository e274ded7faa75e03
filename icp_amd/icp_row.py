"""Drop-in for the numeric functions of the reference's icp_row.py (2-D row-vector
demo): FindTransform returns the combined T (icp_row.py:79-107) and ICP ignores
`tol` (icp_row.py:112-143, no early exit).  Plotting / raw_input are not reproduced.
"""
import numpy as np

from . import rigid
from .icp_3d import Closest, Match, Distance, _is_np, _out  # noqa: F401  (icp_row.py:50-73 are identical)


def FindTransform(source, target):
    as_np = _is_np(source, target)
    R, t = rigid.find_transform(source, target)
    dim = R.shape[0]
    T = R.new_zeros((dim + 1, dim + 1))
    T[:dim, :dim] = R
    T[dim, :dim] = t
    T[dim, dim] = 1.0
    return _out(T, as_np)


def ICP(source, target, tol, run):
    """Runs all `run` iterations (the reference stops only on user input) and returns T
    (the reference returns nothing; the matrix is what its last iteration computed)."""
    as_np = _is_np(source, target)
    r = rigid.rigid_icp(source, target, tol, run, use_tol=False, coherent=len(source) >= rigid.COHERENT_MIN_ROWS)
    for i, e in enumerate(r["err_hist"].cpu().numpy()):
        print("err = %f (%2d)" % (e, i))
    return _out(r["T"], as_np)
