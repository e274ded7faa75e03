"""Drop-in for the numeric functions of the reference's nicp_row.py (2-D non-rigid ICP
demo: 6 parameters per vertex [A(2x2), b(2)] + 3 globals [theta, tx, ty]).  Plotting and
the raw_input loop (nicp_row.py:235-264) are not reproduced; `NICP` below is the same loop
with the manual "R" key press replaced by nicp_3d.py's automatic stiffness schedule.

Q7 note: the reference's MinFunc / D_MinFunc read the module global `source` instead of
their `srcdata` argument in the smoothness term (nicp_row.py:154-155,182,209); the two are
the same array in the reference's only call site, and `srcdata` is used here.
"""
import numpy as np

from . import nicp as _n
from .icp_3d import Closest, Match, _is_np, _out  # noqa: F401  (nicp_row.py:75-89 are the icp_3d functions)
from .nicp_3d import KNN  # noqa: F401  (nicp_row.py:66-70)

num_nn = 4
c_fit = 0.1
c_rigid = 1000
c_smooth = 100
reference_compat = True


def Initialise(source, target):
    """nicp_row.py:94-107."""
    from ._lib import get_context
    ctx = get_context()
    v = ctx.to_dev(source)
    P = _n.initial_params(v.shape[0], 2)
    return (P if _is_np(source) else ctx.to_dev(P)), _n.knn_weights(v, ctx)


def Deform(P, w, source):
    """nicp_row.py:112-126."""
    return _out(_n.deform(P, w, source), _is_np(P, source))


def MinFunc(P, w, srcdata, edges, tgtdata):
    """nicp_row.py:132-160."""
    return _out(_n.residual(P, w, srcdata, edges, tgtdata, (c_fit, c_rigid, c_smooth), reference_compat),
                _is_np(P, srcdata))


def D_MinFunc(P, w, srcdata, edges, tgtdata):
    """nicp_row.py:167-230 (dense; small inputs only)."""
    return _out(_n.dense_jacobian(P, w, srcdata, edges, (c_fit, c_rigid, c_smooth), reference_compat),
                _is_np(P, srcdata))


def NICP(P, w, source, edges, target, tol, run):
    """The nicp_row.py:243-264 loop, non-interactive: returns (deformed, P)."""
    global c_fit, c_rigid, c_smooth
    as_np = _is_np(P, source)
    hist = []
    dd, Pn, c = _n.nicp(P, w, source, edges, target, tol, run, (c_fit, c_rigid, c_smooth), reference_compat,
                        history=hist)
    for i, (err, cr, cs, _) in enumerate(hist):
        print("{0:2d} error = {1:f} [{2:f},{3:f}]".format(i, err, cr, cs))
    c_fit, c_rigid, c_smooth = c
    return _out(dd, as_np), _out(Pn, as_np)
