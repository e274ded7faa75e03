"""Drop-in for the numeric functions of the reference's nicp_row_rt.py ("NICP machinery,
rigid only": 3 parameters [theta, tx, ty]).

The reference fits the three parameters with scipy.optimize.leastsq on Disp/D_Disp
(nicp_row_rt.py:124).  That least-squares problem -- min sum |nn - (src R(theta) + t)|^2
over proper 2-D rotations -- has the closed-form Kabsch solution, which is what
`FitRigid` computes on the GPU (one FindTransform); `Solve` runs the demo's outer loop
(Deform -> Match -> fit) without the plotting / raw_input.
"""
import numpy as np

from . import rigid
from ._lib import get_context
from .icp_3d import Closest, Match, _is_np, _out  # noqa: F401  (nicp_row_rt.py:55-69)


def _rot(rot):
    return np.array([[np.cos(rot), np.sin(rot)], [-np.sin(rot), np.cos(rot)]])


def Deform(dset, rot, tx, ty):
    """dset R(rot) + (tx, ty)  (nicp_row_rt.py:74-80)."""
    return _out(rigid.apply_transform(dset, _rot(float(rot)), np.array([float(tx), float(ty)])), _is_np(dset))


def Disp(p, srcdata, tgtdata):
    """Residual tgt - Deform(src, p), flattened (nicp_row_rt.py:86-88)."""
    ctx = get_context()
    d = ctx.to_dev(tgtdata) - rigid.apply_transform(srcdata, _rot(float(p[0])), np.array([float(p[1]), float(p[2])]))
    return _out(d.reshape(-1), _is_np(srcdata, tgtdata))


def D_Disp(p, srcdata, tgtdata):
    """Jacobian of Disp (nicp_row_rt.py:94-112, live lines): column 0 = -(src dR/dtheta),
    columns 1, 2 = -(1,0), -(0,1) per point."""
    ctx = get_context()
    th = float(p[0])
    dR = np.array([[-np.sin(th), np.cos(th)], [-np.cos(th), -np.sin(th)]])
    d_r = rigid.apply_transform(srcdata, -dR, np.zeros(2))
    n = d_r.shape[0]
    J = ctx.torch.zeros((2 * n, 3), dtype=ctx.torch.float64, device=d_r.device)
    J[:, 0] = d_r.reshape(-1)
    J[0::2, 1] = -1.0
    J[1::2, 2] = -1.0
    return _out(J, _is_np(srcdata))


def FitRigid(srcdata, tgtdata):
    """argmin_p |Disp(p)|^2 in closed form: p = (theta, tx, ty), plus the residual energy."""
    R, t = rigid.find_transform(srcdata, tgtdata)
    R, t = R.cpu().numpy(), t.cpu().numpy()
    p = np.array([np.arctan2(R[0, 1], R[0, 0]), t[0], t[1]])
    r = Disp(p, srcdata, tgtdata)
    r = r if isinstance(r, np.ndarray) else r.cpu().numpy()
    return p, float((r * r).sum())


def Solve(source, target, p0, iterations):
    """The nicp_row_rt.py:121-137 loop without plotting: returns the list of (p, energy)."""
    p = np.asarray(p0, dtype=np.float64)
    out = []
    for _ in range(iterations):
        dd = Deform(source, p[0], p[1], p[2])
        nn = Match(dd, target)
        p, e = FitRigid(source, nn)
        out.append((p, e))
    return out
