"""Drop-in for the hot-path functions of the reference's nicp_3d.py (3-D non-rigid ICP,
embedded deformation): same names, positional arguments and return shapes, computed by
libicp_hip.so on an MI355X.  The Maya glue (nicp_3d.py:25-47,368-457) lives in
icp_amd/meshio.py and icp_amd/maya_plugin.py.

`w` is returned by Initialise as a SparseWeights object (the reference's dense N x N
matrix has 4 non-zeros per column and cannot exist at 10k+ vertices); every function
here also accepts a dense reference-style `w`.
"""
import numpy as np

from . import nicp as _n
from . import rigid
from ._lib import get_context
from .icp_3d import Closest, Distance, FindTransform, ICP, Match, _is_np, _out  # noqa: F401 (nicp_3d.py:62-145 copies)

# nicp_3d.py:17-20 -- module globals, mutated by NICP like the reference's `global` statement
num_nn = 4
c_fit = 0.1
c_rigid = 1000.0
c_smooth = 100.0

# False switches to the self-consistent residual/Jacobian pair (fixes Q5/Q6); the default
# reproduces the reference as coded.
reference_compat = True


def KNN(dset, pt, k):
    """Indices and distances of the k nearest rows (nicp_3d.py:53-57)."""
    ctx = get_context()
    as_np = _is_np(dset, pt)
    q = ctx.to_dev(pt).reshape(1, -1)
    idx, dist = _n.knn(dset, q, k, ctx)
    if as_np:
        return idx[0].cpu().numpy().astype(np.intp), dist[0].cpu().numpy()
    return idx[0], dist[0]


def Initialise(source, target):
    """(P, w): identity A / zero b per vertex + KNN influence weights (nicp_3d.py:150-166)."""
    ctx = get_context()
    as_np = _is_np(source)
    v = ctx.to_dev(source)
    P = _n.initial_params(v.shape[0], v.shape[1])
    w = _n.knn_weights(v, ctx)
    return (P if as_np else ctx.to_dev(P)), w


def PtoAB(P, v):
    """nicp_3d.py:171-175 (views into P)."""
    return P[(v * 12):(v * 12 + 9)].reshape(3, 3), P[(v * 12 + 9):(v * 12 + 12)]


def RotMatrix(x, y, z):
    """nicp_3d.py:180-191 -- three scalars, host arithmetic like the reference."""
    sx, cx, sy, cy, sz, cz = np.sin(x), np.cos(x), np.sin(y), np.cos(y), np.sin(z), np.cos(z)
    return np.array([[cy * cz, cy * sz, -sy],
                     [sx * sy * cz - cx * sz, sx * sy * sz + cx * cz, sx * cy],
                     [cx * sy * cz + sx * sz, cx * sy * sz - sx * cz, cx * cy]])


def D_RotMatrix(x, y, z):
    """nicp_3d.py:196-215."""
    sx, cx, sy, cy, sz, cz = np.sin(x), np.cos(x), np.sin(y), np.cos(y), np.sin(z), np.cos(z)
    d_Rx = np.array([[0, 0, 0],
                     [cx * sy * cz + sx * sz, cx * sy * sz - sx * cz, cx * cy],
                     [-sx * sy * cz + cx * sz, -sx * sy * sz - cx * cz, -sx * cy]])
    d_Ry = np.array([[-sy * cz, -sy * sz, -cy],
                     [sx * cy * cz, sx * cy * sz, -sx * sy],
                     [cx * cy * cz, cx * cy * sz, -cx * sy]])
    d_Rz = np.array([[-cy * sz, cy * cz, 0],
                     [-sx * sy * sz - cx * cz, sx * sy * cz - cx * sz, 0],
                     [-cx * sy * sz + sx * cz, cx * sy * cz + sx * sz, 0]])
    return (d_Rx, d_Ry, d_Rz)


def Deform(P, w, source):
    """nicp_3d.py:220-233."""
    return _out(_n.deform(P, w, source), _is_np(P, source))


def MinFunc(P, w, srcdata, edges, tgtdata):
    """Residual vector [E_fit | E_rigid | E_smooth] (nicp_3d.py:239-267)."""
    r = _n.residual(P, w, srcdata, edges, tgtdata, (c_fit, c_rigid, c_smooth), reference_compat)
    return _out(r, _is_np(P, srcdata))


def D_MinFunc(P, w, srcdata, edges, tgtdata):
    """Dense Jacobian (nicp_3d.py:274-338); small meshes only -- use icp_amd.nicp.jvp / jtvp
    for the matrix-free products the solver uses."""
    J = _n.dense_jacobian(P, w, srcdata, edges, (c_fit, c_rigid, c_smooth), reference_compat)
    return _out(J, _is_np(P, srcdata))


def NICP(P, w, source, edges, target, tol, run):
    """Non-rigid ICP outer loop (nicp_3d.py:344-363); returns (deformed, P).  Prints the
    reference's per-iteration line and halves c_rigid / c_smooth on stall like it does."""
    global c_fit, c_rigid, c_smooth
    as_np = _is_np(P, source)
    hist = []
    dd, Pn, c = _n.nicp(P, w, source, edges, target, tol, run, (c_fit, c_rigid, c_smooth), reference_compat,
                        history=hist)
    for i, (err, cr, cs, _) in enumerate(hist):
        print("{0:2d} error = {1:f} [{2:f},{3:f}]".format(i, err, cr, cs))
    c_fit, c_rigid, c_smooth = c
    return _out(dd, as_np), _out(Pn, as_np)


def main(srcPath=None, tgtPath=None, out=None, tol=0.01, run=50):
    """nicp_3d.py:390-409 with files in place of the Maya selection.  The reference stops after
    Initialise + Match (its NICP / ModifyVertices calls are commented out, :408-409); with `out`
    given this runs them and writes the deformed source."""
    from . import meshio
    print("NICP - Non-rigid Iterative Closest Point")
    sel = meshio.GetModel(*[p for p in (srcPath, tgtPath) if p is not None])
    if sel is None:
        print("You must select 2 models")
        return None
    srcPath, tgtPath = sel
    print("Source model is {0:s}".format(str(meshio.PathName(srcPath))))
    print("Target model is {0:s}".format(str(meshio.PathName(tgtPath))))
    if not (meshio.IsMesh(srcPath) and meshio.IsMesh(tgtPath)):
        print("Selected models are not a mesh")
        return None
    srcPts, srcEdges = meshio.GetVerticesEdges(srcPath)
    tgtPts, tgtEdges = meshio.GetVerticesEdges(tgtPath)
    print("Source mode has {0:d} vertices, {1:d} edges".format(srcPts.shape[0], srcEdges.shape[0]))
    print("Target mode has {0:d} vertices, {1:d} edges".format(tgtPts.shape[0], tgtEdges.shape[0]))
    P0, w = Initialise(srcPts, tgtPts)
    nn = Match(srcPts, tgtPts)
    if out is None:
        return P0, w, nn
    newPts, P = NICP(P0, w, srcPts, srcEdges, tgtPts, tol, run)
    if meshio._is_file(srcPath):
        meshio.ModifyVertices(srcPath, newPts, out=out)
    else:
        meshio.ModifyVertices(srcPath, newPts)       # inside Maya: nicp_3d.py:409 (commented out there)
    return newPts, P


if __name__ == "__main__":
    import sys
    if len(sys.argv) < 3:
        raise SystemExit("usage: python -m icp_amd.nicp_3d SRC.obj TGT.obj [OUT.obj]")
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
