"""Drop-in for the hot-path functions of the reference's icp_3d.py (3-D rigid
point-to-point ICP): same names, positional arguments and return shapes, computed
by libicp_hip.so on an MI355X.

numpy in -> numpy out (like the reference).  torch CUDA tensors in -> CUDA tensors
out, with no host round trip (use that form to keep clouds resident in HBM).
The mesh glue of the reference (GetModel, GetVertices, ArrayToMMatrix; icp_3d.py:16-34,
126-131) is in icp_amd.meshio under the same names: Maya DAG paths inside Maya, file
names (.obj / .npz) elsewhere; `main(src, tgt, out)` below is icp_3d.py:136-158 on files.
The `icp` MPxCommand class (icp_3d.py:160-207) needs Maya to register and is not built.
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
    r = rigid.rigid_icp(source, target, tol, run, coherent=len(source) >= rigid.COHERENT_MIN_ROWS)
    n = int(r["iters"].cpu()[0])
    for i, e in enumerate(r["err_hist"].cpu().numpy()[:n]):
        print("err = %f (%2d)" % (e, i))
    return _out(r["T"], as_np)


def main(srcPath=None, tgtPath=None, out=None, tol=0.10, run=50):
    """icp_3d.py:136-158 with files in place of the Maya selection: loads both meshes, runs
    ICP(src, tgt, 0.10, 50), and (where the reference multiplies the transform node by the
    result) writes the moved source to `out`.  Returns the 4x4 transform."""
    from . import meshio
    print("ICP - Iterative Closest Point")
    sel = meshio.GetModel(*[p for p in (srcPath, tgtPath) if p is not None])
    if sel is None:
        print("You must select 2 models")
        return None
    srcPath, tgtPath = sel
    print("Source model is %s" % meshio.PathName(srcPath))
    print("Target model is %s" % meshio.PathName(tgtPath))
    if not (meshio.IsMesh(srcPath) and meshio.IsMesh(tgtPath)):
        print("Selected models are not a mesh")
        return None
    srcPts = meshio.GetVertices(srcPath)
    tgtPts = meshio.GetVertices(tgtPath)
    print("Source mode has %d vertices" % srcPts.shape[0])
    print("Target mode has %d vertices" % tgtPts.shape[0])
    Trans = ICP(srcPts, tgtPts, tol, run)
    t = meshio.ArrayToMMatrix(Trans)
    if not meshio._is_file(srcPath):
        meshio.ApplyToTransform(srcPath, t)          # inside Maya: icp_3d.py:154-157
    elif out is not None:
        meshio.ModifyVertices(srcPath, meshio.apply_matrix(srcPts, Trans), out=out)
    return Trans


if __name__ == "__main__":
    import sys
    if len(sys.argv) < 3:
        raise SystemExit("usage: python -m icp_amd.icp_3d SRC.obj TGT.obj [OUT.obj]")
    print(main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None))
