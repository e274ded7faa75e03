"""The step either side of the hot path: getting
`[N,3]` vertices and `[E,2]` edges in, and deformed vertices / the 4x4 transform out.

The reference does this through Maya (`GetModel(s)`, `GetVertices`, `GetVerticesEdges`,
`ModifyVertices`, `ArrayToMMatrix`: icp_3d.py:16-34,126-131; nicp_3d.py:25-47,368-385;
nicp3d.cpp:169-177,230-253,424-431).  The same five names are kept here.  A "path" is
either a Maya `MDagPath` -- then the calls go to `maya.api.OpenMaya` exactly as the
reference's do, which needs a Maya interpreter -- or a file name (`.obj`, `.ply`, `.npz`), which
is how the path is driven everywhere else.  Pure host code: no GPU, no libicp_hip.

OBJ subset: `v x y z`, `f a b c ...` (also `a/b/c` tokens, negative indices), `l a b ...`;
edges are the unique undirected polygon sides in order of first appearance, the same rule
as `loadObj` of icp_amd/csrc/nicp3d_core.cpp (Maya's own edge numbering is not
recoverable from a file; NICP's energy does not depend on the edge order).
"""
import os

import numpy as np


def _maya():
    try:
        import maya.api.OpenMaya as om   # only inside Maya
        return om
    except Exception:
        return None


def _is_file(path):
    return isinstance(path, (str, bytes, os.PathLike))


# --------------------------------------------------------------------------- files
def load_obj(path):
    """(vertices [N,3] float64, edges [E,2] int32) of a Wavefront OBJ."""
    verts, order, seen = [], [], set()

    def add_edge(a, b):
        if a == b:
            return
        k = (a, b) if a < b else (b, a)
        if k not in seen:
            seen.add(k)
            order.append(k)

    with open(path, "r") as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v" and len(tok) >= 4:
                verts.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif tok[0] in ("f", "l"):
                ids = []
                for t in tok[1:]:
                    i = int(t.split("/")[0])
                    ids.append(i - 1 if i > 0 else len(verts) + i)
                for a, b in zip(ids[:-1], ids[1:]):
                    add_edge(a, b)
                if tok[0] == "f" and len(ids) > 2:
                    add_edge(ids[-1], ids[0])
    v = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    e = np.asarray(order, dtype=np.int32).reshape(-1, 2)
    if len(e) and (e.min() < 0 or e.max() >= len(v)):
        raise ValueError("%s: face index outside the vertex list" % path)
    return v, e


def save_obj(path, vertices, template=None):
    """Write vertices; with `template` (an OBJ path) every non-vertex line of it is kept, so
    faces / groups / materials survive and only the positions change (what
    `ModifyVertices` does to a Maya mesh, nicp_3d.py:378-385)."""
    vertices = np.asarray(vertices, dtype=np.float64)
    lines = ["v %.17g %.17g %.17g\n" % tuple(p) for p in vertices]
    if template is None:
        with open(path, "w") as f:
            f.writelines(lines)
        return
    out, k = [], 0
    with open(template, "r") as f:
        for line in f:
            if line.split()[:1] == ["v"]:
                if k >= len(lines):
                    raise ValueError("template has more vertices than the array")
                out.append(lines[k])
                k += 1
            else:
                out.append(line)
    if k != len(lines):
        raise ValueError("template has %d vertices, array has %d" % (k, len(lines)))
    with open(path, "w") as f:
        f.writelines(out)


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def load_ply(path):
    """(vertices [N,3] float64, edges [E,2] int32) of a PLY file: ascii or binary (either byte
    order), `vertex` element with x/y/z plus a `face` element with one index list (and/or an
    `edge` element with vertex1/vertex2).  Edges: polygon sides in order of first appearance,
    as for OBJ, then the explicit edge element."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: unterminated PLY header" % path)
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append([tok[1], int(tok[2]), []])
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1][2].append((tok[4], "list", tok[2], tok[3]))
                else:
                    elements[-1][2].append((tok[2], tok[1]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError("%s: unknown PLY format %r" % (path, fmt))
        bo = "<" if fmt != "binary_big_endian" else ">"
        verts, polys, explicit = None, [], []
        text = f.read().split() if fmt == "ascii" else None
        pos = 0
        for name, count, props in elements:
            simple = all(p[1] != "list" for p in props)
            if fmt != "ascii" and simple:
                dt = np.dtype([(p[0], bo + _PLY_TYPES[p[1]]) for p in props])
                rows = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
                get = lambda key: rows[key].astype(np.float64)
            else:
                cols = {p[0]: [] for p in props}
                for _ in range(count):
                    for prop in props:
                        if prop[1] == "list":
                            if fmt == "ascii":
                                k = int(text[pos]); pos += 1
                                vals = [int(float(x)) for x in text[pos:pos + k]]; pos += k
                            else:
                                k = int(np.frombuffer(f.read(np.dtype(_PLY_TYPES[prop[2]]).itemsize), bo + _PLY_TYPES[prop[2]])[0])
                                it = np.dtype(bo + _PLY_TYPES[prop[3]])
                                vals = np.frombuffer(f.read(it.itemsize * k), it).astype(np.int64).tolist()
                            cols[prop[0]].append(vals)
                        else:
                            if fmt == "ascii":
                                cols[prop[0]].append(float(text[pos])); pos += 1
                            else:
                                it = np.dtype(bo + _PLY_TYPES[prop[1]])
                                cols[prop[0]].append(float(np.frombuffer(f.read(it.itemsize), it)[0]))
                get = lambda key: np.asarray(cols[key], dtype=np.float64)
                if name == "face":
                    lists = [p[0] for p in props if p[1] == "list"]
                    if lists:
                        polys = cols[lists[0]]
            if name == "vertex":
                verts = np.stack([get("x"), get("y"), get("z")], 1)
            elif name == "edge" and count:
                explicit = np.stack([get("vertex1"), get("vertex2")], 1).astype(np.int64).tolist()
    if verts is None:
        raise ValueError("%s: no vertex element" % path)
    order, seen = [], set()
    def add_edge(a, b):
        if a == b:
            return
        k = (a, b) if a < b else (b, a)
        if k not in seen:
            seen.add(k)
            order.append(k)
    for ids in polys:
        for a, b in zip(ids[:-1], ids[1:]):
            add_edge(a, b)
        if len(ids) > 2:
            add_edge(ids[-1], ids[0])
    for a, b in explicit:
        add_edge(int(a), int(b))
    e = np.asarray(order, dtype=np.int32).reshape(-1, 2)
    if len(e) and (e.min() < 0 or e.max() >= len(verts)):
        raise ValueError("%s: index outside the vertex list" % path)
    return np.ascontiguousarray(verts, dtype=np.float64), e


def save_ply(path, vertices, edges=None):
    """Binary little-endian PLY with double-precision vertices (+ an `edge` element)."""
    v = np.ascontiguousarray(vertices, dtype="<f8").reshape(-1, 3)
    e = np.zeros((0, 2), "<i4") if edges is None else np.ascontiguousarray(np.asarray(edges).astype("<i4")).reshape(-1, 2)
    with open(path, "wb") as f:
        head = ["ply", "format binary_little_endian 1.0", "element vertex %d" % len(v), "property double x",
                "property double y", "property double z"]
        if len(e):
            head += ["element edge %d" % len(e), "property int vertex1", "property int vertex2"]
        f.write(("\n".join(head) + "\nend_header\n").encode("ascii"))
        f.write(v.tobytes())
        f.write(e.tobytes())


def load_npz(path):
    d = np.load(path)     # allow_pickle stays False
    v = np.asarray(d["vertices"], dtype=np.float64)
    e = np.asarray(d["edges"], dtype=np.int32).reshape(-1, 2) if "edges" in d.files else np.zeros((0, 2), np.int32)
    return v, e


def _load(path):
    p = os.fspath(path)
    if isinstance(p, bytes):
        p = p.decode()
    if p.lower().endswith(".npz"):
        return load_npz(p)
    if p.lower().endswith(".ply"):
        return load_ply(p)
    return load_obj(p)


# --------------------------------------------------------------------------- the reference's names
def GetModel(*paths):
    """icp_3d.py:16-21.  Inside Maya with no arguments: the two selected DAG paths (or None).
    Elsewhere: the two file names given."""
    om = _maya()
    if not paths and om is not None:
        sel = om.MGlobal.getActiveSelectionList()
        if sel.length() != 2:
            return None
        return sel.getDagPath(0), sel.getDagPath(1)
    if len(paths) != 2:
        return None
    return paths[0], paths[1]


def GetVertices(path):
    """icp_3d.py:26-34: the mesh's points as [N,3] float64."""
    if _is_file(path):
        return _load(path)[0]
    om = _maya()
    pts = om.MFnMesh(path).getPoints(om.MSpace.kWorld)
    return np.array([[p.x, p.y, p.z] for p in pts], dtype=np.float64).reshape(-1, 3)


def GetVerticesEdges(path):
    """nicp_3d.py:35-47: (v [N,3], e [E,2]).  The reference returns the edge table as floats
    (Q11); so does this, the consumers cast it."""
    if _is_file(path):
        v, e = _load(path)
        return v, e.astype(np.float64)
    om = _maya()
    mesh = om.MFnMesh(path)
    pts = mesh.getPoints(om.MSpace.kWorld)
    v = np.array([[p.x, p.y, p.z] for p in pts], dtype=np.float64).reshape(-1, 3)
    e = np.array([mesh.getEdgeVertices(i) for i in range(mesh.numEdges)], dtype=np.float64).reshape(-1, 2)
    return v, e


def ModifyVertices(path, dset, out=None):
    """nicp_3d.py:378-385: move the mesh's points to dset.  For a file, writes `out` (default:
    overwrite `path`) keeping the faces of the original."""
    dset = np.asarray(dset.cpu() if hasattr(dset, "is_cuda") else dset, dtype=np.float64)
    if _is_file(path):
        p = os.fspath(path)
        if p.lower().endswith(".npz"):
            v, e = load_npz(p)
            np.savez(out or p, vertices=dset, edges=e)
        elif p.lower().endswith(".ply"):
            v, e = load_ply(p)
            save_ply(out or p, dset, e)
        else:
            tmp = (out or p) + ".tmp"
            save_obj(tmp, dset, template=p)
            os.replace(tmp, out or p)
        return
    om = _maya()
    om.MFnMesh(path).setPoints([om.MPoint(*row) for row in dset], om.MSpace.kWorld)


def IsMesh(path):
    """The `hasFn(om.MFn.kMesh)` test of icp_3d.py:144 / nicp_3d.py:398; files always are."""
    if _is_file(path):
        return True
    om = _maya()
    return bool(path.hasFn(om.MFn.kMesh))


def PathName(path):
    """`partialPathName()` of a DAG path (icp_3d.py:142-143), or the file name."""
    return path if _is_file(path) else path.partialPathName()


def ApplyToTransform(path, t):
    """icp_3d.py:154-157: post-multiply the transform node of a DAG path by the MMatrix t."""
    om = _maya()
    transform = om.MFnTransform(path)
    m = transform.transformation().asMatrix()
    transform.setTransformation(om.MTransformationMatrix(m * t))


def ArrayToMMatrix(arr):
    """icp_3d.py:126-131: the 4x4 row-vector transform as an MMatrix inside Maya, else as the
    same nested tuple MMatrix is built from."""
    arr = np.asarray(arr.cpu() if hasattr(arr, "is_cuda") else arr, dtype=np.float64)
    rows = tuple(tuple(float(arr[r, c]) for c in range(4)) for r in range(4))
    om = _maya()
    return om.MMatrix(rows) if om is not None else rows


def apply_matrix(vertices, T):
    """Row-vector convention of the reference: [x y z 1] . T (what `m*t` does to the
    transform node in icp_3d.py:153-157)."""
    v = np.asarray(vertices, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64)
    return v @ T[:3, :3] + T[3, :3]
