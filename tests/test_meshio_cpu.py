"""Mesh I/O either side of the hot path (icp_amd/meshio.py): pure host code, no GPU."""
import numpy as np
import pytest

import icp_oracle as orc
from icp_amd import meshio

OBJ = """# a quad split in two plus a loose polyline
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0.5
v 2 2 2
vn 0 0 1
f 1/1/1 2/2/1 3/3/1
f 1 3 4
l 4 5
f -5 -4 -2
"""


def test_load_obj_edges_first_appearance(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text(OBJ)
    v, e = meshio.load_obj(str(p))
    assert v.shape == (5, 3) and v.dtype == np.float64
    np.testing.assert_array_equal(v[3], [0, 1, 0.5])
    # sides in order of first appearance, undirected, no duplicates (1-3 appears in both faces;
    # the last face uses negative indices: vertices 1, 2, 4 -> new side 2-4 only)
    assert e.tolist() == [[0, 1], [1, 2], [0, 2], [2, 3], [0, 3], [3, 4], [1, 3]]
    assert e.dtype == np.int32


def test_reference_names_on_files(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text(OBJ)
    assert meshio.GetModel(str(p)) is None                      # icp_3d.py:18-19: needs exactly two
    assert meshio.GetModel(str(p), "t.obj") == (str(p), "t.obj")
    v = meshio.GetVertices(str(p))
    v2, e2 = meshio.GetVerticesEdges(str(p))
    np.testing.assert_array_equal(v, v2)
    assert e2.dtype == np.float64 and e2.shape == (7, 2)        # Q11: the reference holds the edge table as floats
    moved = v + [0.5, -1.0, 2.0]
    out = tmp_path / "o.obj"
    meshio.ModifyVertices(str(p), moved, out=str(out))
    v3, e3 = meshio.GetVerticesEdges(str(out))
    np.testing.assert_array_equal(v3, moved)                    # %.17g round-trips doubles exactly
    np.testing.assert_array_equal(e3, e2)                       # faces kept
    assert "vn 0 0 1" in out.read_text()
    meshio.ModifyVertices(str(out), v)                          # in place
    np.testing.assert_array_equal(meshio.GetVertices(str(out)), v)
    with pytest.raises(ValueError):
        meshio.save_obj(str(tmp_path / "bad.obj"), moved[:3], template=str(p))


def test_npz_and_matrix(tmp_path):
    v, edges, tgt = orc.make_grid_mesh(5, 4, seed=1)
    p = tmp_path / "m.npz"
    np.savez(p, vertices=v, edges=edges)
    v2, e2 = meshio.GetVerticesEdges(str(p))
    np.testing.assert_array_equal(v2, v)
    np.testing.assert_array_equal(e2, edges)
    meshio.ModifyVertices(str(p), tgt)
    np.testing.assert_array_equal(meshio.GetVertices(str(p)), tgt)
    T = np.eye(4)
    T[:3, :3] = orc.rot_matrix(0.1, 0.2, 0.3)
    T[3, :3] = [1, 2, 3]
    rows = meshio.ArrayToMMatrix(T)                             # icp_3d.py:126-131 (nested tuple outside Maya)
    assert isinstance(rows, tuple) and len(rows) == 4 and rows[3][:3] == (1.0, 2.0, 3.0)
    np.testing.assert_allclose(meshio.apply_matrix(v, T), v @ T[:3, :3] + [1, 2, 3])


def test_obj_agrees_with_cxx_loader_rule(tmp_path):
    """Same edge rule as loadObj of nicp3d_core.cpp: a grid mesh written as triangles gives every
    grid side + one diagonal per cell exactly once."""
    nx, ny = 4, 3
    v, edges, _ = orc.make_grid_mesh(nx, ny, seed=0)
    idx = np.arange(nx * ny).reshape(nx, ny)
    lines = ["v %.17g %.17g %.17g" % tuple(p) for p in v]
    for i in range(nx - 1):
        for j in range(ny - 1):
            a, b, c, d = idx[i, j] + 1, idx[i + 1, j] + 1, idx[i + 1, j + 1] + 1, idx[i, j + 1] + 1
            lines += ["f %d %d %d" % (a, b, c), "f %d %d %d" % (a, c, d)]
    p = tmp_path / "g.obj"
    p.write_text("\n".join(lines) + "\n")
    v2, e2 = meshio.load_obj(str(p))
    np.testing.assert_array_equal(v2, v)
    want = {tuple(sorted(map(int, r))) for r in edges}
    got = {tuple(map(int, r)) for r in e2}
    assert got == want and len(e2) == len(edges)


def test_ply_ascii_and_binary(tmp_path):
    v, edges, tgt = orc.make_grid_mesh(4, 3, seed=2)
    # ascii PLY with float vertices + faces (uchar count, int indices) and an extra vertex property
    idx = np.arange(12).reshape(4, 3)
    faces = []
    for i in range(3):
        for j in range(2):
            faces += [(idx[i, j], idx[i + 1, j], idx[i + 1, j + 1]), (idx[i, j], idx[i + 1, j + 1], idx[i, j + 1])]
    txt = ["ply", "format ascii 1.0", "comment made by a test", "element vertex 12", "property double x", "property double y",
           "property double z", "property uchar red", "element face %d" % len(faces), "property list uchar int vertex_indices",
           "end_header"]
    txt += ["%.17g %.17g %.17g 255" % tuple(p) for p in v]
    txt += ["3 %d %d %d" % f for f in faces]
    pa = tmp_path / "a.ply"
    pa.write_text("\n".join(txt) + "\n")
    va, ea = meshio.load_ply(str(pa))
    np.testing.assert_array_equal(va, v)
    assert {tuple(r) for r in ea.tolist()} == {tuple(sorted(map(int, r))) for r in edges} and len(ea) == len(edges)
    # binary round trip through the reference's names
    pb = tmp_path / "b.ply"
    meshio.save_ply(str(pb), v, edges)
    vb, eb = meshio.GetVerticesEdges(str(pb))
    np.testing.assert_array_equal(vb, v)
    assert {tuple(sorted(map(int, r))) for r in eb} == {tuple(sorted(map(int, r))) for r in edges}
    meshio.ModifyVertices(str(pb), tgt, out=str(tmp_path / "c.ply"))
    vc, ec = meshio.GetVerticesEdges(str(tmp_path / "c.ply"))
    np.testing.assert_array_equal(vc, tgt)
    np.testing.assert_array_equal(ec, eb)
    # big-endian binary with float32 vertices
    pe = tmp_path / "e.ply"
    with open(pe, "wb") as f:
        f.write(b"ply\nformat binary_big_endian 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        f.write(np.array([[1, 2, 3], [4, 5, 6]], ">f4").tobytes())
    ve, ee = meshio.load_ply(str(pe))
    np.testing.assert_array_equal(ve, [[1, 2, 3], [4, 5, 6]])
    assert ee.shape == (0, 2)
    with pytest.raises(ValueError):
        (tmp_path / "bad.ply").write_text("plx\n")
        meshio.load_ply(str(tmp_path / "bad.ply"))
