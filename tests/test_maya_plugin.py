"""The Maya shell (icp_amd/maya_plugin.py + the DAG-path branches of icp_amd/meshio.py)
driven through a stand-in for maya.api.OpenMaya (tests/fake_maya.py): command registration as in
icp_3d.py:186-207 / nicp_3d.py:436-457, the selection / mesh checks of main() (icp_3d.py:138-145), and
-- on the GPU -- that `icp` moves the transform node by ICP's result and `nicp -run` writes NICP's
deformed vertices back into the mesh."""
import importlib.util
import os

import numpy as np
import pytest

import fake_maya
import icp_oracle as orc

PLUGIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "icp_amd", "maya_plugin.py")


@pytest.fixture()
def maya():
    om = fake_maya.install()
    spec = importlib.util.spec_from_file_location("icp_maya_plugin_under_test", PLUGIN)   # by path, as Maya loads it
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        yield om, mod
    finally:
        fake_maya.uninstall()


def test_registration_and_selection_checks(maya, capsys):
    om, plug = maya
    reg = {}
    plug.maya_useNewAPI()
    plug.initializePlugin(reg)
    assert sorted(reg) == ["icp", "nicp"]
    with pytest.raises(RuntimeError):
        plug.initializePlugin(reg)                       # second registration fails loudly, like Maya's
    cmd = reg["icp"]()
    assert isinstance(cmd, om.MPxCommand)
    om.MGlobal.selection = [fake_maya.DagPath("only_one", np.zeros((8, 3)))]
    cmd.doIt(om.MArgList())
    assert cmd.result is None and "You must select 2 models" in capsys.readouterr().out
    om.MGlobal.selection = [fake_maya.DagPath("a_mesh", np.zeros((8, 3))), fake_maya.DagPath("a_locator")]
    for name in ("icp", "nicp"):
        c = reg[name]()
        c.doIt(om.MArgList())
        out = capsys.readouterr().out
        assert c.result is None and "Selected models are not a mesh" in out and "a_mesh" in out
    plug.uninitializePlugin(reg)
    assert reg == {}


@pytest.mark.gpu
def test_icp_command_moves_the_transform(maya):
    om, plug = maya
    from icp_amd import icp_3d
    src, tgt = orc.make_rigid_cloud(3000, seed=4)
    reg = {}
    plug.initializePlugin(reg)
    a, b = fake_maya.DagPath("srcShape", src), fake_maya.DagPath("tgtShape", tgt)
    a.matrix = om.MMatrix(np.diag([2.0, 2.0, 2.0, 1.0]))             # an existing transform m: result must be m*t
    om.MGlobal.selection = [a, b]
    cmd = reg["icp"]()
    cmd.doIt(om.MArgList())
    T = icp_3d.ICP(src, tgt, 0.10, 50)
    assert np.array_equal(np.asarray(cmd.result), T)
    np.testing.assert_allclose(a.matrix.a, np.diag([2.0, 2.0, 2.0, 1.0]) @ T, rtol=1e-15)
    assert np.array_equal(a.points, src)                              # the points themselves stay (icp_3d.py:154-157)


@pytest.mark.gpu
def test_nicp_command(maya):
    om, plug = maya
    from icp_amd import nicp_3d
    v, edges, tgt = orc.make_grid_mesh(9, 8, seed=2)
    reg = {}
    plug.initializePlugin(reg)
    a, b = fake_maya.DagPath("srcShape", v, edges), fake_maya.DagPath("tgtShape", tgt)
    om.MGlobal.selection = [a, b]
    cmd = reg["nicp"]()
    cmd.doIt(om.MArgList())                                           # the reference's command: Initialise + Match only
    P0, w, nn = cmd.result
    assert np.array_equal(a.points, v) and nn.shape == v.shape
    saved = (nicp_3d.c_fit, nicp_3d.c_rigid, nicp_3d.c_smooth)
    try:
        cmd.doIt(om.MArgList(["-run"]))
        newPts, P = cmd.result
        nicp_3d.c_fit, nicp_3d.c_rigid, nicp_3d.c_smooth = saved
        P0b, wb = nicp_3d.Initialise(v, tgt)
        want, _ = nicp_3d.NICP(P0b, wb, v, edges.astype(np.float64), tgt, 0.01, 50)
    finally:
        nicp_3d.c_fit, nicp_3d.c_rigid, nicp_3d.c_smooth = saved
    np.testing.assert_allclose(a.points, want, rtol=1e-9, atol=1e-12)
    assert np.abs(a.points - v).max() > 1e-4                          # it did move
