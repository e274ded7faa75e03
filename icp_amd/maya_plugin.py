"""Maya plug-in shell around the drop-in hot path.

Load inside Maya with `cmds.loadPlugin(".../icp_amd/maya_plugin.py")`.  It registers the two
commands of the reference's Python plug-ins under their names -- `icp` (icp_3d.py:162-207) and
`nicp` (nicp_3d.py:412-457) -- whose `doIt` runs `icp_amd.icp_3d.main()` / `icp_amd.nicp_3d.main()`
on the two selected meshes: points out of Maya through `icp_amd.meshio`, every numeric step in
libicp_hip.so, result back into the transform node / the mesh points.

  icp        rigid ICP(src, tgt, 0.10, 50); the source's transform node is post-multiplied by the result
  nicp       what the reference's command does today: Initialise + Match (its NICP call is commented out)
  nicp -run  the full loop NICP(P0, w, src, edges, tgt, 0.01, 50) and ModifyVertices on the source

Only importable where `maya.api.OpenMaya` is (Maya's interpreter); tests drive it with a stand-in
module (tests/fake_maya.py).  Maya is a UI host: nothing here is on the measured path.
"""
import os
import sys

import maya.api.OpenMaya as om

# Maya executes a plug-in file by path, outside its package: make `icp_amd` importable
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


def maya_useNewAPI():
    """Marks the plug-in as written against Maya Python API 2.0 (icp_3d.py:162-167)."""


class PyIcp3dCmd(om.MPxCommand):
    kPluginCmdName = "icp"

    def __init__(self):
        om.MPxCommand.__init__(self)

    @staticmethod
    def cmdCreator():
        return PyIcp3dCmd()

    def doIt(self, args):
        from icp_amd import icp_3d
        self.result = icp_3d.main()


class PyNicp3dCmd(om.MPxCommand):
    kPluginCmdName = "nicp"

    def __init__(self):
        om.MPxCommand.__init__(self)

    @staticmethod
    def cmdCreator():
        return PyNicp3dCmd()

    def doIt(self, args):
        from icp_amd import nicp_3d
        full = False
        try:
            full = len(args) > 0 and args.asString(0) in ("-run", "run")
        except Exception:
            pass
        self.result = nicp_3d.main(out=True) if full else nicp_3d.main()


_COMMANDS = (PyIcp3dCmd, PyNicp3dCmd)


def initializePlugin(plugin):
    pluginFn = om.MFnPlugin(plugin)
    for cmd in _COMMANDS:
        try:
            pluginFn.registerCommand(cmd.kPluginCmdName, cmd.cmdCreator)
        except Exception:
            sys.stderr.write("Failed to register command: %s\n" % cmd.kPluginCmdName)
            raise


def uninitializePlugin(plugin):
    pluginFn = om.MFnPlugin(plugin)
    for cmd in _COMMANDS:
        try:
            pluginFn.deregisterCommand(cmd.kPluginCmdName)
        except Exception:
            sys.stderr.write("Failed to unregister command: %s\n" % cmd.kPluginCmdName)
            raise
