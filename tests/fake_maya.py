"""A stand-in for the handful of `maya.api.OpenMaya` names the Maya glue touches
(icp_amd/meshio.py, icp_amd/maya_plugin.py), so that glue can be exercised without Autodesk Maya.
A test double, nothing more: meshes are numpy arrays held in memory, matrices are 4x4 row-vector
numpy arrays.  install() puts it into sys.modules, uninstall() removes it again."""
import sys
import types

import numpy as np


class MPoint:
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = float(x), float(y), float(z)


class MSpace:
    kWorld = 4


class MFn:
    kMesh = 296
    kTransform = 110


class MMatrix:
    def __init__(self, rows=None):
        self.a = np.eye(4) if rows is None else np.array(rows, dtype=np.float64).reshape(4, 4)

    def __mul__(self, other):
        return MMatrix(self.a @ other.a)


class MTransformationMatrix:
    def __init__(self, m=None):
        self.m = m if m is not None else MMatrix()

    def asMatrix(self):
        return self.m


class DagPath:
    """what MSelectionList.getDagPath returns: a named node with (maybe) a mesh under it"""

    def __init__(self, name, points=None, edges=None):
        self.name = name
        self.points = None if points is None else np.array(points, dtype=np.float64)
        self.edges = np.zeros((0, 2), np.int64) if edges is None else np.asarray(edges).astype(np.int64)
        self.matrix = MMatrix()

    def hasFn(self, fn):
        return fn == MFn.kTransform or (fn == MFn.kMesh and self.points is not None)

    def partialPathName(self):
        return self.name


class MFnMesh:
    def __init__(self, path):
        self.p = path

    def getPoints(self, space):
        return [MPoint(*row) for row in self.p.points]

    def setPoints(self, pts, space):
        assert len(pts) == len(self.p.points)
        self.p.points = np.array([[q.x, q.y, q.z] for q in pts])

    @property
    def numEdges(self):
        return len(self.p.edges)

    def getEdgeVertices(self, i):
        return (int(self.p.edges[i, 0]), int(self.p.edges[i, 1]))


class MFnTransform:
    def __init__(self, path):
        self.p = path

    def transformation(self):
        return MTransformationMatrix(self.p.matrix)

    def setTransformation(self, tm):
        self.p.matrix = tm.asMatrix()


class MSelectionList:
    def __init__(self, items):
        self.items = list(items)

    def length(self):
        return len(self.items)

    def getDagPath(self, i):
        return self.items[i]


class MGlobal:
    selection = []

    @staticmethod
    def getActiveSelectionList():
        return MSelectionList(MGlobal.selection)


class MPxCommand:
    def __init__(self):
        pass


class MFnPlugin:
    """`plugin` is a dict standing for Maya's command registry"""

    def __init__(self, plugin):
        self.reg = plugin

    def registerCommand(self, name, creator):
        if name in self.reg:
            raise RuntimeError("command %s already registered" % name)
        self.reg[name] = creator

    def deregisterCommand(self, name):
        del self.reg[name]


class MArgList(list):
    def asString(self, i):
        return str(self[i])


_NAMES = ("MPoint", "MSpace", "MFn", "MMatrix", "MTransformationMatrix", "MFnMesh", "MFnTransform", "MSelectionList",
          "MGlobal", "MPxCommand", "MFnPlugin", "MArgList")


def install():
    maya, api, om = types.ModuleType("maya"), types.ModuleType("maya.api"), types.ModuleType("maya.api.OpenMaya")
    for n in _NAMES:
        setattr(om, n, globals()[n])
    maya.api, api.OpenMaya = api, om
    sys.modules.update({"maya": maya, "maya.api": api, "maya.api.OpenMaya": om})
    MGlobal.selection = []
    return om


def uninstall():
    for k in ("maya", "maya.api", "maya.api.OpenMaya"):
        sys.modules.pop(k, None)
