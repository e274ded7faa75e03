// Maya command shell of the rewritten nicp3d core: what
// cplugin/nicp3d.cpp:32-36,67-94,169-177,230-253,424-446 does around its numeric methods --
// selection, mesh points / edges out of Maya, deformed points back in, command registration --
// with every numeric step delegated to icphost::nicp3d (nicp3d_core.cpp -> libicp_hip.so).
//
// Compiled only where the Maya devkit is present:
//     make -C icp_amd/csrc maya MAYA_LOCATION=/usr/autodesk/maya
// (defines HAVE_MAYA; the devkit is absent from the build image of this repository, so this file is
// not part of the default build and has not been compiled there.)
//
//   nicp3d          the reference command: Initialise -> Match -> Deform, points written back
//   nicp3d -run     the full non-rigid ICP loop (nicp_3d.py:344-363) the plug-in never reached
#ifdef HAVE_MAYA
#include <maya/MArgList.h>
#include <maya/MDagPath.h>
#include <maya/MFnMesh.h>
#include <maya/MFnPlugin.h>
#include <maya/MGlobal.h>
#include <maya/MPointArray.h>
#include <maya/MPxCommand.h>
#include <maya/MSelectionList.h>
#include <maya/MString.h>

#include <exception>

#include "nicp3d_core.h"

namespace {

using icphost::Array2D;

// nicp3d.cpp:169-177: the first two selected DAG paths
bool selectedPair(MDagPath &src, MDagPath &tgt) {
    MSelectionList sel;
    MGlobal::getActiveSelectionList(sel);
    if (sel.length() != 2) return false;
    return sel.getDagPath(0, src) == MS::kSuccess && sel.getDagPath(1, tgt) == MS::kSuccess;
}

// nicp3d.cpp:230-253: world-space points [n,3] and the edge table [e,2]
MStatus meshArrays(const MDagPath &path, Array2D<double> &vert, Array2D<int> &edges) {
    MStatus st;
    MFnMesh mesh(path, &st);
    if (st != MS::kSuccess) return st;
    MPointArray pts;
    mesh.getPoints(pts, MSpace::kWorld);
    vert = Array2D<double>((int)pts.length(), 3);
    for (unsigned i = 0; i < pts.length(); ++i) {
        vert[i][0] = pts[i].x;
        vert[i][1] = pts[i].y;
        vert[i][2] = pts[i].z;
    }
    edges = Array2D<int>(mesh.numEdges(), 2);
    for (int i = 0; i < mesh.numEdges(); ++i) {
        int2 ends;
        mesh.getEdgeVertices(i, ends);
        edges[i][0] = ends[0];
        edges[i][1] = ends[1];
    }
    return MS::kSuccess;
}

// nicp3d.cpp:424-431
MStatus writePoints(const MDagPath &path, const Array2D<double> &vert) {
    MStatus st;
    MFnMesh mesh(path, &st);
    if (st != MS::kSuccess) return st;
    MPointArray pts((unsigned)vert.dim1());
    for (int i = 0; i < vert.dim1(); ++i) pts.set(MPoint(vert[i][0], vert[i][1], vert[i][2]), (unsigned)i);
    return mesh.setPoints(pts, MSpace::kWorld);
}

class nicp3dCmd : public MPxCommand {
public:
    static void *creator() { return new nicp3dCmd; }

    MStatus doIt(const MArgList &args) override {
        bool full = false;
        for (unsigned i = 0; i < args.length(); ++i)
            if (args.asString(i) == "-run") full = true;
        MDagPath dagSrc, dagTgt;
        if (!selectedPair(dagSrc, dagTgt)) {
            MGlobal::displayInfo("2 models must be selected.");  // nicp3d.cpp:72-77
            return MS::kFailure;
        }
        Array2D<double> ptsSrc, ptsTgt, ptsDD;
        Array2D<int> edgSrc, edgTgt;
        if (meshArrays(dagSrc, ptsSrc, edgSrc) != MS::kSuccess || meshArrays(dagTgt, ptsTgt, edgTgt) != MS::kSuccess) {
            MGlobal::displayInfo("Selected models are not meshes.");
            return MS::kFailure;
        }
        MGlobal::displayInfo(MString("Source mesh has ") + ptsSrc.dim1() + " vertics, " + edgSrc.dim1() + " edges");
        MGlobal::displayInfo(MString("Target mesh has ") + ptsTgt.dim1() + " vertics, " + edgTgt.dim1() + " edges");
        try {
            icphost::nicp3d core(0, /*reference_compat=*/!full);
            if (full) {
                core.runNICP(ptsSrc, edgSrc, ptsTgt, 0.01, 50, ptsDD);
            } else if (core.doIt(ptsSrc, edgSrc, ptsTgt, ptsDD) != 0) {
                return MS::kFailure;
            }
        } catch (const std::exception &e) {
            MGlobal::displayError(MString("nicp3d: ") + e.what());
            return MS::kFailure;
        }
        return writePoints(dagSrc, ptsDD);
    }
};

}  // namespace

// nicp3d.cpp:434-446: same command name, vendor and version strings
MStatus initializePlugin(MObject obj) {
    MFnPlugin plugin(obj, "Registration", "3.0", "Any");
    return plugin.registerCommand("nicp3d", nicp3dCmd::creator);
}

MStatus uninitializePlugin(MObject obj) {
    MFnPlugin plugin(obj);
    return plugin.deregisterCommand("nicp3d");
}
#endif  // HAVE_MAYA
