// Maya-free driver of the nicp3d command core:
//   `nicp3d_cli SRC.obj TGT.obj OUT.obj [--nicp TOL RUN] [--fixed] [--landmarks FILE C_LAND]`
//   default     : what the reference command does (nicp3d.cpp:67-94): Initialise -> Match -> Deform
//   --nicp T R  : full non-rigid ICP (nicp_3d.py:344-363) with tolerance T and at most R outer iterations
//   --fixed     : self-consistent mode (no Q5/Q6/Q8/Q9/Q12 quirks)
//   --landmarks F C : landmark rows for --nicp --fixed; F holds one "vertex x y z" line per landmark (0-based vertex)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>

#include "nicp3d_core.h"

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s SRC.obj TGT.obj OUT.obj [--nicp TOL RUN] [--fixed] [--landmarks FILE C_LAND]\n", argv[0]);
        return 2;
    }
    bool compat = true, full = false;
    double tol = 0.01;
    int run = 50;
    const char *lmFile = nullptr;
    double cLand = 0.0;
    for (int i = 4; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--fixed")) compat = false;
        else if (!std::strcmp(argv[i], "--nicp") && i + 2 < argc) {
            full = true;
            tol = std::atof(argv[++i]);
            run = std::atoi(argv[++i]);
        } else if (!std::strcmp(argv[i], "--landmarks") && i + 2 < argc) {
            lmFile = argv[++i];
            cLand = std::atof(argv[++i]);
        }
    }
    using namespace icphost;
    Array2D<double> ptsSrc, ptsTgt, ptsDD;
    Array2D<int> edgSrc, edgTgt;
    std::string err;
    if (!loadObj(argv[1], ptsSrc, edgSrc, &err) || !loadObj(argv[2], ptsTgt, edgTgt, &err)) {
        std::fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    // the two displayInfo lines of nicp3d.cpp:83-86
    std::printf("Source mesh has %d vertics, %d edges\n", ptsSrc.dim1(), edgSrc.dim1());
    std::printf("Target mesh has %d vertics, %d edges\n", ptsTgt.dim1(), edgTgt.dim1());
    try {
        nicp3d cmd(0, compat);
        if (lmFile) {
            std::ifstream in(lmFile);
            std::vector<double> pos;
            int vtx;
            double x, y, z;
            while (in >> vtx >> x >> y >> z) {
                cmd.landmarkVertex.push_back(vtx);
                pos.insert(pos.end(), {x, y, z});
            }
            if (cmd.landmarkVertex.empty()) {
                std::fprintf(stderr, "no landmarks read from %s\n", lmFile);
                return 1;
            }
            cmd.landmarkPos = Array2D<double>((int)cmd.landmarkVertex.size(), 3);
            std::memcpy(cmd.landmarkPos.data(), pos.data(), pos.size() * sizeof(double));
            cmd.cLand = cLand;
        }
        if (full) {
            std::vector<double> hist;
            int it = cmd.runNICP(ptsSrc, edgSrc, ptsTgt, tol, run, ptsDD, &hist);
            for (int i = 0; i < it; ++i) std::printf("%2d error = %f\n", i, hist[i]);
        } else if (cmd.doIt(ptsSrc, edgSrc, ptsTgt, ptsDD) != 0) {
            std::fprintf(stderr, "2 models must be selected.\n");
            return 1;
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "nicp3d: %s\n", e.what());
        return 1;
    }
    if (!saveObjVertices(argv[3], ptsDD, &err)) {
        std::fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    return 0;
}
