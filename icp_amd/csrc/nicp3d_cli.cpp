// Maya-free driver of the nicp3d command core:
//   `nicp3d_cli SRC.obj TGT.obj OUT.obj [--nicp TOL RUN] [--fixed] [--landmarks FILE C_LAND]`
//   default     : what the reference command does (nicp3d.cpp:67-94): Initialise -> Match -> Deform
//   --nicp T R  : full non-rigid ICP (nicp_3d.py:344-363) with tolerance T and at most R outer iterations
//   --fixed     : self-consistent mode (no Q5/Q6/Q8/Q9/Q12 quirks)
//   --landmarks F C : landmark rows for --nicp --fixed; F holds one "vertex x y z" line per landmark (0-based vertex)
//   `nicp3d_cli --selftest DATA.obj QUERIES.obj [--fixed]` prints what the small methods of the class return
//   (RotMatrix, D_RotMatrix, PtoAB, Closest and KNN per query, KNN on fewer points than k, dataset uploads) in a
//   line format the tests parse (tests/test_gpu_hosts.py)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>

#include "nicp3d_core.h"

static void print_row(const char *tag, const double *a, int count) {
    std::printf("%s", tag);
    for (int i = 0; i < count; ++i) std::printf(" %.17g", a[i]);
    std::printf("\n");
}

static int selftest(int argc, char **argv) {
    using namespace icphost;
    bool compat = true;
    for (int i = 4; i < argc; ++i)
        if (!std::strcmp(argv[i], "--fixed")) compat = false;
    Array2D<double> data, queries;
    Array2D<int> e1, e2;
    std::string err;
    if (!loadObj(argv[2], data, e1, &err) || !loadObj(argv[3], queries, e2, &err)) {
        std::fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    try {
        nicp3d cmd(0, compat);
        Array2D<double> R, dx, dy, dz, A;
        Array1D<double> b;
        cmd.RotMatrix(0.3, -0.4, 0.5, R);
        cmd.D_RotMatrix(0.3, -0.4, 0.5, dx, dy, dz);
        print_row("R", R.data(), 9);
        print_row("dRx", dx.data(), 9);
        print_row("dRy", dy.data(), 9);
        print_row("dRz", dz.data(), 9);
        // PtoAB on a recognisable parameter vector: params[k] = k / 7
        const int nv = 5;
        cmd.params = Array1D<double>(nv * 12 + 6);
        for (int k = 0; k < cmd.params.dim1(); ++k) cmd.params[k] = k / 7.0;
        for (int i : {0, 3}) {
            cmd.PtoAB(i, A, b);
            std::printf("PtoAB %d", i);
            for (int k = 0; k < 9; ++k) std::printf(" %.17g", A.data()[k]);
            for (int k = 0; k < 3; ++k) std::printf(" %.17g", b[k]);
            std::printf("\n");
        }
        for (int q = 0; q < queries.dim1(); ++q) {
            double md = 0.0;
            const int c = cmd.Closest(data, queries[q], &md);
            std::printf("closest %d %d %.17g\n", q, c, md);
        }
        for (int q = 0; q < queries.dim1(); ++q) {
            int idx[6];
            double dist[6];
            const int got = cmd.KNN(data, queries[q], 6, idx, dist);
            std::printf("knn %d %d", q, got);
            for (int k = 0; k < 6; ++k) std::printf(" %d", idx[k]);
            for (int k = 0; k < 6; ++k) std::printf(" %.17g", dist[k]);
            std::printf("\n");
        }
        std::printf("uploads %ld\n", cmd.datasetUploads());
        // fewer points than k (nicp3d.cpp:100-137 returns the heap size)
        Array2D<double> few(4, 3);
        std::memcpy(few.data(), data.data(), sizeof(double) * 12);
        int idx[6];
        double dist[6];
        const int got = cmd.KNN(few, queries[0], 6, idx, dist);
        std::printf("knn_small %d", got);
        for (int k = 0; k < 6; ++k) std::printf(" %d", idx[k]);
        for (int k = 0; k < got; ++k) std::printf(" %.17g", dist[k]);
        std::printf("\n");
        // a changed dataset at the same address must be noticed
        data[0][0] += 1000.0;
        double md = 0.0;
        const int c = cmd.Closest(data, data[0], &md);
        std::printf("closest_after_edit %d %.17g uploads %ld\n", c, md, cmd.datasetUploads());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "nicp3d: %s\n", e.what());
        return 1;
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 4 && !std::strcmp(argv[1], "--selftest")) return selftest(argc, argv);
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
