// C++ host core of the `nicp3d` command (reference: cplugin/nicp3d.cpp) on top of the C-ABI
// in include/icp_hip.h.  The class keeps the reference's method surface
// (nicp3d.cpp:32-51: KNN, Closest, Initialise, Match, Deform, PtoAB, RotMatrix, D_RotMatrix,
// doIt) but every numeric loop is a call into libicp_hip.so instead of TNT dense algebra
// (tnt_array2d_utils.h matmult/addrow, tnt_array1d.h subarray).  Maya (MPxCommand, MFnMesh,
// MGlobal) is replaced by plain arrays in / arrays out; see INTEGRATION.md for the Maya shim.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/icp_hip.h"

namespace icphost {

// Minimal row-major host arrays with the accessors the reference code uses on
// TNT::Array1D / Array2D (dim1(), dim2(), operator[]); plain std::vector storage.
template <class T>
class Array1D {
public:
    Array1D() = default;
    explicit Array1D(int n, T v = T()) : d_(n, v) {}
    int dim1() const { return (int)d_.size(); }
    int dim() const { return dim1(); }
    T &operator[](int i) { return d_[i]; }
    const T &operator[](int i) const { return d_[i]; }
    T *data() { return d_.data(); }
    const T *data() const { return d_.data(); }

private:
    std::vector<T> d_;
};

template <class T>
class Array2D {
public:
    Array2D() = default;
    Array2D(int m, int n, T v = T()) : m_(m), n_(n), d_((size_t)m * n, v) {}
    int dim1() const { return m_; }
    int dim2() const { return n_; }
    T *operator[](int i) { return d_.data() + (size_t)i * n_; }
    const T *operator[](int i) const { return d_.data() + (size_t)i * n_; }
    T *data() { return d_.data(); }
    const T *data() const { return d_.data(); }

private:
    int m_ = 0, n_ = 0;
    std::vector<T> d_;
};

constexpr int NUM_NN = 4;           // nicp3d.cpp:25
constexpr double INIT_C_FIT = 0.1;  // nicp3d.cpp:26-28 (unused by the reference's doIt)
constexpr double INIT_C_RIGID = 1000.0;
constexpr double INIT_C_SMOOTH = 100.0;

class nicp3d {
public:
    // reference_compat keeps Q8 (PtoAB hands every vertex vertex 0's A), Q9 (global
    // Tz = 2.0 after Initialise) and the sqrt-compare of Closest (Q12).
    explicit nicp3d(int device = 0, bool reference_compat = true);
    ~nicp3d();
    nicp3d(const nicp3d &) = delete;
    nicp3d &operator=(const nicp3d &) = delete;

    // nicp3d.cpp:67-94 without Maya: Initialise -> Match -> Deform; returns the deformed source.
    int doIt(const Array2D<double> &ptsSrc, const Array2D<int> &edgSrc, const Array2D<double> &ptsTgt,
             Array2D<double> &ptsDD);
    // the loop the plug-in never got: full NICP (nicp_3d.py:344-363) on the same state
    int runNICP(const Array2D<double> &ptsSrc, const Array2D<int> &edgSrc, const Array2D<double> &ptsTgt, double tol,
                int run, Array2D<double> &ptsDD, std::vector<double> *errHist = nullptr);

    // nicp3d.cpp:40-49
    int KNN(const Array2D<double> &dataset, const double *pt, int k, int *idx, double *dist);
    int Closest(const Array2D<double> &dataset, const double *pt, double *minDist = nullptr);
    void Initialise(const Array2D<double> &vert);
    void Match(const Array2D<double> &source, const Array2D<double> &target, Array2D<double> &matchset);
    void Deform(const Array2D<double> &dataset, Array2D<double> &deformset);
    void PtoAB(int i, Array2D<double> &A, Array1D<double> &b);
    void RotMatrix(double x, double y, double z, Array2D<double> &rot);
    void D_RotMatrix(double x, double y, double z, Array2D<double> &drotx, Array2D<double> &droty,
                     Array2D<double> &drotz);

    Array1D<double> params;         // nicp3d.cpp:38
    // nicp3d.cpp:39 holds a dense N x N `weights`; here: weights[nbr[j][i]][j] == wt[j][i]
    Array2D<int> nbr;
    Array2D<double> wt;
    double c[3] = {INIT_C_FIT, INIT_C_RIGID, INIT_C_SMOOTH};
    // Extended energy for runNICP (include/icp_hip.h nicp_set_weights / nicp_set_landmarks; not in the
    // reference): empty = absent.  fitWeight / rigidWeight per vertex, edgeStiffness per edge,
    // landmarkVertex[k] pulled towards landmarkPos[k] with cLand.  Needs reference_compat = false.
    std::vector<double> fitWeight, rigidWeight, edgeStiffness;
    std::vector<int> landmarkVertex;
    Array2D<double> landmarkPos;
    double cLand = 0.0;
    std::string lastError() const;
    // how many times KNN / Closest / Match uploaded a whole dataset since construction (they keep the device
    // copy of the last dataset / target and re-upload only when the host array changed)
    long datasetUploads() const { return uploads_; }

private:
    void check(int rc, const char *what);
    void releaseGraph();
    // Device copy of a host array, kept between calls: the reference's per-point methods receive the same
    // ref-counted Array2D on every call (nicp3d.cpp:201,266) and pay nothing for it; re-uploading N points per
    // QUERY would cost O(N) PCIe traffic per call here.  A slot is reused while the host pointer, the shape and a
    // 64-bit hash of the contents are unchanged; Initialise drops both slots.
    struct DevCache {
        const void *host = nullptr;
        int rows = 0, cols = 0;
        uint64_t hash = 0;
        void *dev = nullptr;
        size_t cap = 0;
    };
    const double *cached(DevCache &slot, const Array2D<double> &a);
    void dropCache(DevCache &slot);
    DevCache cacheA_, cacheB_;  // A: KNN / Closest dataset, Match target; B: Match source
    void *scratch_ = nullptr;   // one small device block for the per-call query / result of KNN and Closest
    long uploads_ = 0;
    icp_ctx *ctx_ = nullptr;
    nicp_graph *graph_ = nullptr;
    bool compat_;
};

// simple mesh I/O for the CLI (Wavefront OBJ subset: v, f, l)
bool loadObj(const std::string &path, Array2D<double> &vert, Array2D<int> &edges, std::string *err);
bool saveObjVertices(const std::string &path, const Array2D<double> &vert, std::string *err);

}  // namespace icphost
