// See nicp3d_core.h.  Host logic only: every numeric step is a libicp_hip.so call.
#include "nicp3d_core.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>
#include <stdexcept>
#include <utility>

namespace icphost {

namespace {
// RAII device buffer through icp_malloc / icp_free
struct DevBuf {
    icp_ctx *ctx;
    void *p = nullptr;
    DevBuf(icp_ctx *c, size_t bytes) : ctx(c) {
        if (icp_malloc(ctx, bytes, &p) != ICP_OK) throw std::runtime_error(icp_last_error());
    }
    ~DevBuf() { icp_free(ctx, p); }
    template <class T>
    T *as() { return (T *)p; }
};
}  // namespace

nicp3d::nicp3d(int device, bool reference_compat) : compat_(reference_compat) {
    check(icp_ctx_create(device, &ctx_), "icp_ctx_create");
}

nicp3d::~nicp3d() {
    releaseGraph();
    dropCache(cacheA_);
    dropCache(cacheB_);
    if (scratch_) icp_free(ctx_, scratch_);
    icp_ctx_destroy(ctx_);
}

void nicp3d::dropCache(DevCache &slot) {
    if (slot.dev) icp_free(ctx_, slot.dev);
    slot = DevCache();
}

// FNV-1a over the 64-bit words of the array: ~1 ns per point, against ~1 us of PCIe per kilobyte saved
static uint64_t content_hash(const double *p, size_t count) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < count; ++i) {
        uint64_t w;
        std::memcpy(&w, p + i, sizeof(w));
        h = (h ^ w) * 1099511628211ull;
    }
    return h;
}

const double *nicp3d::cached(DevCache &slot, const Array2D<double> &a) {
    const size_t count = (size_t)a.dim1() * a.dim2(), bytes = count * sizeof(double);
    const uint64_t h = content_hash(a.data(), count);
    if (slot.dev && slot.host == a.data() && slot.rows == a.dim1() && slot.cols == a.dim2() && slot.hash == h)
        return (const double *)slot.dev;
    if (slot.cap < bytes) {
        if (slot.dev) icp_free(ctx_, slot.dev);
        slot.dev = nullptr;
        slot.cap = 0;
        check(icp_malloc(ctx_, bytes, &slot.dev), "icp_malloc");
        slot.cap = bytes;
    }
    check(icp_memcpy_h2d(ctx_, slot.dev, a.data(), bytes), "h2d");
    ++uploads_;
    slot.host = a.data();
    slot.rows = a.dim1();
    slot.cols = a.dim2();
    slot.hash = h;
    return (const double *)slot.dev;
}

void nicp3d::releaseGraph() {
    if (graph_) nicp_graph_destroy(ctx_, graph_);
    graph_ = nullptr;
}

std::string nicp3d::lastError() const { return icp_last_error(); }

void nicp3d::check(int rc, const char *what) {
    if (rc != ICP_OK) throw std::runtime_error(std::string(what) + ": " + icp_last_error());
}

// nicp3d.cpp:100-137
// Like the reference, a dataset with fewer than k points yields its n points (the heap size is returned; the
// reference then pops past the empty heap, nicp3d.cpp:130-135 -- here the remaining slots hold -1 / +inf).
int nicp3d::KNN(const Array2D<double> &dataset, const double *pt, int k, int *idx, double *dist) {
    if (k > 8 || k < 1) throw std::invalid_argument("KNN: k must be in 1..8");
    const int kk = k <= 6 ? 6 : 8, n = dataset.dim1(), dim = dataset.dim2();
    if (n < 1) return 0;
    const double *dv = cached(cacheA_, dataset);
    // scratch block: [query 3 doubles | dist 8 doubles | idx 8 int32]
    if (!scratch_) check(icp_malloc(ctx_, 256, &scratch_), "icp_malloc");
    double *dq = (double *)scratch_, *dd = dq + 4;
    int32_t *di = (int32_t *)(dq + 12);
    check(icp_memcpy_h2d(ctx_, dq, pt, sizeof(double) * dim), "h2d");
    check(nicp_knn(ctx_, dv, n, dim, dq, 1, kk, di, dd), "nicp_knn");
    double back[12];  // dist (8 doubles) and idx (8 int32 = 4 doubles) in one copy
    check(icp_memcpy_d2h(ctx_, back, dd, sizeof(back)), "d2h");
    const int32_t *hi = (const int32_t *)(back + 8);
    for (int i = 0; i < k; ++i) {
        idx[i] = hi[i];
        dist[i] = back[i];
    }
    return k < n ? k : n;
}

// nicp3d.cpp:142-164 (compares sqrt distances; Q12)
int nicp3d::Closest(const Array2D<double> &dataset, const double *pt, double *minDist) {
    const int n = dataset.dim1(), dim = dataset.dim2();
    const double *dv = cached(cacheA_, dataset);
    if (!scratch_) check(icp_malloc(ctx_, 256, &scratch_), "icp_malloc");
    double *dq = (double *)scratch_, *dd = dq + 4;
    int32_t *di = (int32_t *)(dq + 12);
    check(icp_memcpy_h2d(ctx_, dq, pt, sizeof(double) * dim), "h2d");
    check(icp_closest(ctx_, dq, 1, dv, n, dim, compat_ ? ICP_COMPARE_SQRT : 0u, di, dd), "icp_closest");
    double back[9];  // d2 at [0], the index in the first int32 of [8]
    check(icp_memcpy_d2h(ctx_, back, dd, sizeof(back)), "d2h");
    int32_t i;
    std::memcpy(&i, back + 8, sizeof(i));
    if (minDist) *minDist = std::sqrt(back[0]);
    return i;
}

// nicp3d.cpp:182-224
void nicp3d::Initialise(const Array2D<double> &vert) {
    const int n = vert.dim1(), dim = vert.dim2();
    if (dim != 3) throw std::invalid_argument("Initialise: vertices must be [N,3]");
    params = Array1D<double>(n * 12 + 6, 0.0);
    for (int i = 0; i < n; ++i) {
        params[i * 12] = 1.0;
        params[i * 12 + 4] = 1.0;
        params[i * 12 + 8] = 1.0;
    }
    if (compat_) params[params.dim1() - 1] = 2.0;  // nicp3d.cpp:194 (debug translation, Q9)
    dropCache(cacheA_);
    dropCache(cacheB_);
    DevBuf dv(ctx_, sizeof(double) * n * 3), dn(ctx_, sizeof(int32_t) * n * NUM_NN), dw(ctx_, sizeof(double) * n * NUM_NN);
    check(icp_memcpy_h2d(ctx_, dv.p, vert.data(), sizeof(double) * n * 3), "h2d");
    check(nicp_knn_weights(ctx_, dv.as<double>(), n, 3, NUM_NN, dn.as<int32_t>(), dw.as<double>()), "nicp_knn_weights");
    nbr = Array2D<int>(n, NUM_NN);
    wt = Array2D<double>(n, NUM_NN);
    check(icp_memcpy_d2h(ctx_, nbr.data(), dn.p, sizeof(int32_t) * n * NUM_NN), "d2h");
    check(icp_memcpy_d2h(ctx_, wt.data(), dw.p, sizeof(double) * n * NUM_NN), "d2h");
    releaseGraph();
    check(nicp_graph_create(ctx_, dn.as<int32_t>(), dw.as<double>(), n, NUM_NN, nullptr, 0, 3, &graph_), "nicp_graph_create");
}

// nicp3d.cpp:259-272
void nicp3d::Match(const Array2D<double> &source, const Array2D<double> &target, Array2D<double> &matchset) {
    const int n = source.dim1(), m = target.dim1(), dim = source.dim2();
    matchset = Array2D<double>(n, dim);
    DevBuf di(ctx_, sizeof(int32_t) * n), dn(ctx_, sizeof(double) * n * dim);
    const double *dt = cached(cacheA_, target), *ds = cached(cacheB_, source);
    check(icp_closest(ctx_, ds, n, dt, m, dim, compat_ ? ICP_COMPARE_SQRT : 0u, di.as<int32_t>(), nullptr), "icp_closest");
    check(icp_gather_points(ctx_, dt, di.as<int32_t>(), n, dim, dn.as<double>()), "icp_gather_points");
    check(icp_memcpy_d2h(ctx_, matchset.data(), dn.p, sizeof(double) * n * dim), "d2h");
}

// nicp3d.cpp:277-371 (live lines :287-290,330-337,347-363)
void nicp3d::Deform(const Array2D<double> &dataset, Array2D<double> &deformset) {
    const int n = dataset.dim1();
    if (!graph_ || params.dim1() != n * 12 + 6) throw std::logic_error("Deform before Initialise");
    deformset = Array2D<double>(n, 3);
    // Q8: the reference's PtoAB builds A from the BASE of the params buffer (the TNT
    // Array1D -> T* conversion ignores the subarray offset), so every vertex uses vertex 0's A.
    std::vector<double> P(params.data(), params.data() + params.dim1());
    if (compat_)
        for (int i = 1; i < n; ++i) std::memcpy(&P[(size_t)i * 12], &P[0], 9 * sizeof(double));
    DevBuf dp(ctx_, sizeof(double) * P.size()), dv(ctx_, sizeof(double) * n * 3), dout(ctx_, sizeof(double) * n * 3);
    check(icp_memcpy_h2d(ctx_, dp.p, P.data(), sizeof(double) * P.size()), "h2d");
    check(icp_memcpy_h2d(ctx_, dv.p, dataset.data(), sizeof(double) * n * 3), "h2d");
    check(nicp_deform(ctx_, graph_, dp.as<double>(), dv.as<double>(), dout.as<double>()), "nicp_deform");
    check(icp_memcpy_d2h(ctx_, deformset.data(), dout.p, sizeof(double) * n * 3), "d2h");
}

// nicp3d.cpp:376-379
void nicp3d::PtoAB(int i, Array2D<double> &A, Array1D<double> &b) {
    A = Array2D<double>(3, 3);
    b = Array1D<double>(3);
    const int src = compat_ ? 0 : i;  // Q8
    for (int k = 0; k < 9; ++k) A[k / 3][k % 3] = params[src * 12 + k];
    for (int k = 0; k < 3; ++k) b[k] = params[i * 12 + 9 + k];
}

// nicp3d.cpp:384-395
void nicp3d::RotMatrix(double x, double y, double z, Array2D<double> &rot) {
    const double sx = std::sin(x), cx = std::cos(x), sy = std::sin(y), cy = std::cos(y), sz = std::sin(z), cz = std::cos(z);
    rot = Array2D<double>(3, 3);
    const double r[9] = {cy * cz, cy * sz, -sy, sx * sy * cz - cx * sz, sx * sy * sz + cx * cz, sx * cy,
                         cx * sy * cz + sx * sz, cx * sy * sz - sx * cz, cx * cy};
    std::memcpy(rot.data(), r, sizeof(r));
}

// nicp3d.cpp:400-422
void nicp3d::D_RotMatrix(double x, double y, double z, Array2D<double> &drotx, Array2D<double> &droty, Array2D<double> &drotz) {
    const double sx = std::sin(x), cx = std::cos(x), sy = std::sin(y), cy = std::cos(y), sz = std::sin(z), cz = std::cos(z);
    drotx = Array2D<double>(3, 3);
    droty = Array2D<double>(3, 3);
    drotz = Array2D<double>(3, 3);
    const double dx[9] = {0, 0, 0, cx * sy * cz + sx * sz, cx * sy * sz - sx * cz, cx * cy,
                          -sx * sy * cz + cx * sz, -sx * sy * sz - cx * cz, -sx * cy};
    const double dy[9] = {-sy * cz, -sy * sz, -cy, sx * cy * cz, sx * cy * sz, -sx * sy, cx * cy * cz, cx * cy * sz, -cx * sy};
    const double dz[9] = {-cy * sz, cy * cz, 0, -sx * sy * sz - cx * cz, sx * sy * cz - cx * sz, 0,
                          -cx * sy * sz + sx * cz, cx * sy * cz + sx * sz, 0};
    std::memcpy(drotx.data(), dx, sizeof(dx));
    std::memcpy(droty.data(), dy, sizeof(dy));
    std::memcpy(drotz.data(), dz, sizeof(dz));
}

// nicp3d.cpp:67-94
int nicp3d::doIt(const Array2D<double> &ptsSrc, const Array2D<int> &edgSrc, const Array2D<double> &ptsTgt,
                 Array2D<double> &ptsDD) {
    (void)edgSrc;  // read but never used by the reference (nicp3d.cpp:80-82)
    if (ptsSrc.dim1() < NUM_NN + 2 || ptsTgt.dim1() < 1) return -1;
    Initialise(ptsSrc);
    Array2D<double> ptsNN;
    Match(ptsSrc, ptsTgt, ptsNN);  // result unused afterwards, as in the reference
    Deform(ptsSrc, ptsDD);
    return 0;
}

int nicp3d::runNICP(const Array2D<double> &ptsSrc, const Array2D<int> &edgSrc, const Array2D<double> &ptsTgt, double tol,
                    int run, Array2D<double> &ptsDD, std::vector<double> *errHist) {
    const int n = ptsSrc.dim1(), m = ptsTgt.dim1(), e = edgSrc.dim1();
    Initialise(ptsSrc);
    params[params.dim1() - 1] = 0.0;  // start from the identity, not the debug Tz
    DevBuf dn(ctx_, sizeof(int32_t) * n * NUM_NN), dw(ctx_, sizeof(double) * n * NUM_NN), de(ctx_, sizeof(int32_t) * 2 * (e ? e : 1)),
        dv(ctx_, sizeof(double) * n * 3), dt(ctx_, sizeof(double) * m * 3), dp(ctx_, sizeof(double) * params.dim1()),
        dout(ctx_, sizeof(double) * n * 3);
    check(icp_memcpy_h2d(ctx_, dn.p, nbr.data(), sizeof(int32_t) * n * NUM_NN), "h2d");
    check(icp_memcpy_h2d(ctx_, dw.p, wt.data(), sizeof(double) * n * NUM_NN), "h2d");
    check(icp_memcpy_h2d(ctx_, de.p, edgSrc.data(), sizeof(int32_t) * 2 * e), "h2d");
    check(icp_memcpy_h2d(ctx_, dv.p, ptsSrc.data(), sizeof(double) * n * 3), "h2d");
    check(icp_memcpy_h2d(ctx_, dt.p, ptsTgt.data(), sizeof(double) * m * 3), "h2d");
    check(icp_memcpy_h2d(ctx_, dp.p, params.data(), sizeof(double) * params.dim1()), "h2d");
    releaseGraph();
    check(nicp_graph_create(ctx_, dn.as<int32_t>(), dw.as<double>(), n, NUM_NN, de.as<int32_t>(), e, 3, &graph_), "nicp_graph_create");
    {   // extended energy, when the caller filled it in
        auto sized = [](const std::vector<double> &a, int want, const char *what) {
            if (!a.empty() && (int)a.size() != want) throw std::runtime_error(std::string(what) + ": wrong length");
        };
        sized(fitWeight, n, "fitWeight");
        sized(rigidWeight, n, "rigidWeight");
        sized(edgeStiffness, e, "edgeStiffness");
        DevBuf df(ctx_, sizeof(double) * (n ? n : 1)), dg(ctx_, sizeof(double) * (n ? n : 1)), ds(ctx_, sizeof(double) * (e ? e : 1));
        if (!fitWeight.empty()) check(icp_memcpy_h2d(ctx_, df.p, fitWeight.data(), sizeof(double) * n), "h2d");
        if (!rigidWeight.empty()) check(icp_memcpy_h2d(ctx_, dg.p, rigidWeight.data(), sizeof(double) * n), "h2d");
        if (!edgeStiffness.empty()) check(icp_memcpy_h2d(ctx_, ds.p, edgeStiffness.data(), sizeof(double) * e), "h2d");
        if (!fitWeight.empty() || !rigidWeight.empty() || !edgeStiffness.empty())
            check(nicp_set_weights(ctx_, graph_, fitWeight.empty() ? nullptr : df.as<double>(),
                                   rigidWeight.empty() ? nullptr : dg.as<double>(), edgeStiffness.empty() ? nullptr : ds.as<double>()),
                  "nicp_set_weights");
        const int nl = (int)landmarkVertex.size();
        if (nl) {
            if (landmarkPos.dim1() != nl || landmarkPos.dim2() != 3) throw std::runtime_error("landmarkPos: must be [landmarks,3]");
            DevBuf dl(ctx_, sizeof(int32_t) * nl), dq(ctx_, sizeof(double) * nl * 3);
            check(icp_memcpy_h2d(ctx_, dl.p, landmarkVertex.data(), sizeof(int32_t) * nl), "h2d");
            check(icp_memcpy_h2d(ctx_, dq.p, landmarkPos.data(), sizeof(double) * nl * 3), "h2d");
            check(nicp_set_landmarks(ctx_, graph_, dl.as<int32_t>(), dq.as<double>(), nl, cLand), "nicp_set_landmarks");
        }
    }
    std::vector<double> hist(run > 0 ? run : 1);
    int iters = 0;
    check(nicp_run(ctx_, graph_, dp.as<double>(), dv.as<double>(), dt.as<double>(), m, tol, run, c,
                   compat_ ? NICP_REFERENCE_COMPAT : 0u, nullptr, dout.as<double>(), hist.data(), &iters),
          "nicp_run");
    ptsDD = Array2D<double>(n, 3);
    check(icp_memcpy_d2h(ctx_, ptsDD.data(), dout.p, sizeof(double) * n * 3), "d2h");
    check(icp_memcpy_d2h(ctx_, params.data(), dp.p, sizeof(double) * params.dim1()), "d2h");
    if (errHist) errHist->assign(hist.begin(), hist.begin() + iters);
    return iters;
}

// ------------------------------------------------------------------ OBJ subset
bool loadObj(const std::string &path, Array2D<double> &vert, Array2D<int> &edges, std::string *err) {
    std::ifstream in(path);
    if (!in) {
        if (err) *err = "cannot open " + path;
        return false;
    }
    std::vector<double> v;
    std::set<std::pair<int, int>> es;
    std::vector<std::pair<int, int>> order;
    auto add_edge = [&](int a, int b) {
        if (a == b) return;
        std::pair<int, int> k(a < b ? a : b, a < b ? b : a);
        if (es.insert(k).second) order.push_back(k);
    };
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        ls >> tag;
        if (tag == "v") {
            double x, y, z;
            if (ls >> x >> y >> z) {
                v.push_back(x);
                v.push_back(y);
                v.push_back(z);
            }
        } else if (tag == "f" || tag == "l") {
            std::vector<int> ids;
            std::string tok;
            while (ls >> tok) {
                int id = std::atoi(tok.substr(0, tok.find('/')).c_str());
                ids.push_back(id > 0 ? id - 1 : (int)(v.size() / 3) + id);
            }
            for (size_t k = 0; k + 1 < ids.size(); ++k) add_edge(ids[k], ids[k + 1]);
            if (tag == "f" && ids.size() > 2) add_edge(ids.back(), ids.front());
        }
    }
    const int n = (int)(v.size() / 3);
    vert = Array2D<double>(n, 3);
    std::memcpy(vert.data(), v.data(), v.size() * sizeof(double));
    edges = Array2D<int>((int)order.size(), 2);
    for (size_t k = 0; k < order.size(); ++k) {
        if (order[k].first < 0 || order[k].second >= n) {
            if (err) *err = "face/line index out of range in " + path;
            return false;
        }
        edges[(int)k][0] = order[k].first;
        edges[(int)k][1] = order[k].second;
    }
    return true;
}

bool saveObjVertices(const std::string &path, const Array2D<double> &vert, std::string *err) {
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f) {
        if (err) *err = "cannot write " + path;
        return false;
    }
    for (int i = 0; i < vert.dim1(); ++i) std::fprintf(f, "v %.17g %.17g %.17g\n", vert[i][0], vert[i][1], vert[i][2]);
    std::fclose(f);
    return true;
}

}  // namespace icphost
