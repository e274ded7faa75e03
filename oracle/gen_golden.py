#!/usr/bin/env python3
"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN FUNCTION BODIES.

Test infrastructure only; runs in the build container (needs /root/reference,
which does not exist on the GPU box).  Nothing from the reference is copied into
the repository: its files are read as text at run time, the line ranges that hold
the pure hot-path functions are sliced out, the Python-2 ``print`` statements
inside those ranges (the only py2-isms there) are replaced by ``pass``, and the
result is exec'd under Python 3 with numpy/scipy injected.  The outputs (data
only) are written as small .npz fixtures together with the numpy/scipy versions.

The reference has no tests or golden vectors of its own; these
fixtures are what pins oracle/icp_oracle.py (tests/test_oracle_golden.py) and,
through it, the HIP path.

Usage:  python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import hashlib
import os
import sys

import numpy as np
import scipy
from scipy import optimize

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import icp_oracle as orc  # noqa: E402  (only for the seeded input generators)


def load_ref(path, ranges, extra_globals=None):
    """exec the given 1-based inclusive line ranges of a reference file."""
    with open(path) as f:
        lines = f.read().split("\n")
    picked = []
    for a, b in ranges:
        for ln in lines[a - 1:b]:
            stripped = ln.lstrip()
            if stripped.startswith("print ") or stripped == "print":
                ln = ln[:len(ln) - len(stripped)] + "pass"
            picked.append(ln)
        picked.append("")
    if not hasattr(np, "product"):
        np.product = np.prod  # numpy<2 name used at nicp_row_rt.py:98
    g = {"np": np, "optimize": optimize}
    if extra_globals:
        g.update(extra_globals)
    exec(compile("\n".join(picked), path, "exec"), g)
    return g


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    meta = dict(numpy=np.__version__, scipy=scipy.__version__)

    icp3 = load_ref(os.path.join(args.ref, "icp_3d.py"), [(39, 121)])
    icprow = load_ref(os.path.join(args.ref, "icp_row.py"), [(50, 107)])
    icpcol = load_ref(os.path.join(args.ref, "icp.py"), [(18, 21)])
    n3 = load_ref(os.path.join(args.ref, "nicp_3d.py"), [(17, 20), (53, 65), (83, 114), (150, 363)])
    # nicp_3d.Match prints each row (nicp_3d.py:76) -> take the silent icp_3d twin
    n3["Match"] = icp3["Match"]
    nrow = load_ref(os.path.join(args.ref, "nicp_row.py"), [(26, 29), (66, 230)])
    nrt = load_ref(os.path.join(args.ref, "nicp_row_rt.py"), [(55, 112)])

    def save(name, **kw):
        kw["meta_numpy"] = np.array(meta["numpy"])
        kw["meta_scipy"] = np.array(meta["scipy"])
        np.savez_compressed(os.path.join(args.out, name + ".npz"), **kw)
        print("wrote", name, {k: getattr(v, "shape", None) for k, v in kw.items() if not k.startswith("meta")})

    def ref_match_idx(Closest, src, tgt):
        return np.array([Closest(tgt, src[i]) for i in range(src.shape[0])], dtype=np.int32)

    # ---------------- 1. closest / match ----------------
    src, tgt = orc.make_rigid_cloud(1000, seed=0)
    idx = ref_match_idx(icp3["Closest"], src, tgt)
    nn = icp3["Match"](src, tgt)
    assert np.array_equal(nn, tgt[idx])
    save("closest_1k_normal3", src=src, tgt=tgt, idx=idx)

    rng = np.random.default_rng(11)
    src = rng.uniform(-1, 1, (4096, 3))
    tgt = rng.uniform(-1, 1, (50000, 3))
    idx = ref_match_idx(icp3["Closest"], src, tgt)
    save("closest_4k_50k_uniform3", seed=np.array(11), idx=idx, src_sha=np.array(sha(src)), tgt_sha=np.array(sha(tgt)))

    rng = np.random.default_rng(12)
    src2 = rng.standard_normal((20, 2))
    tgt2 = rng.standard_normal((50, 2))
    save("closest_20_50_dim2", src=src2, tgt=tgt2, idx=ref_match_idx(icprow["Closest"], src2, tgt2))

    # crafted ties: integer lattice targets (each duplicated later in the array) and
    # queries at cell centres / edge midpoints -> many exactly equidistant targets
    g = np.stack(np.meshgrid(np.arange(6.), np.arange(6.), np.arange(6.), indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(13)
    tgt_t = np.concatenate([g[rng.permutation(len(g))], g[rng.permutation(len(g))]], 0)
    q = np.concatenate([g[:100] + 0.5, g[:100] + np.array([0.5, 0, 0]), g[:50], rng.uniform(0, 5, (50, 3))], 0)
    save("closest_ties3", src=q, tgt=tgt_t, idx=ref_match_idx(icp3["Closest"], q, tgt_t))

    # near ties: target pairs whose squared distance to the query differs by ~1 ulp
    rng = np.random.default_rng(14)
    base = rng.standard_normal((300, 3))
    qn = rng.standard_normal((64, 3))
    near = []
    for i in range(64):
        d = rng.standard_normal(3)
        d /= np.linalg.norm(d)
        r = 0.01 * (1 + rng.random())
        p1 = qn[i] + r * d
        d2 = rng.standard_normal(3)
        d2 /= np.linalg.norm(d2)
        p2 = qn[i] + r * d2
        p2 = np.nextafter(p2, p2 + rng.choice([-1.0, 1.0], 3))
        near += [p1, p2]
    tgt_n = np.concatenate([base, np.array(near)], 0)[rng.permutation(300 + 128)]
    save("closest_nearties3", src=qn, tgt=tgt_n, idx=ref_match_idx(icp3["Closest"], qn, tgt_n))

    # column-major twin (icp.py:18-21): set is [dim,M], pt is [dim,1]
    tc = tgt2.T.copy()
    idx_col = np.array([icpcol["closest"](tc, src2[i][:, None]) for i in range(src2.shape[0])], dtype=np.int32)
    save("closest_col2", src=src2.T.copy(), tgt=tc, idx=idx_col)

    # ---------------- 2. kabsch ----------------
    rng = np.random.default_rng(21)
    s = rng.standard_normal((500, 3)) * np.array([1.0, 2.0, 0.5]) + np.array([3.0, -1.0, 0.2])
    Rt = orc.rot_matrix(0.4, -0.2, 1.1)
    t = s @ Rt + np.array([0.3, 0.1, -0.7]) + 0.01 * rng.standard_normal((500, 3))
    R, tv = icp3["FindTransform"](s, t)
    save("kabsch_3d", src=s, nn=t, R=R, t=tv, dist=np.array(icp3["Distance"](s, t)))
    s2 = rng.standard_normal((40, 2)) + np.array([1.0, 2.0])
    t2 = s2 @ orc.rot2(0.8) + np.array([2.0, 3.5]) + 0.01 * rng.standard_normal((40, 2))
    T2 = icprow["FindTransform"](s2, t2)
    save("kabsch_2d_T", src=s2, nn=t2, T=T2, dist=np.array(icprow["Distance"](s2, t2)))
    # a reflected correspondence set: the reference applies no det fix (Q2)
    tm = s @ np.diag([1.0, 1.0, -1.0]) + 0.01 * rng.standard_normal((500, 3))
    Rm, tvm = icp3["FindTransform"](s, tm)
    save("kabsch_3d_reflect", src=s, nn=tm, R=Rm, t=tvm)

    # ---------------- 3. ICP ----------------
    src, tgt = orc.make_rigid_cloud(1000, seed=0)
    T = icp3["ICP"](src, tgt, 0.10, 50)
    # per-iteration history with the reference's own functions
    ss = src.copy()
    err = 100000000
    idx_h, err_h = [], []
    for i in range(50):
        e = err
        idx = ref_match_idx(icp3["Closest"], ss, tgt)
        nn = tgt[idx]
        R, tv = icp3["FindTransform"](src, nn)
        ss = np.dot(src, R) + tv
        err = icp3["Distance"](nn, ss)
        idx_h.append(idx)
        err_h.append(err)
        if (e - err) < 0.10:
            break
    Th = np.identity(4)
    Th[:3, :3] = R
    Th[3, :3] = tv
    assert np.array_equal(T, Th)
    save("icp_c1", seed=np.array(0), n=np.array(1000), tol=np.array(0.10), run=np.array(50), T=T,
         idx_hist=np.array(idx_h), err_hist=np.array(err_h))

    # a harder start (bigger motion, partial overlap) with a 30-iteration forced run
    rng = np.random.default_rng(31)
    tgt = rng.standard_normal((700, 3))
    src = (tgt[:500] - np.array([0.4, -0.3, 0.2])) @ orc.rot_matrix(0.25, -0.2, 0.3).T
    T = icp3["ICP"](src, tgt, -np.inf, 30)
    save("icp_forced30", src=src, tgt=tgt, T=T)

    # 2-D demo shape (icp_row.py: 20 target samples of sin, random source samples)
    rng = np.random.default_rng(32)
    x = np.linspace(0, 2 * np.pi, 20)
    tgt2 = np.array([x, np.sin(x)]).T
    xs = np.sort(rng.random(20)) * 2 * np.pi
    src2 = np.array([xs, np.sin(xs)]).T @ orc.rot2(0.8) + np.array([2.0, 3.5])
    ss = src2.copy()
    errs = []
    for i in range(20):      # icp_row.py:116-127 without the plotting / prompt
        nn = icprow["Match"](ss, tgt2)
        T2 = icprow["FindTransform"](src2, nn)
        ss = np.dot(src2, T2[0:2, 0:2]) + T2[2, 0:2]
        errs.append(icprow["Distance"](nn, ss))
    save("icp_row_demo", src=src2, tgt=tgt2, T=T2, err_hist=np.array(errs))

    # ---------------- 4. NICP initialise / deform ----------------
    for name, (nx, ny) in {"nicp_init_60": (6, 10), "nicp_init_200": (10, 20)}.items():
        v, edges, tg = orc.make_grid_mesh(nx, ny, seed=41)
        P, w = n3["Initialise"](v, tg)
        n = v.shape[0]
        assert ((w != 0).sum(0) == 4).all()
        # KNN probes
        kidx, kdist = zip(*[n3["KNN"](v, v[j], 6) for j in range(0, n, 7)])
        rng = np.random.default_rng(42)
        Pr = P + 0.05 * rng.standard_normal(P.shape)
        save(name, v=v, edges=edges, tgt=tg, P0=P, w=w, knn_idx=np.array(kidx), knn_dist=np.array(kdist),
             deform_id=n3["Deform"](P, w, v), P_rand=Pr, deform_rand=n3["Deform"](Pr, w, v),
             R=n3["RotMatrix"](0.3, -0.4, 0.5), dR=np.array(n3["D_RotMatrix"](0.3, -0.4, 0.5)))

    # ---------------- 5. NICP step: residual / Jacobian / leastsq ----------------
    def nicp_fixture(name, nx, ny, outer):
        v, edges, tg = orc.make_grid_mesh(nx, ny, seed=51)
        edges = edges.astype(np.int64)
        P0, w = n3["Initialise"](v, tg)
        rng = np.random.default_rng(52)
        Pr = P0 + 0.03 * rng.standard_normal(P0.shape)
        nn = n3["Match"](n3["Deform"](P0, w, v), tg)
        n3["c_fit"], n3["c_rigid"], n3["c_smooth"] = 0.1, 1000.0, 100.0
        r = n3["MinFunc"](Pr, w, v, edges, nn)
        J = n3["D_MinFunc"](Pr, w, v, edges, nn)
        pv = rng.standard_normal(J.shape[1])
        pu = rng.standard_normal(J.shape[0])
        lsq = optimize.leastsq(n3["MinFunc"], P0, (w, v, edges, nn), n3["D_MinFunc"], full_output=1)
        out = dict(v=v, edges=edges.astype(np.int32), tgt=tg, w=w, P0=P0, P_rand=Pr, nn=nn, r=r,
                   pv=pv, pu=pu, Jv=J @ pv, JTu=J.T @ pu,
                   lsq_P=lsq[0], lsq_err=np.array((lsq[2]["fvec"] ** 2).sum()), lsq_nfev=np.array(lsq[2]["nfev"]),
                   lsq_ier=np.array(lsq[4]), lsq_deformed=n3["Deform"](lsq[0], w, v))
        if J.size <= 400000:
            out["J"] = J
        if outer:
            n3["c_fit"], n3["c_rigid"], n3["c_smooth"] = 0.1, 1000.0, 100.0
            dd, Pf = n3["NICP"](P0.copy(), w, v, edges, tg, 0.01, outer)
            out.update(nicp_dd=dd, nicp_P=Pf, nicp_run=np.array(outer),
                       nicp_c=np.array([n3["c_fit"], n3["c_rigid"], n3["c_smooth"]]))
            n3["c_fit"], n3["c_rigid"], n3["c_smooth"] = 0.1, 1000.0, 100.0
        save(name, **out)

    nicp_fixture("nicp_step_24", 4, 6, outer=4)
    nicp_fixture("nicp_step_60", 6, 10, outer=0)

    # 2-D chain (nicp_row.py): 20 source samples, chain edges; module-global `source` (Q7)
    rng = np.random.default_rng(61)
    x = np.linspace(0, 2 * np.pi, 50)
    tgt2 = np.array([x, np.sin(x)]).T
    xs = np.sort(rng.random(20)) * 2 * np.pi
    d = np.array([xs, np.sin(xs)]).T
    for i in range(1, 20):
        d[i, 1] *= (0.5 + float(i) / 20.0)
    src2 = d @ orc.rot2(0.83) + np.array([2.0, 3.5])
    edges2 = np.array([[i, i + 1] for i in range(19)])
    nrow["source"] = src2
    P0, w2 = nrow["Initialise"](src2, tgt2)
    nn2 = nrow["Match"](nrow["Deform"](P0, w2, src2), tgt2)
    Pr = P0 + 0.03 * rng.standard_normal(P0.shape)
    r2 = nrow["MinFunc"](Pr, w2, src2, edges2, nn2)
    J2 = nrow["D_MinFunc"](Pr, w2, src2, edges2, nn2)
    lsq = optimize.leastsq(nrow["MinFunc"], P0, (w2, src2, edges2, nn2), nrow["D_MinFunc"], full_output=1)
    save("nicp_row_20", v=src2, edges=edges2.astype(np.int32), tgt=tgt2, w=w2, P0=P0, P_rand=Pr, nn=nn2, r=r2, J=J2,
         deform_rand=nrow["Deform"](Pr, w2, src2), lsq_P=lsq[0], lsq_err=np.array((lsq[2]["fvec"] ** 2).sum()),
         lsq_deformed=nrow["Deform"](lsq[0], w2, src2))

    # rigid-only LM (nicp_row_rt.py)
    rng = np.random.default_rng(71)
    xs = np.sort(rng.random(20)) * 2 * np.pi
    src3 = np.array([xs, np.sin(xs)]).T @ orc.rot2(0.875) + np.array([2.0, 3.5])
    p = np.array([0.1, -0.2, 0.3])
    nn3 = nrt["Match"](nrt["Deform"](src3, 0.0, 0.0, 0.0), tgt2)
    p0 = np.zeros(3)
    hist = []
    for it in range(6):      # nicp_row_rt.py:121-126 without plotting / prompt
        dd = nrt["Deform"](src3, p0[0], p0[1], p0[2])
        nnk = nrt["Match"](dd, tgt2)
        l = optimize.leastsq(nrt["Disp"], p0, (src3, nnk), nrt["D_Disp"], full_output=1)
        p0 = l[0]
        hist.append(np.append(p0, (l[2]["fvec"] ** 2).sum()))
    save("nicp_row_rt", src=src3, tgt=tgt2, p=p, nn=nn3, disp=nrt["Disp"](p, src3, nn3), d_disp=nrt["D_Disp"](p, src3, nn3),
         deform=nrt["Deform"](src3, p[0], p[1], p[2]), lm_hist=np.array(hist))


if __name__ == "__main__":
    main()
