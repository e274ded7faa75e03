"""Offline experiment (CPU, numpy, small meshes): how many preconditioned-CG iterations the NICP normal
equations need with richer coarse spaces than the 12 global modes nicp.hip deflates today.  Uses the
oracle's dense Jacobian, so it lives with the tests (the only place allowed to import oracle/); a study tool, not collected by pytest:
    python tests/study_coarse_space.py 32 1e-6
Results recorded in DESIGN.md section 7."""
import sys, time
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "oracle"))
import numpy as np
import icp_oracle as orc

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 24
v, edges, tgt = orc.make_grid_mesh(nx, nx, seed=5)
n = len(v)
nbr, wt = orc.knn_weights(v)
C = (0.1, 1000.0, 100.0)
P = np.append(np.tile(np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0.0]), n), np.zeros(6))
nn = tgt[orc.match_idx(v, tgt)[0]]
t0 = time.time()
J = orc.d_min_func(P, nbr, wt, v, edges, nn, C, compat=False)
r = orc.min_func(P, nbr, wt, v, edges, nn, C, compat=False)
print("n", n, "J", J.shape, "built in %.1fs" % (time.time() - t0))
A0 = J.T @ J
d2 = np.diag(A0).copy(); d2[d2 <= 0] = 1.0
lam = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
A = A0 + lam * np.diag(d2)
g = -(J.T @ r)
S = 12
# block Jacobi
Binv = np.zeros_like(A)
for j in range(n):
    sl = slice(j * S, j * S + S)
    Binv[sl, sl] = np.linalg.inv(A[sl, sl])
tl = slice(n * S, n * S + 6)
Binv[tl, tl] = np.diag(1.0 / np.diag(A)[tl])

def zmodes(Pj, vj):
    z = np.zeros((6, 12))
    rows = [Pj[0:3], Pj[3:6], Pj[6:9], vj + Pj[9:12]]
    for rr, x in enumerate(rows):
        o = rr * 3
        z[0, o] = -x[1]; z[0, o + 1] = x[0]
        z[1, o + 1] = -x[2]; z[1, o + 2] = x[1]
        z[2, o] = x[2]; z[2, o + 2] = -x[0]
    z[3, 9] = z[4, 10] = z[5, 11] = 1.0
    return z

def coarse(funcs):
    cols = []
    c = v - v.mean(0)
    c = c / np.abs(c).max()
    for f in funcs:
        phi = f(c)
        for k in range(6):
            col = np.zeros(n * S + 6)
            for j in range(n):
                col[j * S:(j + 1) * S] = phi[j] * zmodes(P[j * S:(j + 1) * S], v[j])[k]
            cols.append(col)
    for k in range(6):
        col = np.zeros(n * S + 6); col[n * S + k] = 1.0; cols.append(col)
    return np.array(cols).T

def pcg(Z, tol=1e-6, maxit=20000):
    if Z is not None:
        E = Z.T @ (A @ Z)
        Einv = np.linalg.pinv(E, rcond=1e-13)
    def M(rv):
        z = Binv @ rv
        if Z is not None:
            z = z + Z @ (Einv @ (Z.T @ rv))
        return z
    x = np.zeros_like(g); rv = g.copy(); z = M(rv); p = z.copy(); rz = rv @ z; rr0 = rv @ rv
    for it in range(1, maxit + 1):
        Ap = A @ p
        al = rz / (p @ Ap)
        x += al * p; rv -= al * Ap
        if rv @ rv <= tol * tol * rr0:
            return it
        z = M(rv); rzn = rv @ z; p = z + (rzn / rz) * p; rz = rzn
    return maxit

one = lambda c: np.ones(len(c))
lin = [lambda c: c[:, 0], lambda c: c[:, 1], lambda c: c[:, 2]]
quad = [lambda c: c[:, 0] ** 2, lambda c: c[:, 1] ** 2, lambda c: c[:, 2] ** 2, lambda c: c[:, 0] * c[:, 1],
        lambda c: c[:, 1] * c[:, 2], lambda c: c[:, 0] * c[:, 2]]
print("block-Jacobi only      :", pcg(None))
print("+ 12 global (current)   :", pcg(coarse([one])))
print("+ affine  (30 columns)  :", pcg(coarse([one] + lin)))
print("+ quadratic (66 columns):", pcg(coarse([one] + lin + quad)))

def patches(k):
    # k x k x k boxes over the bounding box -> indicator functions of the non-empty ones
    c = v - v.min(0)
    c = c / (c.max(0) + 1e-12)
    ids = np.minimum((c * k).astype(int), k - 1)
    key = ids[:, 0] * k * k + ids[:, 1] * k + ids[:, 2]
    fs = []
    for u in np.unique(key):
        m = (key == u).astype(float)
        fs.append(lambda cc, m=m: m)
    return fs
for k in (2, 3, 4, 6):
    fs = patches(k)
    print("patches k=%d (%d patches, %d columns):" % (k, len(fs), 6 * len(fs) + 6), pcg(coarse(fs)))
