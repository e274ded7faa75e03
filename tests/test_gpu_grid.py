"""The exact grid search must return the SAME indices and squared distances as the
brute-force scan (and the C oracle) on every kind of input: that is what makes it a legal
replacement for the reference's O(N*M) Closest loop."""
import numpy as np
import pytest

import c_oracle
import icp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rigid():
    import torch
    assert torch.cuda.is_available()
    from icp_amd import rigid
    return rigid


def both(rigid, src, tgt, **kw):
    gi, gd = rigid.closest_indices(src, tgt, want_d2=True, force_grid=True, **kw)
    bi, bd = rigid.closest_indices(src, tgt, want_d2=True, force_brute=True, **kw)
    return gi.cpu().numpy(), gd.cpu().numpy(), bi.cpu().numpy(), bd.cpu().numpy()


def clouds():
    rng = np.random.default_rng(21)
    out = {}
    out["normal3"] = (rng.standard_normal((30000, 3)), rng.standard_normal((40000, 3)))
    out["uniform2"] = (rng.uniform(-1, 1, (20000, 2)), rng.uniform(-1, 1, (9000, 2)))
    # clustered target, queries partly far outside the target's bounding box
    cl = np.concatenate([rng.standard_normal((20000, 3)) * 0.01 + c for c in rng.uniform(-5, 5, (8, 3))])
    out["clustered_outside"] = (np.concatenate([rng.uniform(-20, 20, (5000, 3)), cl[:5000] + 1e-3]), cl)
    # lattice with duplicates: masses of exact ties, lowest index must win
    g = np.stack(np.meshgrid(np.arange(20.), np.arange(20.), np.arange(20.), indexing="ij"), -1).reshape(-1, 3)
    tg = np.concatenate([g[rng.permutation(len(g))], g[rng.permutation(len(g))]])
    out["lattice_ties"] = (np.concatenate([g[:3000] + 0.5, g[:3000] + [0.5, 0, 0], g[:2000], rng.uniform(-3, 23, (3000, 3))]), tg)
    # strongly anisotropic (a thin sheet) and far from the origin (bounds carry absolute slack)
    sheet = rng.uniform(0, 1, (30000, 3)) * [100.0, 1.0, 1e-3]
    out["sheet"] = (sheet[:8000] + rng.standard_normal((8000, 3)) * [1.0, 0.01, 1e-4], sheet)
    far = rng.standard_normal((20000, 3)) + 1e9
    out["far_origin"] = (far[:6000] + rng.standard_normal((6000, 3)) * 0.1, far)
    # all targets identical / collinear: degenerate boxes
    out["identical"] = (rng.standard_normal((500, 3)), np.tile([[1.0, 2.0, 3.0]], (5000, 1)))
    line = np.zeros((5000, 3))
    line[:, 0] = np.linspace(0, 1, 5000)
    out["collinear"] = (rng.standard_normal((2000, 3)), line)
    out["tiny_target"] = (rng.standard_normal((3000, 3)), rng.standard_normal((3, 3)))
    # far outliers on both sides: the grid spans the bulk only (robust extents), the outliers live in the unbounded
    # border columns; queries among them, beyond them and in the bulk
    bulk = rng.standard_normal((60000, 3))
    bulk[rng.integers(0, 60000, 40)] *= 1e6
    q = rng.standard_normal((20000, 3))
    q[:200] *= 1e6
    q[200:400] = bulk[rng.integers(0, 60000, 200)] * (1 + 1e-9)
    out["far_outliers"] = (q, bulk)
    # decisions far below float resolution (the per-lane pass rejects candidates on an fp32 distance in cell units and must
    # pass on to the fp64 records whatever that cannot separate): every lattice point twice, 1e-10 apart, queries near the
    # midpoints of cells and of edges, so that 2 - 16 targets are within 1e-9 (relative) of the minimum
    g = np.stack(np.meshgrid(np.arange(24.), np.arange(24.), np.arange(24.), indexing="ij"), -1).reshape(-1, 3) * 0.37 + 100.0
    tw = np.concatenate([g, g + rng.standard_normal(g.shape) * 1e-10])[rng.permutation(2 * len(g))]
    qm = np.concatenate([g[:5000] + 0.185, g[:5000] + [0.185, 0, 0], g[:3000]]) + rng.standard_normal((13000, 3)) * 1e-10
    out["below_float_resolution"] = (qm, tw)
    # heavy tails: a Cauchy cloud has no meaningful bounding box at all
    out["cauchy2"] = (rng.standard_cauchy((15000, 2)), rng.standard_cauchy((50000, 2)))
    return out


@pytest.mark.parametrize("name", list(clouds().keys()))
def test_grid_equals_bruteforce(rigid, name):
    src, tgt = clouds()[name]
    gi, gd, bi, bd = both(rigid, src, tgt)
    assert np.array_equal(gi, bi), "%s: %d index mismatches" % (name, (gi != bi).sum())
    assert np.array_equal(gd, bd)
    oi, od = c_oracle.match_idx(src, tgt)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)


def test_grid_compare_sqrt(rigid):
    rng = np.random.default_rng(5)
    tgt = np.concatenate([rng.standard_normal((8000, 3)), [[1.0, 2.0 ** -26, 0.0], [1.0, 0.0, 0.0]]])
    src = np.concatenate([rng.standard_normal((4000, 3)) * 1e-9, rng.standard_normal((4000, 3))])
    gi, gd, bi, bd = both(rigid, src, tgt, compare_sqrt=True)
    oi, _ = c_oracle.match_idx(src, tgt, compare_sqrt=True)
    assert np.array_equal(gi, oi) and np.array_equal(bi, oi)


def test_grid_nonfinite_falls_back(rigid):
    rng = np.random.default_rng(6)
    tgt = rng.standard_normal((6000, 3))
    tgt[4321, 2] = np.nan
    src = rng.standard_normal((3000, 3))
    gi, gd, bi, bd = both(rigid, src, tgt)                 # builder refuses -> brute force -> first NaN wins
    assert np.array_equal(gi, np.full(3000, 4321)) and np.array_equal(bi, gi)
    tgt = rng.standard_normal((6000, 3))
    src[7] = np.nan                                        # NaN QUERY: every d2 is NaN -> index 0
    src[8, 0] = np.inf
    gi, gd, bi, bd = both(rigid, src, tgt)
    assert np.array_equal(gi, bi) and gi[7] == 0


def test_grid_inside_icp(rigid):
    src, tgt = orc.make_rigid_cloud(50000, seed=12)
    a = rigid.rigid_icp(src, tgt, 0.10, 50, want_idx=True, force_grid=True)
    b = rigid.rigid_icp(src, tgt, 0.10, 50, want_idx=True, force_brute=True)
    assert int(a["iters"].cpu()[0]) == int(b["iters"].cpu()[0])
    assert np.array_equal(a["idx"].cpu().numpy(), b["idx"].cpu().numpy())
    # identical correspondences; the sums run over the Morton-ordered source in the grid path
    np.testing.assert_allclose(a["T"].cpu().numpy(), b["T"].cpu().numpy(), rtol=1e-12, atol=1e-13)


def test_target_handle_and_morton_sort(rigid):
    import ctypes
    import torch
    from icp_amd._lib import get_context, ptr
    ctx = get_context()
    rng = np.random.default_rng(31)
    tgt = ctx.to_dev(rng.standard_normal((50000, 3)))
    src = ctx.to_dev(rng.standard_normal((40000, 3)) * 1.2)
    h = ctypes.c_void_p()
    ctx.call("icp_target_create", ptr(tgt), tgt.shape[0], 3, ctypes.byref(h))
    idx = ctx.empty((src.shape[0],), torch.int32)
    d2 = ctx.empty((src.shape[0],))
    for _ in range(2):
        ctx.call("icp_closest_target", h, ptr(src), src.shape[0], 0, ptr(idx), ptr(d2))
    oi, od = c_oracle.match_idx(src.cpu().numpy(), tgt.cpu().numpy())
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(d2.cpu().numpy(), od)
    ctx.call("icp_closest_target", h, ptr(src), 100, 1, ptr(idx), ptr(None))       # compare-sqrt through the handle
    oi2, _ = c_oracle.match_idx(src.cpu().numpy()[:100], tgt.cpu().numpy(), compare_sqrt=True)
    assert np.array_equal(idx.cpu().numpy()[:100], oi2)
    ctx.lib.icp_target_destroy(ctx.handle, h)
    out = torch.empty_like(src)
    order = ctx.empty((src.shape[0],), torch.int32)
    ctx.call("icp_morton_sort", ptr(src), src.shape[0], 3, ptr(out), ptr(order))
    o = order.cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(src.shape[0]))                    # a permutation
    assert np.array_equal(out.cpu().numpy(), src.cpu().numpy()[o])


def test_sharded_ops_single_process(rigid):
    """icp_amd.dist with the real GpuOps (no collective): same answer as the fused loop / oracle."""
    from icp_amd.dist import Comm, GpuOps, sharded_icp
    src, tgt = orc.make_rigid_cloud(40000, seed=14)
    T, hist = sharded_icp(GpuOps(), Comm(enabled=False), src, tgt, 0.10, 50)
    r = rigid.rigid_icp(src, tgt, 0.10, 50)
    n = int(r["iters"].cpu()[0])
    assert len(hist) == n
    np.testing.assert_allclose(T, r["T"].cpu().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(hist, r["err_hist"].cpu().numpy()[:n], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name", ["normal3", "uniform2", "clustered_outside", "lattice_ties", "below_float_resolution", "far_outliers",
                                  "sheet", "far_origin"])
def test_hint_does_not_change_the_result(rigid, name):
    """ICP_HINT_IDX: whatever indices come in -- the true answer, random targets, a constant,
    out-of-range garbage -- the result is the same exact argmin (also with ICP_COMPARE_SQRT).  Hinted argmin searches run
    through grid_pool_kernel (the wave works the neighbour cells off together, fp32 pre-filter, fp64 candidates merged by
    LDS atomics): good hints leave it nothing to merge, random ones make every ball large (lists of more than 384 cells:
    several passes) and most records candidates; on the lattice every merge is a tie on d2 that the index decides; the
    outliers and the far origin are beyond what the filter's floats can hold (stored as NaN / filter off)."""
    src, tgt = clouds()[name]
    n, m = len(src), len(tgt)
    rng = np.random.default_rng(5)
    ref_i, ref_d = rigid.closest_indices(src, tgt, want_d2=True, force_brute=True)
    ref_i, ref_d = ref_i.cpu().numpy(), ref_d.cpu().numpy()
    hints = {"truth": ref_i, "random": rng.integers(0, m, n).astype(np.int32), "zero": np.zeros(n, np.int32),
             "garbage": np.where(rng.random(n) < 0.5, -7, m + 12345).astype(np.int32),
             "mixed": np.where(rng.random(n) < 0.5, ref_i, rng.integers(-5, m + 5, n)).astype(np.int32)}
    for k, h in hints.items():
        gi, gd = rigid.closest_indices(src, tgt, want_d2=True, force_grid=True, hint=h)
        assert np.array_equal(gi.cpu().numpy(), ref_i), (name, k)
        assert np.array_equal(gd.cpu().numpy(), ref_d), (name, k)
    si = rigid.closest_indices(src, tgt, force_grid=True, compare_sqrt=True).cpu().numpy()
    sh = rigid.closest_indices(src, tgt, force_grid=True, compare_sqrt=True, hint=hints["random"]).cpu().numpy()
    assert np.array_equal(si, sh)


@pytest.mark.parametrize("dim,compare_sqrt", [(2, False), (3, True), (2, True)])
def test_grid_inside_icp_2d_and_sqrt(rigid, dim, compare_sqrt):
    """The fused loop with the grid (sorted-position bookkeeping, hints) against the full-scan loop:
    2-D clouds and the C++ comparison of rounded roots (second, `<= threshold` pass)."""
    rng = np.random.default_rng(40 + dim)
    tgt = rng.standard_normal((60000, dim))
    c, s = np.cos(0.2), np.sin(0.2)
    R = np.eye(dim)
    R[:2, :2] = [[c, s], [-s, c]]
    src = ((tgt - 0.05) @ R)[rng.permutation(len(tgt))[:45000]]
    a = rigid.rigid_icp(src, tgt, 1e-3, 25, want_idx=True, force_grid=True, compare_sqrt=compare_sqrt)
    b = rigid.rigid_icp(src, tgt, 1e-3, 25, want_idx=True, force_brute=True, compare_sqrt=compare_sqrt)
    assert int(a["iters"].cpu()[0]) == int(b["iters"].cpu()[0])
    assert np.array_equal(a["idx"].cpu().numpy(), b["idx"].cpu().numpy())
    np.testing.assert_allclose(a["T"].cpu().numpy(), b["T"].cpu().numpy(), rtol=1e-12, atol=1e-13)
    n = int(a["iters"].cpu()[0])
    np.testing.assert_allclose(a["err_hist"].cpu().numpy()[:n], b["err_hist"].cpu().numpy()[:n], rtol=1e-11)


def test_deferred_counters_survive_early_stop_and_resizing(rigid):
    """The deferred-query list has two counters used alternately, each cleared by the deferred pass of the
    search before.  Searches that return at once (run stopped by its tolerance, an odd and an even number
    of them) and a larger cloud that reallocates the list in between must leave every later search exact."""
    rng = np.random.default_rng(5)
    cl = np.concatenate([rng.standard_normal((6000, 3)) * 0.01 + c for c in rng.uniform(-5, 5, (6, 3))])
    far = np.concatenate([rng.uniform(-20, 20, (4000, 3)), cl[:3000] + 1e-3])   # many deferred queries
    oi, od = c_oracle.match_idx(far, cl)

    def check():
        gi, gd = rigid.closest_indices(far, cl, want_d2=True, force_grid=True)
        assert np.array_equal(gi.cpu().numpy(), oi) and np.array_equal(gd.cpu().numpy(), od)

    check()
    src, tgt = orc.make_rigid_cloud(20000, seed=3)
    for run in (40, 41):     # both stop after a few iterations: the rest of the enqueued searches are no-ops
        out = rigid.rigid_icp(src, tgt, 0.10, run, force_grid=True)
        assert int(out["iters"].cpu()[0]) < run - 2
        check()
        check()
    big_s, big_t = orc.make_rigid_cloud(150000, seed=4)    # the list grows: counters start afresh
    a = rigid.rigid_icp(big_s, big_t, 0.10, 30, want_idx=True, force_grid=True)
    b = rigid.rigid_icp(big_s, big_t, 0.10, 30, want_idx=True, force_brute=True)
    assert np.array_equal(a["idx"].cpu().numpy(), b["idx"].cpu().numpy())
    check()


def test_one_far_outlier_does_not_turn_the_grid_into_a_scan(rigid):
    """A single target point at 1e7 used to stretch the grid over the bounding box: the whole cloud fell into one cell
    and every query scanned every target (seconds).  With the extents taken from quantiles the search stays a grid
    search: same answer as the scan, and at least 20x faster than it."""
    import time
    import torch
    rng = np.random.default_rng(5)
    tgt = rng.standard_normal((300000, 3))
    tgt[12345] = [1e7, -1e7, 1e7]
    src = rng.standard_normal((300000, 3))
    rigid.closest_indices(src, tgt, force_grid=True)      # (a first call of this size allocates the library's scratch: milliseconds)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gi = rigid.closest_indices(src, tgt, force_grid=True)
    torch.cuda.synchronize()
    t_grid = time.perf_counter() - t0
    t0 = time.perf_counter()
    bi = rigid.closest_indices(src, tgt, force_brute=True)
    torch.cuda.synchronize()
    t_brute = time.perf_counter() - t0
    assert torch.equal(gi, bi)
    assert t_grid * 20 < t_brute, (t_grid, t_brute)


@pytest.mark.parametrize("dim,use_tol,n", [(3, True, 60000), (3, False, 45001), (2, True, 50000)])
def test_fused_loop_equals_unfused_loop(rigid, dim, use_tol, n):
    """From its second iteration on the loop never writes the moved source: the search kernels apply R, t to the
    original rows themselves and add up the distances to their hints (= the previous iteration's error terms).
    ICP_NO_FUSE keeps the moved source as an array and `apply` / the error as kernels of their own: the same
    transform and correspondences bit for bit, the same number of iterations, errors equal up to summation order.
    (n = 45001: a last block with one query.)"""
    import torch
    rng = np.random.default_rng(50 + dim)
    tgt = rng.standard_normal((n + 7000, dim))
    c, s = np.cos(0.15), np.sin(0.15)
    R = np.eye(dim)
    R[:2, :2] = [[c, s], [-s, c]]
    src = ((tgt - 0.03) @ R)[rng.permutation(len(tgt))[:n]]
    a = rigid.rigid_icp(src, tgt, 1e-3, 25, use_tol=use_tol, want_idx=True, force_grid=True)
    b = rigid.rigid_icp(src, tgt, 1e-3, 25, use_tol=use_tol, want_idx=True, force_grid=True, no_fuse=True)
    k = int(a["iters"].cpu()[0])
    assert k == int(b["iters"].cpu()[0]) and (use_tol or k == 25)
    assert torch.equal(a["T"], b["T"]) and torch.equal(a["idx"], b["idx"])
    np.testing.assert_allclose(a["err_hist"].cpu().numpy()[:k], b["err_hist"].cpu().numpy()[:k], rtol=1e-13)
    a2 = rigid.rigid_icp(src, tgt, 1e-3, 25, use_tol=use_tol, want_idx=True, force_grid=True)
    assert torch.equal(a["err_hist"][:k], a2["err_hist"][:k])      # and reproducible bit for bit


@pytest.mark.parametrize("dim,use_tol,n,kind", [(3, False, 60000, "normal"), (3, True, 45001, "normal"), (2, False, 50000, "normal"),
                                                  (3, False, 20000, "lattice"), (3, False, 300000, "rim")])
def test_coherent_loop_equals_plain_loop(rigid, dim, use_tol, n, kind):
    """ICP_COHERENT: every search also leaves, per source row, the second-best squared distance (a lower bound for every
    target but the match); in later iterations a row that has moved by delta since then keeps its match without a search
    when the match is still nearer than that bound minus delta (triangle inequality, slack for every rounding), and only
    the other rows are searched.  The proof says the full scan would return the same index, so the whole run -- transform,
    correspondences, error history, iteration count -- must equal the plain fused loop's bit for bit: on a Gaussian cloud,
    in 2-D, with the stop test live, on a lattice (masses of exact ties: a tie can never be kept) and on a cloud with a
    sparse rim and rows that are NaN / inf / far away (never kept, or kept for long).  40 iterations, so that rows are
    kept, drift, fail the test and are searched again several times over."""
    import torch
    rng = np.random.default_rng(60 + dim + n)
    if kind == "lattice":
        tgt = rng.integers(0, 24, (n + 3000, dim)).astype(np.float64) * 0.25
    else:
        tgt = rng.standard_normal((n + 7000, dim))
    c, s = np.cos(0.15), np.sin(0.15)
    R = np.eye(dim)
    R[:2, :2] = [[c, s], [-s, c]]
    src = ((tgt - 0.03) @ R)[rng.permutation(len(tgt))[:n]]
    if kind == "rim":
        src[:2000] *= 3.0                     # rows far outside the target: deferred pass, large bounds
        src[5000:5050] = np.nan
        src[6000:6010] = np.inf
        src[7000] = 1e12
    run = 40
    a = rigid.rigid_icp(src, tgt, 1e-7, run, use_tol=use_tol, want_idx=True, force_grid=True)
    b = rigid.rigid_icp(src, tgt, 1e-7, run, use_tol=use_tol, want_idx=True, force_grid=True, coherent=True)
    k = int(a["iters"].cpu()[0])
    assert k == int(b["iters"].cpu()[0]) and (use_tol or k == run)
    bits = lambda t: t.contiguous().view(torch.int64)    # (NaN rows make T NaN: compare bit patterns)
    assert torch.equal(bits(a["T"]), bits(b["T"])) and torch.equal(a["idx"], b["idx"])
    assert torch.equal(bits(a["err_hist"][:k]), bits(b["err_hist"][:k]))
    b2 = rigid.rigid_icp(src, tgt, 1e-7, run, use_tol=use_tol, want_idx=True, force_grid=True, coherent=True)   # a second call: no stale state
    assert torch.equal(bits(b["T"]), bits(b2["T"])) and torch.equal(b["idx"], b2["idx"])


def test_large_target_and_sorted_source_are_built_concurrently(rigid):
    """icp_rigid_run builds the Morton order of the source on a side stream while the target's grid is built on the
    main one.  Above 4 M target points the target is ordered by the same radix sort as the source: the two sorts
    run at the same time and used to share one temporary buffer (silently corrupting both orders once the buffer
    no longer grew between them -- from the second call on, or at once when both clouds have the same size).
    Two calls in a row against the full-scan loop."""
    rng = np.random.default_rng(77)
    m = (1 << 22) + 1000
    tgt = rng.standard_normal((m, 3))
    c, s = np.cos(0.1), np.sin(0.1)
    R = np.array([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]])
    src = ((tgt[rng.permutation(m)[:40000]] - 0.02) @ R)
    b = rigid.rigid_icp(src, tgt, -np.inf, 3, want_idx=True, force_brute=True)
    for _ in range(2):
        a = rigid.rigid_icp(src, tgt, -np.inf, 3, want_idx=True)
        assert np.array_equal(a["idx"].cpu().numpy(), b["idx"].cpu().numpy())
        np.testing.assert_allclose(a["T"].cpu().numpy(), b["T"].cpu().numpy(), rtol=1e-12, atol=1e-13)


def test_deferred_pass_descends_into_coarse_cells(rigid):
    """Rim and far-away queries of a dense cloud: their balls reach coarse cells that hold tens of thousands of
    records; the deferred pass replaces such cells by their children (a stack per half-wave in LDS, several levels
    deep here, cells that do not fit on it scanned whole) instead of scanning them.  Exact all the same: indices and
    squared distances against the full scan, 3-D and 2-D, with and without hints, and it must not take the seconds a
    whole-cell scan per query would.  (35 000 of the 40 000 queries leave the per-lane pass here and the deferred list
    holds a quarter of the queries + 4 096: the lanes that find it full keep their queries and climb on by themselves.)"""
    import time
    import torch
    rng = np.random.default_rng(91)
    for dim in (3, 2):
        tgt = np.concatenate([rng.standard_normal((300000, dim)) * 0.05, rng.standard_normal((200, dim)) * 30.0])
        d = rng.standard_normal((40000, dim))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        src = np.concatenate([d[:20000] * rng.uniform(0.2, 0.6, (20000, 1)),        # the rim of the dense cloud
                              d[20000:35000] * rng.uniform(1.0, 100.0, (15000, 1)),  # far outside it
                              rng.standard_normal((5000, dim)) * 0.05])              # inside
        bi, bd = rigid.closest_indices(src, tgt, want_d2=True, force_brute=True)
        gi, gd = rigid.closest_indices(src, tgt, want_d2=True, force_grid=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gi2, gd2 = rigid.closest_indices(src, tgt, want_d2=True, force_grid=True, hint=bi)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.equal(gi, bi) and torch.equal(gd, bd) and torch.equal(gi2, bi) and torch.equal(gd2, bd)
        assert dt < 0.5, dt
