"""GPU parity tests for the rigid-ICP path: HIP (through the C-ABI) vs the oracle.
Indices must be bit-exact; transforms within 1e-6 relative (north_star)."""
import os

import numpy as np
import pytest

import icp_oracle as orc
import c_oracle
from conftest import golden

pytestmark = pytest.mark.gpu

RTOL = 1e-6


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from icp_amd import rigid, icp_3d, icp_row, icp
    from icp_amd._lib import get_context
    return dict(rigid=rigid, icp_3d=icp_3d, icp_row=icp_row, icp=icp, ctx=get_context(), torch=torch)


def gpu_idx(api, src, tgt, **kw):
    out = api["rigid"].closest_indices(src, tgt, **kw)
    if isinstance(out, tuple):
        return out[0].cpu().numpy(), out[1].cpu().numpy()
    return out.cpu().numpy()


@pytest.mark.parametrize("name", ["closest_1k_normal3", "closest_20_50_dim2", "closest_ties3", "closest_nearties3"])
def test_match_golden(api, name):
    g = golden(name)
    idx, d2 = gpu_idx(api, g["src"], g["tgt"], want_d2=True)
    assert np.array_equal(idx, g["idx"])
    oi, od2 = orc.match_idx(g["src"], g["tgt"])
    assert np.array_equal(d2, od2)          # the squared distances are bit-exact too
    nn = api["icp_3d"].Match(g["src"], g["tgt"])
    assert np.array_equal(nn, g["tgt"][g["idx"]])


def test_match_golden_4k_50k(api):
    g = golden("closest_4k_50k_uniform3")
    rng = np.random.default_rng(int(g["seed"]))
    src = rng.uniform(-1, 1, (4096, 3))
    tgt = rng.uniform(-1, 1, (50000, 3))
    assert np.array_equal(gpu_idx(api, src, tgt), g["idx"])


def test_closest_single_and_col(api):
    g = golden("closest_1k_normal3")
    for i in (0, 17, 999):
        assert api["icp_3d"].Closest(g["tgt"], g["src"][i]) == int(g["idx"][i])
    g = golden("closest_col2")
    for i in range(g["src"].shape[1]):
        assert api["icp"].closest(g["tgt"], g["src"][:, i:i + 1]) == int(g["idx"][i])


# (n, m, dim): exercises the lanes kernel (small n), the tile kernel with Q=1/2/4, target
# splits, ragged tails (m not a multiple of the tile / chunk) and m < one tile
@pytest.mark.parametrize("n,m,dim", [(1, 1, 3), (3, 5, 2), (1, 100003, 3), (257, 4099, 3), (1000, 1000, 3),
                                     (20000, 3001, 3), (20000, 5, 2), (70001, 2049, 2),
                                     (300000, 1027, 3), (600000, 515, 3), (1100000, 260, 3)])
def test_match_vs_c_oracle(api, n, m, dim):
    rng = np.random.default_rng(n * 7 + m)
    src = rng.standard_normal((n, dim))
    tgt = rng.standard_normal((m, dim))
    idx, d2 = gpu_idx(api, src, tgt, want_d2=True)
    oi, od = c_oracle.match_idx(src, tgt)
    assert np.array_equal(idx, oi)
    assert np.array_equal(d2, od)


def test_match_duplicates_lowest_index(api):
    rng = np.random.default_rng(5)
    base = rng.integers(-3, 4, (200, 3)).astype(np.float64)     # heavy duplication
    tgt = np.concatenate([base, base, base])[rng.permutation(600)]
    src = rng.integers(-3, 4, (30000, 3)).astype(np.float64) + 0.5
    idx = gpu_idx(api, src, tgt)
    oi, _ = c_oracle.match_idx(src, tgt)
    assert np.array_equal(idx, oi)
    idx1 = gpu_idx(api, src[:100], tgt)                          # lanes kernel
    assert np.array_equal(idx1, oi[:100])


def test_match_nan_inf(api):
    rng = np.random.default_rng(6)
    tgt = rng.standard_normal((3000, 3))
    tgt[1234, 1] = np.nan
    tgt[2000, 0] = np.nan
    src = rng.standard_normal((20000, 3))
    expect = np.full(20000, 1234)                                # np.argmin: first NaN wins
    assert np.array_equal(gpu_idx(api, src, tgt), expect)
    assert np.array_equal(gpu_idx(api, src[:10], tgt), expect[:10])
    assert orc.closest(tgt, src[0]) == 1234
    tgt = np.full((2000, 3), 1e200)                              # every d2 overflows to +inf -> index 0
    assert np.array_equal(gpu_idx(api, src, tgt), np.zeros(20000))
    assert np.array_equal(gpu_idx(api, src[:7], tgt), np.zeros(7))


def test_compare_sqrt_mode(api):
    # d2 differs by one ulp but the rounded roots tie: numpy picks the smaller d2 (index 1),
    # nicp3d.cpp's sqrt compare keeps the earlier index 0 (Q12)
    tgt = np.array([[1.0, 2.0 ** -26, 0.0], [1.0, 0.0, 0.0], [5.0, 5.0, 5.0]])
    q = np.zeros((1, 3))
    assert orc.closest(tgt, q[0]) == 1 and orc.closest(tgt, q[0], compare_sqrt=True) == 0
    assert gpu_idx(api, q, tgt)[0] == 1
    assert gpu_idx(api, q, tgt, compare_sqrt=True)[0] == 0
    rng = np.random.default_rng(8)
    tgt = np.concatenate([rng.standard_normal((5000, 3)), tgt])
    src = np.concatenate([rng.standard_normal((20000, 3)) * 1e-9, rng.standard_normal((100, 3))])
    oi, _ = c_oracle.match_idx(src, tgt, compare_sqrt=True)
    assert np.array_equal(gpu_idx(api, src, tgt, compare_sqrt=True), oi)
    assert np.array_equal(gpu_idx(api, src[-50:], tgt, compare_sqrt=True), oi[-50:])


def test_empty_and_errors(api):
    from icp_amd._lib import IcpHipError
    tgt = np.zeros((4, 3))
    assert gpu_idx(api, np.zeros((0, 3)), tgt).shape == (0,)
    with pytest.raises(ValueError):
        gpu_idx(api, np.zeros((2, 3)), np.zeros((0, 3)))
    with pytest.raises(ValueError):
        gpu_idx(api, np.zeros((2, 3)), np.zeros((4, 2)))
    with pytest.raises(NameError):
        api["icp_3d"].ICP(np.zeros((3, 3)), tgt, 0.1, 0)


def test_float32_and_other_dtype_tensors_are_converted(api):
    """a float32 CUDA tensor -- torch's default dtype -- must not reach the fp64 kernels as a raw
    pointer.  Inputs of any dtype are converted to fp64 on the context's device: same result as the fp64 path on
    the same (float32-representable) values."""
    torch = api["torch"]
    rng = np.random.default_rng(21)
    src64 = rng.standard_normal((30000, 3)).astype(np.float32).astype(np.float64)
    tgt64 = rng.standard_normal((5000, 3)).astype(np.float32).astype(np.float64)
    ref = gpu_idx(api, src64, tgt64)
    s32 = torch.from_numpy(src64.astype(np.float32)).cuda()
    t32 = torch.from_numpy(tgt64.astype(np.float32)).cuda()
    assert s32.dtype == torch.float32
    assert np.array_equal(api["rigid"].closest_indices(s32, t32).cpu().numpy(), ref)
    assert np.array_equal(api["rigid"].closest_indices(torch.from_numpy(src64.astype(np.float32)), t32.half().float()).cpu().numpy(),
                          gpu_idx(api, src64, t32.half().double().cpu().numpy()))
    T64 = api["icp_3d"].ICP(src64[:2000], tgt64, 0.1, 5)
    T32 = api["icp_3d"].ICP(s32[:2000], t32, 0.1, 5)
    assert T32.dtype == torch.float64 and np.array_equal(T32.cpu().numpy(), T64)
    hint = torch.from_numpy(ref.astype(np.int64)).cuda()        # int64 hints are converted to the int32 the ABI takes
    assert np.array_equal(api["rigid"].closest_indices(s32, t32, hint=hint, force_grid=True).cpu().numpy(), ref)


def test_calls_under_two_streams_are_ordered(api):
    """the context's scratch assumes one stream.  icp_rigid_run issued under stream A and,
    without any host synchronisation, icp_closest under stream B share scratch buffers: the context orders B after
    A (event wait in icp_ctx_set_stream), so both return what they return when run alone -- bit for bit, the
    transform included (the Morton ordering of the source is a stable sort, so every sum has a fixed order)."""
    torch = api["torch"]
    rigid = api["rigid"]
    src, tgt = orc.make_rigid_cloud(200000, seed=9)
    rng = np.random.default_rng(10)
    src2, tgt2 = rng.standard_normal((150000, 3)), rng.standard_normal((180000, 3))
    sd, td, s2d, t2d = (torch.from_numpy(a).cuda() for a in (src, tgt, src2, tgt2))
    alone_run = rigid.rigid_icp(sd, td, -np.inf, 6, want_idx=True)
    torch.cuda.synchronize()
    alone_T, alone_idx = alone_run["T"].clone(), alone_run["idx"].clone()
    alone_c = rigid.closest_indices(s2d, t2d, want_d2=True)
    torch.cuda.synchronize()
    alone_c = (alone_c[0].clone(), alone_c[1].clone())
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        with torch.cuda.stream(a):
            r = rigid.rigid_icp(sd, td, -np.inf, 6, want_idx=True)
        with torch.cuda.stream(b):
            c = rigid.closest_indices(s2d, t2d, want_d2=True)
        with torch.cuda.stream(a):
            r2 = rigid.rigid_icp(sd, td, -np.inf, 6, want_idx=True)
        torch.cuda.synchronize()
        for out in (r, r2):
            assert torch.equal(out["idx"], alone_idx) and torch.equal(out["T"], alone_T)
        assert torch.equal(c[0], alone_c[0]) and torch.equal(c[1], alone_c[1])


def test_distance_and_kabsch_golden(api):
    g = golden("kabsch_3d")
    R, t = api["icp_3d"].FindTransform(g["src"], g["nn"])
    np.testing.assert_allclose(R, g["R"], rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(t, g["t"], rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(api["icp_3d"].Distance(g["src"], g["nn"]), float(g["dist"]), rtol=1e-12)
    g = golden("kabsch_2d_T")
    np.testing.assert_allclose(api["icp_row"].FindTransform(g["src"], g["nn"]), g["T"], rtol=RTOL, atol=1e-9)
    g = golden("kabsch_3d_reflect")                                # Q2: no reflection fix
    R, t = api["icp_3d"].FindTransform(g["src"], g["nn"])
    np.testing.assert_allclose(R, g["R"], rtol=RTOL, atol=1e-9)
    assert np.linalg.det(R) < 0


@pytest.mark.parametrize("n,dim", [(10, 3), (4, 2), (100000, 3), (1 << 20, 3)])
def test_kabsch_vs_oracle(api, n, dim):
    rng = np.random.default_rng(n + dim)
    s = rng.standard_normal((n, dim)) + 5.0            # off-centre: exercises the shifted sums
    Rt = orc.rot_matrix(0.3, 0.2, -0.4) if dim == 3 else orc.rot2(0.7)
    t = s @ Rt + rng.standard_normal(dim) + 1e-3 * rng.standard_normal((n, dim))
    R, tv = api["icp_3d"].FindTransform(s, t)
    Ro, to = c_oracle.find_transform(s, t)
    np.testing.assert_allclose(R, Ro, rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(tv, to, rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(api["icp_3d"].Distance(s, t), orc.distance(s, t), rtol=1e-10)


def test_kabsch_rank_deficient(api):
    """3 points in 3-D: the centred cloud is planar, W has rank 2 and U VT is not unique
    (LAPACK may return a reflection, the Jacobi path completes a right-handed frame;
    SURVEY.md section 7 'SVD degeneracy', parity unpinned).  What IS determined -- the moved
    source points -- must still agree."""
    rng = np.random.default_rng(6)
    s = rng.standard_normal((3, 3)) + 5.0
    t = s @ orc.rot_matrix(0.3, 0.2, -0.4) + rng.standard_normal(3)
    R, tv = api["icp_3d"].FindTransform(s, t)
    Ro, to = c_oracle.find_transform(s, t)
    np.testing.assert_allclose(s @ R + tv, s @ Ro + to, rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(R @ R.T, np.identity(3), atol=1e-12)


def test_icp_c1_golden(api):
    g = golden("icp_c1")
    src, tgt = orc.make_rigid_cloud(int(g["n"]), seed=int(g["seed"]))
    r = api["rigid"].rigid_icp(src, tgt, float(g["tol"]), int(g["run"]), want_idx=True)
    iters = int(r["iters"].cpu()[0])
    assert iters == g["idx_hist"].shape[0]
    np.testing.assert_allclose(r["err_hist"].cpu().numpy()[:iters], g["err_hist"], rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(r["T"].cpu().numpy(), g["T"], rtol=RTOL, atol=1e-9)
    assert np.array_equal(r["idx"].cpu().numpy(), g["idx_hist"][-1])
    T = api["icp_3d"].ICP(src, tgt, float(g["tol"]), int(g["run"]))
    np.testing.assert_allclose(T, g["T"], rtol=RTOL, atol=1e-9)


def test_icp_forced_and_row_golden(api):
    g = golden("icp_forced30")
    r = api["rigid"].rigid_icp(g["src"], g["tgt"], -np.inf, 30)
    assert int(r["iters"].cpu()[0]) == 30
    np.testing.assert_allclose(r["T"].cpu().numpy(), g["T"], rtol=RTOL, atol=1e-9)
    g = golden("icp_row_demo")
    T = api["icp_row"].ICP(g["src"], g["tgt"], 0.1, 20)
    np.testing.assert_allclose(T, g["T"], rtol=RTOL, atol=1e-9)


def test_icp_col_step(api):
    g = golden("closest_col2")
    nn, R, t, moved = api["icp"].icp_step(g["src"], g["tgt"])
    onn, oR, ot, omoved = orc.icp_col_step(g["src"], g["tgt"])
    assert np.array_equal(nn, onn)
    np.testing.assert_allclose(R, oR, rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(t, ot, rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(moved, omoved, rtol=RTOL, atol=1e-9)


def test_icp_100k_iteration_parity(api):
    """Each GPU iteration checked against the oracle on the SAME inputs (the moved
    source the GPU produced), so index parity is exact by construction."""
    n = 100000
    src, tgt = orc.make_rigid_cloud(n, seed=3)
    rigid, torch = api["rigid"], api["torch"]
    ctx = api["ctx"]
    s_d, t_d = ctx.to_dev(src), ctx.to_dev(tgt)
    ss = s_d.clone()
    for it in range(3):
        idx = rigid.closest_indices(ss, t_d)
        oi, _ = c_oracle.match_idx(ss.cpu().numpy(), tgt)
        assert np.array_equal(idx.cpu().numpy(), oi)
        nn = rigid.gather_points(t_d, idx)
        R, t = rigid.find_transform(s_d, nn)
        Ro, to = c_oracle.find_transform(src, tgt[oi])
        np.testing.assert_allclose(R.cpu().numpy(), Ro, rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(t.cpu().numpy(), to, rtol=RTOL, atol=1e-9)
        ss = rigid.apply_transform(s_d, R, t)
        np.testing.assert_allclose(ss.cpu().numpy(), src @ Ro + to, rtol=RTOL, atol=1e-9)
    r = rigid.rigid_icp(s_d, t_d, -np.inf, 3)
    np.testing.assert_allclose(r["T"].cpu().numpy()[:3, :3], R.cpu().numpy(), rtol=1e-12, atol=1e-14)


def test_match_full_size_properties(api):
    """BASELINE config C2 size (2^20 x 2^20): checked through size-independent properties
    -- a permuted copy matches itself exactly (d2 == 0, idx == inverse permutation), and a
    random sample of rows agrees with the C oracle."""
    n = 1 << 20
    rng = np.random.default_rng(1)
    tgt = rng.standard_normal((n, 3))
    perm = rng.permutation(n)
    idx, d2 = gpu_idx(api, tgt[perm], tgt, want_d2=True)
    assert np.array_equal(idx, perm) and not d2.any()
    src, _ = orc.make_rigid_cloud(n, seed=1)
    idx = gpu_idx(api, src, tgt)
    rows = rng.choice(n, 2048, replace=False)
    oi, _ = c_oracle.match_idx(src[rows], tgt)
    assert np.array_equal(idx[rows], oi)


def test_match_c4_size_properties(api):
    """BASELINE config C4 size on one GPU (2^24 targets, one rank's 2^21-row source shard):
    a permuted slice of the target matches itself exactly, and a sample of an independent
    shard agrees with the C oracle."""
    m, n = 1 << 24, 1 << 21
    rng = np.random.default_rng(2)
    tgt = rng.standard_normal((m, 3))
    rows = rng.choice(m, n, replace=False)
    idx, d2 = gpu_idx(api, tgt[rows], tgt, want_d2=True)
    assert np.array_equal(idx, rows) and not d2.any()
    src = tgt[rows] @ orc.rz(0.3).T + 0.01
    idx = gpu_idx(api, src, tgt)
    pick = rng.choice(n, 256, replace=False)
    oi, _ = c_oracle.match_idx(src[pick], tgt)
    assert np.array_equal(idx[pick], oi)


def test_fused_loop_equals_bruteforce_at_full_size(api):
    """The path bench.py times -- icp_rigid_run at 2^20 x 2^20 with the grid search, warm-start hints, sorted-position
    bookkeeping and the XCD 'eighths' block order -- against the same loop on the full scan (ICP_FORCE_BRUTE), 3
    iterations: identical correspondences; transform and error history to 1e-12 (the two loops add the same numbers in
    different orders: the brute loop does not Morton-sort its source).  And bench.py's own check: the last
    correspondences of a K-iteration run equal a brute Match of src . R_{K-1} + t_{K-1}, because the run is
    bit-reproducible."""
    torch = api["torch"]
    rigid = api["rigid"]
    src, tgt = orc.make_rigid_cloud(1 << 20, seed=1)
    sd, td = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    a = rigid.rigid_icp(sd, td, 0.0, 3, use_tol=False, want_idx=True)
    b = rigid.rigid_icp(sd, td, 0.0, 3, use_tol=False, want_idx=True, force_brute=True)
    torch.cuda.synchronize()
    assert torch.equal(a["idx"], b["idx"])
    np.testing.assert_allclose(a["T"].cpu().numpy(), b["T"].cpu().numpy(), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(a["err_hist"].cpu().numpy(), b["err_hist"].cpu().numpy(), rtol=1e-12)
    a2 = rigid.rigid_icp(sd, td, 0.0, 3, use_tol=False, want_idx=True)          # bit-reproducible
    assert torch.equal(a["T"], a2["T"]) and torch.equal(a["idx"], a2["idx"]) and torch.equal(a["err_hist"], a2["err_hist"])
    prev = rigid.rigid_icp(sd, td, 0.0, 2, use_tol=False)["T"]
    ss = rigid.apply_transform(sd, prev[:3, :3].contiguous(), prev[3, :3].contiguous())
    assert torch.equal(rigid.closest_indices(ss, td, force_brute=True), a["idx"])
    # the coherent loop (ICP_COHERENT: rows whose match is provably unchanged skip the search) at the bench's size and
    # iteration count: the plain loop's transform, correspondences and errors bit for bit, and the full scan's answer
    p30 = rigid.rigid_icp(sd, td, 0.0, 30, use_tol=False, want_idx=True)
    c30 = rigid.rigid_icp(sd, td, 0.0, 30, use_tol=False, want_idx=True, coherent=True)
    assert torch.equal(p30["T"], c30["T"]) and torch.equal(p30["idx"], c30["idx"]) and torch.equal(p30["err_hist"], c30["err_hist"])
    prev = rigid.rigid_icp(sd, td, 0.0, 29, use_tol=False, coherent=True)["T"]
    ss = rigid.apply_transform(sd, prev[:3, :3].contiguous(), prev[3, :3].contiguous())
    assert torch.equal(rigid.closest_indices(ss, td, force_brute=True), c30["idx"])


def test_host_icp_turns_the_coherent_loop_on_for_large_clouds(api, capsys):
    """icp_3d.ICP / icp_row.ICP (the reference's entry points, icp_3d.py:102-121) pass ICP_COHERENT from 131 072 source rows
    on: the transform and the printed error lines are the plain loop's bit for bit."""
    rigid = api["rigid"]
    src, tgt = orc.make_rigid_cloud(150000, seed=9)
    assert len(src) >= rigid.COHERENT_MIN_ROWS
    ref = rigid.rigid_icp(src, tgt, 1e-4, 20)
    capsys.readouterr()
    T = api["icp_3d"].ICP(src, tgt, 1e-4, 20)
    lines = capsys.readouterr().out.strip().splitlines()
    k = int(ref["iters"].cpu()[0])
    assert np.array_equal(T, ref["T"].cpu().numpy())
    assert lines == ["err = %f (%2d)" % (e, i) for i, e in enumerate(ref["err_hist"].cpu().numpy()[:k])]
    T2 = api["icp_row"].ICP(src, tgt, 0.0, 12)
    assert np.array_equal(T2, rigid.rigid_icp(src, tgt, 0.0, 12, use_tol=False)["T"].cpu().numpy())


# ---------------------------------------------------------------- icp_rigid_run_dist (source-sharded loop below Python)
@pytest.mark.parametrize("n,m,tol,run,dim", [(60000, 50000, 0.10, 50, 3), (40000, 45000, -np.inf, 7, 3), (3000, 2500, 0.05, 30, 2),
                                             (500, 700, 0.10, 50, 3)])
def test_rigid_run_dist_world1_equals_rigid_run(api, n, m, tol, run, dim):
    """One rank, no callbacks: the piggy-backed loop (stop test one Match late, two correspondence buffers) returns
    what icp_rigid_run returns, bit for bit -- transform, error history, iteration count, last correspondences."""
    from icp_amd import dist as icpdist
    rng = np.random.default_rng(n + m)
    tgt = rng.standard_normal((m, dim))
    A = orc.rot_matrix(0.05, -0.04, 0.06) if dim == 3 else orc.rot2(0.05)
    src = (tgt[rng.permutation(m)[:n] % m] - 0.02) @ A.T if n <= m else rng.standard_normal((n, dim))
    a = api["rigid"].rigid_icp(src, tgt, tol, run, want_idx=True)
    cm = icpdist.RigidComm("cuda", rank=0, world=1)
    b = icpdist.rigid_icp_dist(src, tgt, tol, run, want_idx=True, comm=cm)
    api["torch"].cuda.synchronize()
    k = int(a["iters"].cpu()[0])
    assert k == int(b["iters"].cpu()[0]) and 1 <= k <= run
    assert api["torch"].equal(a["T"], b["T"])
    assert api["torch"].equal(a["err_hist"][:k], b["err_hist"][:k])
    assert api["torch"].equal(a["idx"], b["idx"])
    assert cm.calls == []                                   # a single rank never calls back


def _dist_worker(rank, world, port, q, n, m, tol, run, seed):
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from icp_amd import dist as icpdist
        src, tgt = orc.make_rigid_cloud(m, seed=seed)
        src = src[:n]
        lo, hi = icpdist.shard_bounds(n, world, rank)
        cm = icpdist.RigidComm("cuda:0")
        out = icpdist.rigid_icp_dist(src[lo:hi], tgt, tol, run, want_idx=True, comm=cm)
        torch.cuda.synchronize()
        q.put((rank, dict(T=out["T"].cpu().numpy(), hist=out["err_hist"].cpu().numpy(), iters=int(out["iters"].cpu()[0]),
                          idx=out["idx"].cpu().numpy(), calls=list(cm.calls))))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,m,tol,run,world", [(90000, 100000, 0.10, 50, 2), (70001, 80000, -np.inf, 6, 2),
                                                (50002, 60000, -np.inf, 4, 3)])
def test_rigid_run_dist_two_ranks_one_gpu(api, n, m, tol, run, world):
    """Two ranks, or three ragged ones whose shards stay below the size that gets Morton-sorted (sharing the test box's one GPU, all-reduce over gloo with host staging): same correspondences as
    the single-GPU loop bit for bit, transform / errors to 1e-12 (the sums are added in another order), identical on
    both ranks, and exactly run + 2 collectives per rank of 4, 17 ... 17, 1 doubles whatever the stop test says."""
    import torch.multiprocessing as mp
    src, tgt = orc.make_rigid_cloud(m, seed=13)
    src = src[:n]
    ref = api["rigid"].rigid_icp(src, tgt, tol, run, want_idx=True)
    api["torch"].cuda.synchronize()
    k = int(ref["iters"].cpu()[0])
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [mpc.Process(target=_dist_worker, args=(r, world, port, q, n, m, tol, run, 13)) for r in range(world)]
    for p in procs:
        p.start()
    res = [o for _, o in sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for o in res:
        assert o["iters"] == k
        np.testing.assert_allclose(o["T"], ref["T"].cpu().numpy(), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(o["hist"][:k], ref["err_hist"].cpu().numpy()[:k], rtol=1e-12)
        assert o["calls"] == [4] + [17] * run + [1]
    for o in res[1:]:
        assert np.array_equal(res[0]["T"], o["T"]) and np.array_equal(res[0]["hist"][:k], o["hist"][:k])
    assert np.array_equal(np.concatenate([o["idx"] for o in res]), ref["idx"].cpu().numpy())


_STREAM_HANDOVER = r"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, %r)
from icp_amd import _lib
lib = _lib.load_library()
hip = ctypes.CDLL("libamdhip64.so")
rng = np.random.default_rng(3)
src = torch.from_numpy(rng.standard_normal((5000, 3))).cuda()
tgt = torch.from_numpy(rng.standard_normal((7000, 3))).cuda()
ref = torch.cdist(src, tgt).argmin(1).to(torch.int32)
torch.cuda.synchronize()
ctx = ctypes.c_void_p()
assert lib.icp_ctx_create(0, ctypes.byref(ctx)) == 0
def run(stream):
    rc = lib.icp_ctx_set_stream(ctx, stream)
    assert rc == 0, lib.icp_last_error()
    idx = torch.full((5000,), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = lib.icp_closest(ctx, src.data_ptr(), 5000, tgt.data_ptr(), 7000, 3, 0, idx.data_ptr(), None)
    assert rc == 0, lib.icp_last_error()
    assert lib.icp_sync(ctx) == 0
    assert torch.equal(idx, ref)
for mode in sys.argv[1:]:
    a = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(a)) == 0
    run(a)
    if mode == "detach":
        assert lib.icp_ctx_set_stream(ctx, None) == 0      # the documented hand-over: leave the stream first
    assert hip.hipStreamDestroy(a) == 0
    b = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(b)) == 0
    run(b)                                                  # "destroyed": the previous stream is gone at this call
    run(None)
    assert hip.hipStreamDestroy(b) == 0
    c = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(c)) == 0
    run(c)
    assert lib.icp_ctx_set_stream(ctx, None) == 0
    assert hip.hipStreamDestroy(c) == 0
assert lib.icp_ctx_destroy(ctx) == 0
print("handover ok")
"""


@pytest.mark.parametrize("mode", ["detach", "destroyed"])
def test_stream_handover_after_the_old_stream_is_destroyed(mode):
    """a C host may destroy its stream and hand the context a new one.  With the documented
    detach step (icp_ctx_set_stream(ctx, NULL) first) and without it -- the previous stream already gone when the
    next one is set -- the context must end up on the new stream and stay usable.  Run in a child process: what HIP
    does with a dead stream handle is its own business, and a crash there must read as this test's failure."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _STREAM_HANDOVER % root, mode], capture_output=True, text=True, timeout=110)
    assert out.returncode == 0 and "handover ok" in out.stdout, out.stdout + out.stderr


_SINGLE_PASS = r"""
import sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import icp_oracle as orc
from icp_amd import rigid
src, tgt = orc.make_rigid_cloud(150000, seed=21)
r = rigid.rigid_icp(src, tgt, -np.inf, 4, want_idx=True)
np.save(sys.argv[1], np.concatenate([r["T"].cpu().numpy().ravel(), r["err_hist"].cpu().numpy()[:4], r["idx"].cpu().numpy().astype(np.float64)]))
print("single pass ok")
"""


def test_rigid_run_under_the_single_pass_switch(tmp_path):
    """ICP_GRID_CAP=0 (the documented single-pass search) used to make icp_rigid_run fail at iteration 2 --
    the fused loop was chosen from the flags alone and the search then refused it.  The loop now takes the unfused form
    there: same transform and correspondences bit for bit, errors equal up to summation order."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for cap in (None, "0"):
        env = dict(os.environ)
        env.pop("ICP_GRID_CAP", None)
        if cap is not None:
            env["ICP_GRID_CAP"] = cap
        f = str(tmp_path / ("run_%s.npy" % cap))
        out = subprocess.run([sys.executable, "-c", _SINGLE_PASS % (root, os.path.join(root, "oracle")), f], capture_output=True,
                             text=True, timeout=110, env=env)
        assert out.returncode == 0 and "single pass ok" in out.stdout, out.stdout + out.stderr
        outs.append(np.load(f))
    a, b = outs
    assert np.array_equal(a[:16], b[:16]) and np.array_equal(a[20:], b[20:])
    np.testing.assert_allclose(a[16:20], b[16:20], rtol=1e-12)
