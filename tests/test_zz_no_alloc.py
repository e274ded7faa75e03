"""icp_find_transform_noalloc (include/icp_hip.h): the FindTransform entry point that allocates
nothing.  No counterpart in the reference; checked against the existing icp_find_transform through
the C-ABI, bit for bit (same kernels, same summation trees), on two calls in a row -- the second
with a smaller input, so stale rows of the shared device scratch array from the first call would
show -- and against the oracle at the north-star tolerance."""
import ctypes

import numpy as np
import pytest

import c_oracle

pytestmark = pytest.mark.gpu


def _both(ctx, torch, lib, s_np, t_np):
    from icp_amd._lib import ptr, check
    dim = s_np.shape[1]
    s = torch.from_numpy(s_np).cuda()
    t = torch.from_numpy(t_np).cuda()
    out = []
    for name in ("icp_find_transform", "icp_find_transform_noalloc"):
        # poison the outputs so an entry point that wrote nothing cannot pass
        R = torch.full((dim, dim), float("nan"), dtype=torch.float64, device="cuda")
        tv = torch.full((dim,), float("nan"), dtype=torch.float64, device="cuda")
        ctx.use_current_stream()
        check(getattr(lib, name)(ctx.handle, ptr(s), ptr(t), s.shape[0], dim, ptr(R), ptr(tv)))
        torch.cuda.synchronize()
        out.append((R.cpu().numpy(), tv.cpu().numpy()))
    return out


@pytest.mark.parametrize("dim", [3, 2])
def test_noalloc_matches_find_transform_twice(dim):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from icp_amd._lib import get_context, load_library
    ctx, lib = get_context(), load_library()
    rng = np.random.default_rng(11 + dim)
    for n in (300001, 777):                      # the second call is smaller than the first
        s = rng.standard_normal((n, dim)) + 3.0
        A = np.linalg.qr(rng.standard_normal((dim, dim)))[0]
        t = s @ A + rng.standard_normal(dim) + 1e-3 * rng.standard_normal((n, dim))
        (R0, t0), (R1, t1) = _both(ctx, torch, lib, s, t)
        assert np.isfinite(R1).all() and np.isfinite(t1).all()
        assert np.array_equal(R0, R1) and np.array_equal(t0, t1)
        Ro, to = c_oracle.find_transform(s, t)
        np.testing.assert_allclose(R1, Ro, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(t1, to, rtol=1e-6, atol=1e-9)


def test_noalloc_rejects_oversize_and_bad_args():
    import torch
    from icp_amd._lib import get_context, load_library, ptr
    ctx, lib = get_context(), load_library()
    x = torch.zeros((4, 3), dtype=torch.float64, device="cuda")
    R = torch.zeros((3, 3), dtype=torch.float64, device="cuda")
    tv = torch.zeros((3,), dtype=torch.float64, device="cuda")
    rc = lib.icp_find_transform_noalloc(ctx.handle, ptr(x), ptr(x), (1 << 27) + 1, 3, ptr(R), ptr(tv))
    assert rc == -4 and b"exceeds the limit" in lib.icp_last_error()
    assert lib.icp_find_transform_noalloc(ctx.handle, ptr(x), ptr(x), 0, 3, ptr(R), ptr(tv)) == -1
    assert lib.icp_find_transform_noalloc(ctx.handle, ptr(x), ptr(x), 4, 4, ptr(R), ptr(tv)) == -1
