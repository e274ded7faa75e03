"""ctypes binding of libicp_hip.so (include/icp_hip.h) + device-buffer plumbing.

PyTorch-ROCm is used only for device memory and streams: every call below hands
raw device pointers (``tensor.data_ptr()``) and the current HIP stream to the
C-ABI.  There is NO CPU fallback: if the shared library is missing or no GPU is
visible the product path raises.
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libicp_hip.so")

ICP_COMPARE_SQRT = 1
ICP_FORCE_BRUTE = 2
ICP_FORCE_GRID = 4
ICP_HINT_IDX = 8
ICP_NO_FUSE = 16
ICP_COHERENT = 32

c_dp = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_u32 = ctypes.c_uint32
c_dbl = ctypes.c_double

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/icp_hip.h
_SIGNATURES = {
    "icp_version": [],
    "icp_ctx_create": [c_int, ctypes.POINTER(ctypes.c_void_p)],
    "icp_ctx_destroy": [c_dp],
    "icp_ctx_set_stream": [c_dp, c_dp],
    "icp_sync": [c_dp],
    "icp_malloc": [c_dp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)],
    "icp_free": [c_dp, c_dp],
    "icp_memcpy_h2d": [c_dp, c_dp, c_dp, ctypes.c_size_t],
    "icp_memcpy_d2h": [c_dp, c_dp, c_dp, ctypes.c_size_t],
    "icp_closest": [c_dp, c_dp, c_i64, c_dp, c_i64, c_int, c_u32, c_dp, c_dp],
    "icp_target_create": [c_dp, c_dp, c_i64, c_int, ctypes.POINTER(ctypes.c_void_p)],
    "icp_target_destroy": [c_dp, c_dp],
    "icp_closest_target": [c_dp, c_dp, c_dp, c_i64, c_u32, c_dp, c_dp],
    "icp_morton_sort": [c_dp, c_dp, c_i64, c_int, c_dp, c_dp],
    "icp_gather_points": [c_dp, c_dp, c_dp, c_i64, c_int, c_dp],
    "icp_distance_sum": [c_dp, c_dp, c_dp, c_i64, c_int, c_dp],
    "icp_find_transform": [c_dp, c_dp, c_dp, c_i64, c_int, c_dp, c_dp],
    "icp_find_transform_noalloc": [c_dp, c_dp, c_dp, c_i64, c_int, c_dp, c_dp],
    "icp_apply_transform": [c_dp, c_dp, c_i64, c_int, c_dp, c_dp, c_dp],
    "icp_rigid_run": [c_dp, c_dp, c_i64, c_dp, c_i64, c_int, c_dbl, c_int, c_int, c_u32, c_dp, c_dp, c_dp, c_dp],
    "icp_rigid_run_dist": [c_dp, c_dp, c_i64, c_dp, c_i64, c_int, c_dbl, c_int, c_int, c_u32, c_dp, c_dp, c_dp, c_dp, c_dp],
    "icp_shard_sums": [c_dp, c_dp, c_dp, c_dp, c_i64, c_int, c_dp, c_dp, c_dp],
    "icp_kabsch_solve": [c_dp, c_dp, c_int, c_dp, c_dp, c_dp, c_dp],
    "icp_shard_apply_err": [c_dp, c_dp, c_dp, c_dp, c_i64, c_int, c_dp, c_dp, c_dp, c_dp],
    "icp_column_sums": [c_dp, c_dp, c_i64, c_int, c_dp],
    "icp_bench_grid_query": [c_dp, c_dp, c_i64, c_dp, c_i64, c_int, c_int, c_dp],
    "icp_rccl_unique_id": [c_dp],
    "icp_rccl_create": [c_dp, c_dp, c_int, c_int, ctypes.POINTER(ctypes.c_void_p)],
    "icp_rccl_destroy": [c_dp],
    "nicp_knn": [c_dp, c_dp, c_i64, c_int, c_dp, c_i64, c_int, c_dp, c_dp],
    "nicp_knn_weights": [c_dp, c_dp, c_i64, c_int, c_int, c_dp, c_dp],
    "nicp_graph_create": [c_dp, c_dp, c_dp, c_i64, c_int, c_dp, c_i64, c_int, ctypes.POINTER(ctypes.c_void_p)],
    "nicp_graph_destroy": [c_dp, c_dp],
    "nicp_set_weights": [c_dp, c_dp, c_dp, c_dp, c_dp],
    "nicp_set_landmarks": [c_dp, c_dp, c_dp, c_dp, c_i64, c_dbl],
    "nicp_rows": [c_dp, ctypes.POINTER(c_i64)],
    "nicp_deform": [c_dp, c_dp, c_dp, c_dp, c_dp],
    "nicp_residual": [c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_u32, c_dp],
    "nicp_jvp": [c_dp, c_dp, c_dp, c_dp, c_dp, c_u32, c_dp, c_dp],
    "nicp_jtvp": [c_dp, c_dp, c_dp, c_dp, c_dp, c_u32, c_dp, c_dp],
    "nicp_solve": [c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_u32, c_dp, c_dp],
    "nicp_profile_cg": [c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_u32, c_int, c_dp],
    "nicp_solve_dist": [c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_u32, c_dp, c_dp, c_dp],
    "nicp_run": [c_dp, c_dp, c_dp, c_dp, c_dp, c_i64, c_dbl, c_int, c_dp, c_u32, c_dp, c_dp, c_dp, c_dp],
}


class SolveOpts(ctypes.Structure):
    """nicp_solve_opts of include/icp_hip.h"""
    _fields_ = [("ftol", c_dbl), ("xtol", c_dbl), ("gtol", c_dbl), ("lambda0", c_dbl), ("cg_tol", c_dbl),
                ("max_lm_iters", c_int), ("max_cg_iters", c_int), ("accel", c_int)]


class SolveStats(ctypes.Structure):
    """nicp_solve_stats of include/icp_hip.h"""
    _fields_ = [("err", c_dbl), ("err0", c_dbl), ("lm_iters", c_int), ("cg_iters", c_int), ("nfev", c_int),
                ("info", c_int), ("graph_launches", c_int)]


class CgProfile(ctypes.Structure):
    """nicp_cg_profile of include/icp_hip.h"""
    _fields_ = [("iterations", c_int), ("launches_per_iteration", c_int), ("graph_replay", c_int), ("coarse_columns", c_int),
                ("vertices", c_i64), ("edges", c_i64), ("us_iteration", c_dbl), ("us_kernel", c_dbl * 6)]


NICP_REFERENCE_COMPAT = 1
ICP_COMM_FORCE = 1
ICP_RCCL_ID_BYTES = 128
_RESTYPES = {"icp_last_error": ctypes.c_char_p}

_lib = None
_lock = threading.Lock()
_ctxs = {}


class IcpHipError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen libicp_hip.so and declare every symbol of include/icp_hip.h.
    Raises if the library (or any symbol) is missing -- never falls back."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise IcpHipError(
            "libicp_hip.so not found at %s -- build it with `make -C icp_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback" % p)
    lib = ctypes.CDLL(p)
    lib.icp_last_error.restype = ctypes.c_char_p
    lib.icp_last_error.argtypes = []
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = c_int
    if path is None:
        _lib = lib
    return lib


def exported_symbols():
    return ["icp_last_error"] + list(_SIGNATURES)


def check(rc):
    if rc != 0:
        msg = load_library().icp_last_error()
        raise IcpHipError("libicp_hip error %d: %s" % (rc, (msg or b"").decode()))


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise IcpHipError("no HIP device visible: icp_amd needs an AMD GPU (no CPU fallback)")
    return torch


class Context:
    """One icp_ctx per device; tracks torch's current stream on every call."""

    def __init__(self, device=None):
        torch = _torch()
        self.torch = torch
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.lib = load_library()
        h = ctypes.c_void_p()
        check(self.lib.icp_ctx_create(self.device, ctypes.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.icp_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def use_current_stream(self):
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.icp_ctx_set_stream(self.handle, ctypes.c_void_p(s)))

    def call(self, name, *args):
        self.use_current_stream()
        check(getattr(self.lib, name)(self.handle, *args))

    def call_fast(self, name, *args):
        """call() without re-reading torch's current stream (the caller did use_current_stream())."""
        check(getattr(self.lib, name)(self.handle, *args))

    # -- buffers ---------------------------------------------------------
    def to_dev(self, a, dtype=None):
        """numpy array / torch tensor -> contiguous tensor of `dtype` (default float64: every kernel of the
        library computes in fp64, like the reference's numpy arrays) ON THIS CONTEXT'S DEVICE.  A tensor of
        another dtype (torch's default float32 included) or on another GPU is converted / moved, never handed
        over as it is: the C-ABI takes raw pointers and would read it as fp64 on this device."""
        torch = self.torch
        want = torch.float64 if dtype is None else dtype
        if isinstance(a, torch.Tensor):
            dev = torch.device("cuda", self.device)
            t = a
            if t.dtype != want or t.device != dev:
                t = t.to(device=dev, dtype=want)
            return t.contiguous()
        arr = np.ascontiguousarray(a, dtype=np.float64 if want == torch.float64 else np.int32)
        return torch.from_numpy(arr).cuda(self.device)

    def empty(self, shape, dtype=None):
        torch = self.torch
        return torch.empty(shape, dtype=dtype or torch.float64, device="cuda:%d" % self.device)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def get_context(device=None):
    torch = _torch()
    dev = torch.cuda.current_device() if device is None else int(device)
    with _lock:
        if dev not in _ctxs:
            _ctxs[dev] = Context(dev)
        return _ctxs[dev]


class LibraryRccl:
    """The library's own RCCL communicator (include/icp_hip.h, icp_rccl_*): rank 0 makes the unique id, the host
    distributes it -- here over torch.distributed, whatever its backend -- and every rank joins collectively.  With
    world == 1 no process group is needed (one-rank communicator: rehearsals and tests on one GPU)."""

    def __init__(self, ctx=None, rank=None, world=None, group=None):
        self.ctx = ctx or get_context()
        torch = self.ctx.torch
        lib = self.ctx.lib
        if world is None or world > 1:
            import torch.distributed as dist
            rank = dist.get_rank(group) if rank is None else rank
            world = dist.get_world_size(group) if world is None else world
        self.rank, self.world = int(rank or 0), int(world)
        ident = (ctypes.c_char * ICP_RCCL_ID_BYTES)()
        if self.rank == 0:
            check(lib.icp_rccl_unique_id(ident))
        if self.world > 1:
            import torch.distributed as dist
            dev = "cuda:%d" % self.ctx.device if dist.get_backend(group) == "nccl" else "cpu"
            t = torch.frombuffer(bytearray(bytes(ident)), dtype=torch.uint8).to(dev)
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = (ctypes.c_char * ICP_RCCL_ID_BYTES).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
        h = ctypes.c_void_p()
        torch.cuda.synchronize(self.ctx.device)
        check(lib.icp_rccl_create(self.ctx.handle, ident, self.rank, self.world, ctypes.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.torch.cuda.synchronize(self.ctx.device)
            self.ctx.lib.icp_rccl_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
