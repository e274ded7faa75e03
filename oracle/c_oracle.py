"""ctypes loader of the C oracle (oracle/icp_oracle.c) -- test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.oracle_match.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.oracle_match.restype = None
        L.oracle_distance.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
        L.oracle_distance.restype = ctypes.c_double
        L.oracle_kabsch_sums.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_kabsch_sums.restype = None
        L.oracle_knn_rows.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_int]
        L.oracle_knn_rows.restype = None
        _lib = L
    return _lib


def match_idx(src, tgt, compare_sqrt=False, threads=None):
    src = np.ascontiguousarray(src, dtype=np.float64)
    tgt = np.ascontiguousarray(tgt, dtype=np.float64)
    n, dim = src.shape
    idx = np.empty(n, np.int32)
    d2 = np.empty(n, np.float64)
    threads = threads or min(os.cpu_count() or 1, 16)
    lib().oracle_match(src.ctypes.data, n, tgt.ctypes.data, tgt.shape[0], dim, int(compare_sqrt), idx.ctypes.data,
                       d2.ctypes.data, threads)
    return idx, d2


def find_transform(src, nn):
    """FindTransform with the C sums + numpy's SVD (icp_3d.py:68-90)."""
    src = np.ascontiguousarray(src, dtype=np.float64)
    nn = np.ascontiguousarray(nn, dtype=np.float64)
    n, dim = src.shape
    sm, tm, W = np.empty(dim), np.empty(dim), np.empty((dim, dim))
    lib().oracle_kabsch_sums(src.ctypes.data, nn.ctypes.data, n, dim, sm.ctypes.data, tm.ctypes.data, W.ctypes.data)
    U, S, VT = np.linalg.svd(W)
    R = (U @ VT).T
    return R, tm - sm @ R


def knn_rows(v, k, threads=None):
    v = np.ascontiguousarray(v, dtype=np.float64)
    n, dim = v.shape
    idx = np.empty((n, k), np.int32)
    dist = np.empty((n, k), np.float64)
    threads = threads or min(os.cpu_count() or 1, 16)
    lib().oracle_knn_rows(v.ctypes.data, n, dim, k, idx.ctypes.data, dist.ctypes.data, threads)
    return idx, dist
