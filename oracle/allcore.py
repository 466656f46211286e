"""ctypes wrapper of cpu_allcore.c — the all-core OpenMP CSR aggregation timed by bench.py's cpu_baseline as the "best
CPU" line.  TEST / BENCH INFRASTRUCTURE ONLY; built ON the machine that runs it (-march=native)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgnn_allcore.so")
_lib = None


def lib(rebuild: bool = False):
    global _lib
    if _lib is None or rebuild:
        src = os.path.join(_HERE, "cpu_allcore.c")
        if rebuild or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libgnn_allcore.so"] +
                                  (["ALLCORE_ARCH=-march=native"] if rebuild else []))
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.ac_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def threads() -> int:
    return int(lib().ac_threads())


def build_csr(s, t, n):
    s = np.ascontiguousarray(s, np.int64)
    t = np.ascontiguousarray(t, np.int64)
    rowptr = np.empty(n + 1, np.int64)
    col = np.empty(len(s) + n, np.int32)
    lib().ac_build_csr(_p(s), _p(t), ctypes.c_int64(len(s)), ctypes.c_int64(n), _p(rowptr), _p(col))
    return rowptr, col


def spmm(rowptr, col, x, c=None):
    x = np.ascontiguousarray(x, np.float32)
    n, D = x.shape
    out = np.empty_like(x)
    cp = None if c is None else _p(np.ascontiguousarray(c, np.float32))
    lib().ac_spmm_csr_omp(_p(rowptr), _p(col), _p(x), cp, ctypes.c_int64(n), ctypes.c_int64(D), _p(out))
    return out
