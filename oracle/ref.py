"""ctypes front-end of oracle/_ref/liblap_ref.so -- TEST INFRASTRUCTURE.

liblap_ref.so is the reference's own C++ (LAP/_lapjv_cpp/lapjv.cpp and
lapjv_seeded.cpp) compiled unmodified by oracle/Makefile where /root/reference
exists.  It validates the restatement in jv_oracle.c (tests/golden/make_golden.py,
tests/test_oracle_golden.py) and bench.py times it beside the port on a few
instances (cpu_baseline.port_over_reference).  It is git-ignored and travels to
the GPU box as a prebuilt binary only -- no reference source does.
"""
from __future__ import annotations

import ctypes as ct
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "_ref" / "liblap_ref.so"
REF_ROOT = Path("/root/reference")

_lib = None


def available() -> bool:
    if not LIB_PATH.exists() and REF_ROOT.exists():
        subprocess.run(["make", "-C", str(_HERE), "ref"], check=False, capture_output=True)
    return LIB_PATH.exists()


def _load():
    global _lib
    if _lib is None:
        if not available():
            raise FileNotFoundError(f"{LIB_PATH} not built (needs /root/reference)")
        lib = ct.CDLL(str(LIB_PATH))
        dp = ct.POINTER(ct.c_double)
        llp = ct.POINTER(ct.c_longlong)
        lib.lapjv_seeded.restype = ct.c_int
        lib.lapjv_seeded.argtypes = [dp, ct.c_int, ct.c_int, llp, llp, dp, dp, ct.c_double]
        # int lapjv_internal(const uint_t n, cost_t *cost[], int_t *x, int_t *y)  (C++ linkage)
        fn = getattr(lib, "_Z14lapjv_internaljPPdPiS1_")
        fn.restype = ct.c_int
        fn.argtypes = [ct.c_uint, ct.POINTER(dp), ct.POINTER(ct.c_int), ct.POINTER(ct.c_int)]
        lib.lapjv_internal = fn
        _lib = lib
    return _lib


def seeded_raw(C, u, v, eps: float = 1e-12):
    lib = _load()
    C = np.ascontiguousarray(C, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    n, m = C.shape
    x = np.full(n, -1, dtype=np.int64)
    y = np.full(m, -1, dtype=np.int64)
    dp = ct.POINTER(ct.c_double)
    llp = ct.POINTER(ct.c_longlong)
    ret = lib.lapjv_seeded(C.ctypes.data_as(dp), n, m, x.ctypes.data_as(llp), y.ctypes.data_as(llp),
                           u.ctypes.data_as(dp), v.ctypes.data_as(dp), float(eps))
    return ret, x, y


def dense_raw(C):
    lib = _load()
    C = np.ascontiguousarray(C, dtype=np.float64)
    n = C.shape[0]
    dp = ct.POINTER(ct.c_double)
    rows = (dp * n)()
    base = C.ctypes.data
    for i in range(n):
        rows[i] = ct.cast(base + i * n * 8, dp)
    x = np.empty(n, dtype=np.int32)
    y = np.empty(n, dtype=np.int32)
    ret = lib.lapjv_internal(n, rows, x.ctypes.data_as(ct.POINTER(ct.c_int)),
                             y.ctypes.data_as(ct.POINTER(ct.c_int)))
    return ret, x, y
