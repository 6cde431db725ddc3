"""ctypes front-end of the CPU checker (oracle/jv_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package never does.

`lapjv_seeded` / `lapjv` mirror the reference's Python wrappers
(LAP/lap/_seeded_jv.pyx:14-31, LAP/_lapjv_cpp/_lapjv.pyx:38-129, square path)
including return order, dtypes, exceptions and how `cost` is summed.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "libjv_oracle.so"

BRANCH_NAMES = {0: "none", 1: "ssp", 2: "all_matched", 3: "fallback", 4: "cold"}


class Stats(ct.Structure):
    _fields_ = [(name, ct.c_longlong) for name in (
        "branch", "proj_events", "tight_edges", "free_rows", "arr_fired", "paths", "finds",
        "scan_steps", "scan_elems", "init_elems", "colred_elems", "transfer_rows", "arr_iters")]

    def as_dict(self) -> dict:
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build(force: bool = False) -> Path:
    """Compile the C restatement (and, where /root/reference exists, oracle/_ref)."""
    src = _HERE / "jv_oracle.c"
    stale = (not _LIB_PATH.exists()) or _LIB_PATH.stat().st_mtime < src.stat().st_mtime
    if force or stale:
        subprocess.run(["make", "-C", str(_HERE), "oracle"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ct.CDLL(str(_LIB_PATH))
        dp = ct.POINTER(ct.c_double)
        llp = ct.POINTER(ct.c_longlong)
        ip = ct.POINTER(ct.c_int)
        lib.jvo_lapjv_seeded_ex.restype = ct.c_int
        lib.jvo_lapjv_seeded_ex.argtypes = [dp, ct.c_int, ct.c_int, llp, llp, dp, dp, ct.c_double,
                                            ct.POINTER(Stats), dp, dp]
        lib.jvo_lapjv_seeded.restype = ct.c_int
        lib.jvo_lapjv_seeded.argtypes = [dp, ct.c_int, ct.c_int, llp, llp, dp, dp, ct.c_double]
        lib.jvo_lapjv_dense.restype = ct.c_int
        lib.jvo_lapjv_dense.argtypes = [dp, ct.c_int, ip, ip, ct.POINTER(Stats)]
        _lib = lib
    return _lib


def _dptr(a):
    return a.ctypes.data_as(ct.POINTER(ct.c_double))


def seeded_raw(C, u, v, eps: float = 1e-12, want_duals: bool = False):
    """Returns (ret, x int64[n], y int64[n], stats dict[, u_final, v_final])."""
    lib = _load()
    C = np.ascontiguousarray(C, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    n, m = C.shape
    x = np.full(n, -1, dtype=np.int64)
    y = np.full(m, -1, dtype=np.int64)
    st = Stats()
    uf = np.zeros(n) if want_duals else None
    vf = np.zeros(n) if want_duals else None
    ret = lib.jvo_lapjv_seeded_ex(
        _dptr(C), n, m, x.ctypes.data_as(ct.POINTER(ct.c_longlong)),
        y.ctypes.data_as(ct.POINTER(ct.c_longlong)), _dptr(u), _dptr(v), float(eps),
        ct.byref(st), _dptr(uf) if want_duals else None, _dptr(vf) if want_duals else None)
    if want_duals:
        return ret, x, y, st.as_dict(), uf, vf
    return ret, x, y, st.as_dict()


def lapjv_seeded(C, u, v, eps: float = 1e-12):
    """Mirror of lap.lapjv_seeded (LAP/lap/_seeded_jv.pyx:14-31): returns (x, y, cost)."""
    C = np.asarray(C)
    if C.shape[0] != len(u) or C.shape[1] != len(v):
        raise ValueError("u/v sizes must match C")
    ret, x, y, _ = seeded_raw(C, u, v, eps)
    if ret != 0:
        if ret == -3:
            raise ValueError("Infeasible seed potentials: C - u - v has negatives")
        raise RuntimeError(f"lapjv_seeded internal error (code {ret})")
    n = C.shape[0]
    cost = float(np.sum(np.asarray(C, dtype=np.float64)[np.arange(n), x]))
    return x, y, cost


def dense_raw(C):
    """Cold JV: returns (ret, x int32[n], y int32[n], stats dict)."""
    lib = _load()
    C = np.ascontiguousarray(C, dtype=np.float64)
    n = C.shape[0]
    x = np.empty(n, dtype=np.int32)
    y = np.empty(n, dtype=np.int32)
    st = Stats()
    ret = lib.jvo_lapjv_dense(_dptr(C), n, x.ctypes.data_as(ct.POINTER(ct.c_int)),
                              y.ctypes.data_as(ct.POINTER(ct.c_int)), ct.byref(st))
    return ret, x, y, st.as_dict()


def lapjv(cost, extend_cost: bool = False, cost_limit: float = np.inf, return_cost: bool = True):
    """Mirror of lap.lapjv's square / no-limit path (_lapjv.pyx:38-129)."""
    cost = np.asarray(cost)
    if cost.ndim != 2:
        raise ValueError("2-dimensional array expected")
    if cost.shape[0] != cost.shape[1] and not extend_cost:
        raise ValueError("Square cost array expected. If cost is intentionally "
                         "non-square, pass extend_cost=True.")
    if extend_cost or cost_limit < np.inf:
        raise NotImplementedError("only the square, unlimited path is restated")
    cost_c = np.ascontiguousarray(cost, dtype=np.double)
    ret, x, y, _ = dense_raw(cost_c)
    if ret != 0:
        if ret == -1:
            raise MemoryError("Out of memory.")
        raise RuntimeError("Unknown error (lapjv_internal returned %d)." % ret)
    if return_cost:
        n = cost_c.shape[0]
        return cost_c[np.arange(n), x].sum(), x, y
    return x, y


def serial_elems(stats: dict, n: int) -> int:
    """E of SURVEY.md section 8(d): element visits of the serial phase."""
    return int(stats["init_elems"] + stats["scan_elems"] + stats["colred_elems"]
               + n * stats["transfer_rows"] + n * stats["arr_iters"])


if __name__ == "__main__":  # tiny self-check
    rng = np.random.RandomState(0)
    C = rng.uniform(size=(64, 64))
    u = C.min(1)
    v = (C - u[:, None]).min(0)
    print(lapjv_seeded(C, u, v)[2], lapjv(C)[0], os.getpid())
