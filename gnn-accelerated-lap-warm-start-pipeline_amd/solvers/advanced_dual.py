"""Dual utilities of the hot path, on the GPU (reference: solvers/advanced_dual.py:14-63).

Each function keeps the reference's signature and semantics; the dense row/column sweeps run
as HIP kernels over the C ABI (lapwarm_project_feasible / lapwarm_reduce_costs)."""
from typing import Tuple

import numpy as np

from lap import _hip


def _mat(C):
    return np.ascontiguousarray(np.asarray(C, dtype=float), dtype=np.float64)


def project_feasible(C: np.ndarray, u: np.ndarray, v: np.ndarray,
                     max_rounds: int = 50, tol: float = 1e-12) -> Tuple[np.ndarray, np.ndarray]:
    """u <- min(u, rowmin(C - v)); v <- min(v, colmin(C - u)); until min(C-u-v) >= -tol."""
    C = _mat(C)
    u = np.array(u, dtype=np.float64).copy()
    v = np.array(v, dtype=np.float64).copy()
    n = C.shape[0]
    if n == 0:
        return u, v
    lib = _hip.require_device()
    rc = lib.lapwarm_project_feasible(C.ctypes.data_as(_hip.c_dp), n, u.ctypes.data_as(_hip.c_dp),
                                      v.ctypes.data_as(_hip.c_dp), max(1, int(max_rounds)), float(tol))
    if _hip.check(rc, "project_feasible") != 0:
        raise RuntimeError(f"project_feasible failed (code {rc})")
    return u, v


def _reduce(C, u, v, shift_nonneg, want_matrix=True):
    C = _mat(C)
    u = np.ascontiguousarray(u, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    n = C.shape[0]
    out = np.empty_like(C) if want_matrix else None
    mn = np.zeros(1)
    lib = _hip.require_device()
    rc = lib.lapwarm_reduce_costs(C.ctypes.data_as(_hip.c_dp), n, u.ctypes.data_as(_hip.c_dp),
                                  v.ctypes.data_as(_hip.c_dp), int(bool(shift_nonneg)),
                                  out.ctypes.data_as(_hip.c_dp) if want_matrix else None,
                                  mn.ctypes.data_as(_hip.c_dp))
    if _hip.check(rc, "reduce_costs") != 0:
        raise RuntimeError(f"reduce_costs failed (code {rc})")
    return out, float(mn[0])


def reduce_costs(C: np.ndarray, u: np.ndarray, v: np.ndarray, shift_nonneg: bool = True) -> np.ndarray:
    """C' = C - u 1^T - 1 v^T; with shift_nonneg, minus min(C') when that is negative."""
    if np.asarray(C).shape[0] == 0:
        return np.ascontiguousarray(np.asarray(C, dtype=np.float64))
    return _reduce(C, u, v, shift_nonneg)[0]


def check_dual_feasible(C: np.ndarray, u: np.ndarray, v: np.ndarray, tol: float = 1e-8) -> bool:
    """Raises AssertionError when some reduced cost is below -tol."""
    _, mn = _reduce(C, u, v, False, want_matrix=False)
    if mn < -tol:
        raise AssertionError(f"Dual infeasible: min reduced cost {mn:.3e} < -tol")
    return True
