"""lapjv_seeded: mirror of the Cython wrapper LAP/lap/_seeded_jv.pyx:14-31 over the HIP C ABI."""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import _hip


def _typed(name, a, ndim):
    # the Cython wrapper declares cnp.ndarray[double, ndim=.., mode='c'] arguments: anything
    # else is rejected with ValueError/TypeError before the solver runs
    if not isinstance(a, np.ndarray):
        raise TypeError(f"Argument '{name}' has incorrect type (expected numpy.ndarray, got {type(a).__name__})")
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions (expected %d, got %d)" % (ndim, a.ndim))
    if a.dtype != np.float64:
        raise ValueError("Buffer dtype mismatch, expected 'double' but got '%s'" % a.dtype.name)
    if not a.flags.c_contiguous:
        raise ValueError("ndarray is not C-contiguous")
    return a


def lapjv_seeded(C, u, v, eps: float = 1e-12):
    """Solve the dense square LAP seeded with dual potentials (u, v).

    Returns ``(x, y, cost)`` -- note the order, opposite to :func:`lap.lapjv` -- with
    ``x[i]`` the column of row i and ``y[j]`` the row of column j (int64), and
    ``cost = float(np.sum(C[arange(n), x]))`` summed on the host exactly as the reference does.
    """
    C = _typed("C", C, 2)
    u = _typed("u", u, 1)
    v = _typed("v", v, 1)
    n, m = C.shape
    if u.shape[0] != n or v.shape[0] != m:
        raise ValueError("u/v sizes must match C")
    x = np.full((n,), -1, dtype=np.int64)
    y = np.full((m,), -1, dtype=np.int64)
    if n == 0 or m == 0:
        ret = -2
    else:
        lib = _hip.require_device()
        ret = lib.lapjv_seeded(C.ctypes.data_as(_hip.c_dp), n, m, x.ctypes.data_as(_hip.c_llp),
                               y.ctypes.data_as(_hip.c_llp), u.ctypes.data_as(_hip.c_dp),
                               v.ctypes.data_as(_hip.c_dp), ct.c_double(eps))
        _hip.check(ret, "lapjv_seeded")
    if ret != 0:
        if ret == -3:
            raise ValueError("Infeasible seed potentials: C - u - v has negatives")
        raise RuntimeError(f"lapjv_seeded internal error (code {ret})")
    cost = float(np.sum(C[np.arange(n), x]))
    return x, y, cost
