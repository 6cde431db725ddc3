"""lapjv: mirror of LAP/_lapjv_cpp/_lapjv.pyx:38-129 (square, unlimited path) over the HIP C ABI."""
from __future__ import annotations

import numpy as np

from . import _hip

LARGE_ = 1000000  # LAP/_lapjv_cpp/lapjv.h:4
FP_1_, FP_2_, FP_DYNAMIC_ = 1, 2, 3


def lapjv(cost, extend_cost=False, cost_limit=np.inf, return_cost=True):
    """Jonker-Volgenant on a dense square cost matrix: returns ``(opt, x, y)`` (int32 x, y)."""
    if cost is None:
        raise TypeError("Argument 'cost' must not be None")
    cost = np.asarray(cost)
    if cost.ndim != 2:
        raise ValueError("2-dimensional array expected")
    cost_c = np.ascontiguousarray(cost, dtype=np.double)
    n_rows, n_cols = cost_c.shape
    if n_rows != n_cols and not extend_cost:
        raise ValueError("Square cost array expected. If cost is intentionally "
                         "non-square, pass extend_cost=True.")
    if extend_cost or cost_limit < np.inf:
        raise NotImplementedError(
            "extend_cost / cost_limit (rectangular and thresholded problems, _lapjv.pyx:79-95) are "
            "outside the warm-start hot path and not built here")
    n = n_rows
    x = np.empty((n,), dtype=np.int32)
    y = np.empty((n,), dtype=np.int32)
    if n > 0:
        lib = _hip.require_device()
        ret = lib.lapwarm_lapjv_dense(cost_c.ctypes.data_as(_hip.c_dp), n, x.ctypes.data_as(_hip.c_ip),
                                      y.ctypes.data_as(_hip.c_ip))
        _hip.check(ret, "lapjv")
        if ret != 0:
            if ret == -1:
                raise MemoryError("Out of memory.")
            raise RuntimeError("Unknown error (lapjv_internal returned %d)." % ret)
    if return_cost:
        opt = cost_c[np.arange(n_rows), x].sum()
        return opt, x, y
    return x, y
