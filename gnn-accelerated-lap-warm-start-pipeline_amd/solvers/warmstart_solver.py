"""WarmStartLAPSolver: cold JV on the reduced costs C' = C - u 1^T - 1 v^T
(reference: solvers/warmstart_solver.py:15-66; returns (arange(n), x, C[rows, x].sum())).

On the MI355X the reduced matrix never exists on the host: one copy of C goes to the device,
`lapwarm_warmstart_lapjv` forms C' there (reduce_costs, optional non-negative shift) and runs the
cold JV on it; only the assignment comes back.  `use_lap=False` (the reference's SciPy branch)
reduces on the device and solves with SciPy on the host, as a cross-check.
"""
from __future__ import annotations

import numpy as np

from lap import _hip
from .advanced_dual import reduce_costs
from .lap_solver import _HostSolver, _f64


class WarmStartLAPSolver(_HostSolver):
    name = "WarmStartLAP"

    def __init__(self, use_lap=True):
        self.name = type(self).name
        self.use_lap = use_lap

    def solve(self, C, u, v, shift_nonneg: bool = True):
        C = np.ascontiguousarray(_f64(C))
        n = C.shape[0]
        rows = np.arange(n, dtype=np.int64)
        if n == 0:
            return rows, rows.copy(), 0.0
        if self.use_lap:
            u = np.ascontiguousarray(_f64(u))
            v = np.ascontiguousarray(_f64(v))
            if C.ndim != 2 or C.shape[1] != n or u.shape != (n,) or v.shape != (n,):
                raise ValueError("C must be square and u, v of matching length")
            x = np.empty(n, dtype=np.int32)
            y = np.empty(n, dtype=np.int32)
            lib = _hip.require_device()
            rc = lib.lapwarm_warmstart_lapjv(C.ctypes.data_as(_hip.c_dp), n, u.ctypes.data_as(_hip.c_dp),
                                             v.ctypes.data_as(_hip.c_dp), int(bool(shift_nonneg)),
                                             x.ctypes.data_as(_hip.c_ip), y.ctypes.data_as(_hip.c_ip))
            if _hip.check(rc, "WarmStartLAPSolver") != 0:
                raise RuntimeError(f"lapwarm_warmstart_lapjv failed (code {rc})")
            cols = x.astype(np.int64)
        else:
            import scipy.optimize
            rows, cols = scipy.optimize.linear_sum_assignment(reduce_costs(C, u, v, shift_nonneg=shift_nonneg))
        return rows, cols, float(C[rows, cols].sum())
