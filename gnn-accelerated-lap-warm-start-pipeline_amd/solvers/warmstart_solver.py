"""WarmStartLAPSolver: reduced-cost warm start (reference: solvers/warmstart_solver.py:15-66).
C' = C - u 1^T - 1 v^T is formed on the GPU (reduce_costs), then the cold JV runs on C'."""
from typing import Tuple

import numpy as np
import scipy.optimize

import lap
from .advanced_dual import reduce_costs


class WarmStartLAPSolver:
    def __init__(self, use_lap=True):
        self.name = "WarmStartLAP"
        self.use_lap = use_lap

    def solve(self, C: np.ndarray, u: np.ndarray, v: np.ndarray,
              shift_nonneg: bool = True) -> Tuple[np.ndarray, np.ndarray, float]:
        C = np.asarray(C, dtype=np.float64)
        u = np.asarray(u, dtype=np.float64)
        v = np.asarray(v, dtype=np.float64)
        n = C.shape[0]
        Cprime = reduce_costs(C, u, v, shift_nonneg=shift_nonneg)
        if self.use_lap:
            _, x, _ = lap.lapjv(Cprime, extend_cost=False)
            rows = np.arange(n, dtype=np.int64)
            cols = np.asarray(x, dtype=np.int64)
        else:
            rows, cols = scipy.optimize.linear_sum_assignment(Cprime)
        return rows, cols, float(C[rows, cols].sum())

    def __call__(self, C, u, v):
        return self.solve(C, u, v)
