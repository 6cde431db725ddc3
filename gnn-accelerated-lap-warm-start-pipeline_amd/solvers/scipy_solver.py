"""SciPy baseline wrapper (reference: solvers/scipy_solver.py:13-34). External baseline only."""
from typing import Tuple

import numpy as np
import scipy.optimize


class SciPySolver:
    def __init__(self):
        self.name = "SciPy"

    def solve(self, C: np.ndarray) -> Tuple[np.ndarray, np.ndarray, float]:
        C = np.asarray(C, dtype=np.float64)
        rows, cols = scipy.optimize.linear_sum_assignment(C)
        return rows, cols, float(C[rows, cols].sum())

    __call__ = solve
