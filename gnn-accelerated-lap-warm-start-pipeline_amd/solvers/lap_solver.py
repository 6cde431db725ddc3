"""Host-array solver objects of the harness, over the MI355X library.

The reference's benchmark scripts construct these by name and call `.solve(...)`
(scripts/gnn_benchmark.py:501-503, :393, :415; keyword form scripts/analyze_all_types_pipeline.py:242).
What they must keep (reference solvers/lap_solver.py:33-105): the `.name` strings, the return
conventions -- `LAPSolver` gives (arange(n), x, cost), `SeededLAPSolver` gives (x, y, cost) -- and
how each cost is summed, because "objective bit-identical" includes the rounding of that sum.
"""
from __future__ import annotations

import numpy as np

import lap


class _HostSolver:
    """Common shape of the harness-facing solvers: a display name and call-through."""

    name = "solver"

    def __call__(self, *args, **kwargs):
        return self.solve(*args, **kwargs)


def _f64(a):
    return np.asarray(a, dtype=np.float64)


class LAPSolver(_HostSolver):
    """Cold Jonker-Volgenant (`lap.lapjv`) on the device."""

    name = "LAP"

    def __init__(self):
        self.name = type(self).name

    def solve(self, C):
        C = _f64(C)
        x = np.asarray(lap.lapjv(C, extend_cost=False)[1], dtype=np.int64)
        # the reference adds the matched costs one by one, left to right, with Python's sum
        # (solvers/lap_solver.py:60): the same association order, written as the loop it is
        total = 0
        for i, j in enumerate(x):
            if j >= 0:
                total = total + C[i, j]
        return np.arange(C.shape[0], dtype=np.int64), x, float(total)


class SeededLAPSolver(_HostSolver):
    """`lap.lapjv_seeded(C, u, v)`: warm start from dual potentials, on the device."""

    name = "SeededLAP"

    def __init__(self):
        self.name = type(self).name
        if not hasattr(lap, "lapjv_seeded"):  # the reference raises ImportError here (lap_solver.py:74-79)
            raise ImportError("lap.lapjv_seeded is not available: build liblapwarm_hip.so "
                              "(make -C gnn-accelerated-lap-warm-start-pipeline_amd/csrc)")

    def solve(self, C, u, v):
        x, y, cost = lap.lapjv_seeded(_f64(C), _f64(u), _f64(v))
        return np.asarray(x, dtype=np.int64), np.asarray(y, dtype=np.int64), float(cost)
