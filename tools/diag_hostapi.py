#!/usr/bin/env python3
"""PCIe-inclusive timing of the drop-in host-pointer API (lap.lapjv_seeded / lap.lapjv) with the
reference's methodology (time_solver_rigorous: 5 warm-ups + 30 repeats, median).  Never `value`."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd")]
import numpy as np
import lap
from solvers import SeededLAPSolver, LAPSolver, time_solver_rigorous
from gnn import compute_row_features

for n in (512, 2048, 4096):
    C = np.random.RandomState(42).uniform(0, 1, (n, n))
    u = C.min(1)
    v = (C - u[:, None]).min(0)
    s = SeededLAPSolver()
    t = time_solver_rigorous(lambda: s.solve(C, u, v), 3, 10)
    tf = time_solver_rigorous(lambda: compute_row_features(C), 3, 10)
    print(f"n={n}: lapjv_seeded host API median {t['median']*1e3:.2f} ms (H2D {C.nbytes/2**20:.0f} MiB + solve + D2H); "
          f"compute_row_features host API median {tf['median']*1e3:.2f} ms")
