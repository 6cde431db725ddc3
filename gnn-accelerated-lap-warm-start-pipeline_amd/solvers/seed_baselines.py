"""Classical dual seeds for warm-start experiments, on the GPU
(reference: solvers/seed_baselines.py:18-112).

`seed_row_col_minima` is the reference's construction sweep for sweep (row minima, min-trick,
project_feasible) and is bit-identical to it.  `seed_noisy_optimal` keeps the reference's recipe
(optimal duals + Gaussian noise + projection) but takes the optimal duals from the cold JV solve
on the device instead of SciPy + Bellman-Ford (solvers/advanced_dual.py:85-113): any optimal dual
pair is a valid starting point, so the seeds are equivalent in quality, not bit-identical.
`seed_greedy_matching` depends on the O(n^3) difference-constraint solver
(solvers/dual_computation.py:13-74) and is not part of the hot path.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from lap import _hip
from .advanced_dual import project_feasible


def _row_min(C: np.ndarray, v: Optional[np.ndarray] = None) -> np.ndarray:
    C = np.ascontiguousarray(C, dtype=np.float64)
    n = C.shape[0]
    out = np.empty(n, dtype=np.float64)
    lib = _hip.require_device()
    vv = None if v is None else np.ascontiguousarray(v, dtype=np.float64)
    rc = lib.lapwarm_row_min(C.ctypes.data_as(_hip.c_dp), n,
                             vv.ctypes.data_as(_hip.c_dp) if vv is not None else None,
                             out.ctypes.data_as(_hip.c_dp))
    if _hip.check(rc, "row_min") != 0:
        raise RuntimeError(f"row_min failed (code {rc})")
    return out


def _min_trick(C: np.ndarray, u: np.ndarray) -> np.ndarray:
    C = np.ascontiguousarray(C, dtype=np.float64)
    n = C.shape[0]
    uu = np.ascontiguousarray(u, dtype=np.float64)
    out = np.empty(n, dtype=np.float64)
    lib = _hip.require_device()
    rc = lib.lapwarm_min_trick(C.ctypes.data_as(_hip.c_dp), n, uu.ctypes.data_as(_hip.c_dp),
                               out.ctypes.data_as(_hip.c_dp))
    if _hip.check(rc, "min_trick") != 0:
        raise RuntimeError(f"min_trick failed (code {rc})")
    return out


def seed_row_col_minima(C: np.ndarray, *, project_rounds: int = 50):
    """u = row minima, v = min_i (C_ij - u_i), then project_feasible (seed_baselines.py:18-37)."""
    C = np.asarray(C, dtype=np.float64)
    u = _row_min(C)
    v = _min_trick(C, u)
    return project_feasible(C, u, v, max_rounds=project_rounds)


def seed_noisy_optimal(C: np.ndarray, *, noise_std: float = 0.05,
                       rng: Optional[np.random.Generator] = None, project_rounds: int = 75):
    """Optimal duals (from the device cold JV) + N(0, noise_std) noise, re-projected."""
    import torch
    from gnn.one_gnn import OneGNN
    from gnn.pipeline import WarmStartPipeline
    rng = rng or np.random.default_rng()
    C = np.ascontiguousarray(C, dtype=np.float64)
    pipe = WarmStartPipeline(OneGNN(21), "cuda:0")
    _, u, v, ret = pipe.optimal_duals_batch(torch.from_numpy(C).cuda().unsqueeze(0))
    torch.cuda.synchronize()
    if int(ret[0]) != 0:
        raise RuntimeError(f"cold JV failed (code {int(ret[0])})")
    u_opt, v_opt = u[0].cpu().numpy(), v[0].cpu().numpy()
    u_noisy = u_opt + rng.normal(0.0, noise_std, size=u_opt.shape)
    v_noisy = v_opt + rng.normal(0.0, noise_std, size=v_opt.shape)
    return project_feasible(C, u_noisy, v_noisy, max_rounds=project_rounds)


__all__ = ["seed_row_col_minima", "seed_noisy_optimal"]
