"""NumPy restatement of the reference's row features and dual utilities -- TEST INFRASTRUCTURE.

Checker for the HIP kernels; never imported by the product package.  Pinned by
tests/golden/features_*.npz (generated from the reference by
tests/golden/make_golden.py).

Restates (paths relative to /root/reference):
  gnn/features.py:21-31     positional encodings
  gnn/features.py:161-243   compute_row_features (fp64 statistics -> float32)
  scripts/gnn_benchmark.py:262   the min-trick  v_j = min_i (C_ij - u_i)
  solvers/advanced_dual.py:14-36 project_feasible
  solvers/advanced_dual.py:39-53 reduce_costs
  solvers/advanced_dual.py:56-63 check_dual_feasible
  solvers/seed_baselines.py:18-37 seed_row_col_minima
"""
from __future__ import annotations

import numpy as np

POS_FREQS = (1, 2, 4, 8)  # gnn/features.py:16
EPS = 1e-9                # gnn/features.py:18
N_STATS = 13
ROW_FEATURE_DIM = N_STATS + 2 * len(POS_FREQS)


def positional_encodings(n: int) -> np.ndarray:
    """(n, 8) float32: sin/cos(2*pi*i*f / max(1, n-1)), f in POS_FREQS, interleaved."""
    if n <= 0:
        return np.zeros((0, 2 * len(POS_FREQS)), dtype=np.float32)
    idx = np.arange(n, dtype=np.float64)
    denom = max(1, n - 1)
    out = np.empty((n, 2 * len(POS_FREQS)), dtype=np.float64)
    for c, f in enumerate(POS_FREQS):
        ang = 2.0 * np.pi * idx * f / denom
        out[:, 2 * c] = np.sin(ang)
        out[:, 2 * c + 1] = np.cos(ang)
    return out.astype(np.float32)


def row_statistics(C: np.ndarray, col_min: np.ndarray | None = None) -> np.ndarray:
    """The 13 data-dependent columns, in fp64 (n, 13), before the float32 cast.

    `col_min`: column minima of the FULL matrix when C is only a subset of its rows (large-n
    spot checks); default: the minima of C itself, as in the reference."""
    C = np.asarray(C, dtype=np.float64)
    n, m = C.shape
    lo = C.min(axis=1)
    hi = C.max(axis=1)
    mean = C.mean(axis=1)
    std = C.std(axis=1)  # population (ddof=0): features.py:173
    med = np.median(C, axis=1)
    mad = np.median(np.abs(C - med[:, None]), axis=1)
    mad = np.where(mad < EPS, EPS, mad)

    e = np.exp(-(C - lo[:, None]))
    p = e / (e.sum(axis=1, keepdims=True) + EPS)
    entropy = -(p * np.log(p + EPS)).sum(axis=1)

    srt = np.sort(C, axis=1)
    if m >= 2:
        gap = srt[:, 1] - srt[:, 0]
        competition = gap / ((srt[:, -1] - srt[:, 0]) + EPS)
        difficulty = 1.0 / (np.diff(srt, axis=1).mean(axis=1) + EPS)
    else:
        gap = np.zeros(n)
        competition = np.zeros(n)
        difficulty = np.zeros(n)
    k = min(10, m)
    knear = srt[:, :k]
    k_mean = knear.mean(axis=1)
    k_std = knear.std(axis=1)

    near_best = (C <= lo[:, None] * 1.1).sum(axis=1) / max(1, m)
    if col_min is None:
        col_min = C.min(axis=0)
    col_best = (C == col_min).sum(axis=1) / max(1, m)
    return np.stack([lo, hi, mean, std, mad, entropy, gap, competition, k_mean, k_std,
                     difficulty, near_best, col_best], axis=1)


def compute_row_features(C: np.ndarray) -> np.ndarray:
    """(n, 21) float32, see gnn/features.py:161-243.  n == 0 -> shape (0, 0)."""
    C = np.asarray(C, dtype=np.float64)
    n = C.shape[0]
    if n == 0:
        return np.zeros((0, 0), dtype=np.float32)
    return np.concatenate([row_statistics(C), positional_encodings(n).astype(np.float64)],
                          axis=1).astype(np.float32)


def topk_smallest(C: np.ndarray, k: int = 16) -> np.ndarray:
    """(n, min(k, m)) ascending fp64: what OneGNN's refinement consumes (values only)."""
    C = np.asarray(C, dtype=np.float64)
    k = min(k, C.shape[1])
    return np.sort(C, axis=1)[:, :k]


def min_trick(C: np.ndarray, u: np.ndarray) -> np.ndarray:
    """v_j = min_i (C_ij - u_i) in fp64 (u may be float32; it is widened exactly)."""
    C = np.asarray(C, dtype=np.float64)
    return np.min(C - np.asarray(u)[:, None], axis=0).astype(np.float64)


def project_feasible(C, u, v, max_rounds: int = 50, tol: float = 1e-12):
    C = np.asarray(C, dtype=float)
    u = np.array(u, dtype=float)
    v = np.array(v, dtype=float)
    for _ in range(max(1, int(max_rounds))):
        u = np.minimum(u, (C - v[None, :]).min(axis=1))
        v = np.minimum(v, (C - u[:, None]).min(axis=0))
        if ((C - u[:, None]) - v[None, :]).min() >= -tol:
            break
    return u, v


def reduce_costs(C, u, v, shift_nonneg: bool = True) -> np.ndarray:
    C = np.asarray(C, dtype=float)
    R = (C - np.asarray(u)[:, None]) - np.asarray(v)[None, :]
    if shift_nonneg:
        lo = R.min()
        if lo < 0:
            R = R - lo
    return np.ascontiguousarray(R, dtype=np.float64)


def seed_row_col_minima(C, project_rounds: int = 50):
    """solvers/seed_baselines.py:18-37."""
    C = np.asarray(C, dtype=np.float64)
    u = C.min(axis=1).copy()
    v = (C - u[:, None]).min(axis=0)
    return project_feasible(C, u, v, max_rounds=project_rounds)


def check_dual_feasible(C, u, v, tol: float = 1e-8) -> bool:
    lo = float(((np.asarray(C) - np.asarray(u)[:, None]) - np.asarray(v)[None, :]).min())
    if lo < -tol:
        raise AssertionError(f"Dual infeasible: min reduced cost {lo:.3e} < -tol")
    return True
