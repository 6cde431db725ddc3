"""Synthetic cost-matrix families (inputs only; no solver logic).

Same entry-point names and argument meaning as the reference's generators
(/root/reference/solvers/generators.py:12-178 and the dataset families in
/root/reference/data/generators.py:33-81) so that harness code written against
`solvers.generate_*` keeps working.  The bodies are our own vectorised NumPy;
bit-identical streams with the reference are kept where its construction is a
plain NumPy draw (uniform, clustered, noisy-linear, hard-random, the structured
ones) and are NOT required elsewhere: both the HIP path and the CPU checker
always consume the same arrays (SURVEY.md section 8(d), K3 note).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

LARGE_FILL = 1e6  # sparse fill value, data/generators.py:68


def generate_uniform_costs(n: int, seed: int = 42) -> np.ndarray:
    """U[0,1) from the legacy global stream (solvers/generators.py:12-24)."""
    return np.random.RandomState(seed).uniform(0.0, 1.0, (n, n)).astype(np.float64)


def generate_near_diagonal_costs(n: int, noise_level: float = 0.1, seed: int = 42) -> np.ndarray:
    rs = np.random.RandomState(seed)
    idx = np.arange(n)
    C = 0.1 + 0.9 * (np.abs(idx[:, None] - idx[None, :]) / n)
    C = C + rs.normal(0, noise_level, (n, n))
    return np.maximum(C, 0.001).astype(np.float64)


def generate_sparse_costs(n: int, sparsity_ratio: float = 0.3, seed: int = 42) -> np.ndarray:
    """Uniform costs with ~(1-sparsity_ratio) of the entries raised to a large value,
    keeping a feasible diagonal-permutation (spirit of solvers/generators.py:60-94)."""
    rs = np.random.RandomState(seed)
    C = rs.uniform(0.0, 1.0, (n, n))
    keep = rs.uniform(size=(n, n)) < sparsity_ratio
    perm = rs.permutation(n)
    keep[np.arange(n), perm] = True
    C[~keep] = LARGE_FILL
    return C.astype(np.float64)


def generate_metric_costs(n: int, seed: int = 42) -> np.ndarray:
    """Euclidean distances between n points in [0,100]^2 (solvers/generators.py:97-110)."""
    pts = np.random.RandomState(seed).uniform(0, 100, (n, 2))
    d = pts[:, None, :] - pts[None, :, :]
    return np.sqrt((d * d).sum(-1)).astype(np.float64)


def generate_clustered_costs(n: int, blocks: int = 4, noise: float = 0.1, seed: int = 42) -> np.ndarray:
    """Block structure with cheaper in-cluster costs (solvers/generators.py:113-123)."""
    rng = np.random.default_rng(seed)
    C = rng.uniform(0.0, 1.0, size=(n, n))
    size = max(1, n // max(1, blocks))
    for b in range(blocks):
        lo = b * size
        hi = n if b == blocks - 1 else min(n, (b + 1) * size)
        C[lo:hi, lo:hi] -= 0.4
    C += noise * rng.normal(0.0, 1.0, size=(n, n))
    return np.maximum(C, 0.0).astype(np.float64)


def generate_noisy_linear_costs(n: int, rank: int = 1, noise: float = 0.1, seed: int = 42) -> np.ndarray:
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, rank))
    b = rng.normal(size=(rank, n))
    C = a @ b + rng.normal(scale=noise, size=(n, n))
    C -= C.min()
    return C.astype(np.float64)


def generate_worst_case_costs(n: int) -> np.ndarray:
    idx = np.arange(n)
    return (np.abs(idx[:, None] - (n - 1 - idx[None, :])) + 1).astype(np.float64)


def generate_identity_like_costs(n: int, diagonal_cost: float = 0.0,
                                 off_diagonal_cost: float = 1.0) -> np.ndarray:
    C = np.full((n, n), off_diagonal_cost, dtype=np.float64)
    np.fill_diagonal(C, diagonal_cost)
    return C


def generate_hard_random_costs(n: int, cost_range: Tuple[float, float] = (0.0, 100.0),
                               seed: int = 42) -> np.ndarray:
    rs = np.random.RandomState(seed)
    lo, hi = cost_range
    C = rs.uniform(lo, hi, (n, n))
    C += rs.uniform(0, (hi - lo) * 0.1, n)[:, None]
    C += rs.uniform(0, (hi - lo) * 0.1, n)[None, :]
    return C.astype(np.float64)


# ---- dataset families (data/generators.py:33-81), keyed like SYNTHETIC_FAMILIES ----

def generate_tie_costs(n: int, bins: int = 5, jitter: float = 1e-6, seed: int = 42) -> np.ndarray:
    rng = np.random.default_rng(seed)
    base = rng.integers(0, max(1, bins), size=(n, n)) / max(1, float(bins))
    return (base + jitter * rng.uniform(0.0, 1.0, size=(n, n))).astype(np.float64)


def generate_low_rank_costs(n: int, rank: int = 12, sigma: float = 0.1, seed: int = 42) -> np.ndarray:
    rng = np.random.default_rng(seed)
    a = rng.normal(0.0, 1.0, size=(n, rank))
    b = rng.normal(0.0, 1.0, size=(n, rank))
    return np.maximum(a @ b.T + sigma * rng.normal(0.0, 1.0, size=(n, n)), 0.0).astype(np.float64)


def generate_dataset_sparse_costs(n: int, sparsity: float = 0.3, seed: int = 42) -> np.ndarray:
    """Uniform costs, ~70% of entries set to 1e6, >=1 finite entry per row/col
    (data/generators.py:56-69)."""
    rng = np.random.default_rng(seed)
    C = np.random.RandomState(int(rng.integers(0, np.iinfo(np.uint32).max))).uniform(0, 1, (n, n))
    keep = rng.random(size=(n, n)) < sparsity
    for i in np.flatnonzero(~keep.any(axis=1)):
        keep[i, rng.integers(0, n)] = True
    for j in np.flatnonzero(~keep.any(axis=0)):
        keep[rng.integers(0, n), j] = True
    C[~keep] = LARGE_FILL
    return C.astype(np.float64)


FAMILIES = {
    "uniform": generate_uniform_costs,
    "metric": generate_metric_costs,
    "low_rank": generate_low_rank_costs,
    "block": generate_clustered_costs,
    "clustered": generate_clustered_costs,
    "noisy_linear": generate_noisy_linear_costs,
    "tie": generate_tie_costs,
    "sparse": generate_dataset_sparse_costs,
}


def generate_family(family: str, n: int, seed: int) -> np.ndarray:
    if family not in FAMILIES:
        raise KeyError(f"Unknown family '{family}'. Known families: {sorted(FAMILIES)}")
    return FAMILIES[family](n, seed=seed)


def mixed_batch(batch: int, n: int, families=("uniform", "sparse", "metric", "clustered"),
                seed: int = 1234) -> Tuple[np.ndarray, list]:
    """K3-style batch: `batch` matrices cycling through `families`, per-instance seeds
    drawn from default_rng(seed)."""
    names, per = _mixed_plan(batch, families, seed)
    out = np.empty((batch, n, n), dtype=np.float64)
    for i, (f, s) in enumerate(zip(names, per)):
        out[i] = generate_family(f, n, s)
    return out, names


def _mixed_plan(batch: int, families, seed: int):
    rng = np.random.default_rng(seed)
    per = [int(s) for s in rng.integers(0, np.iinfo(np.uint32).max, size=batch)]
    names = [families[(i * len(families)) // batch] if batch >= len(families)
             else families[i % len(families)] for i in range(batch)]
    return names, per


def mixed_instances(batch: int, n: int, indices, families=("uniform", "sparse", "metric", "clustered"),
                    seed: int = 1234):
    """Only the instances `indices` of mixed_batch(batch, n, families, seed) -- same matrices,
    without materialising the whole batch (32 MiB each at n = 2048)."""
    names, per = _mixed_plan(batch, families, seed)
    return [generate_family(names[i], n, per[i]) for i in indices], [names[i] for i in indices]
