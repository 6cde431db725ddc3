"""`solvers` -- same import surface as the reference's solvers package for the hot path
(/root/reference/solvers/__init__.py:27-58): solver wrappers, dual utilities, timing and the
input generators.  The LAP solves and the dense dual sweeps run on the MI355X through
liblapwarm_hip.so; SciPy is kept only as the external baseline the harness compares against.

Not provided (outside the hot path): LAPMODSolver, compute_oracle_duals,
verification/logging helpers, seed_greedy_matching.
"""
from .scipy_solver import SciPySolver
from .lap_solver import LAPSolver, SeededLAPSolver
from .warmstart_solver import WarmStartLAPSolver
from .timing import time_solver_rigorous
from .advanced_dual import project_feasible, reduce_costs, check_dual_feasible
from .seed_baselines import seed_row_col_minima, seed_noisy_optimal
from .generators import (
    generate_uniform_costs,
    generate_near_diagonal_costs,
    generate_sparse_costs,
    generate_metric_costs,
    generate_clustered_costs,
    generate_noisy_linear_costs,
    generate_worst_case_costs,
    generate_identity_like_costs,
    generate_hard_random_costs,
)

__all__ = [
    "SciPySolver", "LAPSolver", "SeededLAPSolver", "WarmStartLAPSolver", "time_solver_rigorous",
    "project_feasible", "reduce_costs", "check_dual_feasible", "seed_row_col_minima", "seed_noisy_optimal",
    "generate_uniform_costs", "generate_near_diagonal_costs", "generate_sparse_costs",
    "generate_metric_costs", "generate_clustered_costs", "generate_noisy_linear_costs",
    "generate_worst_case_costs", "generate_identity_like_costs", "generate_hard_random_costs",
]
