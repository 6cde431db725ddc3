"""time_solver_rigorous: 5 warm-ups (exceptions swallowed) + 30 timed runs, median et al.
(reference: solvers/timing.py:13-58)."""
import statistics
import time
from typing import Callable, Dict


def time_solver_rigorous(solver_func: Callable, num_warmups: int = 5, num_repeats: int = 30) -> Dict[str, float]:
    for _ in range(num_warmups):
        try:
            solver_func()
        except Exception:
            pass
    times = []
    for _ in range(num_repeats):
        t0 = time.perf_counter()
        try:
            solver_func()
        except Exception as exc:
            return {"success": False, "error": str(exc)}
        times.append(time.perf_counter() - t0)
    if not times:
        return {"success": False, "error": "All runs failed"}
    return {
        "success": True,
        "median": statistics.median(times),
        "mean": statistics.mean(times),
        "std": statistics.stdev(times) if len(times) > 1 else 0.0,
        "min": min(times),
        "max": max(times),
        "num_samples": len(times),
    }
