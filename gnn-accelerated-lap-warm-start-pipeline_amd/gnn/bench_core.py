"""The timed loop of bench.py, factored out so that the multi-rank path is the code under test.

`run_sharded(...)` is what every rank executes: W untimed + K timed steps of a per-rank solve
callable, each followed (when there is more than one rank) by THE one exchange of the sharded
pipeline -- a gather of the int64 assignments to rank 0 -- bracketed by a barrier + device
synchronisation on both sides, MAX over ranks of the elapsed time.  bench.py passes the device
pipeline as the solve callable (backend "nccl" = RCCL over xGMI); tests/test_sharding_gloo.py
passes a CPU stand-in and the "gloo" backend, so the gather / barrier / timing code that runs
under RCCL is exactly the code the CPU test exercises.
"""
from __future__ import annotations

import time
from typing import Callable, Optional

import torch

from .sharding import gather_assignments


def run_sharded(solve_local: Callable[[], dict], steps: int, warmup: int, *, distributed: bool,
                gather_on_host: bool = False, device_sync: Optional[Callable[[], None]] = None,
                sizes=None, after_step: Optional[Callable[[dict], None]] = None):
    """Returns (last_out, elapsed_seconds_max_over_ranks).  `solve_local()` -> dict with "x"
    (assignments of this rank's slice).  On rank 0 the last out carries "x_all" (all ranks)."""
    if distributed:
        import torch.distributed as dist

    def sync():
        if device_sync is not None:
            device_sync()
        if distributed:
            dist.barrier()
            if device_sync is not None:
                device_sync()

    def step():
        out = solve_local()
        if distributed:
            xs = out["x"].cpu() if gather_on_host else out["x"]
            out["x_all"] = gather_assignments(xs, dst=0, sizes=sizes)
        if after_step is not None:
            after_step(out)
        return out

    out = None
    for _ in range(warmup):
        out = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    if distributed:
        dev = out["x"].device if not gather_on_host else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return out, elapsed
