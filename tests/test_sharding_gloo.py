"""The N>1 path on CPU: two gloo ranks run bench.py's OWN timed loop (gnn/bench_core.run_sharded:
warm-up + timed steps, the single gather of assignments per step, barrier-bracketed timing, MAX
over ranks) with the CPU oracle standing in for the per-rank device pipeline.  The code that runs
under RCCL on the GPU node is the code exercised here; only the solve callable and the backend
differ.  Result must equal the unsharded solve."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd"


def _worker(rank, world, port, total, n, q):
    for p in (str(ROOT), str(PKG)):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gnn.bench_core import run_sharded
    from gnn.sharding import shard_bounds
    from oracle import jv
    from solvers.generators import mixed_batch
    C, _ = mixed_batch(total, n, seed=21)
    lo, hi = shard_bounds(total, world, rank)
    calls = []

    def solve_local():  # stands in for WarmStartPipeline.solve_batch on this rank's slice
        xs = []
        for b in range(lo, hi):
            u = C[b].min(1)
            v = (C[b] - u[:, None]).min(0)
            xs.append(jv.seeded_raw(C[b], u, v)[1])
        calls.append(1)
        return {"x": torch.from_numpy(np.stack(xs))}

    sizes = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
    out, elapsed = run_sharded(solve_local, steps=2, warmup=1, distributed=True, gather_on_host=True, sizes=sizes)
    assert len(calls) == 3 and elapsed > 0.0
    x_all = out.get("x_all")
    assert (x_all is not None) == (rank == 0)
    if rank == 0:
        q.put(x_all.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run(total, n, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_two_rank_gather_equals_unsharded():
    sys.path.insert(0, str(PKG))
    from oracle import jv
    from solvers.generators import mixed_batch
    for total in (4, 5):  # even and ragged slices
        got = _run(total, 48)
        C, _ = mixed_batch(total, 48, seed=21)
        for b in range(total):
            u = C[b].min(1)
            v = (C[b] - u[:, None]).min(0)
            assert np.array_equal(got[b], jv.seeded_raw(C[b], u, v)[1])
