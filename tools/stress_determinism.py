#!/usr/bin/env python3
"""Run-to-run determinism of the product build under perturbed timing (diagnostic).
Repeats the K3 seeded batch; on odd repetitions a second stream streams through HBM so that wave
arrival times differ.  Any race in the solver shows as a differing assignment / ret code."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd")]
import numpy as np, torch
from gnn import OneGNN, WarmStartPipeline
from solvers.generators import mixed_batch
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B, n = 32, 2048
Cs, fams = mixed_batch(B, n, seed=1234)
torch.manual_seed(0)
pipe = WarmStartPipeline(OneGNN(21, hidden=192, layers=4).eval(), "cuda:0")
C = torch.from_numpy(Cs).cuda()
u, v = pipe.predict_batch(C)
side = torch.cuda.Stream()
junk = torch.empty((64, 1024, 1024), device="cuda")
x0 = None
bad = 0
for r in range(reps):
    if r % 2 == 1:
        with torch.cuda.stream(side):
            for _ in range(20):
                junk.mul_(1.0001)
    x, y, ret, st = pipe.seeded_batch(C, u, v)
    torch.cuda.synchronize()
    if r and r % 500 == 0:
        print(f"  progress: {r} repetitions, {bad} bad", flush=True)
    xr, rr = x.cpu().numpy(), ret.cpu().numpy()
    if x0 is None:
        x0 = xr
    same = np.array_equal(xr, x0)
    if (rr != 0).any() or not same:
        bad += 1
        print(f"rep {r}: ret nonzero {int((rr != 0).sum())} same_as_first={same} err={st[:,12].cpu().numpy().tolist()}")
print(f"determinism: {reps} repetitions, {bad} bad")
