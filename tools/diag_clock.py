#!/usr/bin/env python3
"""Does the chip hold a low clock while only a few waves run?  Same coop solve with and without a
GEMM loop keeping other CUs busy on a second stream (diagnostic only)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd")]
import numpy as np, torch
from gnn import OneGNN, WarmStartPipeline
from gnn.features import min_trick_device
n, B = int(sys.argv[1]), int(sys.argv[2])
Cs = np.stack([np.random.RandomState(42 + i).uniform(0, 1, (n, n)) for i in range(B)])
pipe = WarmStartPipeline(OneGNN(21, hidden=64, layers=2).eval(), "cuda:0")
C = torch.from_numpy(Cs).cuda(); u = C.min(dim=2).values.contiguous(); v = min_trick_device(C, u)
A = torch.randn(4096, 4096, device="cuda", dtype=torch.float32)
side = torch.cuda.Stream()
def run(heat):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, y, ret, st = pipe.seeded_batch(C, u, v)
    if heat:
        with torch.cuda.stream(side):
            for _ in range(heat):
                A2 = A @ A
    torch.cuda.current_stream().synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt
run(0)
for heat in (0, 0, 200, 1000, 0):
    print(f"n={n} B={B} heater_gemms={heat}: solve {run(heat)*1e3:.1f} ms", flush=True)
