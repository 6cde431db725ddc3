#!/usr/bin/env python3
"""Cold lapjv (ARR-dominated): timing + parity probe (diagnostic).
usage: diag_cold.py n B [family|mixed|uniform|int100|int9] [check] [threads]   (LAPWARM_ARR_LISTS=0: plain row scans)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd")]
import numpy as np, torch
from gnn import OneGNN, WarmStartPipeline
from oracle import jv
from solvers.generators import mixed_batch

n, B = int(sys.argv[1]), int(sys.argv[2])
fam = sys.argv[3] if len(sys.argv) > 3 else "uniform"
check = int(sys.argv[4]) if len(sys.argv) > 4 else 2
if fam == "uniform":
    Cs = np.stack([np.random.RandomState(42 + i).uniform(0, 1, (n, n)) for i in range(B)])
    names = ["uniform"] * B
elif fam in ("int100", "int9"):
    Cs = np.stack([np.random.RandomState(5 + i).randint(1, int(fam[3:]) + 1, (n, n)).astype(np.float64) for i in range(B)])
    names = [fam] * B
elif fam == "mixed":
    Cs, names = mixed_batch(B, n, families=("uniform", "sparse", "metric", "clustered", "low_rank", "noisy_linear", "tie"), seed=77)
else:
    Cs, names = mixed_batch(B, n, families=(fam,), seed=77)
C = torch.from_numpy(Cs).cuda()
hint = int(sys.argv[5]) if len(sys.argv) > 5 else 0   # workgroup size of the solver (0 = the library's choice)
pipe = WarmStartPipeline(OneGNN(21), "cuda:0", threads_hint=hint)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, y, ret, st = pipe.lapjv_batch(C)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = st.cpu().numpy(); x = x.cpu().numpy(); y = y.cpu().numpy(); ret = ret.cpu().numpy()
print(f"n={n} B={B} {fam}: batch {dt*1e3:.1f} ms   ret={sorted(set(ret.tolist()))}")
for b in range(B):
    it, fast, ms = st[b, 11], st[b, 27], st[b, 13] / 1e5
    line = f"  [{b}] {names[b]:12s} {ms:8.1f} ms  ARR iterations {it} (from lists {fast}, {100.0*fast/max(it,1):.1f}%)  paths {st[b,4]}"
    if b < check:
        t0 = time.perf_counter()
        r, xo, yo, so = jv.dense_raw(Cs[b])
        ok = r == ret[b] and np.array_equal(xo, x[b]) and np.array_equal(yo, y[b])
        cnt = so["arr_iters"] == it and so["paths"] == st[b, 4] and so["scan_steps"] == st[b, 6]
        line += f"   oracle {1e3*(time.perf_counter()-t0):.0f} ms exact={bool(ok)} counters_equal={bool(cnt)}"
    print(line, flush=True)
