#!/usr/bin/env python3
"""Cooperative shortest-path kernel: correctness + timing probe (diagnostic).
usage: diag_coop.py n B [family] [reps]   (LAPWARM_COOP_MIN_N / LAPWARM_COOP=0 select the path)"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd")]
import numpy as np, torch
from gnn import OneGNN, WarmStartPipeline
from gnn.features import min_trick_device
from oracle import jv
from solvers.generators import mixed_batch

n, B = int(sys.argv[1]), int(sys.argv[2])
fam = sys.argv[3] if len(sys.argv) > 3 else "uniform"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
check = int(os.environ.get("DIAG_CHECK", "2"))
if fam == "uniform":
    Cs = np.stack([np.random.RandomState(42 + i).uniform(0, 1, (n, n)) for i in range(B)])
elif fam == "mixed":
    Cs, names = mixed_batch(B, n, seed=1234)
else:
    Cs, names = mixed_batch(B, n, families=(fam,), seed=1234)
torch.manual_seed(0)
pipe = WarmStartPipeline(OneGNN(21, hidden=64, layers=2).eval(), "cuda:0")
C = torch.from_numpy(Cs).cuda()
if os.environ.get("DIAG_GNN", "0") == "1":      # the pipeline's own seeds: random-init OneGNN H=192 L=4 (bench.py)
    torch.manual_seed(0)
    pipe = WarmStartPipeline(OneGNN(21, hidden=192, layers=4).eval(), "cuda:0")
    u, v = pipe.predict_batch(C)
    u = u.to(torch.float64)
else:
    u = C.min(dim=2).values.contiguous()      # row-minimum seeds
    v = min_trick_device(C, u)
torch.cuda.synchronize()
for rep in range(reps):
    t0 = time.perf_counter()
    x, y, ret, stats = pipe.seeded_batch(C, u, v)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
st = stats.cpu().numpy(); ret = ret.cpu().numpy(); x = x.cpu().numpy()
coop_paths = [int(s & 0xffffffff) if s >= 0 else -1 for s in st[:, 15]]
reason = [int(s >> 32) if s >= 0 else -1 for s in st[:, 15]]
print(f"n={n} B={B} {fam}: {dt*1e3:.2f} ms/batch  ret={sorted(set(ret.tolist()))} err={sorted(set(st[:,12].tolist()))}")
print(f"   paths={st[:,4].tolist()[:8]} coop_paths={coop_paths[:8]} stop_reason={reason[:8]} steps={st[:,6].tolist()[:8]} "
      f"finds={st[:,5].tolist()[:8]} handoffs={(st[:,26] if not st[0,17:23].any() else st[:,26]*0-1).tolist()[:8]} rounds={[int(v & 0xffffffff) for v in st[:,16]][:8]} same_xcd={[int(v >> 40) for v in st[:,16]][:8]}")
if st[0, 17:23].any():  # -DLAPWARM_COOP_STAMPS build: s_memtime ticks of instance 0's leader, summed over the run
    names = ["row wait", "emit", "staging+publish", "poll", "post (no event)", "post (several events)"]
    print("   leader stamps (M ticks over the run): " + ", ".join(f"{nm}={st[0,17+k]/1e6:.0f}" for k, nm in enumerate(names)))
    mx_poll, mx_work = int(st[0, 23]), int(st[0, 24])
    mn_poll, mn_work = (~int(st[0, 25])) & (2**64 - 1), (~int(st[0, 26])) & (2**64 - 1)
    print(f"   members of instance 0 (M ticks over the run): poll min {mn_poll/1e6:.0f} max {mx_poll/1e6:.0f}; "
          f"stamped rest of a relax round min {mn_work/1e6:.0f} max {mx_work/1e6:.0f}")
tot_steps = st[:, 6].max()
print(f"   slowest instance: {tot_steps} relax steps, {dt*1e6/max(1,tot_steps):.3f} us per step (whole batch time / max steps)")
un, vn = u.cpu().numpy(), v.cpu().numpy()
for b in range(min(check, B)):
    t0 = time.perf_counter()
    r, xo, yo, so = jv.seeded_raw(Cs[b], un[b], vn[b])
    ok = bool(r == ret[b] and (r != 0 or np.array_equal(xo, x[b])))
    cnt_ok = all(st[b, q] == so[k] for q, k in ((4, "paths"), (5, "finds"), (6, "scan_steps"), (7, "scan_elems")))
    print(f"   oracle b={b}: {time.perf_counter()-t0:.2f} s exact={ok} counters_equal={cnt_ok}", flush=True)
