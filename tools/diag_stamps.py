#!/usr/bin/env python3
"""Cycle anatomy of the shortest-path loop from the -DLAPWARM_STAMPS build (diagnostic)."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
os.environ["LAPWARM_HIP_LIB"] = os.environ.get("LAPWARM_STAMP_LIB", str(ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd" / "liblapwarm_hip_stamps.so"))
sys.path[:0] = [str(ROOT), str(ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd")]
import numpy as np, torch
from gnn import OneGNN, WarmStartPipeline
from solvers.generators import mixed_batch
B, n = int(os.environ.get("DIAG_B", 32)), int(os.environ.get("DIAG_N", 2048))
hint = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if n > 2048:  # large-n anatomy: uniform instances only
    Cs = np.stack([np.random.RandomState(42 + i).uniform(0, 1, (n, n)) for i in range(B)])
    fams = ["uniform"] * B
else:
    Cs, fams = mixed_batch(B, n, seed=1234)
torch.manual_seed(0)
pipe = WarmStartPipeline(OneGNN(21, hidden=192, layers=4).eval(), "cuda:0", threads_hint=hint)
C = torch.from_numpy(Cs).cuda()
u, v = pipe.predict_batch(C)
for _ in range(2):
    x, y, ret, st = pipe.seeded_batch(C, u, v)
torch.cuda.synchronize()
st = st.cpu().numpy()
print('ret nonzero:', int((ret != 0).sum()), 'err slots:', st[:,12].tolist()[:8])
names = ["find", "relax:issue loads", "relax:wait+compute+publish", "relax:barrier", "relax:post", "n cnt0", "n cnt1", "n slow", "path end"]
for f in ("uniform", "sparse", "clustered"):
    idx = [b for b in range(B) if fams[b] == f]
    if not idx:
        continue
    s = st[idx].mean(0)
    tot_ms = s[13] / 1e5
    print(f"{f}: kernel {tot_ms:.1f} ms, paths {s[4]:.0f} finds {s[5]:.0f} steps {s[6]:.0f}")
    cyc = s[16:25]
    total_cyc = cyc[0] + cyc[1] + cyc[2] + cyc[3] + cyc[4] + cyc[8]
    for k, nm in enumerate(names):
        if nm.startswith("n "):
            print(f"   {nm:28s} {cyc[k]:10.0f}")
        else:
            per = cyc[k] / (s[5] if k == 0 else (s[4] if k == 8 else s[6]))
            print(f"   {nm:28s} {cyc[k]/1e6:8.2f} Mcycles  ({100*cyc[k]/total_cyc:5.1f}%)  {per:8.0f} cycles each")
    # column-owned search (r02): 9 scatter+B1, 10 scan..B2, 11 classify..B3, 12 apply/replay..B4+relabel, 13 path end
    fn = ["find:scatter+B1", "find:scan..B2", "find:classify..B3", "find:apply..relabel"]
    for k, nm in enumerate(fn):
        print(f"      {nm:24s} {s[16+9+k]/s[5]:8.0f} cycles per find")
    print(f"      path end (dump+barrier)  {s[16+8]/s[4]:8.0f} cycles per path")
    c0, c1, c2 = max(cyc[5], 1), max(cyc[6], 1), max(cyc[7], 1)
    post2 = s[16+4] - s[16+13] - s[16+14]
    print(f"      post | cnt0 {s[16+13]/c0:7.0f}  cnt1 {s[16+14]/c1:7.0f}  slow path (free/multi) {post2/c2:7.0f} cycles each")
    print(f"      tie finds / finds        {s[16+15]/s[5]:8.3f}")
    if os.environ.get("DIAG_BARRIER"):
        print(f"      barrier wait after a 0-event step: {s[16+9]/max(s[16+10],1):7.0f} cycles ({s[16+10]:.0f} steps); "
              f"after a 1-event step: {s[16+11]/max(s[16+12],1):7.0f} cycles ({s[16+12]:.0f} steps); all steps {s[16+3]/s[6]:7.0f}")
    print(f"   stamped total {total_cyc/1e6:.1f} Mcycles -> {total_cyc/ (tot_ms*1e-3) / 1e9:.2f} GHz-equivalent")
