#!/usr/bin/env python3
"""Per-instance anatomy of the K3 batch on the GPU (diagnostic, not a bench)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd")]
import numpy as np, torch
from gnn import OneGNN, WarmStartPipeline
from gnn.features import row_features_device, min_trick_device
from solvers.generators import mixed_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
hints = [int(a) for a in sys.argv[3:]] or [0]
Cs, fams = mixed_batch(B, n, seed=1234)
torch.manual_seed(0)
model = OneGNN(21, hidden=192, layers=4).eval()
C = torch.from_numpy(Cs).cuda()

def timed(fn, reps=3):
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return out, min(ts) * 1e3

pipe = WarmStartPipeline(model, "cuda:0")
(feat, topk), t_feat = timed(lambda: row_features_device(C))
mask = torch.ones(feat.shape[:2], dtype=torch.bool, device="cuda")
with torch.inference_mode():
    u, t_gnn = timed(lambda: pipe.model(feat, mask=mask, topk_values=topk)["u"])
v, t_v = timed(lambda: min_trick_device(C, u))
print(f"stage ms: features {t_feat:.2f}  onegnn {t_gnn:.2f}  min-trick {t_v:.2f}")
for h in hints:
    pipe.threads_hint = h
    (x, y, ret, st), t_s = timed(lambda: pipe.seeded_batch(C, u, v))
    st = st.cpu().numpy()
    print(f"--- threads_hint={h}: seeded batch {t_s:.2f} ms")
    print("  b family     br  free paths finds  steps   arr_it  total_ms prelude_ms")
    for b in range(B):
        print(f"  {b:2d} {fams[b]:9s} {st[b,0]:2d} {st[b,2]:5d} {st[b,4]:5d} {st[b,5]:5d} {st[b,6]:6d} {st[b,11]:8d} {st[b,13]/1e5:9.2f} {st[b,14]/1e5:9.2f}")
(xc, yc, rc, stc), t_c = timed(lambda: pipe.lapjv_batch(C), reps=1)
stc = stc.cpu().numpy()
print(f"--- cold lapjv batch {t_c:.2f} ms; per-instance ms by family:")
for f in sorted(set(fams)):
    idx = [b for b in range(B) if fams[b] == f]
    print(f"  {f:9s} mean {np.mean(stc[idx,13])/1e5:8.2f} max {np.max(stc[idx,13])/1e5:8.2f} arr_it {np.mean(stc[idx,11]):9.0f} paths {np.mean(stc[idx,4]):6.0f} steps {np.mean(stc[idx,6]):8.0f}")
