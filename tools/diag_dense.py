"""Dense sweeps alone on the K3 batch shape (rocprofv3 --kernel-trace --stats gives per-kernel time)."""
import sys
sys.path.insert(0, "gnn-accelerated-lap-warm-start-pipeline_amd")
import torch
from gnn.features import row_features_device

B, n = 32, 2048
fam = sys.argv[1] if len(sys.argv) > 1 else "uniform"
g = torch.Generator(device="cuda").manual_seed(1)
C = torch.rand((B, n, n), dtype=torch.float64, device="cuda", generator=g)
if fam == "sparse":
    keep = torch.rand((B, n, n), device="cuda", generator=g) < 0.3
    C = torch.where(keep, C, torch.full_like(C, 1e6))
elif fam == "ties":
    C = torch.round(C * 8) / 8
for _ in range(2):
    feat, topk = row_features_device(C)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    feat, topk = row_features_device(C)
e1.record()
torch.cuda.synchronize()
print(fam, "row_features_device (colmin + features) ms:", e0.elapsed_time(e1) / 5)
