"""K2-shaped solver launch (64 x 512, optimal duals + min-trick as seeds): wall time, kernel ticks, the serial
(greedy + micro-ARR) part, branches and paths (diagnostic)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd")]
import numpy as np, torch, time
from gnn import OneGNN, WarmStartPipeline
from gnn.features import min_trick_device
B, n = 64, 512
Cs = np.stack([np.random.RandomState(42 + i).uniform(0, 1, (n, n)) for i in range(B)])
C = torch.from_numpy(Cs).cuda()
pipe = WarmStartPipeline(OneGNN(21), "cuda:0")
x, u, v, ret = pipe.optimal_duals_batch(C)
v2 = min_trick_device(C, u)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    xs, ys, rets, st = pipe.seeded_batch(C, u, v2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = st.cpu().numpy()
print("seeded_batch wall %.3f ms; kernel ticks(10ns) mean %.1f max %d; serial part mean %.1f; branches %s; paths %s" % (dt * 1e3, st[:, 13].mean(), st[:, 13].max(), st[:, 14].mean(), np.bincount(st[:, 0].astype(int)).tolist(), st[:, 4].sum()))
