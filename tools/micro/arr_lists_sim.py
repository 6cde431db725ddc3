"""Simulation of the candidate-list rule of the cold row reduction (jv_solver.hip: cold_arr_sweep): share of
iterations that would need the full row scan, per cost family, for 1-3 candidates per column class.
usage: [REBUILD=1] arr_lists_sim.py n family[,family...]"""
import sys, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd")]
from solvers.generators import mixed_batch
LARGE = 1e6
import os
REBUILD = os.environ.get('REBUILD', '0') == '1'
def sim(C, Ds):
    n = C.shape[0]
    v = C.min(axis=0).copy(); ya = C.argmin(axis=0)
    x = -np.ones(n, int); y = ya.copy(); uq = np.ones(n, bool)
    for j in range(n - 1, -1, -1):
        i = y[j]
        if x[i] < 0: x[i] = j
        else: uq[i] = False; y[j] = -1
    fr = []
    for i in range(n):
        if x[i] < 0: fr.append(i)
        elif uq[i]:
            j = x[i]; c = C[i] - v; c[j] = np.inf; v[j] -= min(c.min(), LARGE)
    R = C - v[None, :]
    pad = (-n) % 64
    Rp = np.pad(R, ((0, 0), (0, pad)), constant_values=np.inf).reshape(n, -1, 64)   # [row][k][lane]
    srt = np.sort(Rp, axis=1)
    taus, inl = {}, {}
    for D in Ds:
        taus[D] = srt[:, D, :].min(axis=1) if srt.shape[1] > D else np.full(n, np.inf)
        thr = srt[:, D - 1, :]
        inl[D] = (Rp <= thr[:, None, :]).reshape(n, -1)[:, :n]
    nf = len(fr); fr = fr + [0] * (n - nf)
    tot = 0; fb = {D: 0 for D in Ds}
    for sweep in range(2):
        if nf == 0: break
        cur = rr = 0; newf = 0
        while cur < nf:
            rr += 1; fi = fr[cur]; cur += 1
            c = C[fi] - v
            j1 = int(np.argmin(c)); v1 = c[j1]; c2 = c.copy(); c2[j1] = np.inf; j2 = int(np.argmin(c2)); v2 = c2[j2]
            tot += 1
            for D in Ds:
                if not (inl[D][fi, j1] and inl[D][fi, j2] and v2 < taus[D][fi]):
                    fb[D] += 1
                    if REBUILD:
                        rp = np.pad(c, (0, pad), constant_values=np.inf).reshape(-1, 64)
                        sr = np.sort(rp, axis=0)
                        taus[D][fi] = sr[D].min() if sr.shape[0] > D else np.inf
                        inl[D][fi] = (rp <= sr[D - 1][None, :]).reshape(-1)[:n]
            i0 = y[j1]; vn = v[j1] - (v2 - v1); low = vn < v[j1]
            if rr < cur * n:
                if low: v[j1] = vn
                elif i0 >= 0: j1 = j2; i0 = y[j2]
                if i0 >= 0:
                    if low: cur -= 1; fr[cur] = i0
                    else: fr[newf] = i0; newf += 1
            elif i0 >= 0: fr[newf] = i0; newf += 1
            x[fi] = j1; y[j1] = fi
        nf = newf
    return tot, fb
n = int(sys.argv[1]); fams = sys.argv[2].split(",")
for fam in fams:
    Cs, names = mixed_batch(1, n, families=(fam,), seed=5)
    tot, fb = sim(Cs[0], (1, 2, 3))
    print(f"{fam:12s} n={n}: ARR iterations {tot}, full-scan fallbacks " + ", ".join(f"D={D}: {fb[D]} ({100.0*fb[D]/max(tot,1):.2f}%)" for D in fb), flush=True)
