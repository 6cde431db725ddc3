#!/usr/bin/env python3
"""Diagnostic: the DMA search variant at small sizes (threads_hint forces two columns per thread)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd")]
import numpy as np, torch
from gnn import OneGNN, WarmStartPipeline
from oracle import jv
for n, hint in ((2048, 512), (2048, 1024), (2048, 1024), (1536, 768), (1664, 832), (1920, 960)):
    pipe = WarmStartPipeline(OneGNN(21), "cuda:0", threads_hint=hint)
    B = 3
    Cs = np.stack([np.random.RandomState(100 + i).uniform(0, 1, (n, n)) for i in range(B)])
    us = np.stack([C.min(1) for C in Cs]); vs = np.stack([(C - u[:, None]).min(0) for C, u in zip(Cs, us)])
    x, y, ret, st = pipe.seeded_batch(torch.from_numpy(Cs).cuda(), torch.from_numpy(us).cuda(), torch.from_numpy(vs).cuda())
    torch.cuda.synchronize()
    x = x.cpu().numpy(); st = st.cpu().numpy(); ret = ret.cpu().numpy()
    bad = 0
    for b in range(B):
        r, xo, yo, so = jv.seeded_raw(Cs[b], us[b], vs[b])
        ok = np.array_equal(xo, x[b])
        bad += not ok
        if not ok or b == 0:
            print(f"  n={n} hint={hint} b={b} ret={ret[b]} exact={ok} steps gpu/oracle {st[b,6]}/{so['scan_steps']} finds {st[b,5]}/{so['finds']} paths {st[b,4]}/{so['paths']}")
    print(f"n={n} hint={hint}: {bad}/{B} mismatches")
