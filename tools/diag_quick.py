import sys, os
sys.path[:0]=['/root/repo','/root/repo/gnn-accelerated-lap-warm-start-pipeline_amd']
import numpy as np
import lap
from oracle import jv
n=2048
for seed in (42,43):
    C=np.random.RandomState(seed).uniform(0,1,(n,n)); u=C.min(1); v=(C-u[:,None]).min(0)
    x,y,cost=lap.lapjv_seeded(C,u,v)
    r,xo,yo,st=jv.seeded_raw(C,u,v)
    print(os.environ.get("LAPWARM_HIP_LIB","default")[-20:], seed, "exact" if np.array_equal(x,xo) else "MISMATCH", "perm" if sorted(x.tolist())==list(range(n)) else "notperm", abs(cost-float(C[np.arange(n),xo].sum())))
