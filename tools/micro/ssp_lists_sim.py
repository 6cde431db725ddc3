"""How often would candidate lists certify a minima collection of the shortest-path phase?  
Dense JV shortest paths (lapjv.cpp:153-319 semantics, numpy) from row-minimum seeds + greedy on tight edges; per row a
list of the D smallest c - v of each of 64 column classes built ONCE before the first path, tau_i = smallest value left
out.  A collection is 'certified' when its minimum is < LB = min over the heads of the path of
mind_i + tau_i - (c[i][j_i] - v[j_i])  (the path's own row counts with lb = tau_f).  usage: ssp_lists_sim.py n [D] [seed]"""
import sys
import numpy as np

n = int(sys.argv[1]); D = int(sys.argv[2]) if len(sys.argv) > 2 else 2; seed = int(sys.argv[3]) if len(sys.argv) > 3 else 42
C = np.random.RandomState(seed).uniform(0, 1, (n, n))
u = C.min(axis=1); v = (C - u[:, None]).min(axis=0)
x = -np.ones(n, int); y = -np.ones(n, int)
R = C - u[:, None] - v[None, :]
for i in range(n):                      # greedy: first tight column not yet used
    for j in np.nonzero(R[i] <= 1e-12)[0]:
        if y[j] < 0:
            x[i] = j; y[j] = i; break
free = [i for i in range(n) if x[i] < 0]
# lists at the duals the phase starts from
pad = (-n) % 64
Rv = np.pad(C - v[None, :], ((0, 0), (0, pad)), constant_values=np.inf).reshape(n, -1, 64)
tau = np.sort(Rv, axis=1)[:, D, :].min(axis=1)
steps = finds = cert_finds = 0; paths_all_cert = 0
for f in free:
    dist = C[f] - v; pred = np.full(n, f); todo = np.ones(n, bool); ready = []
    LB = tau[f]; all_cert = True
    while True:
        cand = np.where(todo, dist, np.inf); mind = cand.min(); finds += 1
        if mind < LB: cert_finds += 1
        else: all_cert = False
        scan = list(np.nonzero(cand == mind)[0]); todo[scan] = False
        fin = [j for j in scan if y[j] < 0]
        if fin: jf = fin[0]; break
        done = False
        while scan and not done:
            j = scan.pop(0); i = y[j]; ready.append(j); steps += 1
            h = C[i, j] - v[j] - mind
            LB = min(LB, mind + tau[i] - (C[i, j] - v[j]))
            cred = C[i] - v - h
            imp = todo & (cred < dist)
            dist[imp] = cred[imp]; pred[imp] = i
            new = np.nonzero(imp & (cred == mind))[0]
            for jn in new:
                if y[jn] < 0: jf = jn; done = True; break
                scan.append(jn); todo[jn] = False
        if done: break
    paths_all_cert += all_cert
    for j in ready: v[j] += dist[j] - mind
    j = jf
    while True:
        i = pred[j]; y[j] = i; j, x[i] = x[i], j
        if i == f: break
print(f"n={n} D={D} ({64*D} candidates per row): {len(free)} paths, {steps} relax steps, {finds} minima collections, "
      f"certified by the lists {cert_finds} ({100.0*cert_finds/max(finds,1):.1f}%), paths with every collection certified "
      f"{paths_all_cert} ({100.0*paths_all_cert/max(len(free),1):.1f}%)")
