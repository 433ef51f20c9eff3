"""Diagnostics (GPU): per-iteration sweep-count distribution and CD launch time of a workload."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from insider_amd import api, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
w = workloads.make(name)
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
ds.set_option("profile", 1)
prev_ms = 0.0
for it in range(0, 6):
    A = [a.copy(order="F") for a in w.A0]; C = w.C0.copy(order="F")
    ds.optimize(A, C, w.K, w.lam, w.lam, w.alpha, max_iter=it, global_tol=-1, seed=20240301)
    sw = ds.sweeps().astype(np.int64); pr = ds.profile()
    qs = np.percentile(sw, [0, 10, 50, 90, 99, 99.9, 100]).astype(int).tolist()
    quads = sw[np.argsort(-sw)]  # ideal grouping
    ideal = quads.reshape(-1, 4).max(axis=1).sum() * 4 if len(sw) % 4 == 0 else -1
    nat = sw.reshape(-1, 4).max(axis=1).sum() * 4 if len(sw) % 4 == 0 else -1
    print(f"{name} iteration {it}: sweeps pct[0,10,50,90,99,99.9,100]={qs} mean={sw.mean():.0f} sum={sw.sum()} "
          f"quad-max sum: sorted {ideal} natural {nat}; cd total ms {pr['cd_ms']:.2f} (this iteration ~{pr['cd_ms']-prev_ms:.2f})", flush=True)
    prev_ms = pr['cd_ms']
ds.close()
