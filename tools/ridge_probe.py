"""Diagnostics (GPU): time of the alpha = 0 (ridge) column path, as tune()'s rank sweep uses it (R/insider.R:116-136)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from insider_amd import api, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
w = workloads.make(name)
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
ds.set_option("profile", 1)
for it in (1, 10):
    A = [a.copy(order="F") for a in w.A0]; C = w.C0.copy(order="F")
    t0 = time.perf_counter()
    r = ds.optimize(A, C, w.K, 0.1, 0.1, 0.0, max_iter=it, global_tol=-1, seed=1)
    dt = time.perf_counter() - t0
    pr = ds.profile()
    print(f"{name} alpha=0 iters={it+1}: {dt*1e3/(it+1):.2f} ms/iter, col solve {pr['cd_ms']/max(pr['cd_launches'],1):.3f} ms/launch, "
          f"stats {pr['col_stats_ms']/max(pr['col_stats_launches'],1):.3f} ms, loss {r['loss']:.6g}", flush=True)
ds.close()
