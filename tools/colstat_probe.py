"""Diagnostics (GPU): time of the column-side statistics kernels of a workload with each form forced."""
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
for mode in (1, 0, 2, 3, 1):
    ds.set_option("col_factored", mode)
    A = [a.copy(order="F") for a in w.A0]; C = w.C0.copy(order="F")
    r = ds.optimize(A, C, w.K, w.lam, w.lam, w.alpha, max_iter=5, global_tol=-1, seed=1)
    pr = ds.profile()
    print(f"{name} col_factored={mode}: factored {pr['col_factored']} pair {pr['col_pair']} col stats {pr['col_stats_ms']/pr['col_stats_launches']:.3f} ms per launch, "
          f"loss {r['loss']:.10g}", flush=True)
ds.close()
