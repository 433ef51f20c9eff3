"""Diagnostics (GPU): time the masked-Gram kernel with and without its MFMA drain."""
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
for skip in (0, 1, 0):
    ds.set_option("dbg_skip_drain", skip)
    A = [a.copy(order="F") for a in w.A0]; C = w.C0.copy(order="F")
    ds.optimize(A, C, w.K, w.lam, w.lam, 0.0, max_iter=3, global_tol=-1, seed=1)   # alpha=0: cheap ridge column solve
    pr = ds.profile()
    print(f"{name} skip_drain={skip}: col {pr['col_stats_ms']/pr['col_stats_launches']:.3f} ms  row {pr['row_stats_ms']/pr['row_stats_launches']:.3f} ms", flush=True)
ds.close()
