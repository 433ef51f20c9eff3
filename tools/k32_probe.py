"""Diagnostics (GPU): sweep-kernel rate (coordinate updates / s) of the column update at c3's shape for several K."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from insider_amd import api, workloads
for K in [int(k) for k in sys.argv[1:]] or [28, 30, 31, 32]:
    w = workloads.make("c3", K=K)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    ds.set_option("profile", 1)
    A0, C0 = workloads.init_factors(w.n_levels, K, w.p, 7)
    ds.optimize(A0, C0, K, w.lam, w.lam, w.alpha, max_iter=5, global_tol=-1, seed=3)
    pr = ds.profile()
    print(f"K={K}: cd {pr['cd_ms']:.1f} ms over {pr['cd_launches']} solves, {pr['sweeps']} sweeps, {pr['sweeps'] * K / pr['cd_ms'] / 1e6:.1f} G updates/s", flush=True)
    ds.close()
