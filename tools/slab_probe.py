"""Diagnostics (GPU): time one rank's share of c3 at world sizes 1, 2, 4, 8 on ONE GPU (gene slab only, no exchange):
what strong scaling could reach if the all-reduce were free."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from insider_amd import api, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
w = workloads.make(name)
for world in (1, 2, 4, 8):
    ps = w.p // world
    X = np.asfortranarray(w.X[:, :ps]); Mt = np.asfortranarray(w.M_train[:, :ps]); Me = np.asfortranarray(w.M_test[:, :ps])
    ds = api.InsiderData(X, w.levels, Mt, Me)
    ds.set_option("profile", 1)
    for max_iter in (30, 1, 0):
        best = 1e9
        for rep in range(3):
            A = [a.copy(order="F") for a in w.A0]; C = w.C0[:, :ps].copy(order="F")
            t0 = time.perf_counter()
            ds.optimize(A, C, w.K, w.lam, w.lam, w.alpha, max_iter=max_iter, global_tol=-1, seed=1)
            best = min(best, time.perf_counter() - t0)
        pr = ds.profile()
        print(f"world {world} genes {ps} max_iter {max_iter}: {best*1e3:.2f} ms total, cd {pr['cd_ms']:.2f} col {pr['col_stats_ms']:.2f} "
              f"row {pr['row_stats_ms']:.2f} wall {pr['wall_ms']:.2f}", flush=True)
    ds.close()
