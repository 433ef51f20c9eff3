import sys, os, time
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
ge.build()
from insider_amd import api, workloads
w = workloads.make("c3")
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
ds.set_option("profile", 1)
for seed in (1, 20240301, 2, 3):
    for max_iter in (0, 30):
        for rep in range(2):
            A = [a.copy(order="F") for a in w.A0]; C = w.C0.copy(order="F")
            t0 = time.perf_counter()
            r = ds.optimize(A, C, w.K, w.lam, w.lam, w.alpha, max_iter=max_iter, global_tol=-1, seed=seed)
            dt = time.perf_counter() - t0
            pr = ds.profile()
            print(f"seed {seed} max_iter {max_iter} rep {rep}: {dt*1e3:.2f} ms, cd {pr['cd_ms']:.2f} sweeps {pr['sweeps']} loss {r['loss']:.10g}", flush=True)
