"""Diagnostics (GPU): how sparse is the gene factor C during a fit? (all-zero genes contribute nothing to the row side)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from insider_amd import api, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
w = workloads.make(name)
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
for it in (0, 2, 10, 30):
    A = [a.copy(order="F") for a in w.A0]; C = w.C0.copy(order="F")
    r = ds.optimize(A, C, w.K, w.lam, w.lam, w.alpha, max_iter=it, global_tol=-1, seed=1)
    Cm = r["column_factor"]
    zc = np.mean(np.all(Cm == 0, axis=0)); ze = np.mean(Cm == 0)
    print(f"{name} after {it+1} iterations: all-zero genes {zc:.3f}, zero entries {ze:.3f}, loss {r['loss']:.6g}", flush=True)
ds.close()
