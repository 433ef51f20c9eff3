"""Diagnostics (GPU): would ordering the genes by their sum of squares pack the FIRST column solve of a data set well?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from insider_amd import api, workloads
w = workloads.make(sys.argv[1] if len(sys.argv) > 1 else "c3")
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
A = [a.copy(order="F") for a in w.A0]; C = w.C0.copy(order="F")
ds.optimize(A, C, w.K, w.lam, w.lam, w.alpha, max_iter=0, global_tol=-1, seed=1)
sw = ds.sweeps().astype(np.int64)
yy = np.einsum("ij,ij->j", w.X, w.X)
def waste(order):
    s = sw[order]
    n = len(s) // 4 * 4
    return s[:n].reshape(-1, 4).max(axis=1).sum() * 4 / s[:n].sum() - 1
print("natural order: waste %.3f" % waste(np.arange(len(sw))))
print("by actual sweeps: waste %.3f" % waste(np.argsort(-sw)))
print("by sum of squares: waste %.3f" % waste(np.argsort(-yy)), "corr", np.corrcoef(sw, yy)[0, 1])
ds.close()
