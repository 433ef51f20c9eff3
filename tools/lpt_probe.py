"""Diagnostics (GPU): how well do the previous iteration's sweep counts predict the next one's (quad packing waste)?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from insider_amd import api, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
w = workloads.make(name)
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
hist = []
for it in range(8, 14):
    A = [a.copy(order="F") for a in w.A0]; C = w.C0.copy(order="F")
    ds.optimize(A, C, w.K, w.lam, w.lam, w.alpha, max_iter=it, global_tol=-1, seed=20240301)
    hist.append(ds.sweeps().astype(np.int64))
def waste(pred, b):
    perm = np.argsort(-pred, kind="stable")
    return b[perm].reshape(-1, 4).max(axis=1).sum() * 4 / b.sum()
ema = hist[0].astype(float)
for i in range(1, len(hist) - 1):
    ema = 0.5 * ema + 0.5 * hist[i]
    print(f"iteration {i}->{i+1}: last {waste(hist[i], hist[i+1]):.3f}  mean of last two {waste(hist[i] + hist[i-1], hist[i+1]):.3f}  ema {waste(ema, hist[i+1]):.3f}"
          f"  max of last two {waste(np.maximum(hist[i], hist[i-1]), hist[i+1]):.3f}", flush=True)
for a, b in zip(hist, hist[1:]):
    perm = np.argsort(-a, kind="stable")
    pred = b[perm].reshape(-1, 4).max(axis=1).sum() * 4
    ideal = np.sort(b)[::-1].reshape(-1, 4).max(axis=1).sum() * 4
    print(f"sum {b.sum()}  quad-max with previous iteration's order {pred} ({pred / b.sum():.3f}x)  ideal order {ideal} ({ideal / b.sum():.3f}x)"
          f"  corr {np.corrcoef(a, b)[0, 1]:.3f}", flush=True)
ds.close()
