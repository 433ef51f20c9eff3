"""Diagnostics (GPU): quad-packing waste of the sweep kernel in the COLD outer iterations (0, 1, 2) of a grid point when
the gene order comes from the same iterations of a neighbouring grid point (what bench.py's warm-up call provides),
from the sum of squares (no history), or from an oracle order (ideal)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from insider_amd import api, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
w = workloads.make(name)
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)

def sweeps_of(iters, lam, init_seed, seed):
    A0, C0 = workloads.init_factors(w.n_levels, w.K, w.p, init_seed)
    ds.optimize(A0, C0, w.K, lam, lam, w.alpha, max_iter=iters, global_tol=-1, seed=seed)
    return ds.sweeps().astype(np.int64)

def waste(pred, b):
    perm = np.argsort(-pred, kind="stable")
    return b[perm].reshape(-1, 4).max(axis=1).sum() * 4 / b.sum()

yy = (w.X.astype(np.float64) ** 2 * (w.M_train != 0)).sum(axis=0)
for it in range(3):
    nb = sweeps_of(it, w.lam - 2.0, workloads.INIT_SEED + 1, 20240302)   # the neighbour (warm-up) grid point
    me = sweeps_of(it, w.lam, workloads.INIT_SEED, 20240301)
    print(f"{name} outer iteration {it}: sweeps mean {me.mean():.0f} max {me.max()} min {me.min()}; waste with neighbour's order "
          f"{waste(nb, me):.3f}x, sum-of-squares order {waste(yy, me):.3f}x, own order (ideal) {waste(me, me):.3f}x, "
          f"corr(neighbour, me) {np.corrcoef(nb, me)[0, 1]:.3f}", flush=True)
ds.close()

# the same call's own history as the predictor
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
hist = [sweeps_of(it, w.lam, workloads.INIT_SEED, 20240301) for it in range(5)]
for a in range(1, 5):
    print(f"{name} outer iteration {a}: waste ordered by this call's iteration {a-1}: {waste(hist[a-1], hist[a]):.3f}x, corr {np.corrcoef(hist[a-1], hist[a])[0,1]:.3f}"
          + (f"; by max of the last two {waste(np.maximum(hist[a-1], hist[a-2]), hist[a]):.3f}x" if a > 1 else ""), flush=True)
ds.close()
