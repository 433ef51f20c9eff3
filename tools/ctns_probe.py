"""Diagnostic (GPU): the continuous-covariate path at c3's structure, step by step with timings (prints before every call)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insider_amd import api, workloads
genes = int(sys.argv[1]) if len(sys.argv) > 1 else 192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w = workloads.make("c3", gene_range=(0, genes))
Z = np.asfortranarray(np.random.default_rng(77).standard_normal((w.n, m)))
print("create", flush=True)
t = time.time()
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test, ctns_confounder=Z)
print("created %.2f s" % (time.time() - t), flush=True)
ds.set_option("max_sweeps", 300)
ds.set_option("profile", 1)
A0 = [a.copy(order="F") for a in w.A0] + [np.asfortranarray(np.random.default_rng(9).standard_normal((m, w.K)) * 1e-3)]
for it in (0, 1, 3):
    print("optimize max_iter", it, flush=True)
    t = time.time()
    got = ds.optimize([a.copy(order="F") for a in A0], w.C0.copy(order="F"), w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=it, seed=23,
                      inc_continuous=1)
    print("  done %.2f s, loss %.9g, profile %s" % (time.time() - t, got["loss"], {k: round(v, 2) for k, v in ds.profile().items() if isinstance(v, float)}), flush=True)
ds.close()
print("PROBE_DONE")
