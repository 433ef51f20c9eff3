"""Diagnostics (GPU): how sparse is the column factor during a fit?  Fraction of exactly-zero coefficients after 1, 2, 3, 6, 11, 31
outer iterations of a workload, per gene and per wave of four consecutive genes of the launch order (a coordinate step is a no-op
for a wave only when all four of its genes have the coordinate switched off).   python tools/sparsity_probe.py [c3]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insider_amd import api, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, p, _, _, K, lam, alpha, tuning, f = workloads.CONFIGS[name]
w = workloads.make(name)
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
for iters in (1, 2, 3, 6, 11, 31):
    A0, C0 = workloads.init_factors(w.n_levels, K, p, workloads.INIT_SEED)
    C0 = np.asfortranarray(C0)
    ds.optimize(A0, C0, K, lam, lam, alpha, tuning=tuning, max_iter=iters - 1, global_tol=-1.0, seed=20240301, copy=False)
    Z = (C0 == 0.0)                       # K x p
    perm = ds.debug_array("gene_perm")
    Zp = Z[:, perm[: (p // 4) * 4]].reshape(K, -1, 4)
    wave_zero = Zp.all(axis=2)            # coordinate off in all four genes of a wave of the launch order
    sw = ds.sweeps()
    print(f"{name} after {iters:2d} iterations: zero coefficients {Z.mean():.3f}; all-zero genes {Z.all(axis=0).mean():.3f}; "
          f"(wave, coordinate) pairs off for all four genes {wave_zero.mean():.3f}; weighted by the waves' last sweep counts "
          f"{(wave_zero.mean(axis=0) * sw[perm[: (p // 4) * 4]].reshape(-1, 4).max(axis=1)).sum() / sw[perm[: (p // 4) * 4]].reshape(-1, 4).max(axis=1).sum():.3f}",
          flush=True)
ds.close()
