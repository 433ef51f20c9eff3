"""Diagnostics (GPU): quality of the remaining-length estimate of the multi-pass column solve (insider_cd_reg.hpp)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from insider_amd import api, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
w = workloads.make(name)
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
def waste(pred, b):
    perm = np.argsort(-pred, kind="stable")
    pad = (-len(b)) % 4
    bb = np.concatenate([b[perm], np.zeros(pad, dtype=b.dtype)])
    return bb.reshape(-1, 4).max(axis=1).sum() * 4 / max(b.sum(), 1)
for it in range(3):
    for S0 in (64, 256, 512, 1024):
        ds.set_option("cd_cold_iters", 0)
        for k in range(it):   # bring the factors to outer iteration `it` with single-pass solves... (same iterates either way)
            pass
        ds.set_option("cd_cold_iters", it + 1)
        ds.set_option("cd_pass1", S0)
        A0, C0 = workloads.init_factors(w.n_levels, w.K, w.p, workloads.INIT_SEED)
        ds.optimize(A0, C0, w.K, w.lam, w.lam, w.alpha, max_iter=it, global_tol=-1, seed=20240301)
        T = ds.sweeps().astype(np.int64)
        key = ds.debug_array("cd_key0").astype(np.int64)
        un = key > 0
        if un.sum() < 8:
            print(f"{name} it {it} S0 {S0}: unfinished {un.sum()}", flush=True)
            continue
        rem = T[un] - S0
        k = key[un]
        print(f"{name} it {it} S0 {S0}: unfinished {un.sum()} of {w.p}; remaining mean {rem.mean():.0f} max {rem.max()}; "
              f"estimate/actual median {np.median(k / np.maximum(rem, 1)):.2f} p10 {np.percentile(k / np.maximum(rem, 1), 10):.2f} "
              f"p90 {np.percentile(k / np.maximum(rem, 1), 90):.2f}; corr {np.corrcoef(k, rem)[0, 1]:.3f}; unknown(2^20) {np.mean(k >= 1048576):.3f}; "
              f"pass-2 waste by estimate {waste(k, rem):.3f}x, ideal {waste(rem, rem):.3f}x, by T of a static order {waste(np.arange(len(rem))[::-1], rem):.3f}x", flush=True)
ds.close()
