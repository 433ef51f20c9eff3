"""What do the library's own HIP-event timers (option profile = 1, which bench.py needs for its roofline fields) cost the
timed call?  The default bench workload, the same call with and without them.    python tools/profile_cost.py [workload]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insider_amd import api, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, p, _, _, K, lam, alpha, tuning, f = workloads.CONFIGS[name]
w = workloads.make(name)
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)


def call(iters, seed, init_seed, lam_):
    A0, C0 = workloads.init_factors(w.n_levels, K, p, init_seed)
    ds.sync() if hasattr(ds, "sync") else None
    t0 = time.perf_counter()
    ds.optimize(A0, np.asfortranarray(C0), K, lam_, lam_, alpha, tuning=tuning, max_iter=iters - 1, global_tol=-1.0, seed=seed)
    return time.perf_counter() - t0


for prof in (1, 0, 1, 0):
    ds.set_option("profile", prof)
    call(2, 2, workloads.INIT_SEED + 1, lam - 2.0)
    dt = call(31, 1, workloads.INIT_SEED, lam)
    print(f"profile={prof}: {dt * 1e3 / 31:.3f} ms per outer iteration ({31 / dt:.1f} it/s)", flush=True)
ds.close()
