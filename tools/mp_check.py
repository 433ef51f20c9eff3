import sys, numpy as np
sys.path.insert(0, '/root/repo')
from insider_amd import api, workloads
for K in list(range(2, 33)):
    w = workloads.small(K=K, n=90, p=75, seed=K, f=0.2)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    out = {}
    for mode, p1 in (("single", 0), ("multi", 64), ("multi32", 32)):
        ds.set_option("cd_pass1", p1)
        ds.set_option("cd_cold_iters", 100)
        try:
            r = ds.optimize([a.copy(order="F") for a in w.A0], w.C0.copy(order="F"), w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=2, seed=3, sub_tol=1e-9)
            out[mode] = (r["column_factor"], ds.sweeps().copy())
        except Exception as e:
            out[mode] = repr(e)[:80]
    s = out["single"]
    line = f"K {K}: single max sweeps {s[1].max() if not isinstance(s, str) else s}"
    for m in ("multi", "multi32"):
        v = out[m]
        if isinstance(v, str): line += f" | {m}: {v}"
        else: line += f" | {m}: identical={np.array_equal(v[0], s[0])} nan={np.isnan(v[0]).any()}"
    print(line, flush=True)
    ds.close()
