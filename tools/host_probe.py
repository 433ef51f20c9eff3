import time, numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
C = np.asfortranarray(np.random.default_rng(0).standard_normal((30, 50000)))
ts=[]
for _ in range(10):
    t0=time.perf_counter(); D=C.copy(); ts.append(time.perf_counter()-t0)
print("numpy copy of K x p (12 MB): best %.3f ms median %.3f ms" % (1e3*min(ts), 1e3*sorted(ts)[5]))
