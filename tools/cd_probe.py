"""Diagnostics (GPU): sweep-count distribution of a workload and CD kernel latency / throughput probes."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from insider_amd import api, workloads, _lib

def hist(name, iters):
    w = workloads.make(name)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    ds.set_option("profile", 1)
    for it in iters:
        A = [a.copy(order="F") for a in w.A0]; C = w.C0.copy(order="F")
        ds.optimize(A, C, w.K, w.lam, w.lam, w.alpha, max_iter=it, global_tol=-1, seed=1)
        sw = ds.sweeps(); pr = ds.profile()
        qs = np.percentile(sw, [0, 10, 25, 50, 75, 90, 99, 99.9, 100])
        print(f"{name} after {it+1} iters: sweeps pct[0,10,25,50,75,90,99,99.9,100]={qs.astype(int).tolist()} mean={sw.mean():.0f} "
              f"frac_cap={np.mean(sw>=10000):.4f} cd_ms/launch={pr['cd_ms']/max(pr['cd_launches'],1):.2f}", flush=True)
    ds.close()

def probe(K=30, sweeps=2000, order_mode=0):
    # identical ill-conditioned problems, tol=0 -> every problem runs exactly `sweeps` sweeps
    rng = np.random.default_rng(0)
    X = rng.standard_normal((500, K)) @ (np.eye(K) + 0.5 * rng.standard_normal((K, K)))
    y = X @ rng.standard_normal(K) + rng.standard_normal(500)
    G, q = X.T @ X, X.T @ y
    lib = _lib.load()
    for B in (2, 4, 2048, 8192, 32768):
        api.strong_coordinate_descent(None, None, np.zeros((B, K)), 5.0, 0.4, np.tile(G, (B, 1, 1)), np.tile(q, (B, 1)),
                                      tol=-1.0, max_sweeps=sweeps, order_mode=order_mode)
        ms = lib.insider_hip_last_cd_ms()
        steps = B * sweeps * K
        print(f"K={K} order_mode={order_mode} B={B}: {ms:.2f} ms, {ms*1e6/(sweeps*K):.1f} ns per wave-step, {steps/ms/1e6:.2f} G gene-steps/s", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "probe":      # python tools/cd_probe.py probe 30 16 ...
        om = int(os.environ.get("PROBE_ORDER_MODE", "0"))
        for k in sys.argv[2:]:
            probe(int(k), order_mode=om)
    else:
        probe(30); probe(20); probe(16); probe(48)
        hist("c2", [0, 2, 10])
