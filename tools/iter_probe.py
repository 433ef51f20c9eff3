"""Diagnostics (GPU): per OUTER ITERATION of a cold call, how far is the sweep kernel's time from (i) perfect packing at the
kernel's throughput and (ii) the latency of the longest gene alone?  Runs the same call with max_iter = 0, 1, 2, ... (same
inits and seed, so call t repeats calls 0 .. t-1 and adds one iteration), differences the solve times of consecutive calls and
reads the per-gene sweep counts of the last iteration.          python tools/iter_probe.py [c3|c1|c2|slab] [last_iter]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from insider_amd import api, workloads
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
last = int(sys.argv[2]) if len(sys.argv) > 2 else 10
if which == "slab":
    p = workloads.CONFIGS["c4"][1]
    w = workloads.make("c4", gene_range=(0, p // 8))
    A0, C0 = workloads.init_factors(w.n_levels, w.K, p, 7)
    C0 = np.asfortranarray(C0[:, : p // 8])
else:
    w = workloads.make(which)
    A0, C0 = w.A0, w.C0
tuning = workloads.CONFIGS["c4" if which == "slab" else which][7]
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
ds.set_option("profile", 1)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    ds.set_option(k, float(v))
P, K = C0.shape[1], w.K
prev_cd = prev_col = prev_row = 0.0
rows = []
for it in range(last + 1):
    for rep in range(2):      # second run of the same call: warm caches / clocks
        ds.optimize([a.copy(order="F") for a in A0], C0.copy(order="F"), K, w.lam, w.lam, w.alpha, tuning=tuning, max_iter=it,
                    global_tol=-1, seed=1)
    pr = ds.profile()
    sw = ds.sweeps().astype(np.int64)
    d_cd = pr["cd_ms"] - prev_cd
    rows.append((it, d_cd, sw.sum(), sw.mean(), np.median(sw), np.percentile(sw, 99), sw.max(), pr["col_stats_ms"] - prev_col,
                 pr["wall_ms"]))
    prev_cd, prev_col = pr["cd_ms"], pr["col_stats_ms"]
# throughput of the kernel = the best any iteration reaches
rate = max(r[2] * K / (r[1] * 1e-3) for r in rows if r[1] > 0)
print(f"{which}: p = {P}, K = {K}; best rate {rate:.3e} coordinate updates/s; lone-wave step taken as 33 ns")
print("iter  solve_ms  sweeps/gene  median   p99    max   ideal_ms(throughput)  longest_alone_ms  solve/max(ideal,alone)  stats_ms")
for it, d_cd, tot, mean, med, p99, mx, d_col, wall in rows:
    ideal = tot * K / rate * 1e3
    alone = mx * K * 33e-6
    print(f"{it:4d} {d_cd:9.3f} {mean:11.1f} {med:7.0f} {p99:6.0f} {mx:6d} {ideal:12.3f} {alone:18.3f} {d_cd / max(ideal, alone, 1e-9):14.2f} {d_col:16.3f}")
ds.close()
