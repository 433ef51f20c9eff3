"""Diagnostics (GPU): how predictable is the TAIL of the per-gene sweep counts of a steady-state column solve?
The split column step (option cd_split) only pays if the genes that will be longest in the NEXT solve are known; this prints, for
a c4 / 8 slab (or c3), the overlap between the genes that were longest in outer iteration t + 1 and the genes the launch order
ranks first after iteration t (the smoothed sweep counts).    python tools/tail_probe.py [c3|slab]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from insider_amd import api, workloads
which = sys.argv[1] if len(sys.argv) > 1 else "slab"
if which == "slab":
    p = workloads.CONFIGS["c4"][1]
    w = workloads.make("c4", gene_range=(0, p // 8))
    A0, C0 = workloads.init_factors(w.n_levels, w.K, p, 7)
    C0 = np.asfortranarray(C0[:, : p // 8])
else:
    w = workloads.make("c3")
    A0, C0 = w.A0, w.C0
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
ds.set_option("cd_split", 0)
sw = {}
for it in (7, 8, 9, 10):
    ds.optimize([a.copy(order="F") for a in A0], C0.copy(order="F"), w.K, w.lam, w.lam, w.alpha, max_iter=it, global_tol=-1, seed=1)
    sw[it] = ds.sweeps().astype(np.int64)
    perm = ds.debug_array("gene_perm")          # the launch order made after this solve (for iteration it + 1)
    sw[(it, "perm")] = perm
P = len(sw[7])
for it in (8, 9):
    nxt = sw[it + 1]
    order = sw[(it, "perm")]
    rank = np.empty(P, dtype=np.int64); rank[order] = np.arange(P)
    top = np.argsort(-nxt)
    print(f"{which}: iteration {it + 1}: sweeps mean {nxt.mean():.0f} median {np.median(nxt):.0f} p99 {np.percentile(nxt, 99):.0f} "
          f"p99.9 {np.percentile(nxt, 99.9):.0f} max {nxt.max()}; corr(sweeps_t, sweeps_t+1) = {np.corrcoef(sw[it], nxt)[0, 1]:.2f}")
    for k in (10, 50, 250):
        longest = top[:k]
        for frac in (0.03, 0.10, 0.25):
            hit = np.mean(rank[longest] < frac * P)
            print(f"   of the {k:4d} longest genes, {100 * hit:5.1f} % were among the first {100 * frac:.0f} % of the launch order")
    rest = nxt[rank >= 0.10 * P]
    print(f"   longest gene outside the first 10 % of the launch order: {rest.max()} sweeps (overall max {nxt.max()})")
ds.close()
