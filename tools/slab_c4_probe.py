"""Diagnostics (GPU): one rank's share of the strong-scaled c4 problem on ONE GPU: a gene slab of c4's first p/N genes, 11 outer
iterations (bench.py's c4 command), against the whole problem's 16.5 ms per iteration: what strong scaling can reach before any
exchange cost (the slab's own row factors differ from the full problem's, the timing is representative)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from insider_amd import api, workloads
n, p = workloads.CONFIGS["c4"][0], workloads.CONFIGS["c4"][1]
args = sys.argv[1:]
split = None
if "--split" in args:      # option cd_split of the library (0 never, 2 always; default: what a sharded handle of this size takes)
    i = args.index("--split")
    split = int(args[i + 1])
    del args[i:i + 2]
for N in [int(v) for v in args] or [8, 4, 2]:
    w = workloads.make("c4", gene_range=(0, p // N))
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    ds.set_option("profile", 1)
    if split is not None:
        ds.set_option("cd_split", split)
    def run(iters, seed, init_seed, lam):
        A0, C0 = workloads.init_factors(w.n_levels, w.K, p, init_seed)
        C0 = np.asfortranarray(C0[:, : p // N])
        t0 = time.perf_counter()
        r = ds.optimize(A0, C0, w.K, lam, lam, w.alpha, max_iter=iters - 1, global_tol=-1, seed=seed)
        return time.perf_counter() - t0
    run(1, 2, 8, 3.0)
    dt = run(11, 1, 7, w.lam)
    pr = ds.profile()
    print(f"cd_split={split} c4 / {N}: {p // N} genes: {dt / 11 * 1e3:.3f} ms per outer iteration ({11 / dt:.1f} it/s); ideal from the whole problem: {16.5 / N:.3f} ms; "
          f"cd {pr['cd_ms'] / 11:.3f} ms, statistics {pr['col_stats_ms'] / 11:.3f} ms per iteration", flush=True)
    ds.close()
