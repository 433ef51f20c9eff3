"""Diagnostics (GPU): a gene slab of c4's first p/N genes as a problem of ITS OWN, 11 outer iterations (bench.py's c4 command),
against the whole problem's time per iteration (profiles/single_gpu.json).  NOT the sharded run: the slab's row factors are
estimated from its own genes, so its sweep counts are another problem's (round 4's 3.98 x "before any exchange cost" came from
this).  What a rank of the N-GPU job really does — global row factors, true sweep counts — is tools/scale_replay.py; this probe
remains for the anatomy of a mid-size launch (steady-state iteration, tail of the sweep counts)."""
import sys, os, time
import json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WHOLE_MS = json.load(open(os.path.join(ROOT, "profiles", "single_gpu.json")))["c4"]["ms_per_step"]   # the whole problem on one GPU
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
    print(f"cd_split={split} c4 / {N}: {p // N} genes: {dt / 11 * 1e3:.3f} ms per outer iteration ({11 / dt:.1f} it/s); whole problem / N ({WHOLE_MS:.2f} ms per iteration on one GPU, profiles/single_gpu.json): {WHOLE_MS / N:.3f} ms; "
          f"cd {pr['cd_ms'] / 11:.3f} ms, statistics {pr['col_stats_ms'] / 11:.3f} ms per iteration", flush=True)
    ds.close()
