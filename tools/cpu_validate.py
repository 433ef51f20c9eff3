"""Validates bench.py's CPU-baseline MODEL against real runs of the CPU oracle on this box's host cores (no GPU needed).

bench.py times a bounded sample (G genes, one outer iteration, sweeps capped) and extrapolates: per outer iteration the
reference's formulation costs a fixed time per gene (row update, residual GEMMs, evaluation) plus a time per gene per
coordinate sweep.  Here the same model is checked against
  * c2 (2000 x 20000, K = 20) IN FULL: 31 outer iterations, no sweep cap that bites, the reference's thread counts;
  * c3 (10000 x 50000, K = 30): a 2048-gene slab, outer iterations 0 and 1 from the cold inits, sweeps capped at 3000 per solve
    (a slab this small is badly conditioned — its solves run to 10^5 sweeps uncapped, 17 minutes were not enough — and the
    check is of the cost per gene and sweep, which does not depend on where a solve stops).
Writes one JSON (default gpurun_out/r03/cpu_model_check.json; copy to profiles/r03/): measured wall, the model's prediction
for the same run from an independent sample, and their ratio.    python tools/cpu_validate.py [out.json] [c2|c3|both]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from insider_amd import workloads
from oracle import c_oracle

import threading
_t0 = time.time()


def _heartbeat():      # the GPU box kills a job that prints nothing for seven minutes; the oracle runs are silent for longer
    while True:
        time.sleep(60)
        print(f"[cpu_validate] still running, {time.time() - _t0:.0f} s", flush=True)


threading.Thread(target=_heartbeat, daemon=True).start()
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03", "cpu_model_check.json")
which = sys.argv[2] if len(sys.argv) > 2 else "both"
c_oracle.build()
cores = bench.host_cores()
row_t, col_t = min(10, cores), min(30, cores)
res = {"cpu_model": bench.cpu_model(), "nproc": cores, "row_threads": row_t, "col_threads": col_t,
       "note": "measured = wall clock of the oracle run; model = bench.py's cpu_baseline() sample (its own untimed warm-up, "
               "gene-loop chunk 1, sweeps capped at 120) evaluated at the run's own genes / iterations / sweep count"}


if os.path.exists(out_path):        # a run of one part keeps the other part's record
    try:
        old = json.load(open(out_path))
        for k in ("c2_full", "c3_slab"):
            if k in old:
                res[k] = old[k]
    except ValueError:
        pass


def model_for(name, lam, alpha, genes, iters, total_sweeps):
    m = bench.cpu_baseline(name, lam, alpha, cores, total_sweeps / max(genes * iters, 1), 12.0)["settings"]["reference_threads"]
    per_iter_s = genes * (m["fixed_ms_per_gene"] + m["sweep_ms_per_gene_sweep"] * total_sweeps / (genes * iters)) * 1e-3
    return m, per_iter_s * iters


if which in ("c2", "both"):
    n, p, _, _, K, lam, alpha, tuning, f = workloads.CONFIGS["c2"]
    w = workloads.make("c2")
    t0 = time.perf_counter()
    r = c_oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, lam, lam, alpha, tuning=tuning, max_iter=30,
                          global_tol=-1.0, seed=20240301, row_threads=row_t, col_threads=col_t, max_sweeps=100000)
    wall = time.perf_counter() - t0
    m, pred = model_for("c2", lam, alpha, p, r["iters"], r["total_sweeps"])
    res["c2_full"] = {"shape": [n, p, K], "iterations": r["iters"], "total_sweeps": int(r["total_sweeps"]),
                      "sweeps_per_gene_per_iter": r["total_sweeps"] / (p * r["iters"]), "measured_wall_s": wall,
                      "measured_outer_iterations_per_s": r["iters"] / wall, "phase_seconds": r["phase_seconds"],
                      "model_wall_s": pred, "model_outer_iterations_per_s": r["iters"] / pred, "measured_over_model": wall / pred,
                      "model_parameters": m, "loss": r["loss"], "test_rmse": r["test_rmse"]}
    print("c2 full:", json.dumps(res["c2_full"]), flush=True)
    json.dump(res, open(out_path, "w"), indent=1)

if which in ("c3", "both"):
    n, p, _, _, K, lam, alpha, tuning, f = workloads.CONFIGS["c3"]
    genes = 2048
    w = workloads.make("c3", gene_range=(0, genes))
    c_oracle.set_col_chunk(100)          # the reference's schedule(dynamic, 100): 21 chunks on the column threads
    t0 = time.perf_counter()
    r = c_oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, lam, lam, alpha, tuning=tuning, max_iter=1,
                          global_tol=-1.0, seed=20240301, row_threads=row_t, col_threads=col_t, max_sweeps=3000)
    wall = time.perf_counter() - t0
    m, pred = model_for("c3", lam, alpha, genes, r["iters"], r["total_sweeps"])
    res["c3_slab"] = {"shape": [n, genes, K], "iterations": r["iters"], "total_sweeps": int(r["total_sweeps"]),
                      "sweeps_per_gene_per_iter": r["total_sweeps"] / (genes * r["iters"]), "measured_wall_s": wall,
                      "phase_seconds": r["phase_seconds"], "model_wall_s": pred, "measured_over_model": wall / pred,
                      "model_parameters": m,
                      "note": "a 2048-gene slab is its own problem (its row factors see 2048 genes): the check is of the cost "
                              "model at the run's own sweep count, not of the 50000-gene sweep count.  With the reference's "
                              "schedule(dynamic, 100) the slab is 21 chunks on the column threads: on 16 threads two rounds, the "
                              "second with 5 busy threads, so the balanced model is expected to be low by 32 / 21 = 1.52"}
    print("c3 slab:", json.dumps(res["c3_slab"]), flush=True)
    json.dump(res, open(out_path, "w"), indent=1)
