#!/usr/bin/env python3
"""bench.py — outer-iterations/sec of the INSIDER factorisation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3] [--grid]

A "step" is one outer iteration of the reference's optimize() (src/optimize.cpp:325-410): all covariate row
updates, the column update and the amortised checkpoint work.  The timed region is ONE optimize() call of K outer
iterations from fresh N(0, 1e-6) inits (max_iter = K-1; K = 31 is exactly a tuning_iter = 30 call of tune(),
R/insider.R:163-164) with X, the masks and the level tables already resident in HBM; W warm-up iterations run first
through a separate call at a NEIGHBOURING grid point (other lambda, other inits, other sweep seed), so the timed call
inherits what a real tune() grid point inherits from its predecessor and not a replay of itself.

N = 1: BASELINE config 3 (10000 x 50000, K = 30).  N > 1 (one process per GPU under torchrun): BASELINE config 4
(10000 x 200000, K = 30) STRONG-scaled — the 200000 genes are split N ways, `value` = outer iterations/s of that one
problem; RCCL all-reduces (inside the library, on its stream) of the per-level normal equations per covariate and of the
loss terms per checkpoint.  --workload / --scaling override both.

Prints ONE JSON line on rank 0:
  roofline      the dominant kernel pair = the column step as SURVEY.md 8d prices it (masked Gram/XtY statistics +
                elastic-net sweeps; B_col = 8np + np + 8nK + 16Kp algorithmic bytes) against the 8 TB/s HBM peak,
                average launch durations from HIP events on the library's stream over the timed region
  masked_gram   the statistics kernel alone (BASELINE's metric 2) against the resource that binds it (f64 MFMA for
                the pair-count / look-up / list forms), with its measured HBM bytes
  cd_kernel     the sweep kernel alone
  cpu_baseline  the CPU oracle (reference formulation) on a bounded gene sample, all host threads engaged
  grid          (--grid) BASELINE config 3 as written: tune() over lambda in {1,3,..,19} x alpha in {.2,.3,.4,.5}
"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")   # before anything initialises the HIP runtime (see insider_amd/_lib.py: concurrent fits)
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable by a copy kernel)
FP64_PEAK_TFLOPS = 78.6    # fp64 vector == fp64 matrix (v_mfma_f64_16x16x4) peak on MI355X


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=31)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, help="default: c3 on one GPU, c4 on several")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="N > 1: strong (default) = the workload's p genes split N ways; weak = one slab of p genes per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU work budget of the baseline sample")
    ap.add_argument("--grid", action="store_true", help="also time tune()'s lambda x alpha grid on the resident data set (N = 1)")
    ap.add_argument("--concurrent", type=int, default=1, help="--grid: also time the grid with this many grid points fitted at the same "
                    "time on the one GPU (handles of the shared resident data set, insider_hip_clone)")
    ap.add_argument("--ctns", type=int, default=0, metavar="M", help="add M continuous covariates (N(0, 1) columns of ctns_confounder, "
                    "optimize_continuous_v2, src/optimize.cpp:76-137) to the workload")
    ap.add_argument("--latent", type=int, default=0, metavar="K", help="override the workload's latent dimension (K range study: "
                    "K <= 32 register-resident sweeps, 33..63 one gene per wavefront, > 63 unsupported)")
    ap.add_argument("--seed", type=int, default=20240301)
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="library option (insider_hip_set_option), repeatable")
    return ap.parse_args()


def host_cores():
    """CPU cores this process may actually use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def source_sha():
    """Hash of the sources on disk (insider_amd/_build.py: csrc/, include/, compiler flags).  The library carries the hash
    of the sources it was compiled from; profiles/traffic.json carries the hash of the library its counters were taken on."""
    from insider_amd import _build
    return _build.source_sha()


def cpu_baseline(name, lam, alpha, n_cores, sweeps_per_gene_iter, budget_s):
    """The CPU oracle (the reference's formulation: materialised residual, per-covariate recomputation, cube slices,
    residual-form CD) on a BOUNDED SAMPLE of the same workload, timed on this box's host cores.

    Sample: the first G genes x all samples, one outer iteration, sweeps capped; gene-loop chunk 1 (the reference's
    schedule(dynamic, 100) would put a G < 100 x threads sample on a few threads), so every thread is busy — the
    record carries cpu-time / wall to show it.  Per outer iteration the reference pays a fixed cost per gene (row
    update, the residual GEMMs, evaluation) and a cost per gene PER SWEEP (4 K n_sel flops on the residual); the
    sample's own sweep count is not the workload's (a G-gene problem is not the 50000-gene problem), so the sweep
    phase is scaled by the sweep count the full workload needed (measured on the GPU run; the HIP path and the
    oracle run the same sweeps on the same subproblem, tests/test_gpu_configs.py).  Both thread settings of
    BASELINE.md section 3 are reported, and the labelled covariance-form CPU variant separates algebra from hardware."""
    from insider_amd import workloads
    from oracle import c_oracle
    cn, cp = workloads.CONFIGS[name][0], workloads.CONFIGS[name][1]
    settings = [("reference_threads", min(10, n_cores), min(30, n_cores))]
    if (min(10, n_cores), min(30, n_cores)) != (n_cores, n_cores):
        settings.append(("all_cores", n_cores, n_cores))
    # ~0.7 ms per gene-sweep per core at c3 (4 K n_sel flops, L2-bound): size the sample for the budget
    per_run = budget_s / (len(settings) + 0.25)
    cap = 120
    genes = int(max(4 * n_cores, min(64 * n_cores, per_run * n_cores / (0.7e-3 * cap * (cn / 10000.0)))))
    w = workloads.make(name, gene_range=(0, genes))
    out = {}
    c_oracle.set_col_chunk(1)
    try:
        # untimed pass with one sweep: page in the sample and the oracle's buffers
        c_oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, lam, lam, alpha, tuning=w.tuning,
                          max_iter=0, seed=1, row_threads=n_cores, col_threads=n_cores, max_sweeps=1)
        for label, row_t, col_t in settings + [("covariance_form_variant", n_cores, n_cores)]:
            c_oracle.set_cd_form(1 if label == "covariance_form_variant" else 0)
            t0, c0 = time.perf_counter(), time.process_time()
            res = c_oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, lam, lam, alpha,
                                    tuning=w.tuning, max_iter=0, seed=1, row_threads=row_t, col_threads=col_t,
                                    max_sweeps=cap if label != "covariance_form_variant" else 20 * cap)
            wall, cpu = time.perf_counter() - t0, time.process_time() - c0
            ph = res["phase_seconds"]
            per_gene_fixed = (ph["row"] + ph["residual_eval"]) / genes
            per_gene_sweep = ph["col"] / max(res["total_sweeps"], 1)
            t_iter = cp * (per_gene_fixed + per_gene_sweep * sweeps_per_gene_iter)
            out[label] = {"value": 1.0 / t_iter, "row_threads": row_t, "col_threads": col_t,
                          "fixed_ms_per_gene": per_gene_fixed * 1e3, "sweep_ms_per_gene_sweep": per_gene_sweep * 1e3,
                          "sample_wall_s": wall, "sample_cpu_s": cpu, "cpu_over_wall": cpu / wall,
                          "sample_sweeps": int(res["total_sweeps"])}
        # LIVE check of the model behind the extrapolation, in this run, on this box: the same sample with three times the
        # sweep cap — a different split between the per-gene and the per-sweep cost — timed and compared with what the two
        # parameters fitted above predict for ITS sweep count
        live = None
        try:
            m0 = out["reference_threads"]
            c_oracle.set_cd_form(0)
            t0 = time.perf_counter()
            res = c_oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, lam, lam, alpha, tuning=w.tuning,
                                    max_iter=0, seed=1, row_threads=m0["row_threads"], col_threads=m0["col_threads"], max_sweeps=3 * cap)
            wall = time.perf_counter() - t0
            pred = genes * m0["fixed_ms_per_gene"] * 1e-3 + res["total_sweeps"] * m0["sweep_ms_per_gene_sweep"] * 1e-3
            live = {"sample": f"the same {genes} genes, 1 outer iteration, sweeps capped at {3 * cap}", "sweeps": int(res["total_sweeps"]),
                    "measured_wall_s": wall, "model_wall_s": pred, "measured_over_model": wall / pred}
        except Exception as e:
            live = {"failed": repr(e)}
    finally:
        c_oracle.set_col_chunk(100)
        c_oracle.set_cd_form(0)
    main = out["reference_threads"]
    # (the model behind the extrapolation is checked in THIS run: model_check_live above; full oracle runs of earlier rounds are in
    # profiles/r03/cpu_model_check.json, tools/cpu_validate.py)
    return {"value": main["value"], "unit": "outer-iterations/s", "cores": main["col_threads"], "kind": "port", "model_check_live": live,
            "sample": (f"{name}: first {genes} of {cp} genes x all {cn} samples, 1 outer iteration, sweeps capped at {cap}, "
                       f"gene-loop chunk 1; {main['sample_wall_s']:.1f} s wall, cpu-time/wall {main['cpu_over_wall']:.1f} "
                       f"(row step {main['row_threads']} / column step {main['col_threads']} threads; the reference hard-codes "
                       f"10 / 30, src/optimize.cpp:140,376); fixed {main['fixed_ms_per_gene']:.2f} ms/gene + "
                       f"{main['sweep_ms_per_gene_sweep']:.4f} ms/gene/sweep, scaled to {cp} genes at "
                       f"{sweeps_per_gene_iter:.0f} sweeps/gene/iteration (the full workload's mean, measured on the GPU run)"),
            "cpu_model": cpu_model(), "nproc": n_cores, "omp_proc_bind": os.environ.get("OMP_PROC_BIND", "unset"),
            "omp_places": os.environ.get("OMP_PLACES", "unset"), "settings": out,
            "note": ("covariance_form_variant is NOT the reference's algorithm: the same oracle with covariance-form sweeps "
                     "(2 K^2 instead of 4 K n_sel flops per sweep), reported so that the algebraic part of any GPU/CPU "
                     "ratio can be separated from the hardware part")}


def grid_bench(ds, w, K, steps, warm_start=False, concurrent=1, rank=0, world=1):
    """BASELINE config 3 as written: tune()'s lambda x alpha grid (README.md:79 of the reference: lambda in {1,3,..,19},
    alpha in {0.2,..,0.5}) on the resident data set, tuning_iter = steps - 1, fresh inits per point (R/insider.R:142-174)."""
    from insider_amd import api
    obj = api.Insider()
    obj["data"] = w.X
    obj["confounder"] = w.levels
    obj["inc_continuous"] = 0
    obj["ctns_confounder"] = None
    obj["train_indicator"], obj["test_indicator"] = w.M_train, w.M_test
    obj["params"] = dict(global_tol=-1.0, sub_tol=1e-5, tuning_iter=steps - 1, max_iter=50000)
    obj["seed"] = 7
    obj["_resident_tune"] = ds
    lambdas, alphas = list(range(1, 20, 2)), [0.2, 0.3, 0.4, 0.5]
    timings = []
    import contextlib
    import io
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        res = api.tune(obj, latent_dimension=np.array([K]), lambda_=lambdas, alpha=alphas, timings=timings, warm_start=warm_start,
                       concurrent=concurrent, rank=rank, world=world)
    wall = time.perf_counter() - t0
    clones = obj.get("_tune_clones", [])
    tab = res["reg_tuning"]
    best = tab[int(np.argmin(tab[:, 3]))]
    npts = len(timings)
    return {"points": npts, "lambda": lambdas, "alpha": alphas, "iterations_per_point": steps, "wall_s": wall,
            "warm_start": bool(warm_start), "concurrent": int(concurrent), "table": [[float(v) for v in row] for row in tab],
            "_clones": clones,
            "mean_outer_iterations_per_s": npts * steps / wall,
            "per_point_ms": {"init_draw": 1e3 * float(np.mean([t["init_s"] for t in timings])),
                             "init_draw_not_hidden": 1e3 * float(np.mean([t["init_wait_s"] for t in timings])),
                             "optimize_call": 1e3 * float(np.mean([t["optimize_s"] for t in timings])),
                             "inside_library": float(np.mean([t["library_ms"] for t in timings])),
                             "host_overhead": 1e3 * wall / npts - float(np.mean([t["library_ms"] for t in timings]))},
            "slowest_point_ms": 1e3 * float(max(t["optimize_s"] for t in timings)),
            "fastest_point_ms": 1e3 * float(min(t["optimize_s"] for t in timings)),
            "best": {"lambda": float(best[0]), "alpha": float(best[1]), "test_rmse": float(best[3])},
            "note": ("host_overhead = what the per-point wall time exceeds the time inside the library by: 2 x factor transfer over PCIe, "
                     "Python, and the part of the init draw (numpy, 1.5 M normals, on a helper thread during the previous point's fit) "
                     "that was not hidden; X / masks / lists stay resident")}


def grid_parallel_bench(args, rank, world, local_rank, one_gpu):
    """BASELINE config 3's tune() grid dealt over the ranks (SURVEY.md 8f N1, R/insider.R:142-174): every rank keeps the WHOLE
    c3 data set resident on its GPU and fits the grid points g % world == rank (api.tune(rank, world)); the result tables are
    summed over torch.distributed (RCCL; gloo in the one-GPU rehearsal).  The split of config 3 that has no latency floor:
    the fits are independent, nothing is exchanged inside them."""
    import torch
    import torch.distributed as dist
    from insider_amd import api, workloads
    n, p, _, _, K, lam, alpha, tuning, f = workloads.CONFIGS["c3"]
    t0 = time.perf_counter()
    w = workloads.make("c3")
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test, device=local_rank)
    ds.set_option("profile", 1)
    t_setup = time.perf_counter() - t0
    dist.barrier()
    torch.cuda.synchronize()
    g = grid_bench(ds, w, K, args.steps, concurrent=args.concurrent, rank=rank, world=world)
    torch.cuda.synchronize()
    t = torch.tensor([g["wall_s"]], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    for hd in g.pop("_clones"):
        hd.close()
    ds.close()
    wall = float(t.item())
    tab = np.array(g.pop("table"))
    best = tab[int(np.argmin(tab[:, 3]))]
    return {"workload": f"c3: {n}x{p} fp64, K={K}, 10% held out, the whole data set resident on every GPU", "points": len(tab),
            "points_per_rank": g["points"], "lambda": g["lambda"], "alpha": g["alpha"], "iterations_per_point": args.steps,
            "concurrent_per_gpu": int(args.concurrent), "wall_s_max_over_ranks": wall,
            "mean_outer_iterations_per_s": len(tab) * args.steps / wall,
            "grid_points_per_s": len(tab) / wall, "setup_s_rank0": t_setup,
            "table_complete": bool(np.all(tab[:, 2] > 0)), "table": [[float(v) for v in row] for row in tab], "best": {"lambda": float(best[0]), "alpha": float(best[1]), "test_rmse": float(best[3])},
            "exchange": "one sum-all-reduce of the 40 x 4 result table after the grid (torch.distributed, " + dist.get_backend() + ")",
            "note": "fresh inits of ALL grid points are drawn on every rank in the reference's order: the table does not depend on the rank count"}


def self_launch(n_gpus):
    """Run this script under torch.distributed.run with one rank per GPU, as a child process (nothing in this process has
    initialised the GPU: only argparse / numpy were imported).  Returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # plain `python bench.py --gpus N`: start the N ranks as CHILD processes (one per GPU, torch.distributed.run)
            # before this process has touched the GPU, relay their output and exit with their code
            sys.exit(self_launch(args.gpus))
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}", file=sys.stderr)
        sys.exit(2)
    import torch
    import torch.distributed as dist
    # rehearsal of the N > 1 path on a one-GPU box: every rank on GPU 0, all-reduces staged through the host over gloo
    # (RCCL refuses two ranks on one device); the timing is then not a scaling measurement
    one_gpu = world > 1 and os.environ.get("INSIDER_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from insider_amd import api, dist as idist, workloads

    name = args.workload or ("c3" if world == 1 else "c4")
    scaling = args.scaling or "strong"
    n, p_total, _, _, K, lam, alpha, tuning, f = workloads.CONFIGS[name]
    if args.latent > 0:
        K = args.latent
    weak = scaling == "weak" and world > 1
    slabs = world if weak else 1          # weak scaling: the problem grows with the GPU count
    p_total *= slabs
    lo, hi = idist.shard_range(p_total, rank, world)
    t0 = time.perf_counter()
    w = workloads.make(name, p=p_total, gene_range=(lo, hi), K=K if args.latent > 0 else None)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    Z = None
    if args.ctns > 0:      # continuous covariates: the same N(0, 1) columns on every rank
        Z = np.asfortranarray(np.random.Generator(np.random.PCG64([workloads.DATA_SEED, 7])).standard_normal((n, args.ctns)))
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test, device=local_rank, ctns_confounder=Z)
    t_up = time.perf_counter() - t0
    exchange_vote = None
    if world > 1 and (not one_gpu or os.environ.get("INSIDER_BENCH_REHEARSE_VOTE") == "1"):
        # in-library RCCL (ncclAllReduce on the library's stream) when EVERY rank can join it; else all ranks fall back
        # together — two votes around the blocking join, insider_amd/dist.py:attach_voted.  (One-GPU rehearsal: the vote
        # itself is rehearsed with INSIDER_BENCH_REHEARSE_VOTE=1 + INSIDER_FAIL_COMM_RANK; the fall-back there is the
        # host-staged exchange, RCCL refusing two ranks on one device.)
        exchange, exchange_vote = idist.attach_voted(ds, lo, rank, world, device=local_rank, fallback="staged" if one_gpu else "torch",
                                                     log=lambda m: print(f"[bench rank {rank}] {m}", file=sys.stderr, flush=True))
    else:
        exchange = idist.attach(ds, lo, rank, world, device=local_rank, staged=one_gpu)
    ds.set_option("profile", 1)
    for kv in args.opt:
        k, v = kv.split("=")
        ds.set_option(k, float(v))
    p_loc = hi - lo

    def inits(seed):
        """N(0, 1e-6) inits (R/utils.R:40-43): optimize() updates its factor arguments in place (like the reference)."""
        A0, C0 = workloads.init_factors(w.n_levels, K, p_total, seed)
        if args.ctns > 0:
            A0 = A0 + [np.asfortranarray(np.random.Generator(np.random.PCG64([seed, 9])).normal(0.0, 0.001, size=(args.ctns, K)))]
        return A0, np.asfortranarray(C0[:, lo:hi])

    def run(iters, seed, start, lam_):
        A, C = start
        # copy=False: the factors are updated in place, as the C ABI does it (no second host copy of the result)
        return ds.optimize(A, C, K, lam_, lam_, alpha, tuning=tuning, max_iter=iters - 1, global_tol=-1.0, seed=seed,
                           inc_continuous=1 if args.ctns > 0 else 0, copy=False)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:      # a neighbouring grid point: other lambda, other inits, other sweep-order seed
        run(args.warmup, args.seed + 1, inits(workloads.INIT_SEED + 1), lam - 2.0 if lam > 2.0 else lam + 2.0)
    start = inits(workloads.INIT_SEED)          # host-side start values exist before the clock starts
    sync()
    t0 = time.perf_counter()
    res = run(args.steps, args.seed, start, lam)
    sync()
    dt = time.perf_counter() - t0
    prof = ds.profile()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert res["iters"] == args.steps, (res["iters"], args.steps)

    def copy_bandwidth():
        """Device-to-device copy of 2 GiB (read + write counted), best of 5: what this box's HBM delivers to a plain copy,
        next to the 8 TB/s the roofline is priced against (SURVEY.md 8d asks for both)."""
        try:
            a = torch.empty(1 << 31, dtype=torch.uint8, device="cuda")
            b = torch.empty_like(a)
            best = 0.0
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                b.copy_(a)
                e1.record()
                torch.cuda.synchronize()
                best = max(best, 2.0 * a.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9)
            del a, b
            return best
        except Exception:
            return None

    gp = None
    if world > 1 and args.grid:      # the grid-parallel split of config 3, next to the (unchanged) strong-scaling figure above
        try:
            gp = grid_parallel_bench(args, rank, world, local_rank, one_gpu)
        except Exception as e:
            gp = {"failed": repr(e)}
    if rank == 0:
        gram_ms = prof["col_stats_ms"] / max(prof["col_stats_launches"], 1)
        cd_ms = prof["cd_ms"] / max(prof["cd_launches"], 1)
        col_ms = gram_ms + cd_ms
        T = K * (K + 1) // 2
        # SURVEY.md 8d, per launch over this rank's genes.  Column step with the solve fused: X + uint8 mask + R once + C in / out
        b_col = 8.0 * n * p_loc + 1.0 * n * p_loc + 8.0 * n * K + 2.0 * 8.0 * K * p_loc
        # statistics kernel alone (the record written out): X + mask + R + p (T + K) doubles
        b_gram = 8.0 * n * p_loc + 1.0 * n * p_loc + 8.0 * n * K + 8.0 * p_loc * (T + K)
        fl_alg = 2.0 * f * n * p_loc * T + 2.0 * f * n * p_loc * K
        path = int(ds.info("col_stats_path")) if tuning == 1 else -1
        kern = {2: "k_col_paircnt4 (K <= 31: second product on v_mfma_f64_4x4x4, factor rows in LDS, genes by ticket; else k_col_paircnt), pair-count form (its k_mm_rows2 product, held-out level sums x row factors, forms beside R'R on another stream since round 5 and is not inside this launch time)",
                1: "k_col_factored, look-up form (k_mm_rows product beside R'R on another stream)", 0: "k_list_stats4 / k_list_stats, one rank-one matrix-unit term per held-out entry (v_mfma_f64_4x4x4 tiles for 16 <= K <= 31, else 16x16x4 blocks)",
                -1: "none (tuning = 0: shared R'R, no masked statistics)"}[path]
        mfma_gene = ds.info("col_mfma_per_gene") if tuning == 1 else 0.0
        mfma_tflops = mfma_gene * p_loc * 2048.0 / (gram_ms * 1e-3) / 1e12 if gram_ms > 0 else 0.0
        # measured HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py), valid for the
        # sources they were taken on
        from insider_amd import _lib as ilib
        lib_sha = ilib.library_source_sha()           # what the loaded binary says it was compiled from
        # (PMC figures belong to one problem: a line with --latent / --ctns looks for its own entry, e.g. "c3_K40")
        pmc_key = name + (f"_K{K}" if args.latent > 0 else "") + (f"_ctns{args.ctns}" if args.ctns > 0 else "")
        traffic, tnote = {}, "no profiles/traffic.json entry for this workload"
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(pmc_key, {})
                if ent and world == 1:
                    # valid only for the library it was measured on: compared with the hash the LOADED library reports
                    same = tj.get("source_sha") == lib_sha
                    traffic = ent if same else {}
                    # (the entry's OWN command and commit; files of round 4 carried one top-level pair)
                    tnote = (f"rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE (separate passes, {ent.get('command') or tj.get('command', '?')}); FETCH x 2 per "
                             f"MI355X_MICROARCH.md; taken on source_sha {tj.get('source_sha')} (commit {ent.get('commit') or tj.get('commit', '?')}), "
                             + ("the sources of the library that ran here" if same else
                                f"NOT the library that ran here ({lib_sha}): traffic withheld (null)"))
            except Exception as e:
                tnote = f"profiles/traffic.json unreadable: {e!r}"
        tr_stats = traffic.get("col_stats_bytes_per_launch")
        tr_cd = traffic.get("cd_bytes_per_launch")
        tr_col = (tr_stats + tr_cd) if (tr_stats is not None and tr_cd is not None) else None
        # issue-slot counters of the sweep / statistics kernels (tools/pmc_issue.sh -> profiles/issue.json), same validity rule
        issue, inote = {}, "no profiles/issue.json entry for this workload"
        ipath = os.path.join(ROOT, "profiles", "issue.json")
        if os.path.exists(ipath) and world == 1:
            try:
                ij = json.load(open(ipath))
                if pmc_key in ij:
                    if ij.get("source_sha") == lib_sha:
                        issue = ij[pmc_key]
                        inote = (f"rocprofv3 --pmc SQ_* passes ({issue.get('command') or 'command not recorded for this entry'}; commit "
                                 f"{issue.get('commit') or 'not recorded'}), source_sha {ij.get('source_sha')} = the library that ran here")
                    else:
                        inote = f"profiles/issue.json was taken on source_sha {ij.get('source_sha')}, not on the library that ran here ({lib_sha}): withheld"
            except Exception as e:
                inote = f"profiles/issue.json unreadable: {e!r}"
        st_cd, st_col = ds.info("cd_ms_steady"), (ds.info("col_stats_ms_steady") if tuning == 1 else 0.0)
        cd_updates = prof["sweeps"] * K
        cd_issue = issue.get("sweep_kernel", {})
        cd_clock = 1e9 * cd_issue["clock_GHz"] if cd_issue.get("clock_GHz") else 2.4e9
        out = {
            "metric": f"outer-iterations/sec (masked INSIDER fit, {n} x {p_total}, K={K})",
            "value": slabs * args.steps / dt, "unit": "outer-iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{name}: {n}x{p_total} fp64, K={K}, lambda={lam}, alpha={alpha}, "
                                   f"{int(f * 100)}% held out, tuning={tuning}, levels={list(map(int, w.n_levels))}"
                                   + (f", {args.ctns} continuous covariates" if args.ctns > 0 else ""),
                       "genes_per_gpu": p_loc, "problem_iterations_per_s": args.steps / dt,
                       "value_is": (f"{slabs} x outer iterations/s of the {n}x{p_total} problem (one {name}-sized gene slab per GPU)"
                                    if slabs > 1 else f"outer iterations/s of the {n}x{p_total} problem"),
                       "sub_tol": 1e-5, "global_tol": "off (fixed iteration count)",
                       "timed_call": f"one optimize() of {args.steps} outer iterations from fresh N(0, 1e-6) inits (cold start and the factor transfers included; factors updated in place, as the C ABI does)",
                       "warmup_call": "a neighbouring grid point (other lambda, inits, sweep seed)" if args.warmup > 0 else "none",
                       "parallelism": (f"gene-shard x{world}, exchange: {exchange if isinstance(exchange, str) else type(exchange).__name__}"
                                       + (" (REHEARSAL: all ranks on one GPU, host-staged all-reduce)" if one_gpu else ""))
                                      if world > 1 else "single GPU",
                       "exchange_vote": exchange_vote},
            "roofline": {
                "kernel": ("column step = masked Gram/XtY statistics [" + kern + "] + elastic-net sweeps [k_cd_cols_reg]: the "
                           "dominant kernel pair (SURVEY.md 8d prices them as one 'column-side masked-Gram/XtY (+fused CD)' pass)"),
                "bound": "hbm", "achieved": b_col / (col_ms * 1e-3) / 1e9 if col_ms > 0 else 0.0, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": b_col / (col_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if col_ms > 0 else 0.0,
                "traffic": tr_col, "traffic_note": tnote,
                "algorithmic_bytes_per_launch": b_col, "avg_launch_ms": col_ms,
                "avg_launch_ms_parts": {"statistics": gram_ms, "sweeps": cd_ms}, "launches": prof["cd_launches"],
                "measured_copy_GBs": copy_bandwidth() if world == 1 else None,
                # the same fraction without the cold start (outer iterations >= 5 of the timed call; the first iterations
                # of every call run thousands of sweeps per gene from the N(0, 1e-6) inits)
                "steady_state": ({"avg_launch_ms_parts": {"statistics": st_col, "sweeps": st_cd},
                                  "achieved": b_col / ((st_col + st_cd) * 1e-3) / 1e9,
                                  "frac": b_col / ((st_col + st_cd) * 1e-3) / 1e9 / HBM_PEAK_GBS} if st_cd > 0 else None),
                "note": ("achieved = SURVEY 8d's B_col (8np X + np mask + 8nK R + 16Kp C in/out) / (statistics + sweep kernel time), "
                         "HIP events on the library's stream over the timed call.  The time is dominated by the sweep kernel "
                         "(a sequential K-step recurrence per gene and sweep, hundreds to thousands of sweeps per gene in the "
                         "first outer iterations), not by bytes: measured traffic is far below B_col because neither X nor the "
                         "mask is streamed (held-out lists / dense level-pair counts built once per data set)")},
            "masked_gram": {
                "kernel": kern, "bound": "mfma", "achieved": mfma_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": mfma_tflops / FP64_PEAK_TFLOPS, "avg_launch_ms": gram_ms, "launches": prof["col_stats_launches"],
                "mfma_f64_16x16x4_per_gene": mfma_gene,
                "issue_counters": issue.get("statistics_kernel") or None,
                "traffic": tr_stats,
                "hbm_GBs_measured": tr_stats / (gram_ms * 1e-3) / 1e9 if (tr_stats and gram_ms > 0) else None,
                "hbm_frac_measured": tr_stats / (gram_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if (tr_stats and gram_ms > 0) else None,
                # SURVEY 8d's streaming-formulation bytes over this kernel's time: NOT a roofline fraction (the kernel does
                # not move these bytes); kept because BASELINE's metric 2 is phrased in them
                "algorithmic_equiv_GBs": b_gram / (gram_ms * 1e-3) / 1e9 if gram_ms > 0 else None,
                "algorithmic_equiv_tflops": fl_alg / (gram_ms * 1e-3) / 1e12 if gram_ms > 0 else None,
                "row_update": "merged (per (level, gene) pair)" if prof.get("row_merged") else "per-sample statistics (k_list_stats)",
                "note": ("achieved = executed f64 matrix flops (2048 per v_mfma_f64_16x16x4, 512 per v_mfma_f64_4x4x4) / time against the "
                         "78.6 TF f64 matrix peak; issued back to back from four waves per SIMD the 16x16x4 form sustains 0.60 of that peak on "
                         "this part (matrix unit busy 0.60 at 2.39 GHz), the 4x4x4 form 0.93 - 0.95 (tools/ubench_mfma4.hip); the per-entry "
                         "list form of the same statistics (option col_factored = 0) is the streaming-equivalent kernel")},
            "cd_kernel": {"kernel": ("k_cd_cols_reg (elastic-net coordinate sweeps: 4 genes per wave, Gram matrix in VGPRs, computed-jump dispatch through absolute "
                                      "successor addresses, blocks of two coordinate steps where consecutive coordinates share a slot [option cd_pairs], longest-first gene order)"
                                     if K <= 30 else "k_cd_cols_reg<2, 32> (K = 31, 32: one step per block, 32-bit block offsets)" if K <= 32 else ("k_cd_cols_reg<3, .> (32 < K <= 48: the same kernel with three coordinate slots per lane, the third slot's Gram columns in LDS: 7 VALU + 1 LDS read per step, two waves per SIMD)"
                                                      if K <= 48 else "k_cd_cols<64, 1> (K > 48: one gene per wavefront, Gram matrix in LDS, v_readlane broadcasts)")),
                          "avg_launch_ms": cd_ms, "traffic": tr_cd,
                          "sweeps_per_gene_per_iter": prof["sweeps"] / max(prof["cd_launches"], 1) / p_loc,
                          "coordinate_updates_per_s": cd_updates / max(prof["cd_ms"] * 1e-3, 1e-9),
                          # useful arithmetic: a coordinate update is K fused multiply-adds on the gene's gradient (+ O(1))
                          "fp64_tflops_useful": cd_updates * 2.0 * K / max(prof["cd_ms"] * 1e-3, 1e-9) / 1e12,
                          "fp64_vector_frac": cd_updates * 2.0 * K / max(prof["cd_ms"] * 1e-3, 1e-9) / 1e12 / FP64_PEAK_TFLOPS,
                          # issue model of this design: 6 VALU instructions (4 cycles each) per 4-gene coordinate step and SIMD,
                          # 1024 SIMDs, at the clock the part held during the counter passes (GRBM_GUI_ACTIVE / 8 / kernel time;
                          # 2.4 GHz nominal when no counters are on file): how close the kernel is to ITS ceiling, not a hardware
                          # roofline.  valu_busy_measured = SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles of the kernel (rocprofv3 --pmc)
                          # vector instructions per 4-gene step: 6 (K <= 32), 7 with the third slot (32 < K <= 48); the model does not apply beyond
                          "valu_issue_model_updates_per_s": (1024 * 4 * cd_clock / ((6 if K <= 32 else 7) * 4)) if K <= 48 else None,
                          "valu_issue_model_frac": (cd_updates / max(prof["cd_ms"] * 1e-3, 1e-9) / (1024 * 4 * cd_clock / ((6 if K <= 32 else 7) * 4))) if K <= 48 else None,
                          "valu_issue_model_clock_GHz": cd_clock / 1e9,
                          "valu_busy_measured": cd_issue.get("valu_busy_of_resident_simd_time"),
                          "issue_counters": cd_issue or None, "issue_counters_note": inote,
                          "bound_note": ("taken branches, not instructions: three waves per SIMD (161 VGPRs at K = 30) and a computed jump per block.  A/B on "
                                         "one box in round 5 (profiles/r05/exp): one scalar instruction fewer per step (absolute successor addresses: 9 "
                                         "instead of 10) changes nothing at K = 30 (+2 - 3 % where waves are alone, c1 / c2); a third fewer jumps (blocks of "
                                         "two steps, out of line: ~20 jumps per sweep instead of 30) is 4.5 % fewer sweep-kernel ms at c3 and 6 % at c1; "
                                         "fewer look-ahead touches of the next order row are 5 % slower (DESIGN.md 4.2)"),
                          "share_of_wall": prof["cd_ms"] / (dt * 1e3),
                          # the reference's sweep loop has no cap (src/coordinate_descent.cpp:86-114): solves this call ended
                          # at the library's max_sweeps without convergence (must be 0), and the longest solve
                          "cap_hits": int(ds.info("cap_hits")), "max_gene_sweeps": int(ds.info("max_gene_sweeps")),
                          "max_sweeps": int(ds.info("max_sweeps"))},
            "loss": res["loss"], "train_rmse": res["train_rmse"], "test_rmse": res["test_rmse"], "options": args.opt,
            "library_source_sha": lib_sha, "sources_on_disk_sha": source_sha(),
            "setup_s": {"generate": t_gen, "upload_and_precompute": t_up},
        }
        if world > 1:      # the same problem on ONE GPU, from this repository's own single-GPU run (not measured in this job)
            try:
                sj = json.load(open(os.path.join(ROOT, "profiles", "single_gpu.json")))
                # (the entry measured with THIS command's steps / warm-up when there is one: the cold start weighs differently)
                base = sj.get(f"{name}_s{args.steps}w{args.warmup}") or sj.get(name)
                if base:
                    out["config"]["single_gpu_same_problem"] = base
                proj = json.load(open(os.path.join(ROOT, "profiles", "r05", "scale_projection.json")))
                if name == "c4" and str(world) in proj.get("N", {}):
                    pj = proj["N"][str(world)]["projection"]["assumed"]
                    out["config"]["projected_before_this_run"] = {
                        "value_it_per_s": pj["value_it_per_s"], "speedup_vs_single_gpu": pj["speedup_vs_single_gpu"],
                        "source": ("profiles/r05/scale_projection.json (tools/scale_replay.py: every rank's slab replayed on ONE GPU with the global "
                                   f"level equations, bulk-synchronous sum, all-reduce priced at {pj['alpha_us']} us + bytes / {pj['link_GBs']} GB/s — an "
                                   "assumption, no multi-GPU box in the build pipeline; command --steps 20 --warmup 5)")}
            except Exception:
                pass
        if args.grid and world == 1:
            try:
                out["grid"] = grid_bench(ds, w, K, args.steps)
                out["grid"].pop("_clones")
                if args.concurrent > 1:
                    # k grid points at a time, each on its own handle of the shared resident data set (same tables, bit for bit)
                    gc_ = grid_bench(ds, w, K, args.steps, concurrent=args.concurrent)
                    for hd in gc_.pop("_clones"):
                        hd.close()
                    gc_["identical_to_serial_grid"] = bool(np.array_equal(np.array(gc_.pop("table")), np.array(out["grid"]["table"]), equal_nan=True))
                    gc_["speedup_vs_serial_grid"] = out["grid"]["wall_s"] / gc_["wall_s"]
                    out["grid_concurrent"] = gc_
                # opt-in extension (NOT the reference's behaviour): every grid point starts from its nearest finished neighbour
                gw = grid_bench(ds, w, K, args.steps, warm_start=True)
                gw.pop("_clones")
                cold = np.array(out["grid"]["table"])
                warm = np.array(gw.pop("table"))
                gw["max_abs_test_rmse_difference_vs_cold"] = float(np.max(np.abs(cold[:, 3] - warm[:, 3])))
                gw["best_point_agrees_with_cold"] = bool(np.argmin(cold[:, 3]) == np.argmin(warm[:, 3]))
                out["grid_warm_start"] = gw
            except Exception as e:
                out["grid"] = {"failed": repr(e)}
        if world > 1 and args.grid:
            out["grid_parallel"] = gp
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is an N = 1 figure
            try:
                out["cpu_baseline"] = cpu_baseline(name, lam, alpha, host_cores(),
                                                   out["cd_kernel"]["sweeps_per_gene_per_iter"], args.cpu_seconds)
            except Exception as e:  # the baseline must never take the bench line down
                out["cpu_baseline"] = {"value": None, "unit": "outer-iterations/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    ds.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
