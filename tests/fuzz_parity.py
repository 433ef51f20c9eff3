"""Randomised parity sweep (GPU): random small workloads (shapes, covariate structures, masks, NA entries, lambda / alpha,
masked / unmasked, every form of the statistics kernels, every CD variant) through insider_hip_optimize against the
CPU oracle.  Prints every case that disagrees; exit code 1 when any does.      python tests/fuzz_parity.py [cases] [seed]

A case is a plain dict (draw_case) so that one found by a long sweep can be frozen as a regression input
(tests/golden/fuzz_outliers.json, tests/test_gpu_fuzz.py): FUZZ_DUMP=<file> appends every mismatching case as a JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# the alternative kernel forms a case is re-run under (tools/fuzz_repro.sh; test_fuzz_outliers_*): per-entry list statistics
# on both sides, the group and the LDS-resident CD kernels, single-pass solves, the look-up form of the column statistics
FORMS = ({}, {"col_factored": 0, "row_merged": 0}, {"cd_variant": 1}, {"cd_variant": 2}, {"cd_pass1": 0},
         {"row_counts": 0, "col_factored": 2})


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def draw_case(rng):
    """The next case of the stream (None: the draw is not a valid data set — a level that never occurs)."""
    from insider_amd import workloads
    c = int(rng.integers(1, 5))
    levels = tuple(int(x) for x in rng.integers(1, 13, size=c))
    if max(levels) == 1:
        levels = levels[:-1] + (3,)
    if rng.random() < 0.2:   # a many-level covariate: the GEMM form of the level Gram sums (k_wgemm) needs >= 49 levels
        levels = (int(rng.choice([49, 64, 97, 130])),) + levels[1:]
    n = int(rng.integers(max(16, max(levels) * 2), max(260, max(levels) * 3)))
    p = int(rng.integers(5, 140))
    kchoice = [1, 2, 3, 5, 8, 13, 15, 16, 17, 20, 23, 25, 30, 31, 32, 33, 40, 47, 48, 63]
    if os.environ.get("FUZZ_K"):   # e.g. FUZZ_K=33,36,37,41,44,45,47: a sweep over chosen instantiations
        kchoice = [int(k) for k in os.environ["FUZZ_K"].split(",")]
    K = int(rng.choice(kchoice))
    tuning = int(rng.random() < 0.8)
    kw = dict(n=n, p=p, level_counts=levels, K=K, f=float(rng.uniform(0.03, 0.6)), lam=float(rng.choice([0.3, 1.0, 2.0, 7.0])),
              alpha=float(rng.choice([0.0, 0.1, 0.4, 0.8, 1.0])), tuning=tuning, seed=int(rng.integers(1, 10 ** 6)),
              with_na=bool(rng.random() < 0.3))
    if c >= 2 and rng.random() < 0.25:
        kw["interaction_idx"] = (1, 2)
    opts = dict(row_merged=int(rng.choice([0, 1, 2])), col_factored=int(rng.choice([0, 1, 2, 3])),
                cd_variant=int(rng.choice([0, 0, 0, 1, 2])), row_counts=int(rng.integers(0, 2)), row_gemm=int(rng.integers(0, 2)),
                # multi-pass column solves: first limit (0 = single pass), growth ratio, how many outer iterations use them
                cd_pass1=int(rng.choice([0, 32, 48, 64])), cd_pass_ratio=int(rng.choice([2, 3, 4])),
                cd_cold_iters=int(rng.choice([1, 3, 9])))
    sub_tol = float(rng.choice([1e-5, 1e-5, 1e-8, 1e-11]))   # tight tolerances: hundreds of sweeps, so that passes really split solves
    iters = int(rng.choice([0, 1, 3]))
    seed = int(rng.integers(1, 1000))
    m = int(rng.choice([0, 0, 0, 1, 3]))   # continuous covariates (optimize_continuous_v2)
    if m:
        sub_tol = 1e-5   # their scalar CD stops on sum |du| < 0.1 (src/optimize.cpp:122): a stopping rule at rounding level upstream
                         # (sub_tol 1e-11) flips its pass count on degenerate data (n < K) and the comparison means nothing
    try:
        workloads.small(**kw)
    except AssertionError:
        return None
    scale = float(rng.choice([0.001, 0.3]))
    return dict(kw=kw, opts=opts, sub_tol=sub_tol, iters=iters, seed=seed, m=m, scale=scale)


def tolerances(case):
    tol = 1e-6 if case["m"] else 1e-7   # the continuous update solves an m x m system whose conditioning the data sets
    tol_traj = 1e-8
    if os.environ.get("FUZZ_TIGHT"):   # hunting mode: the suite's base tolerances for every case (finds the rounding-level outliers
        return tol, tol_traj           # that tests/golden/fuzz_outliers.json freezes)
    if case["sub_tol"] < 1e-9:      # the stopping rule |dloss| <= 1e-11 on losses of 1e3 is decided at rounding level: a sweep more
        tol = 5e-6                  # or less on either side moves beta by ~sqrt(tol / D).  (A 3000-case sweep in round 3 had 3 cases
        tol_traj = 1e-7             # of this regime just outside 1e-6 / 1e-8: factors 1.2e-6, trajectories 1.4e-8 and 1.6e-8.)
    return tol, tol_traj


def inputs(case):
    """(workload, A0 list, C0, Z) of a case: everything is a function of the case dict."""
    from insider_amd import workloads
    kw = dict(case["kw"])
    kw["level_counts"] = tuple(kw["level_counts"])
    if "interaction_idx" in kw:
        kw["interaction_idx"] = tuple(kw["interaction_idx"])
    w = workloads.small(**kw)
    rs = np.random.default_rng(case["seed"])
    A = [np.asfortranarray(rs.standard_normal(a.shape) * case["scale"]) for a in w.A0]
    C = np.asfortranarray(rs.standard_normal(w.C0.shape) * case["scale"])
    Z = None
    if case["m"]:
        Z = np.asfortranarray(rs.standard_normal((w.n, case["m"])))
        A = A + [np.asfortranarray(rs.standard_normal((case["m"], w.K)) * case["scale"])]
    return w, A, C, Z


def run_hip(case, extra=None, iters=None, want_sweeps=False):
    """The HIP fit of a case under its own options + `extra`; (result or None, exception or None[, per-gene sweeps])."""
    from insider_amd import api
    w, A, C, Z = inputs(case)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test, ctns_confounder=Z)
    opts = dict(case["opts"])
    opts.update(extra or {})
    for k, v in opts.items():
        ds.set_option(k, v)
    ds.set_option("max_sweeps", 300)
    sw = None
    try:
        got = ds.optimize([a.copy(order="F") for a in A], C.copy(order="F"), w.K, w.lam, w.lam, w.alpha, tuning=w.tuning,
                          max_iter=case["iters"] if iters is None else iters, seed=case["seed"],
                          inc_continuous=1 if case["m"] else 0, sub_tol=case["sub_tol"])
        err = None
        if want_sweeps:
            sw = ds.sweeps()
    except Exception as e:   # both sides must then fail
        got, err = None, e
    ds.close()
    return (got, err, sw) if want_sweeps else (got, err)


def run_oracle(case, iters=None, want_sweeps=False, cd_form=0):
    """cd_form = 0: the parity oracle (the reference's residual-form CD).  cd_form = 1: the SAME oracle with covariance-form
    sweeps (oracle_set_cd_form; a diagnostic, never the checker): what separates the formulation's rounding from a kernel's."""
    from oracle import c_oracle   # test infrastructure: the checker
    w, A, C, Z = inputs(case)
    sink = c_oracle.set_sweep_sink(w.p) if want_sweeps else None
    c_oracle.set_cd_form(cd_form)
    try:
        ref = c_oracle.optimize(w.X, w.levels, w.n_levels, A, C, w.M_train, w.M_test, w.lam, w.lam, w.alpha, tuning=w.tuning,
                                max_iter=case["iters"] if iters is None else iters, seed=case["seed"], max_sweeps=300,
                                sub_tol=case["sub_tol"], **(dict(ctns=Z) if case["m"] else {}))
        rerr = None
    except Exception as e:
        ref, rerr = None, e
    finally:
        c_oracle.set_cd_form(0)
        if want_sweeps:
            c_oracle.set_sweep_sink(None)
    return (ref, rerr, None if sink is None else sink.copy()) if want_sweeps else (ref, rerr)


def errors(got, ref):
    """(row, column, trajectory) relative deviations of a HIP result from the oracle's."""
    rm = ref["row_matrices"]
    rm = [rm[f"factor{i}"] for i in range(len(rm))] if isinstance(rm, dict) else rm     # (another HIP result as the reference)
    e_row = max(relerr(got["row_matrices"][f"factor{i}"], a) for i, a in enumerate(rm))
    e_col = relerr(got["column_factor"], ref["column_factor"])
    tg, tr = got["traj"][:, 1:8], ref["traj"][:, 1:8]
    e_traj = (float(np.nanmax(np.abs(tg - tr) / np.maximum(np.abs(tr), 1e-300))) if tg.shape == tr.shape and tg.size
              else (0.0 if tg.shape == tr.shape else np.inf))
    return e_row, e_col, e_traj


def compare(case, extra=None):
    """None when the HIP fit of the case agrees with the oracle within the sweep's tolerances, else the message."""
    got, err = run_hip(case, extra)
    ref, rerr = run_oracle(case)
    if (got is None) != (ref is None):
        return f"one side failed: hip {err!r} oracle {rerr!r}", None
    if got is None:
        return None, None
    e_row, e_col, e_traj = errors(got, ref)
    tol, tol_traj = tolerances(case)
    line = (f"row {e_row:.2e} col {e_col:.2e} traj {e_traj:.2e} iters {got['iters']} vs {ref['iters']} "
            f"sweeps oracle {ref.get('total_sweeps')}")
    if not (e_row < tol and e_col < tol and e_traj < tol_traj and got["iters"] == ref["iters"]):
        return line, line
    return None, line


def main():
    import __graft_entry__ as ge
    ge.build()
    from oracle import c_oracle
    c_oracle.build()
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    bad = 0
    t0 = time.time()
    # FUZZ_ONLY=<case>: run that case alone (same random stream), FUZZ_OPTS="name=value,...": with these options on top of its own
    only = int(os.environ["FUZZ_ONLY"]) if "FUZZ_ONLY" in os.environ else None
    extra = dict((k, float(v)) for k, v in (kv.split("=") for kv in os.environ.get("FUZZ_OPTS", "").split(",") if kv))
    dump = os.environ.get("FUZZ_DUMP")
    for idx in range(ncases):
        case = draw_case(rng)
        if case is None or (only is not None and idx != only):
            continue   # (every draw of the case has been made: the stream of the later cases is unchanged)
        msg, line = compare(case, extra)
        if only is not None and line:
            print(f"case {idx}: {line} opts {dict(case['opts'], **extra)}", flush=True)
        if msg:
            bad += 1
            print(f"MISMATCH case {idx}: {case['kw']} opts {dict(case['opts'], **extra)} iters {case['iters']} seed {case['seed']} "
                  f"scale {case['scale']} m {case['m']} sub_tol {case['sub_tol']}: {msg}", flush=True)
            if dump:
                with open(dump, "a") as fh:
                    fh.write(json.dumps(dict(case, found_as=f"fuzz_parity.py {ncases} {sys.argv[2] if len(sys.argv) > 2 else 12345} case {idx}",
                                             deviation=msg)) + "\n")
        if idx % 25 == 24:
            print(f"... {idx + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"{ncases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
