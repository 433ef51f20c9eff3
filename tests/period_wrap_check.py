"""Run with INSIDER_HIP_LIB / INSIDER_ORACLE_LIB pointing at builds of the library and the oracle that share a SHORT order
period (-DINSIDER_PERM_PERIOD=64u, tests/test_gpu_period.py): every solve below runs far beyond the period, so the device
code that wraps the order table (tb -> tb0, the look-ahead row, `& (PERIOD - 1)` indexing in the three CD kernels, resumed
passes with start_sweep >= PERIOD) is compared with the oracle sweep by sweep: identical sweep counts, betas to 1e-9.
Exit code 0 = all agree."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert os.environ.get("INSIDER_HIP_LIB") and os.environ.get("INSIDER_ORACLE_LIB"), "variant builds required"
from insider_amd import api, workloads  # noqa: E402
from oracle import c_oracle  # noqa: E402  (test infrastructure: the checker)

PERIOD = 64
bad = 0


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def report(name, ok, detail):
    global bad
    print(("ok   " if ok else "FAIL ") + name + ": " + detail, flush=True)
    bad += 0 if ok else 1


# the periodic sequence itself: sweep s and sweep s + PERIOD have the same order in the variant oracle
for K in (5, 30):
    same = all(c_oracle.sweep_order(K, 3, 1, s) == c_oracle.sweep_order(K, 3, 1, s + PERIOD) for s in range(PERIOD))
    differs = any(c_oracle.sweep_order(K, 3, 1, s) != c_oracle.sweep_order(K, 3, 1, s + 1) for s in range(PERIOD - 1))
    report(f"oracle order period K={K}", same and differs, f"period {PERIOD}")

# (1) the stand-alone batch solver: nearly collinear designs (CD needs thousands of sweeps), tol < 0 = every solve runs exactly to
#     the cap, far from converged — so the iterate at the cap depends on every sweep's order: an order-perturbed oracle run
#     (another outer-iteration stream) lands somewhere else, the library (same stream, wrapped 5 times) on the oracle's iterate
rng = np.random.default_rng(5)
cap = 5 * PERIOD + 7
for K in (3, 16, 20, 30, 32, 40):
    B = 24
    base = rng.standard_normal((B, 60, 1))
    X = base + 0.02 * rng.standard_normal((B, 60, K))
    y = rng.standard_normal((B, 60)) + X.sum(axis=2)
    G = np.einsum("bik,bil->bkl", X, X)
    q = np.einsum("bik,bi->bk", X, y)
    beta, sw = api.strong_coordinate_descent(None, None, np.zeros((B, K)), 0.01, 0.4, G, q, tol=-1.0, seed=11, it=2,
                                             max_sweeps=cap, return_sweeps=True)
    worst, sens = 0.0, np.inf
    for b in range(B):
        rb, rs = c_oracle.strong_cd(X[b], y[b], np.zeros(K), 0.01, 0.4, G[b], q[b], tol=-1.0, seed=11, it=2, max_sweeps=cap)
        ob, _ = c_oracle.strong_cd(X[b], y[b], np.zeros(K), 0.01, 0.4, G[b], q[b], tol=-1.0, seed=11, it=3, max_sweeps=cap)
        worst = max(worst, relerr(beta[b], rb))
        sens = min(sens, relerr(ob, rb))
        assert rs == cap
    ok = worst < 1e-9 and np.all(sw == cap) and (K < 4 or sens > 1e-5)
    report(f"strong_cd batch K={K}", ok, f"rel {worst:.1e} at sweep {cap}; another order stream differs by >= {sens:.1e}")

# (2) the column update on a handle: the three CD kernels, single- and multi-pass (limits 48, 96, 192, ... cross the period,
#     and the later passes resume at start_sweep >= PERIOD); nearly collinear row factors, every solve runs to the cap
w = workloads.small(K=14, n=90, p=44, seed=21, with_na=True)
rs = np.random.default_rng(2)
v = rs.standard_normal(w.K)
A = [np.asfortranarray(np.outer(rs.standard_normal(a.shape[0]), v) + 0.02 * rs.standard_normal(a.shape)) for a in w.A0]
C0 = np.asfortranarray(rs.standard_normal(w.C0.shape) * 0.3)
R = sum(A[i][w.levels[:, i] - 1, :] for i in range(len(A)))
cap = 4 * PERIOD + 9
sink = c_oracle.set_sweep_sink(w.p)
ref, ref_total = c_oracle.optimize_col(w.X, w.M_train, np.asfortranarray(R), C0, 0.01, 0.4, tuning=1, tol=-1.0, seed=17, it=3,
                                       max_sweeps=cap)
ref_sw = sink.copy()
other, _ = c_oracle.optimize_col(w.X, w.M_train, np.asfortranarray(R), C0, 0.01, 0.4, tuning=1, tol=-1.0, seed=17, it=4, max_sweeps=cap)
c_oracle.set_sweep_sink(None)
sens = relerr(other, ref)
for variant in (0, 1, 2):
    for pass1 in (0, 48):
        ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
        ds.set_option("cd_variant", variant)
        ds.set_option("max_sweeps", cap)
        ds.set_option("cd_pass1", pass1)
        ds.set_option("cd_pass_ratio", 2)
        ds.set_option("cd_cold_iters", 9)
        got = ds.optimize_col([a.copy(order="F") for a in A], C0.copy(order="F"), lambda_=0.01, alpha=0.4, tuning=1, tol=-1.0,
                              seed=17, it=3)
        sw = ds.sweeps()
        ds.close()
        report(f"optimize_col cd_variant={variant} cd_pass1={pass1}",
               relerr(got, ref) < 1e-9 and np.array_equal(sw, ref_sw) and int(ref_sw.min()) == cap and sens > 1e-5,
               f"rel {relerr(got, ref):.1e} at sweep {cap} (per-gene sweeps equal {np.array_equal(sw, ref_sw)}); another order stream "
               f"differs by {sens:.1e}")

# (3) a whole fit with multi-pass solves whose pass limits straddle the period
w = workloads.small(K=20, n=120, p=50, seed=8)
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
for k, v in (("max_sweeps", 300), ("cd_pass1", 48), ("cd_pass_ratio", 2), ("cd_cold_iters", 9)):
    ds.set_option(k, v)
got = ds.optimize([a.copy(order="F") for a in w.A0], w.C0.copy(order="F"), w.K, w.lam, w.lam, w.alpha, max_iter=2, seed=4,
                  sub_tol=1e-12)
ds.close()
ref = c_oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, max_iter=2, seed=4,
                        sub_tol=1e-12, max_sweeps=300)
e = relerr(got["column_factor"], ref["column_factor"])
report("fit, multi-pass across the period", e < 1e-7, f"rel C {e:.1e}, oracle sweeps {ref['total_sweeps']}")
print(f"{bad} failures")
sys.exit(1 if bad else 0)
