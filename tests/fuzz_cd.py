"""Randomised parity sweep of insider_hip_strong_cd (GPU) against the oracle's strong_cd on degenerate subproblems:
zero / duplicated / nearly collinear regressors, alpha in {0, ..., 1}, huge and tiny lambda, zero right-hand sides, odd
batch sizes.        python tests/fuzz_cd.py [cases] [seed]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from insider_amd import api
from oracle import c_oracle
c_oracle.build()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for case in range(ncases):
    kchoice = [1, 2, 3, 7, 15, 16, 17, 18, 22, 24, 29, 30, 31, 32, 33, 47, 64]
    if os.environ.get("FUZZ_K"):
        kchoice = [int(k) for k in os.environ["FUZZ_K"].split(",")]
    K = int(rng.choice(kchoice))
    B = int(rng.choice([1, 2, 3, 4, 5, 9, 33]))
    m = int(rng.integers(max(2, K // 2), 3 * K + 20))
    lam = float(rng.choice([1e-3, 0.5, 3.0, 50.0, 1e4]))
    alpha = float(rng.choice([0.0, 0.05, 0.4, 0.9, 1.0]))
    tol = float(rng.choice([1e-5, 1e-9, 1e-12]))
    mode = int(rng.integers(0, 2))
    seed, it = int(rng.integers(0, 1 << 30)), int(rng.integers(0, 100))
    Gs, qs, ws, Xs, ys = [], [], [], [], []
    for b in range(B):
        X = rng.standard_normal((m, K))
        kind = rng.integers(0, 6)
        if kind == 1 and K > 1:
            X[:, rng.integers(0, K)] = 0.0                       # a dead regressor
        if kind == 2 and K > 1:
            X[:, -1] = X[:, 0]                                   # exact duplicate
        if kind == 3 and K > 2:
            X[:, 1] = X[:, 0] + 1e-7 * rng.standard_normal(m)    # nearly collinear
        if kind == 4:
            X *= 1e-3
        y = X @ (rng.standard_normal(K) * (rng.random(K) < 0.5)) + 0.3 * rng.standard_normal(m)
        if kind == 5:
            y[:] = 0.0
        Xs.append(X); ys.append(y); Gs.append(X.T @ X); qs.append(X.T @ y)
        ws.append(rng.standard_normal(K) * float(rng.choice([0.0, 0.1, 5.0])))
    beta, sw = api.strong_coordinate_descent(None, None, np.array(ws), lam, alpha, np.array(Gs), np.array(qs), tol=tol,
                                             seed=seed, it=it, order_mode=mode, max_sweeps=400, return_sweeps=True)
    beta, sw = np.atleast_2d(beta), np.atleast_1d(sw)
    for b in range(B):
        ob, osw = c_oracle.strong_cd(Xs[b], ys[b], ws[b], lam, alpha, Gs[b], qs[b], tol=tol, seed=seed, unit=1000 + b, it=it,
                                     order_mode=mode, max_sweeps=400)
        scale = max(1.0, float(np.max(np.abs(ob))))
        err = float(np.max(np.abs(ob - beta[b]))) / scale
        # same sweep count: the same iterate to 1e-9.  Different counts (the oracle differences two large loss values, so
        # at the tolerance floor its stopping test is noisy; on nearly collinear regressors a few extra sweeps still
        # move beta by 1e-7): both must then sit within the tolerance's reach of the same objective value.
        def objective(bv):
            return 0.5 * bv @ Gs[b] @ bv - qs[b] @ bv + 0.5 * lam * (1 - alpha) * bv @ bv + lam * alpha * np.sum(np.abs(bv))
        fo, fh = objective(ob), objective(beta[b])
        nsw = abs(int(osw) - int(sw[b]))
        ok = np.all(np.isfinite(beta[b])) and (err < 1e-9 if nsw == 0 else
                                               abs(fo - fh) <= 4 * (nsw + 1) * tol + 1e-12 * max(1.0, abs(fo)))
        if not ok:
            bad += 1
            print(f"MISMATCH case {case} b {b}: K {K} B {B} m {m} lam {lam} alpha {alpha} tol {tol} mode {mode} "
                  f"sweeps {sw[b]} vs {osw} err {err:.2e} finite {np.all(np.isfinite(beta[b]))}", flush=True)
            break
print(f"{ncases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
