"""Pins the CPU oracle's elastic-net solver (reference src/coordinate_descent.cpp:56-127).

The reference has no golden vectors (parity unpinned), so the oracle is pinned by mathematics:
a hand-computed K=1 case, KKT certificates, closed-form ridge, scikit-learn's ElasticNet as an independent
solver, and an independent numpy restatement.
"""
import os

import numpy as np
import pytest

from oracle import numpy_oracle as NO


def _problem(m, K, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((m, K)) * scale
    beta_true = rng.standard_normal(K) * (rng.random(K) < 0.6)
    y = X @ beta_true + 0.5 * rng.standard_normal(m)
    return X, y


def kkt_violation(G, q, beta, lam, alpha):
    """Max violation of the elastic-net optimality conditions."""
    g = q - G @ beta
    nz = beta != 0
    v_nz = np.abs(g[nz] - lam * (1 - alpha) * beta[nz] - lam * alpha * np.sign(beta[nz]))
    v_z = np.maximum(np.abs(g[~nz]) - lam * alpha, 0.0)
    return max(v_nz.max(initial=0.0), v_z.max(initial=0.0))


def test_hand_kat_k1(oracle):
    # X=[1,1]', y=[3,1]', lambda=1, alpha=0.5 => u=4, beta=(4-0.5)/(2+0.5)=1.4
    X = np.array([[1.0], [1.0]])
    y = np.array([3.0, 1.0])
    beta, _ = oracle.strong_cd(X, y, np.zeros(1), 1.0, 0.5, X.T @ X, X.T @ y, tol=1e-12)
    assert beta[0] == pytest.approx(1.4, abs=1e-14)
    nb, _ = NO.strong_coordinate_descent(X, y, np.zeros(1), 1.0, 0.5, X.T @ X, X.T @ y, tol=1e-12)
    assert nb[0] == pytest.approx(1.4, abs=1e-14)


@pytest.mark.parametrize("seed,K,lam,alpha", [(0, 5, 2.0, 0.4), (1, 20, 5.0, 0.2), (2, 30, 10.0, 0.5),
                                               (3, 8, 0.5, 0.9), (4, 64, 3.0, 0.3)])
def test_kkt_and_sklearn(oracle, seed, K, lam, alpha):
    from sklearn.linear_model import ElasticNet
    m = 300
    X, y = _problem(m, K, seed)
    G, q = X.T @ X, X.T @ y
    beta, sweeps = oracle.strong_cd(X, y, np.zeros(K), lam, alpha, G, q, tol=1e-13, seed=11, unit=seed)
    assert sweeps >= 1
    assert kkt_violation(G, q, beta, lam, alpha) < 1e-5
    # sklearn minimises 1/(2m)||y-Xb||^2 + a*l1r*|b|_1 + a(1-l1r)/2*||b||^2 = reference objective / m with a=lam/m
    en = ElasticNet(alpha=lam / m, l1_ratio=alpha, fit_intercept=False, tol=1e-14, max_iter=200000)
    en.fit(X, y)
    assert np.max(np.abs(en.coef_ - beta)) < 1e-7


def test_c_matches_numpy_restatement(oracle):
    for seed in range(6):
        K = 3 + 4 * seed
        X, y = _problem(80, K, 100 + seed)
        G, q = X.T @ X, X.T @ y
        w = np.random.default_rng(seed).standard_normal(K) * 0.1
        for tol in (1e-5, 1e-10):
            for mode in (0, 1):
                b1, s1 = oracle.strong_cd(X, y, w, 3.0, 0.4, G, q, tol=tol, seed=77, unit=seed, it=3,
                                          order_mode=mode)
                b2, s2 = NO.strong_coordinate_descent(X, y, w, 3.0, 0.4, G, q, tol=tol, seed=77, unit=seed, it=3,
                                                      order_mode=mode)
                assert s1 == s2
                np.testing.assert_allclose(b1, b2, rtol=0, atol=1e-13)


def test_strong_rule_excludes_and_kkt_readmits(oracle):
    # lambda large enough that the strong rule discards coordinates; the result must still satisfy KKT
    X, y = _problem(200, 12, 5)
    G, q = X.T @ X, X.T @ y
    lam = 0.9 * np.max(np.abs(q)) / 0.5
    beta, _ = oracle.strong_cd(X, y, np.ones(12), lam, 0.5, G, q, tol=1e-13)
    assert np.count_nonzero(beta) < 12
    assert kkt_violation(G, q, beta, lam, 0.5) < 1e-6
    # everything screened out and nothing violating => exact zeros, one (empty) sweep
    lam = 4.0 * np.max(np.abs(q))
    beta, sweeps = oracle.strong_cd(X, y, np.ones(12), lam, 0.5, G, q, tol=1e-13)
    assert np.all(beta == 0.0) and sweeps == 1


def test_order_is_seeded_and_key_unique():
    base = NO.perm_base(123456789012345, 3, 2)
    keys = [NO.perm_key(base, l) for l in range(64)]
    assert len(set(keys)) == 64 and all(k & 63 == l for l, k in enumerate(keys))
    o1 = NO.sweep_order(range(30), 5, 1, 2, 3, 0)
    o2 = NO.sweep_order(range(30), 5, 1, 2, 4, 0)
    assert sorted(o1) == list(range(30)) and o1 != o2 and o1 != list(range(30))


def test_order_sequence_is_periodic(oracle):
    """include/insider_perm.h: the order sequence of a solve repeats after INSIDER_PERM_PERIOD = 16384 sweeps (what lets the
    HIP side keep a sweep-order table of fixed size and run without a sweep cap).  The numpy restatement and the compiled
    oracle must both follow it: a solve stopped at 16384 + 7 sweeps equals, from sweep 16384 on, one restarted there."""
    P = NO.PERM_PERIOD
    assert P == 16384
    assert NO.sweep_order(range(30), 5, 1, 2, 3, 0) == NO.sweep_order(range(30), 5, 1, 2, 3 + P, 0)
    assert NO.sweep_order(range(30), 5, 1, 2, P - 1, 0) != NO.sweep_order(range(30), 5, 1, 2, 2 * P, 0)
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "insider_perm.h")).read()
    assert "#define INSIDER_PERM_PERIOD 16384u" in hdr
    # the compiled oracle applies the same sequence as the numpy restatement, before and after the wrap
    for sweep in (0, 1, 4097, P - 1, P, P + 1, 3 * P + 4097):
        assert oracle.sweep_order(30, 5, 2, sweep) == NO.sweep_order(range(30), 5, 1, 2, sweep, 0), sweep
    assert oracle.sweep_order(30, 5, 2, 4097) == oracle.sweep_order(30, 5, 2, 3 * P + 4097)


def test_perm_uniformity_rough():
    # position of coordinate 0 over many sweeps should be ~uniform over K slots
    K = 8
    counts = np.zeros(K)
    for s in range(4000):
        counts[NO.sweep_order(range(K), 99, 5, 1, s, 0).index(0)] += 1
    assert counts.min() > 350 and counts.max() < 650


def test_ridge_closed_form(oracle):
    rng = np.random.default_rng(3)
    A = rng.standard_normal((40, 9))
    G = A.T @ A + 0.7 * np.eye(9)
    b = rng.standard_normal((9, 3))
    np.testing.assert_allclose(oracle.solve_sympd(G, b), np.linalg.solve(G, b), rtol=1e-12, atol=1e-13)
    # not positive definite => LU fallback of solve(..., likely_sympd)
    N = rng.standard_normal((6, 6))
    v = rng.standard_normal(6)
    np.testing.assert_allclose(oracle.solve_sympd(N, v), np.linalg.solve(N, v), rtol=1e-9, atol=1e-11)


def test_sweep_cap_terminates(oracle):
    X, y = _problem(50, 6, 9)
    beta, sweeps = oracle.strong_cd(X, y, np.zeros(6), 1.0, 0.3, X.T @ X, X.T @ y, tol=0.0, max_sweeps=7)
    assert sweeps == 7 and np.all(np.isfinite(beta))


def test_covariance_form_variant_matches_reference_form(oracle):
    """bench.py's labelled CPU-optimised baseline variant (covariance-form sweeps, oracle.set_cd_form(1)) walks the same
    iterates as the reference's residual-form solver: same sweep counts, same solution; the gene-loop chunk size
    (oracle.set_col_chunk) does not change results."""
    from insider_amd import workloads
    for seed, (K, lam, alpha, tol) in enumerate([(5, 2.0, 0.4, 1e-6), (30, 5.0, 0.4, 1e-5), (17, 200.0, 0.5, 1e-8)]):
        X, y = _problem(250, K, 40 + seed)
        G, q = X.T @ X, X.T @ y
        w0 = np.random.default_rng(seed).standard_normal(K) * 0.1
        b0, s0 = oracle.strong_cd(X, y, w0, lam, alpha, G, q, tol=tol, seed=3, it=seed)
        b1, s1 = oracle.strong_cd_cov(w0, lam, alpha, G, q, tol=tol, seed=3, it=seed)
        assert abs(s0 - s1) <= 1 and np.max(np.abs(b0 - b1)) < 1e-9
    w = workloads.small(n=60, p=230, K=6)
    kw = dict(tuning=1, max_iter=3, seed=5)
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, **kw)
    try:
        oracle.set_col_chunk(1)
        r1 = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, **kw)
        oracle.set_cd_form(1)
        r2 = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, **kw)
    finally:
        oracle.set_col_chunk(100)
        oracle.set_cd_form(0)
    assert np.array_equal(r1["column_factor"], ref["column_factor"]) and r1["total_sweeps"] == ref["total_sweeps"]
    assert np.max(np.abs(r2["column_factor"] - ref["column_factor"])) < 1e-8
    assert abs(r2["total_sweeps"] - ref["total_sweeps"]) <= 0.01 * ref["total_sweeps"] + 2


def test_sweep_order_golden_bytes(oracle):
    """The sweep-order spec is SHARED by the oracle and the HIP kernels (include/insider_perm.h), so the parity tests cannot
    see an edit of it: these committed vectors (tests/golden/perm_golden.json, made by tests/golden/make_perm_golden.py) do.
    Both restatements — the compiled oracle and the numpy one — must reproduce every one of them."""
    import json
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "perm_golden.json")))
    assert len(gold) == 288
    for g in gold:
        assert oracle.sweep_order(g["K"], g["seed"], g["iter"], g["sweep"]) == g["order"], g
        assert NO.sweep_order(range(g["K"]), g["seed"], 0, g["iter"], g["sweep"], 0) == g["order"], g
        assert sorted(g["order"]) == list(range(g["K"]))
