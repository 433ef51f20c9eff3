"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on identical seeded inputs.

Tolerances (fp64): statistics rel 1e-11; one outer iteration rel 1e-9 on the factors;
31 iterations rel 1e-6 on the factors, 1e-9 on the loss trajectory.  The GPU path sums in a different order
(complement statistics on MFMA, level sums) so bitwise equality is not expected.
"""
import numpy as np
import pytest

from insider_amd import _lib, api, workloads

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def _rand_factors(w, seed, scale=0.5):
    rng = np.random.default_rng(seed)
    A = [np.asfortranarray(rng.standard_normal(a.shape) * scale) for a in w.A0]
    C = np.asfortranarray(rng.standard_normal(w.C0.shape) * scale)
    return A, C


def _cp(w):
    """Fresh copies of the inits: optimize() updates float64 Fortran arrays in place, like the reference."""
    return [a.copy(order="F") for a in w.A0], w.C0.copy(order="F")


def _R(w, A):
    return sum(A[i][w.levels[:, i] - 1, :] for i in range(w.levels.shape[1]))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X box")


# the masked statistics have several implementations each (per-entry lists / factored per level, the column side also
# from dense per-gene pair counts); a cost model picks one per data set, so every case runs with each of them forced
PATHS = {"fast": dict(row_merged=2, col_factored=2, row_counts=0), "pair": dict(row_merged=2, col_factored=3, row_counts=1),
         "lists": dict(row_merged=0, col_factored=0)}


# K + 1 <= 16 -> one MFMA block, <= 32 -> 2x2 blocks, ... : cover every block geometry and its edges
@pytest.mark.parametrize("K", [1, 4, 15, 16, 30, 31, 32, 47, 48, 63])
def test_masked_gram_cols_and_rows(oracle, K):
    w = workloads.small(n=150, p=70, level_counts=(6, 5), K=K, f=0.15, seed=K, with_na=True)
    A, C = _rand_factors(w, K + 1)
    R = _R(w, A)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    G, q = ds.masked_gram_cols(R)
    H, b = ds.masked_gram_rows(C)
    ds.close()
    for j in range(w.p):
        XtX, Xty = oracle.masked_gram_col(w.X[:, j], w.M_train[:, j], R)
        assert relerr(G[j], XtX) < 1e-11 and relerr(q[j], Xty) < 1e-11, j
    for r in range(w.n):
        XtX, Xty = oracle.masked_gram_row(w.X, w.M_train, r, C)
        assert relerr(H[r], XtX) < 1e-11 and relerr(b[r], Xty) < 1e-11, r


@pytest.mark.parametrize("K", [16, 19, 20, 23, 24, 25, 27, 28, 30, 31])
def test_masked_gram_on_the_4x4x4_matrix_instruction(oracle, K):
    """16 <= K <= 31: the per-entry statistics kernel tiles the augmented outer products 4 x 4 (k_list_stats4: v_mfma_f64_4x4x4,
    NT = 5 .. 8 tile rows, x in coordinate K, rows read from a tile-pair-interleaved copy of the factor) instead of in 16 x 16
    blocks (option list_fine = 0).  Every NT and the K at its edges, both sides, ragged and empty lines, against the oracle and
    against the 16x16x4 form."""
    w = workloads.small(n=170, p=90, level_counts=(6, 5), K=K, f=0.3, seed=40 + K, with_na=True)
    w.M_train[7, :] = 1          # sample with nothing held out
    w.M_train[:, 3] = 0          # gene with everything held out (172 entries: several list blocks)
    A, C = _rand_factors(w, K + 2)
    R = _R(w, A)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    out = {}
    for fine in (1, 0):
        ds.set_option("list_fine", fine)
        out[fine] = (ds.masked_gram_cols(R), ds.masked_gram_rows(C))
    ds.close()
    (G, q), (H, b) = out[1]
    for j in range(w.p):
        XtX, Xty = oracle.masked_gram_col(w.X[:, j], w.M_train[:, j], R)
        np.testing.assert_allclose(G[j], XtX, rtol=0, atol=1e-10 * max(np.abs(XtX).max(), 1.0))
        np.testing.assert_allclose(q[j], Xty, rtol=0, atol=1e-10 * max(np.abs(Xty).max(), 1.0))
    for r in range(w.n):
        XtX, Xty = oracle.masked_gram_row(w.X, w.M_train, r, C)
        np.testing.assert_allclose(H[r], XtX, rtol=0, atol=1e-10 * max(np.abs(XtX).max(), 1.0))
        np.testing.assert_allclose(b[r], Xty, rtol=0, atol=1e-10 * max(np.abs(Xty).max(), 1.0))
    (G0, q0), (H0, b0) = out[0]
    assert relerr(G, G0) < 1e-13 and relerr(q, q0) < 1e-13 and relerr(H, H0) < 1e-13 and relerr(b, b0) < 1e-13


def test_masked_gram_edge_masks(oracle):
    # empty held-out set, everything held out, ragged sizes (n, p not multiples of the 128-element chunk)
    w = workloads.small(n=131, p=37, level_counts=(4, 3), K=5, f=0.2, seed=3)
    w.M_train[5, :] = 0          # sample with (nearly) everything held out
    w.M_train[6, :] = 1
    w.M_train[:, 0] = 1          # gene with nothing held out
    w.M_train[:, 1] = 0          # gene with everything held out
    A, C = _rand_factors(w, 9)
    R = _R(w, A)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    G, q = ds.masked_gram_cols(R)
    H, b = ds.masked_gram_rows(C)
    ds.close()
    for j in (0, 1, 2, 36):
        XtX, Xty = oracle.masked_gram_col(w.X[:, j], w.M_train[:, j], R)
        np.testing.assert_allclose(G[j], XtX, atol=1e-10)
        np.testing.assert_allclose(q[j], Xty, atol=1e-10)
    assert np.abs(G[1]).max() < 1e-10 and np.abs(q[1]).max() < 1e-10
    for r in (5, 6, 130):
        XtX, Xty = oracle.masked_gram_row(w.X, w.M_train, r, C)
        np.testing.assert_allclose(H[r], XtX, atol=1e-10)
        np.testing.assert_allclose(b[r], Xty, atol=1e-10)


def test_masked_gram_long_lines_exercise_ring_drain(oracle):
    # > 256 held-out entries per line forces mid-line drains of the LDS ring and the segment split on the row side
    w = workloads.small(n=3000, p=20, level_counts=(10, 3), K=7, f=0.5, seed=4)
    A, C = _rand_factors(w, 2)
    R = _R(w, A)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    G, q = ds.masked_gram_cols(R)
    ds.close()
    for j in range(w.p):
        XtX, Xty = oracle.masked_gram_col(w.X[:, j], w.M_train[:, j], R)
        assert relerr(G[j], XtX) < 1e-11 and relerr(q[j], Xty) < 1e-11
    w2 = workloads.small(n=12, p=9000, level_counts=(3, 2), K=7, f=0.4, seed=5)
    A2, C2 = _rand_factors(w2, 3)
    ds = api.InsiderData(w2.X, w2.levels, w2.M_train, w2.M_test)
    H, b = ds.masked_gram_rows(C2)
    ds.close()
    for r in range(w2.n):
        XtX, Xty = oracle.masked_gram_row(w2.X, w2.M_train, r, C2)
        assert relerr(H[r], XtX) < 1e-11 and relerr(b[r], Xty) < 1e-11


@pytest.mark.parametrize("K,lam,alpha,tol", [(1, 1.0, 0.5, 1e-12), (5, 2.0, 0.4, 1e-5), (20, 5.0, 0.2, 1e-8),
                                              (30, 10.0, 0.5, 1e-11), (64, 3.0, 0.3, 1e-7)])
def test_strong_cd_matches_oracle(oracle, K, lam, alpha, tol):
    rng = np.random.default_rng(K)
    B, m = 48, 200
    Gs, qs, ws, Xs, ys = [], [], [], [], []
    for b in range(B):
        X = rng.standard_normal((m, K))
        bt = rng.standard_normal(K) * (rng.random(K) < 0.6)
        y = X @ bt + 0.5 * rng.standard_normal(m)
        Xs.append(X); ys.append(y); Gs.append(X.T @ X); qs.append(X.T @ y); ws.append(rng.standard_normal(K) * 0.1)
    for mode in (0, 1):
        beta, sw = api.strong_coordinate_descent(None, None, np.array(ws), lam, alpha, np.array(Gs), np.array(qs),
                                                 tol=tol, seed=99, it=7, order_mode=mode,
                                                 return_sweeps=True)
        same_sweeps = 0
        for b in range(B):
            ob, osw = oracle.strong_cd(Xs[b], ys[b], ws[b], lam, alpha, Gs[b], qs[b], tol=tol, seed=99, unit=1000 + b,
                                       it=7, order_mode=mode)
            same_sweeps += int(osw == sw[b])
            # identical sweep order; the stopping rule differences the loss differently (exact increments on the
            # GPU, two large numbers in the reference), so allow one sweep of slack at loose tolerances
            assert abs(osw - sw[b]) <= 1
            assert np.max(np.abs(ob - beta[b])) < max(50 * np.sqrt(tol), 1e-9) if osw != sw[b] else \
                np.max(np.abs(ob - beta[b])) < 1e-9
        assert same_sweeps >= B - 2


@pytest.mark.parametrize("K", [2, 3, 8, 15, 16] + list(range(17, 33)) + [33, 35, 36, 37, 40, 41, 43, 44, 45, 47, 48])   # (> 32: three slots, the third in LDS)
def test_strong_cd_every_register_kernel_instantiation(oracle, K):
    """K <= 32 runs the register-resident kernel, instantiated per even K above 16 (insider_cd_reg.hpp), 32 < K <= 48 its
    three-slot form (third slot's Gram columns in LDS; KMAX = 36, 40, 44, 48): every instantiation against the oracle,
    with screened-out coordinates (strong rule) and a partial last wave."""
    rng = np.random.default_rng(100 + K)
    B, m = 7, 120
    Xs = rng.standard_normal((B, m, K)) @ (np.eye(K) + 0.3 * rng.standard_normal((K, K)))
    bt = rng.standard_normal((B, K)) * (rng.random((B, K)) < 0.5)
    ys = np.einsum("bmk,bk->bm", Xs, bt) + 0.3 * rng.standard_normal((B, m))
    Gs = np.einsum("bmk,bml->bkl", Xs, Xs)
    qs = np.einsum("bmk,bm->bk", Xs, ys)
    ws = 0.1 * rng.standard_normal((B, K))
    lam = 0.35 * float(np.max(np.abs(qs)))       # the strong rule screens out part of the coordinates
    beta, sw = api.strong_coordinate_descent(None, None, ws, lam, 0.6, Gs, qs, tol=1e-10, seed=5, it=3,
                                             return_sweeps=True)
    for b in range(B):
        ob, osw = oracle.strong_cd(Xs[b], ys[b], ws[b], lam, 0.6, Gs[b], qs[b], tol=1e-10, seed=5, unit=b, it=3)
        assert abs(osw - sw[b]) <= 1, (K, b, osw, sw[b])
        # one sweep of slack in the stopping rule (see test_strong_cd_matches_oracle) moves beta by ~sqrt(tol)
        assert np.max(np.abs(ob - beta[b])) < (1e-8 if osw == sw[b] else 50 * np.sqrt(1e-10)), (K, b)
        assert np.array_equal(ob == 0, beta[b] == 0)          # identical sparsity pattern


@pytest.mark.parametrize("K", [16, 17, 19, 21, 22, 24, 25, 28, 30, 31, 33, 36, 38, 40, 42, 44, 46, 47, 48])
def test_optimize_column_kernel_instantiations(oracle, K):
    w = workloads.small(K=K, n=90, p=75, seed=K, f=0.2)
    A, C = _cp(w)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    got = ds.optimize(A, C, w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=1, seed=3)
    ds.close()
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha,
                          tuning=1, max_iter=1, seed=3)
    assert relerr(got["column_factor"], ref["column_factor"]) < 1e-8
    assert got["loss"] == pytest.approx(ref["loss"], rel=1e-9)


@pytest.mark.parametrize("K", [3, 16, 17, 20, 24, 27, 31, 32])
@pytest.mark.parametrize("tuning", [1, 0])
def test_ridge_column_kernel_instantiations(oracle, K, tuning):
    """alpha == 0: the register-resident Gauss-Jordan ridge solve (insider_ridge_reg.hpp), every slot geometry."""
    w = workloads.small(K=K, n=100, p=70, seed=K + 50, f=0.2, with_na=(K % 2 == 0))
    A, C = _rand_factors(w, K)
    M = w.M_train if tuning == 1 else np.ones_like(w.M_train)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    got = ds.optimize_col(A, C.copy(order="F"), lambda_=0.7, alpha=0.0, tuning=tuning)
    ds.close()
    ref, _ = oracle.optimize_col(w.X, M, _R(w, A), C, 0.7, 0.0, tuning=tuning)
    assert relerr(got, ref) < 1e-10, relerr(got, ref)


@pytest.mark.parametrize("paths", ["fast", "pair", "lists"])
@pytest.mark.parametrize("kw", [dict(level_counts=(6,), n=90, p=80, K=5),                       # a single covariate
                                dict(level_counts=(300, 3), n=900, p=70, K=6, f=0.3),            # > 255 levels, > 64 per pass
                                dict(level_counts=(3, 4, 2, 5, 3), n=360, p=90, K=7, f=0.2),     # five covariates
                                dict(level_counts=(2, 2), n=40, p=300, K=18, f=0.5, with_na=True),  # long level groups
                                # > 2048 held-out samples per gene: index staging overflows, several 512-entry tiles
                                dict(level_counts=(5, 3), n=3000, p=24, K=4, f=0.8),
                                # ~400 held-out entries per (level, level) cell: the one-byte pair counts overflow
                                dict(level_counts=(2, 2), n=2000, p=16, K=4, f=0.8),
                                # 21 table rows: six k-steps of the count product, two count dwords per lane
                                dict(level_counts=(9, 8, 7, 6), n=700, p=60, K=9, f=0.2),
                                # 36 table rows: beyond the pair-count form (look-up form runs instead)
                                dict(level_counts=(12, 12, 12, 12), n=800, p=50, K=5, f=0.2)],
                         ids=["one-cov", "300-levels", "five-cov", "long-groups", "2400-held-out-per-gene", "count-overflow",
                              "21-table-rows", "36-table-rows"])
def test_statistics_paths_on_odd_covariate_structures(oracle, kw, paths):
    w = workloads.small(seed=91, **kw)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    for k, v in PATHS[paths].items():
        ds.set_option(k, v)
    got = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=10, seed=4)
    pr = ds.profile()
    ds.close()
    assert pr["row_merged"] == (paths != "lists") and pr["col_factored"] == (paths != "lists")
    if paths == "pair":   # the pair-count form needs every count to fit one byte; else the look-up form runs
        assert pr["col_pair"] == (kw["n"] not in (2000, 800))
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, tuning=1,
                          max_iter=10, seed=4)
    for i, a in enumerate(ref["row_matrices"]):
        assert relerr(got["row_matrices"][f"factor{i}"], a) < 1e-8
    assert relerr(got["column_factor"], ref["column_factor"]) < 1e-8
    np.testing.assert_allclose(got["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-9, equal_nan=True)


def test_strong_cd_hand_kat():
    # SURVEY.md 8c item 4
    beta = api.strong_coordinate_descent(None, None, np.zeros(1), 1.0, 0.5, np.array([[2.0]]), np.array([4.0]),
                                         tol=1e-12)
    assert beta[0] == pytest.approx(1.4, abs=1e-14)


def test_strong_cd_kkt_at_tight_tolerance():
    from tests.test_oracle_cd import kkt_violation, _problem
    X, y = _problem(400, 30, 7)
    G, q = X.T @ X, X.T @ y
    for lam, alpha in ((5.0, 0.4), (0.9 * np.max(np.abs(q)) / 0.5, 0.5), (4.0 * np.max(np.abs(q)), 0.5)):
        beta = api.strong_coordinate_descent(X, y, np.ones(30), lam, alpha, G, q, tol=1e-13)
        assert kkt_violation(G, q, beta, lam, alpha) < 1e-5


CASES = {
    "plain": dict(),
    "na": dict(with_na=True),
    "interaction": dict(interaction_idx=(1, 2)),
    "unmasked": dict(tuning=0),
    "ridge": dict(alpha=0.0),
    "ridge_unmasked": dict(alpha=0.0, tuning=0),
    "k17": dict(K=17, n=90, p=120),
    "k30": dict(K=30, n=150, p=90),          # row16 kernel, two slots, 2 padded positions
    "k40": dict(K=40, n=150, p=60),          # K > 32: one gene per wavefront (group kernel), 3x3 MFMA blocks
    "three_cov": dict(level_counts=(4, 3, 5), n=120, p=100),
    # pure lasso (l2 = 0): latent dimensions die in every gene, the next row update returns exactly zero factor columns
    # and XtX_kk + lambda (1 - alpha) = 0 for them (found by tests/fuzz_parity.py: 0 * inf in the sweep)
    "lasso": dict(n=78, p=43, level_counts=(3, 9, 2, 8), K=13, f=0.07, lam=7.0, alpha=1.0, seed=962864),
}


@pytest.mark.parametrize("paths", list(PATHS))
@pytest.mark.parametrize("case", list(CASES))
def test_optimize_one_iteration(oracle, case, paths):
    w = workloads.small(**CASES[case])
    A, C = _rand_factors(w, 5, scale=0.3)   # non-trivial start so that every term of the update matters
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    for k, v in PATHS[paths].items():
        ds.set_option(k, v)
    got = ds.optimize([a.copy(order="F") for a in A], C.copy(order="F"), w.K, w.lam, w.lam, w.alpha, tuning=w.tuning,
                      max_iter=0, seed=17)
    ds.close()
    ref = oracle.optimize(w.X, w.levels, w.n_levels, A, C, w.M_train, w.M_test, w.lam, w.lam, w.alpha,
                          tuning=w.tuning, max_iter=0, seed=17)
    assert got["iters"] == ref["iters"]
    for i, a in enumerate(ref["row_matrices"]):
        assert relerr(got["row_matrices"][f"factor{i}"], a) < 1e-9
    assert relerr(got["column_factor"], ref["column_factor"]) < 1e-9
    np.testing.assert_allclose(got["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-9, equal_nan=True)


@pytest.mark.parametrize("paths", list(PATHS))
@pytest.mark.parametrize("case", list(CASES))
def test_optimize_31_iterations(oracle, case, paths):
    w = workloads.small(**CASES[case])
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    for k, v in PATHS[paths].items():
        ds.set_option(k, v)
    got = ds.optimize([a.copy(order="F") for a in w.A0], w.C0.copy(order="F"), w.K, w.lam, w.lam, w.alpha,
                      tuning=w.tuning, max_iter=30, seed=23)
    ds.close()
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha,
                          tuning=w.tuning, max_iter=30, seed=23)
    assert got["iters"] == ref["iters"] == 31
    assert list(got["traj"][:, 0]) == [-1, 0, 10, 20, 30]
    np.testing.assert_allclose(got["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-9, equal_nan=True)
    assert np.array_equal(got["traj"][:, 9], ref["traj"][:, 9])      # same decay schedule
    for i, a in enumerate(ref["row_matrices"]):
        assert relerr(got["row_matrices"][f"factor{i}"], a) < 1e-6
    assert relerr(got["column_factor"], ref["column_factor"]) < 1e-6
    assert got["loss"] == pytest.approx(ref["loss"], rel=1e-9)
    if w.tuning == 0:
        assert np.isnan(got["test_rmse"])


@pytest.mark.parametrize("kw,m", [(dict(n=48, p=16500, level_counts=(18, 3), K=5, f=0.2), 0),
                                  (dict(n=60, p=16400, level_counts=(50, 4, 3), K=17, f=0.15, with_na=True), 0),
                                  (dict(n=48, p=16390, level_counts=(12, 5), K=6, f=0.2), 2)],
                         ids=["two-cov", "three-cov-K17-gemm", "ctns2"])
def test_streaming_products_and_ticketed_statistics_at_many_genes(oracle, kw, m):
    """From 16384 genes on the row phase's products run on k_mm_rows2 / k_mm_reduce2 (option mm_fast) and the pair-count
    statistics' resident blocks (768 at most) each take many genes by ticket — the suite's other oracle comparisons have a few
    hundred genes and never reach either.  Few samples keep the oracle to seconds.  Against the oracle, against round 4's
    kernels (mm_fast = 0, col_mfma4 = 0), and with the two experimental arrangements of round 5 (join_lean, q_split), which
    may only reorder sums."""
    w = workloads.small(seed=77, **kw)
    rng = np.random.default_rng(6)
    Z = np.asfortranarray(rng.standard_normal((w.n, m))) if m else None
    U0 = [np.asfortranarray(rng.normal(0.0, 0.001, size=(m, w.K)))] if m else []
    out = {}
    for name, opts in (("default", {}), ("round4", dict(mm_fast=0, col_mfma4=0)), ("joins", dict(join_lean=7)), ("qsplit", dict(q_split=1))):
        ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test, ctns_confounder=Z)
        for k, v in PATHS["pair"].items():
            ds.set_option(k, v)
        for k, v in opts.items():
            ds.set_option(k, v)
        A, C = _cp(w)
        out[name] = ds.optimize(A + [u.copy(order="F") for u in U0], C, w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=4, seed=4,
                                inc_continuous=1 if m else 0)
        assert ds.profile()["col_pair"] and ds.profile()["row_merged"]
        ds.close()
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0 + U0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, tuning=1,
                          max_iter=4, seed=4, **(dict(ctns=Z) if m else {}))
    for name, got in out.items():
        np.testing.assert_allclose(got["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-8, equal_nan=True, err_msg=name)
        assert relerr(got["column_factor"], ref["column_factor"]) < 1e-6, name
        for i, a in enumerate(ref["row_matrices"]):
            assert relerr(got["row_matrices"][f"factor{i}"], a) < 1e-6, (name, i)
    for name in ("round4", "joins", "qsplit"):
        np.testing.assert_allclose(out[name]["traj"][:, 1:8], out["default"]["traj"][:, 1:8], rtol=1e-10, equal_nan=True, err_msg=name)
        assert relerr(out[name]["column_factor"], out["default"]["column_factor"]) < 1e-8, name
    assert np.array_equal(out["joins"]["column_factor"], out["default"]["column_factor"])   # the joins change no arithmetic


@pytest.mark.parametrize("opts", [dict(cd_variant=1), dict(cd_variant=2), dict(order_mode=1), dict(max_sweeps=7),
                                  dict(cd_variant=1, max_sweeps=5, order_mode=1),
                                  dict(cd_variant=2, max_sweeps=6, order_mode=1), dict(row_merged=0), dict(row_merged=2), dict(col_factored=0), dict(col_factored=2), dict(col_factored=3),
                                  dict(row_merged=0, col_factored=0)])
def test_optimize_options(oracle, opts):
    # the alternative CD kernels (one group of lanes per gene; LDS-resident row16), cyclic sweep order, the sweep cap,
    # the per-sample row update instead of the merged one, the per-entry column statistics instead of the factored ones
    w = workloads.small(K=9, n=70, p=90, with_na=True)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    for k, v in opts.items():
        ds.set_option(k, v)
    got = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, max_iter=12, seed=31)
    ds.close()
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, max_iter=12,
                          seed=31, order_mode=opts.get("order_mode", 0), max_sweeps=opts.get("max_sweeps", 1 << 24))
    np.testing.assert_allclose(got["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-9, equal_nan=True)
    assert relerr(got["column_factor"], ref["column_factor"]) < 1e-7


@pytest.mark.parametrize("K,limits", [(12, (32, 2)), (30, (32, 3)), (20, (64, 2)), (7, (32, 16)), (40, (32, 2)), (48, (64, 3))])
def test_multipass_column_solve_is_bit_identical(oracle, K, limits):
    """Cold outer iterations solve in passes (options cd_pass1 / cd_pass_ratio / cd_cold_iters, insider_cd_reg.hpp): a limited
    pass stops at a sweep index, the unfinished genes are re-packed by estimated remaining length and continued from their
    saved state.  The iterates must not depend on it: factors, trajectory and sweep counts bit-identical to the single-pass
    solve (cd_pass1 = 0), and parity with the oracle."""
    w = workloads.small(K=K, n=120, p=333, f=0.2, seed=K)          # tight tolerance below: hundreds of sweeps per gene
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    ds.set_option("profile", 1)
    ds.set_option("cd_pass1", 0)                                    # single-pass solves
    kw = dict(tuning=1, max_iter=4, sub_tol=1e-11, seed=9)
    ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=0, sub_tol=1e-11, seed=9)
    assert ds.sweeps().max() > limits[0] + 16, ds.sweeps().max()    # the first pass really stops solves half-way
    one = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, **kw)
    sw_one, tot_one = ds.sweeps(), ds.profile()["sweeps"]
    ds.set_option("cd_cold_iters", 100)
    ds.set_option("cd_pass1", limits[0])
    ds.set_option("cd_pass_ratio", limits[1])
    multi = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, **kw)
    sw_multi, tot_multi = ds.sweeps(), ds.profile()["sweeps"]
    ds.close()
    assert np.array_equal(one["column_factor"], multi["column_factor"])
    assert np.array_equal(one["traj"], multi["traj"], equal_nan=True)
    for i in range(len(w.A0)):
        assert np.array_equal(one["row_matrices"][f"factor{i}"], multi["row_matrices"][f"factor{i}"])
    assert np.array_equal(sw_one, sw_multi) and tot_one == tot_multi
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, tuning=1,
                          max_iter=4, sub_tol=1e-11, seed=9)
    assert relerr(multi["column_factor"], ref["column_factor"]) < 1e-6
    assert abs(tot_multi - ref["total_sweeps"]) <= max(3, 0.002 * ref["total_sweeps"])


@pytest.mark.parametrize("K,frac", [(12, 0.03), (30, 0.1), (20, 0.25), (30, 0.001)])
def test_split_column_step_is_bit_identical(oracle, K, frac):
    """Steady-state outer iterations can run the column step split (option cd_split = 2; OFF by default): the genes predicted longest —
    whole buckets of the launch order, at most cd_long_frac of the genes — get their statistics and their solve on a stream of
    their own, ahead of the statistics of everyone else.  Every gene's record and solve are the same computations as in the
    unsplit step, so factors, trajectory and sweep counts must be bit-identical (and agree with the oracle)."""
    w = workloads.small(K=K, n=150, p=1500, level_counts=(12, 5), f=0.2, seed=40 + K)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    ds.set_option("profile", 1)
    ds.set_option("col_factored", 3)            # the pair-count statistics (what c3 / c4 take): the split path's kernel
    assert int(ds.info("col_stats_path")) == 2
    kw = dict(tuning=1, max_iter=12, seed=9)
    ds.set_option("cd_split", 0)
    one = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, **kw)
    sw_one, tot_one = ds.sweeps(), ds.profile()["sweeps"]
    ds.set_option("cd_split", 2)                # forced (the default engages it by problem size)
    ds.set_option("cd_long_frac", frac)
    two = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, **kw)
    sw_two, tot_two = ds.sweeps(), ds.profile()["sweeps"]
    ds.close()
    assert np.array_equal(one["column_factor"], two["column_factor"])
    assert np.array_equal(one["traj"], two["traj"], equal_nan=True)
    for i in range(len(w.A0)):
        assert np.array_equal(one["row_matrices"][f"factor{i}"], two["row_matrices"][f"factor{i}"])
    assert np.array_equal(sw_one, sw_two) and tot_one == tot_two
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, tuning=1,
                          max_iter=12, seed=9)
    assert relerr(two["column_factor"], ref["column_factor"]) < 1e-6
    np.testing.assert_allclose(two["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-9, equal_nan=True)


@pytest.mark.parametrize("K,levels", [(9, (7, 4)), (30, (12, 5, 3)), (40, (6, 5))])
def test_fused_level_kernel_is_bit_identical(K, levels):
    """The merged row update forms a level's record tail, its normal equations and (single rank, K <= 31) its ridge solve in
    one launch (k_level_merged, option row_fused, default on) instead of three (k_level_pack / k_level_reduce /
    k_level_solve): same arithmetic on the same values, so every result is bit-identical."""
    w = workloads.small(K=K, n=140, p=260, level_counts=levels, f=0.2, seed=70 + K)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    ds.set_option("row_merged", 2)
    runs = []
    for fused in (0, 1):
        ds.set_option("row_fused", fused)
        runs.append(ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=6, seed=5))
    ds.close()
    assert np.array_equal(runs[0]["column_factor"], runs[1]["column_factor"])
    assert np.array_equal(runs[0]["traj"], runs[1]["traj"], equal_nan=True)
    for i in range(len(w.A0)):
        assert np.array_equal(runs[0]["row_matrices"][f"factor{i}"], runs[1]["row_matrices"][f"factor{i}"])


@pytest.mark.parametrize("K,levels,alpha", [(9, (7, 4), 0.4), (23, (2, 16, 8, 107), 0.4), (30, (12, 5, 3), 0.0), (40, (6, 5), 0.3)])
def test_unmasked_row_update_in_one_launch_per_covariate(oracle, K, levels, alpha):
    """tuning = 0 (src/optimize.cpp:178-191): the level equations are the merged update's with empty held-out sums, so the
    unmasked row update is one k_level_merged launch per covariate on an all-zero record (sum_{r in l} s_r from the level-pair
    sample counts), with R rebuilt once per outer iteration, instead of five launches over the samples per covariate (option
    row_fused = 0 keeps those).  Same equations, sum_{r in l} s_r in another order: the fits agree to rounding, and both agree
    with the oracle."""
    w = workloads.small(K=K, n=330, p=240, level_counts=levels, f=0.2, seed=30 + K)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    runs = []
    for fused in (0, 1):
        ds.set_option("row_fused", fused)
        runs.append(ds.optimize(*_cp(w), w.K, w.lam, w.lam, alpha, tuning=0, max_iter=6, seed=5))
    ds.close()
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, alpha, tuning=0,
                          max_iter=6, seed=5)
    for r in runs:
        assert relerr(r["column_factor"], ref["column_factor"]) < 1e-6
        assert np.allclose(r["traj"][:, 3:8], ref["traj"][:, 3:8], rtol=1e-9, atol=0)
        for i in range(len(w.A0)):
            assert relerr(r["row_matrices"][f"factor{i}"], ref["row_matrices"][i]) < 1e-6
    assert relerr(runs[0]["column_factor"], runs[1]["column_factor"]) < 1e-9
    for i in range(len(w.A0)):
        assert relerr(runs[0]["row_matrices"][f"factor{i}"], runs[1]["row_matrices"][f"factor{i}"]) < 1e-9


@pytest.mark.parametrize("K,levels", [(7, (60, 3)), (20, (100, 10)), (30, (130, 7, 2)), (33, (64, 5))])
def test_level_gram_as_gemm_matches_the_rank_one_form(oracle, K, levels):
    """Covariates with many levels get their per-level weighted Gram sums sum_j n_jl c_j c_j' from ONE GEMM over genes
    (k_wgemm: levels x packed index pairs, option row_gemm, default on) instead of one weighted rank-one update per (level,
    gene) (k_wsyrk).  Same sums in another order: the fits agree to rounding, and both agree with the oracle."""
    w = workloads.small(K=K, n=420, p=300, level_counts=levels, f=0.2, seed=90 + K)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    ds.set_option("row_merged", 2)
    runs = []
    for gemm in (0, 1):
        ds.set_option("row_gemm", gemm)
        runs.append(ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=4, seed=5))
    ds.close()
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, tuning=1,
                          max_iter=4, seed=5)
    for r in runs:
        for i, a in enumerate(ref["row_matrices"]):
            assert relerr(r["row_matrices"][f"factor{i}"], a) < 1e-7, i
        assert relerr(r["column_factor"], ref["column_factor"]) < 1e-7
        np.testing.assert_allclose(r["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-8, equal_nan=True)
    for i in range(len(w.A0)):
        assert relerr(runs[0]["row_matrices"][f"factor{i}"], runs[1]["row_matrices"][f"factor{i}"]) < 1e-9
    # the two forms really are different code paths (another summation order)
    assert not all(np.array_equal(runs[0]["row_matrices"][f"factor{i}"], runs[1]["row_matrices"][f"factor{i}"])
                   for i in range(len(w.A0)))


def test_sweep_counts_match_oracle(oracle):
    w = workloads.small(K=12, n=80, p=64)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    ds.set_option("profile", 1)
    ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, max_iter=4, seed=9)
    total = ds.profile()["sweeps"]
    ds.close()
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, max_iter=4,
                          seed=9)
    assert abs(total - ref["total_sweeps"]) <= max(3, 0.002 * ref["total_sweeps"])


def test_inplace_update_and_oneshot_operator(oracle):
    # the reference mutates cfd_factors / column_factor in place (src/optimize.cpp:283-284) and returns copies
    w = workloads.small()
    A = [a.copy(order="F") for a in w.A0]
    C = w.C0.copy(order="F")
    out = api.optimize(w.X, A, C, w.levels, None, w.M_train, w.M_test, 0, w.K, w.lam, w.lam, w.alpha, 1, 1e-10, 1e-5,
                       3, seed=5)
    assert np.array_equal(C, out["column_factor"]) and not np.array_equal(C, w.C0)
    assert np.array_equal(A[0], out["row_matrices"]["factor0"])
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha,
                          max_iter=3, seed=5)
    assert relerr(C, ref["column_factor"]) < 1e-8
    # the List the reference returns holds COPIES (src/optimize.cpp:413): the default; copy=False hands back the arguments
    # themselves (what the C ABI does, and what bench.py times), with the same numbers
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    A1, C1 = [a.copy(order="F") for a in w.A0], w.C0.copy(order="F")
    A2, C2 = [a.copy(order="F") for a in w.A0], w.C0.copy(order="F")
    r1 = ds.optimize(A1, C1, w.K, w.lam, w.lam, w.alpha, max_iter=3, seed=5)
    r2 = ds.optimize(A2, C2, w.K, w.lam, w.lam, w.alpha, max_iter=3, seed=5, copy=False)
    ds.close()
    assert r1["column_factor"] is not C1 and r2["column_factor"] is C2 and r2["row_matrices"]["factor0"] is A2[0]
    assert np.array_equal(C1, C2) and np.array_equal(r1["column_factor"], C2) and np.array_equal(A1[0], A2[0])


def test_global_tol_early_stop(oracle):
    w = workloads.small()
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    got = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, max_iter=200, global_tol=1e-3, seed=1)
    ds.close()
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha,
                          max_iter=200, global_tol=1e-3, seed=1)
    assert got["iters"] == ref["iters"] and got["traj"].shape == ref["traj"].shape


def test_bad_arguments_return_status():
    w = workloads.small()
    lev = w.levels.copy()
    lev[0, 0] = 0
    with pytest.raises(_lib.InsiderError) as e:
        api.InsiderData(w.X, lev, w.M_train, w.M_test, n_levels=w.n_levels)
    assert e.value.status == _lib.ERR_ARG
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    with pytest.raises(_lib.InsiderError) as e:
        ds.optimize(*_cp(w), w.K, tuning=5)
    assert e.value.status == _lib.ERR_ARG
    with pytest.raises(_lib.InsiderError) as e:     # continuous factor asked for, but the handle has no ctns_confounder
        A, C = _cp(w)
        ds.optimize(A + [np.zeros((1, w.K), order="F")], C, w.K, inc_continuous=1)
    assert e.value.status == _lib.ERR_ARG
    ds.close()


def test_handle_reuse_across_ranks_like_tune(oracle):
    # tune() keeps X/M resident and changes K / lambda / alpha between calls (R/insider.R:98-174)
    w = workloads.small(n=80, p=100, K=6)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    # K = 20 / 40 / 6 again: the workspace (statistics records, side-stream buffers) is rebuilt across MFMA block geometries
    for K, lam, alpha in ((3, 1.0, 0.2), (6, 3.0, 0.5), (3, 0.1, 0.0), (20, 2.0, 0.4), (40, 2.0, 0.3), (6, 3.0, 0.5)):
        A0, C0 = workloads.init_factors(w.n_levels, K, w.p, seed=K)
        got = ds.optimize([a.copy(order="F") for a in A0], C0.copy(order="F"), K, lam, lam, alpha, max_iter=5, seed=2)
        ref = oracle.optimize(w.X, w.levels, w.n_levels, A0, C0, w.M_train, w.M_test, lam, lam, alpha, max_iter=5,
                              seed=2)
        assert relerr(got["column_factor"], ref["column_factor"]) < 1e-7
        assert got["test_rmse"] == pytest.approx(ref["test_rmse"], rel=1e-9)
    ds.close()


# ---- BASELINE config c2 at full size: size-independent properties (the oracle is too slow there) --------------
@pytest.fixture(scope="module")
def c2():
    return workloads.make("c2")


def test_c2_full_size_properties(c2):
    w = c2
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    got = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=20, seed=3)
    assert int(ds.info("cap_hits")) == 0, ds.info("max_gene_sweeps")   # no elastic-net solve ended at the sweep cap
    ds.close()
    tr = got["traj"]
    assert np.all(np.diff(tr[:, 7]) < 0)                      # checkpoint losses decrease
    A = [got["row_matrices"][f"factor{i}"] for i in range(len(w.A0))]
    C = got["column_factor"]
    R = sum(A[i][w.levels[:, i] - 1, :] for i in range(w.levels.shape[1]))
    resid = w.X - R @ C
    sse = float(np.sum(resid[w.M_train != 0] ** 2))
    # loss components recomputed independently in numpy from the returned factors
    assert tr[-1, 3] == pytest.approx(sse / 2, rel=1e-10)
    assert tr[-1, 4] == pytest.approx(w.lam * sum(np.sum(a ** 2) for a in A) / 2, rel=1e-12)
    assert tr[-1, 5] == pytest.approx(w.lam * (1 - w.alpha) * np.sum(C ** 2) / 2, rel=1e-12)
    assert tr[-1, 6] == pytest.approx(w.lam * w.alpha * np.sum(np.abs(C)), rel=1e-12)
    assert got["test_rmse"] == pytest.approx(np.sqrt(np.mean(resid[w.M_test != 0] ** 2)), rel=1e-10)
    assert got["train_rmse"] == pytest.approx(np.sqrt(sse / np.count_nonzero(w.M_train)), rel=1e-10)
    assert got["test_rmse"] < 1.2                              # noise sd is 1: the fit generalises


def test_c2_all_ones_mask_equals_unmasked(c2):
    w = c2
    ones = np.ones_like(w.M_train)
    zeros = np.zeros_like(w.M_test)
    ds = api.InsiderData(w.X, w.levels, ones, zeros)
    r1 = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=10, seed=3)
    r0 = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, tuning=0, max_iter=10, seed=3)
    ds.close()
    assert relerr(r1["column_factor"], r0["column_factor"]) < 1e-9
    np.testing.assert_allclose(r1["traj"][:, 3:8], r0["traj"][:, 3:8], rtol=1e-10)


def test_c3_full_size_forms_agree_and_losses_recompute():
    """BASELINE's headline configuration (10000 x 50000, K = 30) at full size: every form of the masked statistics
    (per-entry lists / look-up / pair counts on the column side, per-sample / merged on the row side) gives the same
    11-iteration trajectory, and the loss components recomputed in numpy from the returned factors match."""
    w = workloads.make("c3")
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    runs = {}
    for name, opts in PATHS.items():
        for k, v in opts.items():
            ds.set_option(k, v)
        runs[name] = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=10, seed=3)
        assert int(ds.info("cap_hits")) == 0, (name, ds.info("max_gene_sweeps"))   # the reference has no sweep cap: none may bite
        pr = ds.profile()
        assert pr["col_factored"] == (name != "lists") and pr["col_pair"] == (name == "pair")
    ds.close()
    ref = runs["lists"]
    assert np.all(np.diff(ref["traj"][:, 7]) < 0)                       # checkpoint losses decrease
    for name in ("fast", "pair"):
        np.testing.assert_allclose(runs[name]["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-9, equal_nan=True)
        assert relerr(runs[name]["column_factor"], ref["column_factor"]) < 1e-7
        for i in range(len(w.A0)):
            assert relerr(runs[name]["row_matrices"][f"factor{i}"], ref["row_matrices"][f"factor{i}"]) < 1e-7
    got = runs["pair"]
    A = [got["row_matrices"][f"factor{i}"] for i in range(len(w.A0))]
    C = got["column_factor"]
    R = sum(A[i][w.levels[:, i] - 1, :] for i in range(w.levels.shape[1]))
    sse = sse_te = 0.0
    for b in range(0, w.p, 5000):                                        # gene blocks: bounded host memory
        resid = w.X[:, b:b + 5000] - R @ C[:, b:b + 5000]
        sse += float(np.sum(resid[w.M_train[:, b:b + 5000] != 0] ** 2))
        sse_te += float(np.sum(resid[w.M_test[:, b:b + 5000] != 0] ** 2))
    tr = got["traj"]
    assert tr[-1, 3] == pytest.approx(sse / 2, rel=1e-10)
    assert tr[-1, 4] == pytest.approx(w.lam * sum(np.sum(a ** 2) for a in A) / 2, rel=1e-12)
    assert tr[-1, 5] == pytest.approx(w.lam * (1 - w.alpha) * np.sum(C ** 2) / 2, rel=1e-12)
    assert tr[-1, 6] == pytest.approx(w.lam * w.alpha * np.sum(np.abs(C)), rel=1e-12)
    assert got["test_rmse"] == pytest.approx(np.sqrt(sse_te / np.count_nonzero(w.M_test)), rel=1e-10)
    assert got["train_rmse"] == pytest.approx(np.sqrt(sse / np.count_nonzero(w.M_train)), rel=1e-10)


def test_allreduce_callback_plumbing_single_gpu(oracle):
    """The exchange path of the gene-sharded driver on one GPU: torch.distributed (nccl = RCCL) with world_size 1, the
    library calling back with DEVICE pointers (per covariate per outer iteration + per checkpoint).  A sum over one
    rank is the identity, so the results must equal the unsharded run bit for bit."""
    import os
    import torch
    import torch.distributed as dist
    from insider_amd import dist as idist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        w = workloads.small(K=6, n=70, p=96)
        ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
        plain = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, max_iter=10, seed=4)
        ar = idist.attach(ds, 0, 0, 1, device=0, force=True, mode="torch")
        forced = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, max_iter=10, seed=4)
        ds.close()
        assert np.array_equal(plain["column_factor"], forced["column_factor"])
        assert np.array_equal(plain["traj"], forced["traj"], equal_nan=True)
        KP = 16
        per_iter = [int(L) * (KP * KP + KP) for L in w.n_levels]
        # 11 outer iterations x one all-reduce per covariate, + 6 doubles per loss evaluation (initial + iter 0, 10)
        assert ar.calls == [6] + (per_iter * 1 + [6]) + per_iter * 9 + per_iter + [6]
        assert ar.zero_copy in (True, False)
        print("all-reduce zero-copy aliasing:", ar.zero_copy)
        # the result-table exchange of a grid-parallel tune() on the nccl backend (host table -> device -> all-reduce -> host)
        tab = np.arange(12.0).reshape(3, 4)
        assert np.array_equal(api._grid_sum(tab.copy(), world=2), tab)
    finally:
        dist.destroy_process_group()


def test_in_library_rccl_allreduce_single_gpu(oracle):
    """The in-library collective (insider_hip_comm_init: ncclAllReduce enqueued on the library's stream, no callback into
    the host) with a one-rank RCCL communicator on the one GPU of the box: every exchange point of the sharded driver runs
    through RCCL; a sum over one rank is the identity, so the fit equals the unsharded one bit for bit."""
    from insider_amd import dist as idist
    w = workloads.small(K=6, n=70, p=96)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    plain = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, max_iter=10, seed=4)
    assert idist.attach(ds, 0, 0, 1, device=0, force=True, mode="rccl") == "rccl"
    forced = ds.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, max_iter=10, seed=4)
    # a world > 1 handle with neither a communicator nor a callback refuses to run instead of silently skipping the sum
    ds2 = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    ds2.set_shard(0, 0, 2, None)
    with pytest.raises(_lib.InsiderError) as e:
        ds2.optimize(*_cp(w), w.K, w.lam, w.lam, w.alpha, max_iter=1, seed=4)
    assert e.value.status == _lib.ERR_COMM
    ds2.close()
    ds.close()
    assert np.array_equal(plain["column_factor"], forced["column_factor"])
    assert np.array_equal(plain["traj"], forced["traj"], equal_nan=True)
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, max_iter=10,
                          seed=4)
    np.testing.assert_allclose(forced["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-9, equal_nan=True)


@pytest.mark.parametrize("kw,m", [(dict(), 1), (dict(with_na=True), 2), (dict(tuning=0), 2), (dict(K=17, n=90, p=70), 3)])
def test_optimize_with_continuous_covariates(oracle, kw, m):
    # optimize_continuous_v2 (src/optimize.cpp:76-137) through the weighted level machinery
    w = workloads.small(**kw)
    rng = np.random.default_rng(8)
    Z = np.asfortranarray(rng.standard_normal((w.n, m)))
    U0 = np.asfortranarray(rng.normal(0.0, 0.001, size=(m, w.K)))
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test, ctns_confounder=Z)
    A, C = _cp(w)
    got = ds.optimize(A + [U0.copy(order="F")], C, w.K, w.lam, w.lam, w.alpha, tuning=w.tuning, max_iter=20, seed=5,
                      inc_continuous=1)
    ds.close()
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0 + [U0], w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha,
                          tuning=w.tuning, max_iter=20, seed=5, ctns=Z)
    assert got["iters"] == ref["iters"]
    np.testing.assert_allclose(got["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-8, equal_nan=True)
    for i, a in enumerate(ref["row_matrices"]):
        assert relerr(got["row_matrices"][f"factor{i}"], a) < 1e-6, i
    assert relerr(got["column_factor"], ref["column_factor"]) < 1e-6


@pytest.mark.parametrize("L", [700, 830, 1200])
def test_many_level_covariate_with_continuous_column_vs_oracle(oracle, L):
    """A covariate with so many levels that k_gene_u_cnt's per-wave LDS record (V row + level sums + partials, four waves per
    block) does not fit 64 KB (from about 770 levels on), together with a continuous covariate: the merged row update may
    then not fall back to k_gene_u, which knows nothing of the continuous columns' term z_r' A_c c_j — the data set has to
    take the per-sample path as a whole (decided at insider_hip_create_ex).  L = 700 still runs the merged form."""
    w = workloads.small(n=2 * L + 40, p=48, level_counts=(L, 3), K=4)
    rng = np.random.default_rng(L)
    Z = np.asfortranarray(rng.standard_normal((w.n, 1)))
    U0 = np.asfortranarray(rng.normal(0.0, 0.001, size=(1, w.K)))
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test, ctns_confounder=Z)
    ds.set_option("profile", 1)
    A, C = _cp(w)
    got = ds.optimize(A + [U0.copy(order="F")], C, w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=10, seed=5, inc_continuous=1)
    merged = ds.profile()["row_merged"]
    ds.close()
    assert merged == (L == 700)
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0 + [U0], w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, tuning=1,
                          max_iter=10, seed=5, ctns=Z)
    np.testing.assert_allclose(got["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-8, equal_nan=True)
    for i, a in enumerate(ref["row_matrices"]):
        assert relerr(got["row_matrices"][f"factor{i}"], a) < 1e-6, i
    assert relerr(got["column_factor"], ref["column_factor"]) < 1e-6


def test_operator_level_optimize_with_ctns(oracle):
    # the reference's 16-argument optimize() with inc_continuous = 1 (R/RcppExports.R:20-22)
    w = workloads.small(n=50, p=60)
    rng = np.random.default_rng(2)
    Z = np.asfortranarray(rng.standard_normal((w.n, 2)))
    U0 = np.asfortranarray(rng.normal(0.0, 0.001, size=(2, w.K)))
    A, C = _cp(w)
    out = api.optimize(w.X, A + [U0.copy(order="F")], C, w.levels, Z, w.M_train, w.M_test, 1, w.K, w.lam, w.lam, w.alpha, 1,
                       1e-10, 1e-5, 5, seed=3)
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0 + [U0], w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha,
                          max_iter=5, seed=3, ctns=Z)
    assert relerr(out["column_factor"], ref["column_factor"]) < 1e-7
    assert relerr(out["row_matrices"]["factor2"], ref["row_matrices"][2]) < 1e-7


def test_caller_level_insider_tune_fit(oracle, tmp_path):
    """The R-level workflow mirrored in insider_amd/api.py (R/insider.R:18-216): insider() -> tune() grid -> fit(),
    against the oracle driven with the same masks and the same fresh inits per grid point."""
    rng = np.random.default_rng(4)
    conf = workloads.cyclic_levels(60, (5, 3))
    A = [rng.standard_normal((5, 4)), rng.standard_normal((3, 4))]
    Cs = rng.standard_normal((4, 80))
    data = sum(A[i][conf[:, i] - 1] for i in range(2)) @ Cs + 0.5 * rng.standard_normal((60, 80))
    data[rng.random(data.shape) < 0.03] = np.nan
    obj = api.insider(data, conf, split_ratio=0.15, tuning_iter=12, max_iter=25, seed=99)
    assert not (obj["train_indicator"] & obj["test_indicator"]).any()
    lam, alp = [1.0, 3.0], [0.2, 0.5]
    out = api.tune(obj, latent_dimension=np.array([4]), lambda_=lam, alpha=alp, out_dir=str(tmp_path),
                   rng=np.random.default_rng(7))
    assert out["latent_rank"] == 4 and out["reg_tuning"].shape == (4, 4)
    assert (tmp_path / "insider_R4_reg_tuning_result.csv").exists()           # R/insider.R:172
    # expand.grid(lambda, alpha): lambda varies fastest (R/insider.R:145)
    assert [tuple(r[:2]) for r in out["reg_tuning"]] == [(1.0, 0.2), (3.0, 0.2), (1.0, 0.5), (3.0, 0.5)]
    ref_rng = np.random.default_rng(7)
    n_levels = np.array([5, 3], dtype=np.int32)
    for row in out["reg_tuning"]:
        cfd, col = api._fresh_inits(obj, 4, ref_rng)                            # same draws as tune() made
        ref = oracle.optimize(obj["data"], obj["confounder"], n_levels, cfd, col, obj["train_indicator"],
                              obj["test_indicator"], row[0], row[0], row[1], tuning=1, global_tol=1e-9, sub_tol=1e-5,
                              max_iter=12, seed=99)
        assert row[2] == pytest.approx(ref["train_rmse"], rel=1e-8) and row[3] == pytest.approx(ref["test_rmse"], rel=1e-8)
    # fit(): indicator = train + test, "test" = NA mask, tuning = partition (R/insider.R:207-209)
    obj = api.fit(obj, latent_dimension=4, lambda_=3.0, alpha=0.2, partition=1, rng=np.random.default_rng(11))
    cfd, col = api._fresh_inits(obj, 4, np.random.default_rng(11))
    ref = oracle.optimize(obj["data"], obj["confounder"], n_levels, cfd, col,
                          obj["train_indicator"] + obj["test_indicator"], obj["na_indicator"], 3.0, 3.0, 0.2, tuning=1,
                          global_tol=1e-9, sub_tol=1e-5, max_iter=25, seed=99)
    assert relerr(obj["column_factor"], ref["column_factor"]) < 1e-6
    assert obj["test_rmse"] == pytest.approx(ref["test_rmse"], rel=1e-7)
    assert set(obj["cfd_matrices"]) == {"factor0", "factor1"}


# ---- the block updates as stand-alone operators (optimize_row / optimize_col / fit_interaction arithmetic) ----------
def _residual_without(w, A, C, cov):
    """X minus the contribution of every covariate except `cov` (what optimize() hands optimize_row, :337-339)."""
    others = sum(A[i][w.levels[:, i] - 1, :] for i in range(w.levels.shape[1]) if i != cov)
    return w.X - (others @ C if not np.isscalar(others) else 0.0)


@pytest.mark.parametrize("tuning", [1, 0])
@pytest.mark.parametrize("kw", [dict(K=7), dict(K=20, n=130, p=110, level_counts=(9, 4, 3)), dict(K=33, with_na=True)])
def test_optimize_row_operator(oracle, kw, tuning):
    w = workloads.small(seed=41, **kw)
    A, C = _rand_factors(w, 3)
    M = w.M_train if tuning == 1 else np.ones_like(w.M_train)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    for cov in range(w.levels.shape[1]):
        before = [a.copy() for a in A]
        got = ds.optimize_row([a.copy(order="F") for a in A], C, cov, lambda_=w.lam, tuning=tuning)
        ref = oracle.optimize_row(_residual_without(w, before, C, cov), M, before[cov], C, w.levels[:, cov], C @ C.T,
                                  w.lam, tuning=tuning)
        assert relerr(got, ref) < 1e-9, (cov, relerr(got, ref))
    ds.close()


def test_optimize_row_lambda_zero_is_fit_interaction(oracle):
    """fit_interaction (src/fit_interaction.cpp:36-56): per interaction level, solve(sum C_nz C_nz', sum C_nz resid_nz)
    with NO ridge term, on a residual that excludes the interaction factor itself."""
    w = workloads.small(n=160, p=90, level_counts=(4, 3), K=5, seed=77, f=0.1)
    levels = workloads.interaction_indicator(w.levels, [1, 2])      # interaction inserted 2nd (R/insider.R:40)
    rng = np.random.default_rng(5)
    n_levels = [int(levels[:, i].max()) for i in range(3)]
    A = [np.asfortranarray(rng.standard_normal((L, w.K)) * 0.3) for L in n_levels]
    C = np.asfortranarray(rng.standard_normal((w.K, w.X.shape[1])))
    ds = api.InsiderData(w.X, levels, w.M_train, w.M_test)
    got = ds.optimize_row([a.copy(order="F") for a in A], C, 1, lambda_=0.0, tuning=1)
    ds.close()
    resid = w.X - (A[0][levels[:, 0] - 1] + A[2][levels[:, 2] - 1]) @ C
    want = np.zeros_like(A[1])
    for l in range(n_levels[1]):                                   # direct form, as fit_interaction.cpp:44-54 states it
        G = np.zeros((w.K, w.K))
        q = np.zeros(w.K)
        for r in np.nonzero(levels[:, 1] == l + 1)[0]:
            nz = w.M_train[r] != 0
            G += C[:, nz] @ C[:, nz].T
            q += C[:, nz] @ resid[r, nz]
        want[l] = np.linalg.solve(G, q)
    assert relerr(got, want) < 1e-9
    ref = oracle.optimize_row(resid, w.M_train, A[1], C, levels[:, 1], C @ C.T, 0.0, tuning=1)
    assert relerr(got, ref) < 1e-9


@pytest.mark.parametrize("kw,m", [(dict(level_counts=(100, 10), n=600, p=70, K=30, f=0.1), 0),     # c3's structure: 7 blocks of 16 levels
                                  (dict(level_counts=(50, 5), n=400, p=66, K=20, f=0.1), 0),       # c2's
                                  (dict(level_counts=(37, 9, 3), n=500, p=41, K=15, f=0.2), 0),    # one 16 x 16 block, three covariates
                                  (dict(level_counts=(21,), n=200, p=30, K=31, f=0.2), 0),         # no later covariate: no count product
                                  (dict(level_counts=(9, 8, 7, 6), n=700, p=37, K=9, f=0.2), 0),   # six k-steps, two count dwords per lane
                                  (dict(level_counts=(64, 4), n=500, p=50, K=16, f=0.15, with_na=True), 2),   # continuous columns
                                  (dict(level_counts=(3, 2), n=60, p=9, K=4, f=0.3), 1)],
                         ids=["c3", "c2", "three-cov", "one-cov", "six-steps", "ctns2", "tiny-ctns1"])
def test_pair_count_statistics_on_the_4x4x4_matrix_instruction(oracle, kw, m):
    """k_col_paircnt4 (option col_mfma4 = 1, the default for K <= 31 when the factor rows of all covariates fit LDS): the second
    product of the pair-count statistics on v_mfma_f64_4x4x4 with the rows read in four rotations from LDS, blocks that stay
    resident and walk the genes.  Same fits as k_col_paircnt (col_mfma4 = 0) to rounding, both against the oracle; p is not a
    multiple of four and smaller than the resident grid."""
    w = workloads.small(seed=123, **kw)
    rng = np.random.default_rng(5)
    Z = np.asfortranarray(rng.standard_normal((w.n, m))) if m else None
    U0 = [np.asfortranarray(rng.normal(0.0, 0.001, size=(m, w.K)))] if m else []
    out = {}
    for fine in (1, 0):
        ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test, ctns_confounder=Z)
        for k, v in PATHS["pair"].items():
            ds.set_option(k, v)
        ds.set_option("col_mfma4", fine)
        A, C = _cp(w)
        out[fine] = ds.optimize(A + [u.copy(order="F") for u in U0], C, w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=10, seed=4,
                                inc_continuous=1 if m else 0)
        assert ds.profile()["col_pair"]
        ds.close()
    ref = oracle.optimize(w.X, w.levels, w.n_levels, w.A0 + U0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, tuning=1,
                          max_iter=10, seed=4, **(dict(ctns=Z) if m else {}))
    for fine in (1, 0):
        np.testing.assert_allclose(out[fine]["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-8, equal_nan=True)
        assert relerr(out[fine]["column_factor"], ref["column_factor"]) < 1e-6
    np.testing.assert_allclose(out[1]["traj"][:, 1:8], out[0]["traj"][:, 1:8], rtol=1e-11, equal_nan=True)
    assert relerr(out[1]["column_factor"], out[0]["column_factor"]) < 1e-9


@pytest.mark.parametrize("alpha,tuning", [(0.4, 1), (0.0, 1), (0.3, 0), (0.0, 0), (1.0, 1)])
def test_optimize_col_operator(oracle, alpha, tuning):
    w = workloads.small(K=12, n=140, p=100, seed=43, with_na=True)
    A, C = _rand_factors(w, 9)
    M = w.M_train if tuning == 1 else np.ones_like(w.M_train)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    C_in = C.copy(order="F")
    got = ds.optimize_col(A, C_in, lambda_=w.lam, alpha=alpha, tuning=tuning, tol=1e-9, seed=17, it=4)
    sw = ds.sweeps()
    ds.close()
    assert np.array_equal(got, C_in)                              # updated in place, like the reference's mat&
    ref, ref_sw = oracle.optimize_col(w.X, M, _R(w, A), C, w.lam, alpha, tuning=tuning, tol=1e-9, seed=17, it=4)
    assert relerr(got, ref) < 1e-8, relerr(got, ref)
    if alpha > 0:
        assert abs(int(sw.sum()) - ref_sw) <= max(2, ref_sw // 200)


def test_operators_reject_bad_arguments():
    w = workloads.small(K=4, n=40, p=30)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    A, C = _cp(w)
    with pytest.raises(_lib.InsiderError) as e:
        ds.optimize_row(A, C, 5, lambda_=1.0)
    assert e.value.status == _lib.ERR_ARG
    with pytest.raises(_lib.InsiderError):
        ds.optimize_row(A, C, 0, lambda_=float("nan"))
    with pytest.raises(_lib.InsiderError):
        ds.optimize_col(A, C, tuning=2)
    ds.close()


@pytest.mark.parametrize("K,pairs", [(7, 0), (30, 0), (30, 1), (23, 1), (7, 1)])
def test_device_order_table_matches_golden(K, pairs):
    """The sweep-order table the device kernels read (k_order_table) against the committed golden orders of
    include/insider_perm.h (tests/golden/perm_golden.json): oracle and product share that header, so only fixed bytes can
    catch an edit of it.  Also decodes the successor list the register-resident sweep kernel jumps through."""
    import ctypes as C
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "perm_golden.json")))
    w = workloads.small(K=K, n=60, p=20, seed=3)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    ds.set_option("cd_pairs", pairs)
    checked = 0
    have = {g["K"] for g in gold}
    for seed, it in sorted({(g["seed"], g["iter"]) for g in gold}):
        A, Cm = _cp(w)
        ds.optimize_col(A, Cm, lambda_=w.lam, alpha=0.4, tuning=1, tol=1e-3, seed=seed, it=it)
        rows = 16384 + 1
        ROW = 448                                       # ORDER_ROW: bytes per sweep (insider_kernels.hpp)
        tab = np.zeros(rows * ROW, dtype=np.uint8)
        _lib.check(_lib.load().insider_hip_get_array(ds._h, b"order_table", tab.ctypes.data_as(C.c_void_p), tab.nbytes))
        tab = tab.reshape(rows, ROW)
        for g in gold:
            if (g["seed"], g["iter"]) != (seed, it) or g["K"] != (K if K in have else 30):
                continue
            row = tab[g["sweep"] % 16384]
            if K not in have:      # no golden orders for this K: the row's own order bytes are the walk's reference
                g = dict(g, order=row[:K].tolist())
                assert sorted(g["order"]) == list(range(K))
            assert row[:K].tolist() == g["order"], g
            if K <= 30:
                # successor list (bytes 128..): 64-bit ABSOLUTE code addresses — entry 0 = first block, entry 1 + k = block after
                # coordinate k, the exit block (index KMAX) last; a block's address = table base + 64 * its index, and every sweep
                # visits coordinate 0, whose block IS the base
                kmax = 16 if K <= 16 else (K + 1) & ~1
                pr = row[128:128 + 8 * (1 + kmax)].view(np.uint64)
                if pairs:
                    # routed through the blocks of TWO steps (option cd_pairs): entry 1 + l = the block after the block that ENDS
                    # with coordinate l; pair (a, b) of slot 0 at pair_base + 128 (16 a + b), of slot 1 at pair_base + 128 (256 +
                    # W (a - 16) + (b - 16)), W = kmax - 16; single blocks and the exit block in the table of single steps.  The
                    # exit block is the LARGEST address of the single table (the entries of k >= K and of the last block hold it)
                    exit_addr = int(pr[1 + g["order"][-1]])
                    base = exit_addr - 64 * kmax
                    W, walk, cur, jumps = kmax - 16, [], int(pr[0]), 0
                    pbase = 0

                    def decode(addr, pb):
                        idx, rem = divmod(addr - pb, 128)
                        if rem or idx < 0:
                            return None
                        if idx < 256:
                            return [idx // 16, idx % 16]
                        idx -= 256
                        return [16 + idx // W, 16 + idx % W] if W > 0 and idx < W * W else None
                    greedy, t = 0, 0
                    while t < K:
                        t += 2 if (t + 1 < K and g["order"][t] // 16 == g["order"][t + 1] // 16) else 1
                        greedy += 1
                    # the pair base: the address of the first pair of the greedy cut fixes it
                    t = 0
                    while t < K and not (t + 1 < K and g["order"][t] // 16 == g["order"][t + 1] // 16):
                        t += 1
                    if t < K:
                        a, b = g["order"][t], g["order"][t + 1]
                        first_pair_addr = int(pr[0]) if t == 0 else int(pr[1 + g["order"][t - 1]])
                        pbase = first_pair_addr - 128 * (16 * a + b if a < 16 else 256 + W * (a - 16) + (b - 16))
                    while cur != exit_addr:
                        if base <= cur < exit_addr:
                            assert (cur - base) % 64 == 0
                            walk.append((cur - base) // 64)
                        else:
                            ab = decode(cur, pbase)
                            assert ab is not None and ab[0] // 16 == ab[1] // 16 and ab[0] != ab[1], (cur, pbase)
                            walk += ab
                        jumps += 1
                        assert len(walk) <= K
                        cur = int(pr[1 + walk[-1]])
                    assert walk == g["order"] and jumps == greedy, (walk, g, jumps, greedy)
                    checked += 1
                    continue
                base, unit = int(pr.min()), 64          # INSIDER_REG_BLOCK
                assert base > 0 and base % 4 == 0
                blk = [(int(v) - base) // unit for v in pr]
                assert all((int(v) - base) % unit == 0 for v in pr) and max(blk) == kmax
            elif K <= 32:
                blk = [int(v) // 64 for v in row[128:128 + 4 * (1 + K)].view(np.uint32)]      # K = 31, 32: 32-bit block offsets
            else:
                # 32 < K <= 48: 32-bit block offsets from byte 124 on (INSIDER_REG3_BLOCK = 80 bytes apart)
                blk = [int(v) // 80 for v in row[124:124 + 4 * (1 + K)].view(np.uint32)]
            walk, cur = [], blk[0]
            while len(walk) < K:
                walk.append(cur)
                cur = blk[1 + cur]
            assert walk == g["order"], (walk, g)
            checked += 1
        assert np.array_equal(tab[16384], tab[0])      # the look-ahead row = sweep 16384 = sweep 0 of the next period
    ds.close()
    assert checked == 72
