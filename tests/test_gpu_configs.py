"""GPU parity at the sizes and structures BASELINE.json's configs name (c1..c5), through the C ABI.

* c2 / c3 / c5 STRUCTURE at full n: a gene slab (all samples, every level of every covariate, the real K, the real
  held-out fraction) against the CPU oracle — one outer iteration from a non-trivial start and 11 iterations from
  the N(0, 1e-6) inits, on each of the three statistic paths.  The sweep cap (max_sweeps, honoured identically by
  the HIP path and the oracle) bounds the oracle's residual-form CD to seconds.
* c1 at FULL size (377 x 5000, four covariates L = (2, 16, 8, 107), K = 23, tuning = 0, 31 iterations, no cap)
  against the oracle.
* c5 / c4 at full size: size-independent properties (monotone checkpoint losses, loss components and RMSEs
  recomputed in numpy from the returned factors, default path == per-entry list path).

Tolerances as tests/test_gpu_parity.py (fp64): one iteration rel 1e-9 on the factors, several iterations rel 1e-6,
loss trajectory rel 1e-9.
"""
import os

import numpy as np
import pytest

from insider_amd import _lib, api, workloads

pytestmark = pytest.mark.gpu

PATHS = {"fast": dict(row_merged=2, col_factored=2, row_counts=0), "pair": dict(row_merged=2, col_factored=3, row_counts=1),
         "lists": dict(row_merged=0, col_factored=0), "default": dict()}
SLAB_GENES = 192
SWEEP_CAP = 300


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X box")


def _slab(name):
    """Gene slab [0, SLAB_GENES) of BASELINE config `name` (all samples), its oracle results and a resident handle."""
    from oracle import c_oracle
    w = workloads.make(name, gene_range=(0, SLAB_GENES))
    rng = np.random.default_rng(5)
    A1 = [np.asfortranarray(rng.standard_normal(a.shape) * 0.3) for a in w.A0]
    C1 = np.asfortranarray(rng.standard_normal(w.C0.shape) * 0.3)
    c_oracle.set_col_chunk(1)          # 192 genes would otherwise be two chunks of 100 = two threads
    try:
        threads = c_oracle.num_procs()
        ref1 = c_oracle.optimize(w.X, w.levels, w.n_levels, A1, C1, w.M_train, w.M_test, w.lam, w.lam, w.alpha,
                                 tuning=1, max_iter=0, seed=17, max_sweeps=SWEEP_CAP, col_threads=threads,
                                 row_threads=threads)
        ref11 = c_oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha,
                                  tuning=1, max_iter=10, seed=23, max_sweeps=SWEEP_CAP, col_threads=threads,
                                  row_threads=threads)
    finally:
        c_oracle.set_col_chunk(100)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    return dict(w=w, A1=A1, C1=C1, ref1=ref1, ref11=ref11, ds=ds)


@pytest.fixture(scope="module")
def c3_slab():
    s = _slab("c3")
    yield s
    s["ds"].close()


@pytest.fixture(scope="module")
def c5_slab():
    s = _slab("c5")
    yield s
    s["ds"].close()


@pytest.fixture(scope="module")
def c2_slab():
    s = _slab("c2")
    yield s
    s["ds"].close()


def _check_slab(s, paths, expect_levels):
    w, ds = s["w"], s["ds"]
    assert w.n == workloads.CONFIGS[w.name][0] and list(w.n_levels) == expect_levels and w.K == workloads.CONFIGS[w.name][4]
    for k in ("row_merged", "col_factored", "row_counts"):          # back to the defaults, then this path's choices
        ds.set_option(k, 1)
    for k, v in PATHS[paths].items():
        ds.set_option(k, v)
    ds.set_option("max_sweeps", SWEEP_CAP)
    got = ds.optimize([a.copy(order="F") for a in s["A1"]], s["C1"].copy(order="F"), w.K, w.lam, w.lam, w.alpha,
                      tuning=1, max_iter=0, seed=17)
    ref = s["ref1"]
    for i, a in enumerate(ref["row_matrices"]):
        assert relerr(got["row_matrices"][f"factor{i}"], a) < 1e-9, (paths, i)
    assert relerr(got["column_factor"], ref["column_factor"]) < 1e-9
    np.testing.assert_allclose(got["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-9, equal_nan=True)
    got = ds.optimize([a.copy(order="F") for a in w.A0], w.C0.copy(order="F"), w.K, w.lam, w.lam, w.alpha,
                      tuning=1, max_iter=10, seed=23)
    ref = s["ref11"]
    assert got["iters"] == ref["iters"] == 11 and list(got["traj"][:, 0]) == [-1, 0, 10]
    np.testing.assert_allclose(got["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-9, equal_nan=True)
    assert np.array_equal(got["traj"][:, 9], ref["traj"][:, 9])      # same decay schedule
    for i, a in enumerate(ref["row_matrices"]):
        assert relerr(got["row_matrices"][f"factor{i}"], a) < 1e-6, (paths, i)
    assert relerr(got["column_factor"], ref["column_factor"]) < 1e-6


@pytest.mark.parametrize("paths", list(PATHS))
def test_c3_structure_full_n_slab_vs_oracle(c3_slab, paths):
    """n = 10000 samples, 100 x 10 levels, K = 30, 10 % held out (byte-packed pair counts, 2x2 MFMA blocks)."""
    _check_slab(c3_slab, paths, [100, 10])


@pytest.mark.parametrize("paths", list(PATHS))
def test_c5_structure_full_n_slab_vs_oracle(c5_slab, paths):
    """n = 5000, 3 covariates + interaction(1, 2) = 200 levels inserted second (R/insider.R:34-40), K = 25."""
    _check_slab(c5_slab, paths, [20, 200, 10, 25])


@pytest.mark.parametrize("paths", list(PATHS))
def test_c2_structure_full_n_slab_vs_oracle(c2_slab, paths):
    """BASELINE config 2's own structure: n = 2000 samples, 50 x 5 levels, K = 20 (the four-waves-per-SIMD instantiation of
    the sweep kernel, KMAX = 20; 2 x 2 MFMA blocks with 11 padding coordinates), 10 % held out."""
    _check_slab(c2_slab, paths, [50, 5])


def test_c3_structure_deep_sweeps_vs_oracle():
    """The deep-sweep regime at real size, which the capped slab tests above never reach: c3's structure at full n (10000
    samples, 100 x 10 levels, K = 30, 10 % held out), 64 genes, the reference's cold N(0, 1e-6) inits, two outer
    iterations with the sweep cap at 20000 — the first iteration of such a call runs thousands of sweeps per gene through
    the multi-pass continuation [64, 256) -> [256, 1024) -> [1024, 4096) -> [4096, end) of the default path.  Same sweeps
    (total within 1 per solve), same factors, same trajectory as the oracle's residual-form CD under the same cap.  (A
    64-gene slab is a badly conditioned problem of its own — row factors estimated from 64 genes — and some of its solves
    do not converge in 100000 sweeps on either side; the full-size runs of c2 .. c5 assert that no solve reaches the cap.)"""
    from oracle import c_oracle
    genes, cap = 64, 20000
    w = workloads.make("c3", gene_range=(0, genes))
    assert w.n == 10000 and list(w.n_levels) == [100, 10] and w.K == 30
    c_oracle.set_col_chunk(1)
    try:
        threads = c_oracle.num_procs()
        ref = c_oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, tuning=1,
                                max_iter=1, seed=41, max_sweeps=cap, col_threads=threads, row_threads=threads)
    finally:
        c_oracle.set_col_chunk(100)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    ds.set_option("profile", 1)
    ds.set_option("max_sweeps", cap)
    got = ds.optimize([a.copy(order="F") for a in w.A0], w.C0.copy(order="F"), w.K, w.lam, w.lam, w.alpha, tuning=1,
                      max_iter=1, seed=41)
    prof, longest = ds.profile(), int(ds.info("max_gene_sweeps"))
    ds.close()
    assert longest > 4096, longest                                        # the last pass range [4096, end) really ran
    assert ref["total_sweeps"] > 2000 * genes                            # thousands of sweeps per gene in this call
    assert abs(prof["sweeps"] - ref["total_sweeps"]) <= 2 * genes, (prof["sweeps"], ref["total_sweeps"])   # +-1 per solve
    assert got["iters"] == ref["iters"] == 2
    np.testing.assert_allclose(got["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-9, equal_nan=True)
    for i, a in enumerate(ref["row_matrices"]):
        assert relerr(got["row_matrices"][f"factor{i}"], a) < 1e-6, i
    assert relerr(got["column_factor"], ref["column_factor"]) < 1e-6


def test_c1_full_size_vs_oracle():
    """BASELINE config 1 (README.md:91-118 shapes): 377 x 5000, L = (2, 16, 8, 107), K = 23, lambda = 10, alpha = 0.4,
    fit()'s unmasked path (tuning = 0), 31 outer iterations, no sweep cap — the whole configuration against the oracle."""
    from oracle import c_oracle
    w = workloads.make("c1")
    assert (w.n, w.p, w.K, w.tuning) == (377, 5000, 23, 0) and list(w.n_levels) == [2, 16, 8, 107]
    threads = c_oracle.num_procs()
    ref = c_oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, tuning=0,
                            max_iter=30, seed=29, col_threads=threads, row_threads=threads)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    ds.set_option("profile", 1)
    got = ds.optimize([a.copy(order="F") for a in w.A0], w.C0.copy(order="F"), w.K, w.lam, w.lam, w.alpha, tuning=0,
                      max_iter=30, seed=29)
    sweeps = ds.profile()["sweeps"]
    ds.close()
    assert got["iters"] == ref["iters"] == 31 and list(got["traj"][:, 0]) == [-1, 0, 10, 20, 30]
    np.testing.assert_allclose(got["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-9, equal_nan=True)
    assert np.array_equal(got["traj"][:, 9], ref["traj"][:, 9])
    for i, a in enumerate(ref["row_matrices"]):
        assert relerr(got["row_matrices"][f"factor{i}"], a) < 1e-6, i
    assert relerr(got["column_factor"], ref["column_factor"]) < 1e-6
    assert np.isnan(got["test_rmse"])                                    # uninitialised in the reference (:264)
    assert abs(sweeps - ref["total_sweeps"]) <= 0.002 * ref["total_sweeps"]


def _full_size_properties(name, iters, compare_lists):
    w = workloads.make(name)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    ds.set_option("profile", 1)
    got = ds.optimize([a.copy(order="F") for a in w.A0], w.C0.copy(order="F"), w.K, w.lam, w.lam, w.alpha, tuning=1,
                      max_iter=iters - 1, seed=3)
    prof = ds.profile()
    assert int(ds.info("cap_hits")) == 0, (name, ds.info("max_gene_sweeps"))   # no solve ended at the sweep cap
    if compare_lists:     # the per-entry list kernels give the same trajectory as whatever the cost models chose
        ds.set_option("row_merged", 0)
        ds.set_option("col_factored", 0)
        ref = ds.optimize([a.copy(order="F") for a in w.A0], w.C0.copy(order="F"), w.K, w.lam, w.lam, w.alpha, tuning=1,
                          max_iter=iters - 1, seed=3)
        np.testing.assert_allclose(got["traj"][:, 1:8], ref["traj"][:, 1:8], rtol=1e-9, equal_nan=True)
        assert relerr(got["column_factor"], ref["column_factor"]) < 1e-7
    ds.close()
    tr = got["traj"]
    assert np.all(np.diff(tr[:, 7]) < 0)                                 # checkpoint losses decrease
    A = [got["row_matrices"][f"factor{i}"] for i in range(len(w.A0))]
    C = got["column_factor"]
    R = sum(A[i][w.levels[:, i] - 1, :] for i in range(w.levels.shape[1]))
    sse = sse_te = 0.0
    ntr = nte = 0
    for b in range(0, w.p, 5000):                                        # gene blocks: bounded host memory
        resid = w.X[:, b:b + 5000] - R @ C[:, b:b + 5000]
        mtr, mte = w.M_train[:, b:b + 5000] != 0, w.M_test[:, b:b + 5000] != 0
        sse += float(np.sum(resid[mtr] ** 2))
        sse_te += float(np.sum(resid[mte] ** 2))
        ntr += int(mtr.sum())
        nte += int(mte.sum())
    assert tr[-1, 3] == pytest.approx(sse / 2, rel=1e-10)
    assert tr[-1, 4] == pytest.approx(w.lam * sum(np.sum(a ** 2) for a in A) / 2, rel=1e-12)
    assert tr[-1, 5] == pytest.approx(w.lam * (1 - w.alpha) * np.sum(C ** 2) / 2, rel=1e-12)
    assert tr[-1, 6] == pytest.approx(w.lam * w.alpha * np.sum(np.abs(C)), rel=1e-12)
    assert got["test_rmse"] == pytest.approx(np.sqrt(sse_te / nte), rel=1e-10)
    assert got["train_rmse"] == pytest.approx(np.sqrt(sse / ntr), rel=1e-10)
    assert got["test_rmse"] < 1.2                                        # noise sd is 1: the fit generalises
    return prof


def test_c5_full_size_properties():
    """BASELINE config 5 at full size: 5000 x 50000, 3 covariates + the 200-level interaction, K = 25."""
    _full_size_properties("c5", 11, compare_lists=True)


def test_c4_full_size_properties():
    """BASELINE config 4 at full size on ONE GPU: 10000 x 200000, K = 30 (16 GB of X; the multi-GPU configuration's
    whole problem as a single slab)."""
    import psutil
    free = psutil.virtual_memory().available
    if free < 40e9:      # X (16 GB) + two masks + the generator's blocks + the residual check: do not drive a small host out of memory
        pytest.skip(f"needs ~40 GB of free host memory, {free / 1e9:.0f} GB available")
    _full_size_properties("c4", 11, compare_lists=False)


def test_c3_structure_full_n_slab_with_continuous_covariates_vs_oracle():
    """Continuous covariates at real size (optimize_continuous_v2, src/optimize.cpp:76-137,340-351): c3's
    structure at full n (10000 samples, 100 x 10 levels, K = 30, 10 % held out), a 192-gene slab, plus TWO N(0, 1) columns
    of ctns_confounder — one outer iteration from a non-trivial start (factors 1e-9) and three from the cold inits (1e-6,
    trajectory 1e-9) against the oracle's literal scalar passes over the n x p residual."""
    from oracle import c_oracle
    w = workloads.make("c3", gene_range=(0, SLAB_GENES))
    m = 2
    Z = np.asfortranarray(np.random.default_rng(77).standard_normal((w.n, m)))
    rng = np.random.default_rng(5)
    A1 = [np.asfortranarray(rng.standard_normal(a.shape) * 0.3) for a in w.A0] + [np.asfortranarray(rng.standard_normal((m, w.K)) * 0.3)]
    C1 = np.asfortranarray(rng.standard_normal(w.C0.shape) * 0.3)
    A0 = [a.copy(order="F") for a in w.A0] + [np.asfortranarray(np.random.default_rng(9).standard_normal((m, w.K)) * 1e-3)]
    c_oracle.set_col_chunk(1)
    try:
        # the cores this process may really use (affinity, cgroup quota): the continuous update's scalar passes are barrier-bound,
        # and an OpenMP team larger than the CPU share (omp_get_num_procs sees the whole host) spins for minutes
        threads = len(os.sched_getaffinity(0))
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                threads = min(threads, max(1, int(int(quota) / int(period))))
        except Exception:
            pass
        threads = max(1, min(threads, 16))
        ref1 = c_oracle.optimize(w.X, w.levels, w.n_levels, A1, C1, w.M_train, w.M_test, w.lam, w.lam, w.alpha, tuning=1, max_iter=0,
                                 seed=17, max_sweeps=SWEEP_CAP, col_threads=threads, row_threads=threads, ctns=Z)
        ref4 = c_oracle.optimize(w.X, w.levels, w.n_levels, A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, tuning=1, max_iter=2,
                                 seed=23, max_sweeps=SWEEP_CAP, col_threads=threads, row_threads=threads, ctns=Z)
    finally:
        c_oracle.set_col_chunk(100)
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test, ctns_confounder=Z)
    ds.set_option("max_sweeps", SWEEP_CAP)
    got = ds.optimize([a.copy(order="F") for a in A1], C1.copy(order="F"), w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=0, seed=17,
                      inc_continuous=1)
    assert len(ref1["row_matrices"]) == 3
    for i, a in enumerate(ref1["row_matrices"]):
        assert relerr(got["row_matrices"][f"factor{i}"], a) < 1e-9, i
    assert relerr(got["column_factor"], ref1["column_factor"]) < 1e-9
    np.testing.assert_allclose(got["traj"][:, 1:8], ref1["traj"][:, 1:8], rtol=1e-9, equal_nan=True)
    got = ds.optimize([a.copy(order="F") for a in A0], w.C0.copy(order="F"), w.K, w.lam, w.lam, w.alpha, tuning=1, max_iter=2, seed=23,
                      inc_continuous=1)
    ds.close()
    assert got["iters"] == ref4["iters"] == 3
    np.testing.assert_allclose(got["traj"][:, 1:8], ref4["traj"][:, 1:8], rtol=1e-9, equal_nan=True)
    for i, a in enumerate(ref4["row_matrices"]):
        assert relerr(got["row_matrices"][f"factor{i}"], a) < 1e-6, i
    assert relerr(got["column_factor"], ref4["column_factor"]) < 1e-6
