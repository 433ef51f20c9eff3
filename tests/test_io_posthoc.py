"""the .RData-free I/O (flat binary / .npy), the command-line driver and the post-hoc regression.
CPU: formats, argument handling, loud failure without a GPU, glm_interaction against an explicit stacked least squares.
GPU: files -> `python -m insider_amd.fit` (a child process) -> files, against the CPU oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from insider_amd import flatio, posthoc, workloads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flat_round_trip(tmp_path):
    w = workloads.small(n=23, p=17, K=3, with_na=True)
    Z = np.random.default_rng(0).standard_normal((w.n, 2))
    d = flatio.write_flat(str(tmp_path / "in"), w.X, w.levels, w.M_train, w.M_test, ctns=Z)
    assert os.path.getsize(os.path.join(d, "X.f64")) == 8 * w.n * w.p           # raw column-major doubles, what R's writeBin emits
    assert np.array_equal(np.fromfile(os.path.join(d, "X.f64"))[: w.n], w.X[:, 0])
    got = flatio.read_flat(d)
    assert np.array_equal(got["X"], w.X) and np.array_equal(got["levels"], w.levels)
    assert np.array_equal(got["train"], w.M_train) and np.array_equal(got["test"], w.M_test) and np.array_equal(got["ctns"], Z)
    with pytest.raises(ValueError):
        flatio.read_raw(os.path.join(d, "X.f64"), (w.n + 1, w.p))
    flatio.write_result(str(tmp_path / "out"), "flat", w.A0, w.C0, {"loss": 1.0})
    assert np.array_equal(flatio.read_raw(str(tmp_path / "out" / "A1.f64"), w.A0[1].shape), w.A0[1])
    assert np.array_equal(flatio.read_raw(str(tmp_path / "out" / "C.f64"), w.C0.shape), w.C0)


def test_cli_arguments_and_loud_failure_without_gpu(tmp_path):
    from insider_amd import _lib, fit
    with pytest.raises(SystemExit):
        fit.parse(["--x", "a.npy"])                                   # no levels
    with pytest.raises(SystemExit):
        fit.parse(["--flat", "d"])                                    # a fit needs rank / lambda / alpha
    a = fit.parse(["--flat", "d", "--rank", "4", "--lambda", "2", "--alpha", "0.3"])
    assert (a.rank, a.lam, a.alpha, a.partition) == (4, 2.0, 0.3, None)
    if _lib.device_count() == 0:                                      # on the CPU box: the product path has no fallback
        w = workloads.small(n=12, p=10, K=2)
        d = flatio.write_flat(str(tmp_path / "in"), w.X, w.levels, w.M_train, w.M_test)
        with pytest.raises(_lib.InsiderError) as e:
            fit.main(["--flat", d, "--rank", "2", "--lambda", "1", "--alpha", "0.2", "--out", str(tmp_path / "o")])
        assert e.value.status == _lib.ERR_NO_DEVICE


def test_glm_interaction_matches_stacked_least_squares():
    """R/glm_interaction.R:2-30 builds, per level, the stacked design (t(column_factor) once per member sample) and fits
    glm(response ~ . - 1, gaussian): compare the closed form with that literal construction + textbook OLS inference."""
    from scipy import stats
    rng = np.random.default_rng(4)
    n, p, K = 30, 40, 3
    Cm = rng.standard_normal((K, p))
    ind = np.repeat(np.arange(1, 7), 5)
    truth = rng.standard_normal((6, K))
    resid = truth[ind - 1] @ Cm + 0.3 * rng.standard_normal((n, p))
    coeff, pval = posthoc.glm_interaction(resid, None, ind, Cm)
    assert coeff.shape == pval.shape == (6, K)
    for i in range(1, 7):
        ids = np.flatnonzero(ind == i)
        F = np.vstack([Cm.T] * len(ids))                              # :17-21
        yv = resid[ids].ravel()
        beta, *_ = np.linalg.lstsq(F, yv, rcond=None)
        dof = F.shape[0] - K
        s2 = np.sum((yv - F @ beta) ** 2) / dof
        se = np.sqrt(s2 * np.diag(np.linalg.inv(F.T @ F)))
        np.testing.assert_allclose(coeff[i - 1], beta, rtol=1e-10)
        np.testing.assert_allclose(pval[i - 1], 2 * stats.t.sf(np.abs(beta / se), dof), rtol=1e-8, atol=1e-300)
    assert np.max(np.abs(coeff - truth)) < 0.2 and np.all(pval[np.abs(truth) > 0.5] < 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("carrier", ["flat", "npy"])
def test_cli_round_trip_matches_oracle(tmp_path, oracle, carrier):
    from insider_amd import api
    w = workloads.small(n=50, p=64, K=4, with_na=True)
    out = str(tmp_path / "out")
    if carrier == "flat":
        d = flatio.write_flat(str(tmp_path / "in"), w.X, w.levels, w.M_train, w.M_test)
        src = ["--flat", d]
    else:
        for name, arr in (("X", w.X), ("L", w.levels), ("tr", w.M_train), ("te", w.M_test)):
            np.save(str(tmp_path / f"{name}.npy"), arr)
        src = ["--x", str(tmp_path / "X.npy"), "--levels", str(tmp_path / "L.npy"), "--train-mask", str(tmp_path / "tr.npy"),
               "--test-mask", str(tmp_path / "te.npy")]
    cmd = [sys.executable, "-m", "insider_amd.fit", *src, "--rank", str(w.K), "--lambda", str(w.lam), "--alpha", str(w.alpha),
           "--partition", "1", "--max-iter", "15", "--global-tol", "-1", "--seed", "11", "--out", out]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    # the same inits the driver drew (R/utils.R:40-43 order: every A_i, then C)
    rng = np.random.default_rng(11)
    A0 = [np.asfortranarray(api.init_parameters(int(L) * w.K, rng=rng).reshape((-1, w.K), order="F")) for L in w.n_levels]
    C0 = np.asfortranarray(api.init_parameters(w.K * w.p, rng=rng).reshape((w.K, -1), order="F"))
    ref = oracle.optimize(w.X, w.levels, w.n_levels, A0, C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha, tuning=1,
                          max_iter=15, global_tol=-1.0, seed=11)
    if carrier == "flat":
        Cg = flatio.read_raw(os.path.join(out, "C.f64"), (w.K, w.p))
        A1 = flatio.read_raw(os.path.join(out, "A1.f64"), (int(w.n_levels[1]), w.K))
    else:
        Cg, A1 = np.load(os.path.join(out, "C.npy")), np.load(os.path.join(out, "A1.npy"))
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(Cg, ref["column_factor"]) < 1e-7 and rel(A1, ref["row_matrices"][1]) < 1e-7
    res = json.load(open(os.path.join(out, "result.json")))
    assert res["iters"] == ref["iters"] == 16
    assert res["loss"] == pytest.approx(ref["loss"], rel=1e-9) and line["test_rmse"] == pytest.approx(ref["test_rmse"], rel=1e-9)
    # the post-hoc regression on the fitted residual runs on the returned factors
    A = [flatio.read_raw(os.path.join(out, f"A{i}.f64"), (int(L), w.K)) if carrier == "flat" else np.load(os.path.join(out, f"A{i}.npy"))
         for i, L in enumerate(w.n_levels)]
    R = sum(A[i][w.levels[:, i] - 1, :] for i in range(w.levels.shape[1]))
    coeff, pval = posthoc.glm_interaction(w.X - R @ Cg, w.M_train, w.levels[:, 1], Cg)
    assert coeff.shape == (int(w.n_levels[1]), w.K) and np.all((pval >= 0) & (pval <= 1))


@pytest.mark.gpu
def test_cli_tune_writes_the_grid_tables(tmp_path):
    """`--tune`: insider() + tune() of R/insider.R:18-176 from files: the rank sweep at (lambda, alpha) = (0.1, 0) picks the
    rank with the smallest test RMSE (:136), then the lambda x alpha grid (lambda fastest, :145-147) is fitted at that rank."""
    w = workloads.small(n=60, p=48, K=4)
    np.save(str(tmp_path / "X.npy"), w.X)
    np.save(str(tmp_path / "L.npy"), w.levels)
    out = str(tmp_path / "tune")
    cmd = [sys.executable, "-m", "insider_amd.fit", "--x", str(tmp_path / "X.npy"), "--levels", str(tmp_path / "L.npy"), "--tune",
           "--ranks", "2", "4", "--lambdas", "1", "3", "--alphas", "0.2", "0.5", "--tuning-iter", "6", "--seed", "5", "--out", out]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.load(open(os.path.join(out, "tune.json")))
    rank_tab, reg_tab = np.array(res["rank_tuning"]), np.array(res["reg_tuning"])
    assert rank_tab.shape == (2, 3) and list(rank_tab[:, 0]) == [2, 4]
    assert res["latent_rank"] == int(rank_tab[np.argmin(rank_tab[:, 2]), 0])
    assert reg_tab.shape == (4, 4)
    assert [tuple(v) for v in reg_tab[:, :2]] == [(1.0, 0.2), (3.0, 0.2), (1.0, 0.5), (3.0, 0.5)]      # expand.grid order
    assert np.all(np.isfinite(reg_tab[:, 2:])) and np.all(reg_tab[:, 2:] > 0)
