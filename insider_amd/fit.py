"""Command-line driver: the INSIDER fit on the MI355X from files, no R and no .RData.

    python -m insider_amd.fit --flat DIR                     # inputs written by r/insider_hip.R:insider_write_flat
    python -m insider_amd.fit --x X.npy --levels L.npy [--train-mask M.npy --test-mask T.npy] [--ctns Z.npy]
        --rank K --lambda 5 --alpha 0.4 [--partition 0|1] [--max-iter N] [--out DIR] [--out-format npy|flat]
    ... --tune --ranks 10 12 14 --lambdas 1 3 5 --alphas 0.2 0.4   # tune()'s rank sweep + lambda x alpha grid

Semantics are those of insider_amd.api (the mirror of R/insider.R): with masks given, `--partition 1` fits on the
train entries (optimize(tuning = 1)) and reports the test RMSE; without masks (or `--partition 0`) every non-NA entry is
used (fit()'s default, R/insider.R:190-216; NaN entries of X are the NA set).  Inits are N(0, 0.001^2)
(R/utils.R:40-43) from --seed.  Output: A<i> (L_i x K), C (K x p) and result.json {train_rmse, test_rmse, loss,
iters, traj} in --out.  There is no CPU fallback: without a visible MI355X the command fails with the library's status.
"""
import argparse
import json
import os
import sys

import numpy as np


def parse(argv=None):
    ap = argparse.ArgumentParser(prog="python -m insider_amd.fit", description=__doc__.split("\n\n")[0])
    ap.add_argument("--flat", help="directory in the insider-flat-1 layout (X.f64, levels.i32, train.u8, test.u8, manifest.json)")
    ap.add_argument("--x", help="n x p expression matrix (.npy / .csv); NaN = NA")
    ap.add_argument("--levels", help="n x c categorical covariates, 1-based level ids (.npy / .csv)")
    ap.add_argument("--ctns", help="n x m continuous covariates (.npy / .csv)")
    ap.add_argument("--train-mask", help="n x p 0/1 (.npy)")
    ap.add_argument("--test-mask", help="n x p 0/1 (.npy)")
    ap.add_argument("--interaction", type=int, nargs="+", help="1-based covariate columns to interact (R/insider.R:28-40)")
    ap.add_argument("--rank", type=int, help="latent dimension K")
    ap.add_argument("--lambda", dest="lam", type=float)
    ap.add_argument("--alpha", type=float)
    ap.add_argument("--partition", type=int, default=None, choices=(0, 1))
    ap.add_argument("--max-iter", type=int, default=50000)
    ap.add_argument("--global-tol", type=float, default=1e-9)
    ap.add_argument("--sub-tol", type=float, default=1e-5)
    ap.add_argument("--seed", type=int, default=0x1D5EED)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--tune", action="store_true")
    ap.add_argument("--ranks", type=int, nargs="+")
    ap.add_argument("--lambdas", type=float, nargs="+")
    ap.add_argument("--alphas", type=float, nargs="+")
    ap.add_argument("--tuning-iter", type=int, default=30)
    ap.add_argument("--warm-start", action="store_true",
                    help="--tune: start every (lambda, alpha) grid point from its nearest finished neighbour's factors "
                         "(opt-in; the reference draws fresh inits per point, R/insider.R:152-161)")
    ap.add_argument("--split-ratio", type=float, default=0.1)
    ap.add_argument("--out", default="insider_fit_out")
    ap.add_argument("--out-format", choices=("npy", "flat"), default=None)
    a = ap.parse_args(argv)
    if not a.flat and not (a.x and a.levels):
        ap.error("give --flat DIR or --x and --levels")
    if not a.tune and (a.rank is None or a.lam is None or a.alpha is None):
        ap.error("a fit needs --rank, --lambda and --alpha (or use --tune)")
    return a


def load_inputs(a):
    from . import flatio
    if a.flat:
        d = flatio.read_flat(a.flat)
        X, lev, tr, te, Z = d["X"], d["levels"], d["train"], d["test"], d["ctns"]
    else:
        X = flatio.load_matrix(a.x, np.float64)
        lev = flatio.load_matrix(a.levels).astype(np.int32).reshape(X.shape[0], -1)
        tr = flatio.load_matrix(a.train_mask) if a.train_mask else None
        te = flatio.load_matrix(a.test_mask) if a.test_mask else None
        Z = flatio.load_matrix(a.ctns, np.float64) if a.ctns else None
    return X, lev, tr, te, Z


def main(argv=None):
    a = parse(argv)
    from . import api, flatio
    X, lev, tr, te, Z = load_inputs(a)
    X = np.array(X, dtype=np.float64, order="F")
    na = np.isnan(X)
    X[na] = 0.0                                                           # R/insider.R:26
    if a.interaction:
        from .workloads import interaction_indicator
        lev = interaction_indicator(np.asarray(lev, dtype=np.int32), tuple(a.interaction))
    n, p = X.shape
    fmt = a.out_format or ("flat" if a.flat else "npy")
    if a.tune:
        # the caller-level path: insider() draws its own hold-out (R/utils.R:78-117) unless masks were given
        obj = api.insider(np.where(na, np.nan, X), lev, ctns_confounder=Z, split_ratio=a.split_ratio, global_tol=a.global_tol,
                          sub_tol=a.sub_tol, tuning_iter=a.tuning_iter, max_iter=a.max_iter, device=a.device, seed=a.seed)
        if tr is not None and te is not None:
            obj["train_indicator"] = np.asfortranarray(np.asarray(tr) != 0, dtype=np.uint8)
            obj["test_indicator"] = np.asfortranarray(np.asarray(te) != 0, dtype=np.uint8)
        res = api.tune(obj, latent_dimension=np.array(a.ranks if a.ranks else [a.rank]),
                       lambda_=a.lambdas if a.lambdas else (a.lam if a.lam is not None else 0.1),
                       alpha=a.alphas if a.alphas else (a.alpha if a.alpha is not None else 0.0), out_dir=None,
                       rng=np.random.default_rng(a.seed), warm_start=a.warm_start)
        os.makedirs(a.out, exist_ok=True)
        out = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in res.items()}
        with open(os.path.join(a.out, "tune.json"), "w") as f:
            json.dump(out, f, indent=1)
        print(json.dumps({"latent_rank": res["latent_rank"], "out": a.out}))
        return 0
    if tr is None:
        tr = ~na
    if te is None:
        te = np.zeros((n, p), dtype=bool)
    tr = np.asfortranarray(np.asarray(tr) != 0, dtype=np.uint8)
    te = np.asfortranarray(np.asarray(te) != 0, dtype=np.uint8)
    partition = a.partition if a.partition is not None else (1 if te.any() else 0)
    if partition == 0:                                                    # R/insider.R:207-208: train + test, "test" = NA
        tr, te = np.asfortranarray((tr | te) & ~na, dtype=np.uint8), np.asfortranarray(na, dtype=np.uint8)
    ds = api.InsiderData(X, lev, tr, te, device=a.device, ctns_confounder=Z)
    rng = np.random.default_rng(a.seed)
    K = a.rank
    A0 = [np.asfortranarray(api.init_parameters(int(L) * K, rng=rng).reshape((-1, K), order="F")) for L in ds.n_levels]
    if Z is not None:
        A0.append(np.asfortranarray(api.init_parameters(ds.m * K, rng=rng).reshape((-1, K), order="F")))
    C0 = np.asfortranarray(api.init_parameters(K * p, rng=rng).reshape((K, -1), order="F"))
    res = ds.optimize(A0, C0, K, a.lam, a.lam, a.alpha, tuning=partition, global_tol=a.global_tol, sub_tol=a.sub_tol,
                      max_iter=a.max_iter, seed=a.seed, inc_continuous=1 if Z is not None else 0)
    ds.close()
    summary = dict(train_rmse=res["train_rmse"], test_rmse=None if np.isnan(res["test_rmse"]) else res["test_rmse"],
                   loss=res["loss"], iters=res["iters"], rank=K, **{"lambda": a.lam}, alpha=a.alpha, partition=partition,
                   n=n, p=p, n_levels=[int(v) for v in ds.n_levels], traj=np.where(np.isnan(res["traj"]), None, res["traj"]).tolist())
    flatio.write_result(a.out, fmt, list(res["row_matrices"].values()), res["column_factor"], summary)
    print(json.dumps({k: summary[k] for k in ("train_rmse", "test_rmse", "loss", "iters")} | {"out": a.out}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
