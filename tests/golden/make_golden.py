"""Generates tests/golden/insider_golden.npz from the CPU oracle (oracle/insider_oracle.c).

The reference ships no golden vectors and cannot be run here (parity unpinned), so these
fixtures freeze the ORACLE's outputs on small seeded inputs: they guard the oracle against regressions and give
the GPU parity tests a target that does not depend on the oracle being rebuilt on the GPU box.
Inputs are regenerated from seeds by insider_amd.workloads (numpy PCG64); only the expected outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from insider_amd import workloads  # noqa: E402
from oracle import c_oracle  # noqa: E402

CASES = {
    "masked": dict(n=48, p=72, level_counts=(6, 4), K=5, f=0.15, seed=101),
    "masked_na_interaction": dict(n=60, p=50, level_counts=(5, 3), K=7, f=0.2, seed=102, with_na=True,
                                  interaction_idx=(1, 2)),
    "unmasked": dict(n=40, p=64, level_counts=(4, 5), K=4, f=0.1, seed=103, tuning=0),
    "ridge": dict(n=40, p=48, level_counts=(4, 2), K=6, f=0.1, seed=104, alpha=0.0),
    # BASELINE config 5's structure in small: three covariates plus the interaction of columns 1 and 2 inserted second
    "c5_structure": dict(n=240, p=96, level_counts=(6, 4, 5), K=9, f=0.1, seed=105, interaction_idx=(1, 2)),
}
MAX_ITER, SEED = 20, 77


def run(name):
    w = workloads.small(**CASES[name])
    res = c_oracle.optimize(w.X, w.levels, w.n_levels, w.A0, w.C0, w.M_train, w.M_test, w.lam, w.lam, w.alpha,
                            tuning=w.tuning, max_iter=MAX_ITER, seed=SEED)
    return w, res


if __name__ == "__main__":
    # existing cases are kept as committed (the oracle's OpenMP reductions differ in the last bits from run to run);
    # `python tests/golden/make_golden.py --all` regenerates everything
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "insider_golden.npz")
    out = dict(np.load(path)) if os.path.exists(path) and "--all" not in sys.argv else {}
    for name in CASES:
        if name + "/traj" in out:
            continue
        w, res = run(name)
        out[name + "/traj"] = res["traj"]
        out[name + "/C"] = res["column_factor"]
        for i, a in enumerate(res["row_matrices"]):
            out[f"{name}/A{i}"] = a
        out[name + "/scalars"] = np.array([res["train_rmse"], res["test_rmse"], res["loss"], res["iters"],
                                           res["total_sweeps"]], dtype=float)
        # a checksum of the inputs so that a change of the generator is detected rather than silently compared
        out[name + "/input_sum"] = np.array([w.X.sum(), float(w.M_train.sum()), float(w.M_test.sum()),
                                             float(w.levels.sum()), w.C0.sum()])
    np.savez_compressed(path, **out)
    print("wrote", len(out), "arrays")
