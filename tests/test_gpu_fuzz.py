"""Randomised parity sweeps on the GPU (tests/fuzz_parity.py, tests/fuzz_cd.py): random shapes, covariate structures,
masks, penalties and kernel forms through the C ABI against the CPU oracle.  A bounded number of cases runs with the
suite; INSIDER_FUZZ_CASES=N runs N of each (the sweeps that found the alpha = 1 zero-denominator bug ran 300-1000)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = os.environ.get("INSIDER_FUZZ_CASES", "120")


@pytest.mark.gpu
@pytest.mark.parametrize("script,seed", [("fuzz_parity.py", 11), ("fuzz_cd.py", 12)])
def test_randomised_parity_sweep(script, seed):
    r = subprocess.run([sys.executable, os.path.join(HERE, script), CASES, str(seed)], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_repeated_calls_are_bitwise_reproducible_and_leak_free():
    r = subprocess.run([sys.executable, os.path.join(HERE, "soak_determinism.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


# ---- the frozen outliers of the long randomised sweeps ---------------------------------------------------
# tests/golden/fuzz_outliers.json: cases (explicit parameters, tests/fuzz_parity.py:draw_case) whose HIP fit ends OUTSIDE the
# suite's base tolerances against the oracle (factors 1e-7, trajectory 1e-8); found by a 7000-case sweep with FUZZ_TIGHT=1
# (25 such cases: all with sub_tol <= 1e-8, most with K close to or above the number of training samples of a gene).  The claim
# these tests make executable: the deviation is the FORMULATION's, not a kernel's.  The library runs coordinate descent in
# covariance form (gradient = Xty - XtX beta), the reference and the parity oracle on the residual vector; on an
# ill-conditioned subproblem the two round differently (cancellation in Xty - XtX beta), sometimes enough to flip a stopping
# decision (|loss change| <= tol at rounding level, src/coordinate_descent.cpp:114), and the difference feeds through the
# following outer iterations.  Executable form: (i) every kernel form of the library gives the same fit; (ii) the library
# agrees with the ORACLE RUN IN COVARIANCE FORM (same C code, oracle_set_cd_form(1)) within the base tolerances, or their
# per-gene sweep counts show a stopping decision that differs; (iii) the two forms of the oracle differ from EACH OTHER — on
# the CPU alone — by as much as the library differs from the parity oracle; (iv) the deviation stays within the recorded one.
def _outliers():
    import json
    path = os.path.join(HERE, "golden", "fuzz_outliers.json")
    return json.load(open(path)) if os.path.exists(path) else []


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(len(_outliers())))
def test_fuzz_outlier_is_not_a_kernels(idx):
    import numpy as np
    sys.path.insert(0, os.path.dirname(HERE))
    from tests import fuzz_parity as fz
    case = _outliers()[idx]
    ref, rerr = fz.run_oracle(case)
    assert ref is not None, rerr
    base, err = fz.run_hip(case)
    assert base is not None, err
    dev = fz.errors(base, ref)
    assert base["iters"] == ref["iters"]
    # (i) every kernel form of the library gives the same fit: per-entry / look-up / pair-count statistics, the three CD
    #     kernels, single- and multi-pass solves agree with each other orders of magnitude more closely than with the oracle
    for form in fz.FORMS[1:]:
        got, err = fz.run_hip(case, form)
        assert got is not None, (form, err)
        cross = fz.errors(got, base)
        assert max(cross[0], cross[1]) < 1e-9 and cross[2] < 1e-10, (form, cross, dev)
        assert got["iters"] == base["iters"]
    # (ii) against the oracle in COVARIANCE form: base tolerances, unless a stopping decision differs (per-gene sweep counts of
    #      the last column step of fits of 0, 1, ... outer iterations)
    cov, cerr = fz.run_oracle(case, cd_form=1)
    assert cov is not None, cerr
    dev_cov = fz.errors(base, cov)
    flipped = None
    for it in range(case["iters"] + 1):
        _, _, sw_h = fz.run_hip(case, iters=it, want_sweeps=True)
        _, _, sw_o = fz.run_oracle(case, iters=it, want_sweeps=True, cd_form=1)
        if sw_h is None or sw_o is None or not np.array_equal(sw_h, sw_o):
            flipped = it
            break
    assert flipped is not None or (max(dev_cov[0], dev_cov[1]) < 1e-7 and dev_cov[2] < 1e-8), (dev_cov, dev)
    # (iii) what explains the deviation: the oracle's two forms differ from each other, with no GPU involved, by as much as the
    #       library from the parity oracle — or (cases of round 5: both oracle forms agree to 1e-14) a gene's stopping decision
    #       fell on the other side of the tolerance than the oracle's (found above), which moves the result by about the
    #       tolerance itself
    form_gap = fz.errors(dict(row_matrices={f"factor{i}": a for i, a in enumerate(cov["row_matrices"])},
                              column_factor=cov["column_factor"], traj=cov["traj"]), ref)
    assert flipped is not None or max(form_gap[0], form_gap[1]) > 0.2 * max(dev[0], dev[1]), (form_gap, dev, flipped)
    # (iv) the deviation stays within what the fixture records (x 3: not bit-stable across compilers of the oracle)
    tol = case["tolerated"]
    assert dev[0] <= 3 * tol["row"] and dev[1] <= 3 * tol["col"] and dev[2] <= 3 * tol["traj"], (dev, tol)
