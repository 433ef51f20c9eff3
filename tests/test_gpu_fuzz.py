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
# suite's base tolerances against the oracle (factors 1e-7, trajectory 1e-8).  The claim these tests make executable: the
# deviation is not a kernel's — every kernel form of the library produces the same fit — but a stopping decision at rounding
# level (|loss change| <= tol, src/coordinate_descent.cpp:114) that falls differently in the oracle's residual-form loss
# difference and then amplifies through the following outer iterations.
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
    # (i) every kernel form of the library gives the same fit: per-entry / look-up / pair-count statistics, the three CD
    #     kernels, single- and multi-pass solves agree with each other orders of magnitude more closely than with the oracle
    for form in fz.FORMS[1:]:
        got, err = fz.run_hip(case, form)
        assert got is not None, (form, err)
        cross = fz.errors(got, base)
        assert max(cross[0], cross[1]) < 1e-9 and cross[2] < 1e-10, (form, cross, dev)
        assert got["iters"] == base["iters"]
    # (ii) HIP and oracle agree until their stopping decisions first differ: fits of 0, 1, ... outer iterations, per-gene
    #      sweep counts of the last column step on both sides
    first = None
    for it in range(case["iters"] + 1):
        g, _, sw_h = fz.run_hip(case, iters=it, want_sweeps=True)
        r, _, sw_o = fz.run_oracle(case, iters=it, want_sweeps=True)
        same = sw_h is not None and sw_o is not None and np.array_equal(sw_h, sw_o)
        if same:
            e = fz.errors(g, r)
            assert max(e[0], e[1]) < 1e-7 and e[2] < 1e-8, (it, e)          # no decision has differed yet: base tolerances hold
        elif first is None:
            first = it
            differ = np.flatnonzero(sw_h != sw_o)
            assert differ.size <= max(2, sw_o.size // 10), (it, differ.size)  # a few genes, not a systematic difference
            keep = np.setdiff1d(np.arange(sw_o.size), differ)
            ck, cr = g["column_factor"][:, keep], r["column_factor"][:, keep]
            assert np.linalg.norm(ck - cr) / max(np.linalg.norm(cr), 1e-300) < 1e-7  # the other genes' solves still agree
    assert first is not None, "no stopping decision differs: this case should meet the base tolerances"
    # (iii) the deviation at the end stays within what the fixture records (x 3: packing-independent, but not bit-stable
    #       across compilers of the oracle)
    tol = case["tolerated"]
    assert dev[0] <= 3 * tol["row"] and dev[1] <= 3 * tol["col"] and dev[2] <= 3 * tol["traj"], (dev, tol)
    assert base["iters"] == ref["iters"]
