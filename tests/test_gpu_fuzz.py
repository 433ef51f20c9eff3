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
