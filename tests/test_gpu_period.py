"""The periodic sweep-order table on the DEVICE (include/insider_perm.h): the library and the oracle are both
rebuilt with a period of 64 sweeps, and solves of several hundred sweeps are compared — identical per-gene sweep counts and
iterates across the wrap, in all three CD kernels, through insider_hip_strong_cd, and in multi-pass solves whose pass limits
and resume points lie beyond the period.  (The production period, 16384, would need solves of > 16384 sweeps per case.)"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.gpu
def test_order_table_wrap_matches_oracle_with_a_short_period():
    sys.path.insert(0, ROOT)
    from insider_amd import _build
    out = os.path.join(HERE, "_build")
    os.makedirs(out, exist_ok=True)
    hip = os.path.join(out, "libinsider_hip_p64.so")
    orc = os.path.join(out, "libinsider_oracle_p64.so")
    _build.build_library(force=True, extra_flags=["-DINSIDER_PERM_PERIOD=64u"], out=hip)
    subprocess.check_call(["gcc", "-O2", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-std=c11", "-DINSIDER_PERM_PERIOD=64u", "-shared",
                           "-o", orc, os.path.join(ROOT, "oracle", "insider_oracle.c"), "-lm"])
    env = dict(os.environ, INSIDER_HIP_LIB=hip, INSIDER_ORACLE_LIB=orc)
    r = subprocess.run([sys.executable, os.path.join(HERE, "period_wrap_check.py")], capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
