"""Soak (GPU): repeated optimize() calls on one handle must be bitwise reproducible (the side streams only carry
scheduling hints and operands the main chain waits for) and leak no device memory.   python tests/soak_determinism.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import ctypes
from insider_amd import api, workloads
_hip = ctypes.CDLL("libamdhip64.so")


def free_device_bytes():
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert _hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value


w = workloads.make("c2")
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
ref = {}
free0 = None
bad = 0
for rep in range(24):
    lam, alpha = ((5.0, 0.4), (1.0, 0.2), (9.0, 0.5))[rep % 3]
    A = [a.copy(order="F") for a in w.A0]; C = w.C0.copy(order="F")
    r = ds.optimize(A, C, w.K, lam, lam, alpha, max_iter=12, global_tol=-1, seed=5)
    key = (lam, alpha)
    sig = (r["column_factor"].tobytes(), tuple(r["row_matrices"][k].tobytes() for k in sorted(r["row_matrices"])), r["traj"].tobytes())
    if key in ref and ref[key] != sig:
        bad += 1
        print(f"rep {rep} {key}: result differs from the first run with these penalties", flush=True)
    ref.setdefault(key, sig)
    free = free_device_bytes()
    if rep == 3:
        free0 = free
    if rep > 3 and free < free0 - (64 << 20):
        bad += 1
        print(f"rep {rep}: free device memory fell from {free0} to {free}", flush=True)
ds.close()
print("soak:", "ok" if not bad else f"{bad} problems")
sys.exit(1 if bad else 0)
