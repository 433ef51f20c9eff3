"""Freezes the sweep-order spec (include/insider_perm.h) as golden bytes: tests/golden/perm_golden.json.

The oracle and the HIP kernels BOTH take the coordinate order of a sweep from include/insider_perm.h (the reference's
randperm, src/coordinate_descent.cpp:89, is irreproducible), so the two cannot disagree about it — and a silent edit of the
header would move both together.  These fixed (seed, outer iteration, sweep, K) -> order vectors make such an edit fail
tests/test_oracle_cd.py::test_sweep_order_golden_bytes (oracle + numpy restatement) and
tests/test_gpu_parity.py::test_device_order_table_matches_golden (the table the device kernels read).

    python tests/golden/make_perm_golden.py        (only when the spec is changed ON PURPOSE)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle  # noqa: E402

POINTS = [(seed, it, sweep, K) for seed in (0, 17, 20240301, (1 << 40) + 12345) for it in (0, 3, 30) for sweep in (0, 1, 255, 16383, 16384, 40000)
          for K in (1, 7, 30, 63)]

if __name__ == "__main__":
    out = [dict(seed=s, iter=i, sweep=w, K=K, order=c_oracle.sweep_order(K, s, i, w)) for (s, i, w, K) in POINTS]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "perm_golden.json")
    json.dump(out, open(path, "w"), separators=(",", ":"))
    print("wrote", len(out), "orders to", path)
