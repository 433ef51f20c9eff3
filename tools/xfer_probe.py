"""Diagnostics (GPU box): what the factor transfers of one optimize() call cost: 12 MB (K x p doubles at c3) host -> device and
back, from pageable memory (what the C ABI's caller-owned buffers are) and from pinned memory.   python tools/xfer_probe.py"""
import time, numpy as np, torch
n = 30 * 50000
for pinned in (False, True):
    h = torch.empty(n, dtype=torch.float64, pin_memory=pinned)
    h.normal_()
    d = torch.empty(n, dtype=torch.float64, device="cuda")
    for name, fn in (("h2d", lambda: d.copy_(h)), ("d2h", lambda: h.copy_(d))):
        ts = []
        for _ in range(20):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"{'pinned' if pinned else 'pageable'} {name}: median {1e3 * ts[10]:.3f} ms, best {1e3 * ts[0]:.3f} ms = {n * 8 / ts[0] / 1e9:.1f} GB/s")
