"""Replay-based projection of the gene-sharded multi-GPU run on ONE GPU.

What a rank of `bench.py --gpus N` does is fixed by (i) its gene slab and (ii) the GLOBAL level equations it receives from
the all-reduce of every covariate's row update (and the global loss terms at checkpoints): the row factors of all ranks are
bit-identical, so every rank's column step solves its genes against the same R.  This tool therefore

  1. RECORDS: runs the whole problem once on the one GPU as a world-1 job with `force_allreduce` — the path a sharded rank
     takes (level equations formed, handed to the exchange, then solved) — and copies every buffer the exchange sees (the
     global sums, by construction) to the host, for the warm-up call and the timed call of bench.py's command;
  2. REPLAYS: for N in {2, 4, 8} and EVERY rank r < N: a handle on rank r's slab with world = N whose exchange callback
     overwrites the slab's own partial sums with the recorded global ones (a device-to-device copy enqueued on the library's
     stream: no host synchronisation, like ncclAllReduce) and brackets it with HIP events.  The rank thus computes exactly
     what it would in the N-GPU job — its own row-phase partial sums, the replicated solves, its genes' sweeps at the TRUE
     sweep counts — with an exchange that costs nothing;
  3. PROJECTS: the job is bulk-synchronous at every exchange, so  T_N = sum over the segments between exchanges of the
     slowest rank's segment + the priced all-reduces + the slowest rank's time outside the events (factor transfers).
     The all-reduce price is an ASSUMPTION (no multi-GPU box in this pipeline), stated in the output with a sensitivity row.

Usage: python tools/scale_replay.py [--workload c4] [--steps 20] [--warmup 5] [--ranks 8,4,2] [--out FILE]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import __graft_entry__ as ge  # noqa: E402

ge.build()
from insider_amd import _lib, api, dist as idist, workloads  # noqa: E402

hip = C.CDLL("libamdhip64.so")
vp = C.c_void_p
hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
hip.hipFree.argtypes = [vp]
hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
hip.hipStreamSynchronize.argtypes = [vp]
hip.hipEventCreate.argtypes = [C.POINTER(vp)]
hip.hipEventRecord.argtypes = [vp, vp]
hip.hipEventSynchronize.argtypes = [vp]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), vp, vp]
hip.hipEventDestroy.argtypes = [vp]
H2D, D2H, D2D = 1, 2, 3


def ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")


class Recorder:
    """Exchange callback of the whole-problem run: keeps a host copy of every buffer the exchange is handed."""

    def __init__(self):
        self.bufs = []

    def __call__(self, ptr, count, stream):
        ck(hip.hipStreamSynchronize(stream), "sync")
        host = np.empty(count, dtype=np.float64)
        ck(hip.hipMemcpy(host.ctypes.data, ptr, count * 8, D2H), "D2H")
        self.bufs.append(host)


class Replayer:
    """Exchange callback of a slab run: the recorded global buffer replaces the slab's partial sums, stream-ordered."""

    def __init__(self, record):
        self.counts = [len(b) for b in record]
        flat = np.concatenate(record)
        self.dev = vp()
        ck(hip.hipMalloc(C.byref(self.dev), flat.nbytes), "hipMalloc")
        ck(hip.hipMemcpy(self.dev, flat.ctypes.data, flat.nbytes, H2D), "H2D")
        self.off = np.concatenate([[0], np.cumsum(self.counts)]).astype(np.int64) * 8
        self.reset()

    def reset(self):
        self.k = 0
        self.events = []       # (before, after) per exchange

    def __call__(self, ptr, count, stream):
        if self.k >= len(self.counts) or count != self.counts[self.k]:
            raise RuntimeError(f"replay out of step: exchange {self.k} has {count} doubles, recorded "
                               f"{self.counts[self.k] if self.k < len(self.counts) else None}")
        e0, e1 = vp(), vp()
        ck(hip.hipEventCreate(C.byref(e0)), "event")
        ck(hip.hipEventCreate(C.byref(e1)), "event")
        ck(hip.hipEventRecord(e0, stream), "record")
        ck(hip.hipMemcpyAsync(ptr, self.dev.value + int(self.off[self.k]), count * 8, D2D, stream), "D2D")
        ck(hip.hipEventRecord(e1, stream), "record")
        self.events.append((e0, e1))
        self.k += 1

    def segments_ms(self):
        """Device time between consecutive exchanges: [after exchange k-1 -> before exchange k], k = 1 .. n-1."""
        out = []
        for k in range(1, len(self.events)):
            t = C.c_float()
            ck(hip.hipEventSynchronize(self.events[k][0]), "event sync")
            ck(hip.hipEventElapsedTime(C.byref(t), self.events[k - 1][1], self.events[k][0]), "elapsed")
            out.append(float(t.value))
        for e0, e1 in self.events:
            hip.hipEventDestroy(e0)
            hip.hipEventDestroy(e1)
        return out

    def close(self):
        hip.hipFree(self.dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ranks", default="8,4,2")
    ap.add_argument("--seed", type=int, default=20240301)
    ap.add_argument("--out", default="gpurun_out/r05/scale_projection.json")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE")
    args = ap.parse_args()
    name = args.workload
    n, p, _, _, K, lam, alpha, tuning, f = workloads.CONFIGS[name]
    t0 = time.perf_counter()
    w = workloads.make(name)
    print(f"generated {name} ({n} x {p}) in {time.perf_counter() - t0:.1f} s", flush=True)
    lam_w = lam - 2.0 if lam > 2.0 else lam + 2.0     # bench.py's warm-up call: a neighbouring grid point

    def inits(seed, lo, hi):
        A0, C0 = workloads.init_factors(w.n_levels, K, p, seed)
        return A0, np.asfortranarray(C0[:, lo:hi])

    def calls(ds, lo, hi):
        """bench.py's two calls; returns the wall time of the timed one."""
        if args.warmup > 0:
            A, Cm = inits(workloads.INIT_SEED + 1, lo, hi)
            ds.optimize(A, Cm, K, lam_w, lam_w, alpha, tuning=tuning, max_iter=args.warmup - 1, global_tol=-1.0, seed=args.seed + 1,
                        copy=False)
        A, Cm = inits(workloads.INIT_SEED, lo, hi)
        t1 = time.perf_counter()
        res = ds.optimize(A, Cm, K, lam, lam, alpha, tuning=tuning, max_iter=args.steps - 1, global_tol=-1.0, seed=args.seed, copy=False)
        return time.perf_counter() - t1, res

    def options(ds):
        ds.set_option("profile", 1)
        for kv in args.opt:
            k, v = kv.split("=")
            ds.set_option(k, float(v))

    # ---- 1. the whole problem on this GPU: plain (the one-GPU figure), then recorded ---------------------------------------
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    options(ds)
    t_one, res_one = calls(ds, 0, p)
    prof_one = ds.profile()
    print(f"whole problem, plain: {t_one * 1e3:.1f} ms for {args.steps} iterations = {args.steps / t_one:.2f} it/s", flush=True)
    rec = Recorder()
    ds.set_shard(0, 0, 1, rec)
    ds.set_option("force_allreduce", 1)
    n_warm = 0
    if args.warmup > 0:
        A, Cm = inits(workloads.INIT_SEED + 1, 0, p)
        ds.optimize(A, Cm, K, lam_w, lam_w, alpha, tuning=tuning, max_iter=args.warmup - 1, global_tol=-1.0, seed=args.seed + 1, copy=False)
        n_warm = len(rec.bufs)
    A, Cm = inits(workloads.INIT_SEED, 0, p)
    res_rec = ds.optimize(A, Cm, K, lam, lam, alpha, tuning=tuning, max_iter=args.steps - 1, global_tol=-1.0, seed=args.seed, copy=False)
    ds.close()
    record = rec.bufs
    sizes = sorted(set(len(b) for b in record))
    print(f"recorded {len(record)} exchanges ({n_warm} in the warm-up call), payloads (doubles): {sizes}; "
          f"loss {res_rec['loss']:.12g} (plain run {res_one['loss']:.12g})", flush=True)
    rp = Replayer(record)
    timed = list(range(n_warm, len(record)))           # exchanges of the timed call
    payload_bytes = [8 * len(record[k]) for k in timed]

    # ---- 2. every rank of every N, replayed ----------------------------------------------------------------------------------
    out = {"workload": f"{name}: {n} x {p}, K = {K}", "command": f"bench.py --gpus N --steps {args.steps} --warmup {args.warmup}",
           "library_source_sha": _lib.library_source_sha(), "options": args.opt,
           "single_gpu": {"wall_ms": t_one * 1e3, "value_it_per_s": args.steps / t_one, "cd_ms": prof_one["cd_ms"],
                          "col_stats_ms": prof_one["col_stats_ms"], "sweeps": prof_one["sweeps"]},
           "exchanges_per_timed_call": len(timed), "exchange_payload_bytes": sorted(set(payload_bytes)),
           "method": ("every rank's slab run on one GPU with the recorded GLOBAL level equations / loss terms replacing its partial sums "
                      "(stream-ordered device copy): the rank's true work at the true sweep counts, exchange cost zero; bulk-synchronous "
                      "combination: sum over segments of the slowest rank + priced all-reduces + slowest rank's time outside the events"),
           "N": {}}
    for N in [int(v) for v in args.ranks.split(",")]:
        ranks = []
        for r in range(N):
            lo, hi = idist.shard_range(p, r, N)
            t1 = time.perf_counter()
            ds = api.InsiderData(np.asfortranarray(w.X[:, lo:hi]), w.levels, np.asfortranarray(w.M_train[:, lo:hi]),
                                 np.asfortranarray(w.M_test[:, lo:hi]))
            t_up = time.perf_counter() - t1
            options(ds)
            rp.reset()
            ds.set_shard(lo, r, N, rp)
            wall, res = calls(ds, lo, hi)
            prof = ds.profile()
            seg = rp.segments_ms()
            # seg[k] ends at exchange k + 1; the timed call's exchanges are n_warm ... : its inner segments are seg[n_warm:].  (The one
            # that ends at its FIRST exchange starts in the warm-up call: dropped; that part of the timed call — factor upload, the
            # evaluation of the initial values — and the part behind its last exchange are inside `outside_ms` below.)
            seg_t = seg[n_warm:]
            dev_span = sum(seg_t)
            ranks.append({"rank": r, "genes": hi - lo, "wall_ms": wall * 1e3, "segments_ms": seg_t, "device_span_ms": dev_span,
                          "outside_ms": wall * 1e3 - dev_span, "cd_ms": prof["cd_ms"], "col_stats_ms": prof["col_stats_ms"],
                          "sweeps": prof["sweeps"], "max_gene_sweeps": int(ds.info("max_gene_sweeps")), "loss": res["loss"],
                          "upload_s": t_up})
            ds.close()
            print(f"N = {N} rank {r}: {hi - lo} genes, wall {wall * 1e3:.1f} ms, between exchanges {dev_span:.1f} ms, cd {prof['cd_ms']:.1f} ms, "
                  f"statistics {prof['col_stats_ms']:.1f} ms, sweeps {prof['sweeps']}", flush=True)
        nseg = min(len(rk["segments_ms"]) for rk in ranks)
        bsp = sum(max(rk["segments_ms"][k] for rk in ranks) for k in range(nseg))
        outside = max(rk["outside_ms"] for rk in ranks)
        slowest_wall = max(rk["wall_ms"] for rk in ranks)
        proj = {}
        for label, a_us, bw in (("assumed", 25.0, 50e9), ("optimistic", 10.0, 150e9), ("pessimistic", 50.0, 25e9)):
            # ring / tree all-reduce of B bytes over N ranks on xGMI: alpha + 2 (N - 1) / N * B / bw   (alpha: launch + N-rank latency)
            t_ar = sum(a_us * 1e-3 + 2.0 * (N - 1) / N * b / bw * 1e3 for b in payload_bytes)
            T = bsp + outside + t_ar
            proj[label] = {"alpha_us": a_us, "link_GBs": bw / 1e9, "allreduce_ms_total": t_ar, "T_ms": T,
                           "value_it_per_s": args.steps / (T * 1e-3), "speedup_vs_single_gpu": (t_one * 1e3) / T}
        out["N"][str(N)] = {"ranks": ranks, "bsp_segments_ms": bsp, "outside_events_ms_max": outside, "slowest_rank_wall_ms": slowest_wall,
                            "zero_cost_exchange": {"T_ms": bsp + outside, "speedup_vs_single_gpu": (t_one * 1e3) / (bsp + outside)},
                            "projection": proj,
                            "loss_matches_whole_run": bool(all(abs(rk["loss"] - res_rec["loss"]) <= 1e-9 * abs(res_rec["loss"]) for rk in ranks))}
        print(f"N = {N}: slowest rank alone {slowest_wall:.1f} ms; bulk-synchronous {bsp + outside:.1f} ms (zero-cost exchange) = "
              f"{(t_one * 1e3) / (bsp + outside):.2f} x; with the assumed all-reduce price {proj['assumed']['T_ms']:.1f} ms = "
              f"{proj['assumed']['speedup_vs_single_gpu']:.2f} x ({proj['assumed']['value_it_per_s']:.1f} it/s)", flush=True)
    rp.close()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
