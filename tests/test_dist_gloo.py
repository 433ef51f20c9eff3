"""Gene-axis sharding rehearsed on the CPU with gloo, world_size 2.

What runs here is the exchange protocol of insider_amd/dist.py and the algebra the HIP driver shards by: every term of
a level's ridge normal equations (src/optimize.cpp:161-175) is a sum over genes, so each rank reduces its gene slab to
the L_i x (K^2 + K) per-level equations and ONE sum-all-reduce per covariate makes them global; the loss needs one
all-reduce of {SSE_train, SSE_test, sum c^2, sum |c|, #train, #test}.  The slab partials are formed in numpy exactly
the way the kernels form them (complement statistics over held-out entries + per-level sums of X), and the reduced
result must equal the CPU oracle's update on the UNSHARDED problem.
"""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from insider_amd import dist as idist
from insider_amd import workloads


def test_shard_range_partitions():
    for p, world in ((50000, 8), (7, 3), (5, 8), (200000, 4)):
        spans = [idist.shard_range(p, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == p
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def _slab_level_equations(w, A, Cm, cov, lo, hi):
    """This rank's share of covariate `cov`'s per-level equations, formed like the kernels do."""
    X, M, lev = w.X[:, lo:hi], w.M_train[:, lo:hi], w.levels
    Cs = Cm[:, lo:hi]
    K, L = Cm.shape[0], int(w.n_levels[cov])
    CCt = Cs @ Cs.T                                   # this slab's part of C C'
    eq = np.zeros((L, K * K + K))
    s_all = sum(A[m][lev[:, m] - 1, :] for m in range(lev.shape[1]) if m != cov)     # s_r = sum_{m != cov} A_m[level]
    for l in range(L):
        members = np.flatnonzero(lev[:, cov] == l + 1)
        XtX = len(members) * CCt
        Xty = Cs @ X[members, :].sum(axis=0)          # (S_i C')[l]: per-level sum of X rows, then times C'
        for r in members:
            held = M[r, :] == 0
            Hc = Cs[:, held] @ Cs[:, held].T          # complement Gram
            bc = Cs[:, held] @ X[r, held]             # complement XtY
            XtX -= Hc
            Xty += -bc - CCt @ s_all[r] + Hc @ s_all[r]
        eq[l, :K * K] = XtX.ravel()
        eq[l, K * K:] = Xty
    return eq


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = workloads.small(n=60, p=90, level_counts=(5, 4), K=4, f=0.15, seed=11, with_na=True)
        rng = np.random.default_rng(3)
        A = [rng.standard_normal(a.shape) * 0.3 for a in w.A0]
        Cm = rng.standard_normal(w.C0.shape) * 0.3
        lo, hi = idist.shard_range(w.p, rank, world)
        ar = idist.HostAllreduce()
        K, lam = w.K, 1.7
        new_A = [a.copy() for a in A]
        for cov in range(w.levels.shape[1]):          # Gauss-Seidel over covariates: one all-reduce each
            eq = np.ascontiguousarray(_slab_level_equations(w, new_A, Cm, cov, lo, hi))
            ar(eq.ctypes.data, eq.size, 0)            # the callback signature the C ABI uses: (pointer, count, stream)
            for l in range(eq.shape[0]):
                XtX = eq[l, :K * K].reshape(K, K) + lam * np.eye(K)
                new_A[cov][l] = np.linalg.solve(XtX, eq[l, K * K:])
        # loss terms: slab sums + one all-reduce of 6 doubles
        R = sum(new_A[m][w.levels[:, m] - 1, :] for m in range(w.levels.shape[1]))
        resid = w.X[:, lo:hi] - R @ Cm[:, lo:hi]
        tr, te = w.M_train[:, lo:hi] != 0, w.M_test[:, lo:hi] != 0
        buf = np.array([np.sum(resid[tr] ** 2), np.sum(resid[te] ** 2), np.sum(Cm[:, lo:hi] ** 2),
                        np.sum(np.abs(Cm[:, lo:hi])), tr.sum(), te.sum()], dtype=np.float64)
        ar(buf.ctypes.data, buf.size, 0)
        q.put((rank, [a.copy() for a in new_A], buf.copy(), list(ar.calls)))
    finally:
        dist.destroy_process_group()


def test_two_rank_row_update_and_loss_match_unsharded(oracle):
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # both ranks end with bit-identical row factors (they solve the same reduced systems)
    for a0, a1 in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(a0, a1)
    assert np.array_equal(outs[0][2], outs[1][2])
    # ... equal to the oracle's row updates on the unsharded problem (reference formulation: Gauss-Seidel residual)
    w = workloads.small(n=60, p=90, level_counts=(5, 4), K=4, f=0.15, seed=11, with_na=True)
    rng = np.random.default_rng(3)
    A = [rng.standard_normal(a.shape) * 0.3 for a in w.A0]
    Cm = rng.standard_normal(w.C0.shape) * 0.3
    gram = Cm @ Cm.T
    ref_A = [a.copy() for a in A]
    resid = w.X - sum(ref_A[m][w.levels[:, m] - 1, :] for m in range(2)) @ Cm
    for cov in range(2):
        resid = resid + ref_A[cov][w.levels[:, cov] - 1, :] @ Cm                       # src/optimize.cpp:338
        ref_A[cov] = oracle.optimize_row(resid, w.M_train, ref_A[cov], Cm, w.levels[:, cov], gram, 1.7, 1)
        resid = resid - ref_A[cov][w.levels[:, cov] - 1, :] @ Cm                       # :354
    for got, ref in zip(outs[0][1], ref_A):
        np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12)
    R = sum(ref_A[m][w.levels[:, m] - 1, :] for m in range(2))
    full = w.X - R @ Cm
    exp = [np.sum(full[w.M_train != 0] ** 2), np.sum(full[w.M_test != 0] ** 2), np.sum(Cm ** 2), np.sum(np.abs(Cm)),
           np.count_nonzero(w.M_train), np.count_nonzero(w.M_test)]
    np.testing.assert_allclose(outs[0][2], exp, rtol=1e-10)
    # exchange volume: one all-reduce per covariate of L_i (K^2 + K) doubles, one of 6 for the loss
    assert outs[0][3] == [5 * (16 + 4), 4 * (16 + 4), 6]


# ---- grid-parallel tune(): the host logic, with the device fit stubbed out ----------------------
class _FakeData:
    """Stands in for the HBM-resident data set: a deterministic function of the inits and hyper-parameters."""

    def optimize(self, cfd, col, K, l1, l2, a, tuning, gtol, stol, iters, seed=0, inc_continuous=0):
        s = float(sum(np.sum(m) for m in cfd) + np.sum(col))
        return dict(train_rmse=1e3 * s + l1, test_rmse=1e3 * s + 10 * a + K)


def _tune_worker(rank, world, port, q):
    import torch.distributed as dist
    from insider_amd import api
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        obj = api.Insider(params=dict(global_tol=1e-9, sub_tol=1e-5, tuning_iter=3, max_iter=5), inc_continuous=0,
                          confounder=workloads.cyclic_levels(12, (3, 2)), data=np.zeros((12, 9)), seed=1)
        obj["_resident_tune"] = _FakeData()
        out = api.tune(obj, latent_dimension=np.array([2, 3, 4]), lambda_=[1.0, 2.0, 3.0], alpha=[0.1, 0.2],
                       rng=np.random.default_rng(5), rank=rank, world=world)
        q.put((rank, out["rank_tuning"], out["latent_rank"], out["reg_tuning"]))
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_grid_parallel_tune_matches_serial():
    ctx = mp.get_context("spawn")
    results = {}
    for world in (1, 2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        q = ctx.Queue()
        procs = [ctx.Process(target=_tune_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        results[world] = outs
    serial = results[1][0]
    for r in results[2]:   # every rank of the 2-rank job ends with the serial tables
        np.testing.assert_allclose(r[1], serial[1], rtol=1e-13)
        assert r[2] == serial[2]
        np.testing.assert_allclose(r[3], serial[3], rtol=1e-13)
    assert serial[3].shape == (6, 4) and serial[1].shape == (3, 3)


# ---- tune(concurrent=k): the host logic around the clones, with the device fit stubbed out ---------------------------------
class _FakeHandle(_FakeData):
    def __init__(self, src=None):
        self._h = 1
        self._options = dict(src._options) if src is not None else {}
        self.fits = 0

    def clone(self):
        return _FakeHandle(self)

    def set_option(self, name, value):
        self._options[name] = float(value)

    def optimize(self, *a, **kw):
        self.fits += 1
        return super().optimize(*a, **kw)

    def profile(self):
        return dict(wall_ms=0.0)

    def close(self):
        self._h = None


def _concurrent_obj():
    from insider_amd import api
    obj = api.Insider(params=dict(global_tol=1e-9, sub_tol=1e-5, tuning_iter=3, max_iter=5), inc_continuous=0,
                      confounder=workloads.cyclic_levels(12, (3, 2)), data=np.zeros((12, 9)), seed=1)
    obj["_resident_tune"] = _FakeHandle()
    return obj


def test_concurrent_tune_host_logic_survives_a_failing_producer_and_closed_clones(monkeypatch):
    """(i) an exception while drawing the inits must end tune(concurrent=k) with that exception, not leave the
    workers blocked on the queue for ever; (ii) clones closed since the last call (bench.py closes them) are not re-used;
    (iii) options set on the data set after the clones were made reach them."""
    import threading
    from insider_amd import api
    grid = dict(latent_dimension=np.array([3]), lambda_=[1.0, 2.0, 3.0], alpha=[0.1, 0.2])
    obj = _concurrent_obj()
    serial = api.tune(obj, rng=np.random.default_rng(5), **grid)["reg_tuning"]
    conc = api.tune(obj, rng=np.random.default_rng(5), concurrent=3, **grid)["reg_tuning"]
    np.testing.assert_array_equal(conc, serial)
    clones = list(obj["_tune_clones"])
    assert len(clones) == 2 and obj["_resident_tune"].fits + sum(c.fits for c in clones) == 12   # 6 serial + 6 dealt over the handles
    # (ii) + (iii)
    clones[0].close()
    obj["_resident_tune"].set_option("cd_pass1", 128)
    again = api.tune(obj, rng=np.random.default_rng(5), concurrent=3, **grid)["reg_tuning"]
    np.testing.assert_array_equal(again, serial)
    assert clones[0] not in obj["_tune_clones"] and len(obj["_tune_clones"]) == 2
    assert all(c._h and c._options.get("cd_pass1") == 128.0 for c in obj["_tune_clones"])
    # (i): the third draw raises
    calls = {"n": 0}
    real = api._fresh_inits

    def failing(*a, **kw):
        calls["n"] += 1
        if calls["n"] == 3:
            raise MemoryError("no room for the inits")
        return real(*a, **kw)

    monkeypatch.setattr(api, "_fresh_inits", failing)
    box = {}

    def run():
        try:
            api.tune(obj, rng=np.random.default_rng(5), concurrent=2, **grid)
            box["out"] = "returned"
        except BaseException as e:
            box["out"] = e

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(timeout=60)
    assert not t.is_alive(), "tune(concurrent=2) hangs when the producer of the inits fails"
    assert isinstance(box["out"], MemoryError)


# ---- the exchange vote of a multi-rank job (insider_amd/dist.py:attach_voted), world_size 2 over gloo --------------------------
class _VoteHandle:
    """Records what attach_voted does to a handle; comm_init fails on the ranks listed in `bad_join`."""

    def __init__(self, rank, bad_join=()):
        self.rank, self.bad_join, self.log = rank, bad_join, []

    def set_shard(self, gene_offset, rank, world, allreduce=None):
        self.log.append(("set_shard", type(allreduce).__name__))

    def comm_init(self, uid, rank, world):
        self.log.append(("comm_init", len(uid)))
        if rank in self.bad_join:
            raise RuntimeError("join failed")


def _vote_worker(rank, world, port, case, q):
    import torch.distributed as dist
    from insider_amd import _lib, dist as idist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("INSIDER_FAIL_COMM_RANK", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        idist._make_unique_id = lambda: b"\x07" * _lib.COMM_ID_BYTES if case != "no_id" else (_ for _ in ()).throw(RuntimeError("no id"))
        if case == "prepare_fails_on_1":
            os.environ["INSIDER_FAIL_COMM_RANK"] = "1"
        ds = _VoteHandle(rank, bad_join=(1,) if case == "join_fails_on_1" else ())
        ex, rep = idist.attach_voted(ds, 100 * rank, rank, world, fallback="staged")
        q.put((rank, ex if isinstance(ex, str) else type(ex).__name__, rep, ds.log))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["all_join", "prepare_fails_on_1", "join_fails_on_1", "no_id"])
def test_exchange_vote_keeps_all_ranks_on_one_path(case):
    """the first real multi-GPU run must not die (or hang) in glue.  Whatever one rank suffers, all ranks
    end on the same exchange; a rank that could not prepare keeps every rank out of the blocking join."""
    ctx = mp.get_context("spawn")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    q = ctx.Queue()
    procs = [ctx.Process(target=_vote_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    paths = {o[1] for o in outs}
    assert len(paths) == 1, outs                                   # every rank on the same exchange
    joins = [sum(1 for e in o[3] if e[0] == "comm_init") for o in outs]
    if case == "all_join":
        assert paths == {"rccl"} and joins == [1, 1] and all(o[2]["ready_min"] == 1 and o[2]["joined_min"] == 1 for o in outs)
    elif case == "join_fails_on_1":
        assert paths == {"StagedHostAllreduce"} and joins == [1, 1] and all(o[2]["joined_min"] == 0 for o in outs)
        assert all(o[3][-1] == ("set_shard", "StagedHostAllreduce") for o in outs)     # the half-made communicator is replaced
    else:
        assert paths == {"StagedHostAllreduce"} and joins == [0, 0]                    # nobody entered the blocking join
        assert all(o[2]["ready_min"] == 0 and o[2]["joined_min"] is None and o[2]["path"] == "staged" for o in outs)
