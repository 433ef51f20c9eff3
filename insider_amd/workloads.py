"""Synthetic INSIDER workloads (BASELINE.json configs c1..c5).

Generator = the reference's tests/simulation.rmd:19-63 scaled up: cyclic level
ids so that every level tuple occurs (v1_dis / v2_dis, :40-46), A_i* ~ N(0,1),
C* ~ N(0,1) with 30 % of the gene columns zeroed (:25-26), X = (sum Z_i A_i*) C*
+ N(0,1) (:59-63).  Hold-out mask = uniform element-wise sample without
replacement of floor(f*n*p) entries (R/utils.R:88-100, there with set.seed(123)).
Inits i.i.d. N(0, 0.001^2) (R/utils.R:40-43).  Interaction indicator = index of
the unique tuple of the selected columns, inserted as column 2
(R/insider.R:34-40).  Seeds: data 20240301, mask 123, init 7.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

DATA_SEED = 20240301
MASK_SEED = 123
INIT_SEED = 7


@dataclass
class Workload:
    name: str
    X: np.ndarray            # n x p float64, Fortran order
    levels: np.ndarray       # n x c int32, Fortran order, 1-based
    n_levels: np.ndarray     # c int32
    M_train: np.ndarray      # n x p uint8, Fortran order
    M_test: np.ndarray       # n x p uint8, Fortran order
    K: int
    lam: float
    alpha: float
    tuning: int
    A0: List[np.ndarray] = field(default_factory=list)   # L_i x K, Fortran order
    C0: Optional[np.ndarray] = None                      # K x p, Fortran order

    @property
    def n(self):
        return self.X.shape[0]

    @property
    def p(self):
        return self.X.shape[1]


CONFIGS = {
    # name: (n, p, base level counts, interaction_idx (1-based) or None, K, lambda, alpha, tuning, held-out f)
    "c1": (377, 5000, (2, 8, 107), (1, 2), 23, 10.0, 0.4, 0, 0.1),
    "c2": (2000, 20000, (50, 5), None, 20, 5.0, 0.4, 1, 0.1),
    "c3": (10000, 50000, (100, 10), None, 30, 5.0, 0.4, 1, 0.1),
    "c4": (10000, 200000, (100, 10), None, 30, 5.0, 0.4, 1, 0.1),
    "c5": (5000, 50000, (20, 10, 25), (1, 2), 25, 5.0, 0.4, 1, 0.1),
}


def cyclic_levels(n, level_counts):
    """Mixed-radix cyclic level ids (tests/simulation.rmd:40-46 generalised): last covariate cycles fastest."""
    lev = np.zeros((n, len(level_counts)), dtype=np.int32, order="F")
    r = np.arange(n, dtype=np.int64)
    stride = 1
    for i in range(len(level_counts) - 1, -1, -1):
        # every level (and as many tuples as possible) must occur even when n < prod(L_i)
        st = max(1, min(stride, n // int(np.prod(level_counts[: i + 1]))))
        lev[:, i] = (r // st) % level_counts[i] + 1
        stride *= level_counts[i]
    return lev


def interaction_indicator(confounder, interaction_idx):
    """R/insider.R:34-40: level = index (1-based, order of first appearance as R's unique()) of the tuple of the
    selected columns; inserted as the second column."""
    sel = confounder[:, [i - 1 for i in interaction_idx]]
    seen = {}
    inter = np.zeros(confounder.shape[0], dtype=np.int32)
    for r in range(sel.shape[0]):
        key = tuple(int(v) for v in sel[r])
        if key not in seen:
            seen[key] = len(seen) + 1
        inter[r] = seen[key]
    out = np.column_stack([confounder[:, 0], inter, confounder[:, 1:]]).astype(np.int32)
    return np.asfortranarray(out)


GENE_BLOCK = 1024   # X noise and hold-out masks are generated per block of genes from block-keyed streams


def _block_counts(n, p, f, seed):
    """How many of the floor(f*n*p) held-out entries fall in each gene block: sequential multivariate
    hypergeometric draws, so that (counts, then uniform choice inside each block) is exactly a uniform sample
    without replacement of the whole matrix (R/utils.R:88-100) while every block can be generated on its own."""
    rng = np.random.Generator(np.random.PCG64([seed, 0]))
    nblk = (p + GENE_BLOCK - 1) // GENE_BLOCK
    sizes = np.array([n * (min((b + 1) * GENE_BLOCK, p) - b * GENE_BLOCK) for b in range(nblk)], dtype=np.int64)
    k = int(np.floor(n * p * f))
    counts = np.zeros(nblk, dtype=np.int64)
    remaining_total, remaining_k = int(sizes.sum()), k
    for b in range(nblk):
        if remaining_k == 0:
            break
        good, bad = int(sizes[b]), remaining_total - int(sizes[b])
        if bad <= 0:
            counts[b] = remaining_k
        elif good < 10 ** 9 and bad < 10 ** 9:
            counts[b] = rng.hypergeometric(good, bad, remaining_k)
        else:
            # numpy's hypergeometric refuses populations of 1e9 and more (config c4: n p = 2e9): at that size its
            # normal limit is exact to far below one count; the sequential scheme still makes the total exactly k
            N = good + bad
            mean = remaining_k * good / N
            var = remaining_k * (good / N) * (bad / N) * (N - remaining_k) / (N - 1)
            counts[b] = int(min(max(round(rng.normal(mean, np.sqrt(var))), max(0, remaining_k - bad)), min(good, remaining_k)))
        remaining_total -= good
        remaining_k -= int(counts[b])
    return counts


def _block_test_mask(n, width, k, seed, b):
    """Exactly k test entries, uniform without replacement, inside an n x width block (Fortran order)."""
    rng = np.random.Generator(np.random.PCG64([seed, 2, b]))
    tot = n * width
    test = np.zeros(tot, dtype=np.uint8)
    if k > 0:
        if tot <= 2_000_000:
            test[rng.choice(tot, size=k, replace=False)] = 1
        else:  # random keys + k-th smallest: same distribution
            keys = rng.random(tot, dtype=np.float32)
            thr = np.partition(keys, k - 1)[k - 1]
            sel = np.flatnonzero(keys <= thr)
            if sel.size > k:  # ties at the threshold: drop surplus deterministically
                tie = np.flatnonzero(keys[sel] == thr)
                sel = np.delete(sel, tie[: sel.size - k])
            test[sel] = 1
    return test.reshape((n, width), order="F")


def holdout_masks(n, p, f, seed=MASK_SEED, gene_range=None):
    """Exactly floor(f*n*p) held-out entries over the WHOLE n x p matrix, uniform without replacement
    (R/utils.R:88-100); returns the (train, test) masks of the requested gene slab. No NA entries."""
    lo, hi = gene_range if gene_range is not None else (0, p)
    counts = _block_counts(n, p, f, seed)
    test = np.zeros((n, hi - lo), dtype=np.uint8, order="F")
    blocks = range(lo // GENE_BLOCK, (hi + GENE_BLOCK - 1) // GENE_BLOCK)

    def one(b):
        b0, b1 = b * GENE_BLOCK, min((b + 1) * GENE_BLOCK, p)
        m = _block_test_mask(n, b1 - b0, int(counts[b]), seed, b)
        s0, s1 = max(b0, lo), min(b1, hi)
        test[:, s0 - lo:s1 - lo] = m[:, s0 - b0:s1 - b0]

    _pmap(one, blocks)
    train = (1 - test).astype(np.uint8, order="F")
    return train, test


def _pmap(fn, items, workers=None):
    items = list(items)
    if len(items) <= 1:
        for it in items:
            fn(it)
        return
    import os
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers or min(16, os.cpu_count() or 1)) as ex:
        list(ex.map(fn, items))


def init_factors(n_levels, K, p, seed=INIT_SEED):
    """init_parameters (R/utils.R:40-43): N(0, 0.001^2) for every A_i and for C."""
    rng = np.random.Generator(np.random.PCG64(seed))
    A0 = [np.asfortranarray(rng.normal(0.0, 0.001, size=(int(L), K))) for L in n_levels]
    C0 = np.asfortranarray(rng.normal(0.0, 0.001, size=(K, p)))
    return A0, C0


def make(name=None, n=None, p=None, level_counts=None, interaction_idx=None, K=None, lam=5.0, alpha=0.4, tuning=1,
         f=0.1, data_seed=DATA_SEED, mask_seed=MASK_SEED, init_seed=INIT_SEED, gene_range=None, k_true=None):
    """Build a workload. ``name`` picks a BASELINE config; explicit arguments override / define a custom one.

    ``gene_range=(lo, hi)`` keeps only that gene slab of X / masks / C0 (generated per-gene-block so a slab of
    the full workload can be produced without materialising the rest).
    """
    if name is not None and name in CONFIGS:
        cn, cp, cl, ci, cK, clam, calpha, ctun, cf = CONFIGS[name]
        n = n or cn
        p = p or cp
        level_counts = level_counts or cl
        interaction_idx = interaction_idx if interaction_idx is not None else ci
        K = K or cK
        lam, alpha, tuning, f = clam, calpha, ctun, cf
    name = name or "custom"
    levels = cyclic_levels(n, level_counts)
    if interaction_idx:
        levels = interaction_indicator(levels, interaction_idx)
    n_levels = np.array([levels[:, i].max() for i in range(levels.shape[1])], dtype=np.int32)
    for i in range(levels.shape[1]):  # level ids must be exactly 1..L_i (src/optimize.cpp:175,286)
        assert np.array_equal(np.unique(levels[:, i]), np.arange(1, n_levels[i] + 1))
    kt = k_true or K
    rng = np.random.Generator(np.random.PCG64(data_seed))
    Astar = [rng.standard_normal((int(L), kt)) for L in n_levels]
    Rstar = sum(Astar[i][levels[:, i] - 1, :] for i in range(levels.shape[1]))
    Cstar = rng.standard_normal((kt, p))
    Cstar[:, rng.choice(p, size=int(0.3 * p), replace=False)] = 0.0
    lo, hi = gene_range if gene_range is not None else (0, p)
    # noise is drawn per 1024-gene block from a block-keyed stream so any slab reproduces the full matrix
    X = np.empty((n, hi - lo), dtype=np.float64, order="F")
    blk = GENE_BLOCK

    def one(b):
        b0, b1 = b * blk, min((b + 1) * blk, p)
        brng = np.random.Generator(np.random.PCG64([data_seed, 1, b]))
        noise = brng.standard_normal((b1 - b0, n)).T      # Fortran-ordered n x width block
        s0, s1 = max(b0, lo), min(b1, hi)
        X[:, s0 - lo:s1 - lo] = Rstar @ Cstar[:, s0:s1] + noise[:, s0 - b0:s1 - b0]

    _pmap(one, range(lo // blk, (hi + blk - 1) // blk))
    if tuning == 1 and f > 0:
        Mtr, Mte = holdout_masks(n, p, f, mask_seed, gene_range=(lo, hi))
    else:
        Mtr = np.ones((n, hi - lo), dtype=np.uint8, order="F")
        Mte = np.zeros((n, hi - lo), dtype=np.uint8, order="F")
    A0, C0 = init_factors(n_levels, K, p, init_seed)
    if gene_range is not None:
        C0 = np.asfortranarray(C0[:, lo:hi])
    return Workload(name=name, X=X, levels=levels, n_levels=n_levels, M_train=Mtr, M_test=Mte, K=K, lam=lam,
                    alpha=alpha, tuning=tuning, A0=A0, C0=C0)


def small(n=40, p=60, level_counts=(5, 4), K=4, f=0.15, lam=2.0, alpha=0.4, tuning=1, seed=1, interaction_idx=None,
          with_na=False):
    """A tiny workload for the pure-Python / parity tests."""
    w = make(n=n, p=p, level_counts=level_counts, interaction_idx=interaction_idx, K=K, lam=lam, alpha=alpha,
             tuning=tuning, f=f, data_seed=seed, mask_seed=seed + 1, init_seed=seed + 2)
    if with_na:  # NA entries: x = 0, excluded from train AND test (R/utils.R:84-86, R/insider.R:26)
        rng = np.random.Generator(np.random.PCG64(seed + 3))
        na = rng.random((n, p)) < 0.05
        w.X[na] = 0.0
        w.M_train[na] = 0
        w.M_test[na] = 0
    return w
