"""Host-side mirror of the reference's operator interface for the hot path.

Operator level (names and argument meaning of /root/reference/R/RcppExports.R:4-22):
    optimize(data, cfd_factors, column_factor, cfd_indicators, ctns_confounder, train_indicator,
             test_indicator, inc_continuous, latent_dim, lambda1, lambda2, alpha, tuning, global_tol,
             sub_tol, max_iter)
    strong_coordinate_descent(X, y, wstart, lambda_, alpha, XtX, Xty, tol)
    optimize_continuous_v2(data, indicator, updating_factor, c_factor, updating_confd, gram, lambda_, tuning)
Caller level (R/insider.R:18-216, R/utils.R:40-43,78-117) — R is absent from this pipeline, so the R S3 API is
mirrored here in Python with the same names, defaults and error behaviour:
    insider(), tune(), fit(), ratio_splitter(), init_parameters()

All compute goes through libinsider_hip.so (insider_amd/_lib.py); nothing here falls back to the CPU.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import InsiderError

DEFAULT_SEED = 0x1D5EED


class InsiderData:
    """A data set resident in HBM (insider_hip_create). Reused across optimize() calls, e.g. by tune()'s grid."""

    def __init__(self, data, cfd_indicators, train_indicator, test_indicator, device=0, n_levels=None,
                 ctns_confounder=None):
        lib = _lib.load()
        X = _lib.f64(data)
        n, p = X.shape
        lev = np.asfortranarray(np.asarray(cfd_indicators).reshape(n, -1), dtype=np.int32)
        c = lev.shape[1]
        if n_levels is None:
            # R/insider.R:107: factor_num <- length(unique(confounder[, i])); ids must be exactly 1..L_i
            n_levels = np.array([len(np.unique(lev[:, i])) for i in range(c)], dtype=np.int32)
        n_levels = np.ascontiguousarray(n_levels, dtype=np.int32)
        Mtr = np.asfortranarray(train_indicator, dtype=np.uint8)
        Mte = np.asfortranarray(test_indicator, dtype=np.uint8)
        if Mtr.shape != (n, p) or Mte.shape != (n, p):
            raise InsiderError(_lib.ERR_ARG, "indicator shape must match data")
        self.n, self.p, self.c = n, p, c
        self.n_levels = n_levels
        self._h = C.c_void_p()
        if ctns_confounder is not None:
            Z = _lib.f64(np.asarray(ctns_confounder, dtype=np.float64).reshape(n, -1))
            self.m = Z.shape[1]
            zp = _lib.ptr(Z)
        else:
            self.m, zp = 0, None
        _lib.check(lib.insider_hip_create_ex(_lib.ptr(X), n, p, _lib.ptr(lev, C.c_int32), c,
                                             _lib.ptr(n_levels, C.c_int32), zp, self.m, _lib.ptr(Mtr, C.c_uint8),
                                             _lib.ptr(Mte, C.c_uint8), int(device), C.byref(self._h)))
        self._cb = None  # keeps the ctypes callback alive
        # measurement knob: INSIDER_HIP_OPTIONS="name=value,name=value" applies library options to every handle this process
        # creates (A/B of an option through tools that do not expose it)
        for kv in filter(None, os.environ.get("INSIDER_HIP_OPTIONS", "").split(",")):
            name, value = kv.split("=")
            self.set_option(name.strip(), float(value))

    def clone(self):
        """Another handle on the SAME resident data set (insider_hip_clone): the device copy of X, the lists and the
        count tables are shared, the factor workspace, streams and options (copied as they stand) are its own.  Handles of
        one data set may fit at the same time from different threads (tune(concurrent=k))."""
        other = object.__new__(InsiderData)
        other.n, other.p, other.c, other.m, other.n_levels = self.n, self.p, self.c, self.m, self.n_levels
        other._h = C.c_void_p()
        other._cb = None
        other._options = dict(getattr(self, "_options", {}))       # the library copies the options as they stand
        _lib.check(_lib.load().insider_hip_clone(self._h, C.byref(other._h)))
        return other

    def set_option(self, name, value):
        _lib.check(_lib.load().insider_hip_set_option(self._h, name.encode(), float(value)))
        self.__dict__.setdefault("_options", {})[name] = float(value)      # (what tune(concurrent=k) re-applies to its clones)

    def set_shard(self, gene_offset, rank, world, allreduce=None):
        """allreduce(ptr:int, count:int, stream:int) -> None sums `count` doubles at device pointer `ptr` across ranks in
        place, ordered against the library's HIP stream `stream` (see include/insider_hip.h)."""
        if allreduce is None:
            cb = C.cast(None, _lib.ALLREDUCE_FN)
        else:
            def _tramp(_user, ptr, count, stream):
                try:
                    allreduce(int(ptr), int(count), int(stream or 0))
                    return 0
                except Exception as e:  # never unwind through the C frame
                    print(f"[insider_amd] all-reduce callback failed: {e!r}", flush=True)
                    return 1
            cb = _lib.ALLREDUCE_FN(_tramp)
        self._cb = cb
        _lib.check(_lib.load().insider_hip_set_shard(self._h, int(gene_offset), int(rank), int(world), cb, None))

    def _marshal(self, cfd_factors, column_factor, K, inc_continuous):
        """F-ordered float64 views (or copies) of the factors plus the pointer array the C ABI takes."""
        A = []
        shapes = [int(L) for L in self.n_levels] + ([self.m] if inc_continuous else [])
        if len(cfd_factors) != len(shapes):
            raise InsiderError(_lib.ERR_ARG, f"expected {len(shapes)} row-factor matrices, got {len(cfd_factors)}")
        for i, a in enumerate(cfd_factors):
            a = np.asarray(a)
            if a.shape != (shapes[i], K):
                raise InsiderError(_lib.ERR_ARG, f"cfd_factors[{i}] must be {shapes[i]} x {K}")
            A.append(a if (a.dtype == np.float64 and a.flags.f_contiguous) else _lib.f64(a).copy(order="F"))
        Cm = np.asarray(column_factor)
        if Cm.shape != (K, self.p):
            raise InsiderError(_lib.ERR_ARG, f"column_factor must be {K} x {self.p}")
        Cw = Cm if (Cm.dtype == np.float64 and Cm.flags.f_contiguous) else _lib.f64(Cm).copy(order="F")
        Aptrs = (C.POINTER(C.c_double) * len(A))(*[_lib.ptr(a) for a in A])
        return A, Cw, Aptrs

    def optimize_row(self, cfd_factors, column_factor, cov, lambda_=1.0, tuning=1, inc_continuous=0):
        """One row update of covariate `cov` (optimize_row, src/optimize.cpp:139-198 as called at :339); returns the
        new L_cov x K factor and updates cfd_factors[cov] in place when it is an F-ordered float64 array."""
        K = int(np.asarray(column_factor).shape[0])
        A, Cw, Aptrs = self._marshal(cfd_factors, column_factor, K, inc_continuous)
        _lib.check(_lib.load().insider_hip_optimize_row(self._h, Aptrs, _lib.ptr(Cw), int(inc_continuous), K, int(cov),
                                                        float(lambda_), int(tuning)))
        i = min(int(cov), len(A) - 1)
        if A[i] is not cfd_factors[i] and isinstance(cfd_factors[i], np.ndarray):
            cfd_factors[i][...] = A[i]
        return A[i].copy()

    def optimize_col(self, cfd_factors, column_factor, lambda_=1.0, alpha=0.1, tuning=1, tol=1e-5, seed=DEFAULT_SEED,
                     it=0, inc_continuous=0):
        """One column update (optimize_col, src/optimize.cpp:200-253 as called at :376); returns the new K x p factor
        and updates column_factor in place when it is an F-ordered float64 array."""
        K = int(np.asarray(column_factor).shape[0])
        A, Cw, Aptrs = self._marshal(cfd_factors, column_factor, K, inc_continuous)
        _lib.check(_lib.load().insider_hip_optimize_col(self._h, Aptrs, _lib.ptr(Cw), int(inc_continuous), K,
                                                        float(lambda_), float(alpha), int(tuning), float(tol), int(seed),
                                                        int(it)))
        if Cw is not column_factor and isinstance(column_factor, np.ndarray):
            column_factor[...] = Cw
        return Cw.copy()

    def optimize(self, cfd_factors, column_factor, latent_dim, lambda1=1.0, lambda2=1.0, alpha=0.1, tuning=1,
                 global_tol=1e-10, sub_tol=1e-5, max_iter=10000, seed=DEFAULT_SEED, inc_continuous=0, traj_cap=4096,
                 copy=True):
        """copy=False: the returned factors ARE the (updated in place) arguments instead of copies of them — what the C ABI
        itself does; the reference's List holds copies (src/optimize.cpp:413), hence the default."""
        lib = _lib.load()
        K = int(latent_dim)
        A, Cw, Aptrs = self._marshal(cfd_factors, column_factor, K, inc_continuous)
        traj = np.full((traj_cap, _lib.TRAJ_STRIDE), np.nan)
        tr, te, lo = C.c_double(), C.c_double(), C.c_double()
        rows, iters = C.c_int(), C.c_int()
        _lib.check(lib.insider_hip_optimize(self._h, Aptrs, _lib.ptr(Cw), int(inc_continuous), K, float(lambda1),
                                            float(lambda2), float(alpha), int(tuning), float(global_tol),
                                            float(sub_tol), int(max_iter), int(seed), C.byref(tr), C.byref(te),
                                            C.byref(lo), _lib.ptr(traj), traj_cap, C.byref(rows), C.byref(iters)))
        # the reference mutates cfd_factors / column_factor in place AND returns copies (src/optimize.cpp:283-284,413)
        for src, dst in zip(A, cfd_factors):
            if src is not dst and isinstance(dst, np.ndarray):
                dst[...] = src
        if Cw is not column_factor and isinstance(column_factor, np.ndarray):
            column_factor[...] = Cw
        return dict(row_matrices={f"factor{i}": (a.copy() if copy else a) for i, a in enumerate(A)},
                    column_factor=Cw.copy() if copy else Cw,
                    train_rmse=tr.value, test_rmse=te.value, loss=lo.value, traj=traj[: rows.value].copy(),
                    iters=iters.value)

    def masked_gram_cols(self, R):
        R = _lib.f64(R)
        K = R.shape[1]
        G = np.zeros((self.p, K, K))
        q = np.zeros((self.p, K))
        _lib.check(_lib.load().insider_hip_masked_gram_cols(self._h, _lib.ptr(R), K, _lib.ptr(G), _lib.ptr(q)))
        return G, q

    def masked_gram_rows(self, Cmat):
        Cmat = _lib.f64(Cmat)
        K = Cmat.shape[0]
        H = np.zeros((self.n, K, K))
        b = np.zeros((self.n, K))
        _lib.check(_lib.load().insider_hip_masked_gram_rows(self._h, _lib.ptr(Cmat), K, _lib.ptr(H), _lib.ptr(b)))
        return H, b

    def profile(self):
        out = np.zeros(12)
        _lib.check(_lib.load().insider_hip_get_profile(self._h, _lib.ptr(out)))
        return dict(col_stats_launches=int(out[0]), col_stats_ms=out[1], row_stats_launches=int(out[2]),
                    row_stats_ms=out[3], cd_launches=int(out[4]), cd_ms=out[5], test_launches=int(out[6]),
                    test_ms=out[7], wall_ms=out[8], iters=int(out[9]), sweeps=int(out[10]),
                    col_factored=bool(int(out[11]) & 1), row_merged=bool(int(out[11]) & 2),
                    col_pair=bool(int(out[11]) & 4))

    def info(self, name):
        """A fact about the handle (insider_hip_get_info): "col_stats_path", "col_mfma_per_gene", ..."""
        out = C.c_double()
        _lib.check(_lib.load().insider_hip_get_info(self._h, name.encode(), C.byref(out)))
        return out.value

    def comm_init(self, unique_id, rank, world):
        """Join the in-library RCCL communicator (insider_hip_comm_init) after set_shard(); collective over all ranks."""
        buf = (C.c_char * _lib.COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _lib.check(_lib.load().insider_hip_comm_init(self._h, buf, int(rank), int(world)))

    def debug_array(self, name):
        """An internal per-gene int32 array (insider_hip_get_array): "cd_key0", "cd_key1", "gene_perm"."""
        out = np.zeros(self.p, dtype=np.int32)
        _lib.check(_lib.load().insider_hip_get_array(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def sweeps(self):
        """Per-gene sweep counts of the last column update."""
        out = np.zeros(self.p, dtype=np.int32)
        _lib.check(_lib.load().insider_hip_get_sweeps(self._h, _lib.ptr(out, C.c_int32)))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.load().insider_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------------------
# operator level — R/RcppExports.R:4-22
# ---------------------------------------------------------------------------------------------------------------
def optimize(data, cfd_factors, column_factor, cfd_indicators, ctns_confounder, train_indicator, test_indicator,
             inc_continuous, latent_dim, lambda1=1.0, lambda2=1.0, alpha=0.1, tuning=1, global_tol=1e-10, sub_tol=1e-5,
             max_iter=10000, seed=DEFAULT_SEED, device=0):
    """optimize() of R/RcppExports.R:20-22 (src/optimize.cpp:255-422): one-shot upload + fit, through
    ``insider_hip_optimize_oneshot_ex`` — the symbol r/insider_hip_shim.c binds for the R package.

    Returns dict(row_matrices, column_factor, train_rmse, test_rmse, loss) like the reference's List (:417-421);
    float64 Fortran-ordered ``cfd_factors`` / ``column_factor`` arrays are also updated in place (:283-284).
    """
    if tuning not in (0, 1):
        raise InsiderError(_lib.ERR_ARG, "Parameter tuning should be either 0 or 1!")
    if inc_continuous not in (0, 1):
        raise InsiderError(_lib.ERR_ARG, "The value of prarameter inc_continuous can only be 0 or 1.")
    lib = _lib.load()
    X = _lib.f64(data)
    n, p = X.shape
    K = int(latent_dim)
    lev = np.asfortranarray(np.asarray(cfd_indicators).reshape(n, -1), dtype=np.int32)
    c = lev.shape[1]
    n_levels = np.ascontiguousarray([len(np.unique(lev[:, i])) for i in range(c)], dtype=np.int32)   # R/insider.R:107
    Mtr = np.asfortranarray(train_indicator, dtype=np.uint8)
    Mte = np.asfortranarray(test_indicator, dtype=np.uint8)
    if Mtr.shape != (n, p) or Mte.shape != (n, p):
        raise InsiderError(_lib.ERR_ARG, "indicator shape must match data")
    if inc_continuous == 1:
        if ctns_confounder is None:
            raise InsiderError(_lib.ERR_ARG, "inc_continuous = 1 needs ctns_confounder (n x m)")
        Z = _lib.f64(np.asarray(ctns_confounder, dtype=np.float64).reshape(n, -1))
        m, zp = Z.shape[1], _lib.ptr(Z)
    else:
        m, zp = 0, None
    shapes = [int(L) for L in n_levels] + ([m] if inc_continuous else [])
    if len(cfd_factors) != len(shapes):
        raise InsiderError(_lib.ERR_ARG, f"expected {len(shapes)} row-factor matrices, got {len(cfd_factors)}")
    A = []
    for i, a in enumerate(cfd_factors):
        a = np.asarray(a)
        if a.shape != (shapes[i], K):
            raise InsiderError(_lib.ERR_ARG, f"cfd_factors[{i}] must be {shapes[i]} x {K}")
        A.append(a if (a.dtype == np.float64 and a.flags.f_contiguous) else _lib.f64(a).copy(order="F"))
    Cm = np.asarray(column_factor)
    if Cm.shape != (K, p):
        raise InsiderError(_lib.ERR_ARG, f"column_factor must be {K} x {p}")
    Cw = Cm if (Cm.dtype == np.float64 and Cm.flags.f_contiguous) else _lib.f64(Cm).copy(order="F")
    Aptrs = (C.POINTER(C.c_double) * len(A))(*[_lib.ptr(a) for a in A])
    tr, te, lo = C.c_double(), C.c_double(), C.c_double()
    _lib.check(lib.insider_hip_optimize_oneshot_ex(_lib.ptr(X), n, p, Aptrs, _lib.ptr(Cw), _lib.ptr(lev, C.c_int32), c,
                                                   _lib.ptr(n_levels, C.c_int32), zp, m, _lib.ptr(Mtr, C.c_uint8),
                                                   _lib.ptr(Mte, C.c_uint8), int(inc_continuous), K, float(lambda1),
                                                   float(lambda2), float(alpha), int(tuning), float(global_tol),
                                                   float(sub_tol), int(max_iter), int(seed), int(device), C.byref(tr),
                                                   C.byref(te), C.byref(lo)))
    for src, dst in zip(A, cfd_factors):
        if src is not dst and isinstance(dst, np.ndarray):
            dst[...] = src
    if Cw is not column_factor and isinstance(column_factor, np.ndarray):
        column_factor[...] = Cw
    return dict(row_matrices={f"factor{i}": a.copy() for i, a in enumerate(A)}, column_factor=Cw.copy(),
                train_rmse=tr.value, test_rmse=te.value, loss=lo.value)


def strong_coordinate_descent(X, y, wstart, lambda_, alpha, XtX=None, Xty=None, tol=1e-5, seed=DEFAULT_SEED, it=0,
                              order_mode=0, max_sweeps=1 << 24, device=0, return_sweeps=False):
    """strong_coordinate_descent() of R/RcppExports.R:8-10 (src/coordinate_descent.cpp:56-127).

    Single problem (the reference's signature): ``X`` m x K, ``y`` m; ``XtX`` / ``Xty`` are formed on the device as X'X
    and X'y when they are not given (``insider_hip_strong_cd_xy``).  Batched use: XtX of shape (B, K, K), Xty / wstart
    of shape (B, K); X and y are then not read (covariance form).
    """
    w = np.ascontiguousarray(wstart, dtype=np.float64)
    if XtX is None or Xty is None or np.ndim(Xty) == 1:
        if X is None and (XtX is None or Xty is None):
            raise InsiderError(_lib.ERR_ARG, "pass (X, y), or XtX and Xty")
        K = int(w.shape[0])
        Xf = _lib.f64(X) if X is not None else None
        yf = np.ascontiguousarray(y, dtype=np.float64) if y is not None else None
        if Xf is not None and (Xf.ndim != 2 or Xf.shape[1] != K or yf is None or yf.shape != (Xf.shape[0],)):
            raise InsiderError(_lib.ERR_ARG, "X must be m x K and y of length m")
        G = _lib.f64(XtX) if XtX is not None else None
        q = np.ascontiguousarray(Xty, dtype=np.float64) if Xty is not None else None
        if (G is not None and G.shape != (K, K)) or (q is not None and q.shape != (K,)):
            raise InsiderError(_lib.ERR_ARG, "XtX must be K x K and Xty of length K")
        beta = np.zeros(K)
        sw = np.zeros(1, dtype=np.int32)
        _lib.check(_lib.load().insider_hip_strong_cd_xy(
            _lib.ptr(Xf) if Xf is not None else None, _lib.ptr(yf) if yf is not None else None,
            int(Xf.shape[0]) if Xf is not None else 0, K, _lib.ptr(w), float(lambda_), float(alpha),
            _lib.ptr(G) if G is not None else None, _lib.ptr(q) if q is not None else None, float(tol), int(seed), int(it),
            int(order_mode), int(max_sweeps), int(device), _lib.ptr(beta), _lib.ptr(sw, C.c_int32)))
        return (beta, int(sw[0])) if return_sweeps else beta
    G = np.ascontiguousarray(XtX, dtype=np.float64)
    q = np.ascontiguousarray(Xty, dtype=np.float64)
    B, K = q.shape
    if G.shape != (B, K, K) or w.shape != (B, K):
        raise InsiderError(_lib.ERR_ARG, "XtX must be (B,K,K) and Xty/wstart (B,K)")
    beta = np.zeros((B, K))
    sw = np.zeros(B, dtype=np.int32)
    _lib.check(_lib.load().insider_hip_strong_cd(_lib.ptr(G), _lib.ptr(q), _lib.ptr(w), K, B, float(lambda_),
                                                 float(alpha), float(tol), int(seed), int(it),
                                                 int(order_mode), int(max_sweeps), int(device), _lib.ptr(beta),
                                                 _lib.ptr(sw, C.c_int32)))
    return (beta, sw) if return_sweeps else beta


def optimize_continuous_v2(data, indicator, updating_factor, c_factor, updating_confd, gram, lambda_, tuning, device=0):
    """optimize_continuous_v2() of R/RcppExports.R:16-18 (src/optimize.cpp:76-137), the reference's eight arguments, through
    ``insider_hip_optimize_continuous_v2``: the update of one continuous covariate's K-vector against the matrix ``data``
    (n x p; inside optimize() the residual with this column's contribution added back, :344-345).  ``updating_factor`` is
    updated IN PLACE when it is a float64 array (the reference's ``rowvec&``) and returned.  ``indicator`` is read only when
    tuning = 1, ``gram`` only when tuning = 0 — as in the reference."""
    if tuning not in (0, 1):
        raise InsiderError(_lib.ERR_ARG, "Parameter tuning should be either 0 or 1!")
    D = _lib.f64(data)
    n, p = D.shape
    Cm = _lib.f64(c_factor)
    K = Cm.shape[0]
    z = np.ascontiguousarray(np.asarray(updating_confd, dtype=np.float64).reshape(-1))
    u = np.ascontiguousarray(np.asarray(updating_factor, dtype=np.float64).reshape(-1)).copy()
    if Cm.shape != (K, p) or z.shape != (n,) or u.shape != (K,):
        raise InsiderError(_lib.ERR_ARG, "c_factor must be K x p, updating_confd of length n, updating_factor of length K")
    M = g = None
    if tuning == 1:
        M = np.asfortranarray(np.asarray(indicator) != 0, dtype=np.uint8)
        if M.shape != (n, p):
            raise InsiderError(_lib.ERR_ARG, "indicator shape must match data")
    else:
        g = _lib.f64(gram)
        if g.shape != (K, K):
            raise InsiderError(_lib.ERR_ARG, "gram must be K x K")
    _lib.check(_lib.load().insider_hip_optimize_continuous_v2(
        _lib.ptr(D), n, p, _lib.ptr(M, C.c_uint8) if M is not None else None, _lib.ptr(u), _lib.ptr(Cm), K, _lib.ptr(z),
        _lib.ptr(g) if g is not None else None, float(lambda_), int(tuning), int(device)))
    if isinstance(updating_factor, np.ndarray) and updating_factor.dtype == np.float64:
        updating_factor.reshape(-1)[...] = u
    return u


def solve_sympd(A, b, device=0, return_route=False):
    """solve(A, b, solve_opts::likely_sympd) (src/optimize.cpp:175,190,226,240), batched: A (B, K, K) or (K, K), b (B, K)
    or (K,).  Cholesky first, Gaussian elimination with partial pivoting when A is not positive definite."""
    Am = np.asarray(A, dtype=np.float64)
    bm = np.asarray(b, dtype=np.float64)
    single = Am.ndim == 2
    if single:
        Am, bm = Am[None], bm[None]
    B, K = bm.shape
    if Am.shape != (B, K, K):
        raise InsiderError(_lib.ERR_ARG, "A must be (B,K,K) and b (B,K)")
    Af = np.ascontiguousarray(np.transpose(Am, (0, 2, 1)))     # every block column-major
    bf = np.ascontiguousarray(bm)
    x = np.zeros((B, K))
    route = np.zeros(B, dtype=np.int32)
    _lib.check(_lib.load().insider_hip_solve_sympd(_lib.ptr(Af), _lib.ptr(bf), K, B, int(device), _lib.ptr(x),
                                                   _lib.ptr(route, C.c_int32)))
    if single:
        x, route = x[0], int(route[0])
    return (x, route) if return_route else x


# ---------------------------------------------------------------------------------------------------------------
# caller level — R/insider.R, R/utils.R
# ---------------------------------------------------------------------------------------------------------------
def init_parameters(size, init_mean=0.0, init_std=0.001, rng=None):
    """R/utils.R:40-43: rnorm(size, mean, sd). ``rng`` replaces R's global RNG."""
    rng = rng if rng is not None else np.random.default_rng()
    return rng.normal(init_mean, init_std, size=size)


def ratio_splitter(data, ratio=0.1, rm_na_col=True, seed=123):
    """R/utils.R:78-117: element-wise hold-out without replacement; NA -> 0 and excluded; all-zero columns of the
    train set dropped.  (numpy PCG64 stands in for R's set.seed(123); sample().)"""
    data = np.array(data, dtype=np.float64, order="F")
    na = np.isnan(data)
    data[na] = 0.0
    train = ~na
    rng = np.random.Generator(np.random.PCG64(seed))
    existing = np.flatnonzero((~na).ravel(order="F"))
    k = int(np.floor(existing.size * ratio))
    test_idx = rng.choice(existing, size=k, replace=False)
    test = np.zeros(data.size, dtype=bool)
    test[test_idx] = True
    test = test.reshape(data.shape, order="F")
    testset = np.where(test, data, 0.0)
    trainset = np.where(test, 0.0, data)
    train &= ~test
    num_per_col = (trainset != 0).sum(axis=0)
    print(f"number of all zero columns removed: {int((num_per_col == 0).sum())}")
    keep = num_per_col != 0 if rm_na_col else np.ones(data.shape[1], dtype=bool)
    return dict(trainset=trainset[:, keep], testset=testset[:, keep], train_indicator=train[:, keep],
                test_indicator=test[:, keep], na_indicator=na[:, keep], kept_columns=keep)


class Insider(dict):
    """The reference's S3 object of class "insider" (a list, R/insider.R:24)."""


def insider(data, confounder, ctns_confounder=None, interaction_idx=None, split_ratio=0.1, global_tol=1e-9,
            sub_tol=1e-5, tuning_iter=30, max_iter=50000, device=0, seed=DEFAULT_SEED):
    """insider() of R/insider.R:18-67."""
    data = np.asarray(data, dtype=np.float64)
    confounder = np.asarray(confounder)
    if confounder.ndim == 1:
        confounder = confounder[:, None]
    dataset = ratio_splitter(data, ratio=split_ratio)
    obj = Insider()
    keep = dataset["kept_columns"]
    d = np.array(data[:, keep], dtype=np.float64, order="F")
    d[np.isnan(d)] = 0.0                                      # R/insider.R:26 (intent: NA -> 0)
    obj["data"] = d
    if interaction_idx is not None and len(interaction_idx) > 1 and \
            all(isinstance(v, (int, np.integer)) for v in interaction_idx):
        if max(interaction_idx) > confounder.shape[1]:
            raise ValueError("The interaction_idx is out of the range of confounder!")   # R/insider.R:30-32
        from .workloads import interaction_indicator
        obj["confounder"] = interaction_indicator(confounder.astype(np.int32), tuple(interaction_idx))  # :34-40
    elif interaction_idx is None:
        obj["confounder"] = np.asfortranarray(confounder, dtype=np.int32)                 # :43
    else:
        raise ValueError("The interaction_idx should be integers and its length must be greater than or equal to 2!")
    if ctns_confounder is not None:
        obj["inc_continuous"] = 1
        obj["ctns_confounder"] = np.asarray(ctns_confounder, dtype=np.float64)
    else:
        obj["inc_continuous"] = 0
        obj["ctns_confounder"] = np.zeros((confounder.shape[0], 1))
    obj["train_indicator"] = np.asfortranarray(dataset["train_indicator"], dtype=np.uint8)   # :57-59
    obj["test_indicator"] = np.asfortranarray(dataset["test_indicator"], dtype=np.uint8)
    obj["na_indicator"] = np.asfortranarray(dataset["na_indicator"], dtype=np.uint8)
    obj["params"] = dict(global_tol=global_tol, sub_tol=sub_tol, tuning_iter=tuning_iter, max_iter=max_iter)
    obj["device"] = device
    obj["seed"] = seed
    return obj


def _resident(obj, which):
    """HBM-resident data set for the tune (train/test masks) or fit (train+test / NA masks) call pattern."""
    key = "_resident_" + which
    if key not in obj:
        if which == "tune":
            tr, te = obj["train_indicator"], obj["test_indicator"]
        else:  # R/insider.R:207-208: indicator = train + test, "test" = NA mask
            tr, te = obj["train_indicator"] + obj["test_indicator"], obj["na_indicator"]
        obj[key] = InsiderData(obj["data"], obj["confounder"], tr, te, device=obj.get("device", 0),
                               ctns_confounder=obj["ctns_confounder"] if obj["inc_continuous"] == 1 else None)
    return obj[key]


def _fresh_inits(obj, latent_rank, rng):
    conf = obj["confounder"]
    cfd = [np.asfortranarray(init_parameters(len(np.unique(conf[:, i])) * latent_rank, rng=rng)
                             .reshape((-1, latent_rank), order="F")) for i in range(conf.shape[1])]     # :106-109
    if obj["inc_continuous"] == 1:
        cfd.append(np.asfortranarray(init_parameters(obj["ctns_confounder"].shape[1] * latent_rank, rng=rng)
                                     .reshape((-1, latent_rank), order="F")))                            # :111-113
    col = np.asfortranarray(init_parameters(latent_rank * obj["data"].shape[1], rng=rng)
                            .reshape((latent_rank, -1), order="F"))                                      # :114
    return cfd, col


def _grid_sum(rows, world):
    """Combine the per-rank result tables of a grid-parallel tune(): every rank filled only its own rows."""
    if world <= 1:
        return rows
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(rows))
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def _nearest_finished(done, g, n_lambda):
    """Index of the finished grid point closest to g on the (alpha row, lambda column) lattice (expand.grid order: lambda
    fastest), ties to the lower index; None when nothing is finished yet."""
    best, bd = None, None
    for h in done:
        d = abs(h // n_lambda - g // n_lambda) + abs(h % n_lambda - g % n_lambda)
        if bd is None or d < bd or (d == bd and h < best):
            best, bd = h, d
    return best


def _tune_handles(obj, ds, k):
    """k handles on the resident tune() data set: the data set's own plus k - 1 clones (insider_hip_clone: shared device
    arrays, private workspaces), kept on the object for the next call."""
    # clones whose handle has been closed since (bench.py closes them between grids) are dropped; the others are dropped too
    # when they belong to another data set handle than `ds` (a re-created resident data set)
    clones = [hd for hd in obj.get("_tune_clones", []) if getattr(hd, "_h", None) and getattr(hd, "_src", None) is ds]
    while len(clones) < k - 1:
        hd = ds.clone()
        hd._src = ds
        clones.append(hd)
    obj["_tune_clones"] = clones
    # a clone copies its source's options as they stood at clone time: bring the ones set on `ds` since then across
    for hd in clones[: k - 1]:
        for name, value in getattr(ds, "_options", {}).items():
            if getattr(hd, "_options", {}).get(name) != value:
                hd.set_option(name, value)
    return [ds] + clones[: k - 1]


def tune(obj, latent_dimension=None, lambda_=0.1, alpha=0.0, out_dir=None, rng=None, rank=0, world=1, timings=None,
         warm_start=False, concurrent=1):
    """tune() of R/insider.R:81-176.  ``out_dir``: where to write the reference's CSVs (None = do not write).

    ``warm_start`` (opt-in, NOT the reference's behaviour, which draws fresh N(0, 0.001^2) inits for every grid point,
    R/insider.R:152-161): the fit of (lambda, alpha) grid point g starts from the fitted factors of the nearest grid point
    this rank has already finished instead of from its fresh draw (which is still drawn, so the generator state — and
    every point that does start cold — is unchanged).  The first outer iterations of a cold fit run thousands of
    coordinate sweeps per gene to leave the near-zero inits; a neighbouring optimum is a few hundred sweeps away.
    IT CHANGES THE ANSWER, not only the time: a tuning_iter = 30 fit is not converged, so a fit that starts near a
    neighbour's optimum ends elsewhere than the cold fit of the same point, and the grid's argmin may move.  Measured
    (profiles/r04/grid_c3_k2.json, grid_c2_k4.json): the selected (lambda, alpha) DIFFERED from the cold grid's at config 3
    and at config 2 (`best_point_agrees_with_cold` false in both; largest test-RMSE difference over the 40 points 2.4e-5
    and 4e-4).  Use it to explore a grid quickly; re-run the chosen neighbourhood cold before reporting a selection.

    ``rank`` / ``world``: grid-parallel tuning across the GPUs of a node: every rank keeps the
    whole data set resident, grid point g is fitted by rank g % world and the result tables are summed over
    torch.distributed.  The fresh inits of ALL grid points are drawn on every rank, in the reference's order, so
    the tables do not depend on ``world``.  ``timings``: a list that receives one dict per fitted point
    (init_s = drawing the fresh inits, optimize_s = the optimize() call, library_ms = time inside the library).

    ``concurrent`` = k > 1: k grid points of this rank are fitted AT THE SAME TIME on the one GPU, each on its own handle of
    the shared resident data set (InsiderData.clone) from its own host thread.  The grid points are independent fits
    (R/insider.R:145-164), and a data set of the size real INSIDER inputs have does not fill an MI355X with one fit: its
    column step is bound by its longest gene's sequential sweep chain, which a second fit overlaps.  The inits are still
    drawn by ONE generator in the reference's order, and every point's result is bit-identical to the serial grid's
    (tests/test_gpu_parity.py::test_concurrent_tune_is_bit_identical).  Not combinable with ``warm_start`` (whose
    starting points depend on the order in which fits finish)."""
    if concurrent > 1 and warm_start:
        raise ValueError("tune(): concurrent > 1 and warm_start exclude each other")
    import time as _time
    lat = np.atleast_1d(latent_dimension) if latent_dimension is not None else np.array([])
    lam = np.atleast_1d(np.asarray(lambda_, dtype=float))
    alp = np.atleast_1d(np.asarray(alpha, dtype=float))
    if lat.size == 0 or not np.issubdtype(lat.dtype, np.integer):
        raise ValueError("TUNNING: The element of latent_dimension, lambda, and alpha should be integer, numeric, "
                         "and numeric.")                                                                 # :83-85
    if lat.size <= 1 and lam.size <= 1 and alp.size <= 1:
        raise ValueError("TUNNING: The length of either latent_dimension or lambda and alpha should be greater "
                         "than 1.")                                                                      # :87-89
    prm = obj["params"]
    rng = rng if rng is not None else np.random.default_rng(obj.get("seed", DEFAULT_SEED))
    ds = _resident(obj, "tune")
    rank_tuning, reg_tuning = None, None
    if lat.size > 1:                                                                                     # :98-132
        rows = np.zeros((lat.size, 3))
        for g, latent_rank in enumerate(lat):
            cfd, col = _fresh_inits(obj, int(latent_rank), rng)
            if g % world != rank:
                continue
            print(f"Latent rank:  {int(latent_rank)} ---------------------------------")
            if lam.size == 1 and alp.size == 1:
                l_, a_ = float(lam[0]), float(alp[0])
            else:
                l_, a_ = 0.1, 0.0                                                                        # :120-121
            fitted = ds.optimize(cfd, col, int(latent_rank), l_, l_, a_, 1, prm["global_tol"], prm["sub_tol"],
                                 prm["tuning_iter"], seed=obj.get("seed", DEFAULT_SEED),
                                 inc_continuous=obj["inc_continuous"])
            rows[g] = (int(latent_rank), fitted["train_rmse"], fitted["test_rmse"])
            if out_dir is not None and world == 1:
                np.savetxt(os.path.join(out_dir, "insider_rank_tuning_result.csv"), rows[: g + 1], delimiter=",")
        rank_tuning = _grid_sum(rows, world)
        if out_dir is not None and world > 1 and rank == 0:
            np.savetxt(os.path.join(out_dir, "insider_rank_tuning_result.csv"), rank_tuning, delimiter=",")
        latent_rank = int(lat[int(np.argmin(rank_tuning[:, 2]))])                                        # :136
    else:
        latent_rank = int(lat[0])
    if lam.size > 1 or alp.size > 1:                                                                     # :142-174
        grid = [(round(float(l_), 2), round(float(a_), 2)) for a_ in alp for l_ in lam]   # expand.grid: lambda fastest
        rows = np.zeros((len(grid), 4))
        csv = os.path.join(out_dir, f"insider_R{latent_rank}_reg_tuning_result.csv") if out_dir is not None else None
        # the fresh inits of grid point g + 1 are drawn (same generator, same order: R/insider.R:152-161) on a helper
        # thread while the GPU fits point g: numpy's generator and the ctypes call both release the GIL
        from concurrent.futures import ThreadPoolExecutor

        def _draw():
            t_0 = _time.perf_counter()
            v = _fresh_inits(obj, latent_rank, rng)
            return v, _time.perf_counter() - t_0

        if concurrent > 1:
            import queue
            import threading
            handles = _tune_handles(obj, ds, int(concurrent))
            ready = queue.Queue(maxsize=2 * len(handles))      # inits drawn ahead of the fits, in the reference's order
            lock = threading.Lock()
            errors = []

            def _producer():
                try:
                    for g in range(len(grid)):
                        if errors:
                            break
                        v, t_draw = _draw()
                        if g % world == rank:
                            ready.put((g, v, t_draw))
                except BaseException as e:      # a failed draw (MemoryError, ...) must not leave the workers waiting for ever
                    errors.append(e)
                finally:
                    for _ in handles:           # one sentinel per worker, whatever happened above
                        ready.put(None)

            def _worker(hd):
                while True:
                    item = ready.get()
                    if item is None or errors:
                        if item is not None:
                            continue        # drain after a failure elsewhere
                        return
                    g, (cfd, col), t_draw = item
                    l_r, a_r = grid[g]
                    t_1 = _time.perf_counter()
                    try:
                        fitted = hd.optimize(cfd, col, latent_rank, l_r, l_r, a_r, 1, prm["global_tol"], prm["sub_tol"],
                                             prm["tuning_iter"], seed=obj.get("seed", DEFAULT_SEED),
                                             inc_continuous=obj["inc_continuous"])
                    except Exception as e:      # reported by the caller's thread
                        errors.append(e)
                        continue
                    with lock:
                        print(f"parameter grid: {l_r},{a_r} ---------------------------------")
                        rows[g] = (l_r, a_r, fitted["train_rmse"], fitted["test_rmse"])
                        if timings is not None:
                            timings.append(dict(lambda_=l_r, alpha=a_r, init_s=t_draw, init_wait_s=0.0, warm_from=None,
                                                optimize_s=_time.perf_counter() - t_1, library_ms=hd.profile()["wall_ms"]))

            threads = [threading.Thread(target=_producer)] + [threading.Thread(target=_worker, args=(hd,)) for hd in handles]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errors:
                raise errors[0]
            if csv and world == 1:
                np.savetxt(csv, rows, delimiter=",")
        finished = {}       # warm_start: grid index -> (row factors, column factor) of the points this rank has fitted
        with ThreadPoolExecutor(max_workers=1) as pool:
            nxt = pool.submit(_draw) if concurrent <= 1 else None
            for g, (l_r, a_r) in enumerate(grid if concurrent <= 1 else []):                             # :147-150
                t_w = _time.perf_counter()
                (cfd, col), t_draw = nxt.result()
                t_wait = _time.perf_counter() - t_w
                if g + 1 < len(grid):
                    nxt = pool.submit(_draw)
                if g % world != rank:
                    continue
                t_1 = _time.perf_counter()
                print(f"parameter grid: {l_r},{a_r} ---------------------------------")
                src = _nearest_finished(finished, g, lam.size) if warm_start else None
                if src is not None:
                    cfd = [a.copy(order="F") for a in finished[src][0]]
                    col = finished[src][1].copy(order="F")
                fitted = ds.optimize(cfd, col, latent_rank, l_r, l_r, a_r, 1, prm["global_tol"], prm["sub_tol"],
                                     prm["tuning_iter"], seed=obj.get("seed", DEFAULT_SEED),
                                     inc_continuous=obj["inc_continuous"])
                if warm_start:
                    finished[g] = (list(fitted["row_matrices"].values()), fitted["column_factor"])
                    for old_g in [h for h in finished if h < g - lam.size * world - world]:   # keep about one alpha row back
                        del finished[old_g]
                if timings is not None:
                    timings.append(dict(lambda_=l_r, alpha=a_r, init_s=t_draw, init_wait_s=t_wait, warm_from=src,
                                        optimize_s=_time.perf_counter() - t_1, library_ms=ds.profile()["wall_ms"]))
                rows[g] = (l_r, a_r, fitted["train_rmse"], fitted["test_rmse"])
                if csv and world == 1:
                    np.savetxt(csv, rows[: g + 1], delimiter=",")
        reg_tuning = _grid_sum(rows, world)
        if csv and world > 1 and rank == 0:
            np.savetxt(csv, reg_tuning, delimiter=",")
    return dict(rank_tuning=rank_tuning, latent_rank=latent_rank, reg_tuning=reg_tuning)


def fit(obj, latent_dimension=None, lambda_=None, alpha=None, partition=0, rng=None):
    """fit() of R/insider.R:190-216."""
    prm = obj["params"]
    rng = rng if rng is not None else np.random.default_rng(obj.get("seed", DEFAULT_SEED))
    K = int(latent_dimension)
    cfd, col = _fresh_inits(obj, K, rng)
    ds = _resident(obj, "fit")
    fitted = ds.optimize(cfd, col, K, float(lambda_), float(lambda_), float(alpha), int(partition), prm["global_tol"],
                         prm["sub_tol"], prm["max_iter"], seed=obj.get("seed", DEFAULT_SEED),
                         inc_continuous=obj["inc_continuous"])
    obj["cfd_matrices"] = fitted["row_matrices"]                                                         # :211-213
    obj["column_factor"] = fitted["column_factor"]
    obj["test_rmse"] = fitted["test_rmse"]
    obj["train_rmse"] = fitted["train_rmse"]
    obj["loss"] = fitted["loss"]
    obj["traj"] = fitted["traj"]
    return obj
