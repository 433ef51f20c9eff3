"""ctypes binding of oracle/insider_oracle.c (test infrastructure, see package docstring)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# INSIDER_ORACLE_LIB: another build of the same oracle (tests/test_gpu_period.py: INSIDER_PERM_PERIOD = 64 on both sides)
_SO = os.environ.get("INSIDER_ORACLE_LIB") or os.path.join(_HERE, "_build", "libinsider_oracle.so")
TRAJ_STRIDE = 10
TRAJ_COLS = ("iter", "train_rmse", "test_rmse", "sse_half", "row_reg_half", "col_reg_half", "l1_reg", "loss",
             "delta_loss", "decay")

_lib = None


def build(force=False):
    """Compile the oracle with gcc (a few seconds)."""
    if os.environ.get("INSIDER_ORACLE_LIB"):
        return _SO
    deps = [os.path.join(_HERE, "insider_oracle.c"), os.path.join(_HERE, "Makefile"),
            os.path.join(os.path.dirname(_HERE), "include", "insider_perm.h")]   # the sweep-order spec is shared with the HIP side
    if force or not os.path.exists(_SO) or any(os.path.getmtime(_SO) < os.path.getmtime(d) for d in deps):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "all"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.oracle_num_procs.restype = C.c_int
    return _lib


def _f64(a):
    return np.asfortranarray(a, dtype=np.float64)


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def num_procs():
    return int(lib().oracle_num_procs())


def set_col_chunk(chunk=100):
    """OpenMP chunk of the per-gene loops (reference: schedule(dynamic, 100), src/optimize.cpp:213,243).  A bounded gene
    sample needs chunk 1 to occupy every thread; results do not depend on it (genes are independent)."""
    lib().oracle_set_col_chunk(C.c_int(int(chunk)))


_sink = None


def set_sweep_sink(p=None):
    """Test diagnostic: per-gene sweep counts of the LAST column step of the following calls land in the returned int32
    array (None switches it off)."""
    global _sink
    if p is None:
        lib().oracle_set_sweep_sink(None)
        _sink = None
        return None
    _sink = np.zeros(int(p), dtype=np.int32)
    lib().oracle_set_sweep_sink(_sink.ctypes.data_as(C.c_void_p))
    return _sink


def set_cd_form(form=0):
    """0 = residual-form CD (the reference's formulation; the parity oracle), 1 = covariance-form sweeps (LABELLED
    CPU-optimised variant for bench.py's cpu_baseline, never used as the checker)."""
    lib().oracle_set_cd_form(C.c_int(int(form)))


def sweep_order(K, seed, it, sweep, order_mode=0):
    """Coordinate order of sweep `sweep` (include/insider_perm.h as the compiled oracle applies it)."""
    out = np.zeros(K, dtype=np.int32)
    rc = lib().oracle_sweep_order(C.c_int(K), C.c_uint64(seed), C.c_uint32(it), C.c_uint32(sweep), C.c_int(order_mode),
                                  _p(out, C.c_int32))
    if rc:
        raise RuntimeError("oracle_sweep_order failed")
    return out.tolist()


def strong_cd_cov(wstart, lam, alpha, XtX, Xty, tol=1e-5, seed=0, it=0, order_mode=0, max_sweeps=1 << 24):
    """Covariance-form variant of strong_cd (see set_cd_form). Returns (beta, sweeps)."""
    XtX = _f64(XtX)
    Xty = _f64(Xty)
    w = _f64(wstart)
    K = Xty.shape[0]
    beta = np.zeros(K)
    sw = C.c_int(0)
    rc = lib().oracle_strong_cd_cov(C.c_int(K), _p(w), C.c_double(lam), C.c_double(alpha), _p(XtX), _p(Xty),
                                    C.c_double(tol), C.c_uint64(seed), C.c_uint32(it), C.c_int(order_mode),
                                    C.c_int(max_sweeps), _p(beta), C.byref(sw))
    if rc:
        raise RuntimeError(f"oracle_strong_cd_cov failed rc={rc}")
    return beta, sw.value


def solve_sympd(A, b):
    A = _f64(A)
    b = _f64(b)
    K = A.shape[0]
    nrhs = 1 if b.ndim == 1 else b.shape[1]
    x = np.zeros_like(b, order="F")
    rc = lib().oracle_solve_sympd(_p(A), _p(b), C.c_int(K), C.c_int(nrhs), _p(x))
    if rc:
        raise RuntimeError(f"oracle_solve_sympd failed rc={rc}")
    return x


def strong_cd(X, y, wstart, lam, alpha, XtX, Xty, tol=1e-5, seed=0, unit=0, it=0, order_mode=0, max_sweeps=1 << 24):
    """strong_coordinate_descent (reference src/coordinate_descent.cpp:56-127). Returns (beta, sweeps)."""
    X = _f64(X)
    y = _f64(y)
    m, K = X.shape
    XtX = _f64(XtX)
    Xty = _f64(Xty)
    w = _f64(wstart)
    beta = np.zeros(K)
    sw = C.c_int(0)
    rc = lib().oracle_strong_cd(_p(X), _p(y), C.c_int(m), C.c_int(K), _p(w), C.c_double(lam), C.c_double(alpha),
                                _p(XtX), _p(Xty), C.c_double(tol), C.c_uint64(seed), C.c_uint32(unit),
                                C.c_uint32(it), C.c_int(order_mode), C.c_int(max_sweeps), _p(beta), C.byref(sw))
    if rc:
        raise RuntimeError(f"oracle_strong_cd failed rc={rc}")
    return beta, sw.value


def masked_gram_col(xcol, mcol, R):
    """(XtX, Xty) of one gene as src/optimize.cpp:216-222 forms them."""
    R = _f64(R)
    n, K = R.shape
    xcol = _f64(xcol)
    mcol = np.ascontiguousarray(mcol, dtype=np.uint8)
    gram = _f64(R.T @ R)
    XtX = np.zeros((K, K), order="F")
    Xty = np.zeros(K)
    lib().oracle_masked_gram_col(_p(xcol), _p(mcol, C.c_uint8), _p(R), C.c_int(n), C.c_int(K), _p(gram), _p(XtX),
                                 _p(Xty))
    return XtX, Xty


def masked_gram_row(V, M, r, Cmat):
    """(XtX, Xty) of sample r as src/optimize.cpp:162-171 forms them (V = matrix being regressed)."""
    V = _f64(V)
    M = np.asfortranarray(M, dtype=np.uint8)
    Cmat = _f64(Cmat)
    n, p = V.shape
    K = Cmat.shape[0]
    gram = _f64(Cmat @ Cmat.T)
    XtX = np.zeros((K, K), order="F")
    Xty = np.zeros(K)
    vptr = C.cast(V.ctypes.data + 8 * r, C.POINTER(C.c_double))
    mptr = C.cast(M.ctypes.data + r, C.POINTER(C.c_uint8))
    lib().oracle_masked_gram_row(vptr, mptr, C.c_int(n), C.c_int(p), _p(Cmat), C.c_int(K), _p(gram), _p(XtX),
                                 _p(Xty))
    return XtX, Xty


def optimize_row(residual, M, A, Cmat, levels, gram, lam, tuning=1, n_threads=8):
    residual = _f64(residual)
    M = np.asfortranarray(M, dtype=np.uint8)
    A = _f64(A).copy(order="F")
    Cmat = _f64(Cmat)
    gram = _f64(gram)
    levels = np.ascontiguousarray(levels, dtype=np.int32)
    n, p = residual.shape
    L, K = A.shape
    rc = lib().oracle_optimize_row(_p(residual), _p(M, C.c_uint8), _p(A), _p(Cmat), _p(levels, C.c_int32), _p(gram),
                                   C.c_double(lam), C.c_int(tuning), C.c_int(n), C.c_int(p), C.c_int(K), C.c_int(L),
                                   C.c_int(n_threads))
    if rc:
        raise RuntimeError(f"oracle_optimize_row failed rc={rc}")
    return A


def optimize_col(X, M, R, Cmat, lam, alpha, tuning=1, tol=1e-5, seed=0, it=0, order_mode=0, max_sweeps=1 << 24,
                 n_threads=8, gene_offset=0):
    X = _f64(X)
    M = np.asfortranarray(M, dtype=np.uint8)
    R = _f64(R)
    Cout = _f64(Cmat).copy(order="F")
    n, p = X.shape
    K = R.shape[1]
    sw = C.c_int64(0)
    rc = lib().oracle_optimize_col(_p(X), _p(M, C.c_uint8), _p(R), _p(Cout), C.c_double(lam), C.c_double(alpha),
                                   C.c_int(tuning), C.c_double(tol), C.c_int(n), C.c_int(p), C.c_int(K),
                                   C.c_uint64(seed), C.c_uint32(it), C.c_int(order_mode), C.c_int(max_sweeps),
                                   C.c_int(n_threads), C.c_int64(gene_offset), C.byref(sw))
    if rc:
        raise RuntimeError(f"oracle_optimize_col failed rc={rc}")
    return Cout, sw.value


def optimize(X, levels, n_levels, A_list, Cmat, M_train, M_test, lam1, lam2, alpha, tuning=1, global_tol=1e-10,
             sub_tol=1e-5, max_iter=10000, seed=0, order_mode=0, max_sweeps=1 << 24, row_threads=10, col_threads=30,
             traj_cap=4096, ctns=None):
    """The reference's optimize() (src/optimize.cpp:255-422), categorical covariates only.

    Returns dict(row_matrices, column_factor, train_rmse, test_rmse, loss, traj, iters, total_sweeps);
    inputs are not modified.
    """
    X = _f64(X)
    n, p = X.shape
    levels = np.asfortranarray(levels, dtype=np.int32).reshape(n, -1, order="F")
    c = levels.shape[1]
    n_levels = np.ascontiguousarray(n_levels, dtype=np.int32)
    A = [_f64(a).copy(order="F") for a in A_list]
    Cout = _f64(Cmat).copy(order="F")
    K = Cout.shape[0]
    Mtr = np.asfortranarray(M_train, dtype=np.uint8)
    Mte = np.asfortranarray(M_test, dtype=np.uint8)
    Aptrs = (C.POINTER(C.c_double) * len(A))(*[_p(a) for a in A])
    if ctns is not None:
        ctns = _f64(np.asarray(ctns, dtype=np.float64).reshape(n, -1))
        m = ctns.shape[1]
        assert len(A) == c + 1 and A[c].shape == (m, K)
    else:
        m = 0
        assert len(A) == c
    traj = np.full((traj_cap, TRAJ_STRIDE), np.nan)
    tr = C.c_double()
    te = C.c_double()
    lo = C.c_double()
    rows = C.c_int()
    iters = C.c_int()
    sw = C.c_int64()
    phases = np.zeros(3)
    rc = lib().oracle_optimize(_p(X), C.c_int(n), C.c_int(p), _p(levels, C.c_int32), C.c_int(c),
                               _p(n_levels, C.c_int32), _p(ctns) if m else None, C.c_int(m), Aptrs, _p(Cout),
                               _p(Mtr, C.c_uint8), _p(Mte, C.c_uint8),
                               C.c_int(K), C.c_double(lam1), C.c_double(lam2), C.c_double(alpha), C.c_int(tuning),
                               C.c_double(global_tol), C.c_double(sub_tol), C.c_uint32(max_iter), C.c_uint64(seed),
                               C.c_int(order_mode), C.c_int(max_sweeps), C.c_int(row_threads), C.c_int(col_threads),
                               C.byref(tr), C.byref(te), C.byref(lo), _p(traj), C.c_int(traj_cap), C.byref(rows),
                               C.byref(iters), C.byref(sw), _p(phases))
    if rc:
        raise RuntimeError(f"oracle_optimize failed rc={rc}")
    return dict(row_matrices=A, column_factor=Cout, train_rmse=tr.value, test_rmse=te.value, loss=lo.value,
                traj=traj[:rows.value].copy(), iters=iters.value, total_sweeps=sw.value,
                phase_seconds=dict(row=phases[0], col=phases[1], residual_eval=phases[2]))


def optimize_continuous(data, M, u, Cmat, z, gram, lam, tuning=1, n_threads=8):
    """optimize_continuous_v2 (reference src/optimize.cpp:76-137); returns the updated K-vector."""
    data = _f64(data)
    M = np.asfortranarray(M, dtype=np.uint8)
    Cmat = _f64(Cmat)
    gram = _f64(gram)
    z = _f64(z)
    u = _f64(u).copy()
    n, p = data.shape
    K = Cmat.shape[0]
    rc = lib().oracle_optimize_continuous(_p(data), _p(M, C.c_uint8), _p(u), _p(Cmat), _p(z), _p(gram), C.c_double(lam),
                                          C.c_int(tuning), C.c_int(n), C.c_int(p), C.c_int(K), C.c_int(n_threads))
    if rc:
        raise RuntimeError(f"oracle_optimize_continuous failed rc={rc}")
    return u
