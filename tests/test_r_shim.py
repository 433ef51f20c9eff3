"""r/insider_hip_shim.c — the R-side binding of the C ABI — compiled with -Wall -Werror and EXECUTED against a stand-in
for the R C API (tests/stubs/R: headers with R's documented signatures + mock_r.c), because the image has no R.

CPU: the file compiles, registers its routines with the arities r/insider_hip.R calls them with, turns bad arguments
into R errors, and answers "no device" with NULL + a warning (the R wrapper then falls back to `_insider_optimize`).
GPU: optimize() through the shim's 16-argument entry equals the ctypes path bit for bit, updates the factors in place
like the reference (src/optimize.cpp:283-284), and the reference's tune()-style call sequence (same data objects, new
inits per grid point, R/insider.R:142-174) re-uses ONE resident handle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBS = os.path.join(ROOT, "tests", "stubs", "R")
SHIM = os.path.join(ROOT, "r", "insider_hip_shim.c")
BUILD = os.path.join(ROOT, "tests", "_build")


@pytest.fixture(scope="module")
def shim():
    ge.build()
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libinsider_shim_mock.so")
    srcs = [SHIM, os.path.join(STUBS, "mock_r.c")]
    if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in srcs + [os.path.join(STUBS, "Rinternals.h")]):
        subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-Wall", "-Wno-strict-prototypes", "-fPIC", "-shared", "-o", so,
                               "-I" + STUBS, "-I" + os.path.join(ROOT, "include"), *srcs,
                               "-L" + os.path.join(ROOT, "insider_amd"), "-linsider_hip",
                               "-Wl,-rpath," + os.path.join(ROOT, "insider_amd"), "-lm"])
    lib = C.CDLL(so)
    vp = C.c_void_p
    for name, res, args in (("mock_nil", vp, []), ("mock_real_matrix", vp, [vp, C.c_int, C.c_int]),
                            ("mock_real_vector", vp, [vp, C.c_int]), ("mock_int_matrix", vp, [vp, C.c_int, C.c_int]),
                            ("mock_list", vp, [C.c_int]), ("mock_list_set", None, [vp, C.c_int, vp]),
                            ("mock_list_get", vp, [vp, C.c_int]), ("mock_list_get_named", vp, [vp, C.c_char_p]),
                            ("mock_real_ptr", C.POINTER(C.c_double), [vp]), ("mock_type", C.c_int, [vp]),
                            ("mock_len", C.c_int, [vp]), ("mock_is_nil", C.c_int, [vp]), ("mock_last_error", C.c_char_p, []),
                            ("mock_last_warning", C.c_char_p, []), ("mock_warning_count", C.c_int, []),
                            ("mock_preserved_count", C.c_int, []), ("mock_finalized_count", C.c_int, []),
                            ("mock_routine_args", C.c_int, [C.c_char_p]), ("mock_run_finalizers", None, []),
                            ("mock_call", vp, [C.c_char_p, C.c_int, C.POINTER(vp)]), ("R_init_insiderhip", None, [vp]),
                            ("R_unload_insiderhip", None, [vp])):
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    lib.R_init_insiderhip(None)
    return lib


class R:
    """Tiny helper around the mock: numpy -> SEXP and .Call."""

    def __init__(self, lib):
        self.lib = lib

    def real(self, a):
        a = np.asfortranarray(a, dtype=np.float64)
        if a.ndim == 2:
            return self.lib.mock_real_matrix(a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1])
        return self.lib.mock_real_vector(a.ctypes.data_as(C.c_void_p), a.size)

    def integer(self, a):
        a = np.asfortranarray(a, dtype=np.int32)
        a2 = a.reshape(a.shape[0], -1, order="F")
        return self.lib.mock_int_matrix(a2.ctypes.data_as(C.c_void_p), a2.shape[0], a2.shape[1])

    def scalar(self, v):
        return self.real(np.array([float(v)]))

    def list(self, items):
        l = self.lib.mock_list(len(items))
        for i, it in enumerate(items):
            self.lib.mock_list_set(l, i, it)
        return l

    def to_numpy(self, sexp, shape):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(self.lib.mock_real_ptr(sexp), shape=(n,)).reshape(shape, order="F").copy()

    def call(self, name, *args):
        arr = (C.c_void_p * max(len(args), 1))(*args)
        out = self.lib.mock_call(name.encode(), len(args), arr)
        if out is None:
            raise RuntimeError(self.lib.mock_last_error().decode())
        return out


def test_shim_compiles_warning_free_against_the_r_api_stand_in():
    subprocess.check_call(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + STUBS,
                           "-I" + os.path.join(ROOT, "include"), SHIM])


def test_registered_routines_match_the_r_wrappers(shim):
    arities = {"insider_hip_available_R": 0, "insider_hip_optimize_R": 19, "insider_hip_strong_cd_R": 10,
               "insider_hip_optimize_continuous_v2_R": 9,
               "insider_hip_create_R": 9, "insider_hip_optimize_handle_R": 14, "insider_hip_destroy_R": 1,
               "insider_hip_cache_clear_R": 0, "insider_hip_cache_stats_R": 0}
    for name, n in arities.items():
        assert shim.mock_routine_args(name.encode()) == n
    import re
    rsrc = open(os.path.join(ROOT, "r", "insider_hip.R")).read()
    for name, n in arities.items():   # every .Call in r/insider_hip.R passes exactly the registered number of arguments
        for m in re.finditer(r'\.Call\("%s"((?:[^()]|\([^()]*\))*)\)' % name, rsrc):
            args = [a for a in m.group(1).split(",") if a.strip()]
            assert len(args) == n, (name, args)
    # the wrapper's CPU fall-back is announced, not silent (K > 63, no device: the shim warns, the wrapper adds a message)
    assert 'message("insider_hip: optimize() runs on the CPU reference' in rsrc


def _problem(r, seed=3, n=48, p=80, levels=(6, 4), K=5):
    from insider_amd import workloads
    w = workloads.make(n=n, p=p, level_counts=levels, K=K, lam=2.0, alpha=0.4, f=0.15, data_seed=seed, mask_seed=seed + 1,
                       init_seed=seed + 2)
    sx = dict(data=r.real(w.X), lev=r.integer(w.levels), ctns=r.real(np.zeros((n, 1))), train=r.integer(w.M_train),
              test=r.integer(w.M_test))
    return w, sx


def _optimize_args(r, w, sx, A, Cm, lam, resident=1, K=None, seed=11):
    K = K or w.K
    return (sx["data"], r.list(A), Cm, sx["lev"], sx["ctns"], sx["train"], sx["test"], r.scalar(0), r.scalar(K), r.scalar(lam),
            r.scalar(lam), r.scalar(w.alpha), r.scalar(1), r.scalar(-1.0), r.scalar(1e-5), r.scalar(5), r.scalar(seed),
            r.scalar(0), r.scalar(resident))


def test_no_device_is_a_fallback_not_an_error(shim):
    """On a box without a GPU the binding must hand control back to the package's CPU path: NULL + a warning."""
    from insider_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    r = R(shim)
    w, sx = _problem(r)
    A = [r.real(a) for a in w.A0]
    Cm = r.real(w.C0)
    before = shim.mock_warning_count()
    for resident in (1, 0):
        out = r.call("insider_hip_optimize_R", *_optimize_args(r, w, sx, A, Cm, 2.0, resident=resident))
        assert shim.mock_is_nil(out)
    assert shim.mock_warning_count() == before + 2 and b"CPU reference" in shim.mock_last_warning()
    assert shim.mock_preserved_count() == 0          # nothing was cached
    avail = r.call("insider_hip_available_R")
    assert shim.mock_len(avail) == 1


def test_bad_arguments_become_r_errors(shim):
    r = R(shim)
    w, sx = _problem(r)
    A = [r.real(a) for a in w.A0]
    with pytest.raises(RuntimeError, match="numeric matrices"):
        r.call("insider_hip_optimize_R", *_optimize_args(r, w, dict(sx, data=sx["lev"]), A, r.real(w.C0), 2.0))
    with pytest.raises(RuntimeError, match="one matrix per covariate"):
        r.call("insider_hip_optimize_R", *_optimize_args(r, w, sx, A[:1], r.real(w.C0), 2.0))
    with pytest.raises(RuntimeError, match="not a handle"):
        r.call("insider_hip_optimize_handle_R", sx["data"], r.list(A), r.real(w.C0), r.scalar(2), r.scalar(0), r.scalar(w.K),
               r.scalar(1), r.scalar(1), r.scalar(0.4), r.scalar(1), r.scalar(-1), r.scalar(1e-5), r.scalar(3), r.scalar(1))
    with pytest.raises(RuntimeError, match="takes 19 arguments"):
        r.call("insider_hip_optimize_R", sx["data"])


@pytest.mark.gpu
def test_tune_style_calls_reuse_one_resident_handle_and_match_ctypes(shim):
    from insider_amd import api
    r = R(shim)
    w, sx = _problem(r)
    r.call("insider_hip_cache_clear_R")
    base = r.to_numpy(r.call("insider_hip_cache_stats_R"), (3,))
    ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
    rng = np.random.default_rng(5)
    for g, lam in enumerate((2.0, 3.0, 5.0)):          # tune(): same data objects, fresh inits per grid point
        A0 = [np.asfortranarray(rng.normal(0, 1e-3, a.shape)) for a in w.A0]
        C0 = np.asfortranarray(rng.normal(0, 1e-3, w.C0.shape))
        A = [r.real(a) for a in A0]
        Cm = r.real(C0)
        out = r.call("insider_hip_optimize_R", *_optimize_args(r, w, sx, A, Cm, lam))
        assert not shim.mock_is_nil(out)
        ref = ds.optimize([a.copy(order="F") for a in A0], C0.copy(order="F"), w.K, lam, lam, w.alpha, tuning=1, max_iter=5,
                          global_tol=-1.0, seed=11)
        Cgot = r.to_numpy(shim.mock_list_get_named(out, b"column_factor"), C0.shape)
        assert np.array_equal(Cgot, ref["column_factor"])
        assert np.array_equal(r.to_numpy(Cm, C0.shape), ref["column_factor"])          # in place, like the reference
        rows = shim.mock_list_get_named(out, b"row_matrices")
        for i, a in enumerate(A0):
            assert np.array_equal(r.to_numpy(shim.mock_list_get(rows, i), a.shape), ref["row_matrices"][f"factor{i}"])
            assert np.array_equal(r.to_numpy(A[i], a.shape), ref["row_matrices"][f"factor{i}"])
        for key in ("train_rmse", "test_rmse", "loss"):
            assert r.to_numpy(shim.mock_list_get_named(out, key.encode()), (1,))[0] == ref[key]
    ds.close()
    st = r.to_numpy(r.call("insider_hip_cache_stats_R"), (3,)) - base
    assert st[0] == 2 and st[1] == 1 and st[2] == 1                  # one upload, two re-uses, one live handle
    # a cache hit with factor matrices of ANOTHER row count (the handle's level counts came from the first call's): an R error,
    # not an out-of-bounds read of the caller's matrices
    bad = [r.real(np.zeros((a.shape[0] + (1 if i == 0 else 0), w.K))) for i, a in enumerate(w.A0)]
    with pytest.raises(RuntimeError, match="was created with"):
        r.call("insider_hip_optimize_R", *_optimize_args(r, w, sx, bad, r.real(w.C0), 2.0))
    # another data object (what R's copy-on-modify produces when the user changes the matrix): a second handle
    sx2 = dict(sx, data=r.real(w.X * 1.0))
    out = r.call("insider_hip_optimize_R", *_optimize_args(r, w, sx2, [r.real(a) for a in w.A0], r.real(w.C0), 2.0))
    assert not shim.mock_is_nil(out)
    st = r.to_numpy(r.call("insider_hip_cache_stats_R"), (3,)) - base
    assert st[1] == 2 and st[2] == 2
    # the one-shot form (resident = FALSE) gives the same numbers and caches nothing
    A = [r.real(a) for a in w.A0]
    Cm = r.real(w.C0)
    one = r.call("insider_hip_optimize_R", *_optimize_args(r, w, sx, A, Cm, 2.0, resident=0))
    A2 = [r.real(a) for a in w.A0]
    Cm2 = r.real(w.C0)
    res = r.call("insider_hip_optimize_R", *_optimize_args(r, w, sx, A2, Cm2, 2.0, resident=1))
    assert np.array_equal(r.to_numpy(Cm, w.C0.shape), r.to_numpy(Cm2, w.C0.shape))
    assert r.to_numpy(shim.mock_list_get_named(one, b"loss"), (1,))[0] == r.to_numpy(shim.mock_list_get_named(res, b"loss"), (1,))[0]
    # K beyond the library's limit: NULL + warning -> the R wrapper falls back to the CPU reference
    K = 64
    Abig = [r.real(np.zeros((a.shape[0], K))) for a in w.A0]
    out = r.call("insider_hip_optimize_R", *_optimize_args(r, w, sx, Abig, r.real(np.zeros((K, w.C0.shape[1]))), 2.0, K=K))
    # the fallback is VISIBLE: the warning names the status, the reason (the library's K range) and what happens next
    wtxt = shim.mock_last_warning()
    assert shim.mock_is_nil(out) and b"CPU reference" in wtxt and b"K must be in 1..63" in wtxt and b"status 6" in wtxt, wtxt
    r.call("insider_hip_cache_clear_R")
    assert shim.mock_preserved_count() == 0
    assert r.to_numpy(r.call("insider_hip_cache_stats_R"), (3,))[2] == 0


@pytest.mark.gpu
def test_explicit_handle_lifecycle(shim):
    r = R(shim)
    w, sx = _problem(r, seed=9)
    A = [r.real(a) for a in w.A0]
    Cm = r.real(w.C0)
    h = r.call("insider_hip_create_R", sx["data"], r.list(A), sx["lev"], sx["ctns"], sx["train"], sx["test"], r.scalar(0),
               r.scalar(w.K), r.scalar(0))
    assert shim.mock_type(h) == 22
    args = (r.list(A), Cm, r.scalar(2), r.scalar(0), r.scalar(w.K), r.scalar(2.0), r.scalar(2.0), r.scalar(0.4), r.scalar(1),
            r.scalar(-1.0), r.scalar(1e-5), r.scalar(3), r.scalar(7))
    out = r.call("insider_hip_optimize_handle_R", h, *args)
    assert np.isfinite(r.to_numpy(shim.mock_list_get_named(out, b"loss"), (1,))[0])
    r.call("insider_hip_destroy_R", h)
    with pytest.raises(RuntimeError, match="destroyed"):
        r.call("insider_hip_optimize_handle_R", h, *args)
    # a handle nobody destroys is freed by its finalizer (R's garbage collector; the mock runs them on request)
    h2 = r.call("insider_hip_create_R", sx["data"], r.list(A), sx["lev"], sx["ctns"], sx["train"], sx["test"], r.scalar(0),
                r.scalar(w.K), r.scalar(0))
    n0 = shim.mock_finalized_count()
    shim.mock_run_finalizers()
    assert shim.mock_finalized_count() == n0 + 1
    with pytest.raises(RuntimeError, match="destroyed"):
        r.call("insider_hip_optimize_handle_R", h2, *args)


@pytest.mark.gpu
def test_optimize_continuous_v2_through_the_shim_updates_in_place(shim):
    """The reference's eight arguments (R/RcppExports.R:16-18): the K-vector is updated in place (rowvec&), the routine's
    result equals the ctypes path bit for bit, for both `tuning` values."""
    from insider_amd import api
    r = R(shim)
    rng = np.random.default_rng(21)
    n, p, K = 60, 90, 7
    D = np.asfortranarray(rng.standard_normal((n, p)))
    M = np.asfortranarray(rng.random((n, p)) > 0.15, dtype=np.int32)
    Cm = np.asfortranarray(rng.standard_normal((K, p)) * 0.3)
    z = rng.standard_normal(n)
    u0 = rng.standard_normal(K) * 0.1
    gram = np.asfortranarray(Cm @ Cm.T)
    for tuning in (1, 0):
        u = r.real(u0.copy())
        out = r.call("insider_hip_optimize_continuous_v2_R", r.real(D), r.integer(M), u, r.real(Cm), r.real(z), r.real(gram),
                     r.scalar(1.5), r.scalar(tuning), r.scalar(0))
        assert not shim.mock_is_nil(out)
        ref = api.optimize_continuous_v2(D, M, u0.copy(), Cm, z, gram, 1.5, tuning)
        assert np.array_equal(r.to_numpy(u, (K,)), ref) and not np.array_equal(ref, u0)
    with pytest.raises(RuntimeError, match="tuning should be either 0 or 1"):
        r.call("insider_hip_optimize_continuous_v2_R", r.real(D), r.integer(M), r.real(u0.copy()), r.real(Cm), r.real(z), r.real(gram),
               r.scalar(1.5), r.scalar(2), r.scalar(0))
