/*
 * insider_hip_shim.c — R-side binding of libinsider_hip.so for the kai0511/insider package.
 *
 * Compiled INTO the R package next to its own sources (copy this file and include/insider_hip.h to src/, add
 * `PKG_LIBS += -L$(INSIDER_HIP_DIR) -linsider_hip -Wl,-rpath,$(INSIDER_HIP_DIR)` to src/Makevars) or as a stand-alone
 * `R CMD SHLIB insider_hip_shim.c -linsider_hip` object loaded with dyn.load().  Plain R C API: needs neither Rcpp nor
 * Armadillo.  It replaces, argument for argument, the hot-path .Call entries of the reference
 *     _insider_optimize                    (16 args)  src/RcppExports.cpp:87-110, R/RcppExports.R:20-22
 *     _insider_strong_coordinate_descent   ( 8 args)  src/RcppExports.cpp:35-50,  R/RcppExports.R:8-10
 *     _insider_optimize_continuous_v2      ( 8 args)  src/RcppExports.cpp:69-85,  R/RcppExports.R:16-18
 * with an optional trailing `seed` (the reference draws its sweep order from R's global RNG through Rcpp::RNGScope,
 * src/RcppExports.cpp:38,90) and `device`.  R wrappers: r/insider_hip.R.
 *
 * The data set stays RESIDENT in HBM across calls: the reference's tune() calls optimize() once per grid point with the
 * same `data`, indicator and mask objects (R/insider.R:142-174), so the binding keeps a small cache of library handles
 * keyed on the identity of those R objects (their data pointers and dimensions).  A cached handle holds a reference to
 * the objects it was built from (the external pointer's `prot` slot), so R can neither free them nor reuse their
 * addresses while the entry lives; modifying one of them in R makes R copy it (copy-on-modify), which changes the key.
 * An unmodified `tune()` loop therefore uploads X once (4.5 GB over PCIe + 0.5 s of list building at 10000 x 50000)
 * instead of once per grid point.  The handle is an EXTPTRSXP whose finalizer calls insider_hip_destroy().
 *
 * Status INSIDER_ERR_NO_DEVICE / INSIDER_ERR_UNSUPPORTED (no MI355X visible; K > 63, n or p >= 2^23) is not an R error:
 * the entry returns NULL and r/insider_hip.R falls back to the package's own `_insider_optimize`.  Every other failure is
 * an R error carrying insider_hip_last_error() (never exit(1), src/optimize.cpp:249-251).
 *
 * No R in this repository's image: the file is compiled (gcc -Wall -Werror) and EXECUTED against a small stand-in for
 * the R C API (tests/stubs/R: headers + mock_r.c) by tests/test_r_shim.py — the cache, the in-place factor update and
 * the fallback statuses on the CPU, a fit through the shim against the ctypes path on the GPU.  It has not been built
 * against a real R.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "insider_hip.h"

/* R hands the masks over as INTEGER matrices (R/insider.R:57-59; Rcpp would copy them to fp64,
 * src/RcppExports.cpp:96-97); logical and double matrices are accepted too. */
static uint8_t *mask_u8(SEXP m, size_t count)
{
    uint8_t *out = (uint8_t *)R_alloc(count, 1);
    if (TYPEOF(m) == INTSXP || TYPEOF(m) == LGLSXP) {
        const int *v = INTEGER(m);
        for (size_t e = 0; e < count; e++) out[e] = (uint8_t)(v[e] != 0 && v[e] != NA_INTEGER);
    } else if (TYPEOF(m) == REALSXP) {
        const double *v = REAL(m);
        for (size_t e = 0; e < count; e++) out[e] = (uint8_t)(v[e] != 0.0 && !ISNAN(v[e]));
    } else Rf_error("insider_hip: indicator matrices must be integer, logical or numeric");
    return out;
}

/* cfd_indicators: the reference takes `const umat&` (R double or integer, copied and cast; src/RcppExports.cpp:94) */
static int32_t *levels_i32(SEXP lev, size_t count)
{
    int32_t *out = (int32_t *)R_alloc(count, sizeof(int32_t));
    if (TYPEOF(lev) == INTSXP) memcpy(out, INTEGER(lev), count * sizeof(int32_t));
    else if (TYPEOF(lev) == REALSXP) { const double *v = REAL(lev); for (size_t e = 0; e < count; e++) out[e] = (int32_t)v[e]; }
    else Rf_error("insider_hip: cfd_indicators must be integer or numeric");
    return out;
}

/* address of an R vector's payload: the identity of the object as long as something references it */
static const void *payload(SEXP x)
{
    switch (TYPEOF(x)) {
    case REALSXP: return (const void *)REAL(x);
    case INTSXP: case LGLSXP: return (const void *)INTEGER(x);
    default: return NULL;
    }
}

SEXP insider_hip_available_R(void) { return Rf_ScalarLogical(insider_hip_device_count() > 0); }

/* ---- resident handles ------------------------------------------------------------------------------------------- */
static void handle_finalizer(SEXP ptr)
{
    insider_hip_handle *h = (insider_hip_handle *)R_ExternalPtrAddr(ptr);
    if (h) insider_hip_destroy(h);
    R_ClearExternalPtr(ptr);
}

static insider_hip_handle *handle_of(SEXP ptr)
{
    if (TYPEOF(ptr) != EXTPTRSXP) Rf_error("insider_hip: not a handle");
    insider_hip_handle *h = (insider_hip_handle *)R_ExternalPtrAddr(ptr);
    if (!h) Rf_error("insider_hip: the handle has been destroyed");
    return h;
}

/* the factors' row counts give L_i (R/insider.R:107 sizes them as length(unique(confounder[, i])) x K) */
static int factor_pointers(SEXP cfd_factors, int c, int inc, int K, double **A, int32_t *n_levels)
{
    const int nfac = Rf_length(cfd_factors);
    if (nfac != c + (inc == 1 ? 1 : 0)) Rf_error("insider_hip: cfd_factors must hold one matrix per covariate (+ the continuous one)");
    for (int i = 0; i < nfac; i++) {
        SEXP a = VECTOR_ELT(cfd_factors, i);
        if (TYPEOF(a) != REALSXP || Rf_ncols(a) != K) Rf_error("insider_hip: cfd_factors[[%d]] must be a numeric L x K matrix", i + 1);
        if (A) A[i] = REAL(a);                                  /* in place, src/optimize.cpp:283-284 */
        if (n_levels && i < c) n_levels[i] = Rf_nrows(a);
    }
    return nfac;
}

/* insider_hip_create_R(data, cfd_factors, cfd_indicators, ctns_confounder, train_indicator, test_indicator,
 *                      inc_continuous, latent_dim, device) -> external pointer (or NULL: no device / unsupported size).
 * cfd_factors is read for the level counts only. */
SEXP insider_hip_create_R(SEXP data, SEXP cfd_factors, SEXP cfd_indicators, SEXP ctns_confounder, SEXP train_indicator,
                          SEXP test_indicator, SEXP inc_continuous, SEXP latent_dim, SEXP device)
{
    if (TYPEOF(data) != REALSXP) Rf_error("insider_hip: data must be a numeric matrix");
    const int64_t n = Rf_nrows(data), p = Rf_ncols(data);
    const int c = Rf_ncols(cfd_indicators), inc = Rf_asInteger(inc_continuous), K = Rf_asInteger(latent_dim);
    if (Rf_nrows(cfd_indicators) != n) Rf_error("insider_hip: cfd_indicators must have one row per sample");
    if (inc != 0 && inc != 1) Rf_error("The value of prarameter inc_continuous can only be 0 or 1.");   /* src/optimize.cpp:270-272 */
    int32_t *n_levels = (int32_t *)R_alloc((size_t)(c > 0 ? c : 1), sizeof(int32_t));
    factor_pointers(cfd_factors, c, inc, K, NULL, n_levels);
    const size_t np = (size_t)n * (size_t)p;
    if ((size_t)Rf_xlength(train_indicator) != np || (size_t)Rf_xlength(test_indicator) != np) Rf_error("insider_hip: indicator shape must match data");
    const double *ctns = NULL;
    int m = 0;
    if (inc == 1) {   /* the reference ignores ctns_confounder otherwise (src/optimize.cpp:276-291) */
        if (TYPEOF(ctns_confounder) != REALSXP || Rf_nrows(ctns_confounder) != n) Rf_error("insider_hip: ctns_confounder must be numeric n x m");
        ctns = REAL(ctns_confounder);
        m = Rf_ncols(ctns_confounder);
    }
    insider_hip_handle *h = NULL;
    const int rc = insider_hip_create_ex(REAL(data), n, p, levels_i32(cfd_indicators, (size_t)n * c), c, n_levels, ctns, m,
                                         mask_u8(train_indicator, np), mask_u8(test_indicator, np), Rf_asInteger(device), &h);
    if (rc == INSIDER_ERR_NO_DEVICE || rc == INSIDER_ERR_UNSUPPORTED) {
        Rf_warning("insider_hip (status %d): %s; using the CPU reference", rc, insider_hip_last_error());
        return R_NilValue;
    }
    if (rc != INSIDER_OK) Rf_error("insider_hip (status %d): %s", rc, insider_hip_last_error());
    /* prot: the R objects the handle was built from stay referenced while it lives (see the cache below) */
    SEXP keep = PROTECT(Rf_allocVector(VECSXP, 5));
    SET_VECTOR_ELT(keep, 0, data);
    SET_VECTOR_ELT(keep, 1, cfd_indicators);
    SET_VECTOR_ELT(keep, 2, train_indicator);
    SET_VECTOR_ELT(keep, 3, test_indicator);
    SET_VECTOR_ELT(keep, 4, inc == 1 ? ctns_confounder : R_NilValue);
    SEXP ptr = PROTECT(R_MakeExternalPtr(h, Rf_install("insider_hip_handle"), keep));
    R_RegisterCFinalizerEx(ptr, handle_finalizer, TRUE);
    UNPROTECT(2);
    return ptr;
}

SEXP insider_hip_destroy_R(SEXP ptr)
{
    if (TYPEOF(ptr) == EXTPTRSXP) handle_finalizer(ptr);
    return R_NilValue;
}

static SEXP result_list(SEXP cfd_factors, SEXP column_factor, int nfac, double tr, double te, double loss)
{
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 5)), nm = PROTECT(Rf_allocVector(STRSXP, 5));
    SEXP rows = PROTECT(Rf_allocVector(VECSXP, nfac)), rnm = PROTECT(Rf_allocVector(STRSXP, nfac));
    for (int i = 0; i < nfac; i++) {                            /* List row_matrices{"factor0", ...}, :413-416 */
        char key[32];
        snprintf(key, sizeof key, "factor%d", i);
        SET_VECTOR_ELT(rows, i, Rf_duplicate(VECTOR_ELT(cfd_factors, i)));
        SET_STRING_ELT(rnm, i, Rf_mkChar(key));
    }
    Rf_setAttrib(rows, R_NamesSymbol, rnm);
    SET_VECTOR_ELT(out, 0, rows);                        SET_STRING_ELT(nm, 0, Rf_mkChar("row_matrices"));
    SET_VECTOR_ELT(out, 1, Rf_duplicate(column_factor)); SET_STRING_ELT(nm, 1, Rf_mkChar("column_factor"));
    SET_VECTOR_ELT(out, 2, Rf_ScalarReal(tr));           SET_STRING_ELT(nm, 2, Rf_mkChar("train_rmse"));
    SET_VECTOR_ELT(out, 3, Rf_ScalarReal(te));           SET_STRING_ELT(nm, 3, Rf_mkChar("test_rmse"));
    SET_VECTOR_ELT(out, 4, Rf_ScalarReal(loss));         SET_STRING_ELT(nm, 4, Rf_mkChar("loss"));
    Rf_setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(4);
    return out;
}

/* insider_hip_optimize_handle_R(handle, cfd_factors, column_factor, n_covariates, inc_continuous, latent_dim, lambda1,
 *                               lambda2, alpha, tuning, global_tol, sub_tol, max_iter, seed)
 * One optimize() on a resident data set: returns list(row_matrices, column_factor, train_rmse, test_rmse, loss) — the
 * fields of src/optimize.cpp:413-421 — and, like the reference (:283-284 and the mat& parameter), has updated
 * cfd_factors / column_factor IN PLACE.  NULL when the library cannot run this call (K > 63): the caller falls back. */
SEXP insider_hip_optimize_handle_R(SEXP handle, SEXP cfd_factors, SEXP column_factor, SEXP n_covariates, SEXP inc_continuous,
                                   SEXP latent_dim, SEXP lambda1, SEXP lambda2, SEXP alpha, SEXP tuning, SEXP global_tol,
                                   SEXP sub_tol, SEXP max_iter, SEXP seed)
{
    insider_hip_handle *h = handle_of(handle);
    if (TYPEOF(column_factor) != REALSXP) Rf_error("insider_hip: column_factor must be a numeric matrix");
    const int c = Rf_asInteger(n_covariates), K = Rf_asInteger(latent_dim), inc = Rf_asInteger(inc_continuous);
    if (Rf_nrows(column_factor) != K) Rf_error("insider_hip: column_factor must be K x p");
    double **A = (double **)R_alloc((size_t)(c + 1), sizeof(double *));
    const int nfac = factor_pointers(cfd_factors, c, inc, K, A, NULL);
    double tr = NA_REAL, te = NA_REAL, loss = NA_REAL;
    const int rc = insider_hip_optimize(h, A, REAL(column_factor), inc, K, Rf_asReal(lambda1), Rf_asReal(lambda2),
                                        Rf_asReal(alpha), Rf_asInteger(tuning), Rf_asReal(global_tol), Rf_asReal(sub_tol),
                                        (uint32_t)Rf_asReal(max_iter), (uint64_t)Rf_asReal(seed), &tr, &te, &loss, NULL, 0,
                                        NULL, NULL);
    if (rc == INSIDER_ERR_UNSUPPORTED || rc == INSIDER_ERR_NO_DEVICE) {
        Rf_warning("insider_hip (status %d): %s; using the CPU reference", rc, insider_hip_last_error());
        return R_NilValue;
    }
    if (rc != INSIDER_OK) Rf_error("insider_hip (status %d): %s", rc, insider_hip_last_error());   /* an R error, never exit(1) */
    return result_list(cfd_factors, column_factor, nfac, tr, te, loss);
}

/* ---- the cache: the reference's unmodified tune() / fit() reach the resident path through optimize()'s 16 arguments ---- */
#define CACHE_SLOTS 4
static struct {
    const void *data, *lev, *train, *test, *ctns;
    int64_t n, p;
    int c, inc, device;
    int *n_levels;            /* rows of each cfd_factors element when the handle was created (c entries, malloc'd) */
    SEXP ptr;                 /* preserved external pointer, or NULL */
    unsigned long stamp;
} g_cache[CACHE_SLOTS];
static unsigned long g_stamp = 0, g_hits = 0, g_misses = 0;

SEXP insider_hip_cache_clear_R(void)
{
    for (int s = 0; s < CACHE_SLOTS; s++)
        if (g_cache[s].ptr) {
            handle_finalizer(g_cache[s].ptr);
            R_ReleaseObject(g_cache[s].ptr);
            g_cache[s].ptr = NULL;
            free(g_cache[s].n_levels);
            g_cache[s].n_levels = NULL;
        }
    return R_NilValue;
}

/* c(hits, misses, live handles): lets a caller (and the tests) see that tune() re-used the upload */
SEXP insider_hip_cache_stats_R(void)
{
    SEXP out = PROTECT(Rf_allocVector(REALSXP, 3));
    int live = 0;
    for (int s = 0; s < CACHE_SLOTS; s++) live += g_cache[s].ptr != NULL;
    REAL(out)[0] = (double)g_hits;
    REAL(out)[1] = (double)g_misses;
    REAL(out)[2] = (double)live;
    UNPROTECT(1);
    return out;
}

/* the handle of this (data, indicators, masks) tuple: cached, or created and cached (evicting the least recently used) */
static SEXP cached_handle(SEXP data, SEXP cfd_factors, SEXP cfd_indicators, SEXP ctns_confounder, SEXP train_indicator,
                          SEXP test_indicator, SEXP inc_continuous, SEXP latent_dim, SEXP device)
{
    const int inc = Rf_asInteger(inc_continuous), dev = Rf_asInteger(device);
    const void *kd = payload(data), *kl = payload(cfd_indicators), *ktr = payload(train_indicator),
               *kte = payload(test_indicator), *kc = inc == 1 ? payload(ctns_confounder) : NULL;
    const int64_t n = Rf_nrows(data), p = Rf_ncols(data);
    const int c = Rf_ncols(cfd_indicators);
    int victim = 0;
    for (int s = 0; s < CACHE_SLOTS; s++) {
        if (g_cache[s].ptr && g_cache[s].data == kd && g_cache[s].lev == kl && g_cache[s].train == ktr && g_cache[s].test == kte &&
            g_cache[s].ctns == kc && g_cache[s].n == n && g_cache[s].p == p && g_cache[s].c == c && g_cache[s].inc == inc &&
            g_cache[s].device == dev && R_ExternalPtrAddr(g_cache[s].ptr)) {
            /* the key is the identity of the data objects: the factor shapes of THIS call must still be the ones the handle
             * was built for (its level counts come from them), or upload_factors would read past the caller's matrices */
            if (TYPEOF(cfd_factors) != VECSXP || Rf_length(cfd_factors) < c) Rf_error("insider_hip: cfd_factors must be a list of %d matrices", c);
            for (int i = 0; i < c; i++)
                if (Rf_nrows(VECTOR_ELT(cfd_factors, i)) != g_cache[s].n_levels[i])
                    Rf_error("insider_hip: cfd_factors[[%d]] has %d rows, the resident data set was created with %d levels for that "
                             "covariate (insider_hip_cache_clear() drops it)", i + 1, Rf_nrows(VECTOR_ELT(cfd_factors, i)),
                             g_cache[s].n_levels[i]);
            g_cache[s].stamp = ++g_stamp;
            ++g_hits;
            return g_cache[s].ptr;
        }
        if (!g_cache[s].ptr) victim = s;
        else if (g_cache[victim].ptr && g_cache[s].stamp < g_cache[victim].stamp) victim = s;
    }
    ++g_misses;
    if (g_cache[victim].ptr) {          /* free the evicted data set's HBM now, not at some later garbage collection */
        handle_finalizer(g_cache[victim].ptr);
        R_ReleaseObject(g_cache[victim].ptr);
        g_cache[victim].ptr = NULL;
        free(g_cache[victim].n_levels);
        g_cache[victim].n_levels = NULL;
    }
    SEXP ptr = insider_hip_create_R(data, cfd_factors, cfd_indicators, ctns_confounder, train_indicator, test_indicator,
                                    inc_continuous, latent_dim, device);
    if (ptr == R_NilValue) return ptr;
    R_PreserveObject(ptr);
    g_cache[victim].n_levels = (int *)malloc(sizeof(int) * (size_t)(c > 0 ? c : 1));
    for (int i = 0; i < c && g_cache[victim].n_levels; i++) g_cache[victim].n_levels[i] = Rf_nrows(VECTOR_ELT(cfd_factors, i));
    g_cache[victim].data = kd; g_cache[victim].lev = kl; g_cache[victim].train = ktr; g_cache[victim].test = kte;
    g_cache[victim].ctns = kc; g_cache[victim].n = n; g_cache[victim].p = p; g_cache[victim].c = c; g_cache[victim].inc = inc;
    g_cache[victim].device = dev; g_cache[victim].ptr = ptr; g_cache[victim].stamp = ++g_stamp;
    return ptr;
}

/* optimize(data, cfd_factors, column_factor, cfd_indicators, ctns_confounder, train_indicator, test_indicator,
 *          inc_continuous, latent_dim, lambda1, lambda2, alpha, tuning, global_tol, sub_tol, max_iter [, seed, device,
 *          resident])
 * The reference's 16 arguments.  resident != 0 (default in r/insider_hip.R): through the handle cache; 0: upload, fit,
 * free (insider_hip_optimize_oneshot_ex).  NULL = not run here, fall back to `_insider_optimize`. */
SEXP insider_hip_optimize_R(SEXP data, SEXP cfd_factors, SEXP column_factor, SEXP cfd_indicators, SEXP ctns_confounder,
                            SEXP train_indicator, SEXP test_indicator, SEXP inc_continuous, SEXP latent_dim,
                            SEXP lambda1, SEXP lambda2, SEXP alpha, SEXP tuning, SEXP global_tol, SEXP sub_tol,
                            SEXP max_iter, SEXP seed, SEXP device, SEXP resident)
{
    if (TYPEOF(data) != REALSXP || TYPEOF(column_factor) != REALSXP) Rf_error("insider_hip: data / column_factor must be numeric matrices");
    const int64_t n = Rf_nrows(data), p = Rf_ncols(data);
    const int c = Rf_ncols(cfd_indicators), K = Rf_asInteger(latent_dim), inc = Rf_asInteger(inc_continuous);
    if (Rf_asInteger(resident) != 0) {
        SEXP h = cached_handle(data, cfd_factors, cfd_indicators, ctns_confounder, train_indicator, test_indicator,
                               inc_continuous, latent_dim, device);
        if (h == R_NilValue) return h;
        SEXP nc = PROTECT(Rf_ScalarInteger(c));
        SEXP out = insider_hip_optimize_handle_R(h, cfd_factors, column_factor, nc, inc_continuous, latent_dim, lambda1, lambda2,
                                                 alpha, tuning, global_tol, sub_tol, max_iter, seed);
        UNPROTECT(1);
        return out;
    }
    int32_t *n_levels = (int32_t *)R_alloc((size_t)(c > 0 ? c : 1), sizeof(int32_t));
    double **A = (double **)R_alloc((size_t)(c + 1), sizeof(double *));
    const int nfac = factor_pointers(cfd_factors, c, inc, K, A, n_levels);
    const size_t np = (size_t)n * (size_t)p;
    const double *ctns = NULL;
    int m = 0;
    if (inc == 1) {
        if (TYPEOF(ctns_confounder) != REALSXP || Rf_nrows(ctns_confounder) != n) Rf_error("insider_hip: ctns_confounder must be numeric n x m");
        ctns = REAL(ctns_confounder);
        m = Rf_ncols(ctns_confounder);
    }
    double tr = NA_REAL, te = NA_REAL, loss = NA_REAL;
    const int rc = insider_hip_optimize_oneshot_ex(
        REAL(data), n, p, A, REAL(column_factor), levels_i32(cfd_indicators, (size_t)n * c), c, n_levels, ctns, m,
        mask_u8(train_indicator, np), mask_u8(test_indicator, np), inc, K, Rf_asReal(lambda1), Rf_asReal(lambda2),
        Rf_asReal(alpha), Rf_asInteger(tuning), Rf_asReal(global_tol), Rf_asReal(sub_tol), (uint32_t)Rf_asReal(max_iter),
        (uint64_t)Rf_asReal(seed), Rf_asInteger(device), &tr, &te, &loss);
    if (rc == INSIDER_ERR_UNSUPPORTED || rc == INSIDER_ERR_NO_DEVICE) {
        Rf_warning("insider_hip (status %d): %s; using the CPU reference", rc, insider_hip_last_error());
        return R_NilValue;
    }
    if (rc != INSIDER_OK) Rf_error("insider_hip (status %d): %s", rc, insider_hip_last_error());   /* an R error, never exit(1) */
    return result_list(cfd_factors, column_factor, nfac, tr, te, loss);
}

/* strong_coordinate_descent(X, y, wstart, lambda, alpha, XtX, Xty, tol [, seed, device]) -> numeric K-vector
 * (NULL: not run here, fall back to `_insider_strong_coordinate_descent`) */
SEXP insider_hip_strong_cd_R(SEXP X, SEXP y, SEXP wstart, SEXP lambda, SEXP alpha, SEXP XtX, SEXP Xty, SEXP tol,
                             SEXP seed, SEXP device)
{
    const int K = Rf_length(wstart);
    const int64_t m = Rf_isNull(X) ? 0 : Rf_nrows(X);
    if (!Rf_isNull(X) && (Rf_ncols(X) != K || Rf_length(y) != m)) Rf_error("insider_hip: X must be m x K and y of length m");
    if (!Rf_isNull(XtX) && (Rf_nrows(XtX) != K || Rf_ncols(XtX) != K || Rf_length(Xty) != K)) Rf_error("insider_hip: XtX must be K x K, Xty of length K");
    SEXP beta = PROTECT(Rf_allocVector(REALSXP, K));
    const int rc = insider_hip_strong_cd_xy(Rf_isNull(X) ? NULL : REAL(X), Rf_isNull(y) ? NULL : REAL(y), m, K, REAL(wstart),
                                            Rf_asReal(lambda), Rf_asReal(alpha), Rf_isNull(XtX) ? NULL : REAL(XtX),
                                            Rf_isNull(Xty) ? NULL : REAL(Xty), Rf_asReal(tol), (uint64_t)Rf_asReal(seed), 0u,
                                            0 /* hashed random order */, 1 << 24 /* no effective sweep cap, like the reference */, Rf_asInteger(device), REAL(beta), NULL);
    UNPROTECT(1);
    if (rc == INSIDER_ERR_UNSUPPORTED || rc == INSIDER_ERR_NO_DEVICE) return R_NilValue;
    if (rc != INSIDER_OK) Rf_error("insider_hip (status %d): %s", rc, insider_hip_last_error());
    return beta;
}

/* optimize_continuous_v2(data, indicator, updating_factor, c_factor, updating_confd, gram, lambda, tuning [, device])
 * -> TRUE, with updating_factor updated IN PLACE (the reference's rowvec&, src/optimize.cpp:76-77; its .Call returns
 * R_NilValue, src/RcppExports.cpp:69-85); NULL: not run here, fall back to `_insider_optimize_continuous_v2`. */
SEXP insider_hip_optimize_continuous_v2_R(SEXP data, SEXP indicator, SEXP updating_factor, SEXP c_factor, SEXP updating_confd,
                                          SEXP gram, SEXP lambda, SEXP tuning, SEXP device)
{
    if (TYPEOF(data) != REALSXP || TYPEOF(c_factor) != REALSXP || TYPEOF(updating_factor) != REALSXP ||
        TYPEOF(updating_confd) != REALSXP)
        Rf_error("insider_hip: data, updating_factor, c_factor and updating_confd must be numeric");
    const int64_t n = Rf_nrows(data), p = Rf_ncols(data);
    const int K = Rf_nrows(c_factor), tun = Rf_asInteger(tuning);
    if (Rf_ncols(c_factor) != p || Rf_length(updating_factor) != K || Rf_length(updating_confd) != n)
        Rf_error("insider_hip: c_factor must be K x p, updating_factor of length K, updating_confd of length n");
    const size_t np = (size_t)n * (size_t)p;
    const uint8_t *ind = NULL;
    const double *g = NULL;
    if (tun == 1) {       /* the reference reads `indicator` only here (:79-126) and `gram` only when tuning = 0 (:127-131) */
        if ((size_t)Rf_xlength(indicator) != np) Rf_error("insider_hip: indicator shape must match data");
        ind = mask_u8(indicator, np);
    } else if (tun == 0) {
        if (TYPEOF(gram) != REALSXP || Rf_nrows(gram) != K || Rf_ncols(gram) != K) Rf_error("insider_hip: gram must be numeric K x K");
        g = REAL(gram);
    }
    const int rc = insider_hip_optimize_continuous_v2(REAL(data), n, p, ind, REAL(updating_factor), REAL(c_factor), K,
                                                      REAL(updating_confd), g, Rf_asReal(lambda), tun, Rf_asInteger(device));
    if (rc == INSIDER_ERR_UNSUPPORTED || rc == INSIDER_ERR_NO_DEVICE) {
        Rf_warning("insider_hip (status %d): %s; using the CPU reference", rc, insider_hip_last_error());
        return R_NilValue;
    }
    if (rc != INSIDER_OK) Rf_error("insider_hip (status %d): %s", rc, insider_hip_last_error());   /* an R error, never exit(1) (:133-136) */
    return Rf_ScalarLogical(1);
}

static const R_CallMethodDef insider_hip_calls[] = {
    {"insider_hip_available_R", (DL_FUNC)&insider_hip_available_R, 0},
    {"insider_hip_optimize_R", (DL_FUNC)&insider_hip_optimize_R, 19},
    {"insider_hip_strong_cd_R", (DL_FUNC)&insider_hip_strong_cd_R, 10},
    {"insider_hip_optimize_continuous_v2_R", (DL_FUNC)&insider_hip_optimize_continuous_v2_R, 9},
    {"insider_hip_create_R", (DL_FUNC)&insider_hip_create_R, 9},
    {"insider_hip_optimize_handle_R", (DL_FUNC)&insider_hip_optimize_handle_R, 14},
    {"insider_hip_destroy_R", (DL_FUNC)&insider_hip_destroy_R, 1},
    {"insider_hip_cache_clear_R", (DL_FUNC)&insider_hip_cache_clear_R, 0},
    {"insider_hip_cache_stats_R", (DL_FUNC)&insider_hip_cache_stats_R, 0},
    {NULL, NULL, 0}};

/* stand-alone build (`R CMD SHLIB -o insiderhip.so ...`); inside the insider package append the entries to the
 * CallEntries table of src/RcppExports.cpp:112-120 instead */
void R_init_insiderhip(DllInfo *dll)
{
    R_registerRoutines(dll, NULL, insider_hip_calls, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}

void R_unload_insiderhip(DllInfo *dll)
{
    (void)dll;
    insider_hip_cache_clear_R();   /* free the HBM of every cached data set */
}
