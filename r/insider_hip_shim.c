/*
 * insider_hip_shim.c — R-side binding of libinsider_hip.so for the kai0511/insider package.
 *
 * Compiled INTO the R package next to its own sources (copy this file and include/insider_hip.h to src/, add
 * `PKG_LIBS += -L$(INSIDER_HIP_DIR) -linsider_hip -Wl,-rpath,$(INSIDER_HIP_DIR)` to src/Makevars) or as a stand-alone
 * `R CMD SHLIB insider_hip_shim.c -linsider_hip` object loaded with dyn.load().  Plain R C API: needs neither Rcpp nor
 * Armadillo.  It replaces, argument for argument, the two hot-path .Call entries of the reference
 *     _insider_optimize                    (16 args)  src/RcppExports.cpp:87-110, R/RcppExports.R:20-22
 *     _insider_strong_coordinate_descent   ( 8 args)  src/RcppExports.cpp:35-50,  R/RcppExports.R:8-10
 * with an optional trailing `seed` (the reference draws its sweep order from R's global RNG through Rcpp::RNGScope,
 * src/RcppExports.cpp:38,90) and `device`.  R wrappers: r/insider_hip.R.
 *
 * NOT BUILT OR RUN IN THIS REPOSITORY'S PIPELINE: the image has no R (no R.h / Rinternals.h).  The C ABI it calls is
 * the tested surface (tests/test_gpu_boundary.py drives the same symbols with the same argument order via ctypes).
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stdint.h>
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

SEXP insider_hip_available_R(void) { return Rf_ScalarLogical(insider_hip_device_count() > 0); }

/* optimize(data, cfd_factors, column_factor, cfd_indicators, ctns_confounder, train_indicator, test_indicator,
 *          inc_continuous, latent_dim, lambda1, lambda2, alpha, tuning, global_tol, sub_tol, max_iter [, seed, device])
 * Returns list(row_matrices, column_factor, train_rmse, test_rmse, loss) — the fields of src/optimize.cpp:413-421 —
 * and, like the reference (:283-284 and the mat& parameter), has updated cfd_factors / column_factor IN PLACE. */
SEXP insider_hip_optimize_R(SEXP data, SEXP cfd_factors, SEXP column_factor, SEXP cfd_indicators, SEXP ctns_confounder,
                            SEXP train_indicator, SEXP test_indicator, SEXP inc_continuous, SEXP latent_dim,
                            SEXP lambda1, SEXP lambda2, SEXP alpha, SEXP tuning, SEXP global_tol, SEXP sub_tol,
                            SEXP max_iter, SEXP seed, SEXP device)
{
    if (TYPEOF(data) != REALSXP || TYPEOF(column_factor) != REALSXP) Rf_error("insider_hip: data / column_factor must be numeric matrices");
    const int64_t n = Rf_nrows(data), p = Rf_ncols(data);
    const int c = Rf_ncols(cfd_indicators), K = Rf_asInteger(latent_dim), inc = Rf_asInteger(inc_continuous);
    const int nfac = Rf_length(cfd_factors);
    if (nfac != c + (inc == 1 ? 1 : 0)) Rf_error("insider_hip: cfd_factors must hold one matrix per covariate (+ the continuous one)");
    int32_t *n_levels = (int32_t *)R_alloc((size_t)c, sizeof(int32_t));
    double **A = (double **)R_alloc((size_t)nfac, sizeof(double *));
    for (int i = 0; i < nfac; i++) {
        SEXP a = VECTOR_ELT(cfd_factors, i);
        if (TYPEOF(a) != REALSXP || Rf_ncols(a) != K) Rf_error("insider_hip: cfd_factors[[%d]] must be a numeric L x K matrix", i + 1);
        A[i] = REAL(a);                                         /* in place, src/optimize.cpp:283-284 */
        if (i < c) n_levels[i] = Rf_nrows(a);
    }
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
    if (rc != INSIDER_OK) Rf_error("insider_hip (status %d): %s", rc, insider_hip_last_error());   /* an R error, never exit(1) */
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

/* strong_coordinate_descent(X, y, wstart, lambda, alpha, XtX, Xty, tol [, seed, device]) -> numeric K-vector */
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
                                            0 /* hashed random order */, 10000, Rf_asInteger(device), REAL(beta), NULL);
    if (rc != INSIDER_OK) { UNPROTECT(1); Rf_error("insider_hip (status %d): %s", rc, insider_hip_last_error()); }
    UNPROTECT(1);
    return beta;
}

static const R_CallMethodDef insider_hip_calls[] = {
    {"insider_hip_available_R", (DL_FUNC)&insider_hip_available_R, 0},
    {"insider_hip_optimize_R", (DL_FUNC)&insider_hip_optimize_R, 18},
    {"insider_hip_strong_cd_R", (DL_FUNC)&insider_hip_strong_cd_R, 10},
    {NULL, NULL, 0}};

/* stand-alone build (`R CMD SHLIB -o insiderhip.so ...`); inside the insider package append the three entries to the
 * CallEntries table of src/RcppExports.cpp:112-120 instead */
void R_init_insiderhip(DllInfo *dll)
{
    R_registerRoutines(dll, NULL, insider_hip_calls, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
