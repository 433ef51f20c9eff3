/*
 * Stand-in for R's <Rinternals.h>: ONLY the part of the R C API that r/insider_hip_shim.c uses, declared with R's
 * documented names and signatures ("Writing R Extensions", section 5/6), implemented for tests by mock_r.c in this
 * directory.  Written for this repository (the image has no R); it is test scaffolding, not a copy of R's header.
 */
#ifndef INSIDER_STUB_RINTERNALS_H
#define INSIDER_STUB_RINTERNALS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SEXPREC *SEXP;
typedef int R_len_t;
typedef ptrdiff_t R_xlen_t;
typedef unsigned int SEXPTYPE;
typedef enum { FALSE = 0, TRUE = 1 } Rboolean;

#define NILSXP 0
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define EXTPTRSXP 22
#define CHARSXP 9
#define SYMSXP 1

extern SEXP R_NilValue;
extern SEXP R_NamesSymbol;
extern SEXP R_DimSymbol;
extern int R_NaInt;
extern double R_NaReal;
#define NA_INTEGER R_NaInt
#define NA_REAL R_NaReal
int R_IsNaN_or_NA(double x);
#define ISNAN(x) R_IsNaN_or_NA(x)

int TYPEOF(SEXP x);
double *REAL(SEXP x);
int *INTEGER(SEXP x);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);

SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n);
SEXP Rf_duplicate(SEXP x);
SEXP Rf_mkChar(const char *s);
SEXP Rf_install(const char *name);
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP value);
SEXP Rf_ScalarReal(double v);
SEXP Rf_ScalarInteger(int v);
SEXP Rf_ScalarLogical(int v);
int Rf_asInteger(SEXP x);
double Rf_asReal(SEXP x);
int Rf_nrows(SEXP x);
int Rf_ncols(SEXP x);
R_len_t Rf_length(SEXP x);
R_xlen_t Rf_xlength(SEXP x);
Rboolean Rf_isNull(SEXP x);

SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
void R_PreserveObject(SEXP x);
void R_ReleaseObject(SEXP x);

typedef void (*R_CFinalizer_t)(SEXP);
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot);
void *R_ExternalPtrAddr(SEXP s);
void R_ClearExternalPtr(SEXP s);
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit);

char *R_alloc(size_t n, int size);
void Rf_error(const char *fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
void Rf_warning(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#ifdef __cplusplus
}
#endif
#endif
