/*
 * mock_r.c — a small stand-in for the R runtime behind tests/stubs/R/Rinternals.h, so that r/insider_hip_shim.c can be
 * compiled and EXECUTED without R (tests/test_r_shim.py).  Vectors are malloc'ed and never collected (the tests are
 * short-lived), Rf_error longjmps to the frame mock_call() set up and leaves the message in mock_last_error(), external
 * pointer finalizers run when the test calls mock_run_finalizers().  Test scaffolding only.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <math.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <string.h>

struct SEXPREC {
    int type;
    R_xlen_t len;
    void *data;          /* payload: double / int / SEXP array, C string, external address */
    int nrow, ncol;      /* dim attribute (0 = none) */
    SEXP names, tag, prot;
    R_CFinalizer_t fin;
    int preserved;
};

static struct SEXPREC nil_rec = {NILSXP, 0, NULL, 0, 0, NULL, NULL, NULL, NULL, 0};
static struct SEXPREC names_rec = {SYMSXP, 0, (void *)"names", 0, 0, NULL, NULL, NULL, NULL, 0};
static struct SEXPREC dim_rec = {SYMSXP, 0, (void *)"dim", 0, 0, NULL, NULL, NULL, NULL, 0};
SEXP R_NilValue = &nil_rec, R_NamesSymbol = &names_rec, R_DimSymbol = &dim_rec;
int R_NaInt = INT32_MIN;
double R_NaReal;

static jmp_buf *g_jmp = NULL;
static char g_error[1024], g_warning[1024];
static int g_warnings = 0, g_preserved = 0, g_finalized = 0;
static const R_CallMethodDef *g_routines = NULL;
static SEXP g_extptrs[256];
static int g_nextptr = 0;
static void *g_ralloc[4096];
static int g_nralloc = 0;

__attribute__((constructor)) static void mock_init(void) { R_NaReal = NAN; }

int R_IsNaN_or_NA(double x) { return isnan(x); }
int TYPEOF(SEXP x) { return x->type; }
double *REAL(SEXP x) { if (x->type != REALSXP) Rf_error("REAL() of a non-numeric object"); return (double *)x->data; }
int *INTEGER(SEXP x) { if (x->type != INTSXP && x->type != LGLSXP) Rf_error("INTEGER() of a non-integer object"); return (int *)x->data; }
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) { if (x->type != VECSXP || i < 0 || i >= x->len) Rf_error("VECTOR_ELT out of range"); return ((SEXP *)x->data)[i]; }
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) { if (x->type != VECSXP || i < 0 || i >= x->len) Rf_error("SET_VECTOR_ELT out of range"); ((SEXP *)x->data)[i] = v; return v; }
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v) { if (x->type != STRSXP || i < 0 || i >= x->len) Rf_error("SET_STRING_ELT out of range"); ((SEXP *)x->data)[i] = v; }

SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n)
{
    SEXP s = (SEXP)calloc(1, sizeof(struct SEXPREC));
    size_t el = type == REALSXP ? sizeof(double) : (type == INTSXP || type == LGLSXP) ? sizeof(int) : sizeof(SEXP);
    s->type = (int)type;
    s->len = n;
    s->data = calloc((size_t)(n > 0 ? n : 1), el);
    if (type == VECSXP || type == STRSXP) for (R_xlen_t i = 0; i < n; i++) ((SEXP *)s->data)[i] = R_NilValue;
    return s;
}

SEXP Rf_duplicate(SEXP x)
{
    if (x == R_NilValue) return x;
    SEXP s = Rf_allocVector((SEXPTYPE)x->type, x->len);
    size_t el = x->type == REALSXP ? sizeof(double) : (x->type == INTSXP || x->type == LGLSXP) ? sizeof(int) : sizeof(SEXP);
    memcpy(s->data, x->data, (size_t)x->len * el);
    s->nrow = x->nrow; s->ncol = x->ncol; s->names = x->names;
    return s;
}

SEXP Rf_mkChar(const char *str) { SEXP s = (SEXP)calloc(1, sizeof(struct SEXPREC)); s->type = CHARSXP; s->data = strdup(str); s->len = (R_xlen_t)strlen(str); return s; }
SEXP Rf_install(const char *name) { SEXP s = Rf_mkChar(name); s->type = SYMSXP; return s; }
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP value)
{
    if (name == R_NamesSymbol) x->names = value;
    else if (name == R_DimSymbol) { x->nrow = INTEGER(value)[0]; x->ncol = INTEGER(value)[1]; }
    return value;
}
SEXP Rf_ScalarReal(double v) { SEXP s = Rf_allocVector(REALSXP, 1); REAL(s)[0] = v; return s; }
SEXP Rf_ScalarInteger(int v) { SEXP s = Rf_allocVector(INTSXP, 1); INTEGER(s)[0] = v; return s; }
SEXP Rf_ScalarLogical(int v) { SEXP s = Rf_allocVector(LGLSXP, 1); INTEGER(s)[0] = v != 0; return s; }
int Rf_asInteger(SEXP x)
{
    if (x->len < 1) return R_NaInt;
    if (x->type == INTSXP || x->type == LGLSXP) return INTEGER(x)[0];
    if (x->type == REALSXP) return isnan(REAL(x)[0]) ? R_NaInt : (int)REAL(x)[0];
    return R_NaInt;
}
double Rf_asReal(SEXP x)
{
    if (x->len < 1) return R_NaReal;
    if (x->type == REALSXP) return REAL(x)[0];
    if (x->type == INTSXP || x->type == LGLSXP) return INTEGER(x)[0] == R_NaInt ? R_NaReal : (double)INTEGER(x)[0];
    return R_NaReal;
}
int Rf_nrows(SEXP x) { return x->nrow ? x->nrow : (int)x->len; }
int Rf_ncols(SEXP x) { return x->nrow ? x->ncol : 1; }
R_len_t Rf_length(SEXP x) { return (R_len_t)x->len; }
R_xlen_t Rf_xlength(SEXP x) { return x->len; }
Rboolean Rf_isNull(SEXP x) { return x == R_NilValue || x->type == NILSXP ? TRUE : FALSE; }
SEXP Rf_protect(SEXP x) { return x; }
void Rf_unprotect(int n) { (void)n; }
void R_PreserveObject(SEXP x) { x->preserved++; g_preserved++; }
void R_ReleaseObject(SEXP x) { x->preserved--; g_preserved--; }

SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot)
{
    SEXP s = (SEXP)calloc(1, sizeof(struct SEXPREC));
    s->type = EXTPTRSXP; s->data = p; s->tag = tag; s->prot = prot;
    if (g_nextptr < 256) g_extptrs[g_nextptr++] = s;
    return s;
}
void *R_ExternalPtrAddr(SEXP s) { return s->type == EXTPTRSXP ? s->data : NULL; }
void R_ClearExternalPtr(SEXP s) { s->data = NULL; }
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit) { (void)onexit; s->fin = fun; }

char *R_alloc(size_t n, int size)
{
    void *q = calloc(n ? n : 1, (size_t)size);
    if (g_nralloc < 4096) g_ralloc[g_nralloc++] = q;
    return (char *)q;
}

void Rf_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    if (g_jmp) longjmp(*g_jmp, 1);
    fprintf(stderr, "mock R: error outside mock_call(): %s\n", g_error);
    abort();
}
void Rf_warning(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_warning, sizeof g_warning, fmt, ap);
    va_end(ap);
    g_warnings++;
}

int R_registerRoutines(DllInfo *info, const void *c, const R_CallMethodDef *call, const void *f, const void *e)
{
    (void)info; (void)c; (void)f; (void)e;
    g_routines = call;
    return 1;
}
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value) { (void)info; return value; }

/* ---- what the Python test drives ----------------------------------------------------------------------------------- */
SEXP mock_nil(void) { return R_NilValue; }
SEXP mock_real_matrix(const double *src, int nrow, int ncol)   /* column-major copy, like an R numeric matrix */
{
    SEXP s = Rf_allocVector(REALSXP, (R_xlen_t)nrow * ncol);
    if (src) memcpy(s->data, src, (size_t)nrow * ncol * sizeof(double));
    s->nrow = nrow; s->ncol = ncol;
    return s;
}
SEXP mock_real_vector(const double *src, int n) { SEXP s = Rf_allocVector(REALSXP, n); if (src) memcpy(s->data, src, (size_t)n * sizeof(double)); return s; }
SEXP mock_int_matrix(const int *src, int nrow, int ncol)
{
    SEXP s = Rf_allocVector(INTSXP, (R_xlen_t)nrow * ncol);
    if (src) memcpy(s->data, src, (size_t)nrow * ncol * sizeof(int));
    s->nrow = nrow; s->ncol = ncol;
    return s;
}
SEXP mock_list(int n) { return Rf_allocVector(VECSXP, n); }
void mock_list_set(SEXP l, int i, SEXP v) { SET_VECTOR_ELT(l, i, v); }
SEXP mock_list_get(SEXP l, int i) { return VECTOR_ELT(l, i); }
SEXP mock_list_get_named(SEXP l, const char *name)
{
    if (l->type != VECSXP || !l->names) return R_NilValue;
    for (R_xlen_t i = 0; i < l->len; i++)
        if (strcmp((const char *)((SEXP *)l->names->data)[i]->data, name) == 0) return ((SEXP *)l->data)[i];
    return R_NilValue;
}
double *mock_real_ptr(SEXP s) { return (double *)s->data; }
int mock_type(SEXP s) { return s->type; }
int mock_len(SEXP s) { return (int)s->len; }
int mock_is_nil(SEXP s) { return s == R_NilValue; }
const char *mock_last_error(void) { return g_error; }
const char *mock_last_warning(void) { return g_warning; }
int mock_warning_count(void) { return g_warnings; }
int mock_preserved_count(void) { return g_preserved; }
int mock_finalized_count(void) { return g_finalized; }
int mock_routine_args(const char *name)
{
    for (const R_CallMethodDef *r = g_routines; r && r->name; r++) if (strcmp(r->name, name) == 0) return r->numArgs;
    return -1;
}
/* garbage collection stand-in: run the finalizer of every external pointer nothing preserves */
void mock_run_finalizers(void)
{
    for (int i = 0; i < g_nextptr; i++) {
        SEXP s = g_extptrs[i];
        if (s->fin && s->data && s->preserved == 0) { s->fin(s); g_finalized++; }
    }
}
void mock_free_transient(void) { for (int i = 0; i < g_nralloc; i++) free(g_ralloc[i]); g_nralloc = 0; }

/* .Call of a registered routine with nargs SEXP arguments; NULL (C) on an R error (message in mock_last_error()) */
typedef SEXP (*fn0)(void);
SEXP mock_call(const char *name, int nargs, SEXP *a)
{
    DL_FUNC f = NULL;
    for (const R_CallMethodDef *r = g_routines; r && r->name; r++)
        if (strcmp(r->name, name) == 0) {
            if (r->numArgs != nargs) { snprintf(g_error, sizeof g_error, "%s takes %d arguments, got %d", name, r->numArgs, nargs); return NULL; }
            f = r->fun;
        }
    if (!f) { snprintf(g_error, sizeof g_error, "no routine %s", name); return NULL; }
    jmp_buf jb;
    g_jmp = &jb;
    g_error[0] = 0;
    SEXP out = NULL;
    if (setjmp(jb) == 0) {
        typedef SEXP (*F)();
        F g = (F)f;
        switch (nargs) {
        case 0: out = ((fn0)f)(); break;
        case 1: out = g(a[0]); break;
        case 9: out = g(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]); break;
        case 10: out = g(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9]); break;
        case 14: out = g(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13]); break;
        case 19: out = g(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17], a[18]); break;
        default: snprintf(g_error, sizeof g_error, "mock_call: unsupported arity %d", nargs); out = NULL;
        }
    } else out = NULL;
    g_jmp = NULL;
    mock_free_transient();
    return out;
}
