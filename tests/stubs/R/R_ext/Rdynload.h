/* Stand-in for R's <R_ext/Rdynload.h>: routine registration as r/insider_hip_shim.c uses it. */
#ifndef INSIDER_STUB_RDYNLOAD_H
#define INSIDER_STUB_RDYNLOAD_H
#include <Rinternals.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef void *(*DL_FUNC)();   /* unprototyped, as in R: every routine is cast to it */
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo *info, const void *cRoutines, const R_CallMethodDef *callRoutines, const void *fortranRoutines,
                       const void *externalRoutines);
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value);
#ifdef __cplusplus
}
#endif
#endif
