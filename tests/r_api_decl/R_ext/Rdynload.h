/* TEST INFRASTRUCTURE ONLY — see ../Rinternals.h. */
#ifndef S4B_TEST_R_DYNLOAD_H
#define S4B_TEST_R_DYNLOAD_H
#include "../Rinternals.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef void* (*DL_FUNC)(void);
typedef struct { const char* name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo*, const void*, const R_CallMethodDef*, const void*, const void*);
Rboolean R_useDynamicSymbols(DllInfo*, Rboolean);
#ifdef __cplusplus
}
#endif
#endif
