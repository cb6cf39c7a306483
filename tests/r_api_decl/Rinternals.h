/* TEST INFRASTRUCTURE ONLY — declarations of the subset of R's C API that shim/init_shim.cpp uses, so that the shim can be
 * syntax- and type-checked (`g++ -fsyntax-only -Itests/r_api_decl`) on a machine without R.  Declarations only: nothing here
 * can be linked or run; a real build includes R's own headers from `R CMD config --cppflags` instead. */
#ifndef S4B_TEST_RINTERNALS_H
#define S4B_TEST_RINTERNALS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct SEXPREC* SEXP;
typedef ptrdiff_t R_xlen_t;
typedef enum { FALSE = 0, TRUE } Rboolean;
typedef unsigned int SEXPTYPE;
#define NILSXP 0
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define RAWSXP 24
typedef unsigned char Rbyte;
extern SEXP R_NilValue, R_GlobalEnv, R_NamesSymbol, R_DimSymbol, R_DimNamesSymbol, R_ClassSymbol, R_RowNamesSymbol, R_SeedsSymbol, R_UnboundValue;
extern int R_NaInt;
#define NA_INTEGER R_NaInt
#define NA_LOGICAL R_NaInt
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
SEXP Rf_allocVector(SEXPTYPE, R_xlen_t);
SEXP Rf_allocMatrix(SEXPTYPE, int, int);
SEXP Rf_install(const char*);
SEXP Rf_mkChar(const char*);
SEXP Rf_mkString(const char*);
SEXP Rf_getAttrib(SEXP, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
SEXP Rf_findVar(SEXP, SEXP);
void Rf_defineVar(SEXP, SEXP, SEXP);
SEXP Rf_lang4(SEXP, SEXP, SEXP, SEXP);
SEXP Rf_eval(SEXP, SEXP);
SEXP Rf_ScalarReal(double);
SEXP Rf_ScalarInteger(int);
SEXP R_do_slot(SEXP, SEXP);
int R_has_slot(SEXP, SEXP);
R_xlen_t Rf_xlength(SEXP);
int Rf_length(SEXP);
int Rf_asInteger(SEXP);
int Rf_asLogical(SEXP);
double Rf_asReal(SEXP);
Rboolean Rf_isNull(SEXP);
Rboolean Rf_isReal(SEXP);
Rboolean Rf_isInteger(SEXP);
Rboolean Rf_isLogical(SEXP);
Rboolean Rf_isFunction(SEXP);
Rboolean Rf_isEnvironment(SEXP);
Rboolean Rf_isString(SEXP);
Rboolean Rf_isNewList(SEXP);
double* REAL(SEXP);
int* INTEGER(SEXP);
int* LOGICAL(SEXP);
Rbyte* RAW(SEXP);
SEXP VECTOR_ELT(SEXP, R_xlen_t);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
const char* R_CHAR(SEXP);
#define CHAR(x) R_CHAR(x)
void Rf_error(const char*, ...) __attribute__((noreturn));
void Rf_warning(const char*, ...);
void Rprintf(const char*, ...);
int R_IsNA(double);
#define ISNA(x) R_IsNA(x)
extern double R_NaReal;
typedef void (*R_CFinalizer_t)(SEXP);
SEXP R_MakeExternalPtr(void* p, SEXP tag, SEXP prot);
void* R_ExternalPtrAddr(SEXP);
void R_ClearExternalPtr(SEXP);
void R_RegisterCFinalizerEx(SEXP, R_CFinalizer_t, Rboolean onexit);
Rboolean R_ToplevelExec(void (*fun)(void*), void* data);
void R_CheckUserInterrupt(void);
void Rf_onintr(void);
#ifdef __cplusplus
}
#endif
#endif
