// TEST INFRASTRUCTURE ONLY — an in-memory implementation of the subset of R's C API that shim/init_shim.cpp uses (the functions
// declared in tests/r_api_decl), so that the shim can be LINKED AND EXECUTED without R: SEXPs are tagged structs, Rf_error is a
// C++ exception, `.Random.seed` lives in a global table, closures are C callbacks, R_ToplevelExec / R_CheckUserInterrupt model
// the user interrupt.  It is not an R stand-in and is never shipped: it exists so that tests/test_shim_exec.py can drive the
// twelve `.Call` routines with the argument lists the reference's R code passes (R/stan4bart_fit.R:42-57,579; R/generics.R:190,667)
// and compare them with the ctypes path.  The `mock_*` functions at the bottom are the test harness's handle on it.
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

enum { M_SYMSXP = 1, M_CLOSXP = 3, M_ENVSXP = 4, M_LANGSXP = 6, M_CHARSXP = 9, M_EXTPTRSXP = 22, M_S4SXP = 25 };

typedef SEXP (*mock_closure_fn)(SEXP a, SEXP b, SEXP c, void* user);

struct SEXPREC {
  unsigned type = NILSXP;
  std::vector<double> r; std::vector<int> i; std::vector<unsigned char> raw; std::vector<SEXP> v;
  std::string s;                                  // CHARSXP / SYMSXP
  std::map<std::string, SEXP> attr, slots;        // attributes by symbol name; S4 slots
  std::map<std::string, SEXP> vars;               // ENVSXP
  void* ext = nullptr; SEXP prot = nullptr; R_CFinalizer_t fin = nullptr;
  mock_closure_fn fn = nullptr; void* fnUser = nullptr;   // CLOSXP
};

struct MockRError { std::string msg; };
struct MockInterrupt {};

namespace {
std::vector<SEXP> g_all;            // every object ever allocated (freed by mock_reset)
int g_protect = 0, g_protectMax = 0, g_underflow = 0;
bool g_interruptPending = false;
std::string g_printed, g_lastError;
std::vector<std::string> g_warnings;
const R_CallMethodDef* g_routines = nullptr;
SEXP mk(unsigned type) { SEXP s = new SEXPREC; s->type = type; g_all.push_back(s); return s; }
SEXP sym(const char* n) { static std::map<std::string, SEXP> tab; auto it = tab.find(n); if (it != tab.end()) return it->second; SEXP s = new SEXPREC; s->type = M_SYMSXP; s->s = n; tab[n] = s; return s; }
SEXP the_nil() { static SEXP n = new SEXPREC; return n; }
SEXP the_global() { static SEXP e = [] { SEXP x = new SEXPREC; x->type = M_ENVSXP; return x; }(); return e; }
SEXP the_unbound() { static SEXP u = [] { SEXP x = new SEXPREC; x->type = M_SYMSXP; x->s = "<unbound>"; return x; }(); return u; }
}  // namespace

extern "C" {

SEXP R_NilValue = the_nil(), R_GlobalEnv = the_global(), R_UnboundValue = the_unbound();
SEXP R_NamesSymbol = sym("names"), R_DimSymbol = sym("dim"), R_DimNamesSymbol = sym("dimnames"), R_ClassSymbol = sym("class"),
     R_RowNamesSymbol = sym("row.names"), R_SeedsSymbol = sym(".Random.seed");
int R_NaInt = INT32_MIN;
double R_NaReal = [] { union { double d; unsigned long long u; } x; x.u = 0x7FF00000000007A2ull; return x.d; }();   // R's NA_real_ payload 1954

SEXP Rf_protect(SEXP s) { ++g_protect; if (g_protect > g_protectMax) g_protectMax = g_protect; return s; }
void Rf_unprotect(int n) { g_protect -= n; if (g_protect < 0) { ++g_underflow; g_protect = 0; } }
SEXP Rf_allocVector(SEXPTYPE t, R_xlen_t n) {
  SEXP s = mk(t);
  switch (t) {
    case LGLSXP: case INTSXP: s->i.assign((size_t)n, 0); break;
    case REALSXP: s->r.assign((size_t)n, 0.0); break;
    case RAWSXP: s->raw.assign((size_t)n, 0); break;
    case STRSXP: case VECSXP: s->v.assign((size_t)n, t == STRSXP ? mk(M_CHARSXP) : R_NilValue); break;
    default: throw MockRError{"mock Rf_allocVector: unsupported type"};
  }
  return s;
}
SEXP Rf_allocMatrix(SEXPTYPE t, int nr, int nc) {
  SEXP s = Rf_allocVector(t, (R_xlen_t)nr * nc);
  SEXP d = Rf_allocVector(INTSXP, 2); d->i[0] = nr; d->i[1] = nc;
  s->attr["dim"] = d;
  return s;
}
SEXP Rf_install(const char* n) { return sym(n); }
SEXP Rf_mkChar(const char* c) { SEXP s = mk(M_CHARSXP); s->s = c; return s; }
SEXP Rf_mkString(const char* c) { SEXP s = Rf_allocVector(STRSXP, 1); s->v[0] = Rf_mkChar(c); return s; }
SEXP Rf_getAttrib(SEXP x, SEXP name) { auto it = x->attr.find(name->s); return it == x->attr.end() ? R_NilValue : it->second; }
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP v) { if (v == R_NilValue) x->attr.erase(name->s); else x->attr[name->s] = v; return v; }
SEXP Rf_findVar(SEXP name, SEXP env) { auto it = env->vars.find(name->s); return it == env->vars.end() ? R_UnboundValue : it->second; }
void Rf_defineVar(SEXP name, SEXP v, SEXP env) { env->vars[name->s] = v; }
SEXP Rf_lang4(SEXP f, SEXP a, SEXP b, SEXP c) { SEXP s = mk(M_LANGSXP); s->v = {f, a, b, c}; return s; }
SEXP Rf_eval(SEXP call, SEXP) {
  if (call->type != M_LANGSXP || call->v.size() != 4 || call->v[0]->type != M_CLOSXP || !call->v[0]->fn) throw MockRError{"mock Rf_eval: not a call of a mock closure"};
  return call->v[0]->fn(call->v[1], call->v[2], call->v[3], call->v[0]->fnUser);
}
SEXP Rf_ScalarReal(double x) { SEXP s = Rf_allocVector(REALSXP, 1); s->r[0] = x; return s; }
SEXP Rf_ScalarInteger(int x) { SEXP s = Rf_allocVector(INTSXP, 1); s->i[0] = x; return s; }
SEXP R_do_slot(SEXP obj, SEXP name) {
  auto it = obj->slots.find(name->s);
  if (it == obj->slots.end()) throw MockRError{"no slot of name \"" + name->s + "\" for this object"};
  return it->second;
}
int R_has_slot(SEXP obj, SEXP name) { return obj->slots.count(name->s) ? 1 : 0; }
R_xlen_t Rf_xlength(SEXP s) {
  switch (s->type) {
    case LGLSXP: case INTSXP: return (R_xlen_t)s->i.size();
    case REALSXP: return (R_xlen_t)s->r.size();
    case RAWSXP: return (R_xlen_t)s->raw.size();
    case STRSXP: case VECSXP: return (R_xlen_t)s->v.size();
    case NILSXP: return 0;
    default: return 1;
  }
}
int Rf_length(SEXP s) { return (int)Rf_xlength(s); }
int Rf_asInteger(SEXP s) {
  if (Rf_xlength(s) < 1) return R_NaInt;
  if (s->type == INTSXP || s->type == LGLSXP) return s->i[0];
  if (s->type == REALSXP) return R_IsNA(s->r[0]) || s->r[0] != s->r[0] ? R_NaInt : (int)s->r[0];
  return R_NaInt;
}
int Rf_asLogical(SEXP s) {
  if (Rf_xlength(s) < 1) return R_NaInt;
  if (s->type == LGLSXP) return s->i[0];
  if (s->type == INTSXP) return s->i[0] == R_NaInt ? R_NaInt : (s->i[0] != 0);
  if (s->type == REALSXP) return s->r[0] != s->r[0] ? R_NaInt : (s->r[0] != 0.0);
  return R_NaInt;
}
double Rf_asReal(SEXP s) {
  if (Rf_xlength(s) < 1) return R_NaReal;
  if (s->type == REALSXP) return s->r[0];
  if (s->type == INTSXP || s->type == LGLSXP) return s->i[0] == R_NaInt ? R_NaReal : (double)s->i[0];
  return R_NaReal;
}
Rboolean Rf_isNull(SEXP s) { return s->type == NILSXP ? TRUE : FALSE; }
Rboolean Rf_isReal(SEXP s) { return s->type == REALSXP ? TRUE : FALSE; }
Rboolean Rf_isInteger(SEXP s) { return s->type == INTSXP ? TRUE : FALSE; }
Rboolean Rf_isLogical(SEXP s) { return s->type == LGLSXP ? TRUE : FALSE; }
Rboolean Rf_isFunction(SEXP s) { return s->type == M_CLOSXP ? TRUE : FALSE; }
Rboolean Rf_isEnvironment(SEXP s) { return s->type == M_ENVSXP ? TRUE : FALSE; }
Rboolean Rf_isString(SEXP s) { return s->type == STRSXP ? TRUE : FALSE; }
Rboolean Rf_isNewList(SEXP s) { return (s->type == VECSXP || s->type == NILSXP) ? TRUE : FALSE; }
double* REAL(SEXP s) { if (s->type != REALSXP) throw MockRError{"REAL() can only be applied to a 'numeric'"}; return s->r.data(); }
int* INTEGER(SEXP s) { if (s->type != INTSXP && s->type != LGLSXP) throw MockRError{"INTEGER() can only be applied to a 'integer'"}; return s->i.data(); }
int* LOGICAL(SEXP s) { if (s->type != LGLSXP) throw MockRError{"LOGICAL() can only be applied to a 'logical'"}; return s->i.data(); }
Rbyte* RAW(SEXP s) { if (s->type != RAWSXP) throw MockRError{"RAW() can only be applied to a 'raw'"}; return s->raw.data(); }
SEXP VECTOR_ELT(SEXP s, R_xlen_t i) { if (s->type != VECSXP || i < 0 || (size_t)i >= s->v.size()) throw MockRError{"VECTOR_ELT out of range / not a list"}; return s->v[(size_t)i]; }
SEXP SET_VECTOR_ELT(SEXP s, R_xlen_t i, SEXP x) { if (s->type != VECSXP || i < 0 || (size_t)i >= s->v.size()) throw MockRError{"SET_VECTOR_ELT out of range / not a list"}; s->v[(size_t)i] = x; return x; }
SEXP STRING_ELT(SEXP s, R_xlen_t i) { if (s->type != STRSXP || i < 0 || (size_t)i >= s->v.size()) throw MockRError{"STRING_ELT out of range / not a character vector"}; return s->v[(size_t)i]; }
void SET_STRING_ELT(SEXP s, R_xlen_t i, SEXP x) { if (s->type != STRSXP || i < 0 || (size_t)i >= s->v.size()) throw MockRError{"SET_STRING_ELT out of range"}; s->v[(size_t)i] = x; }
const char* R_CHAR(SEXP s) { return s->s.c_str(); }
void Rf_error(const char* fmt, ...) {
  char buf[4096]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  throw MockRError{buf};
}
void Rf_warning(const char* fmt, ...) { char buf[4096]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap); g_warnings.push_back(buf); }
void Rprintf(const char* fmt, ...) { char buf[4096]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap); g_printed += buf; }
int R_IsNA(double x) { union { double d; unsigned long long u; } v; v.d = x; return x != x && (unsigned)(v.u & 0xFFFFFFFFull) == 1954u; }
SEXP R_MakeExternalPtr(void* p, SEXP, SEXP prot) { SEXP s = mk(M_EXTPTRSXP); s->ext = p; s->prot = prot; return s; }
void* R_ExternalPtrAddr(SEXP s) { return s->ext; }
void R_ClearExternalPtr(SEXP s) { s->ext = nullptr; }
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t f, Rboolean) { s->fin = f; }
Rboolean R_ToplevelExec(void (*fun)(void*), void* data) { try { fun(data); } catch (const MockInterrupt&) { return FALSE; } catch (const MockRError&) { return FALSE; } return TRUE; }
void R_CheckUserInterrupt(void) { if (g_interruptPending) { g_interruptPending = false; throw MockInterrupt{}; } }
void Rf_onintr(void) { throw MockInterrupt{}; }
void GetRNGstate(void) {
  // (an R session materialises .Random.seed on the first draw; the harness seeds it explicitly: nothing to do when it exists)
  if (R_GlobalEnv->vars.find(".Random.seed") == R_GlobalEnv->vars.end()) throw MockRError{"mock: .Random.seed was not set by the test harness"};
}
void PutRNGstate(void) {}
int R_registerRoutines(DllInfo*, const void*, const R_CallMethodDef* calls, const void*, const void*) { g_routines = calls; return 1; }
Rboolean R_useDynamicSymbols(DllInfo*, Rboolean) { return FALSE; }

void R_init_stan4bart(DllInfo*);

// ================================================================================================ harness (ctypes)
void mock_init(void) { static bool done = false; if (!done) { R_init_stan4bart(nullptr); done = true; } }
const char* mock_last_error(void) { return g_lastError.c_str(); }
const char* mock_printed(void) { return g_printed.c_str(); }
void mock_clear_printed(void) { g_printed.clear(); }
int mock_num_warnings(void) { return (int)g_warnings.size(); }
int mock_protect_depth(void) { return g_protect; }
int mock_protect_underflows(void) { return g_underflow; }
void mock_set_interrupt_pending(int v) { g_interruptPending = v != 0; }
SEXP mock_nil(void) { return R_NilValue; }
SEXP mock_real(const double* x, int64_t n) { SEXP s = Rf_allocVector(REALSXP, (R_xlen_t)n); if (n) std::memcpy(s->r.data(), x, (size_t)n * 8); return s; }
SEXP mock_int(const int* x, int64_t n) { SEXP s = Rf_allocVector(INTSXP, (R_xlen_t)n); if (n) std::memcpy(s->i.data(), x, (size_t)n * 4); return s; }
SEXP mock_lgl(int v) { SEXP s = Rf_allocVector(LGLSXP, 1); s->i[0] = v; return s; }
SEXP mock_str(const char* c) { return Rf_mkString(c); }
SEXP mock_raw(const unsigned char* x, int64_t n) { SEXP s = Rf_allocVector(RAWSXP, (R_xlen_t)n); if (n) std::memcpy(s->raw.data(), x, (size_t)n); return s; }
SEXP mock_list(int64_t n) { return Rf_allocVector(VECSXP, (R_xlen_t)n); }
void mock_list_set(SEXP l, int64_t i, const char* name, SEXP v) {
  SET_VECTOR_ELT(l, (R_xlen_t)i, v);
  if (name) {
    SEXP nm = Rf_getAttrib(l, R_NamesSymbol);
    if (Rf_isNull(nm)) { nm = Rf_allocVector(STRSXP, Rf_xlength(l)); Rf_setAttrib(l, R_NamesSymbol, nm); }
    SET_STRING_ELT(nm, (R_xlen_t)i, Rf_mkChar(name));
  }
}
void mock_set_dim(SEXP x, int nr, int nc) { SEXP d = Rf_allocVector(INTSXP, 2); d->i[0] = nr; d->i[1] = nc; x->attr["dim"] = d; }
void mock_set_attr(SEXP x, const char* name, SEXP v) { x->attr[name] = v; }
SEXP mock_s4(void) { return mk(M_S4SXP); }
void mock_set_slot(SEXP obj, const char* name, SEXP v) { obj->slots[name] = v; }
SEXP mock_env(void) { return mk(M_ENVSXP); }
SEXP mock_closure(mock_closure_fn fn, void* user) { SEXP s = mk(M_CLOSXP); s->fn = fn; s->fnUser = user; return s; }
void mock_set_seed(const int* seed626) { SEXP s = Rf_allocVector(INTSXP, 626); std::memcpy(s->i.data(), seed626, 626 * 4); R_GlobalEnv->vars[".Random.seed"] = s; }
int mock_get_seed(int* out626) { auto it = R_GlobalEnv->vars.find(".Random.seed"); if (it == R_GlobalEnv->vars.end()) return 1; std::memcpy(out626, it->second->i.data(), 626 * 4); return 0; }
// accessors
int mock_type(SEXP s) { return (int)s->type; }
int64_t mock_length(SEXP s) { return (int64_t)Rf_xlength(s); }
double* mock_real_ptr(SEXP s) { return s->r.data(); }
int* mock_int_ptr(SEXP s) { return s->i.data(); }
unsigned char* mock_raw_ptr(SEXP s) { return s->raw.data(); }
SEXP mock_elt(SEXP s, int64_t i) { return s->v[(size_t)i]; }
const char* mock_chars(SEXP charsxp) { return charsxp->s.c_str(); }
SEXP mock_get_attr(SEXP s, const char* name) { auto it = s->attr.find(name); return it == s->attr.end() ? R_NilValue : it->second; }
// run the finalizer R's garbage collector would run
void mock_finalize(SEXP extptr) { if (extptr->fin) extptr->fin(extptr); }

// .Call(name, args...): 0 = returned normally (result in *out), 1 = R error (mock_last_error), 2 = interrupt, 3 = unknown routine / arity
int mock_call(const char* name, int nargs, SEXP* args, SEXP* out) {
  mock_init();
  const R_CallMethodDef* m = g_routines;
  for (; m && m->name; ++m) if (std::strcmp(m->name, name) == 0) break;
  if (!m || !m->name || m->numArgs != nargs) { g_lastError = std::string("no routine ") + name + " with that arity"; return 3; }
  try {
    SEXP r = R_NilValue;
    switch (nargs) {
      case 0: r = ((SEXP (*)(void))m->fun)(); break;
      case 1: r = ((SEXP (*)(SEXP))m->fun)(args[0]); break;
      case 3: r = ((SEXP (*)(SEXP, SEXP, SEXP))m->fun)(args[0], args[1], args[2]); break;
      case 4: r = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP))m->fun)(args[0], args[1], args[2], args[3]); break;
      case 5: r = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP))m->fun)(args[0], args[1], args[2], args[3], args[4]); break;
      case 6: r = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))m->fun)(args[0], args[1], args[2], args[3], args[4], args[5]); break;
      default: g_lastError = "arity not wired in the mock"; return 3;
    }
    if (out) *out = r;
    return 0;
  } catch (const MockRError& e) { g_lastError = e.msg; g_protect = 0; return 1; }   // (R unwinds the protect stack at the top level)
  catch (const MockInterrupt&) { g_lastError = "interrupt"; g_protect = 0; return 2; }
}

}  // extern "C"
