// R `.Call` shim: the reference's twelve registered routines (reference src/init.cpp:1215-1229, same names, same arity,
// same argument and result shapes) implemented on top of the C-ABI of include/stan4bart_amd.h.  With this file built into the
// package's shared object instead of the reference's src/, R/stan4bart_fit.R and R/generics.R run unchanged: they only ever
// talk to these routines.
//
//   R CMD SHLIB shim/init_shim.cpp -I include -L stan4bart_amd/csrc -ls4b      (needs R; not built in this repository's image)
//   g++ -std=c++17 -fsyntax-only -Iinclude -Itests/r_api_decl shim/init_shim.cpp   (what tests/test_shim.py checks here)
//
// What it unpacks (and where the reference does the same):
//   bartControl / bartData / bartModel  dbarts S4 objects; the slots stan4bart relies on (SURVEY.md §8b "Argument schemas";
//                                       the reference hands them to dbarts' own initializeControl/Data/Model, src/init.cpp:215-225)
//   stanData      named list, exactly the 44 names of dataNames[] (reference src/stan_sampler.cpp:67-80), matched by name
//   stanControl   named list: seed (required) + 12 optional fields with the reference's defaults (src/stan_sampler.cpp:395-458)
//   commonControl named list of 12 fields (src/init.cpp:1015-1051)
// R's generator: dbarts draws from unif_rand() in single-chain mode; the device needs the Mersenne-Twister state itself, so
// every entry point that draws copies `.Random.seed` in and writes the advanced state back (GetRNGstate / PutRNGstate
// bracket in the reference: src/init.cpp:259,298,750,919).
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

#include <cstdint>
#include <cstring>
#include <set>
#include <string>
#include <limits>
#include <vector>

#include "stan4bart_amd.h"

namespace {

// ------------------------------------------------------------------------------------------------ small SEXP helpers
SEXP list_element(SEXP list, const char* name) {
  SEXP names = Rf_getAttrib(list, R_NamesSymbol);
  if (Rf_isNull(names)) return R_NilValue;
  for (R_xlen_t i = 0; i < Rf_xlength(list); ++i)
    if (std::strcmp(CHAR(STRING_ELT(names, i)), name) == 0) return VECTOR_ELT(list, i);
  return R_NilValue;
}
SEXP need_element(SEXP list, const char* name, const char* what) {
  SEXP e = list_element(list, name);
  if (Rf_isNull(e)) Rf_error("%s requires '%s' to be specified", what, name);
  return e;
}
SEXP slot(SEXP obj, const char* name) { return R_do_slot(obj, Rf_install(name)); }
bool has_slot(SEXP obj, const char* name) { return R_has_slot(obj, Rf_install(name)) != 0; }
int int_or(SEXP e, int dflt) { return (Rf_isNull(e) || Rf_length(e) == 0 || Rf_asInteger(e) == NA_INTEGER) ? dflt : Rf_asInteger(e); }
double real_or(SEXP e, double dflt) { return (Rf_isNull(e) || Rf_length(e) == 0 || ISNA(Rf_asReal(e))) ? dflt : Rf_asReal(e); }
const double* reals_or_null(SEXP e) { return (Rf_isNull(e) || Rf_length(e) == 0) ? nullptr : REAL(e); }
std::vector<int32_t> ints(SEXP e) {
  std::vector<int32_t> v((size_t)Rf_xlength(e));
  if (Rf_isInteger(e) || Rf_isLogical(e)) for (size_t i = 0; i < v.size(); ++i) v[i] = INTEGER(e)[i];
  else if (Rf_isReal(e)) for (size_t i = 0; i < v.size(); ++i) v[i] = (int32_t)REAL(e)[i];
  return v;
}
void check(int status) { if (status != 0) Rf_error("%s", s4b_last_error()); }

// ------------------------------------------------------------------------------------------------ R's generator state
// .Random.seed = c(kind code, mti, mt[624]); the device path needs Mersenne-Twister / Inversion (R's defaults since 3.6.0)
void read_r_rng(uint32_t state[S4B_R_RNG_WORDS]) {
  GetRNGstate(); PutRNGstate();   // materialises .Random.seed if the session has not drawn yet
  SEXP seed = Rf_findVar(R_SeedsSymbol, R_GlobalEnv);
  if (seed == R_UnboundValue || !Rf_isInteger(seed) || Rf_xlength(seed) != 626 || INTEGER(seed)[0] % 100 != 3)
    Rf_error("stan4bart on the GPU needs RNGkind('Mersenne-Twister', 'Inversion')");
  for (int i = 0; i < S4B_R_RNG_WORDS; ++i) state[i] = (uint32_t)INTEGER(seed)[1 + i];
}
void write_r_rng(const uint32_t state[S4B_R_RNG_WORDS]) {
  SEXP old = Rf_findVar(R_SeedsSymbol, R_GlobalEnv);
  SEXP seed = PROTECT(Rf_allocVector(INTSXP, 626));
  INTEGER(seed)[0] = INTEGER(old)[0];
  for (int i = 0; i < S4B_R_RNG_WORDS; ++i) INTEGER(seed)[1 + i] = (int)state[i];
  Rf_defineVar(R_SeedsSymbol, seed, R_GlobalEnv);
  UNPROTECT(1);
}

// ------------------------------------------------------------------------------------------------ sampler objects
struct Sampler {                 // reference `Sampler`, src/init.cpp:124-173
  s4b_sampler* h = nullptr;
  int64_t numPars = 0, n = 0, nTest = 0, p = 0, nTrees = 0;
  bool keepFits = true, keepTrees = false, kModeled = false; int verbose = 0, refresh = 200;
  SEXP callback = R_NilValue, callbackEnv = R_NilValue;
  std::vector<std::string> parNames;
  // per run(): collected callback results
  std::vector<double> cbValues; R_xlen_t cbLength = 0; bool interrupted = false;
};
struct StoredSampler {           // reference `StoredBARTSampler`, src/init.cpp:175-188: the kept trees of every chain
  std::vector<s4b_sampler*> chains; int64_t p = 0, nTrees = 0;
};
std::set<SEXP>* activeSamplers = nullptr;
std::set<SEXP>* activeStored = nullptr;

void sampler_finalizer(SEXP ptr) {
  Sampler* s = static_cast<Sampler*>(R_ExternalPtrAddr(ptr));
  if (!s) return;
  if (activeSamplers) activeSamplers->erase(ptr);
  s4b_free(s->h);
  delete s;
  R_ClearExternalPtr(ptr);
}
void stored_finalizer(SEXP ptr) {
  StoredSampler* s = static_cast<StoredSampler*>(R_ExternalPtrAddr(ptr));
  if (!s) return;
  if (activeStored) activeStored->erase(ptr);
  for (s4b_sampler* h : s->chains) s4b_free(h);
  delete s;
  R_ClearExternalPtr(ptr);
}
Sampler& sampler_of(SEXP ptr, const char* who) {
  Sampler* s = static_cast<Sampler*>(R_ExternalPtrAddr(ptr));
  if (!s) Rf_error("%s called on NULL external pointer", who);
  return *s;
}
StoredSampler& stored_of(SEXP ptr, const char* who) {
  StoredSampler* s = static_cast<StoredSampler*>(R_ExternalPtrAddr(ptr));
  if (!s) Rf_error("%s called on NULL external pointer", who);
  return *s;
}

// ------------------------------------------------------------------------------------------------ callbacks into R
// per-iteration user callback (reference src/init.cpp:849-911): callback(yhat.train, yhat.test, stan.pars) evaluated in callbackEnv
int callback_trampoline(void* user, const double* train, const double* test, const double* pars, int32_t numPars) {
  Sampler& s = *static_cast<Sampler*>(user);
  SEXP tr = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)s.n));
  std::memcpy(REAL(tr), train, (size_t)s.n * sizeof(double));
  SEXP te = R_NilValue;
  if (s.nTest > 0) { te = Rf_allocVector(REALSXP, (R_xlen_t)s.nTest); std::memcpy(REAL(te), test, (size_t)s.nTest * sizeof(double)); }
  PROTECT(te);
  SEXP sp = PROTECT(Rf_allocVector(REALSXP, numPars));
  std::memcpy(REAL(sp), pars, (size_t)numPars * sizeof(double));
  SEXP nm = PROTECT(Rf_allocVector(STRSXP, numPars));
  for (int i = 0; i < numPars; ++i) SET_STRING_ELT(nm, i, Rf_mkChar(s.parNames[(size_t)i].c_str()));
  Rf_setAttrib(sp, R_NamesSymbol, nm);
  SEXP call = PROTECT(Rf_lang4(s.callback, tr, te, sp));
  SEXP res = PROTECT(Rf_eval(call, Rf_isNull(s.callbackEnv) ? R_GlobalEnv : s.callbackEnv));
  if (!Rf_isNull(res) && Rf_isReal(res) && Rf_xlength(res) > 0) {
    if (s.cbLength == 0) s.cbLength = Rf_xlength(res);
    if (Rf_xlength(res) == s.cbLength) s.cbValues.insert(s.cbValues.end(), REAL(res), REAL(res) + s.cbLength);
  }
  UNPROTECT(6);
  return 0;
}
// progress + user interrupt (reference: "iter k / n" src/init.cpp:752-754; R_CheckUserInterrupt src/stan_sampler.hpp:44-48).
// R_CheckUserInterrupt longjmps; run it under R_ToplevelExec so that the C++ frames of the library unwind normally.
void check_interrupt(void*) { R_CheckUserInterrupt(); }
int progress_trampoline(void* user, int32_t iter, int32_t numIter, int32_t) {
  Sampler& s = *static_cast<Sampler*>(user);
  if (s.refresh > 0 && s.verbose > 1 && iter % s.refresh == 0) Rprintf("  iter %.3d / %.3d\n", iter, numIter);
  if (R_ToplevelExec(check_interrupt, nullptr) == FALSE) { s.interrupted = true; return 1; }
  return 0;
}

// ------------------------------------------------------------------------------------------------ argument unpacking
struct StanDataHold { std::vector<int32_t> p, l, v, u, numNormals; };   // integer vectors R may hand over as doubles

s4b_stan_data unpack_stan_data(SEXP d, StanDataHold& hold) {
  // all 44 names of the reference's dataNames[] must be present (src/stan_sampler.cpp:118-139); the ones this path does not
  // consume (intercept priors: has_intercept is always 0, R/stan4bart_fit.R:259-365; len_y / lb_y / ub_y) are still required
  static const char* const names[44] = {
    "N", "K", "X", "len_y", "lb_y", "ub_y", "y", "has_intercept", "is_binary", "prior_dist", "prior_dist_for_intercept",
    "prior_dist_for_aux", "has_weights", "weights", "offset_", "prior_scale", "prior_scale_for_intercept", "prior_scale_for_aux",
    "prior_mean", "prior_mean_for_intercept", "prior_mean_for_aux", "prior_df", "prior_df_for_intercept", "prior_df_for_aux",
    "global_prior_df", "global_prior_scale", "slab_df", "slab_scale", "num_normals", "t", "p", "l", "q", "len_theta_L", "shape",
    "scale", "len_concentration", "concentration", "len_regularization", "regularization", "num_non_zero", "w", "v", "u"};
  SEXP nm = Rf_getAttrib(d, R_NamesSymbol);
  if (Rf_isNull(nm)) Rf_error("names for stanData object cannot be NULL");
  for (const char* n : names) {
    bool found = false;
    for (R_xlen_t i = 0; i < Rf_xlength(nm) && !found; ++i) found = std::strcmp(CHAR(STRING_ELT(nm, i)), n) == 0;
    if (!found) Rf_error("stanData requires '%s' to be specified", n);
  }
  s4b_stan_data s; std::memset(&s, 0, sizeof(s));
  s.N = Rf_asInteger(list_element(d, "N")); s.K = Rf_asInteger(list_element(d, "K"));
  s.is_binary = Rf_asInteger(list_element(d, "is_binary")); s.has_intercept = Rf_asInteger(list_element(d, "has_intercept"));
  s.has_weights = Rf_asInteger(list_element(d, "has_weights"));
  s.prior_dist = Rf_asInteger(list_element(d, "prior_dist")); s.prior_dist_for_aux = Rf_asInteger(list_element(d, "prior_dist_for_aux"));
  SEXP X = list_element(d, "X");   // array[1] of an N x K matrix (continuous.stan: matrix[N, K] X[1]): column-major N x K either way
  if (!Rf_isReal(X) || Rf_xlength(X) != (R_xlen_t)s.N * s.K) Rf_error("X must be a real N x K matrix");
  s.X = REAL(X);
  SEXP y = list_element(d, "y");
  if (!Rf_isReal(y) || Rf_xlength(y) != s.N) Rf_error("y must be a real vector of length N");
  s.y = REAL(y);
  s.weights = s.has_weights ? reals_or_null(list_element(d, "weights")) : nullptr;
  s.prior_scale = reals_or_null(list_element(d, "prior_scale")); s.prior_mean = reals_or_null(list_element(d, "prior_mean"));
  s.prior_df = reals_or_null(list_element(d, "prior_df"));
  s.prior_scale_for_aux = real_or(list_element(d, "prior_scale_for_aux"), 1.0);
  s.prior_mean_for_aux = real_or(list_element(d, "prior_mean_for_aux"), 0.0);
  s.prior_df_for_aux = real_or(list_element(d, "prior_df_for_aux"), 1.0);
  s.t = Rf_asInteger(list_element(d, "t")); s.q = Rf_asInteger(list_element(d, "q"));
  s.len_theta_L = Rf_asInteger(list_element(d, "len_theta_L"));
  s.len_concentration = Rf_asInteger(list_element(d, "len_concentration")); s.len_regularization = Rf_asInteger(list_element(d, "len_regularization"));
  hold.p = ints(list_element(d, "p")); hold.l = ints(list_element(d, "l"));
  s.p = hold.p.data(); s.l = hold.l.data();
  s.shape = reals_or_null(list_element(d, "shape")); s.scale = reals_or_null(list_element(d, "scale"));
  s.concentration = reals_or_null(list_element(d, "concentration")); s.regularization = reals_or_null(list_element(d, "regularization"));
  s.num_non_zero = Rf_asInteger(list_element(d, "num_non_zero"));
  s.w = reals_or_null(list_element(d, "w"));
  hold.v = ints(list_element(d, "v")); hold.u = ints(list_element(d, "u"));
  s.v = hold.v.data(); s.u = hold.u.data();
  s.global_prior_df = real_or(list_element(d, "global_prior_df"), 1.0); s.global_prior_scale = real_or(list_element(d, "global_prior_scale"), 1.0);
  s.slab_df = real_or(list_element(d, "slab_df"), 1.0); s.slab_scale = real_or(list_element(d, "slab_scale"), 1.0);
  hold.numNormals = ints(list_element(d, "num_normals"));
  s.num_normals = hold.numNormals.empty() ? nullptr : hold.numNormals.data();
  return s;
}

s4b_stan_control unpack_stan_control(SEXP c) {
  // seed is required; the 12 others default as in the reference (src/stan_sampler.cpp:395-458)
  if (Rf_isNull(Rf_getAttrib(c, R_NamesSymbol))) Rf_error("names for stanControl object cannot be NULL");
  s4b_stan_control s; std::memset(&s, 0, sizeof(s));
  s.seed = (uint32_t)Rf_asInteger(need_element(c, "seed", "stanControl"));
  s.init_r = real_or(list_element(c, "init_r"), 2.0);
  s.skip = int_or(list_element(c, "skip"), -1);                       // NA: max(1, (2000 - warmup) / 1000), resolved by the library
  s.adapt_gamma = real_or(list_element(c, "adapt_gamma"), 0.05);
  s.adapt_delta = real_or(list_element(c, "adapt_delta"), 0.8);
  s.adapt_kappa = real_or(list_element(c, "adapt_kappa"), 0.75);
  s.adapt_init_buffer = (uint32_t)int_or(list_element(c, "adapt_init_buffer"), 75);
  s.adapt_term_buffer = (uint32_t)int_or(list_element(c, "adapt_term_buffer"), 50);
  s.adapt_window = (uint32_t)int_or(list_element(c, "adapt_window"), 25);
  s.adapt_t0 = real_or(list_element(c, "adapt_t0"), 10.0);
  s.stepsize = real_or(list_element(c, "stepsize"), 1.0);
  s.stepsize_jitter = real_or(list_element(c, "stepsize_jitter"), 0.0);
  s.max_treedepth = int_or(list_element(c, "max_treedepth"), 10);
  s.hmc_mode = int_or(list_element(c, "hmc_mode"), 0);                // extension of this library (0: sufficient statistics)
  return s;
}

void unpack_bart(SEXP control, SEXP data, SEXP model, s4b_bart_control& bc, s4b_bart_data& bd, std::vector<int32_t>& nCuts) {
  std::memset(&bc, 0, sizeof(bc)); std::memset(&bd, 0, sizeof(bd));
  int splitProbsLen = -1;
  // dbartsControl (dbarts R sources; R/stan4bart_fit.R:437-452 sets n.trees, n.thin, keepTrees, binary)
  bc.n_trees = Rf_asInteger(slot(control, "n.trees"));
  bc.n_thin = Rf_asInteger(slot(control, "n.thin"));
  bc.keep_trees = Rf_asLogical(slot(control, "keepTrees")) == TRUE ? 1 : 0;
  bc.node_capacity = 0;
  // dbartsModel (R/stan4bart_fit.R:455-479): cgm(power, base) tree prior, normal(k) leaf prior, node.scale, proposal mix
  SEXP treePrior = slot(model, "tree.prior");
  bc.power = Rf_asReal(slot(treePrior, "power")); bc.base = Rf_asReal(slot(treePrior, "base"));
  SEXP nodePrior = slot(model, "node.prior");
  bc.k = has_slot(nodePrior, "k") ? real_or(slot(nodePrior, "k"), 2.0) : 2.0;
  // a hyperprior on k (bart_args$k = chi(df, scale), R/stan4bart_fit.R:460-465; R/stan4bart.R:202): model@node.hyperprior is then a
  // dbartsChiHyperprior(degreesOfFreedom, scale) instead of a fixed value — k is sampled (s4b_bart_control.k_hyper_df / k_hyper_scale) and
  // starts from dbarts' default 2
  bc.k_hyper_df = 0.0; bc.k_hyper_scale = std::numeric_limits<double>::infinity();
  if (has_slot(model, "node.hyperprior")) {
    SEXP hp = slot(model, "node.hyperprior");
    if (!Rf_isNull(hp) && has_slot(hp, "degreesOfFreedom")) {
      bc.k_hyper_df = Rf_asReal(slot(hp, "degreesOfFreedom"));
      bc.k_hyper_scale = has_slot(hp, "scale") ? Rf_asReal(slot(hp, "scale")) : std::numeric_limits<double>::infinity();
      if (!(bc.k_hyper_df > 0.0) || !(bc.k_hyper_scale > 0.0)) Rf_error("node.hyperprior: degreesOfFreedom and scale must be positive");
      bc.k = 2.0;
    } else if (!Rf_isNull(hp) && has_slot(hp, "k")) bc.k = real_or(slot(hp, "k"), bc.k);      // dbartsFixedHyperprior(k)
  }
  // cgm(split.probs = ): tree.prior@splitProbabilities, one weight per predictor (numeric(0) = equally likely)
  if (has_slot(treePrior, "splitProbabilities")) {
    SEXP sp = slot(treePrior, "splitProbabilities");
    if (!Rf_isNull(sp) && Rf_xlength(sp) > 0) {
      if (!Rf_isReal(sp)) Rf_error("tree.prior@splitProbabilities must be numeric");
      bc.split_probs = REAL(sp);      // (length checked against data@x below; borrowed: the S4 object outlives the call)
      splitProbsLen = (int)Rf_xlength(sp);
    }
  }
  // dbartsControl(useQuantiles = ) travels through bart_args (R/stan4bart_fit.R:440-444)
  bc.interface_version = S4B_INTERFACE_VERSION;
  bc.use_quantiles = (has_slot(control, "useQuantiles") && Rf_asLogical(slot(control, "useQuantiles")) == TRUE) ? 1 : 0;
  bc.node_scale = Rf_asReal(slot(model, "node.scale"));
  bc.birth_or_death_prob = Rf_asReal(slot(model, "p.birth_death")); bc.swap_prob = Rf_asReal(slot(model, "p.swap"));
  bc.change_prob = Rf_asReal(slot(model, "p.change")); bc.birth_prob = Rf_asReal(slot(model, "p.birth"));
  // dbartsData: x (n x p), n.cuts (attribute of the data or of the control: R/stan4bart_fit.R:446-451), x.test
  SEXP x = slot(data, "x");
  SEXP dims = Rf_getAttrib(x, R_DimSymbol);
  if (!Rf_isReal(x) || Rf_isNull(dims) || Rf_length(dims) != 2) Rf_error("data@x must be a real matrix");
  bd.n = INTEGER(dims)[0]; bd.p = INTEGER(dims)[1]; bd.x = REAL(x);
  if (splitProbsLen >= 0 && splitProbsLen != bd.p) Rf_error("tree.prior@splitProbabilities must have one entry per predictor");
  SEXP cuts = has_slot(data, "n.cuts") ? slot(data, "n.cuts") : Rf_getAttrib(control, Rf_install("n.cuts"));
  nCuts = ints(cuts);
  if (nCuts.size() == 1) nCuts.assign((size_t)bd.p, nCuts[0]);
  if ((int)nCuts.size() != bd.p) Rf_error("n.cuts must have one entry per predictor");
  bd.n_cuts = nCuts.data();
  SEXP xt = slot(data, "x.test");
  if (!Rf_isNull(xt) && Rf_xlength(xt) > 0) {
    SEXP dt = Rf_getAttrib(xt, R_DimSymbol);
    if (!Rf_isReal(xt) || Rf_isNull(dt) || INTEGER(dt)[1] != bd.p) Rf_error("data@x.test must be a real matrix with as many columns as data@x");
    bd.n_test = INTEGER(dt)[0]; bd.x_test = REAL(xt);
  }
}

}  // namespace

extern "C" {

// stan4bart_create(bartControl, bartData, bartModel, stanData, stanControl, commonControl) — reference src/init.cpp:190-310
static SEXP createSampler(SEXP bartControlExpr, SEXP bartDataExpr, SEXP bartModelExpr, SEXP stanDataExpr, SEXP stanControlExpr, SEXP commonControlExpr) {
  s4b_bart_control bc; s4b_bart_data bd; std::vector<int32_t> nCuts;
  unpack_bart(bartControlExpr, bartDataExpr, bartModelExpr, bc, bd, nCuts);
  StanDataHold hold;
  s4b_stan_data sd = unpack_stan_data(stanDataExpr, hold);
  s4b_stan_control sc = unpack_stan_control(stanControlExpr);
  // the 12 commonControl fields (src/init.cpp:199-202, 1015-1051)
  s4b_common_control cc; std::memset(&cc, 0, sizeof(cc));
  cc.warmup = int_or(list_element(commonControlExpr, "warmup"), 1000);
  cc.iter = int_or(list_element(commonControlExpr, "iter"), 2000);
  cc.verbose = int_or(list_element(commonControlExpr, "verbose"), 0);
  cc.refresh = int_or(list_element(commonControlExpr, "refresh"), 200);                 // NA -> 200 (src/init.cpp:1040-1041)
  SEXP isBinary = need_element(commonControlExpr, "is_binary", "commonControl");
  if (Rf_asLogical(isBinary) == NA_LOGICAL) Rf_error("is_binary cannot be NA");
  cc.is_binary = Rf_asLogical(isBinary) == TRUE ? 1 : 0;
  SEXP offset = list_element(commonControlExpr, "offset");
  cc.offset = (Rf_isNull(offset) || Rf_length(offset) == 0 || !Rf_isReal(offset)) ? nullptr : REAL(offset);
  cc.offset_type = int_or(list_element(commonControlExpr, "offset_type"), 0);
  cc.bart_offset_init = reals_or_null(list_element(commonControlExpr, "bart_offset_init"));
  cc.sigma_init = real_or(list_element(commonControlExpr, "sigma_init"), 1.0);
  SEXP keepFits = need_element(commonControlExpr, "keep_fits", "commonControl");
  if (Rf_asLogical(keepFits) == NA_LOGICAL) Rf_error("keep_fits cannot be NA");
  cc.keep_fits = Rf_asLogical(keepFits) == TRUE ? 1 : 0;
  cc.device = int_or(list_element(commonControlExpr, "device"), 0);                     // extension: HIP device ordinal of this chain
  SEXP callback = list_element(commonControlExpr, "callback");
  if (!Rf_isNull(callback) && !Rf_isFunction(callback)) Rf_error("callback must be a function or NULL");
  SEXP callbackEnv = list_element(commonControlExpr, "callbackEnv");
  if (!Rf_isNull(callbackEnv) && !Rf_isEnvironment(callbackEnv)) Rf_error("callbackEnv must be an environment or NULL");

  Sampler* s = new Sampler;
  s->keepFits = cc.keep_fits != 0; s->keepTrees = bc.keep_trees != 0; s->kModeled = bc.k_hyper_df > 0.0; s->verbose = cc.verbose; s->refresh = cc.refresh;
  s->callback = callback; s->callbackEnv = callbackEnv;
  if (!Rf_isNull(callback)) { cc.callback = callback_trampoline; cc.callback_user = s; }
  uint32_t rng[S4B_R_RNG_WORDS];
  read_r_rng(rng);
  if (s4b_create(&bc, &bd, &sd, &sc, &cc, rng, &s->h) != 0) { delete s; Rf_error("%s", s4b_last_error()); }
  check(s4b_get_r_rng_state(s->h, rng));
  write_r_rng(rng);
  int64_t dims[5];
  check(s4b_get_dims(s->h, dims));
  s->numPars = dims[0]; s->n = dims[1]; s->nTest = dims[2]; s->p = dims[3]; s->nTrees = dims[4];
  {
    std::vector<char> buf((size_t)64 * (size_t)(s->numPars + 8));
    check(s4b_get_stan_par_names(s->h, buf.data(), buf.size()));
    std::string all(buf.data());
    size_t pos = 0;
    while (pos <= all.size()) { size_t e = all.find('\n', pos); if (e == std::string::npos) e = all.size(); s->parNames.push_back(all.substr(pos, e - pos)); pos = e + 1; }
  }
  check(s4b_set_progress(s->h, progress_trampoline, s));
  // the callback closure and its environment must outlive the sampler: kept alive as the external pointer's protected value
  SEXP keep = PROTECT(Rf_allocVector(VECSXP, 2));
  SET_VECTOR_ELT(keep, 0, callback); SET_VECTOR_ELT(keep, 1, callbackEnv);
  SEXP result = PROTECT(R_MakeExternalPtr(s, R_NilValue, keep));
  R_RegisterCFinalizerEx(result, sampler_finalizer, TRUE);
  if (activeSamplers) activeSamplers->insert(result);
  UNPROTECT(2);
  return result;
}

// stan4bart_run(sampler, numIter, isWarmup, resultsType) — reference src/init.cpp:678-965.  resultsType: 0 both, 1 bart, 2 stan;
// R passes the string "both" (R/stan4bart_fit.R:49), which the reference's integer reader maps to the default "both"
static SEXP run(SEXP samplerExpr, SEXP numIterExpr, SEXP isWarmupExpr, SEXP resultsTypeExpr) {
  Sampler& s = sampler_of(samplerExpr, "run");
  const int numIter = Rf_asInteger(numIterExpr);
  if (numIter == NA_INTEGER || numIter < 1) Rf_error("numIter must be a positive integer");
  const bool isWarmup = Rf_asLogical(isWarmupExpr) == TRUE;
  int resultsType = (Rf_isInteger(resultsTypeExpr) || Rf_isReal(resultsTypeExpr)) ? Rf_asInteger(resultsTypeExpr) : 0;
  if (resultsType == NA_INTEGER || resultsType < 0 || resultsType > 2) resultsType = 0;
  const bool doStan = resultsType != 1, doBart = resultsType != 2;
  if (s.verbose > 0)
    Rprintf("starting %s, %d draws, %s\n", isWarmup ? "warmup" : "sampling", numIter,
            resultsType == 0 ? "both BART and Stan" : (resultsType == 1 ? "BART only" : "Stan only"));
  const R_xlen_t S = s.keepFits ? numIter : 1;
  int protectCount = 0;
  SEXP stan = R_NilValue, bart = R_NilValue;
  s4b_results out; std::memset(&out, 0, sizeof(out));
  if (s.keepFits && doStan) {   // [num_pars x S] with the parameter names as row names (reference src/stan_sampler.cpp:577-596)
    stan = PROTECT(Rf_allocMatrix(REALSXP, (int)s.numPars, (int)S)); ++protectCount;
    SEXP dimnames = PROTECT(Rf_allocVector(VECSXP, 2)); ++protectCount;
    SEXP rows = PROTECT(Rf_allocVector(STRSXP, (R_xlen_t)s.numPars)); ++protectCount;
    for (int64_t i = 0; i < s.numPars; ++i) SET_STRING_ELT(rows, (R_xlen_t)i, Rf_mkChar(s.parNames[(size_t)i].c_str()));
    SET_VECTOR_ELT(dimnames, 0, rows); SET_VECTOR_ELT(dimnames, 1, R_NilValue);
    Rf_setAttrib(stan, R_DimNamesSymbol, dimnames);
    out.stan = REAL(stan);
  }
  if (s.keepFits && doBart) {   // list(sigma, train, test, varcount[, k]) (reference src/bart_util.cpp:13-81: the fifth element only when k is modeled)
    const int nb = s.kModeled ? 5 : 4;
    bart = PROTECT(Rf_allocVector(VECSXP, nb)); ++protectCount;
    SEXP sigma = Rf_allocVector(REALSXP, S); SET_VECTOR_ELT(bart, 0, sigma);
    SEXP train = Rf_allocMatrix(REALSXP, (int)s.n, (int)S); SET_VECTOR_ELT(bart, 1, train);
    SEXP test = R_NilValue;
    if (s.nTest > 0) { test = Rf_allocMatrix(REALSXP, (int)s.nTest, (int)S); SET_VECTOR_ELT(bart, 2, test); }
    SEXP varcount = Rf_allocMatrix(INTSXP, (int)s.p, (int)S); SET_VECTOR_ELT(bart, 3, varcount);
    SEXP nm = PROTECT(Rf_allocVector(STRSXP, nb)); ++protectCount;
    SET_STRING_ELT(nm, 0, Rf_mkChar("sigma")); SET_STRING_ELT(nm, 1, Rf_mkChar("train")); SET_STRING_ELT(nm, 2, Rf_mkChar("test")); SET_STRING_ELT(nm, 3, Rf_mkChar("varcount"));
    if (s.kModeled) { SEXP kd = Rf_allocVector(REALSXP, S); SET_VECTOR_ELT(bart, 4, kd); SET_STRING_ELT(nm, 4, Rf_mkChar("k")); out.bart_k = REAL(kd); }
    Rf_setAttrib(bart, R_NamesSymbol, nm);
    out.bart_sigma = REAL(sigma); out.bart_train = REAL(train); out.bart_test = s.nTest > 0 ? REAL(test) : nullptr; out.bart_varcount = INTEGER(varcount);
  }
  s.cbValues.clear(); s.cbLength = 0; s.interrupted = false;
  uint32_t rng[S4B_R_RNG_WORDS];
  read_r_rng(rng);
  check(s4b_set_r_rng_state(s.h, rng));
  const int status = s4b_run(s.h, numIter, isWarmup ? 1 : 0, resultsType, s.keepFits ? &out : nullptr);
  if (s4b_get_r_rng_state(s.h, rng) == 0) write_r_rng(rng);
  if (s.interrupted) { UNPROTECT(protectCount); Rf_onintr(); return R_NilValue; }
  if (status != 0) { UNPROTECT(protectCount); Rf_error("%s", s4b_last_error()); }
  // list(stan = , bart = , callback = ) or, without kept fits, list(callback = )
  SEXP callbackResults = R_NilValue;
  if (!Rf_isNull(s.callback) && s.cbLength > 0) {
    const R_xlen_t got = (R_xlen_t)s.cbValues.size() / s.cbLength;
    callbackResults = PROTECT(Rf_allocMatrix(REALSXP, (int)s.cbLength, (int)got)); ++protectCount;
    std::memcpy(REAL(callbackResults), s.cbValues.data(), s.cbValues.size() * sizeof(double));
  }
  const bool withCb = !Rf_isNull(s.callback);
  const int len = s.keepFits ? (doStan ? 1 : 0) + (doBart ? 1 : 0) + (withCb ? 1 : 0) : 1;
  SEXP result = PROTECT(Rf_allocVector(VECSXP, len)); ++protectCount;
  SEXP names = PROTECT(Rf_allocVector(STRSXP, len)); ++protectCount;
  int pos = 0;
  if (s.keepFits && doStan) { SET_VECTOR_ELT(result, pos, stan); SET_STRING_ELT(names, pos++, Rf_mkChar("stan")); }
  if (s.keepFits && doBart) { SET_VECTOR_ELT(result, pos, bart); SET_STRING_ELT(names, pos++, Rf_mkChar("bart")); }
  if (!s.keepFits || withCb) { SET_VECTOR_ELT(result, pos, callbackResults); SET_STRING_ELT(names, pos++, Rf_mkChar("callback")); }
  Rf_setAttrib(result, R_NamesSymbol, names);
  UNPROTECT(protectCount);
  return result;
}

// stan4bart_printInitialSummary — reference src/init.cpp:971-993
static SEXP printInitialSummary(SEXP samplerExpr) { check(s4b_print_initial_summary(sampler_of(samplerExpr, "printInitialSummary").h)); return R_NilValue; }

// stan4bart_disengageAdaptation — reference src/init.cpp:995-1004
static SEXP disengageAdaptation(SEXP samplerExpr) { check(s4b_disengage_adaptation(sampler_of(samplerExpr, "disengageAdaptation").h)); return R_NilValue; }

// stan4bart_finalize() — reference src/init.cpp:1182-1211: frees whatever is still alive when the package unloads (R/hooks.R:1-11)
static SEXP finalize(void) {
  if (activeSamplers) { std::set<SEXP> live(*activeSamplers); for (SEXP p : live) sampler_finalizer(p); delete activeSamplers; activeSamplers = nullptr; }
  if (activeStored) { std::set<SEXP> live(*activeStored); for (SEXP p : live) stored_finalizer(p); delete activeStored; activeStored = nullptr; }
  return R_NilValue;
}

// stan4bart_exportBARTState(sampler) — reference src/init.cpp:409-416: here a raw vector (the library's relocatable byte string)
static SEXP exportBARTState(SEXP samplerExpr) {
  Sampler& s = sampler_of(samplerExpr, "exportBARTState");
  int64_t size = 0;
  check(s4b_export_bart_state(s.h, nullptr, 0, &size));
  SEXP raw = PROTECT(Rf_allocVector(RAWSXP, (R_xlen_t)size));
  check(s4b_export_bart_state(s.h, RAW(raw), size, &size));
  UNPROTECT(1);
  return raw;
}

// stan4bart_createStoredBARTSampler(bartControl, bartData, bartModel, states) — reference src/init.cpp:418-446.  `states` is the
// list of per-chain exported states (R/stan4bart_fit.R:572-580); control / data / model are only consulted for consistency: the
// exported state carries cut points, scales and trees
static SEXP createStoredBARTSampler(SEXP bartControlExpr, SEXP, SEXP, SEXP statesExpr) {
  if (!Rf_isNewList(statesExpr) || Rf_xlength(statesExpr) < 1) Rf_error("state must be a list with one exported state per chain");
  StoredSampler* st = new StoredSampler;
  const int device = 0;
  for (R_xlen_t c = 0; c < Rf_xlength(statesExpr); ++c) {
    SEXP raw = VECTOR_ELT(statesExpr, c);
    s4b_sampler* h = nullptr;
    if (s4b_create_stored_bart_sampler(RAW(raw), (int64_t)Rf_xlength(raw), device, &h) != 0) {
      for (s4b_sampler* g : st->chains) s4b_free(g);
      delete st;
      Rf_error("%s", s4b_last_error());
    }
    st->chains.push_back(h);
  }
  int64_t dims[5];
  check(s4b_get_dims(st->chains[0], dims));
  st->p = dims[3]; st->nTrees = dims[4];
  if (Rf_asInteger(slot(bartControlExpr, "n.trees")) != (int)st->nTrees) Rf_warning("stored state has %d trees, control says %d", (int)st->nTrees, Rf_asInteger(slot(bartControlExpr, "n.trees")));
  SEXP result = PROTECT(R_MakeExternalPtr(st, R_NilValue, R_NilValue));
  R_RegisterCFinalizerEx(result, stored_finalizer, TRUE);
  if (activeStored) activeStored->insert(result);
  UNPROTECT(1);
  return result;
}

// stan4bart_predictBART(storedSampler, x_test, offset_test) — reference src/init.cpp:354-403: [n_test x samples (x chains)]
static SEXP predictBART(SEXP storedExpr, SEXP xTestExpr, SEXP offsetTestExpr) {
  StoredSampler& st = stored_of(storedExpr, "predictBART");
  if (Rf_isNull(xTestExpr)) return R_NilValue;
  if (!Rf_isReal(xTestExpr)) Rf_error("x.test must be of type real");
  SEXP dims = Rf_getAttrib(xTestExpr, R_DimSymbol);
  if (Rf_isNull(dims) || Rf_length(dims) != 2 || INTEGER(dims)[1] != (int)st.p) Rf_error("dimensions of x_test must be n.test x %d", (int)st.p);
  const int64_t nTest = INTEGER(dims)[0];
  const double* offset = nullptr;
  if (!Rf_isNull(offsetTestExpr)) {
    if (!Rf_isReal(offsetTestExpr)) Rf_error("offset.test must be of type real");
    if (Rf_xlength(offsetTestExpr) != 1 || !ISNA(REAL(offsetTestExpr)[0])) {
      if (Rf_xlength(offsetTestExpr) != nTest) Rf_error("length of offset.test must equal number of rows in x.test");
      offset = REAL(offsetTestExpr);
    }
  }
  int64_t numSamples = 0;
  check(s4b_predict_bart_offset(st.chains[0], REAL(xTestExpr), nTest, offset, nullptr, &numSamples));
  const size_t numChains = st.chains.size();
  SEXP result = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)(nTest * numSamples * (int64_t)numChains)));
  for (size_t c = 0; c < numChains; ++c) {
    int64_t got = 0;
    check(s4b_predict_bart_offset(st.chains[c], REAL(xTestExpr), nTest, offset, REAL(result) + (size_t)c * (size_t)(nTest * numSamples), &got));
    if (got != numSamples) { UNPROTECT(1); Rf_error("chains hold different numbers of kept draws"); }
  }
  SEXP rdims = PROTECT(Rf_allocVector(INTSXP, numChains > 1 ? 3 : 2));
  INTEGER(rdims)[0] = (int)nTest; INTEGER(rdims)[1] = (int)numSamples;
  if (numChains > 1) INTEGER(rdims)[2] = (int)numChains;
  Rf_setAttrib(result, R_DimSymbol, rdims);
  UNPROTECT(2);
  return result;
}

// stan4bart_getParametricMean(sampler) — reference src/init.cpp:332-347
static SEXP getParametricMean(SEXP samplerExpr) {
  Sampler& s = sampler_of(samplerExpr, "getParametricMean");
  SEXP result = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)s.n));
  check(s4b_get_parametric_mean(s.h, REAL(result)));
  UNPROTECT(1);
  return result;
}

// stan4bart_getBARTDataRange(sampler) — reference src/init.cpp:316-330: c(min, max)
static SEXP getBARTDataRange(SEXP samplerExpr) {
  Sampler& s = sampler_of(samplerExpr, "getBARTDataRange");
  SEXP result = PROTECT(Rf_allocVector(REALSXP, 2));
  check(s4b_get_bart_data_range(s.h, REAL(result)));
  UNPROTECT(1);
  return result;
}

// 1-based R index vector (or NULL = all) -> 0-based, with the reference's bound messages (src/init.cpp:467-478)
static bool index_vector(SEXP e, size_t have, const char* what, std::vector<int32_t>& out) {
  if (Rf_isNull(e)) return false;
  out = ints(e);
  if (out.size() > have) Rf_error("%d %s specified but only %d in sampler", (int)out.size(), what, (int)have);
  for (int32_t& v : out) { if (v < 1 || (size_t)v > have) Rf_error("%s index out of range", what); v -= 1; }
  return true;
}

// stan4bart_printTrees(storedSampler, chainIndices, sampleIndices, treeIndices) — reference src/init.cpp:448-512
static SEXP printTrees(SEXP storedExpr, SEXP chainIndicesExpr, SEXP sampleIndicesExpr, SEXP treeIndicesExpr) {
  StoredSampler& st = stored_of(storedExpr, "printTrees");
  int64_t numSamples = 0;
  check(s4b_predict_bart(st.chains[0], nullptr, 0, nullptr, &numSamples));
  std::vector<int32_t> ci, si, ti;
  const bool hasC = index_vector(chainIndicesExpr, st.chains.size(), "chains", ci);
  const bool hasS = index_vector(sampleIndicesExpr, (size_t)numSamples, "samples", si);
  const bool hasT = index_vector(treeIndicesExpr, (size_t)st.nTrees, "trees", ti);
  for (size_t k = 0; k < (hasC ? ci.size() : st.chains.size()); ++k) {
    const size_t c = hasC ? (size_t)ci[k] : k;
    if (st.chains.size() > 1) Rprintf("chain %d:\n", (int)c + 1);
    check(s4b_print_trees(st.chains[c], hasS ? si.data() : nullptr, (int64_t)si.size(), hasT ? ti.data() : nullptr, (int64_t)ti.size()));
  }
  return R_NilValue;
}

// stan4bart_getTrees(storedSampler, chainIndices, sampleIndices, treeIndices, current) — reference src/init.cpp:514-671:
// data.frame(chain (several chains only), sample, tree, n, var (1-based; -1 = leaf), value)
static SEXP getTrees(SEXP storedExpr, SEXP chainIndicesExpr, SEXP sampleIndicesExpr, SEXP treeIndicesExpr, SEXP currentExpr) {
  StoredSampler& st = stored_of(storedExpr, "getTrees");
  if (Rf_asLogical(currentExpr) == TRUE) Rf_error("current = TRUE needs a live sampler: a stored sampler only holds kept trees");
  int64_t numSamples = 0;
  check(s4b_predict_bart(st.chains[0], nullptr, 0, nullptr, &numSamples));
  std::vector<int32_t> ci, si, ti;
  const bool hasC = index_vector(chainIndicesExpr, st.chains.size(), "chains", ci);
  const bool hasS = index_vector(sampleIndicesExpr, (size_t)numSamples, "samples", si);
  const bool hasT = index_vector(treeIndicesExpr, (size_t)st.nTrees, "trees", ti);
  const size_t nChainsSel = hasC ? ci.size() : st.chains.size();
  std::vector<int32_t> chain, smp, tree, nobs, var, split; std::vector<double> value;
  for (size_t k = 0; k < nChainsSel; ++k) {
    const size_t c = hasC ? (size_t)ci[k] : k;
    int64_t m = 0;
    check(s4b_get_kept_trees_indexed(st.chains[c], hasS ? si.data() : nullptr, (int64_t)si.size(), hasT ? ti.data() : nullptr, (int64_t)ti.size(), 0,
                                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &m));
    const size_t o = smp.size();
    chain.resize(o + (size_t)m, (int32_t)c + 1); smp.resize(o + (size_t)m); tree.resize(o + (size_t)m); nobs.resize(o + (size_t)m);
    var.resize(o + (size_t)m); split.resize(o + (size_t)m); value.resize(o + (size_t)m);
    check(s4b_get_kept_trees_indexed(st.chains[c], hasS ? si.data() : nullptr, (int64_t)si.size(), hasT ? ti.data() : nullptr, (int64_t)ti.size(), m,
                                     smp.data() + o, tree.data() + o, nobs.data() + o, var.data() + o, split.data() + o, value.data() + o, &m));
  }
  const R_xlen_t rows = (R_xlen_t)smp.size();
  const bool withChain = st.chains.size() > 1;
  const int cols = 5 + (withChain ? 1 : 0);
  SEXP df = PROTECT(Rf_allocVector(VECSXP, cols));
  SEXP names = PROTECT(Rf_allocVector(STRSXP, cols));
  int pos = 0;
  auto int_col = [&](const char* nm, const std::vector<int32_t>& v, int add) {
    SEXP col = Rf_allocVector(INTSXP, rows);
    SET_VECTOR_ELT(df, pos, col);
    for (R_xlen_t i = 0; i < rows; ++i) INTEGER(col)[i] = v[(size_t)i] + add;
    SET_STRING_ELT(names, pos++, Rf_mkChar(nm));
  };
  if (withChain) int_col("chain", chain, 0);
  int_col("sample", smp, 1); int_col("tree", tree, 1); int_col("n", nobs, 0);
  {
    SEXP col = Rf_allocVector(INTSXP, rows); SET_VECTOR_ELT(df, pos, col);
    for (R_xlen_t i = 0; i < rows; ++i) INTEGER(col)[i] = var[(size_t)i] >= 0 ? var[(size_t)i] + 1 : -1;
    SET_STRING_ELT(names, pos++, Rf_mkChar("var"));
  }
  { SEXP col = Rf_allocVector(REALSXP, rows); SET_VECTOR_ELT(df, pos, col); std::memcpy(REAL(col), value.data(), (size_t)rows * sizeof(double)); SET_STRING_ELT(names, pos++, Rf_mkChar("value")); }
  Rf_setAttrib(df, R_NamesSymbol, names);
  SEXP cls = PROTECT(Rf_mkString("data.frame"));
  Rf_setAttrib(df, R_ClassSymbol, cls);
  SEXP rn = PROTECT(Rf_allocVector(INTSXP, 2));          // compact row names c(NA, -rows)
  INTEGER(rn)[0] = NA_INTEGER; INTEGER(rn)[1] = -(int)rows;
  Rf_setAttrib(df, R_RowNamesSymbol, rn);
  UNPROTECT(4);
  return df;
}

#define S4B_DEF(name, fn, nargs) {name, (DL_FUNC)(void*)&fn, nargs}
// the reference's registration table, name for name and arity for arity (src/init.cpp:1215-1229)
static const R_CallMethodDef callMethods[] = {
  S4B_DEF("stan4bart_create", createSampler, 6),
  S4B_DEF("stan4bart_run", run, 4),
  S4B_DEF("stan4bart_printInitialSummary", printInitialSummary, 1),
  S4B_DEF("stan4bart_disengageAdaptation", disengageAdaptation, 1),
  S4B_DEF("stan4bart_finalize", finalize, 0),
  S4B_DEF("stan4bart_exportBARTState", exportBARTState, 1),
  S4B_DEF("stan4bart_createStoredBARTSampler", createStoredBARTSampler, 4),
  S4B_DEF("stan4bart_predictBART", predictBART, 3),
  S4B_DEF("stan4bart_getParametricMean", getParametricMean, 1),
  S4B_DEF("stan4bart_getBARTDataRange", getBARTDataRange, 1),
  S4B_DEF("stan4bart_printTrees", printTrees, 4),
  S4B_DEF("stan4bart_getTrees", getTrees, 5),
  {NULL, NULL, 0}
};
#undef S4B_DEF

void R_init_stan4bart(DllInfo* info) {
  R_registerRoutines(info, NULL, callMethods, NULL, NULL);
  R_useDynamicSymbols(info, FALSE);
  activeSamplers = new std::set<SEXP>();
  activeStored = new std::set<SEXP>();
}

}  // extern "C"
