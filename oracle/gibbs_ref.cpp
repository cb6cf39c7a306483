// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the reference's blocked Gibbs driver:
//   createSampler   reference src/init.cpp:190-310   (init order: Stan init + init_stepsize against
//                   offset_ = 0, then setOffset(updateScale) / setSigma / trees from prior / one
//                   un-recorded BART sweep / first stanOffset)
//   run             reference src/init.cpp:678-965   (loop body :752-917)
//   disengage       reference src/init.cpp:995-1004
// exported with the same C interface as the product (include/stan4bart_amd.h) under the
// `orc_` prefix, so tests drive both through one wrapper.  Single-threaded, as the reference
// (R/stan4bart_fit.R:437-439).
#define S4B_PREFIX orc_
#include "../include/stan4bart_amd.h"

#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "bart_ref.hpp"
#include "stan_ref.hpp"

using namespace oracle;

namespace {
thread_local std::string g_err;

enum { OFFSET_DEFAULT = 0, OFFSET_FIXEF, OFFSET_RANEF, OFFSET_BART, OFFSET_PARAMETRIC };
}  // namespace

struct KeptSample { std::vector<int32_t> st; std::vector<double> mu; std::vector<size_t> treeStart, leafStart; double min, range; };

struct s4b_sampler {
  bool keepTrees = false; std::vector<KeptSample> kept;
  int warmup = 0, iter = 0, verbose = 0, refresh = 0;
  bool binary = false, keepFits = true;
  int offsetType = 0;
  std::vector<double> userOffset; bool hasUserOffset = false;
  s4b_callback_fn callback = nullptr; void* callbackUser = nullptr;
  RRng rrng;
  std::unique_ptr<StanModel> model;
  std::unique_ptr<NutsSampler> nuts;
  std::unique_ptr<BartFit> bart;
  std::vector<double> bartOffset, stanOffset, bartLatents;
  std::vector<double> row;   // last written Stan sample row (x_curr)
  size_t n = 0;
  long treeUpdates = 0;
  // stored sampler (orc_create_stored_bart_sampler): what predict needs, without a fit
  bool stored = false; size_t storedP = 0; int storedT = 0; std::vector<int> storedNumCuts; std::vector<std::vector<double>> storedCuts;
};

static StanData convert(const s4b_stan_data& d) {
  StanData s;
  s.N = d.N; s.K = d.K;
  s.X.assign(d.X, d.X + (size_t)d.N * d.K);
  s.y.assign(d.y, d.y + d.N);
  s.is_binary = d.is_binary; s.has_intercept = d.has_intercept;
  s.prior_dist = d.prior_dist; s.prior_dist_for_aux = d.prior_dist_for_aux;
  if (d.K) { s.prior_scale.assign(d.prior_scale, d.prior_scale + d.K); s.prior_mean.assign(d.prior_mean, d.prior_mean + d.K);
             s.prior_df.assign(d.prior_df, d.prior_df + d.K); }
  s.prior_scale_for_aux = d.prior_scale_for_aux; s.prior_mean_for_aux = d.prior_mean_for_aux; s.prior_df_for_aux = d.prior_df_for_aux;
  s.global_prior_df = d.global_prior_df; s.global_prior_scale = d.global_prior_scale; s.slab_df = d.slab_df; s.slab_scale = d.slab_scale;
  if (d.prior_dist == 7) { if (!d.num_normals) throw std::invalid_argument("product_normal needs num_normals"); s.num_normals.assign(d.num_normals, d.num_normals + d.K); }
  s.t = d.t; s.q = d.q; s.len_theta_L = d.len_theta_L;
  if (d.t) { s.p.assign(d.p, d.p + d.t); s.l.assign(d.l, d.l + d.t); s.shape.assign(d.shape, d.shape + d.t); s.scale.assign(d.scale, d.scale + d.t); }
  if (d.len_concentration) s.concentration.assign(d.concentration, d.concentration + d.len_concentration);
  if (d.len_regularization) s.regularization.assign(d.regularization, d.regularization + d.len_regularization);
  if (d.num_non_zero) { s.w.assign(d.w, d.w + d.num_non_zero); s.v.assign(d.v, d.v + d.num_non_zero); }
  s.u.assign(d.N + 1, 0);
  if (d.u) s.u.assign(d.u, d.u + d.N + 1);
  s.has_weights = d.has_weights;
  if (d.has_weights) s.weights.assign(d.weights, d.weights + d.N);
  s.offset_.assign(d.N, 0.0);
  return s;
}

namespace {
struct Blob {
  std::vector<unsigned char> b;
  template <class T> void put(const T* p, size_t n) { const unsigned char* q = (const unsigned char*)p; b.insert(b.end(), q, q + n * sizeof(T)); }
  template <class T> void one(T v) { put(&v, 1); }
};
struct BlobIn {
  const unsigned char* p; size_t n, pos;
  template <class T> void get(T* dst, size_t k) { if (pos + k * sizeof(T) > n) throw std::invalid_argument("exported state: truncated"); if (k) std::memcpy(dst, p + pos, k * sizeof(T)); pos += k * sizeof(T); }
  template <class T> T one() { T v; get(&v, 1); return v; }
};
}

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }

int orc_create(const s4b_bart_control* bc, const s4b_bart_data* bd, const s4b_stan_data* sd,
               const s4b_stan_control* sc, const s4b_common_control* cc, const uint32_t* r_rng_state,
               s4b_sampler** out) {
  try {
    if (bc->interface_version != S4B_INTERFACE_VERSION) throw std::invalid_argument("s4b_bart_control.interface_version must be S4B_INTERFACE_VERSION");
    std::unique_ptr<s4b_sampler> sp(new s4b_sampler);
    s4b_sampler& s = *sp;
    s.warmup = cc->warmup; s.iter = cc->iter; s.verbose = cc->verbose; s.refresh = cc->refresh;
    s.binary = cc->is_binary != 0; s.keepFits = cc->keep_fits != 0; s.offsetType = cc->offset_type;
    s.callback = cc->callback; s.callbackUser = cc->callback_user;
    s.keepTrees = bc->keep_trees != 0;
    s.n = (size_t)bd->n;
    if (bd->n != sd->N) throw std::invalid_argument("bart data n != stan data N");
    if (!(cc->sigma_init > 0)) throw std::invalid_argument("sigma_init must be > 0");
    if (cc->offset) { s.userOffset.assign(cc->offset, cc->offset + s.n); s.hasUserOffset = true; }
    s.rrng.mti = (int)r_rng_state[0];
    std::memcpy(s.rrng.mt, r_rng_state + 1, 624 * sizeof(uint32_t));

    s.model.reset(new StanModel(convert(*sd)));
    StanControl ctl;
    ctl.seed = sc->seed; ctl.init_radius = sc->init_r; ctl.skip = sc->skip;
    if (ctl.skip <= 0) { ctl.skip = (2000 - s.warmup) / 1000; if (ctl.skip < 1) ctl.skip = 1; }
    ctl.adapt_gamma = sc->adapt_gamma; ctl.adapt_delta = sc->adapt_delta; ctl.adapt_kappa = sc->adapt_kappa; ctl.adapt_t0 = sc->adapt_t0;
    ctl.init_buffer = sc->adapt_init_buffer; ctl.term_buffer = sc->adapt_term_buffer; ctl.window = sc->adapt_window;
    ctl.stepsize = sc->stepsize; ctl.stepsize_jitter = sc->stepsize_jitter; ctl.max_treedepth = sc->max_treedepth;
    s.nuts.reset(new NutsSampler(*s.model, ctl, 1, s.warmup));
    s.row.assign((size_t)s.model->n_row, 0.0);

    BartConfig cfg;
    cfg.numTrees = bc->n_trees; cfg.thin = bc->n_thin > 0 ? bc->n_thin : 1; cfg.binary = s.binary;
    cfg.base = bc->base; cfg.power = bc->power; cfg.k = bc->k; cfg.nodeScale = bc->node_scale;
    if (bc->k_hyper_df > 0.0) {       // normal(k = chi(df, scale)): k is redrawn every sweep (bart_ref.hpp drawK)
      if (!(bc->k_hyper_scale > 0.0)) throw std::invalid_argument("k_hyper_scale must be positive (or +Inf)");
      if (!(bc->k > 0.0)) throw std::invalid_argument("k must be positive");
      cfg.kDf = bc->k_hyper_df; cfg.kScale = bc->k_hyper_scale;
    }
    cfg.birthOrDeathProb = bc->birth_or_death_prob; cfg.swapProb = bc->swap_prob; cfg.changeProb = bc->change_prob; cfg.birthProb = bc->birth_prob;
    if (bc->split_probs) cfg.splitProbs.assign(bc->split_probs, bc->split_probs + bd->p);
    cfg.useQuantiles = bc->use_quantiles != 0;
    std::vector<int> ncuts(bd->n_cuts, bd->n_cuts + bd->p);
    s.bart.reset(new BartFit(cfg, s.n, (size_t)bd->p, bd->x, sd->y, ncuts.data(), (size_t)bd->n_test, bd->x_test, &s.rrng));
    if (sd->has_weights) {
      if (s.binary) throw std::invalid_argument("observation weights are not available for binary responses");
      s.bart->weights.assign(sd->weights, sd->weights + s.n);
    }

    s.bartOffset.assign(s.n, 0.0); s.stanOffset.assign(s.n, 0.0);
    if (s.binary) s.bartLatents.assign(s.n, 0.0);
    const double* boi = cc->bart_offset_init;
    if (s.hasUserOffset) {
      if (s.offsetType != OFFSET_BART) {
        s.bartOffset = s.userOffset;
        if (boi && s.offsetType == OFFSET_DEFAULT) for (size_t i = 0; i < s.n; ++i) s.bartOffset[i] += boi[i];
      } else if (boi) s.bartOffset.assign(boi, boi + s.n);
    } else if (boi) s.bartOffset.assign(boi, boi + s.n);

    s.bart->setOffset(s.bartOffset.data(), true);
    if (!s.binary) s.bart->setSigma(cc->sigma_init);
    s.bart->sampleTreesFromPrior();
    BartResults first;
    s.bart->runSampler(first);
    for (size_t j = 0; j < s.n; ++j) first.train[j] -= s.bartOffset[j];
    if (s.hasUserOffset && s.offsetType == OFFSET_BART) s.stanOffset = s.userOffset;
    else {
      s.stanOffset = first.train;
      if (s.hasUserOffset && s.offsetType == OFFSET_DEFAULT) for (size_t j = 0; j < s.n; ++j) s.stanOffset[j] += s.userOffset[j];
    }
    s.model->set_offset(s.stanOffset.data());
    if (s.binary) { s.bart->getLatents(s.bartLatents.data()); s.model->set_response(s.bartLatents.data()); }
    *out = sp.release();
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

int orc_run(s4b_sampler* sp, int32_t numIter, int32_t isWarmup, int32_t resultsType, s4b_results* out) {
  if (sp->stored) { g_err = "this call needs a live sampler: a stored BART sampler only predicts"; return 1; }
  try {
    if (!sp) throw std::invalid_argument("run called on NULL sampler");
    if (numIter < 1) throw std::invalid_argument("num_iter must be >= 1");
    s4b_sampler& s = *sp;
    const size_t n = s.n, nTest = s.bart->nTest, p = s.bart->p;
    const int numPars = s.model->n_row;
    const bool doStan = resultsType == 0 || resultsType == 2, doBart = resultsType == 0 || resultsType == 1;
    size_t slot = 0;
    BartResults res;
    for (int iter = 0; iter < numIter; ++iter) {
      if (doStan) {
        s.nuts->run(s.row.data());
        const double* cons = s.row.data() + 7;
        if (!s.hasUserOffset) s.model->parametric_mean(cons, s.bartOffset.data(), true, true);
        else switch (s.offsetType) {
          case OFFSET_DEFAULT:
            s.model->parametric_mean(cons, s.bartOffset.data(), true, true);
            for (size_t j = 0; j < n; ++j) s.bartOffset[j] += s.userOffset[j];
            break;
          case OFFSET_BART: s.model->parametric_mean(cons, s.bartOffset.data(), true, true); break;
          case OFFSET_RANEF:
            s.model->parametric_mean(cons, s.bartOffset.data(), true, false);
            for (size_t j = 0; j < n; ++j) s.bartOffset[j] += s.userOffset[j];
            break;
          case OFFSET_FIXEF:
            s.model->parametric_mean(cons, s.bartOffset.data(), false, true);
            for (size_t j = 0; j < n; ++j) s.bartOffset[j] += s.userOffset[j];
            break;
          case OFFSET_PARAMETRIC: s.bartOffset = s.userOffset; break;
        }
        if (!s.binary) s.bart->setSigma(cons[s.model->aux_pos()]);
        if (out && out->stan) std::memcpy(out->stan + slot * numPars, s.row.data(), numPars * sizeof(double));
        int update_scale_mod = 1 << (8 * iter / numIter);
        s.bart->setOffset(s.bartOffset.data(), isWarmup && iter % update_scale_mod == 0);
      }
      if (doBart) {
        s.bart->runSampler(res);
        s.treeUpdates += (long)s.bart->cfg.numTrees * s.bart->cfg.thin;
        for (size_t j = 0; j < n; ++j) res.train[j] -= s.bartOffset[j];
        if (s.hasUserOffset && s.offsetType == OFFSET_BART) s.stanOffset = s.userOffset;
        else {
          s.stanOffset = res.train;
          if (s.hasUserOffset && s.offsetType == OFFSET_DEFAULT) for (size_t j = 0; j < n; ++j) s.stanOffset[j] += s.userOffset[j];
        }
        s.model->set_offset(s.stanOffset.data());
        if (s.binary) { s.bart->getLatents(s.bartLatents.data()); s.model->set_response(s.bartLatents.data()); }
        if (out) {
          if (out->bart_sigma) out->bart_sigma[slot] = res.sigma;
          if (out->bart_k) out->bart_k[slot] = res.k;
          if (out->bart_train) std::memcpy(out->bart_train + slot * n, res.train.data(), n * sizeof(double));
          if (out->bart_test && nTest) std::memcpy(out->bart_test + slot * nTest, res.test.data(), nTest * sizeof(double));
          if (out->bart_varcount) for (size_t j = 0; j < p; ++j) out->bart_varcount[slot * p + j] = (int32_t)res.varcount[j];
        }
        if (s.keepTrees && !isWarmup) {
          KeptSample ks; ks.min = s.bart->scaleMin; ks.range = s.bart->scaleRange;
          for (int t = 0; t < s.bart->cfg.numTrees; ++t) {
            ks.treeStart.push_back(ks.st.size() / 2); ks.leafStart.push_back(ks.mu.size());
            s.bart->serializeTree(t, ks.st, ks.mu);
          }
          s.kept.push_back(std::move(ks));
        }
        if (s.callback) s.callback(s.callbackUser, res.train.data(), nTest ? res.test.data() : nullptr, s.row.data(), numPars);
      }
      if (s.keepFits) ++slot;
    }
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

int orc_disengage_adaptation(s4b_sampler* s) {
  if (s->stored) { g_err = "this call needs a live sampler: a stored BART sampler only predicts"; return 1; }
  if (!s) { g_err = "disengageAdaptation called on NULL sampler"; return 1; }
  s->nuts->disengage_adaptation();
  return 0;
}

int orc_print_initial_summary(s4b_sampler* s) {
  if (!s) { g_err = "printInitialSummary called on NULL sampler"; return 1; }
  std::printf("stan4bart oracle: n = %zu, p = %zu, trees = %d, stan params = %d\n", s->n, s->bart->p, s->bart->cfg.numTrees, s->model->D);
  return 0;
}

int orc_get_parametric_mean(s4b_sampler* s, double* out) {
  if (!s) { g_err = "getParametricMean called on NULL sampler"; return 1; }
  s->model->parametric_mean(s->row.data() + 7, out, true, true);
  return 0;
}

int orc_get_bart_data_range(s4b_sampler* s, double out[2]) {
  if (!s) { g_err = "getBARTDataRange called on NULL sampler"; return 1; }
  out[0] = s->bart->scaleMin; out[1] = s->bart->scaleMax;
  return 0;
}

int orc_get_r_rng_state(s4b_sampler* s, uint32_t* st) { st[0] = (uint32_t)s->rrng.mti; std::memcpy(st + 1, s->rrng.mt, 624 * 4); return 0; }
int orc_set_r_rng_state(s4b_sampler* s, const uint32_t* st) { s->rrng.mti = (int)st[0]; std::memcpy(s->rrng.mt, st + 1, 624 * 4); return 0; }

int orc_get_dims(s4b_sampler* s, int64_t d[5]) {
  if (s->stored) { d[0] = 0; d[1] = 0; d[2] = 0; d[3] = (int64_t)s->storedP; d[4] = s->storedT; return 0; }
  d[0] = s->model->n_row; d[1] = (int64_t)s->n; d[2] = (int64_t)s->bart->nTest; d[3] = (int64_t)s->bart->p; d[4] = s->bart->cfg.numTrees;
  return 0;
}

int orc_get_stan_par_names(s4b_sampler* s, char* buf, size_t cap) {
  const StanModel& m = *s->model;
  std::string o = "lp__\naccept_stat__\nstepsize__\ntreedepth__\nn_leapfrog__\ndivergent__\nenergy__";
  auto add = [&](const char* base, int cnt) { for (int i = 1; i <= cnt; ++i) o += "\n" + std::string(base) + "." + std::to_string(i); };
  add("z_beta", m.n_z_beta); add("global", m.hs);
  for (int k = 1; k <= (m.hs ? m.dat.K : 0); ++k) for (int j = 1; j <= m.hs; ++j) o += "\nlocal." + std::to_string(j) + "." + std::to_string(k);
  add("caux", m.hs > 0 ? 1 : 0);
  for (int k = 1; k <= m.n_mix; ++k) o += "\nmix.1." + std::to_string(k);
  add("one_over_lambda", m.n_lambda);
  add("z_b", m.dat.q); add("z_T", m.len_z_T); add("rho", m.len_rho); add("zeta", m.len_conc); add("tau", m.dat.t);
  if (!m.dat.is_binary) { add("aux_unscaled", 1); add("aux", 1); }
  add("beta", m.dat.K); add("b", m.dat.q); add("theta_L", m.dat.len_theta_L);
  if (o.size() + 1 > cap) { g_err = "name buffer too small"; return 1; }
  std::memcpy(buf, o.c_str(), o.size() + 1);
  return 0;
}

int orc_get_trees(s4b_sampler* s, int64_t cap, int32_t* tree, int32_t* n_obs, int32_t* var, int32_t* split, double* value, int64_t* num_nodes) {
  if (s->stored) { g_err = "this call needs a live sampler: a stored BART sampler only predicts"; return 1; }
  int64_t cnt = 0;
  for (int t = 0; t < s->bart->cfg.numTrees; ++t) {
    std::vector<int32_t> st; std::vector<double> mu;
    s->bart->serializeTree(t, st, mu);
    // n_obs of an internal node = sum of its leaves' counts (preorder recursion)
    size_t nn = st.size() / 2; std::vector<int32_t> cntObs(nn, 0);
    {
      size_t cursor = 0;
      struct Rec { static int32_t go(const std::vector<int32_t>& st, std::vector<int32_t>& c, size_t& k) {
        size_t me = k++;
        if (st[2 * me] < 0) { c[me] = st[2 * me + 1]; return c[me]; }
        int32_t a = go(st, c, k); int32_t b = go(st, c, k); c[me] = a + b; return c[me]; } };
      Rec::go(st, cntObs, cursor);
    }
    size_t leaf = 0;
    for (size_t k = 0; k < nn; ++k, ++cnt) {
      if (cnt < cap && tree) {
        tree[cnt] = t; n_obs[cnt] = cntObs[k]; var[cnt] = st[2 * k];
        if (st[2 * k] >= 0) { split[cnt] = st[2 * k + 1]; value[cnt] = s->bart->cuts[(size_t)st[2 * k]][(size_t)st[2 * k + 1]]; }
        else { split[cnt] = -1; value[cnt] = mu[leaf]; }
      }
      if (st[2 * k] < 0) ++leaf;
    }
  }
  *num_nodes = cnt;
  return 0;
}

int orc_set_trace(s4b_sampler* s, int32_t enable) { s->bart->keepTrace = enable != 0; s->bart->trace.clear(); return 0; }
int orc_get_trace(s4b_sampler* s, int64_t cap, int32_t* out, int64_t* num) {
  int64_t m = (int64_t)s->bart->trace.size();
  *num = m;
  for (int64_t i = 0; i < m && i < cap; ++i) std::memcpy(out + 5 * i, &s->bart->trace[(size_t)i], 5 * sizeof(int32_t));
  s->bart->trace.clear();
  return 0;
}
int orc_get_leaf_assignment(s4b_sampler* s, int32_t t, int32_t* out) {
  std::vector<int32_t> a; s->bart->leafAssignment(t, a); std::memcpy(out, a.data(), a.size() * sizeof(int32_t)); return 0;
}
int orc_get_counters(s4b_sampler* s, int64_t out[3]) { out[0] = s->model->gradEvals; out[1] = s->treeUpdates; out[2] = 0; return 0; }

int orc_get_nuts_stats(s4b_sampler* s, double out[4]) {
  out[0] = (double)s->nuts->tot_transitions; out[1] = (double)s->nuts->tot_depth; out[2] = (double)s->nuts->tot_leapfrog; out[3] = (double)s->nuts->tot_divergent;
  return 0;
}
int orc_profile_sweep(s4b_sampler*, int32_t, double out[8]) { for (int i = 0; i < 8; ++i) out[i] = 0.0; return 0; }
int orc_stream_probe(int32_t, int64_t, int32_t, double out[4]) { for (int i = 0; i < 4; ++i) out[i] = 0.0; return 0; }
int orc_profile_leapfrog(s4b_sampler*, int32_t, double out[8]) { for (int i = 0; i < 8; ++i) out[i] = 0.0; return 0; }
int orc_predict_bart(s4b_sampler* s, const double* x_test, int64_t n_test, double* out, int64_t* num_samples) {
  *num_samples = (int64_t)s->kept.size();
  if (!out) return 0;
  const size_t P = s->stored ? s->storedP : s->bart->p;
  const int T = s->stored ? s->storedT : s->bart->cfg.numTrees;
  const std::vector<int>& numCuts = s->stored ? s->storedNumCuts : s->bart->numCuts;
  const std::vector<std::vector<double>>& cuts = s->stored ? s->storedCuts : s->bart->cuts;
  std::vector<uint16_t> xb(P * (size_t)n_test);
  for (size_t j = 0; j < P; ++j) for (int64_t i = 0; i < n_test; ++i) {
    int c = 0; while (c < numCuts[j] && x_test[j * (size_t)n_test + i] > cuts[j][(size_t)c]) ++c;
    xb[j * (size_t)n_test + i] = (uint16_t)c;
  }
  for (size_t k = 0; k < s->kept.size(); ++k) {
    const KeptSample& ks = s->kept[k];
    for (int64_t i = 0; i < n_test; ++i) {
      double fit = 0.0;
      for (int t = 0; t < T; ++t) {
        // walk the preorder serialisation of tree t: (var, split) pairs, leaves as (-1, count);
        // at an internal node go to the left child (next entry) or skip the left subtree
        size_t pos = ks.treeStart[(size_t)t], leaf = ks.leafStart[(size_t)t];
        while (ks.st[2 * pos] >= 0) {
          bool right = (int)xb[(size_t)ks.st[2 * pos] * (size_t)n_test + i] > ks.st[2 * pos + 1];
          ++pos;
          if (right) { int depth = 1; while (depth > 0) { if (ks.st[2 * pos] >= 0) ++depth; else { --depth; ++leaf; } ++pos; } }
        }
        fit += ks.mu[leaf];
      }
      out[(size_t)k * (size_t)n_test + i] = s->binary ? fit : (fit + 0.5) * ks.range + ks.min;
    }
  }
  return 0;
}

int orc_get_kept_trees(s4b_sampler* s, int64_t sample, int64_t cap, int32_t* smp, int32_t* tree, int32_t* n_obs, int32_t* var, int32_t* split, double* value,
                       int64_t* num_nodes) {
  const int64_t S = (int64_t)s->kept.size();
  if (sample >= S) { g_err = "sample index out of range"; return 1; }
  const int T = s->stored ? s->storedT : s->bart->cfg.numTrees;
  const std::vector<std::vector<double>>& cuts = s->stored ? s->storedCuts : s->bart->cuts;
  int64_t cnt = 0;
  for (int64_t k = sample < 0 ? 0 : sample; k < (sample < 0 ? S : sample + 1); ++k) {
    const KeptSample& ks = s->kept[(size_t)k];
    for (int t = 0; t < T; ++t) {
      // the serialisation is already preorder: (var, split) for internal nodes, (-1, count) for leaves; counts of the
      // internal nodes are the sums over their subtrees
      size_t start = ks.treeStart[(size_t)t], leaf = ks.leafStart[(size_t)t];
      size_t end = start; { int open = 1; while (open > 0) { if (ks.st[2 * end] >= 0) ++open; else --open; ++end; } }
      std::vector<int32_t> nsub(end - start, 0);
      for (size_t i = end; i-- > start;) {
        if (ks.st[2 * i] < 0) nsub[i - start] = ks.st[2 * i + 1];
        else { size_t l = i + 1; int open = 1; size_t r = l; while (open > 0) { if (ks.st[2 * r] >= 0) ++open; else --open; ++r; } nsub[i - start] = nsub[l - start] + nsub[r - start]; }
      }
      for (size_t i = start; i < end; ++i) {
        if (cnt < cap && tree) {
          smp[cnt] = (int32_t)k; tree[cnt] = t; n_obs[cnt] = nsub[i - start];
          if (ks.st[2 * i] >= 0) { var[cnt] = ks.st[2 * i]; split[cnt] = ks.st[2 * i + 1]; value[cnt] = cuts[(size_t)ks.st[2 * i]][(size_t)ks.st[2 * i + 1]]; }
          else { var[cnt] = -1; split[cnt] = -1; value[cnt] = ks.mu[leaf++]; }
        } else if (ks.st[2 * i] < 0) ++leaf;
        ++cnt;
      }
    }
  }
  *num_nodes = cnt;
  return 0;
}

// exportBARTState / createStoredBARTSampler (reference src/init.cpp:409-446): the oracle's own byte layout
int orc_export_bart_state(s4b_sampler* s, void* buf, int64_t cap, int64_t* size) {
  try {
    Blob o;
    const size_t P = s->stored ? s->storedP : s->bart->p;
    const int T = s->stored ? s->storedT : s->bart->cfg.numTrees;
    const std::vector<int>& numCuts = s->stored ? s->storedNumCuts : s->bart->numCuts;
    const std::vector<std::vector<double>>& cuts = s->stored ? s->storedCuts : s->bart->cuts;
    o.one<uint32_t>(0x5343524fu); o.one<uint64_t>(P); o.one<int32_t>(T); o.one<int32_t>(s->binary ? 1 : 0); o.one<uint64_t>(s->kept.size());
    for (size_t j = 0; j < P; ++j) { o.one<int32_t>(numCuts[j]); o.put(cuts[j].data(), (size_t)numCuts[j]); }
    for (const KeptSample& k : s->kept) {
      o.one<double>(k.min); o.one<double>(k.range);
      o.one<uint64_t>(k.st.size()); o.put(k.st.data(), k.st.size());
      o.one<uint64_t>(k.mu.size()); o.put(k.mu.data(), k.mu.size());
      for (int t = 0; t < T; ++t) { o.one<uint64_t>(k.treeStart[(size_t)t]); o.one<uint64_t>(k.leafStart[(size_t)t]); }
    }
    *size = (int64_t)o.b.size();
    if (buf && cap >= *size) std::memcpy(buf, o.b.data(), o.b.size());
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
int orc_create_stored_bart_sampler(const void* state, int64_t size, int32_t, s4b_sampler** out) {
  try {
    BlobIn in{(const unsigned char*)state, (size_t)(size < 0 ? 0 : size), 0};
    if (!state || in.one<uint32_t>() != 0x5343524fu) throw std::invalid_argument("not an exported oracle BART state");
    std::unique_ptr<s4b_sampler> s(new s4b_sampler());
    s->stored = true;
    s->storedP = (size_t)in.one<uint64_t>(); s->storedT = in.one<int32_t>(); s->binary = in.one<int32_t>() != 0;
    const size_t S = (size_t)in.one<uint64_t>();
    s->storedNumCuts.resize(s->storedP); s->storedCuts.resize(s->storedP);
    for (size_t j = 0; j < s->storedP; ++j) { s->storedNumCuts[j] = in.one<int32_t>(); s->storedCuts[j].resize((size_t)s->storedNumCuts[j]); in.get(s->storedCuts[j].data(), s->storedCuts[j].size()); }
    s->kept.resize(S);
    for (KeptSample& k : s->kept) {
      k.min = in.one<double>(); k.range = in.one<double>();
      k.st.resize((size_t)in.one<uint64_t>()); in.get(k.st.data(), k.st.size());
      k.mu.resize((size_t)in.one<uint64_t>()); in.get(k.mu.data(), k.mu.size());
      k.treeStart.resize((size_t)s->storedT); k.leafStart.resize((size_t)s->storedT);
      for (int t = 0; t < s->storedT; ++t) { k.treeStart[(size_t)t] = (size_t)in.one<uint64_t>(); k.leafStart[(size_t)t] = (size_t)in.one<uint64_t>(); }
    }
    *out = s.release();
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
// sampler state as a byte string (layout: include/stan4bart_amd.h, s4b_get_state); the oracle's own writer / reader
int orc_get_state(s4b_sampler* s, void* buf, int64_t cap, int64_t* size) {
  try {
    if (s->stored) throw std::invalid_argument("this call needs a live sampler");
    NutsSampler& ns = *s->nuts; BartFit& bf = *s->bart;
    const int D = ns.D; const size_t n = s->n;
    Blob o;
    s4b_state_header hd; std::memset(&hd, 0, sizeof(hd));
    hd.magic = S4B_STATE_MAGIC; hd.version = 1; hd.n = (int64_t)n; hd.n_trees = bf.cfg.numTrees; hd.num_unconstrained = D; hd.is_binary = s->binary ? 1 : 0; hd.p = (int32_t)bf.p;
    if (bf.cfg.kDf > 0.0) std::memcpy(&hd.reserved[0], &bf.cfg.k, 8);      // (the current value of a modeled k)
    o.one(hd);
    o.put(ns.cont_params.data(), (size_t)D); o.put(ns.inv_metric.data(), (size_t)D); o.put(ns.wf_m.data(), (size_t)D); o.put(ns.wf_m2.data(), (size_t)D);
    const double sc6[6] = {ns.nom_epsilon, ns.sa_mu, ns.sa_counter, ns.sa_s_bar, ns.sa_x_bar, ns.wf_n};
    o.put(sc6, 6);
    const double last[7] = {ns.lp_, ns.accept_stat_, ns.epsilon, (double)ns.depth_, (double)ns.n_leapfrog_, ns.divergent_ ? 1.0 : 0.0, ns.energy_};
    o.put(last, 7);
    const uint32_t win[8] = {ns.num_warmup_, ns.adapt_init_buffer_, ns.adapt_term_buffer_, ns.adapt_base_window_, ns.adapt_window_counter_, ns.adapt_next_window_,
                             ns.adapt_window_size_, ns.adapt_flag ? 1u : 0u};
    o.put(win, 8);
    const uint32_t ec[2] = {ns.rng.x1, ns.rng.x2};
    o.put(ec, 2);
    uint32_t rr[626]; rr[0] = (uint32_t)s->rrng.mti; std::memcpy(rr + 1, s->rrng.mt, 624 * 4); rr[625] = 0u;
    o.put(rr, 626);
    const double sc4[4] = {bf.scaleMin, bf.scaleMax, bf.scaleRange, s->binary ? 1.0 : bf.sigma * bf.scaleRange};
    o.put(sc4, 4);
    o.put(bf.offset.data(), n); o.put(bf.totalFits.data(), n);
    if (s->binary) o.put(bf.probitLatents.data(), n);
    for (int t = 0; t < bf.cfg.numTrees; ++t) {
      std::vector<int32_t> st; std::vector<double> mu;
      bf.serializeTree(t, st, mu);
      o.one<int32_t>((int32_t)(st.size() / 2)); o.one<int32_t>((int32_t)mu.size());
      o.put(st.data(), st.size()); o.put(mu.data(), mu.size());
    }
    *size = (int64_t)o.b.size();
    if (buf && cap >= *size) std::memcpy(buf, o.b.data(), o.b.size());
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
int orc_set_state(s4b_sampler* s, const void* buf, int64_t size) {
  try {
    if (s->stored) throw std::invalid_argument("this call needs a live sampler");
    NutsSampler& ns = *s->nuts; BartFit& bf = *s->bart;
    const int D = ns.D; const size_t n = s->n;
    BlobIn in{(const unsigned char*)buf, (size_t)(size < 0 ? 0 : size), 0};
    s4b_state_header hd = in.one<s4b_state_header>();
    if (hd.magic != S4B_STATE_MAGIC || hd.version != 1u) throw std::invalid_argument("not a stan4bart sampler state");
    if (hd.n != (int64_t)n || hd.n_trees != bf.cfg.numTrees || hd.num_unconstrained != D || (hd.is_binary != 0) != s->binary || hd.p != (int32_t)bf.p)
      throw std::invalid_argument("sampler state: dimensions do not match this sampler");
    {
      double kk; std::memcpy(&kk, &hd.reserved[0], 8);
      if (bf.cfg.kDf > 0.0) {
        if (!(kk > 0.0) || !std::isfinite(kk)) throw std::invalid_argument("sampler state: k must be positive and finite");
        bf.cfg.k = kk;
      } else if (kk != 0.0) throw std::invalid_argument("sampler state: it carries the value of a modeled k, this sampler's k is fixed");
    }
    in.get(ns.cont_params.data(), (size_t)D); in.get(ns.inv_metric.data(), (size_t)D); in.get(ns.wf_m.data(), (size_t)D); in.get(ns.wf_m2.data(), (size_t)D);
    double sc6[6]; in.get(sc6, 6);
    ns.nom_epsilon = sc6[0]; ns.sa_mu = sc6[1]; ns.sa_counter = sc6[2]; ns.sa_s_bar = sc6[3]; ns.sa_x_bar = sc6[4]; ns.wf_n = sc6[5];
    double last[7]; in.get(last, 7);
    ns.lp_ = last[0]; ns.accept_stat_ = last[1]; ns.epsilon = last[2]; ns.depth_ = (int)last[3]; ns.n_leapfrog_ = (int)last[4]; ns.divergent_ = last[5] != 0.0; ns.energy_ = last[6];
    uint32_t win[8]; in.get(win, 8);
    ns.num_warmup_ = win[0]; ns.adapt_init_buffer_ = win[1]; ns.adapt_term_buffer_ = win[2]; ns.adapt_base_window_ = win[3]; ns.adapt_window_counter_ = win[4];
    ns.adapt_next_window_ = win[5]; ns.adapt_window_size_ = win[6]; ns.adapt_flag = win[7] != 0;
    uint32_t ec[2]; in.get(ec, 2); ns.rng.x1 = ec[0]; ns.rng.x2 = ec[1];
    ns.z.q = ns.cont_params;
    uint32_t rr[626]; in.get(rr, 626);
    s->rrng.mti = (int)rr[0]; std::memcpy(s->rrng.mt, rr + 1, 624 * 4);
    double sc4[4]; in.get(sc4, 4);
    bf.scaleMin = sc4[0]; bf.scaleMax = sc4[1]; bf.scaleRange = sc4[2]; bf.sigma = s->binary ? 1.0 : sc4[3] / sc4[2];
    in.get(bf.offset.data(), n); in.get(bf.totalFits.data(), n);
    if (s->binary) in.get(bf.probitLatents.data(), n);
    bf.refreshRescaledResponse();
    for (int t = 0; t < bf.cfg.numTrees; ++t) {
      const int32_t nn = in.one<int32_t>(), nl = in.one<int32_t>();
      if (nn < 1 || nl < 1 || 2 * nl - 1 != nn) throw std::invalid_argument("sampler state: malformed tree");
      std::vector<int32_t> st((size_t)nn * 2); std::vector<double> mu((size_t)nl);
      in.get(st.data(), st.size()); in.get(mu.data(), mu.size());
      bf.importTree(t, st.data(), nn, mu.data(), nl);
    }
    // derived host state: the sample row, the offsets of both blocks, Stan's offset / response (src/init.cpp:828-847)
    s->row[0] = last[0]; s->row[1] = last[1]; s->row[2] = last[2]; s->row[3] = last[3]; s->row[4] = last[4]; s->row[5] = last[5]; s->row[6] = last[6];
    s->model->write_array(ns.cont_params, s->row.data() + 7);
    s->bartOffset = bf.offset;
    if (s->hasUserOffset && s->offsetType == OFFSET_BART) s->stanOffset = s->userOffset;
    else {
      for (size_t j = 0; j < n; ++j) s->stanOffset[j] = s->binary ? bf.totalFits[j] : (bf.totalFits[j] + 0.5) * bf.scaleRange + bf.scaleMin;
      if (s->hasUserOffset && s->offsetType == OFFSET_DEFAULT) for (size_t j = 0; j < n; ++j) s->stanOffset[j] += s->userOffset[j];
    }
    s->model->set_offset(s->stanOffset.data());
    if (s->binary) { bf.getLatents(s->bartLatents.data()); s->model->set_response(s->bartLatents.data()); }
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
void orc_free(s4b_sampler* s) { delete s; }

// ---- small extras used only by tests: direct access to the RNG restatements and the model ----
void orc_test_r_rng(uint32_t seed, int32_t n_unif, double* unif, int32_t n_norm, double* norm, int32_t n_exp, double* ex,
                    int32_t n_idx, double dn, double* idx) {
  RRng r; r.set_seed(seed);
  for (int i = 0; i < n_unif; ++i) unif[i] = r.unif_rand();
  for (int i = 0; i < n_norm; ++i) norm[i] = r.norm_rand();
  for (int i = 0; i < n_exp; ++i) ex[i] = r.exp_rand();
  for (int i = 0; i < n_idx; ++i) idx[i] = r.unif_index(dn);
}
void orc_test_r_seed_state(uint32_t seed, uint32_t* state) { RRng r; r.set_seed(seed); state[0] = (uint32_t)r.mti; std::memcpy(state + 1, r.mt, 624 * 4); }
double orc_test_qnorm(double p) { return RRng::qnorm(p); }
uint32_t orc_test_ecuyer_nth(uint32_t nth) { Ecuyer1988 e; e.x1 = 1; e.x2 = 1; uint32_t v = 0; for (uint32_t i = 0; i < nth; ++i) v = e.next(); return v; }
void orc_test_boost_draws(uint32_t seed, uint32_t chain, int32_t n_u, double* u, int32_t n_norm, double* norm) {
  Ecuyer1988 e; e.create(seed, chain);
  for (int i = 0; i < n_u; ++i) u[i] = e.uniform01();
  for (int i = 0; i < n_norm; ++i) norm[i] = boost_normal(e);
}
// log density + gradient of the Stan model at an arbitrary unconstrained point
// test hook: `count` draws of rgamma(shape, scale) from an R generator seeded like set.seed(seed) (tests/test_bart_args.py)
int orc_test_rgamma(uint32_t seed, double shape, double scale, int32_t count, double* out) {
  try {
    RRng r; r.set_seed(seed);
    for (int32_t i = 0; i < count; ++i) out[i] = r.rgamma(shape, scale);
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
int orc_test_log_prob_grad(s4b_sampler* s, const double* q, double* lp, double* grad) {
  std::vector<double> qv(q, q + s->model->D), g;
  *lp = s->model->log_prob_grad(qv, g);
  std::memcpy(grad, g.data(), g.size() * sizeof(double));
  return 0;
}
int orc_test_num_unconstrained(s4b_sampler* s) { return s->model->D; }

}  // extern "C"
