// Host orchestration of one chain of the blocked Gibbs sampler on top of a device layer.
//
// Mirrors the reference's C++ driver — `Sampler` (reference src/init.cpp:124-173), createSampler
// (:190-310) and run (:678-965, loop body :752-917) — with the two blocks re-targeted:
//   * BART block: device-resident (DevLayer::sweep); nothing O(N) crosses PCIe inside the loop.
//   * Stan block: NUTS control flow on the host (stan_host.hpp), O(N) sums on the device, by default
//     folded into sufficient statistics once per Gibbs iteration.
// `Dev` is the device layer.  The product instantiates it with the HIP layer (dev_hip.hip); a CPU
// emulation of the same interface exists only under tests/emul to exercise this file without a GPU.
#ifndef S4B_SAMPLER_CORE_HPP
#define S4B_SAMPLER_CORE_HPP

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/stan4bart_amd.h"
#include "dev_common.hpp"
#include "stan_host.hpp"

namespace s4b {

enum { OFFSET_DEFAULT = 0, OFFSET_FIXEF, OFFSET_RANEF, OFFSET_BART, OFFSET_PARAMETRIC };
enum { OBS_R = 0, OBS_OFF, OBS_LAT, OBS_Y };   // observation-length device arrays addressable through download_obs / upload_obs

// a node of a kept tree (predict): children are slot ids relative to the tree's first record
struct PackedNode { int16_t var; uint16_t cut; int16_t left, right; double mu; int32_t n; int32_t pad; };   // n: observations in the node

// everything the device layer needs at creation (host pointers, valid during init() only)
struct DevInit {
  int64_t n = 0, nTest = 0; int32_t P = 0, T = 0, nc = 256, device = 0;
  const uint16_t* xbin = nullptr;      // [P][n]
  const uint16_t* xbinTest = nullptr;  // [P][nTest]
  const int32_t* numCuts = nullptr;
  const double* y = nullptr;
  const double* userOffset = nullptr;
  const double* weights = nullptr;     // [n] or null
  ModelView model;
  int32_t K = 0, q = 0, binary = 0;
  const double* X = nullptr;           // n x K column-major
  const double* w = nullptr; const int32_t* v = nullptr; const int32_t* u = nullptr; int64_t nnz = 0;
  int32_t traceCap = 0;
};

template <class Dev>
class SamplerCore {
 public:
  // stored sampler (stan4bart_createStoredBARTSampler, reference src/init.cpp:418-446): only the exported kept trees
  SamplerCore(const void* state, int64_t size, int device) {
    stored_ = true;
    Reader r{(const unsigned char*)state, (size_t)(size < 0 ? 0 : size), 0};
    if (!state || r.u32() != STATE_MAGIC) throw std::invalid_argument("not an exported stan4bart_amd BART state");
    if (r.u32() != 1u) throw std::invalid_argument("exported BART state: unknown version");
    P_ = (int)r.u32(); T_ = (int)r.u32(); binary_ = r.u32() != 0;
    const uint64_t S = r.u64(), numNodes = r.u64();
    if (P_ < 1 || T_ < 1) throw std::invalid_argument("exported BART state: bad dimensions");
    numCuts_.resize((size_t)P_); r.get(numCuts_.data(), (size_t)P_ * 4);
    cuts_.resize((size_t)P_);
    for (int j = 0; j < P_; ++j) { if (numCuts_[(size_t)j] < 0 || numCuts_[(size_t)j] > 65534) throw std::invalid_argument("exported BART state: bad cut count");
                                   cuts_[(size_t)j].resize((size_t)numCuts_[(size_t)j]); r.get(cuts_[(size_t)j].data(), cuts_[(size_t)j].size() * 8); }
    keptScale_.resize((size_t)S * 2); r.get(keptScale_.data(), keptScale_.size() * 8);
    keptTreeStart_.resize((size_t)S * (size_t)T_); r.get(keptTreeStart_.data(), keptTreeStart_.size() * 8);
    keptNodes_.resize((size_t)numNodes); r.get(keptNodes_.data(), keptNodes_.size() * sizeof(PackedNode));
    // The byte string may come from another process (parallel.py gathers them across ranks) or from disk: nothing in it is
    // trusted.  Every tree must be a contiguous node range whose links stay inside the range and form a tree (each node reached
    // at most once from the root: no cycles, no shared children), with rules inside the cut tables — otherwise the prediction
    // kernel would read out of bounds or never terminate.
    {
      const size_t numTreesAll = keptTreeStart_.size();
      std::vector<uint8_t> seen; std::vector<int> stack;
      for (size_t k = 0; k < numTreesAll; ++k) {
        const int64_t st = keptTreeStart_[k];
        const int64_t en = k + 1 < numTreesAll ? keptTreeStart_[k + 1] : (int64_t)numNodes;
        if (st < 0 || en > (int64_t)numNodes || en <= st) throw std::invalid_argument("exported BART state: tree offset out of range");
        const int64_t count = en - st;
        if (count > 32767) throw std::invalid_argument("exported BART state: tree too large");
        const PackedNode* base = keptNodes_.data() + st;
        seen.assign((size_t)count, 0); stack.assign(1, 0);
        while (!stack.empty()) {
          const int nd = stack.back(); stack.pop_back();
          if (seen[(size_t)nd]) throw std::invalid_argument("exported BART state: node reached twice (cycle or shared child)");
          seen[(size_t)nd] = 1;
          const PackedNode& p = base[nd];
          if (p.var >= 0) {
            if (p.var >= P_) throw std::invalid_argument("exported BART state: predictor index out of range");
            if ((int)p.cut >= numCuts_[(size_t)p.var]) throw std::invalid_argument("exported BART state: cut index out of range");
            if (p.left < 0 || p.left >= count || p.right < 0 || p.right >= count) throw std::invalid_argument("exported BART state: child link out of range");
            stack.push_back(p.right); stack.push_back(p.left);
          } else if (p.var != NODE_LEAF) throw std::invalid_argument("exported BART state: unused slot linked into a tree");
        }
      }
    }
    dev_.init_stored(device, P_);
  }
  // stan4bart_exportBARTState (reference src/init.cpp:409-416): needed size; written when the buffer is large enough
  int64_t export_state(void* buf, int64_t cap) const {
    size_t need = 4 * 5 + 8 * 2 + (size_t)P_ * 4;
    for (int j = 0; j < P_; ++j) need += cuts_[(size_t)j].size() * 8;
    need += keptScale_.size() * 8 + keptTreeStart_.size() * 8 + keptNodes_.size() * sizeof(PackedNode);
    if (!buf || cap < (int64_t)need) return (int64_t)need;
    unsigned char* o = (unsigned char*)buf;
    auto put = [&](const void* src, size_t n) { if (n) std::memcpy(o, src, n); o += n; };
    const uint32_t head[5] = {STATE_MAGIC, 1u, (uint32_t)P_, (uint32_t)T_, binary_ ? 1u : 0u};
    const uint64_t cnt[2] = {(uint64_t)keptScale_.size() / 2, (uint64_t)keptNodes_.size()};
    put(head, sizeof(head)); put(cnt, sizeof(cnt)); put(numCuts_.data(), (size_t)P_ * 4);
    for (int j = 0; j < P_; ++j) put(cuts_[(size_t)j].data(), cuts_[(size_t)j].size() * 8);
    put(keptScale_.data(), keptScale_.size() * 8); put(keptTreeStart_.data(), keptTreeStart_.size() * 8);
    put(keptNodes_.data(), keptNodes_.size() * sizeof(PackedNode));
    return (int64_t)need;
  }

  SamplerCore(const s4b_bart_control* bc, const s4b_bart_data* bd, const s4b_stan_data* sd, const s4b_stan_control* sc,
              const s4b_common_control* cc, const uint32_t* rstate) {
    if (!bc || !bd || !sd || !sc || !cc || !rstate) throw std::invalid_argument("create: NULL argument");
    // (first thing read of the struct: a caller built against an older, shorter layout is refused before any newer field is touched)
    if (bc->interface_version != S4B_INTERFACE_VERSION) throw std::invalid_argument("s4b_bart_control.interface_version must be S4B_INTERFACE_VERSION (the caller was compiled against another revision of stan4bart_amd.h)");
    if (bd->n != sd->N) throw std::invalid_argument("bart data n != stan data N");
    if (bd->n < 1 || bd->p < 1) throw std::invalid_argument("bart data must have n >= 1, p >= 1");
    if (bc->n_trees < 1) throw std::invalid_argument("n_trees must be >= 1");
    if (!(cc->sigma_init > 0)) throw std::invalid_argument("sigma_init must be > 0");
    if ((cc->is_binary != 0) != (sd->is_binary != 0)) throw std::invalid_argument("common_control.is_binary and stan_data.is_binary disagree");
    binary_ = cc->is_binary != 0;
    if (sd->has_weights) {
      if (!sd->weights) throw std::invalid_argument("has_weights = 1 but weights is NULL");
      if (cc->is_binary) throw std::invalid_argument("observation weights are not available for binary responses");
      for (int64_t i = 0; i < sd->N; ++i) if (!(sd->weights[i] > 0.0)) throw std::invalid_argument("weights must be positive");
      weights_.assign(sd->weights, sd->weights + sd->N);
    }
    if (sd->has_intercept) throw std::invalid_argument("has_intercept = 1 is not supported (BART supplies the intercept)");
    if (sd->prior_dist < 0 || sd->prior_dist > 7) throw std::invalid_argument("prior_dist must be in 0..7");
    if ((sd->prior_dist == 3 || sd->prior_dist == 4) && cc->is_binary) throw std::invalid_argument("hs priors scale with the residual sd: not available for binary responses");
    if (sd->prior_dist == 7 && !sd->num_normals) throw std::invalid_argument("product_normal needs num_normals");
    n_ = (size_t)bd->n; P_ = bd->p; T_ = bc->n_trees; nTest_ = (size_t)bd->n_test;
    warmup_ = cc->warmup; verbose_ = cc->verbose; refresh_ = cc->refresh; keepFits_ = cc->keep_fits != 0; offsetType_ = cc->offset_type;
    callback_ = cc->callback; callbackUser_ = cc->callback_user;
    thin_ = bc->n_thin > 0 ? bc->n_thin : 1;
    keepTrees_ = bc->keep_trees != 0;
    nc_ = bc->node_capacity > 0 ? bc->node_capacity : 256;
    if (nc_ < 3 || nc_ > 32000) throw std::invalid_argument("node_capacity must be in [3, 32000]");
    if (cc->offset) { userOffset_.assign(cc->offset, cc->offset + n_); hasUserOffset_ = true; }
    rng_.mti = (int32_t)rstate[0]; rng_.pad = 0;
    std::memcpy(rng_.mt, rstate + 1, 624 * sizeof(uint32_t));

    // ---- Stan spec + host copies of the design (the reference copies them too: stan_sampler.cpp:197-249)
    StanSpec sp;
    sp.N = sd->N; sp.K = sd->K; sp.q = sd->q; sp.t = sd->t; sp.len_theta_L = sd->len_theta_L;
    sp.is_binary = binary_ ? 1 : 0; sp.prior_dist = sd->prior_dist; sp.prior_dist_for_aux = sd->prior_dist_for_aux;
    if (sd->K) { sp.prior_scale.assign(sd->prior_scale, sd->prior_scale + sd->K); sp.prior_mean.assign(sd->prior_mean, sd->prior_mean + sd->K);
                 sp.prior_df.assign(sd->prior_df, sd->prior_df + sd->K); }
    sp.prior_scale_for_aux = sd->prior_scale_for_aux; sp.prior_mean_for_aux = sd->prior_mean_for_aux; sp.prior_df_for_aux = sd->prior_df_for_aux;
    sp.global_prior_df = sd->global_prior_df; sp.global_prior_scale = sd->global_prior_scale; sp.slab_df = sd->slab_df; sp.slab_scale = sd->slab_scale;
    if (sd->prior_dist == 7) sp.num_normals.assign(sd->num_normals, sd->num_normals + sd->K);
    if (sd->t) { sp.p.assign(sd->p, sd->p + sd->t); sp.l.assign(sd->l, sd->l + sd->t); sp.shape.assign(sd->shape, sd->shape + sd->t);
                 sp.scale.assign(sd->scale, sd->scale + sd->t); }
    if (sd->len_concentration) sp.concentration.assign(sd->concentration, sd->concentration + sd->len_concentration);
    if (sd->len_regularization) sp.regularization.assign(sd->regularization, sd->regularization + sd->len_regularization);
    {
      int qq = 0; for (int i = 0; i < sd->t; ++i) qq += sd->p[i] * sd->l[i];
      if (qq != sd->q) throw std::invalid_argument("q != sum(p * l)");
      for (int64_t e = 0; e < sd->num_non_zero; ++e) if (sd->v[e] < 0 || sd->v[e] >= sd->q) throw std::invalid_argument("CSR column index out of range");
    }
    model_.reset(new HostModel(sp));
    // test hook (tests/test_priors.py): 1 = reverse-mode tape for every gradient, 2 = closed form AND tape, compared on every evaluation
    if (const char* e = std::getenv("S4B_GRADIENT_CHECK")) model_->gradientCheck = std::atoi(e);
    K_ = sd->K; q_ = sd->q; hmcMode_ = sc->hmc_mode;
    cX_.assign((size_t)K_, 0.0); cZ_.assign((size_t)q_, 0.0);

    // ---- BART data: cut points + binning on the host (one-off), model constants
    numCuts_.assign(bd->n_cuts, bd->n_cuts + P_);
    for (int j = 0; j < P_; ++j) if (numCuts_[(size_t)j] < 0 || numCuts_[(size_t)j] > 65534) throw std::invalid_argument("n_cuts must be in [0, 65534]");
    make_cuts(bd->x, bc->use_quantiles != 0);
    std::vector<uint16_t> xbin((size_t)P_ * n_), xbinTest((size_t)P_ * nTest_);
    bin_matrix(bd->x, n_, xbin);
    if (nTest_) bin_matrix(bd->x_test, nTest_, xbinTest);

    DevInit di;
    di.n = (int64_t)n_; di.nTest = (int64_t)nTest_; di.P = P_; di.T = T_; di.nc = nc_; di.device = cc->device;
    di.xbin = xbin.data(); di.xbinTest = nTest_ ? xbinTest.data() : nullptr; di.numCuts = numCuts_.data();
    di.y = sd->y; di.userOffset = hasUserOffset_ ? userOffset_.data() : nullptr; di.weights = weights_.empty() ? nullptr : weights_.data();
    di.model.P = P_; di.model.Pvalid = 0;
    for (int j = 0; j < P_; ++j) if (numCuts_[(size_t)j] > 0) ++di.model.Pvalid;
    if (di.model.Pvalid == 0) throw std::invalid_argument("no predictor has a cut point");
    di.model.numCuts = nullptr; di.model.scratch = nullptr; di.model.splitProbs = nullptr;
    if (bc->split_probs) {   // cgm(split.probs = ): one positive weight per predictor (reference R/stan4bart_fit.R:466-475)
      splitProbs_.assign(bc->split_probs, bc->split_probs + P_);
      for (double w : splitProbs_) if (!(w > 0.0) || !std::isfinite(w)) throw std::invalid_argument("split_probs must be positive and finite");
      di.model.splitProbs = splitProbs_.data();
    }
    di.model.base = bc->base; di.model.power = bc->power;
    di.model.pBD = bc->birth_or_death_prob; di.model.pSwap = bc->swap_prob; di.model.pChange = bc->change_prob; di.model.pBirth = bc->birth_prob;
    if (!(bc->k > 0.0) || !std::isfinite(bc->k)) throw std::invalid_argument("k must be positive and finite");
    kModeled_ = bc->k_hyper_df > 0.0; kFixed_ = bc->k;
    if (kModeled_ && !(bc->k_hyper_scale > 0.0)) throw std::invalid_argument("k_hyper_scale must be positive (or +Inf)");
    if (kModeled_ && !std::isfinite(bc->k_hyper_df)) throw std::invalid_argument("k_hyper_df must be finite");
    di.model.leafPrec = leaf_precision(bc->k, T_, bc->node_scale);
    {   // lookup tables of the tree prior, computed with the host libm (device decisions then use the same values)
      pgDepth_.resize(S4B_MAX_DEPTH); logPg_.resize(S4B_MAX_DEPTH); log1mPg_.resize(S4B_MAX_DEPTH);
      for (int d = 0; d < S4B_MAX_DEPTH; ++d) {
        pgDepth_[(size_t)d] = bc->base / std::pow(1.0 + (double)d, bc->power);
        logPg_[(size_t)d] = std::log(pgDepth_[(size_t)d]); log1mPg_[(size_t)d] = std::log(1.0 - pgDepth_[(size_t)d]);
      }
      int maxCuts = 1; for (int j = 0; j < P_; ++j) maxCuts = std::max(maxCuts, numCuts_[(size_t)j]);
      logInt_.resize((size_t)std::max(P_, maxCuts) + 2);
      for (size_t k = 0; k < logInt_.size(); ++k) logInt_[k] = std::log((double)k);
      di.model.pgDepth = pgDepth_.data(); di.model.logPg = logPg_.data(); di.model.log1mPg = log1mPg_.data();
      di.model.logInt = logInt_.data(); di.model.logIntLen = (int32_t)logInt_.size();
    }
    di.K = K_; di.q = q_; di.binary = binary_ ? 1 : 0; di.X = sd->X; di.w = sd->w; di.v = sd->v; di.u = sd->u; di.nnz = sd->num_non_zero; nnz_ = sd->num_non_zero;
    di.traceCap = 1 << 16;
    hostModelView_ = di.model; hostModelView_.numCuts = numCuts_.data();   // (table pointers are host pointers)
    dev_.init(di);
    if (kModeled_) dev_.set_k_hyper(bc->k_hyper_df, bc->k_hyper_scale, bc->node_scale, bc->k);     // normal(k = chi(df, scale)): redrawn after every sweep

    // ---- Stan sampler: init + init_stepsize run against offset_ = 0 and the raw y (reference
    //      interruptable_sampler.hpp:150,175-176 happen before any BART fit exists)
    if (hmcMode_ == 0) { build_gram(sd); haveGram_ = true; }
    dev_.stan_inputs(/*mode raw y*/ 0, false, cX_.data(), cZ_.data(), &s0_, nullptr);
    model_->lik = [this](const double* beta, const double* b, double* gX, double* gZ) { return likelihood(beta, b, gX, gZ); };
    NutsControl nc;
    nc.seed = sc->seed; nc.init_r = sc->init_r; nc.skip = sc->skip;
    if (nc.skip <= 0) { nc.skip = (2000 - warmup_) / 1000; if (nc.skip < 1) nc.skip = 1; }
    nc.gamma = sc->adapt_gamma; nc.delta = sc->adapt_delta; nc.kappa = sc->adapt_kappa; nc.t0 = sc->adapt_t0;
    nc.init_buffer = sc->adapt_init_buffer; nc.term_buffer = sc->adapt_term_buffer; nc.window = sc->adapt_window;
    nc.stepsize = sc->stepsize; nc.jitter = sc->stepsize_jitter; nc.max_depth = sc->max_treedepth;
    nuts_.reset(new Nuts(*model_, nc, 1, warmup_));
    row_.assign((size_t)(7 + model_->sp.n_constrained), 0.0);

    // ---- BART init (reference src/init.cpp:236-273)
    std::vector<double> bartOffset(n_, 0.0);
    const double* boi = cc->bart_offset_init;
    if (hasUserOffset_) {
      if (offsetType_ != OFFSET_BART) {
        bartOffset = userOffset_;
        if (boi && offsetType_ == OFFSET_DEFAULT) for (size_t i = 0; i < n_; ++i) bartOffset[i] += boi[i];
      } else if (boi) bartOffset.assign(boi, boi + n_);
    } else if (boi) bartOffset.assign(boi, boi + n_);
    dev_.offset_from_host(bartOffset.data());
    dev_.rescale(true);
    if (!binary_) { dev_.set_sigma(cc->sigma_init); sigma_ = cc->sigma_init; } else sigma_ = 1.0;
    sample_trees_from_prior();
    dev_.sweep(thin_);
    treeUpdates_ += (long)T_ * thin_;
    dev_.stan_inputs(stan_mode(), false, cX_.data(), cZ_.data(), &s0_, nullptr);
    check_device();
  }

  // stan4bart_run (reference src/init.cpp:678-965)
  void run(int numIter, bool isWarmup, int resultsType, s4b_results* out) {
    live();
    if (numIter < 1) throw std::invalid_argument("num_iter must be >= 1");
    dev_.reset_fused_scales();     // (history-independent at every call boundary: see DevHip::reset_fused_scales)
    const bool doStan = resultsType == 0 || resultsType == 2, doBart = resultsType == 0 || resultsType == 1;
    const int numPars = (int)row_.size();
    size_t slot = 0;
    std::vector<double> train, test;
    const bool wantTrain = (out && out->bart_train) || callback_;
    // keep_fits = FALSE keeps one result slot that every iteration overwrites (reference src/init.cpp:725-733): without a
    // callback only the last iteration's draw is observable, so nothing O(N) crosses PCIe before it
    const bool lastOnly = !keepFits_ && !callback_;
    if (wantTrain) train.resize(n_);
    if (nTest_) test.resize(nTest_);
    // reference src/init.cpp:745-747, 752-754: start line at verbose > 0, "iter k / n" every `refresh` iterations at verbose > 1.
    // With a progress hook (s4b_set_progress) the hook is called instead of printing, every `refresh` iterations (every
    // iteration when refresh <= 0), and may cancel the run by returning non-zero (the R shim polls R_CheckUserInterrupt there,
    // as the reference does per transition: src/stan_sampler.hpp:44-48)
    if (verbose_ > 0 && !progress_)
      std::printf("starting %s, %d draws, %s\n", isWarmup ? "warmup" : "sampling", numIter,
                  resultsType == 0 ? "both BART and Stan" : (resultsType == 1 ? "BART only" : "Stan only"));
    std::fflush(stdout);
    const bool timing = std::getenv("S4B_HOST_TIMING") != nullptr;
    double tph[4] = {0, 0, 0, 0};
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int iter = 0; iter < numIter; ++iter) {
      if (progress_) {
        if (refresh_ <= 0 || (iter + 1) % refresh_ == 0 || iter == 0)
          if (progress_(progressUser_, iter + 1, numIter, isWarmup ? 1 : 0) != 0) throw std::runtime_error("interrupted by the progress callback");
      } else if (refresh_ > 0 && verbose_ > 1 && (iter + 1) % refresh_ == 0) { std::printf("  iter %.3d / %.3d\n", iter + 1, numIter); std::fflush(stdout); }
      double t0 = timing ? now() : 0;
      if (doStan) {
        nuts_->run(row_.data());
        if (timing) { tph[0] += now() - t0; t0 = now(); }
        const double* cons = row_.data() + 7;
        const double* beta = cons + model_->sp.beta_pos();
        const double* b = cons + model_->sp.b_pos();
        if (!hasUserOffset_) dev_.offset_from_params(beta, b, 1, 1, 0);
        else switch (offsetType_) {
          case OFFSET_DEFAULT: dev_.offset_from_params(beta, b, 1, 1, 1); break;
          case OFFSET_BART: dev_.offset_from_params(beta, b, 1, 1, 0); break;
          case OFFSET_RANEF: dev_.offset_from_params(beta, b, 1, 0, 1); break;
          case OFFSET_FIXEF: dev_.offset_from_params(beta, b, 0, 1, 1); break;
          case OFFSET_PARAMETRIC: dev_.offset_from_params(beta, b, 0, 0, 1); break;
        }
        if (!binary_) { sigma_ = cons[model_->sp.aux_pos()]; dev_.set_sigma(sigma_); }
        if (out && out->stan) std::memcpy(out->stan + slot * (size_t)numPars, row_.data(), (size_t)numPars * sizeof(double));
        int update_scale_mod = 1 << (8 * iter / numIter);
        dev_.rescale(isWarmup && iter % update_scale_mod == 0);
        if (timing) { tph[1] += now() - t0; t0 = now(); }
      }
      if (doBart) {
        treeUpdates_ += (long)T_ * thin_;
        const bool emit = !lastOnly || iter == numIter - 1;
        if (nTest_ && emit && ((out && out->bart_test) || callback_)) dev_.request_test_fits();
        if (out && emit && out->bart_varcount) dev_.request_var_counts();
        dev_.sweep_and_stan_inputs(thin_, stan_mode(), wantTrain && emit, cX_.data(), cZ_.data(), &s0_, (wantTrain && emit) ? train.data() : nullptr);
        if (timing) { tph[2] += now() - t0; t0 = now(); }
        if (nTest_ && emit && ((out && out->bart_test) || callback_)) dev_.test_fits(test.data());
        if (out && emit) {
          if (out->bart_sigma) out->bart_sigma[slot] = sigma_;
          if (out->bart_k) out->bart_k[slot] = kModeled_ ? dev_.k_current() : kFixed_;
          if (out->bart_train) std::memcpy(out->bart_train + slot * n_, train.data(), n_ * sizeof(double));
          if (out->bart_test && nTest_) std::memcpy(out->bart_test + slot * nTest_, test.data(), nTest_ * sizeof(double));
          if (out->bart_varcount) dev_.var_counts(out->bart_varcount + slot * (size_t)P_);
        }
        if (keepTrees_ && !isWarmup) keep_current_trees();
        // (a non-zero return stops the run after this iteration: the reference's callback is R code whose error unwinds run(),
        // src/init.cpp:855)
        if (callback_ && callback_(callbackUser_, train.data(), nTest_ ? test.data() : nullptr, row_.data(), numPars) != 0)
          throw std::runtime_error("stopped by the per-iteration callback");
        if (timing) tph[3] += now() - t0;
      }
      if (keepFits_) ++slot;
    }
    if (timing) std::fprintf(stderr, "S4B host ms/iter: nuts %.3f  offset+rescale issue %.3f  sweep + stan inputs (wait) %.3f  results %.3f\n",
                             1e3 * tph[0] / numIter, 1e3 * tph[1] / numIter, 1e3 * tph[2] / numIter, 1e3 * tph[3] / numIter);
    check_device();
  }

  void disengage_adaptation() { live(); nuts_->disengage(); }
  void set_progress(s4b_progress_fn fn, void* user) { progress_ = fn; progressUser_ = user; }

  void parametric_mean(double* out) {
    live();
    const double* cons = row_.data() + 7;
    dev_.param_mean_to_host(cons + model_->sp.beta_pos(), cons + model_->sp.b_pos(), out);
  }
  void data_range(double out[2]) { live(); ScaleState s; dev_.get_scale(s); out[0] = s.min; out[1] = s.max; }
  void get_rng(uint32_t* st) { live(); dev_.download_rng(rng_); st[0] = (uint32_t)rng_.mti; std::memcpy(st + 1, rng_.mt, 624 * 4); }
  void set_rng(const uint32_t* st) { live(); rng_.mti = (int32_t)st[0]; rng_.pad = 0; std::memcpy(rng_.mt, st + 1, 624 * 4); dev_.upload_rng(rng_); }
  void dims(int64_t d[5]) { d[0] = (int64_t)row_.size(); d[1] = (int64_t)n_; d[2] = (int64_t)nTest_; d[3] = P_; d[4] = T_; }
  std::string par_names() const {
    live();
    const StanSpec& m = model_->sp;
    std::string o = "lp__\naccept_stat__\nstepsize__\ntreedepth__\nn_leapfrog__\ndivergent__\nenergy__";
    auto add = [&](const char* base, int cnt) { for (int i = 1; i <= cnt; ++i) o += "\n" + std::string(base) + "." + std::to_string(i); };
    add("z_beta", m.n_z_beta); add("global", m.hs);
    for (int k = 1; k <= (m.hs ? m.K : 0); ++k) for (int j = 1; j <= m.hs; ++j) o += "\nlocal." + std::to_string(j) + "." + std::to_string(k);
    add("caux", m.hs > 0 ? 1 : 0);
    for (int k = 1; k <= m.n_mix; ++k) o += "\nmix.1." + std::to_string(k);
    add("one_over_lambda", m.n_lambda);
    add("z_b", m.q); add("z_T", m.len_z_T); add("rho", m.len_rho); add("zeta", m.len_conc); add("tau", m.t);
    if (!m.is_binary) { add("aux_unscaled", 1); add("aux", 1); }
    add("beta", m.K); add("b", m.q); add("theta_L", m.len_theta_L);
    return o;
  }
  void print_summary() const {
    live();
    std::printf("stan4bart_amd sampler: n = %zu, p = %d, trees = %d, node capacity = %d, unconstrained stan params = %d, hmc mode = %s\n",
                n_, P_, T_, nc_, model_->sp.D, hmcMode_ == 0 ? "sufficient statistics" : "per-leapfrog kernels");
  }

  // flattened live trees, preorder (stan4bart_getTrees layout; reference src/init.cpp:583-665)
  int64_t get_trees(int64_t cap, int32_t* tree, int32_t* n_obs, int32_t* var, int32_t* split, double* value) {
    live();
    HostTrees h; download_trees(h);
    int64_t cnt = 0;
    for (int t = 0; t < T_; ++t) {
      TreeView tv = h.view(t, nc_);
      std::vector<int32_t> ncount((size_t)nc_, 0);
      { int nd, k; Walker<TreeView> w(tv, 0);   // post-order: counts of internal nodes
        while (w.next(nd, k)) { if (k == 0) ncount[(size_t)nd] = h.cnt[(size_t)t * nc_ + nd]; else if (k == 2) ncount[(size_t)nd] = ncount[(size_t)tv.left.get(nd)] + ncount[(size_t)tv.right.get(nd)]; } }
      int nd, k; Walker<TreeView> w(tv, 0);
      while (w.next(nd, k)) {
        if (k == 2) continue;
        if (cnt < cap && tree) {
          tree[cnt] = t; n_obs[cnt] = ncount[(size_t)nd];
          if (k == 1) { var[cnt] = tv.var.get(nd); split[cnt] = tv.cut.get(nd); value[cnt] = cuts_[(size_t)tv.var.get(nd)][(size_t)tv.cut.get(nd)]; }
          else { var[cnt] = -1; split[cnt] = -1; value[cnt] = h.mu[(size_t)t * nc_ + nd]; }
        }
        ++cnt;
      }
    }
    return cnt;
  }
  // stan4bart_getTrees(current = FALSE) over the kept draws (reference src/init.cpp:514-671; extract(fit, "trees", sampleNums = , treeNums = )):
  // 0-based index vectors select draws / trees (NULL = all, in order); same flattened layout as get_trees plus the draw index
  int64_t get_kept_trees_indexed(const int32_t* sampleIdx, int64_t numSamples, const int32_t* treeIdx, int64_t numTrees, int64_t cap, int32_t* smp,
                                 int32_t* tree, int32_t* n_obs, int32_t* var, int32_t* split, double* value) const {
    const int64_t S = (int64_t)keptScale_.size() / 2;
    if (sampleIdx && numSamples > S) throw std::invalid_argument(std::to_string(numSamples) + " samples specified but only " + std::to_string(S) + " in sampler");
    if (treeIdx && numTrees > T_) throw std::invalid_argument(std::to_string(numTrees) + " trees specified but only " + std::to_string(T_) + " in sampler");
    const int64_t nS = sampleIdx ? numSamples : S, nT = treeIdx ? numTrees : (int64_t)T_;
    int64_t cnt = 0;
    std::vector<int> stack;
    for (int64_t a = 0; a < nS; ++a) {
      const int64_t k = sampleIdx ? (int64_t)sampleIdx[a] : a;
      if (k < 0 || k >= S) throw std::invalid_argument("sample index out of range");
      for (int64_t b = 0; b < nT; ++b) {
        const int t = treeIdx ? (int)treeIdx[b] : (int)b;
        if (t < 0 || t >= T_) throw std::invalid_argument("tree index out of range");
        const PackedNode* base = keptNodes_.data() + keptTreeStart_[(size_t)(k * T_ + t)];
        stack.assign(1, 0);
        while (!stack.empty()) {   // preorder
          const int nd = stack.back(); stack.pop_back();
          const PackedNode& p = base[nd];
          if (cnt < cap && tree) {
            smp[cnt] = (int32_t)k; tree[cnt] = t; n_obs[cnt] = p.n;
            if (p.var >= 0) { var[cnt] = p.var; split[cnt] = p.cut; value[cnt] = cuts_[(size_t)p.var][(size_t)p.cut]; }
            else { var[cnt] = -1; split[cnt] = -1; value[cnt] = p.mu; }
          }
          ++cnt;
          if (p.var >= 0) { stack.push_back(p.right); stack.push_back(p.left); }
        }
      }
    }
    return cnt;
  }
  int64_t get_kept_trees(int64_t sample, int64_t cap, int32_t* smp, int32_t* tree, int32_t* n_obs, int32_t* var, int32_t* split, double* value) const {
    const int64_t S = (int64_t)keptScale_.size() / 2;
    if (sample >= S) throw std::invalid_argument("sample index out of range");
    if (sample < 0) return get_kept_trees_indexed(nullptr, 0, nullptr, 0, cap, smp, tree, n_obs, var, split, value);
    const int32_t one = (int32_t)sample;
    return get_kept_trees_indexed(&one, 1, nullptr, 0, cap, smp, tree, n_obs, var, split, value);
  }
  // stan4bart_printTrees (reference src/init.cpp:448-512 hands the indices to dbarts' printer, whose text format is not part of
  // the reference tree): one line per node, indented by depth — "var <= cut value [n]" for rules, "mu [n]" for leaves
  void print_trees(const int32_t* sampleIdx, int64_t numSamples, const int32_t* treeIdx, int64_t numTrees) const {
    const int64_t S = (int64_t)keptScale_.size() / 2;
    const int64_t nS = sampleIdx ? numSamples : S, nT = treeIdx ? numTrees : (int64_t)T_;
    if (sampleIdx && numSamples > S) throw std::invalid_argument(std::to_string(numSamples) + " samples specified but only " + std::to_string(S) + " in sampler");
    if (treeIdx && numTrees > T_) throw std::invalid_argument(std::to_string(numTrees) + " trees specified but only " + std::to_string(T_) + " in sampler");
    std::vector<std::pair<int, int>> stack;
    for (int64_t a = 0; a < nS; ++a) {
      const int64_t k = sampleIdx ? (int64_t)sampleIdx[a] : a;
      if (k < 0 || k >= S) throw std::invalid_argument("sample index out of range");
      for (int64_t b = 0; b < nT; ++b) {
        const int t = treeIdx ? (int)treeIdx[b] : (int)b;
        if (t < 0 || t >= T_) throw std::invalid_argument("tree index out of range");
        std::printf("sample %lld tree %d:\n", (long long)k + 1, t + 1);
        const PackedNode* base = keptNodes_.data() + keptTreeStart_[(size_t)(k * T_ + t)];
        stack.assign(1, std::make_pair(0, 0));
        while (!stack.empty()) {
          const int nd = stack.back().first, depth = stack.back().second; stack.pop_back();
          const PackedNode& p = base[nd];
          for (int d = 0; d < depth; ++d) std::printf("  ");
          if (p.var >= 0) { std::printf("x%d <= %g [n = %d]\n", p.var + 1, cuts_[(size_t)p.var][(size_t)p.cut], p.n); stack.push_back(std::make_pair((int)p.right, depth + 1)); stack.push_back(std::make_pair((int)p.left, depth + 1)); }
          else std::printf("mu = %g [n = %d]\n", p.mu, p.n);
        }
      }
    }
    std::fflush(stdout);
  }
  // stan4bart_predictBART over the trees kept while sampling (reference src/init.cpp:354-403)
  int64_t predict(const double* xTest, int64_t nT, double* out, const double* offsetTest = nullptr) {
    const int64_t S = (int64_t)keptScale_.size() / 2;
    if (!out) return S;
    if (nT < 1 || !xTest) throw std::invalid_argument("predict: x_test must have at least one row");
    if (S == 0) return 0;
    std::vector<uint16_t> xb((size_t)P_ * (size_t)nT);
    bin_matrix(xTest, (size_t)nT, xb);
    dev_.predict_stored(xb.data(), nT, keptNodes_.data(), keptNodes_.size(), keptTreeStart_.data(), S, T_, keptScale_.data(), binary_ ? 1 : 0, out);
    // offset_test of stan4bart_predictBART (reference src/init.cpp:377-384): added to every draw's prediction
    if (offsetTest) for (int64_t k = 0; k < S; ++k) for (int64_t i = 0; i < nT; ++i) out[(size_t)k * (size_t)nT + (size_t)i] += offsetTest[i];
    return S;
  }
  // ---- sampler state as a byte string (layout: include/stan4bart_amd.h, s4b_get_state)
  int64_t get_state(void* buf, int64_t cap) {
    live();
    const int D = model_->sp.D;
    HostTrees h; download_trees(h);
    std::vector<std::vector<int32_t>> nodes((size_t)T_); std::vector<std::vector<double>> leafMu((size_t)T_);
    size_t treeBytes = 0;
    for (int t = 0; t < T_; ++t) {
      TreeView tv = h.view(t, nc_);
      int nd, k; Walker<TreeView> w(tv, 0);
      while (w.next(nd, k)) {
        if (k == 2) continue;
        if (k == 1) { nodes[(size_t)t].push_back(tv.var.get(nd)); nodes[(size_t)t].push_back((int32_t)tv.cut.get(nd)); }
        else { nodes[(size_t)t].push_back(-1); nodes[(size_t)t].push_back(h.cnt[(size_t)t * nc_ + nd]); leafMu[(size_t)t].push_back(h.mu[(size_t)t * nc_ + nd]); }
      }
      treeBytes += 8 + nodes[(size_t)t].size() * 4 + leafMu[(size_t)t].size() * 8;
    }
    const size_t need = sizeof(s4b_state_header) + (size_t)(4 * D + 6 + 7) * 8 + (8 + 2 + 626) * 4 + 4 * 8 + (size_t)(binary_ ? 3 : 2) * n_ * 8 + treeBytes;
    if (!buf || cap < (int64_t)need) return (int64_t)need;
    unsigned char* o = (unsigned char*)buf;
    auto put = [&](const void* src, size_t k) { if (k) std::memcpy(o, src, k); o += k; };
    s4b_state_header hd; std::memset(&hd, 0, sizeof(hd));
    hd.magic = S4B_STATE_MAGIC; hd.version = 1; hd.n = (int64_t)n_; hd.n_trees = T_; hd.num_unconstrained = D; hd.is_binary = binary_ ? 1 : 0; hd.p = P_;
    if (kModeled_) { const double kk = dev_.k_current(); std::memcpy(&hd.reserved[0], &kk, 8); }
    put(&hd, sizeof(hd));
    Nuts::State ns; nuts_->get_state(ns);
    put(ns.q.data(), (size_t)D * 8); put(ns.inv_metric.data(), (size_t)D * 8); put(ns.wm.data(), (size_t)D * 8); put(ns.wm2.data(), (size_t)D * 8);
    const double sc6[6] = {ns.stepsize, ns.mu, ns.counter, ns.s_bar, ns.x_bar, ns.wn};
    put(sc6, sizeof(sc6)); put(ns.last, sizeof(ns.last));
    uint32_t win[8]; for (int i = 0; i < 7; ++i) win[i] = ns.window[i]; win[7] = (uint32_t)ns.adapting;
    put(win, sizeof(win)); put(ns.rng, sizeof(ns.rng));
    uint32_t rr[626]; get_rng(rr); rr[625] = 0u;
    put(rr, sizeof(rr));
    ScaleState scl; dev_.get_scale(scl);
    const double sc4[4] = {scl.min, scl.max, scl.range, scl.sigmaData};
    put(sc4, sizeof(sc4));
    std::vector<double> off(n_), R(n_), aux(n_);
    dev_.download_obs(OBS_OFF, off.data()); dev_.download_obs(OBS_R, R.data());
    dev_.download_obs(binary_ ? OBS_LAT : OBS_Y, aux.data());
    put(off.data(), n_ * 8);
    if (binary_) { for (size_t i = 0; i < n_; ++i) R[i] = aux[i] - R[i]; }
    else for (size_t i = 0; i < n_; ++i) R[i] = ((aux[i] - off[i] - scl.min) / scl.range - 0.5) - R[i];
    put(R.data(), n_ * 8);
    if (binary_) put(aux.data(), n_ * 8);
    for (int t = 0; t < T_; ++t) {
      const int32_t cnt[2] = {(int32_t)(nodes[(size_t)t].size() / 2), (int32_t)leafMu[(size_t)t].size()};
      put(cnt, sizeof(cnt)); put(nodes[(size_t)t].data(), nodes[(size_t)t].size() * 4); put(leafMu[(size_t)t].data(), leafMu[(size_t)t].size() * 8);
    }
    return (int64_t)need;
  }
  void set_state(const void* buf, int64_t size) {
    live();
    const int D = model_->sp.D;
    Reader r{(const unsigned char*)buf, (size_t)(size < 0 ? 0 : size), 0};
    r.what = "sampler state: truncated";
    s4b_state_header hd;
    if (!buf) throw std::invalid_argument("set_state: NULL buffer");
    r.get(&hd, sizeof(hd));
    if (hd.magic != S4B_STATE_MAGIC) throw std::invalid_argument("not a stan4bart sampler state");
    if (hd.version != 1u) throw std::invalid_argument("sampler state: unknown version");
    if (hd.n != (int64_t)n_ || hd.n_trees != T_ || hd.num_unconstrained != D || (hd.is_binary != 0) != binary_ || hd.p != P_)
      throw std::invalid_argument("sampler state: dimensions do not match this sampler");
    Nuts::State ns;
    ns.q.resize((size_t)D); ns.inv_metric.resize((size_t)D); ns.wm.resize((size_t)D); ns.wm2.resize((size_t)D);
    r.get(ns.q.data(), (size_t)D * 8); r.get(ns.inv_metric.data(), (size_t)D * 8); r.get(ns.wm.data(), (size_t)D * 8); r.get(ns.wm2.data(), (size_t)D * 8);
    double sc6[6]; r.get(sc6, sizeof(sc6)); r.get(ns.last, sizeof(ns.last));
    ns.stepsize = sc6[0]; ns.mu = sc6[1]; ns.counter = sc6[2]; ns.s_bar = sc6[3]; ns.x_bar = sc6[4]; ns.wn = sc6[5];
    uint32_t win[8]; r.get(win, sizeof(win)); r.get(ns.rng, sizeof(ns.rng));
    for (int i = 0; i < 7; ++i) ns.window[i] = win[i];
    ns.adapting = (int32_t)win[7];
    uint32_t rr[626]; r.get(rr, sizeof(rr));
    if (rr[0] > 624u) throw std::invalid_argument("sampler state: generator position out of range");
    double sc4[4]; r.get(sc4, sizeof(sc4));
    if (!(sc4[2] > 0.0) || !(sc4[3] > 0.0)) throw std::invalid_argument("sampler state: response range and sigma must be positive");
    std::vector<double> off(n_), fits(n_), lat;
    r.get(off.data(), n_ * 8); r.get(fits.data(), n_ * 8);
    if (binary_) { lat.resize(n_); r.get(lat.data(), n_ * 8); }
    // trees: preorder -> node slots in preorder (slot ids carry no meaning: every move addresses nodes by traversal order)
    HostTrees h; h.alloc(T_, nc_);
    for (int t = 0; t < T_; ++t) {
      int32_t cnt[2]; r.get(cnt, sizeof(cnt));
      if (cnt[0] < 1 || cnt[0] > nc_ || cnt[1] < 1 || 2 * cnt[1] - 1 != cnt[0]) throw std::invalid_argument("sampler state: tree does not fit node_capacity");
      std::vector<int32_t> nd((size_t)cnt[0] * 2); std::vector<double> mu((size_t)cnt[1]);
      r.get(nd.data(), nd.size() * 4); r.get(mu.data(), mu.size() * 8);
      const size_t o = (size_t)t * nc_;
      int leaf = 0; std::vector<int> open;   // parents still waiting for their right child
      for (int i = 0; i < cnt[0]; ++i) {
        int par = -1;
        if (i > 0) {
          if (open.empty()) throw std::invalid_argument("sampler state: malformed tree");
          par = open.back();
          if (h.left[o + (size_t)par] < 0) h.left[o + (size_t)par] = (int16_t)i; else { h.right[o + (size_t)par] = (int16_t)i; open.pop_back(); }
        }
        h.parent[o + (size_t)i] = (int16_t)par;
        const int32_t v = nd[(size_t)2 * i], s2 = nd[(size_t)2 * i + 1];
        if (v >= 0) {
          if (v >= P_ || s2 < 0 || s2 >= numCuts_[(size_t)v]) throw std::invalid_argument("sampler state: rule out of range");
          h.var[o + (size_t)i] = (int16_t)v; h.cut[o + (size_t)i] = (uint16_t)s2; open.push_back(i);
        } else {
          if (leaf >= cnt[1]) throw std::invalid_argument("sampler state: malformed tree");
          h.var[o + (size_t)i] = NODE_LEAF; h.mu[o + (size_t)i] = mu[(size_t)leaf++]; h.cnt[o + (size_t)i] = s2;
        }
      }
      if (!open.empty() || leaf != cnt[1]) throw std::invalid_argument("sampler state: malformed tree");
      h.hwm[(size_t)t] = cnt[0];
    }
    // nothing non-finite reaches the device or the adaptation state (the blob may come from a file or another process)
    {
      auto finite = [](const double* x, size_t k) { for (size_t i = 0; i < k; ++i) if (!std::isfinite(x[i])) return false; return true; };
      bool ok = finite(ns.q.data(), (size_t)D) && finite(ns.inv_metric.data(), (size_t)D) && finite(ns.wm.data(), (size_t)D) && finite(ns.wm2.data(), (size_t)D) &&
                finite(sc6, 6) && finite(sc4, 4) && finite(off.data(), n_) && finite(fits.data(), n_) && (!binary_ || finite(lat.data(), n_)) &&
                finite(h.mu.data(), h.mu.size());
      for (int i = 0; i < 7 && ok; ++i) ok = !std::isnan(ns.last[i]);           // (energy__ / lp__ may legitimately be infinite)
      for (int i = 0; i < D && ok; ++i) ok = ns.inv_metric[(size_t)i] > 0.0;
      if (!ok || !(sc6[0] > 0.0)) throw std::invalid_argument("sampler state: non-finite value, non-positive step size or inverse metric");
    }
    double kState = 0.0;
    std::memcpy(&kState, &hd.reserved[0], 8);
    if (kModeled_) {
      if (!(kState > 0.0) || !std::isfinite(kState)) throw std::invalid_argument("sampler state: k must be positive and finite (the state of a sampler with a fixed k carries none)");
    } else if (kState != 0.0) throw std::invalid_argument("sampler state: it carries the value of a modeled k, this sampler's k is fixed");
    // ---- commit
    if (kModeled_) dev_.set_k(kState);
    nuts_->set_state(ns);
    dev_.reset_fused_scales();
    nuts_->current_row(row_.data());
    set_rng(rr);
    sigma_ = sc4[3];
    dev_.set_scale(sc4[0], sc4[1], sc4[2], sc4[3]);
    dev_.upload_obs(OBS_OFF, off.data());
    if (binary_) { dev_.upload_obs(OBS_LAT, lat.data()); for (size_t i = 0; i < n_; ++i) fits[i] = lat[i] - fits[i]; }
    else {
      std::vector<double> y(n_); dev_.download_obs(OBS_Y, y.data());
      for (size_t i = 0; i < n_; ++i) fits[i] = ((y[i] - off[i] - sc4[0]) / sc4[2] - 0.5) - fits[i];
    }
    dev_.upload_obs(OBS_R, fits.data());
    dev_.upload_trees(h.var.data(), h.cut.data(), h.left.data(), h.right.data(), h.parent.data(), h.mu.data(), h.hwm.data());
    dev_.upload_counts(h.cnt.data());
    dev_.assign_leaves_only();
    dev_.stan_inputs(stan_mode(), false, cX_.data(), cZ_.data(), &s0_, nullptr);
    check_device();
  }

  void set_trace(bool on) { live(); dev_.set_trace(on); }
  void set_device_sharing(int chains) { live(); dev_.set_device_sharing(chains); }
  int hmc_mode() const { live(); return hmcMode_; }
  void set_hmc_mode(int mode) {
    live();
    if (mode != 0 && mode != 1) throw std::invalid_argument("hmc_mode must be 0 (sufficient statistics) or 1 (one device evaluation per leapfrog)");
    if (mode == 0 && !haveGram_) throw std::invalid_argument("this sampler was created with hmc_mode 1: it has no Gram matrix to evaluate from sufficient statistics");
    hmcMode_ = mode;
  }
  int64_t get_trace(int64_t cap, int32_t* out) { live(); return dev_.get_trace(cap, out); }
  void leaf_assignment(int t, int32_t* out) {
    live();
    if (t < 0 || t >= T_) throw std::invalid_argument("tree index out of range");
    HostTrees h; download_trees(h);
    TreeView tv = h.view(t, nc_);
    std::vector<int16_t> list((size_t)nc_); std::vector<int32_t> rank((size_t)nc_, -1);
    PtrArr<int16_t> la(list.data());
    int nl = tv_list_leaves(tv, 0, la);
    for (int i = 0; i < nl; ++i) rank[(size_t)list[(size_t)i]] = i;
    std::vector<uint16_t> leaf(n_);
    dev_.download_leaf_plane(t, leaf.data());
    for (size_t i = 0; i < n_; ++i) out[i] = rank[leaf[i]];
  }
  void profile_sweep(int nSweeps, double out[8]) { live(); if (nSweeps < 1) throw std::invalid_argument("n_sweeps must be >= 1"); dev_.profile_sweep(nSweeps, thin_, out); out[7] = (double)n_; check_device(); }
  // measurement hook: the O(N) sums of one leapfrog on the device at the current draw; out[3] = N, out[4] = SURVEY §8d's
  // algorithmic bytes per evaluation N (8K + 12z + 20), z = non-zeros of Z per row
  void profile_leapfrog(int nEvals, double out[8]) {
    live();
    if (nEvals < 1) throw std::invalid_argument("n_evals must be >= 1");
    const double* cons = row_.data() + 7;
    for (int i = 0; i < 8; ++i) out[i] = 0.0;
    dev_.profile_leapfrog(nEvals, cons + model_->sp.beta_pos(), cons + model_->sp.b_pos(), out);
    const double z = n_ ? (double)nnz_ / (double)n_ : 0.0;
    out[3] = (double)n_; out[4] = (double)n_ * (8.0 * K_ + 12.0 * z + 20.0);
    check_device();
  }
  void nuts_stats(double out[4]) { live(); nuts_->totals(out); }
  void counters(int64_t out[3]) { live(); out[0] = model_->gradEvals; out[1] = treeUpdates_; out[2] = dev_.launches(); }
  Dev& dev() { return dev_; }

 private:
  static constexpr uint32_t STATE_MAGIC = 0x54423453u;   // "S4BT"
  struct Reader {
    const unsigned char* p; size_t n, pos; const char* what = "exported BART state: truncated";
    void get(void* dst, size_t k) { if (k > n || pos > n - k) throw std::invalid_argument(what); if (k) std::memcpy(dst, p + pos, k); pos += k; }
    uint32_t u32() { uint32_t v; get(&v, 4); return v; }
    uint64_t u64() { uint64_t v; get(&v, 8); return v; }
  };
  bool stored_ = false; int64_t nnz_ = 0;
  std::vector<double> weights_;
  void live() const { if (stored_) throw std::invalid_argument("this call needs a live sampler: a stored BART sampler only predicts"); }
  struct HostTrees {
    std::vector<int16_t> var, left, right, parent, na, dep; std::vector<uint16_t> cut; std::vector<double> mu; std::vector<int32_t> cnt, hwm;
    TreeView view(int t, int nc) { size_t o = (size_t)t * nc; return make_tree_view(var.data() + o, cut.data() + o, left.data() + o, right.data() + o, parent.data() + o, nc, na.data(), dep.data()); }
    void alloc(int T, int nc) { size_t m = (size_t)T * nc; na.assign((size_t)nc, 0); dep.assign((size_t)nc, 0); var.assign(m, NODE_FREE); left.assign(m, -1); right.assign(m, -1); parent.assign(m, -1);
                                cut.assign(m, 0); mu.assign(m, 0.0); cnt.assign(m, 0); hwm.assign((size_t)T, 1); }
  };
  // one kept draw = the used node slots of every tree, packed as 16-byte records
  void keep_current_trees() {
    HostTrees h; download_trees(h);
    for (int t = 0; t < T_; ++t) {
      keptTreeStart_.push_back((int64_t)keptNodes_.size());
      const size_t o = (size_t)t * nc_;
      for (int i = 0; i < h.hwm[(size_t)t]; ++i) {
        PackedNode pn; pn.var = h.var[o + i]; pn.cut = h.cut[o + i]; pn.left = h.left[o + i]; pn.right = h.right[o + i]; pn.mu = h.mu[o + i];
        pn.n = pn.var == NODE_LEAF ? h.cnt[o + i] : 0; pn.pad = 0;
        keptNodes_.push_back(pn);
      }
      {   // observation counts of the internal nodes: children before parents
        PackedNode* base = keptNodes_.data() + keptTreeStart_.back();
        TreeView tv = h.view(t, nc_);
        int nd, k; Walker<TreeView> w(tv, 0);
        while (w.next(nd, k)) if (k == 2) base[nd].n = base[base[nd].left].n + base[base[nd].right].n;
      }
    }
    ScaleState sc; dev_.get_scale(sc);
    keptScale_.push_back(sc.min); keptScale_.push_back(sc.range);
  }
  void download_trees(HostTrees& h) { h.alloc(T_, nc_); dev_.download_trees(h.var.data(), h.cut.data(), h.left.data(), h.right.data(), h.parent.data(), h.mu.data(), h.cnt.data(), h.hwm.data()); }

  int stan_mode() const {   // how the Stan offset is formed from the BART fit (reference src/init.cpp:831-839)
    if (hasUserOffset_ && offsetType_ == OFFSET_BART) return 2;      // stanOffset = userOffset
    if (hasUserOffset_ && offsetType_ == OFFSET_DEFAULT) return 3;   // bart fit + userOffset
    return 1;                                                        // bart fit
  }

  // cut points of every predictor: uniform between the column extremes (dbartsControl useQuantiles = FALSE, the default), or
  // from the distinct values (useQuantiles = TRUE; forwarded by bart_args, reference R/stan4bart_fit.R:440-444): with at most
  // n.cuts + 1 distinct values a cut between every two neighbours, otherwise n.cuts cuts at evenly spaced ranks of the sorted
  // distinct values, each half-way between two neighbours.  (dbarts' rule restated: unpinned)  The cut COUNT of a
  // predictor may shrink below n.cuts then.
  void make_cuts(const double* x, bool quantiles) {
    cuts_.resize((size_t)P_);
    for (int j = 0; j < P_; ++j) {
      const double* col = x + (size_t)j * n_;
      int m = numCuts_[(size_t)j];
      if (quantiles) {
        std::vector<double> u(col, col + n_);
        std::sort(u.begin(), u.end());
        u.erase(std::unique(u.begin(), u.end()), u.end());
        const size_t nu = u.size();
        size_t numCuts, step, offset;
        if (m == 0) { numCuts = 0; step = 1; offset = 0; }      // n_cuts = 0 is a legal request: no rule can use the column (nu / numCuts below would divide by zero)
        else if (nu <= (size_t)m + 1) { numCuts = nu - 1; step = 1; offset = 0; }
        else { numCuts = (size_t)m; step = nu / numCuts; offset = step / 2; }
        cuts_[(size_t)j].resize(numCuts);
        for (size_t k = 0; k < numCuts; ++k) {
          const size_t idx = std::min(k * step + offset, nu - 2);
          cuts_[(size_t)j][k] = 0.5 * (u[idx] + u[idx + 1]);
        }
        numCuts_[(size_t)j] = (int32_t)numCuts;
        continue;
      }
      double mn = col[0], mx = col[0];
      for (size_t i = 1; i < n_; ++i) { if (col[i] < mn) mn = col[i]; if (col[i] > mx) mx = col[i]; }
      cuts_[(size_t)j].resize((size_t)m);
      for (int c = 0; c < m; ++c) cuts_[(size_t)j][(size_t)c] = mn + (double)(c + 1) * (mx - mn) / (double)(m + 1);
    }
  }
  void bin_matrix(const double* x, size_t m, std::vector<uint16_t>& out) const {
    auto work = [&](int j0, int j1) {
      for (int j = j0; j < j1; ++j) {
        const std::vector<double>& c = cuts_[(size_t)j];
        const double* col = x + (size_t)j * m; uint16_t* o = out.data() + (size_t)j * m;
        for (size_t i = 0; i < m; ++i) o[i] = (uint16_t)(std::lower_bound(c.begin(), c.end(), col[i]) - c.begin());   // #cuts strictly below x
      }
    };
    unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), (unsigned)P_);
    if ((size_t)P_ * m < (1u << 22)) nt = 1;
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; ++k) th.emplace_back(work, (int)((size_t)P_ * k / nt), (int)((size_t)P_ * (k + 1) / nt));
    for (auto& t : th) t.join();
  }

  // dbarts sampleTreesFromPrior (reference src/init.cpp:261): structure by recursive growth, leaf values from the prior
  void grow(TreeView& tv, int& hwm, int node) {
    tv_fill_info_node(tv, hostModelView_, node);
    double pg = tv_growth(tv, hostModelView_, node);
    if (pg <= 0.0) return;
    if (!(r_unif(&rng_) < pg)) return;
    int v = tv_draw_var(tv, hostModelView_, node, &rng_);
    int lo, hi; tv_interval(tv, hostModelView_, node, v, lo, hi);
    int s = r_unif_int(&rng_, lo, hi + 1);
    int L = tv_alloc(tv, hwm); if (L < 0) throw std::runtime_error("node capacity exceeded while sampling trees from the prior");
    tv.var.set(L, NODE_LEAF);
    int R = tv_alloc(tv, hwm); if (R < 0) throw std::runtime_error("node capacity exceeded while sampling trees from the prior");
    tv.var.set(node, (int16_t)v); tv.cut.set(node, (uint16_t)s); tv.left.set(node, (int16_t)L); tv.right.set(node, (int16_t)R);
    tv.var.set(L, NODE_LEAF); tv.parent.set(L, (int16_t)node); tv.left.set(L, -1); tv.right.set(L, -1);
    tv.var.set(R, NODE_LEAF); tv.parent.set(R, (int16_t)node); tv.left.set(R, -1); tv.right.set(R, -1);
    grow(tv, hwm, L); grow(tv, hwm, R);
  }
  void sample_trees_from_prior() {
    HostTrees h; h.alloc(T_, nc_);
    std::vector<int16_t> list((size_t)nc_);
    for (int t = 0; t < T_; ++t) {
      TreeView tv = h.view(t, nc_);
      tv.var.set(0, NODE_LEAF); tv.parent.set(0, -1);
      int hwm = 1;
      grow(tv, hwm, 0);
      PtrArr<int16_t> la(list.data());
      int nl = tv_list_leaves(tv, 0, la);
      for (int i = 0; i < nl; ++i) h.mu[(size_t)t * nc_ + list[(size_t)i]] = r_norm(&rng_) / std::sqrt(hostModelView_.leafPrec);
      h.hwm[(size_t)t] = hwm;
    }
    dev_.upload_trees(h.var.data(), h.cut.data(), h.left.data(), h.right.data(), h.parent.data(), h.mu.data(), h.hwm.data());
    dev_.upload_rng(rng_);
    dev_.assign_leaves_and_residual();
  }

  void var_counts(int32_t* out) {
    HostTrees h; download_trees(h);
    for (int j = 0; j < P_; ++j) out[j] = 0;
    for (int t = 0; t < T_; ++t) { TreeView tv = h.view(t, nc_); int nd, k; Walker<TreeView> w(tv, 0); while (w.next(nd, k)) if (k == 1) ++out[tv.var.get(nd)]; }
  }

  // G = [X Z]'[X Z]: constant over the whole run (hmc_mode 0).  Z'Z is block-sparse (levels of one grouping
  // factor never co-occur), so G is kept in CSR form: a leapfrog costs O(nnz(G)), not O((K+q)^2).
  void build_gram(const s4b_stan_data* sd) {
    const int M = K_ + q_;
    std::vector<int> idx; std::vector<double> val;
    auto row_of = [&](int64_t i) {
      idx.clear(); val.clear();
      for (int k = 0; k < K_; ++k) { idx.push_back(k); val.push_back(sd->X[(size_t)k * sd->N + i]); }
      if (q_) for (int e = sd->u[i]; e < sd->u[i + 1]; ++e) { idx.push_back(K_ + sd->v[e]); val.push_back(sd->w[e]); }
      return weights_.empty() ? 1.0 : weights_[(size_t)i];   // weighted likelihood: G = [X Z]' W [X Z]
    };
    gramPtr_.assign((size_t)M + 1, 0); gramCol_.clear(); gram_.clear();
    if ((size_t)M * (size_t)M <= ((size_t)1 << 22)) {   // small: a dense scratch matrix (32 MB at most), compressed afterwards
      std::vector<double> dense((size_t)M * M, 0.0);
      for (int64_t i = 0; i < sd->N; ++i) {
        const double wi = row_of(i);
        for (size_t a = 0; a < idx.size(); ++a) for (size_t b = 0; b < idx.size(); ++b) dense[(size_t)idx[a] * M + idx[b]] += wi * val[a] * val[b];
      }
      for (int a = 0; a < M; ++a) {
        for (int b = 0; b < M; ++b) if (dense[(size_t)a * M + b] != 0.0) { gramCol_.push_back(b); gram_.push_back(dense[(size_t)a * M + b]); }
        gramPtr_[(size_t)a + 1] = (int)gramCol_.size();
      }
      // a few dozen coefficients whose Gram matrix is mostly filled (crossed grouping factors): rows padded to a multiple of four, so that
      // the matrix-vector product of a leapfrog is straight vector code instead of indexed loads
      gramDense_.clear(); gramLd_ = 0;
      if (M <= 64 && gramCol_.size() * 3 >= (size_t)M * M) {
        gramLd_ = (M + 3) & ~3;
        gramDense_.assign((size_t)M * gramLd_, 0.0);
        for (int a = 0; a < M; ++a) for (int b = 0; b < M; ++b) gramDense_[(size_t)a * gramLd_ + b] = dense[(size_t)a * M + b];
      }
      return;
    }
    // many groups (e.g. a grouping factor with 1e5 levels): G is never formed densely.  Entries are accumulated per (row, column)
    // key in a hash map — a row of [X Z] has K + z non-zeros, so the work is N (K + z)^2 and the memory nnz(G)
    std::unordered_map<uint64_t, double> acc;
    acc.reserve((size_t)sd->N < ((size_t)1 << 22) ? (size_t)sd->N * 4 : ((size_t)1 << 24));
    for (int64_t i = 0; i < sd->N; ++i) {
      const double wi = row_of(i);
      for (size_t a = 0; a < idx.size(); ++a) for (size_t b = 0; b < idx.size(); ++b) acc[(uint64_t)idx[a] * (uint64_t)M + (uint64_t)idx[b]] += wi * val[a] * val[b];
    }
    std::vector<std::pair<uint64_t, double>> ent(acc.begin(), acc.end());
    std::sort(ent.begin(), ent.end(), [](const std::pair<uint64_t, double>& x, const std::pair<uint64_t, double>& y) { return x.first < y.first; });
    for (const auto& e : ent) {
      if (e.second == 0.0) continue;
      const int a = (int)(e.first / (uint64_t)M);
      gramCol_.push_back((int)(e.first % (uint64_t)M)); gram_.push_back(e.second);
      gramPtr_[(size_t)a + 1] = (int)gramCol_.size();
    }
    for (int a = 0; a < M; ++a) if (gramPtr_[(size_t)a + 1] < gramPtr_[(size_t)a]) gramPtr_[(size_t)a + 1] = gramPtr_[(size_t)a];
  }
  // ss = e'We, gX = X'We, gZ = Z'We with e = (y - offset) - X beta - Z b (W = I without weights)
  double likelihood(const double* beta, const double* b, double* gX, double* gZ) {
    if (hmcMode_ != 0) return dev_.leapfrog_sums(beta, b, gX, gZ);
    // (hundreds of calls per Gibbs iteration once the chain runs deep NUTS trees: theta = [beta; b] contiguous, no branch per entry;
    // same products in the same order as the two-array form)
    const int M = K_ + q_;
    if (thetaBuf_.size() < (size_t)M + 4) thetaBuf_.assign((size_t)M + 4, 0.0);
    double* const th = thetaBuf_.data();
    for (int k = 0; k < K_; ++k) th[k] = beta[k];
    for (int j = 0; j < q_; ++j) th[K_ + j] = b[j];
    if (gramLd_) {
      const double* const G = gramDense_.data(); const int ld = gramLd_;
      double ss = s0_;
      for (int a = 0; a < M; ++a) {
        const double* const r = G + (size_t)a * ld;
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
        for (int c = 0; c < ld; c += 4) { p0 += r[c] * th[c]; p1 += r[c + 1] * th[c + 1]; p2 += r[c + 2] * th[c + 2]; p3 += r[c + 3] * th[c + 3]; }
        const double ga = (p0 + p2) + (p1 + p3);
        const double ca = a < K_ ? cX_[(size_t)a] : cZ_[(size_t)(a - K_)];
        ss += th[a] * (ga - 2.0 * ca);
        if (a < K_) gX[a] = ca - ga; else gZ[a - K_] = ca - ga;
      }
      return ss;
    }
    const int* const ptr = gramPtr_.data(); const int* const col = gramCol_.data(); const double* const g = gram_.data();
    double ss = s0_;
    for (int a = 0; a < M; ++a) {
      double ga = 0.0;
      for (int e = ptr[a]; e < ptr[a + 1]; ++e) ga += g[e] * th[col[e]];
      const double ca = a < K_ ? cX_[(size_t)a] : cZ_[(size_t)(a - K_)];
      ss += th[a] * (ga - 2.0 * ca);
      if (a < K_) gX[a] = ca - ga; else gZ[a - K_] = ca - ga;
    }
    return ss;
  }
  void check_device() {
    int32_t e = dev_.error_flags();
    if (e & S4B_ERR_NODE_CAPACITY) throw std::runtime_error("a tree outgrew node_capacity; re-create the sampler with a larger bart_control.node_capacity");
    if (e & S4B_ERR_INTERNAL) throw std::runtime_error("internal error: a hand-shake of the control kernel timed out (device error word " + std::to_string(e) + ")");
    if (e & S4B_ERR_TRACE_OVERFLOW) throw std::runtime_error("trace buffer overflow: call get_trace more often");
  }

  Dev dev_;
  size_t n_ = 0, nTest_ = 0; int P_ = 0, T_ = 0, nc_ = 256, thin_ = 1, K_ = 0, q_ = 0, hmcMode_ = 0; bool haveGram_ = false;
  std::vector<double> thetaBuf_;
  int warmup_ = 0, verbose_ = 0, refresh_ = 0, offsetType_ = 0; s4b_progress_fn progress_ = nullptr; void* progressUser_ = nullptr; bool keepFits_ = true, hasUserOffset_ = false, binary_ = false;
  std::vector<double> userOffset_;
  s4b_callback_fn callback_ = nullptr; void* callbackUser_ = nullptr;
  MTState rng_;
  std::vector<int32_t> numCuts_; std::vector<std::vector<double>> cuts_; std::vector<double> splitProbs_;
  std::vector<double> pgDepth_, logPg_, log1mPg_, logInt_;
  ModelView hostModelView_;
  std::unique_ptr<HostModel> model_; std::unique_ptr<Nuts> nuts_;
  bool keepTrees_ = false, kModeled_ = false; double kFixed_ = 2.0;
  std::vector<PackedNode> keptNodes_; std::vector<int64_t> keptTreeStart_; std::vector<double> keptScale_;
  std::vector<double> row_, cX_, cZ_, gram_, gramDense_; int gramLd_ = 0; std::vector<int> gramPtr_, gramCol_; double s0_ = 0, sigma_ = 1;
  long treeUpdates_ = 0;
};

}  // namespace s4b
#endif
