// TEST INFRASTRUCTURE ONLY — CPU emulation of the device layer interface used by
// stan4bart_amd/csrc/sampler_core.hpp.  It lets the `-m "not gpu"` tests exercise the product's host
// logic (Gibbs order, NUTS, tape gradient, propose/decide control code, C-ABI) in a container without a
// GPU.  It is never linked into the product library: the product's only device layer is dev_hip.hip,
// and the product fails loudly when no GPU is present.  Kernels are restated here as plain loops with
// the same per-observation arithmetic as the HIP kernels (reduction ORDER differs, as it does on the GPU).
#ifndef S4B_DEV_CPU_HPP
#define S4B_DEV_CPU_HPP

#include <cstring>
#include <vector>
#include "../../stan4bart_amd/csrc/sampler_core.hpp"

namespace s4b {

class DevCpu {
 public:
  static void probe_stream(int, int64_t, int, double out[4]) { for (int i = 0; i < 4; ++i) out[i] = 0.0; }
  void bind() {}
  void init_stored(int, int P) { P_ = P; }
  void init(const DevInit& d) {
    di_ = d;
    n_ = (size_t)d.n; nTest_ = (size_t)d.nTest; P_ = d.P; T_ = d.T; nc_ = d.nc; K_ = d.K; q_ = d.q; binary_ = d.binary != 0;
    xbin_.assign(d.xbin, d.xbin + (size_t)P_ * n_);
    if (nTest_) xbinTest_.assign(d.xbinTest, d.xbinTest + (size_t)P_ * nTest_);
    numCuts_.assign(d.numCuts, d.numCuts + P_);
    y_.assign(d.y, d.y + n_);
    if (d.userOffset) user_.assign(d.userOffset, d.userOffset + n_);
    X_.assign(d.X, d.X + n_ * (size_t)K_);
    if (d.nnz) { w_.assign(d.w, d.w + d.nnz); v_.assign(d.v, d.v + d.nnz); }
    u_.assign(n_ + 1, 0); if (d.u) u_.assign(d.u, d.u + n_ + 1);
    R_.assign(n_, 0.0); off_.assign(n_, 0.0); offNew_.assign(n_, 0.0); e0_.assign(n_, 0.0);
    leaf_.assign((size_t)T_ * n_, 0);
    size_t m = (size_t)T_ * nc_;
    var_.assign(m, NODE_FREE); left_.assign(m, -1); right_.assign(m, -1); parent_.assign(m, -1); cut_.assign(m, 0); mu_.assign(m, 0.0); cnt_.assign(m, 0);
    hwm_.assign((size_t)T_, 1);
    for (int t = 0; t < T_; ++t) var_[(size_t)t * nc_] = NODE_LEAF;
    cna_.assign(m, 0); cdep_.assign(m, 0); cleaf_.assign(m, 0); cpre_.assign(m, 0); cpost_.assign(m, 0);
    cnl_.assign((size_t)T_, 0); cni_.assign((size_t)T_, 0); cg_.assign((size_t)T_, 0); cgn_.assign((size_t)T_, 0); cvalid_.assign((size_t)T_, 0); clogpi_.assign((size_t)T_, 0.0);
    for (int s = 0; s < 2; ++s) {
      sc_[s].pvar.assign((size_t)nc_, 0); sc_[s].pleft.assign((size_t)nc_, 0); sc_[s].pright.assign((size_t)nc_, 0); sc_[s].pparent.assign((size_t)nc_, 0);
      sc_[s].pcut.assign((size_t)nc_, 0); sc_[s].binA.assign((size_t)nc_, 0); sc_[s].binB.assign((size_t)nc_, 0); sc_[s].list.assign((size_t)nc_, 0);
      sc_[s].insub.assign((size_t)nc_, 0); sc_[s].muOld.assign((size_t)nc_, 0.0);
      sc_[s].pna.assign((size_t)nc_, 0); sc_[s].pdep.assign((size_t)nc_, 0);
      sc_[s].work.assign((size_t)14 * nc_, 0.0);
    }
    binCnt_.assign((size_t)2 * nc_, 0.0); binSum_.assign((size_t)2 * nc_, 0.0); binWt_.assign((size_t)2 * nc_, 0.0);
    if (d.weights) wts_.assign(d.weights, d.weights + n_);
    trace_.assign((size_t)d.traceCap, StepRecord{});
    // arrays view
    a_ = BartArrays{};
    a_.n = d.n; a_.npad = d.n; a_.P = P_; a_.T = T_; a_.nc = nc_; a_.grid = 1; a_.binCap = 2 * nc_; a_.traceCap = d.traceCap;
    a_.nTest = d.nTest; a_.nTestPad = d.nTest;
    a_.xbin = xbin_.data(); a_.xbinTest = xbinTest_.data(); a_.y = y_.data(); a_.R = R_.data(); a_.off = off_.data(); a_.offNew = offNew_.data();
    a_.userOffset = user_.empty() ? nullptr : user_.data(); a_.leaf = leaf_.data();
    a_.var = var_.data(); a_.left = left_.data(); a_.right = right_.data(); a_.parent = parent_.data(); a_.cut = cut_.data(); a_.mu = mu_.data();
    a_.cnt = cnt_.data(); a_.hwm = hwm_.data();
    a_.cna = cna_.data(); a_.cdep = cdep_.data(); a_.cleaf = cleaf_.data(); a_.cpre = cpre_.data(); a_.cpost = cpost_.data();
    a_.cnl = cnl_.data(); a_.cni = cni_.data(); a_.cg = cg_.data(); a_.cgn = cgn_.data(); a_.cvalid = cvalid_.data(); a_.clogpi = clogpi_.data();
    for (int s = 0; s < 2; ++s) {
      StepScratch& c = a_.sc[s];
      c.pvar = sc_[s].pvar.data(); c.pleft = sc_[s].pleft.data(); c.pright = sc_[s].pright.data(); c.pparent = sc_[s].pparent.data(); c.pcut = sc_[s].pcut.data();
      c.binA = sc_[s].binA.data(); c.binB = sc_[s].binB.data(); c.list = sc_[s].list.data(); c.insub = sc_[s].insub.data(); c.muOld = sc_[s].muOld.data();
      c.prop = &sc_[s].prop; c.accepted = &sc_[s].accepted;
      c.pna = sc_[s].pna.data(); c.pdep = sc_[s].pdep.data(); c.work = sc_[s].work.data();
    }
    a_.partCnt = nullptr; a_.partSum = nullptr; a_.binCnt = binCnt_.data(); a_.binSum = binSum_.data();
    a_.wts = wts_.empty() ? nullptr : wts_.data(); a_.partWt = nullptr; a_.binWt = binWt_.data();
    a_.rng = &rng_; a_.scale = &scale_; a_.numCuts = numCuts_.data();
    a_.trace = trace_.data(); a_.traceCount = &traceCount_; a_.errFlag = &err_;
    a_.model = d.model; a_.model.numCuts = numCuts_.data(); a_.traceOn = 0;
    // initial scale from the raw response, offset 0; R = yRescaled (all tree fits are zero)
    double mn = y_[0], mx = y_[0];
    for (size_t i = 1; i < n_; ++i) { if (y_[i] < mn) mn = y_[i]; if (y_[i] > mx) mx = y_[i]; }
    scale_ = ScaleState{}; scale_.min = mn; scale_.max = mx; scale_.range = mx - mn; scale_.min0 = mn; scale_.range0 = mx - mn; scale_.sigmaData = 1.0; scale_.sigma = 1.0 / (mx - mn);
    for (size_t i = 0; i < n_; ++i) R_[i] = (y_[i] - mn) / scale_.range - 0.5;
    if (binary_) {   // probit: no rescaling, sigma = 1, latents start at 2y - 1 (stored with the offset removed)
      scale_.min = -0.5; scale_.max = 0.5; scale_.range = 1.0; scale_.min0 = -0.5; scale_.range0 = 1.0; scale_.sigmaData = 1.0; scale_.sigma = 1.0;
      lat_.assign(n_, 0.0);
      for (size_t i = 0; i < n_; ++i) { lat_[i] = 2.0 * y_[i] - 1.0; R_[i] = lat_[i]; }
    }
  }

  void upload_rng(const MTState& s) { rng_ = s; }
  void download_rng(MTState& s) { s = rng_; }
  void upload_trees(const int16_t* var, const uint16_t* cut, const int16_t* left, const int16_t* right, const int16_t* parent, const double* mu, const int32_t* hwm) {
    size_t m = (size_t)T_ * nc_;
    std::memcpy(var_.data(), var, m * 2); std::memcpy(cut_.data(), cut, m * 2); std::memcpy(left_.data(), left, m * 2); std::memcpy(right_.data(), right, m * 2);
    std::memcpy(parent_.data(), parent, m * 2); std::memcpy(mu_.data(), mu, m * 8); std::memcpy(hwm_.data(), hwm, (size_t)T_ * 4);
    std::fill(cvalid_.begin(), cvalid_.end(), 0);
  }
  void download_trees(int16_t* var, uint16_t* cut, int16_t* left, int16_t* right, int16_t* parent, double* mu, int32_t* cnt, int32_t* hwm) {
    size_t m = (size_t)T_ * nc_;
    std::memcpy(var, var_.data(), m * 2); std::memcpy(cut, cut_.data(), m * 2); std::memcpy(left, left_.data(), m * 2); std::memcpy(right, right_.data(), m * 2);
    std::memcpy(parent, parent_.data(), m * 2); std::memcpy(mu, mu_.data(), m * 8); std::memcpy(cnt, cnt_.data(), m * 4); std::memcpy(hwm, hwm_.data(), (size_t)T_ * 4);
  }
  void download_leaf_plane(int t, uint16_t* out) { std::memcpy(out, &leaf_[(size_t)t * n_], n_ * 2); }
  void get_scale(ScaleState& s) { s = scale_; }
  int32_t error_flags() { return err_; }
  int64_t launches() const { return launches_; }
  void set_trace(bool on) { a_.traceOn = on ? 1 : 0; traceCount_ = 0; }
  int64_t get_trace(int64_t cap, int32_t* out) {
    int64_t m = traceCount_;
    for (int64_t i = 0; i < m && i < cap; ++i) std::memcpy(out + 5 * i, &trace_[(size_t)i], 20);
    traceCount_ = 0;
    return m;
  }

  // ---- offsets and response rescaling (dbarts setOffset / setSigma)
  void offset_from_host(const double* off) { std::memcpy(offNew_.data(), off, n_ * 8); }
  void offset_from_params(const double* beta, const double* b, int fixed, int random, int addUser) {
    for (size_t i = 0; i < n_; ++i) {
      double eta = 0.0;
      if (fixed) for (int k = 0; k < K_; ++k) eta += X_[(size_t)k * n_ + i] * beta[k];
      if (random && q_) for (int e = u_[i]; e < u_[i + 1]; ++e) eta += w_[(size_t)e] * b[v_[(size_t)e]];
      if (addUser) eta += user_[i];
      offNew_[i] = eta;
    }
  }
  void param_mean_to_host(const double* beta, const double* b, double* out) {
    for (size_t i = 0; i < n_; ++i) {
      double eta = 0.0;
      for (int k = 0; k < K_; ++k) eta += X_[(size_t)k * n_ + i] * beta[k];
      if (q_) for (int e = u_[i]; e < u_[i + 1]; ++e) eta += w_[(size_t)e] * b[v_[(size_t)e]];
      out[i] = eta;
    }
  }
  void set_sigma(double s) { scale_.sigmaData = s; scale_.sigma = s / scale_.range; }
  void get_latents(double* out) { for (size_t i = 0; i < n_; ++i) out[i] = lat_[i] + off_[i]; }
  void rescale(bool update) {
    if (binary_) {   // keep latent + offset invariant
      for (size_t i = 0; i < n_; ++i) { double nl = lat_[i] + (off_[i] - offNew_[i]); R_[i] += nl - lat_[i]; lat_[i] = nl; }
      off_.swap(offNew_); a_.off = off_.data(); a_.offNew = offNew_.data();
      return;
    }
    ScaleState& s = scale_;
    s.min0 = s.min; s.range0 = s.range; s.shiftPerTree = 0.0;
    if (update) {
      double mn = y_[0] - offNew_[0], mx = mn;
      for (size_t i = 1; i < n_; ++i) { double v = y_[i] - offNew_[i]; if (v < mn) mn = v; if (v > mx) mx = v; }
      s.min = mn; s.max = mx; s.range = mx - mn;
      s.shiftPerTree = (s.min0 + 0.5 * s.range0 - s.min - 0.5 * s.range) / (double)T_;
      s.sigma = s.sigmaData / s.range;
      for (size_t k = 0; k < mu_.size(); ++k) mu_[k] = (s.range0 * mu_[k] + s.shiftPerTree) / s.range;
    }
    for (size_t i = 0; i < n_; ++i) {
      double yOld = (y_[i] - off_[i] - s.min0) / s.range0 - 0.5;
      double F = yOld - R_[i];
      if (update) F = (s.range0 * F + (double)T_ * s.shiftPerTree) / s.range;
      double yNew = (y_[i] - offNew_[i] - s.min) / s.range - 0.5;
      R_[i] = yNew - F;
    }
    off_.swap(offNew_);
    a_.off = off_.data(); a_.offNew = offNew_.data();
  }

  // ---- trees
  void assign_leaves_and_residual() {
    for (size_t i = 0; i < n_; ++i) R_[i] = binary_ ? lat_[i] : (y_[i] - off_[i] - scale_.min) / scale_.range - 0.5;
    for (int t = 0; t < T_; ++t) {
      TreeView tv = tree_view(a_, t);
      for (size_t i = 0; i < n_; ++i) {
        int nd = 0;
        while (tv.var.get(nd) >= 0) nd = (xbin_[(size_t)tv.var.get(nd) * n_ + i] <= tv.cut.get(nd)) ? tv.left.get(nd) : tv.right.get(nd);
        leaf_[(size_t)t * n_ + i] = (uint16_t)nd;
        R_[i] -= mu_[(size_t)t * nc_ + nd];
      }
    }
  }
  void assign_leaves_only() {
    for (int t = 0; t < T_; ++t) {
      TreeView tv = tree_view(a_, t);
      for (size_t i = 0; i < n_; ++i) {
        int nd = 0;
        while (tv.var.get(nd) >= 0) nd = (xbin_[(size_t)tv.var.get(nd) * n_ + i] <= tv.cut.get(nd)) ? tv.left.get(nd) : tv.right.get(nd);
        leaf_[(size_t)t * n_ + i] = (uint16_t)nd;
      }
    }
  }
  std::vector<double>& obs_array(int which) {
    switch (which) {
      case OBS_R: return R_;
      case OBS_OFF: return off_;
      case OBS_LAT: if (!binary_) throw std::invalid_argument("latents exist only for binary responses"); return lat_;
      case OBS_Y: return y_;
    }
    throw std::invalid_argument("unknown observation array");
  }
  void download_obs(int which, double* out) { std::memcpy(out, obs_array(which).data(), n_ * 8); }
  void upload_obs(int which, const double* in) {
    if (which == OBS_Y) throw std::invalid_argument("the response is fixed at creation");
    std::memcpy(obs_array(which).data(), in, n_ * 8);
  }
  void upload_counts(const int32_t* cnt) { std::memcpy(cnt_.data(), cnt, cnt_.size() * 4); }
  void set_scale(double mn, double mx, double range, double sigmaData) {
    scale_ = ScaleState{}; scale_.min = mn; scale_.max = mx; scale_.range = range; scale_.min0 = mn; scale_.range0 = range; scale_.sigmaData = sigmaData; scale_.sigma = sigmaData / range;
  }
  void sweep(int thin) {
    for (int k = 0; k < thin; ++k) {
      propose_step(a_, 0); ++launches_;
      for (int t = 0; t < T_; ++t) {
        stats(t); ++launches_;
        control_step(a_, t, t + 1 < T_ ? t + 1 : -1); ++launches_;
        apply(t); ++launches_;
      }
      if (binary_) sample_latents();
      if (kModeled_) draw_k();
    }
  }
  // ---- k hyperprior (dev_common.hpp k_hyper_*): the same step the HIP layer runs as k_draw_k
  void set_k_hyper(double df, double scale, double nodeScale, double k0) { kModeled_ = true; kh_.df = df; kh_.invScale2 = std::isinf(scale) ? 0.0 : 1.0 / (scale * scale); kh_.nodeScale = nodeScale; kCur_ = k0; }
  double k_current() const { return kCur_; }
  void set_k(double k) { kCur_ = k; a_.model.leafPrec = leaf_precision(k, T_, kh_.nodeScale); }
  void draw_k() {
    double sumSq = 0.0, leaves = 0.0;
    for (int t = 0; t < T_; ++t) { double s, m; k_hyper_tree_stats(a_, t, s, m); sumSq += s; leaves += m; }
    set_k(k_hyper_draw(&rng_, kh_, T_, sumSq, leaves, kCur_));
  }
  bool kModeled_ = false; KHyper kh_{0, 0, 1}; double kCur_ = 2.0;
  // dbarts probit step: z_i ~ N(fit_i + offset_i, 1) truncated by y_i, sequential draws from R's generator
  double lower_trunc_std_normal(double lower) {
    double x;
    if (lower < 0.0) { x = r_norm(&rng_); while (x < lower) x = r_norm(&rng_); }
    else {
      double a = 0.5 * (lower + std::sqrt(lower * lower + 4.0)), u, r;
      do { x = r_exp(&rng_) / a + lower; u = r_unif(&rng_); double d = x - a; r = std::exp(-0.5 * d * d); } while (u > r);
    }
    return x;
  }
  void sample_latents() {
    for (size_t i = 0; i < n_; ++i) {
      double F = lat_[i] - R_[i];
      double mean = F + off_[i];
      double z = (y_[i] > 0.0) ? mean + lower_trunc_std_normal(0.0 - mean) : mean - lower_trunc_std_normal(mean - 0.0);
      lat_[i] = z - off_[i];
      R_[i] = lat_[i] - F;
    }
  }
  void predict_stored(const uint16_t* xb, int64_t nT, const PackedNode* nodes, size_t, const int64_t* treeStart, int64_t S, int T, const double* scale,
                      int binary, double* out) {
    for (int64_t k = 0; k < S; ++k) for (int64_t i = 0; i < nT; ++i) {
      double f = 0.0;
      for (int t = 0; t < T; ++t) {
        const PackedNode* base = nodes + treeStart[k * T + t];
        int nd = 0;
        while (base[nd].var >= 0) nd = (xb[(size_t)base[nd].var * (size_t)nT + i] <= base[nd].cut) ? base[nd].left : base[nd].right;
        f += base[nd].mu;
      }
      out[(size_t)k * (size_t)nT + i] = binary ? f : (f + 0.5) * scale[2 * k + 1] + scale[2 * k];
    }
  }
  void profile_leapfrog(int, const double*, const double*, double* out) { out[0] = out[1] = out[2] = 0.0; }
  void set_device_sharing(int) {}
  void set_tree_path(int path) { if (path < 0 || path > 5 || path == 3) throw std::invalid_argument("tree path must be 0 (automatic), 1 (two-kernel), 2 (fused), 4 (persistent) or 5 (persistent, streaming pass)"); pathReq_ = path; }
  void get_tree_path(int32_t out[2]) const { out[0] = pathReq_; out[1] = 1; }
  void fused_stats(int64_t out[2]) const { out[0] = 0; out[1] = 0; }
  void reset_fused_scales() {}
  void sweep_stats(int64_t out[2]) const { out[0] = 0; out[1] = 0; }
  int64_t sweep_busy() const { return 0; }
  void set_test_hook(int, int64_t) { throw std::invalid_argument("set_test_hook: the emulation of the device layer has no persistent launch"); }
  void sweep_spec(int64_t out[4]) const { out[0] = 0; out[1] = 0; out[2] = 0; out[3] = 0; }
  void profile_sweep(int nSweeps, int thin, double* out) { for (int i = 0; i < 8; ++i) out[i] = 0.0; for (int k = 0; k < nSweeps; ++k) sweep(thin); }
  void request_test_fits() {}      // (the device layer queues them behind the Stan inputs: nothing to overlap here)
  void request_var_counts() {}
  void var_counts(int32_t* out) {
    for (int j = 0; j < P_; ++j) out[j] = 0;
    for (int t = 0; t < T_; ++t) { TreeView tv = tree_view(a_, t); int nd, k; Walker<TreeView> w(tv, 0); while (w.next(nd, k)) if (k == 1) ++out[tv.var.get(nd)]; }
  }
  void test_fits(double* out) {
    for (size_t i = 0; i < nTest_; ++i) {
      double f = 0.0;
      for (int t = 0; t < T_; ++t) {
        TreeView tv = tree_view(a_, t);
        int nd = 0;
        while (tv.var.get(nd) >= 0) nd = (xbinTest_[(size_t)tv.var.get(nd) * nTest_ + i] <= tv.cut.get(nd)) ? tv.left.get(nd) : tv.right.get(nd);
        f += mu_[(size_t)t * nc_ + nd];
      }
      out[i] = binary_ ? f : (f + 0.5) * scale_.range + scale_.min;
    }
  }

  // ---- Stan inputs
  void sweep_and_stan_inputs(int thin, int mode, bool wantTrain, double* cX, double* cZ, double* s0, double* trainOut) {
    sweep(thin); stan_inputs(mode, wantTrain, cX, cZ, s0, trainOut);
  }
  void stan_inputs(int mode, bool wantTrain, double* cX, double* cZ, double* s0, double* trainOut) {
    double ss = 0.0;
    for (int k = 0; k < K_; ++k) cX[k] = 0.0;
    for (int j = 0; j < q_; ++j) cZ[j] = 0.0;
    for (size_t i = 0; i < n_; ++i) {
      double fit = 0.0, resp = y_[i];
      if (mode != 0) {
        if (binary_) { fit = lat_[i] - R_[i]; resp = lat_[i] + off_[i]; }   // Stan's response is the latent incl. the offset
        else { double yr = (y_[i] - off_[i] - scale_.min) / scale_.range - 0.5; fit = ((yr - R_[i]) + 0.5) * scale_.range + scale_.min; }
      }
      double so = mode == 0 ? 0.0 : mode == 1 ? fit : mode == 2 ? user_[i] : fit + user_[i];
      double e = resp - so;
      e0_[i] = e;
      const double we = wts_.empty() ? e : wts_[i] * e;
      ss += we * e;
      for (int k = 0; k < K_; ++k) cX[k] += X_[(size_t)k * n_ + i] * we;
      if (q_) for (int z = u_[i]; z < u_[i + 1]; ++z) cZ[v_[(size_t)z]] += w_[(size_t)z] * we;
      if (wantTrain && trainOut) trainOut[i] = fit;
    }
    *s0 = ss;
  }
  double leapfrog_sums(const double* beta, const double* b, double* gX, double* gZ) {
    double ss = 0.0;
    for (int k = 0; k < K_; ++k) gX[k] = 0.0;
    for (int j = 0; j < q_; ++j) gZ[j] = 0.0;
    for (size_t i = 0; i < n_; ++i) {
      double e = e0_[i];
      for (int k = 0; k < K_; ++k) e -= X_[(size_t)k * n_ + i] * beta[k];
      if (q_) for (int z = u_[i]; z < u_[i + 1]; ++z) e -= w_[(size_t)z] * b[v_[(size_t)z]];
      const double we = wts_.empty() ? e : wts_[i] * e;
      ss += we * e;
      for (int k = 0; k < K_; ++k) gX[k] += X_[(size_t)k * n_ + i] * we;
      if (q_) for (int z = u_[i]; z < u_[i + 1]; ++z) gZ[v_[(size_t)z]] += w_[(size_t)z] * we;
    }
    return ss;
  }

 private:
  void stats(int t) {
    const StepScratch& c = a_.sc[t & 1];
    const Proposal& pr = *c.prop;
    int nb = pr.nbA + pr.nbB;
    for (int k = 0; k < nb; ++k) { binCnt_[(size_t)k] = 0.0; binSum_[(size_t)k] = 0.0; binWt_[(size_t)k] = 0.0; }
    const bool wt = !wts_.empty();
    const double* mu = &mu_[(size_t)t * nc_];
    const uint16_t* leaf = &leaf_[(size_t)t * n_];
    for (size_t i = 0; i < n_; ++i) {
      int lf = leaf[i];
      const double wi = wt ? wts_[i] : 1.0;
      double r = (R_[i] + mu[lf]) * wi;
      int a = c.binA[lf];
      binCnt_[(size_t)a] += 1.0; binSum_[(size_t)a] += r; binWt_[(size_t)a] += wi;
      if (c.insub[lf]) {
        int nd = pr.node;
        while (c.pvar[nd] >= 0) nd = (xbin_[(size_t)c.pvar[nd] * n_ + i] <= c.pcut[nd]) ? c.pleft[nd] : c.pright[nd];
        int b = c.binB[nd];
        binCnt_[(size_t)b] += 1.0; binSum_[(size_t)b] += r; binWt_[(size_t)b] += wi;
      }
    }
  }
  void apply(int t) {
    const StepScratch& c = a_.sc[t & 1];
    const Proposal& pr = *c.prop;
    TreeView tv = tree_view(a_, t);
    const double* mu = &mu_[(size_t)t * nc_];
    uint16_t* leaf = &leaf_[(size_t)t * n_];
    const bool acc = *c.accepted != 0;
    for (size_t i = 0; i < n_; ++i) {
      int lf = leaf[i], nl = lf;
      if (acc && c.insub[lf]) {
        int nd = pr.node;
        while (tv.var.get(nd) >= 0) nd = (xbin_[(size_t)tv.var.get(nd) * n_ + i] <= tv.cut.get(nd)) ? tv.left.get(nd) : tv.right.get(nd);
        nl = nd; leaf[i] = (uint16_t)nd;
      }
      R_[i] = (R_[i] + c.muOld[lf]) - mu[nl];
    }
  }

  struct Scratch { std::vector<int16_t> pvar, pleft, pright, pparent, binA, binB, list, pna, pdep; std::vector<double> work; std::vector<uint16_t> pcut; std::vector<uint8_t> insub;
                   std::vector<double> muOld; Proposal prop; int32_t accepted = 0; };
  DevInit di_;
  size_t n_ = 0, nTest_ = 0; int P_ = 0, T_ = 0, nc_ = 0, K_ = 0, q_ = 0;
  std::vector<uint16_t> xbin_, xbinTest_, leaf_, cut_;
  std::vector<int32_t> numCuts_, cnt_, hwm_, v_, u_;
  std::vector<double> y_, user_, X_, w_, R_, off_, offNew_, e0_, mu_, binCnt_, binSum_, binWt_, wts_, lat_; bool binary_ = false;
  std::vector<int16_t> var_, left_, right_, parent_, cna_, cdep_, cleaf_, cpre_, cpost_;
  std::vector<int32_t> cnl_, cni_, cg_, cgn_, cvalid_; std::vector<double> clogpi_;
  Scratch sc_[2];
  std::vector<StepRecord> trace_;
  MTState rng_; ScaleState scale_; BartArrays a_;
  int32_t traceCount_ = 0, err_ = 0; int64_t launches_ = 0; int pathReq_ = 0;
};

}  // namespace s4b
#endif
