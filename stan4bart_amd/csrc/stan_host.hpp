// Host side of the Stan block of the MI355X path.
//
// What stays on the host is O(D) per gradient: the unconstrain->constrain transforms, make_theta_L /
// make_b, the priors and Jacobians of reference src/stan_files/continuous.stan:261-429, and the NUTS
// control flow of reference src/include/stan/mcmc/hmc/nuts/base_nuts.hpp:78-352.  Everything O(N)
// (X beta + Z b, residual sum of squares, X'e, Z'e) is produced by HIP kernels and enters through
// the `Likelihood` callback, either once per Gibbs iteration as sufficient statistics
// (c = [X Z]'(y - offset), s0 = |y - offset|^2 with the constant Gram matrix G = [X Z]'[X Z]) or once
// per leapfrog (hmc_mode = 1).
//
// The O(D) part is differentiated with a small reverse-mode tape (the reference uses Stan's
// reverse-mode AD: src/include/stan/math/rev/functor/gradient.hpp:46-56).
#ifndef S4B_STAN_HOST_HPP
#define S4B_STAN_HOST_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <chrono>
#include <cstdio>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

namespace s4b {

// ---------------------------------------------------------------- reverse-mode tape
// The model's control flow depends on the data description only (StanSpec), never on parameter values: the tape of a gradient has
// the same nodes in the same order every time.  Every node therefore carries its operation, and once recorded a tape is REPLAYED —
// values and partial derivatives recomputed in place, node by node, with the formulas that recorded them — instead of being rebuilt
// through the model code (hundreds of gradients per Gibbs iteration once the chain runs deep NUTS trees).
enum TapeCode : int { T_CONST = 0, T_INPUT, T_ADD, T_SUB, T_MUL, T_DIV, T_ADDC, T_SUBC, T_MULC, T_DIVC, T_CSUB, T_CDIV, T_SQRT, T_EXP, T_LOG,
                      T_SQUARE, T_LOG1M, T_STDNORMAL, T_INVLOGIT, T_INVLOGIT_LOGJAC };
class Tape {
 public:
  struct Op { double val; int a, b; double da, db; double c; int code; int k; };
  struct Ops {
    std::vector<Op> buf; size_t n = 0;
    size_t size() const { return n; }
    void clear() { n = 0; }
    void reserve(size_t k) { if (buf.size() < k) buf.resize(k); }
    Op& operator[](size_t i) { return buf[i]; }
    const Op& operator[](size_t i) const { return buf[i]; }
    int push(const Op& o) { if (n == buf.size()) buf.resize(buf.empty() ? 1024 : 2 * buf.size()); buf[n] = o; return (int)n++; }
  } ops;
  std::vector<double> adj;
  void clear() { ops.clear(); }
  // value and partial derivatives of one node from its operands
  static inline void eval(Op& o, const Op* all) {
    const double x = o.a >= 0 ? all[o.a].val : 0.0, y = o.b >= 0 ? all[o.b].val : 0.0, c = o.c;
    switch (o.code) {
      case T_CONST: case T_INPUT: break;
      case T_ADD: o.val = x + y; o.da = 1; o.db = 1; break;
      case T_SUB: o.val = x - y; o.da = 1; o.db = -1; break;
      case T_MUL: o.val = x * y; o.da = y; o.db = x; break;
      case T_DIV: { const double r = x / y; o.val = r; o.da = 1.0 / y; o.db = -r / y; break; }
      case T_ADDC: o.val = x + c; o.da = 1; break;
      case T_SUBC: o.val = x - c; o.da = 1; break;
      case T_MULC: o.val = x * c; o.da = c; break;
      case T_DIVC: o.val = x / c; o.da = 1.0 / c; break;
      case T_CSUB: o.val = c - x; o.da = -1; break;
      case T_CDIV: { const double r = c / x; o.val = r; o.da = -r / x; break; }
      case T_SQRT: { const double r = std::sqrt(x); o.val = r; o.da = 0.5 / r; break; }
      case T_EXP: { const double e = std::exp(x); o.val = e; o.da = e; break; }
      case T_LOG: o.val = std::log(x); o.da = 1.0 / x; break;
      case T_SQUARE: o.val = x * x; o.da = 2 * x; break;
      case T_LOG1M: o.val = std::log1p(-x); o.da = -1.0 / (1.0 - x); break;
      case T_STDNORMAL: o.val = -0.5 * x * x - 0.91893853320467274178; o.da = -x; break;
      case T_INVLOGIT: { const double il = 1.0 / (1.0 + std::exp(-x)); o.val = il; o.da = il * (1.0 - il); break; }
      case T_INVLOGIT_LOGJAC: {   // log-Jacobian of lub_constrain(x, 0, 1): -|x| - 2 log1p(exp(-|x|))
        const double ax = std::fabs(x), sg = x >= 0 ? 1.0 : -1.0, e = std::exp(-ax);
        o.val = -ax - 2.0 * std::log1p(e); o.da = -sg + 2.0 * sg * e / (1.0 + e); break;
      }
    }
  }
  int node(int code, int a, int b, double c) {
    Op o; o.val = 0; o.a = a; o.b = b; o.da = 0; o.db = 0; o.c = c; o.code = code; o.k = -1;
    const int i = ops.push(o);
    eval(ops[(size_t)i], ops.buf.data());
    return i;
  }
  int leaf(double v) { Op o; o.val = v; o.a = -1; o.b = -1; o.da = 0; o.db = 0; o.c = 0; o.code = T_CONST; o.k = -1; return ops.push(o); }
  int input(int k, double v) { Op o; o.val = v; o.a = -1; o.b = -1; o.da = 0; o.db = 0; o.c = 0; o.code = T_INPUT; o.k = k; return ops.push(o); }
  double val(int i) const { return ops[(size_t)i].val; }
  // the recorded nodes again, for new input values (the caller has stored them in the input nodes)
  void replay() {
    Op* const all = ops.buf.data();
    const size_t n = ops.size();
    for (size_t i = 0; i < n; ++i) if (all[i].code > T_INPUT) eval(all[i], all);
  }
  // propagate: caller seeds adj (size = ops.size()) then calls backward()
  void backward() {
    for (int i = (int)ops.size() - 1; i >= 0; --i) {
      const Op& o = ops[(size_t)i];
      double g = adj[(size_t)i];
      if (g == 0.0) continue;
      if (o.a >= 0) adj[(size_t)o.a] += g * o.da;
      if (o.b >= 0) adj[(size_t)o.b] += g * o.db;
    }
  }
};

struct TV {   // tape variable handle
  Tape* t; int i;
  double v() const { return t->val(i); }
};
inline TV operator+(TV a, TV b) { return {a.t, a.t->node(T_ADD, a.i, b.i, 0)}; }
inline TV operator-(TV a, TV b) { return {a.t, a.t->node(T_SUB, a.i, b.i, 0)}; }
inline TV operator*(TV a, TV b) { return {a.t, a.t->node(T_MUL, a.i, b.i, 0)}; }
inline TV operator/(TV a, TV b) { return {a.t, a.t->node(T_DIV, a.i, b.i, 0)}; }
inline TV operator+(TV a, double c) { return {a.t, a.t->node(T_ADDC, a.i, -1, c)}; }
inline TV operator-(TV a, double c) { return {a.t, a.t->node(T_SUBC, a.i, -1, c)}; }
inline TV operator*(TV a, double c) { return {a.t, a.t->node(T_MULC, a.i, -1, c)}; }
inline TV operator*(double c, TV a) { return a * c; }
inline TV operator/(TV a, double c) { return {a.t, a.t->node(T_DIVC, a.i, -1, c)}; }
inline TV operator-(double c, TV a) { return {a.t, a.t->node(T_CSUB, a.i, -1, c)}; }
inline TV operator/(double c, TV a) { return {a.t, a.t->node(T_CDIV, a.i, -1, c)}; }
inline TV tsqrt(TV a) { return {a.t, a.t->node(T_SQRT, a.i, -1, 0)}; }
inline TV texp(TV a) { return {a.t, a.t->node(T_EXP, a.i, -1, 0)}; }
inline TV tlog(TV a) { return {a.t, a.t->node(T_LOG, a.i, -1, 0)}; }
inline TV tsquare(TV a) { return {a.t, a.t->node(T_SQUARE, a.i, -1, 0)}; }
inline TV tlog1m(TV a) { return {a.t, a.t->node(T_LOG1M, a.i, -1, 0)}; }
// log N(z | 0, 1) as one node
inline TV tstd_normal_lpdf(TV z) { return {z.t, z.t->node(T_STDNORMAL, z.i, -1, 0)}; }

// ---------------------------------------------------------------- model description (the stanData list)
struct StanSpec {
  int64_t N = 0; int K = 0, q = 0, t = 0, len_theta_L = 0;
  int is_binary = 0, prior_dist = 1, prior_dist_for_aux = 3;
  std::vector<double> prior_scale, prior_mean, prior_df;
  double prior_scale_for_aux = 1, prior_mean_for_aux = 0, prior_df_for_aux = 1;
  double global_prior_df = 1, global_prior_scale = 1, slab_df = 1, slab_scale = 1;   // hs, hs_plus
  std::vector<int> num_normals;                                                       // product_normal
  std::vector<int> p, l;
  std::vector<double> shape, scale, concentration, regularization;
  // derived (continuous.stan transformed data, src/stan_sampler.cpp:167-182)
  int len_z_T = 0, len_rho = 0, len_conc = 0, D = 0, n_constrained = 0;
  int hs = 0, n_z_beta = 0, n_mix = 0, n_lambda = 0;   // extra blocks of the hs / laplace / lasso / product_normal priors
  std::vector<double> delta;
  void finish() {
    int sum_p = 0; len_z_T = 0; delta.clear();
    for (int i = 0; i < t; ++i) {
      sum_p += p[i];
      if (p[i] > 1) for (int j = 0; j < p[i]; ++j) delta.push_back(concentration[(size_t)j]);
      for (int j = 3; j <= p[i]; ++j) len_z_T += p[i] - 1;
    }
    len_rho = sum_p - t; len_conc = (int)delta.size();
    hs = prior_dist == 3 ? 2 : (prior_dist == 4 ? 4 : 0);
    n_z_beta = K;
    if (prior_dist == 7) { n_z_beta = 0; for (int k = 0; k < K; ++k) n_z_beta += num_normals[(size_t)k]; }
    n_mix = (prior_dist == 5 || prior_dist == 6) ? K : 0; n_lambda = prior_dist == 6 ? 1 : 0;
    D = n_z_beta + hs + hs * K + (hs > 0 ? 1 : 0) + n_mix + n_lambda + q + len_z_T + len_rho + len_conc + t + (is_binary ? 0 : 1);
    n_constrained = D + (is_binary ? 0 : 1) + K + q + len_theta_L;
  }
  int aux_pos() const { return D; }
  int beta_pos() const { return D + (is_binary ? 0 : 1); }
  int b_pos() const { return beta_pos() + K; }
};

// O(N) part: given beta, b, sigma returns ss = sum (y - offset - X beta - Z b)^2 and fills
// gX = X'e, gZ = Z'e (e the residual).  Implemented by the device layer.
typedef std::function<double(const double* beta, const double* b, double* gX, double* gZ)> Likelihood;

class HostModel {
 public:
  StanSpec sp;
  Likelihood lik;
  long gradEvals = 0;
  explicit HostModel(const StanSpec& s) : sp(s) { sp.finish(); plan_closed_form(); }
  // (test hook) 0: the closed-form gradient where it applies (default); 1: the tape everywhere; 2: both, compared (throws on a difference)
  int gradientCheck = 0;

  struct Fwd { TV sigma; std::vector<TV> beta, b, theta_L, constrained; TV lp; bool wantConstrained = true; };

  void forward(Tape& tp, const std::vector<double>& qv, Fwd& F, std::vector<int>& qidx, bool jacobian) const {
    tp.clear();
    tp.ops.reserve(4096);
    qidx.resize((size_t)sp.D);
    for (int i = 0; i < sp.D; ++i) qidx[(size_t)i] = tp.input(i, qv[(size_t)i]);
    auto Q = [&](int i) { return TV{&tp, qidx[(size_t)i]}; };
    TV lp{&tp, tp.leaf(0.0)};
    int pos = 0;
    std::vector<TV>& z_beta = w_[0]; std::vector<TV>& z_b = w_[1]; std::vector<TV>& z_T = w_[2];
    std::vector<TV>& rho = w_[3]; std::vector<TV>& zeta = w_[4]; std::vector<TV>& tau = w_[5];
    for (auto& v : w_) v.clear();
    for (int k = 0; k < sp.n_z_beta; ++k) z_beta.push_back(Q(pos++));
    // lower-bounded blocks of the shrinkage priors (continuous.stan:266-270; arrays of vectors are read array-major)
    auto lb0e = [&](int idx) { TV x = Q(idx); if (jacobian) lp = lp + x; return texp(x); };
    std::vector<TV>& global = sGlobal_; std::vector<TV>& caux = sCaux_; std::vector<TV>& lambda1 = sLambda_; std::vector<TV>& mix = sMix_;
    std::vector<std::vector<TV>>& local = sLocal_;
    global.clear(); caux.clear(); lambda1.clear(); mix.clear(); local.resize((size_t)sp.hs); for (auto& v : local) v.clear();
    for (int j = 0; j < sp.hs; ++j) global.push_back(lb0e(pos++));
    for (int j = 0; j < sp.hs; ++j) for (int k = 0; k < sp.K; ++k) local[(size_t)j].push_back(lb0e(pos++));
    if (sp.hs > 0) caux.push_back(lb0e(pos++));
    for (int k = 0; k < sp.n_mix; ++k) mix.push_back(lb0e(pos++));
    if (sp.n_lambda) lambda1.push_back(lb0e(pos++));
    for (int j = 0; j < sp.q; ++j) z_b.push_back(Q(pos++));
    for (int j = 0; j < sp.len_z_T; ++j) z_T.push_back(Q(pos++));
    for (int j = 0; j < sp.len_rho; ++j) {   // lub_constrain(x, 0, 1): inv_logit, log-Jacobian -|x| - 2 log1p(exp(-|x|))
      TV x = Q(pos++);
      rho.push_back(TV{&tp, tp.node(T_INVLOGIT, x.i, -1, 0)});
      if (jacobian) lp = lp + TV{&tp, tp.node(T_INVLOGIT_LOGJAC, x.i, -1, 0)};
    }
    auto lb0 = [&](int idx) { TV x = Q(idx); if (jacobian) lp = lp + x; return texp(x); };
    for (int j = 0; j < sp.len_conc; ++j) zeta.push_back(lb0(pos++));
    for (int j = 0; j < sp.t; ++j) tau.push_back(lb0(pos++));
    TV aux_unscaled{&tp, -1};
    if (!sp.is_binary) aux_unscaled = lb0(pos++);

    TV aux{&tp, tp.leaf(1.0)};
    if (!sp.is_binary) {
      if (sp.prior_dist_for_aux == 0) aux = aux_unscaled;
      else { aux = aux_unscaled * sp.prior_scale_for_aux; if (sp.prior_dist_for_aux <= 2) aux = aux + sp.prior_mean_for_aux; }
    }
    F.sigma = aux;
    F.beta.clear();
    if (sp.prior_dist <= 2) for (int k = 0; k < sp.K; ++k) {
      if (sp.prior_dist == 0) F.beta.push_back(z_beta[(size_t)k]);
      else if (sp.prior_dist == 1) F.beta.push_back(z_beta[(size_t)k] * sp.prior_scale[(size_t)k] + sp.prior_mean[(size_t)k]);
      else F.beta.push_back(cornish_fisher(z_beta[(size_t)k], sp.prior_df[(size_t)k]) * sp.prior_scale[(size_t)k] + sp.prior_mean[(size_t)k]);
    } else if (sp.hs > 0) {   // hs_prior / hsplus_prior (continuous.stan:124-144), error_scale = aux
      TV c2 = caux[0] * (sp.slab_scale * sp.slab_scale);
      TV tauG = global[0] * tsqrt(global[1]) * sp.global_prior_scale * aux;
      for (int k = 0; k < sp.K; ++k) {
        TV lam = local[0][(size_t)k] * tsqrt(local[1][(size_t)k]);
        if (sp.hs == 4) lam = lam * (local[2][(size_t)k] * tsqrt(local[3][(size_t)k]));
        TV lam2 = tsquare(lam);
        TV tilde = tsqrt(c2 * lam2 / (c2 + tsquare(tauG) * lam2));
        F.beta.push_back(z_beta[(size_t)k] * tilde * tauG);
      }
    } else if (sp.prior_dist == 5) {
      for (int k = 0; k < sp.K; ++k) F.beta.push_back(tsqrt(mix[(size_t)k] * 2.0) * sp.prior_scale[(size_t)k] * z_beta[(size_t)k] + sp.prior_mean[(size_t)k]);
    } else if (sp.prior_dist == 6) {
      for (int k = 0; k < sp.K; ++k)
        F.beta.push_back(lambda1[0] * sp.prior_scale[(size_t)k] * tsqrt(mix[(size_t)k] * 2.0) * z_beta[(size_t)k] + sp.prior_mean[(size_t)k]);
    } else {               // product_normal
      int zp = 0;
      for (int k = 0; k < sp.K; ++k) {
        TV bk = z_beta[(size_t)zp++];
        for (int n2 = 2; n2 <= sp.num_normals[(size_t)k]; ++n2) bk = bk * z_beta[(size_t)zp++];
        F.beta.push_back(bk * std::pow(sp.prior_scale[(size_t)k], (double)sp.num_normals[(size_t)k]) + sp.prior_mean[(size_t)k]);
      }
    }
    theta_L(tp, aux, tau, zeta, rho, z_T, F.theta_L);
    make_b(tp, z_b, F.theta_L, F.b);
    F.constrained.clear();
    if (F.wantConstrained) {
    for (auto& x : z_beta) F.constrained.push_back(x);
    for (auto& x : global) F.constrained.push_back(x);
    for (int k = 0; k < (sp.hs ? sp.K : 0); ++k) for (int j = 0; j < sp.hs; ++j) F.constrained.push_back(local[(size_t)j][(size_t)k]);   // vector-index-major
    for (auto& x : caux) F.constrained.push_back(x);
    for (auto& x : mix) F.constrained.push_back(x);
    for (auto& x : lambda1) F.constrained.push_back(x);
    for (auto& x : z_b) F.constrained.push_back(x);
    for (auto& x : z_T) F.constrained.push_back(x);
    for (auto& x : rho) F.constrained.push_back(x);
    for (auto& x : zeta) F.constrained.push_back(x);
    for (auto& x : tau) F.constrained.push_back(x);
    if (!sp.is_binary) F.constrained.push_back(aux_unscaled);
    }

    const double HALF_LOG_2PI = 0.91893853320467274178;
    if (!sp.is_binary && sp.prior_dist_for_aux > 0 && sp.prior_scale_for_aux > 0) {
      const double log_half = -0.693147180559945286;
      if (sp.prior_dist_for_aux == 1) lp = lp + (tsquare(aux_unscaled) * -0.5 - HALF_LOG_2PI) - log_half;
      else if (sp.prior_dist_for_aux == 2) {
        double nu = sp.prior_df_for_aux;
        TV tt = tlog(tsquare(aux_unscaled) / nu + 1.0) * (-(nu + 1.0) / 2.0);
        lp = lp + (tt + (lg((nu + 1.0) / 2.0) - lg(nu / 2.0) - 0.5 * std::log(nu * M_PI))) - log_half;
      } else lp = lp - aux_unscaled;
    }
    if (sp.prior_dist >= 1) for (auto& z : z_beta) lp = lp + tstd_normal_lpdf(z);
    {
      const double log_half = -0.693147180559945286;
      auto half_normal = [&](const std::vector<TV>& v) { for (auto& x : v) lp = lp + tstd_normal_lpdf(x); lp = lp - log_half; };
      auto inv_gamma = [&](TV x, double al, double be) { lp = lp + (tlog(x) * (-(al + 1.0)) - (1.0 / x) * be + (al * std::log(be) - lg(al))); };
      if (sp.hs > 0) {
        half_normal(local[0]);
        for (int k = 0; k < sp.K; ++k) inv_gamma(local[1][(size_t)k], 0.5 * sp.prior_df[(size_t)k], 0.5 * sp.prior_df[(size_t)k]);
        if (sp.hs == 4) {
          half_normal(local[2]);
          for (int k = 0; k < sp.K; ++k) inv_gamma(local[3][(size_t)k], 0.5 * sp.prior_scale[(size_t)k], 0.5 * sp.prior_scale[(size_t)k]);
        }
        lp = lp + tstd_normal_lpdf(global[0]) - log_half;
        inv_gamma(global[1], 0.5 * sp.global_prior_df, 0.5 * sp.global_prior_df);
        inv_gamma(caux[0], 0.5 * sp.slab_df, 0.5 * sp.slab_df);
      }
      for (auto& x : mix) lp = lp - x;   // exponential_lpdf(mix | 1)
      if (sp.n_lambda) {                 // chi_square_lpdf(one_over_lambda | prior_df[1])
        double nu = sp.prior_df[0];
        lp = lp + (tlog(lambda1[0]) * (0.5 * nu - 1.0) - lambda1[0] * 0.5 - (0.5 * nu * std::log(2.0) + lg(0.5 * nu)));
      }
    }
    for (auto& z : z_b) lp = lp + tstd_normal_lpdf(z);
    for (auto& z : z_T) lp = lp + tstd_normal_lpdf(z);
    int pos_reg = 0, pos_rho = 0;
    for (int i = 0; i < sp.t; ++i) if (sp.p[(size_t)i] > 1) {
      int m = sp.p[(size_t)i] - 1;
      std::vector<double>& s1 = sS1_; std::vector<double>& s2 = sS2_; s1.assign((size_t)m, 0.0); s2.assign((size_t)m, 0.0);
      double nu = sp.regularization[(size_t)pos_reg++] + 0.5 * (sp.p[(size_t)i] - 2);
      s1[0] = nu; s2[0] = nu;
      for (int j = 2; j <= m; ++j) { nu -= 0.5; s1[(size_t)j - 1] = 0.5 * j; s2[(size_t)j - 1] = nu; }
      for (int j = 0; j < m; ++j) {
        TV r = rho[(size_t)(pos_rho + j)];
        double lbeta = lg(s1[(size_t)j]) + lg(s2[(size_t)j]) - lg(s1[(size_t)j] + s2[(size_t)j]);
        lp = lp + (tlog(r) * (s1[(size_t)j] - 1.0) + tlog1m(r) * (s2[(size_t)j] - 1.0) - lbeta);
      }
      pos_rho += m;
    }
    for (int j = 0; j < sp.len_conc; ++j) lp = lp + (tlog(zeta[(size_t)j]) * (sp.delta[(size_t)j] - 1.0) - zeta[(size_t)j] - lg(sp.delta[(size_t)j]));
    for (int j = 0; j < sp.t; ++j) lp = lp + (tlog(tau[(size_t)j]) * (sp.shape[(size_t)j] - 1.0) - tau[(size_t)j] - lg(sp.shape[(size_t)j]));
    F.lp = lp;
  }

  double log_prob_grad(const std::vector<double>& qv, std::vector<double>& grad) {
    if (closedForm_ && gradientCheck != 1) {
      ++gradEvals;
      grad.resize((size_t)sp.D);
      const double lp = closed_form_grad(qv.data(), grad.data());
      if (gradientCheck == 2) {
        std::vector<double> g2; const double lp2 = tape_grad(qv, g2); --gradEvals;
        // (far outside the typical set — a diverging trajectory — components of 1e9 are differences of terms of 1e17: the two
        // evaluation orders then agree to the conditioning of the sum, not to 1e-9 of the component)
        double gmax = 0.0; for (double x : g2) if (std::isfinite(x)) gmax = std::max(gmax, std::fabs(x));
        auto off = [gmax](double a, double b) { return std::fabs(a - b) > 1e-9 * (std::fabs(a) + std::fabs(b)) + 1e-11 * gmax + 1e-300 && !(std::isnan(a) && std::isnan(b)) && !(std::isinf(a) && a == b); };
        bool bad = off(lp, lp2);
        for (int i = 0; i < sp.D; ++i) bad = bad || off(grad[(size_t)i], g2[(size_t)i]);
        if (bad) {
          std::string m = "closed-form gradient differs from the tape: lp " + std::to_string(lp) + " vs " + std::to_string(lp2);
          char buf[96];
          for (int i = 0; i < sp.D; ++i) if (off(grad[(size_t)i], g2[(size_t)i])) { std::snprintf(buf, sizeof buf, "; g[%d] %.17g vs %.17g (q %.6g)", i, grad[(size_t)i], g2[(size_t)i], qv[(size_t)i]); m += buf; }
          throw std::logic_error(m);
        }
      }
      return lp;
    }
    return tape_grad(qv, grad);
  }
  double tape_grad(const std::vector<double>& qv, std::vector<double>& grad) {
    ++gradEvals;
    Fwd& F = fwd_; std::vector<int>& qidx = qidx_;
    F.wantConstrained = false;
#ifdef S4B_NUTS_TIMING
    const auto t0 = std::chrono::steady_clock::now();
#endif
    if (tapeHoldsGradient_) {      // same nodes as last time: new inputs, values and partials recomputed in place
      for (int i = 0; i < sp.D; ++i) tape_.ops[(size_t)qidx[(size_t)i]].val = qv[(size_t)i];
      tape_.replay();
    } else { forward(tape_, qv, F, qidx, true); tapeHoldsGradient_ = true; }
#ifdef S4B_NUTS_TIMING
    const auto t1 = std::chrono::steady_clock::now();
#endif
    std::vector<double>& beta = bufBeta_; std::vector<double>& b = bufB_; std::vector<double>& gX = bufGX_; std::vector<double>& gZ = bufGZ_;
    beta.resize((size_t)sp.K); b.resize((size_t)sp.q); gX.assign((size_t)sp.K, 0.0); gZ.assign((size_t)sp.q, 0.0);
    for (int k = 0; k < sp.K; ++k) beta[(size_t)k] = F.beta[(size_t)k].v();
    for (int j = 0; j < sp.q; ++j) b[(size_t)j] = F.b[(size_t)j].v();
    double sigma = F.sigma.v();
    double ss = lik(beta.data(), b.data(), gX.data(), gZ.data());
    double s2 = sigma * sigma, N = (double)sp.N;
    double ll = -0.5 * ss / s2 - N * std::log(sigma) - N * 0.91893853320467274178;
    tape_.adj.assign(tape_.ops.size(), 0.0);
    tape_.adj[(size_t)F.lp.i] = 1.0;
    tape_.adj[(size_t)F.sigma.i] += ss / (s2 * sigma) - N / sigma;
    for (int k = 0; k < sp.K; ++k) tape_.adj[(size_t)F.beta[(size_t)k].i] += gX[(size_t)k] / s2;
    for (int j = 0; j < sp.q; ++j) tape_.adj[(size_t)F.b[(size_t)j].i] += gZ[(size_t)j] / s2;
#ifdef S4B_NUTS_TIMING
    const auto t2 = std::chrono::steady_clock::now();
#endif
    tape_.backward();
    grad.resize((size_t)sp.D);
    for (int i = 0; i < sp.D; ++i) grad[(size_t)i] = tape_.adj[(size_t)qidx[(size_t)i]];
#ifdef S4B_NUTS_TIMING
    { const auto t3 = std::chrono::steady_clock::now(); auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      tFwd_ += us(t0, t1); tLik_ += us(t1, t2); tBwd_ += us(t2, t3);
      if (gradEvals % 5000 == 0) std::fprintf(stderr, "S4B gradient us: forward %.2f lik %.2f backward %.2f (tape %zu nodes)\n", tFwd_ / gradEvals, tLik_ / gradEvals, tBwd_ / gradEvals, tape_.ops.size()); }
#endif
    return F.lp.v() + ll;
  }

  // write_array_impl (continuous.hpp:2640-2938): constrained params, then aux, beta, b, theta_L
  void write_array(const std::vector<double>& qv, double* out) {
    Fwd F; std::vector<int> qidx;
    F.wantConstrained = true;
    tapeHoldsGradient_ = false;      // (this recording replaces the gradient's nodes)
    forward(tape_, qv, F, qidx, false);
    int o = 0;
    for (auto& x : F.constrained) out[o++] = x.v();
    if (!sp.is_binary) out[o++] = F.sigma.v();
    for (auto& x : F.beta) out[o++] = x.v();
    for (auto& x : F.b) out[o++] = x.v();
    for (auto& x : F.theta_L) out[o++] = x.v();
  }

  bool has_closed_form() const { return closedForm_; }

 private:
  // ---- closed-form gradient of the default model family -----------------------------------------------------------------------------
  // Once the chain runs deep NUTS trees (~1000 leapfrogs per Gibbs iteration at n = 1e6) the O(D) part of a gradient IS the Stan block's
  // cost in hmc_mode 0.  For the configuration the reference fits by default — normal (or flat) coefficient prior, decov with at most
  // two coefficients per grouping term (random intercepts, one random slope), any aux prior — log pi and its gradient are written out
  // below: the same quantities as forward() (reference continuous.stan:261-429, make_theta_L :2-59, make_b :61-94, decov_lp :96-122)
  // without tape nodes.  Everything else (hs / laplace / lasso / product_normal / student_t coefficients, p_i > 2) stays on the tape;
  // tests/test_priors.py and gradientCheck = 2 compare the two.
  struct CfTerm { int p, l, zb0, rho, zeta, tau; double scale, betaA, lbeta, delta0, delta1, lgDelta0, lgDelta1, shape, lgShape; };
  struct CfWork { double tau, A, rho, e, z0, z1, S, pi0, pi1, trace, sd1, T21, sq; };   // forward values of one term, kept for the backward pass
  std::vector<CfWork> cfWork_;
  bool closedForm_ = false;
  std::vector<CfTerm> cfTerms_;
  int cfRho0_ = 0, cfZeta0_ = 0, cfTau0_ = 0, cfAux_ = 0;
  double cfAuxConst_ = 0;
  std::vector<double> cfBeta_, cfB_, cfGX_, cfGZ_, cfTh_;
  void plan_closed_form() {
    closedForm_ = false;
    if (sp.prior_dist > 1 || sp.hs || sp.n_mix || sp.n_lambda || sp.len_z_T) return;
    for (int i = 0; i < sp.t; ++i) if (sp.p[(size_t)i] > 2 || sp.p[(size_t)i] < 1) return;
    cfRho0_ = sp.K + sp.q; cfZeta0_ = cfRho0_ + sp.len_rho; cfTau0_ = cfZeta0_ + sp.len_conc; cfAux_ = cfTau0_ + sp.t;
    cfTerms_.clear();
    int zb = sp.K, rho = cfRho0_, zeta = cfZeta0_, reg = 0, dl = 0;
    for (int i = 0; i < sp.t; ++i) {
      CfTerm c{}; c.p = sp.p[(size_t)i]; c.l = sp.l[(size_t)i]; c.zb0 = zb; c.tau = cfTau0_ + i; c.scale = sp.scale[(size_t)i];
      c.shape = sp.shape[(size_t)i]; c.lgShape = lg(c.shape);
      c.rho = -1; c.zeta = -1;
      if (c.p == 2) {
        c.rho = rho++; c.zeta = zeta; zeta += 2;
        const double nu = sp.regularization[(size_t)reg++];          // (p = 2: one Beta(nu, nu) on rho)
        c.betaA = nu; c.lbeta = lg(nu) + lg(nu) - lg(nu + nu);
        c.delta0 = sp.delta[(size_t)dl]; c.delta1 = sp.delta[(size_t)dl + 1]; dl += 2;
        c.lgDelta0 = lg(c.delta0); c.lgDelta1 = lg(c.delta1);
      }
      zb += c.p * c.l;
      cfTerms_.push_back(c);
    }
    if (!sp.is_binary && sp.prior_dist_for_aux == 2) {
      const double nu = sp.prior_df_for_aux;
      cfAuxConst_ = lg((nu + 1.0) / 2.0) - lg(nu / 2.0) - 0.5 * std::log(nu * M_PI);
    }
    cfBeta_.assign((size_t)sp.K + 1, 0.0); cfB_.assign((size_t)sp.q + 1, 0.0); cfGX_.assign((size_t)sp.K + 1, 0.0); cfGZ_.assign((size_t)sp.q + 1, 0.0);
    cfTh_.assign((size_t)(3 * sp.t + 1), 0.0); cfWork_.assign((size_t)sp.t + 1, CfWork{});
    closedForm_ = true;
  }
  double closed_form_grad(const double* q, double* g) {
    const double HALF_LOG_2PI = 0.91893853320467274178, LOG_HALF = -0.693147180559945286;
    const int K = sp.K, nq = sp.q, t = sp.t;
    double lp = 0.0;
    // ---- aux (sigma)
    double u = 1.0, aux = 1.0, dAuxDu = 0.0;
    if (!sp.is_binary) {
      u = std::exp(q[cfAux_]); lp += q[cfAux_];
      if (sp.prior_dist_for_aux == 0) { aux = u; dAuxDu = 1.0; }
      else { aux = u * sp.prior_scale_for_aux; dAuxDu = sp.prior_scale_for_aux; if (sp.prior_dist_for_aux <= 2) aux += sp.prior_mean_for_aux; }
    }
    // ---- coefficients
    double* const beta = cfBeta_.data(); double* const b = cfB_.data();
    if (sp.prior_dist == 0) for (int k = 0; k < K; ++k) beta[k] = q[k];
    else for (int k = 0; k < K; ++k) { beta[k] = q[k] * sp.prior_scale[(size_t)k] + sp.prior_mean[(size_t)k]; lp += -0.5 * q[k] * q[k] - HALF_LOG_2PI; }
    // ---- theta_L, b (T = [[T00, 0], [T10, T11]] per term with two coefficients)
    double* const th = cfTh_.data();     // per term: T00 (or theta), T10, T11
    for (int i = 0; i < t; ++i) {
      const CfTerm& c = cfTerms_[(size_t)i];
      CfWork& w = cfWork_[(size_t)i];
      const double tau = std::exp(q[c.tau]); lp += q[c.tau];
      const double A = tau * c.scale * aux;
      w.tau = tau; w.A = A;
      const double* zb = q + c.zb0;
      double* bo = b + (c.zb0 - K);
      if (c.p == 1) {
        th[3 * i] = A;
        for (int s = 0; s < c.l; ++s) { bo[s] = A * zb[s]; lp += -0.5 * zb[s] * zb[s] - HALF_LOG_2PI; }
      } else {
        const double x = q[c.rho], ax = std::fabs(x), e = std::exp(-ax);
        const double rho = x >= 0 ? 1.0 / (1.0 + e) : e / (1.0 + e);
        lp += -ax - 2.0 * std::log1p(e);
        const double z0 = std::exp(q[c.zeta]), z1 = std::exp(q[c.zeta + 1]); lp += q[c.zeta] + q[c.zeta + 1];
        const double trace = A * A * 2.0, S = z0 + z1, pi0 = z0 / S, pi1 = z1 / S;
        const double sd0 = std::sqrt(pi0 * trace), sd1 = std::sqrt(pi1 * trace), T21 = rho * 2.0 - 1.0, sq = std::sqrt(1.0 - T21 * T21);
        const double T00 = sd0, T10 = sd1 * T21, T11 = sd1 * sq;
        th[3 * i] = T00; th[3 * i + 1] = T10; th[3 * i + 2] = T11;
        w.rho = rho; w.e = e; w.z0 = z0; w.z1 = z1; w.S = S; w.pi0 = pi0; w.pi1 = pi1; w.trace = trace; w.sd1 = sd1; w.T21 = T21; w.sq = sq;
        for (int j = 0; j < c.l; ++j) {
          const double a0 = zb[2 * j], a1 = zb[2 * j + 1];
          bo[2 * j] = T00 * a0; bo[2 * j + 1] = T10 * a0 + T11 * a1;
          lp += (-0.5 * a0 * a0 - HALF_LOG_2PI) + (-0.5 * a1 * a1 - HALF_LOG_2PI);
        }
        lp += std::log(rho) * (c.betaA - 1.0) + std::log1p(-rho) * (c.betaA - 1.0) - c.lbeta;
        lp += (q[c.zeta] * (c.delta0 - 1.0) - z0 - c.lgDelta0) + (q[c.zeta + 1] * (c.delta1 - 1.0) - z1 - c.lgDelta1);
      }
      lp += q[c.tau] * (c.shape - 1.0) - tau - c.lgShape;
    }
    // ---- aux prior
    double dLpDu = 0.0;
    if (!sp.is_binary && sp.prior_dist_for_aux > 0 && sp.prior_scale_for_aux > 0) {
      if (sp.prior_dist_for_aux == 1) { lp += (-0.5 * u * u - HALF_LOG_2PI) - LOG_HALF; dLpDu = -u; }
      else if (sp.prior_dist_for_aux == 2) {
        const double nu = sp.prior_df_for_aux;
        lp += std::log(u * u / nu + 1.0) * (-(nu + 1.0) / 2.0) + cfAuxConst_ - LOG_HALF; dLpDu = -(nu + 1.0) * u / (nu + u * u);
      } else { lp -= u; dLpDu = -1.0; }
    }
    // ---- the O(N) part (device kernels or the Gram form)
    double* const gX = cfGX_.data(); double* const gZ = cfGZ_.data();
    for (int k = 0; k < K; ++k) gX[k] = 0.0;
    for (int j = 0; j < nq; ++j) gZ[j] = 0.0;
    const double ss = lik(beta, b, gX, gZ);
    const double s2 = aux * aux, N = (double)sp.N, is2 = 1.0 / s2;
    const double ll = -0.5 * ss / s2 - N * std::log(aux) - N * HALF_LOG_2PI;
    double dAux = ss / (s2 * aux) - N / aux;
    // ---- backward
    if (sp.prior_dist == 0) for (int k = 0; k < K; ++k) g[k] = gX[k] * is2;
    else for (int k = 0; k < K; ++k) g[k] = gX[k] * is2 * sp.prior_scale[(size_t)k] - q[k];
    for (int i = 0; i < t; ++i) {
      const CfTerm& c = cfTerms_[(size_t)i];
      const double* zb = q + c.zb0; const double* gb = gZ + (c.zb0 - K); double* gz = g + c.zb0;
      const CfWork& w = cfWork_[(size_t)i];
      const double tau = w.tau, A = w.A;
      double dA = 0.0;
      if (c.p == 1) {
        double dTh = 0.0;
        for (int s = 0; s < c.l; ++s) { const double db = gb[s] * is2; dTh += db * zb[s]; gz[s] = db * A - zb[s]; }
        dA = dTh;
      } else {
        const double T00 = th[3 * i], T10 = th[3 * i + 1], T11 = th[3 * i + 2];
        double d00 = 0.0, d10 = 0.0, d11 = 0.0;
        for (int j = 0; j < c.l; ++j) {
          const double a0 = zb[2 * j], a1 = zb[2 * j + 1], db0 = gb[2 * j] * is2, db1 = gb[2 * j + 1] * is2;
          d00 += db0 * a0; d10 += db1 * a0; d11 += db1 * a1;
          gz[2 * j] = db0 * T00 + db1 * T10 - a0; gz[2 * j + 1] = db1 * T11 - a1;
        }
        const double sg = q[c.rho] >= 0 ? 1.0 : -1.0, e = w.e, rho = w.rho, z0 = w.z0, z1 = w.z1;
        const double trace = w.trace, S = w.S, pi0 = w.pi0, pi1 = w.pi1, sd0 = T00, sd1 = w.sd1, T21 = w.T21, sq = w.sq;
        const double dsd0 = d00, dsd1 = d10 * T21 + d11 * sq, dT21 = d10 * sd1 - d11 * sd1 * (T21 / sq);
        const double dpi0 = dsd0 * trace * (0.5 / sd0), dpi1 = dsd1 * trace * (0.5 / sd1);
        const double dtrace = dsd0 * pi0 * (0.5 / sd0) + dsd1 * pi1 * (0.5 / sd1);
        const double dS = -(dpi0 * pi0 + dpi1 * pi1) / S;
        double dz0 = dpi0 / S + dS, dz1 = dpi1 / S + dS;
        dA = dtrace * 4.0 * A;
        double drho = 2.0 * dT21 + (c.betaA - 1.0) / rho - (c.betaA - 1.0) / (1.0 - rho);
        g[c.rho] = drho * rho * (1.0 - rho) + (-sg + 2.0 * sg * e / (1.0 + e));
        g[c.zeta] = dz0 * z0 + 1.0 + (c.delta0 - 1.0) - z0;
        g[c.zeta + 1] = dz1 * z1 + 1.0 + (c.delta1 - 1.0) - z1;
      }
      const double dtau = dA * c.scale * aux;
      dAux += dA * tau * c.scale;
      g[c.tau] = dtau * tau + 1.0 + (c.shape - 1.0) - tau;
    }
    if (!sp.is_binary) g[cfAux_] = (dAux * dAuxDu + dLpDu) * u + 1.0;
    return lp + ll;
  }

  // the log-gamma constants of the priors depend on the data only: computed once, reused by every gradient
  mutable std::vector<std::pair<double, double>> lgMemo_;
  double lg(double x) const {
    for (const auto& e : lgMemo_) if (e.first == x) return e.second;
    const double v = std::lgamma(x);
    if (lgMemo_.size() < 64) lgMemo_.emplace_back(x, v);
    return v;
  }
  mutable std::vector<TV> sGlobal_, sCaux_, sLambda_, sMix_, sT_, sPi_; mutable std::vector<std::vector<TV>> sLocal_;
  mutable std::vector<double> sS1_, sS2_; mutable std::vector<int> sIdx_;
#ifdef S4B_NUTS_TIMING
  double tFwd_ = 0, tLik_ = 0, tBwd_ = 0;
#endif
  Tape tape_; bool tapeHoldsGradient_ = false;
  mutable std::vector<TV> w_[6];
  Fwd fwd_; std::vector<int> qidx_; std::vector<double> bufBeta_, bufB_, bufGX_, bufGZ_;
  static TV cornish_fisher(TV z, double df) {
    TV z2 = tsquare(z), z3 = z2 * z, z5 = z2 * z3, z7 = z2 * z5, z9 = z2 * z7;
    double df2 = df * df, df3 = df2 * df, df4 = df2 * df2;
    return z + (z3 + z) / (4 * df) + (z5 * 5.0 + z3 * 16.0 + z * 3.0) / (96 * df2) +
           (z7 * 3.0 + z5 * 19.0 + z3 * 17.0 - z * 15.0) / (384 * df3) +
           (z9 * 79.0 + z7 * 776.0 + z5 * 1482.0 - z3 * 1920.0 - z * 945.0) / (92160 * df4);
  }
  void theta_L(Tape& tp, TV dispersion, const std::vector<TV>& tau, const std::vector<TV>& zeta, const std::vector<TV>& rho,
               const std::vector<TV>& z_T, std::vector<TV>& out) const {
    out.clear();
    int zeta_mark = 0, rho_mark = 0, z_T_mark = 0;
    for (int i = 0; i < sp.t; ++i) {
      int nc = sp.p[(size_t)i];
      TV A = tau[(size_t)i] * sp.scale[(size_t)i] * dispersion;
      if (nc == 1) { out.push_back(A); continue; }
      TV zero{&tp, tp.leaf(0.0)};
      std::vector<TV>& T = sT_; T.assign((size_t)(nc * nc), zero);
      auto at = [&](int r, int c) -> TV& { return T[(size_t)(r * nc + c)]; };
      TV trace = tsquare(A) * (double)nc;
      TV sum_pi = zeta[(size_t)zeta_mark];
      for (int j = 1; j < nc; ++j) sum_pi = sum_pi + zeta[(size_t)(zeta_mark + j)];
      std::vector<TV>& pi = sPi_; pi.clear();
      for (int j = 0; j < nc; ++j) pi.push_back(zeta[(size_t)(zeta_mark + j)] / sum_pi);
      zeta_mark += nc;
      TV std_dev = tsqrt(pi[0] * trace);
      at(0, 0) = std_dev;
      std_dev = tsqrt(pi[1] * trace);
      TV T21 = rho[(size_t)rho_mark] * 2.0 - 1.0;
      rho_mark += 1;
      at(1, 1) = std_dev * tsqrt(1.0 - tsquare(T21));
      at(1, 0) = std_dev * T21;
      for (int r = 2; r <= nc - 1; ++r) {
        TV dot = tsquare(z_T[(size_t)z_T_mark]);
        for (int c = 1; c < r; ++c) dot = dot + tsquare(z_T[(size_t)(z_T_mark + c)]);
        TV scale_factor = tsqrt(rho[(size_t)rho_mark] / dot) * std_dev;
        for (int c = 0; c < r; ++c) at(r, c) = z_T[(size_t)(z_T_mark + c)] * scale_factor;
        z_T_mark += r;
        std_dev = tsqrt(pi[(size_t)r] * trace);
        at(r, r) = tsqrt(1.0 - rho[(size_t)rho_mark]) * std_dev;
        rho_mark += 1;
      }
      for (int c = 0; c < nc; ++c) for (int r = c; r < nc; ++r) out.push_back(at(r, c));
    }
  }
  void make_b(Tape& tp, const std::vector<TV>& z_b, const std::vector<TV>& th, std::vector<TV>& b) const {
    b.clear(); b.reserve((size_t)sp.q);
    int b_mark = 0, tm = 0;
    for (int i = 0; i < sp.t; ++i) {
      int nc = sp.p[(size_t)i];
      if (nc == 1) {
        for (int s = 0; s < sp.l[(size_t)i]; ++s) b.push_back(th[(size_t)tm] * z_b[(size_t)(b_mark + s)]);
        b_mark += sp.l[(size_t)i]; tm += 1;
      } else {
        std::vector<int>& idx = sIdx_; idx.assign((size_t)(nc * nc), -1);   // theta_L index of T[r][c], column-major lower triangle
        for (int c = 0; c < nc; ++c) for (int r = c; r < nc; ++r) idx[(size_t)(r * nc + c)] = tm++;
        for (int j = 0; j < sp.l[(size_t)i]; ++j) {
          for (int r = 0; r < nc; ++r) {
            TV acc = th[(size_t)idx[(size_t)(r * nc)]] * z_b[(size_t)b_mark];
            for (int c = 1; c <= r; ++c) acc = acc + th[(size_t)idx[(size_t)(r * nc + c)]] * z_b[(size_t)(b_mark + c)];
            b.push_back(acc);
          }
          b_mark += nc;
        }
      }
    }
    (void)tp;
  }
};

// ---------------------------------------------------------------- Stan's random stream
// boost::ecuyer1988 + uniform_01 + uniform_real + ziggurat normal (see DESIGN.md "Random streams")
class StanRng {
 public:
  void create(uint32_t seed, uint32_t chain) {
    x1_ = seed % 2147483563u; if (!x1_) x1_ = 1;
    x2_ = seed % 2147483399u; if (!x2_) x2_ = 1;
    uint64_t z = (uint64_t(1) << 50) * chain;
    x1_ = (uint32_t)(mulpow(40014, z, 2147483563ull) * x1_ % 2147483563ull);
    x2_ = (uint32_t)(mulpow(40692, z, 2147483399ull) * x2_ % 2147483399ull);
    build_tables();
  }
  uint32_t raw() {
    x1_ = (uint32_t)(40014ull * x1_ % 2147483563ull);
    x2_ = (uint32_t)(40692ull * x2_ % 2147483399ull);
    return x2_ < x1_ ? x1_ - x2_ : x1_ - x2_ + 2147483562u;
  }
  double u01() { return (double)(raw() - 1u) / 2147483562.0; }
  void get_state(uint32_t out[2]) const { out[0] = x1_; out[1] = x2_; }
  void set_state(const uint32_t in[2]) { x1_ = in[0]; x2_ = in[1]; }
  double uniform(double a, double b) { for (;;) { double r = u01() * (b - a) + a; if (r < b) return r; } }
  double normal() {
    for (;;) {
      double u; int bucket; pair8(u, bucket);
      int sign = (bucket & 1) * 2 - 1, i = bucket >> 1;
      double x = u * nx_[i];
      if (x < nx_[i + 1]) return x * sign;
      if (i == 0) {
        for (;;) { double tx = expo() / nx_[1], ty = expo(); if (2.0 * ty > tx * tx) return (tx + nx_[1]) * sign; }
      }
      double y = ny_[i] + u01() * (ny_[i + 1] - ny_[i]);
      if (y < std::exp(-0.5 * x * x)) return x * sign;
    }
  }
 private:
  uint32_t x1_ = 1, x2_ = 1;
  double nx_[129], ny_[129], ex_[257], ey_[257];
  static uint64_t mulpow(uint64_t a, uint64_t e, uint64_t m) { uint64_t r = 1; a %= m; while (e) { if (e & 1) r = r * a % m; a = a * a % m; e >>= 1; } return r; }
  uint32_t digit30() { uint32_t u; do { u = raw() - 1u; } while (u >= (1u << 30)); return u; }
  void pair8(double& r, int& bucket) {   // 8 integer bits + 53-bit fraction out of three 30-bit digits
    uint32_t a = digit30(), b = digit30(), c = digit30();
    bucket = (int)(a & 255u);
    r = std::ldexp((double)(a >> 8), -22);
    r = std::ldexp(r + (double)b, -30);
    r = 0.5 * (r + (double)(c & 1u));
  }
  double expo() {   // unit exponential, 256-layer ziggurat
    double shift = 0.0;
    for (;;) {
      double u; int i; pair8(u, i);
      double x = u * ex_[i];
      if (x < ex_[i + 1]) return shift + x;
      if (i == 0) { shift += ex_[1]; continue; }
      double y = ey_[i] + u01() * (ey_[i + 1] - ey_[i]);
      if (y < std::exp(-x)) return shift + x;
    }
  }
  void build_tables() {
    const double r = 3.442619855899, v = 9.91256303526217e-3;
    nx_[1] = r; ny_[1] = std::exp(-0.5 * r * r); nx_[0] = v / ny_[1]; ny_[0] = 0;
    for (int i = 2; i < 128; ++i) { ny_[i] = ny_[i - 1] + v / nx_[i - 1]; nx_[i] = std::sqrt(-2.0 * std::log(ny_[i])); }
    nx_[128] = 0; ny_[128] = 1;
    const double re = 7.69711747013104972, ve = 3.949659822581572e-3;
    ex_[1] = re; ey_[1] = std::exp(-re); ex_[0] = ve / ey_[1]; ey_[0] = 0;
    for (int i = 2; i < 256; ++i) { ey_[i] = ey_[i - 1] + ve / ex_[i - 1]; ex_[i] = -std::log(ey_[i]); }
    ex_[256] = 0; ey_[256] = 1;
  }
};

// ---------------------------------------------------------------- NUTS with diagonal metric adaptation
struct NutsControl {
  uint32_t seed = 0; double init_r = 2.0; int skip = 1;
  double gamma = 0.05, delta = 0.8, kappa = 0.75, t0 = 10;
  unsigned init_buffer = 75, term_buffer = 50, window = 25;
  double stepsize = 1, jitter = 0; int max_depth = 10;
};

class Nuts {
 public:
  typedef std::vector<double> V;
  struct Pt { V q, p, g; double Vv = 0; };

  Nuts(HostModel& m, const NutsControl& c, unsigned chain, int num_warmup) : model_(m), D_(m.sp.D), skip_(c.skip) {
    rng_.create(c.seed, chain);
    // services::util::initialize: uniform(-R, R) starts, at most 100 attempts
    bool ok = false;
    for (int attempt = 0; attempt < (c.init_r == 0 ? 1 : 100) && !ok; ++attempt) {
      cont_.assign((size_t)D_, 0.0);
      if (c.init_r != 0) for (int i = 0; i < D_; ++i) cont_[(size_t)i] = rng_.uniform(-c.init_r, c.init_r);
      V g;
      double lp = model_.log_prob_grad(cont_, g);
      if (!std::isfinite(lp)) continue;
      lp = model_.log_prob_grad(cont_, g);
      double s = 0; for (double x : g) s += x;
      ok = std::isfinite(s);
    }
    if (!ok) throw std::domain_error("Initialization failed.");
    inv_metric_.assign((size_t)D_, 1.0);
    z_.q = cont_; z_.p.assign((size_t)D_, 0.0); z_.g.assign((size_t)D_, 0.0);
    wm_.assign((size_t)D_, 0.0); wm2_.assign((size_t)D_, 0.0);
    if (c.stepsize > 0) nom_eps_ = c.stepsize;
    if (c.jitter > 0 && c.jitter < 1) jitter_ = c.jitter;
    if (c.max_depth > 0) max_depth_ = c.max_depth;
    mu_ = std::log(10 * c.stepsize);
    if (c.delta > 0 && c.delta < 1) delta_ = c.delta;
    if (c.gamma > 0) gamma_ = c.gamma;
    if (c.kappa > 0) kappa_ = c.kappa;
    if (c.t0 > 0) t0_ = c.t0;
    next_window_ = init_buffer_ + window_size_ - 1;   // restart() with the zero defaults
    set_windows((unsigned)(num_warmup * c.skip), c.init_buffer, c.term_buffer, c.window);
    init_stepsize();
  }

  // interruptable_sampler::run: skip-1 unsaved transitions then one saved; fills the sample row
  void run(double* row) {
    for (int s = 0; s < skip_; ++s) transition();
    row[0] = lp_; row[1] = accept_; row[2] = eps_; row[3] = depth_; row[4] = n_leapfrog_; row[5] = divergent_ ? 1 : 0; row[6] = energy_;
    model_.write_array(cont_, row + 7);
  }
  void disengage() { adapting_ = false; nom_eps_ = std::exp(x_bar_); }
  // running totals over all transitions since creation: {transitions, sum treedepth__, sum n_leapfrog__, divergent transitions}
  void totals(double out[4]) const { out[0] = (double)nTrans_; out[1] = (double)sumDepth_; out[2] = (double)sumLeap_; out[3] = (double)nDiv_; }

  // Everything that carries over from one transition to the next (s4b_get_state / s4b_set_state: checkpoint / resume and
  // the teacher-forced parity tests).  Momentum, gradient and potential are recomputed at the start of a transition
  // (base_nuts.hpp:85-91), so they are not part of the state.
  struct State {
    V q, inv_metric, wm, wm2;
    double stepsize = 0, mu = 0, counter = 0, s_bar = 0, x_bar = 0, wn = 0;
    int32_t adapting = 1;
    uint32_t window[7] = {0, 0, 0, 0, 0, 0, 0};   // num_warmup, init_buffer, term_buffer, base_window, window_counter, next_window, window_size
    uint32_t rng[2] = {1, 1};
    double last[7] = {0, 0, 0, 0, 0, 0, 0};       // sampler columns of the last emitted row
  };
  void get_state(State& s) const {
    s.q = cont_; s.inv_metric = inv_metric_; s.wm = wm_; s.wm2 = wm2_;
    s.stepsize = nom_eps_; s.mu = mu_; s.counter = counter_; s.s_bar = s_bar_; s.x_bar = x_bar_; s.wn = wn_;
    s.adapting = adapting_ ? 1 : 0;
    const uint32_t w[7] = {num_warmup_, init_buffer_, term_buffer_, base_window_, window_counter_, next_window_, window_size_};
    for (int i = 0; i < 7; ++i) s.window[i] = w[i];
    rng_.get_state(s.rng);
    const double l[7] = {lp_, accept_, eps_, (double)depth_, (double)n_leapfrog_, divergent_ ? 1.0 : 0.0, energy_};
    for (int i = 0; i < 7; ++i) s.last[i] = l[i];
  }
  void set_state(const State& s) {
    if ((int)s.q.size() != D_ || (int)s.inv_metric.size() != D_ || (int)s.wm.size() != D_ || (int)s.wm2.size() != D_)
      throw std::invalid_argument("sampler state: wrong number of unconstrained parameters");
    cont_ = s.q; z_.q = s.q; inv_metric_ = s.inv_metric; wm_ = s.wm; wm2_ = s.wm2;
    nom_eps_ = s.stepsize; mu_ = s.mu; counter_ = s.counter; s_bar_ = s.s_bar; x_bar_ = s.x_bar; wn_ = s.wn;
    adapting_ = s.adapting != 0;
    num_warmup_ = s.window[0]; init_buffer_ = s.window[1]; term_buffer_ = s.window[2]; base_window_ = s.window[3];
    window_counter_ = s.window[4]; next_window_ = s.window[5]; window_size_ = s.window[6];
    rng_.set_state(s.rng);
    lp_ = s.last[0]; accept_ = s.last[1]; eps_ = s.last[2]; depth_ = (int)s.last[3]; n_leapfrog_ = (int)s.last[4];
    divergent_ = s.last[5] != 0.0; energy_ = s.last[6];
  }
  // the sample row of the current point (sampler columns as last emitted)
  void current_row(double* row) {
    row[0] = lp_; row[1] = accept_; row[2] = eps_; row[3] = depth_; row[4] = n_leapfrog_; row[5] = divergent_ ? 1 : 0; row[6] = energy_;
    model_.write_array(cont_, row + 7);
  }

 private:
  HostModel& model_; int D_, skip_;
  StanRng rng_;
  Pt z_; V inv_metric_, cont_;
  double nom_eps_ = 0.1, eps_ = 0.1, jitter_ = 0; int max_depth_ = 10;
  double lp_ = 0, accept_ = 0, energy_ = 0; int depth_ = 0, n_leapfrog_ = 0; bool divergent_ = false, adapting_ = true;
  double mu_ = 0.5, delta_ = 0.5, gamma_ = 0.05, kappa_ = 0.75, t0_ = 10, counter_ = 0, s_bar_ = 0, x_bar_ = 0;
  unsigned num_warmup_ = 0, init_buffer_ = 0, term_buffer_ = 0, base_window_ = 0, window_counter_ = 0, next_window_ = 0, window_size_ = 0;
  double wn_ = 0; V wm_, wm2_;
  long nTrans_ = 0, sumDepth_ = 0, sumLeap_ = 0, nDiv_ = 0;

  double kinetic() const { double s = 0; for (int i = 0; i < D_; ++i) s += z_.p[(size_t)i] * (inv_metric_[(size_t)i] * z_.p[(size_t)i]); return 0.5 * s; }
  double hamiltonian() const { return kinetic() + z_.Vv; }
  void grad() { z_.Vv = -model_.log_prob_grad(z_.q, z_.g); for (double& x : z_.g) x = -x; }
  void draw_momentum() { for (int i = 0; i < D_; ++i) z_.p[(size_t)i] = rng_.normal() / std::sqrt(inv_metric_[(size_t)i]); }
  void leapfrog(double e) {
    for (int i = 0; i < D_; ++i) z_.p[(size_t)i] -= (0.5 * e) * z_.g[(size_t)i];
    for (int i = 0; i < D_; ++i) z_.q[(size_t)i] += e * (inv_metric_[(size_t)i] * z_.p[(size_t)i]);
    grad();
    for (int i = 0; i < D_; ++i) z_.p[(size_t)i] -= (0.5 * e) * z_.g[(size_t)i];
  }
  V sharp() const { V r((size_t)D_); for (int i = 0; i < D_; ++i) r[(size_t)i] = inv_metric_[(size_t)i] * z_.p[(size_t)i]; return r; }
  static double finite_or_inf(double h) { return std::isnan(h) ? std::numeric_limits<double>::infinity() : h; }

  void init_stepsize() {
    Pt z0 = z_;
    if (nom_eps_ == 0 || nom_eps_ > 1e7 || std::isnan(nom_eps_)) return;
    const double thr = std::log(0.8);
    draw_momentum(); grad();
    double H0 = hamiltonian();
    leapfrog(nom_eps_);
    double dH = H0 - finite_or_inf(hamiltonian());
    int dir = dH > thr ? 1 : -1;
    for (;;) {
      z_ = z0;
      draw_momentum(); grad();
      double H0b = hamiltonian();
      leapfrog(nom_eps_);
      double d = H0b - finite_or_inf(hamiltonian());
      if (dir == 1 && !(d > thr)) break;
      if (dir == -1 && !(d < thr)) break;
      nom_eps_ = dir == 1 ? 2 * nom_eps_ : 0.5 * nom_eps_;
      if (nom_eps_ > 1e7) throw std::runtime_error("Posterior is improper. Please check your model.");
      if (nom_eps_ == 0) throw std::runtime_error("No acceptably small step size could be found. Perhaps the posterior is not continuous?");
    }
    z_ = z0;
  }

  static double lse(double a, double b) {
    const double ninf = -std::numeric_limits<double>::infinity();
    if (a == ninf) return b;
    if (b == ninf) return a;
    return a > b ? a + std::log1p(std::exp(b - a)) : b + std::log1p(std::exp(a - b));
  }
  static bool uturn_ok(const V& sharp_minus, const V& sharp_plus, const V& rho) {
    double a = 0, b = 0;
    for (size_t i = 0; i < rho.size(); ++i) { a += sharp_plus[i] * rho[i]; b += sharp_minus[i] * rho[i]; }
    return a > 0 && b > 0;
  }
  static V vsum(const V& a, const V& b) { V r(a.size()); for (size_t i = 0; i < a.size(); ++i) r[i] = a[i] + b[i]; return r; }

  struct Acc { int n_leapfrog = 0; double sum_metro = 0; };

  // A point a subtree proposes: position, potential and Hamiltonian.  (The reference copies the whole phase-space point,
  // base_nuts.hpp:283; what is read of a proposal afterwards is its position, its potential (lp__) and H (energy__), and H of a point
  // is the value the leaf computed from the same p and V.)
  struct Prop { V q; double Vv = 0, H = 0; };
  // work space of one recursion level of subtree(): allocated once (a level is never active twice at the same time), so that a leapfrog
  // costs no heap traffic — about a thousand of them per Gibbs iteration once the chain runs deep trees
  struct Level { V p_init_end, sharp_init_end, rho_init, p_final_beg, sharp_final_beg, rho_final; Prop propose_final; };
  std::vector<Level> lv_;
  void ensure_levels() {
    if ((int)lv_.size() == max_depth_ + 1) return;
    lv_.resize((size_t)max_depth_ + 1);
    for (Level& l : lv_)
      for (V* v : {&l.p_init_end, &l.sharp_init_end, &l.rho_init, &l.p_final_beg, &l.sharp_final_beg, &l.rho_final, &l.propose_final.q}) v->assign((size_t)D_, 0.0);
  }

  // base_nuts::build_tree (base_nuts.hpp:247-352).  Every quantity is computed by the same operations in the same order as the
  // straightforward transcription (per-element arithmetic; dot products summed from element 0 upwards) — the loops are fused, and the six
  // dot products of the three U-turn criteria of a merge run side by side (six independent chains of additions instead of six loops that
  // each wait for their own previous addition): the chains of a sampler are unchanged bit for bit.
  bool subtree(int depth, Prop& propose, V& sharp_beg, V& sharp_end, V& rho, V& p_beg, V& p_end, double H0, double sign,
               double& lsw, Acc& acc) {
    const double ninf = -std::numeric_limits<double>::infinity();
    const int D = D_;
    if (depth == 0) {
      const double e = sign * eps_, he = 0.5 * e;
      {
        double* __restrict__ const p = z_.p.data(); double* __restrict__ const q = z_.q.data();
        const double* __restrict__ const g = z_.g.data(); const double* __restrict__ const im = inv_metric_.data();
        for (int i = 0; i < D; ++i) { const double pi = p[i] - he * g[i]; p[i] = pi; q[i] += e * (im[i] * pi); }
      }
      grad();
      ++acc.n_leapfrog;
      double kin = 0.0;
      {
        double* __restrict__ const p = z_.p.data(); const double* __restrict__ const g = z_.g.data(); const double* __restrict__ const im = inv_metric_.data();
        double* __restrict__ const sb = sharp_beg.data(); double* __restrict__ const se = sharp_end.data(); double* __restrict__ const rh = rho.data();
        double* __restrict__ const pb = p_beg.data(); double* __restrict__ const pe = p_end.data();
        for (int i = 0; i < D; ++i) {
          const double pi = p[i] - he * g[i], sh = im[i] * pi;
          p[i] = pi; kin += pi * sh; sb[i] = sh; se[i] = sh; rh[i] += pi; pb[i] = pi; pe[i] = pi;
        }
      }
      const double hRaw = 0.5 * kin + z_.Vv, h = finite_or_inf(hRaw);
      if (h - H0 > 1000.0) divergent_ = true;
      lsw = lse(lsw, H0 - h);
      acc.sum_metro += (H0 - h > 0) ? 1.0 : std::exp(H0 - h);
      std::copy(z_.q.begin(), z_.q.end(), propose.q.begin()); propose.Vv = z_.Vv; propose.H = hRaw;
      return !divergent_;
    }
    Level& L = lv_[(size_t)depth];
    double lsw_init = ninf;
    std::fill(L.rho_init.begin(), L.rho_init.end(), 0.0);
    if (!subtree(depth - 1, propose, sharp_beg, L.sharp_init_end, L.rho_init, p_beg, L.p_init_end, H0, sign, lsw_init, acc)) return false;
    double lsw_final = ninf;
    std::fill(L.rho_final.begin(), L.rho_final.end(), 0.0);
    if (!subtree(depth - 1, L.propose_final, L.sharp_final_beg, sharp_end, L.rho_final, L.p_final_beg, p_end, H0, sign, lsw_final, acc)) return false;
    double lsw_sub = lse(lsw_init, lsw_final);
    lsw = lse(lsw, lsw_sub);
    if (lsw_final > lsw_sub) std::swap(propose, L.propose_final);      // (the level's slot is overwritten before it is read again)
    else if (rng_.u01() < std::exp(lsw_final - lsw_sub)) std::swap(propose, L.propose_final);
    // rho_sub = rho_init + rho_final; rho += rho_sub; the three criteria (base_nuts.hpp:339-351): the whole subtree, its first half
    // extended by the first momentum of the second, its second half extended by the last momentum of the first
    double a1 = 0, b1 = 0, a2 = 0, b2 = 0, a3 = 0, b3 = 0;
    {
      const double* __restrict__ const ri = L.rho_init.data(); const double* __restrict__ const rf = L.rho_final.data();
      const double* __restrict__ const pfb = L.p_final_beg.data(); const double* __restrict__ const pie = L.p_init_end.data();
      const double* __restrict__ const sb = sharp_beg.data(); const double* __restrict__ const se = sharp_end.data();
      const double* __restrict__ const sfb = L.sharp_final_beg.data(); const double* __restrict__ const sie = L.sharp_init_end.data();
      double* __restrict__ const rh = rho.data();
      for (int i = 0; i < D; ++i) {
        const double rs = ri[i] + rf[i], t2 = ri[i] + pfb[i], t3 = rf[i] + pie[i];
        rh[i] += rs;
        a1 += se[i] * rs; b1 += sb[i] * rs;
        a2 += sfb[i] * t2; b2 += sb[i] * t2;
        a3 += se[i] * t3; b3 += sie[i] * t3;
      }
    }
    return (a1 > 0 && b1 > 0) & (a2 > 0 && b2 > 0) & (a3 > 0 && b3 > 0);
  }

  void nuts_transition() {
    const double ninf = -std::numeric_limits<double>::infinity();
    eps_ = nom_eps_;
    if (jitter_) eps_ *= 1.0 + jitter_ * (2.0 * rng_.u01() - 1.0);
    ensure_levels();
    z_.q = cont_;
    draw_momentum(); grad();
    const double H0 = hamiltonian();
    Pt& fwd = tFwd_; Pt& bck = tBck_; fwd = z_; bck = z_;
    Prop& sample = tSample_; Prop& propose = tPropose_;
    sample.q = z_.q; sample.Vv = z_.Vv; sample.H = H0; propose = sample;
    V& p_ff = tV_[0]; V& s_ff = tV_[1]; V& p_fb = tV_[2]; V& s_fb = tV_[3]; V& p_bf = tV_[4]; V& s_bf = tV_[5]; V& p_bb = tV_[6]; V& s_bb = tV_[7];
    V& rho = tV_[8]; V& rho_f = tV_[9]; V& rho_b = tV_[10];
    p_ff = z_.p; s_ff = sharp(); p_fb = z_.p; s_fb = s_ff; p_bf = z_.p; s_bf = s_ff; p_bb = z_.p; s_bb = s_ff;
    rho = z_.p;
    double lsw = 0;
    Acc acc;
    depth_ = 0; divergent_ = false;
    while (depth_ < max_depth_) {
      rho_f.assign((size_t)D_, 0.0); rho_b.assign((size_t)D_, 0.0);
      bool valid; double lsw_sub = ninf;
      if (rng_.u01() > 0.5) {
        z_ = fwd; rho_b = rho; p_bf = p_ff; s_bf = s_ff;
        valid = subtree(depth_, propose, s_fb, s_ff, rho_f, p_fb, p_ff, H0, 1, lsw_sub, acc);
        fwd = z_;
      } else {
        z_ = bck; rho_f = rho; p_fb = p_bb; s_fb = s_bb;
        valid = subtree(depth_, propose, s_bf, s_bb, rho_b, p_bf, p_bb, H0, -1, lsw_sub, acc);
        bck = z_;
      }
      if (!valid) break;
      ++depth_;
      if (lsw_sub > lsw) sample = propose;
      else if (rng_.u01() < std::exp(lsw_sub - lsw)) sample = propose;
      lsw = lse(lsw, lsw_sub);
      rho = vsum(rho_b, rho_f);
      bool ok = uturn_ok(s_bb, s_ff, rho);
      ok &= uturn_ok(s_bb, s_fb, vsum(rho_b, p_fb));
      ok &= uturn_ok(s_bf, s_ff, vsum(rho_f, p_bf));
      if (!ok) break;
    }
    n_leapfrog_ = acc.n_leapfrog;
    ++nTrans_; sumDepth_ += depth_; sumLeap_ += acc.n_leapfrog; if (divergent_) ++nDiv_;
    accept_ = acc.sum_metro / (double)acc.n_leapfrog;
    z_.q = sample.q; z_.Vv = sample.Vv;     // (momentum and gradient are redrawn / recomputed when the next transition starts)
    energy_ = sample.H;
    cont_ = z_.q; lp_ = -z_.Vv;
  }
  Pt tFwd_, tBck_; Prop tSample_, tPropose_; V tV_[11];

  void transition() {
    nuts_transition();
    if (!adapting_) return;
    // dual averaging
    ++counter_;
    double a = accept_ > 1 ? 1 : accept_;
    double eta = 1.0 / (counter_ + t0_);
    s_bar_ = (1.0 - eta) * s_bar_ + eta * (delta_ - a);
    double x = mu_ - s_bar_ * std::sqrt(counter_) / gamma_;
    double x_eta = std::pow(counter_, -kappa_);
    x_bar_ = (1.0 - x_eta) * x_bar_ + x_eta * x;
    nom_eps_ = std::exp(x);
    // windowed variance
    bool in_window = window_counter_ >= init_buffer_ && window_counter_ < num_warmup_ - term_buffer_ && window_counter_ != num_warmup_;
    if (in_window) {
      ++wn_;
      for (int i = 0; i < D_; ++i) { double d = z_.q[(size_t)i] - wm_[(size_t)i]; wm_[(size_t)i] += d / wn_; wm2_[(size_t)i] += d * (z_.q[(size_t)i] - wm_[(size_t)i]); }
    }
    bool window_end = window_counter_ == next_window_ && window_counter_ != num_warmup_;
    if (window_end) {
      next_window();
      for (int i = 0; i < D_; ++i) {
        double var = wn_ > 1 ? wm2_[(size_t)i] / (wn_ - 1.0) : inv_metric_[(size_t)i];
        inv_metric_[(size_t)i] = (wn_ / (wn_ + 5.0)) * var + 1e-3 * (5.0 / (wn_ + 5.0));
      }
      wn_ = 0; wm_.assign((size_t)D_, 0.0); wm2_.assign((size_t)D_, 0.0);
      ++window_counter_;
      init_stepsize();
      mu_ = std::log(10 * nom_eps_);
      counter_ = 0; s_bar_ = 0; x_bar_ = 0;
      return;
    }
    ++window_counter_;
  }
  void set_windows(unsigned num_warmup, unsigned init_buffer, unsigned term_buffer, unsigned base_window) {
    if (num_warmup < 20) return;
    if (init_buffer + base_window + term_buffer > num_warmup) {
      num_warmup_ = num_warmup;
      init_buffer_ = (unsigned)(0.15 * num_warmup);
      term_buffer_ = (unsigned)(0.10 * num_warmup);
      base_window_ = num_warmup - (init_buffer_ + term_buffer_);
      return;   // the vendored Stan does not restart() here (windowed_adaptation.hpp:45-74)
    }
    num_warmup_ = num_warmup; init_buffer_ = init_buffer; term_buffer_ = term_buffer; base_window_ = base_window;
    window_counter_ = 0; window_size_ = base_window_; next_window_ = init_buffer_ + window_size_ - 1;
  }
  void next_window() {
    if (next_window_ == num_warmup_ - term_buffer_ - 1) return;
    window_size_ *= 2;
    next_window_ = window_counter_ + window_size_;
    if (next_window_ == num_warmup_ - term_buffer_ - 1) return;
    unsigned boundary = next_window_ + 2 * window_size_;
    if (boundary >= num_warmup_ - term_buffer_) next_window_ = num_warmup_ - term_buffer_ - 1;
  }
};

}  // namespace s4b
#endif
