// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the reference's Stan block:
//   model      src/stan_files/continuous.stan (whole file), cross-checked with
//              src/stan_files/continuous.hpp:2168-2638 (log_prob_impl), :2640-2938 (write_array_impl),
//              :3626-3768 (set_response / set_offset / get_aux / get_parametric_mean)
//   NUTS       src/include/stan/mcmc/hmc/nuts/base_nuts.hpp:78-352, adapt_diag_e_nuts.hpp:25-49,
//              hmc/base_hmc.hpp:81-143 (init_stepsize), hamiltonians/diag_e_metric.hpp:20-50,
//              integrators/{base,expl}_leapfrog.hpp
//   adaptation src/include/stan/mcmc/{stepsize_adaptation,var_adaptation,windowed_adaptation}.hpp,
//              math/prim/fun/welford_var_estimator.hpp
//   driver     src/interruptable_sampler.hpp:118-210, services/util/{initialize,generate_transitions,
//              create_rng}.hpp, io/random_var_context.hpp:43-82
// Reverse-mode AD is replaced by: forward-mode dual numbers over the O(D) parameter transforms
// (so the Stan program is followed line by line) + the closed-form adjoint of the Gaussian
// likelihood for the O(N) part.  Supported prior families: prior_dist in {0 none, 1 normal,
// 2 student_t (Cornish-Fisher), 3 hs, 4 hs_plus, 5 laplace, 6 lasso, 7 product_normal}, prior_dist_for_aux in
// {0,1,2,3}, decov covariance prior.
#ifndef ORACLE_STAN_REF_HPP
#define ORACLE_STAN_REF_HPP

#include <vector>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <string>
#include "boost_rng.hpp"

namespace oracle {

// ------------------------------------------------------------------ forward-mode dual
struct Dual {
  double v = 0.0;
  std::vector<double> d;
  Dual() {}
  explicit Dual(size_t D) : v(0.0), d(D, 0.0) {}
  Dual(double v_, size_t D) : v(v_), d(D, 0.0) {}
  size_t D() const { return d.size(); }
};
inline Dual cst(double c, size_t D) { return Dual(c, D); }
inline Dual operator+(const Dual& a, const Dual& b) { Dual r(a.v + b.v, a.D()); for (size_t i = 0; i < r.D(); ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
inline Dual operator-(const Dual& a, const Dual& b) { Dual r(a.v - b.v, a.D()); for (size_t i = 0; i < r.D(); ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
inline Dual operator*(const Dual& a, const Dual& b) { Dual r(a.v * b.v, a.D()); for (size_t i = 0; i < r.D(); ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
inline Dual operator/(const Dual& a, const Dual& b) { Dual r(a.v / b.v, a.D()); for (size_t i = 0; i < r.D(); ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v; return r; }
inline Dual operator+(const Dual& a, double c) { Dual r = a; r.v += c; return r; }
inline Dual operator-(const Dual& a, double c) { Dual r = a; r.v -= c; return r; }
inline Dual operator*(const Dual& a, double c) { Dual r = a; r.v *= c; for (double& x : r.d) x *= c; return r; }
inline Dual operator*(double c, const Dual& a) { return a * c; }
inline Dual operator/(const Dual& a, double c) { return a * (1.0 / c); }
inline Dual operator-(double c, const Dual& a) { Dual r(c - a.v, a.D()); for (size_t i = 0; i < r.D(); ++i) r.d[i] = -a.d[i]; return r; }
inline Dual operator-(const Dual& a) { return 0.0 - a; }
inline Dual unary(const Dual& a, double f, double df) { Dual r(f, a.D()); for (size_t i = 0; i < r.D(); ++i) r.d[i] = df * a.d[i]; return r; }
inline Dual sqrt(const Dual& a) { double s = std::sqrt(a.v); return unary(a, s, 0.5 / s); }
inline Dual exp(const Dual& a) { double e = std::exp(a.v); return unary(a, e, e); }
inline Dual log(const Dual& a) { return unary(a, std::log(a.v), 1.0 / a.v); }
inline Dual square(const Dual& a) { return unary(a, a.v * a.v, 2.0 * a.v); }
inline Dual log1m(const Dual& a) { return unary(a, std::log1p(-a.v), -1.0 / (1.0 - a.v)); }

// ------------------------------------------------------------------ model
struct StanData {
  int64_t N = 0; int K = 0;
  std::vector<double> X;           // N x K column-major
  std::vector<double> y;
  int is_binary = 0, has_intercept = 0;
  int prior_dist = 1, prior_dist_for_aux = 3;
  std::vector<double> prior_scale, prior_mean, prior_df;
  double prior_scale_for_aux = 1.0, prior_mean_for_aux = 0.0, prior_df_for_aux = 1.0;
  int t = 0; std::vector<int> p, l; int q = 0; int len_theta_L = 0;
  std::vector<double> shape, scale, concentration, regularization;
  std::vector<double> w; std::vector<int> v; std::vector<int> u;   // CSR of Z
  int has_weights = 0; std::vector<double> weights;
  double global_prior_df = 1, global_prior_scale = 1, slab_df = 1, slab_scale = 1;
  std::vector<int> num_normals;
  std::vector<double> offset_;
};

class StanModel {
 public:
  StanData dat;
  int len_z_T = 0, len_rho = 0, len_conc = 0;
  int hs = 0, n_z_beta = 0, n_mix = 0, n_lambda = 0;   // extra blocks of the hs / laplace / lasso / product_normal priors
  std::vector<double> delta;
  int D = 0;
  int n_constrained = 0, n_row = 0;   // write_array length; +7 sampler columns
  mutable long gradEvals = 0;

  explicit StanModel(const StanData& d) : dat(d) {
    if (dat.has_intercept) throw std::invalid_argument("has_intercept = 1 is not supported (BART supplies the intercept)");
    if (dat.prior_dist < 0 || dat.prior_dist > 7) throw std::invalid_argument("prior_dist must be in 0..7");
    hs = dat.prior_dist == 3 ? 2 : (dat.prior_dist == 4 ? 4 : 0);
    if (hs && dat.is_binary) throw std::invalid_argument("hs priors scale with the residual sd: not available for binary responses");
    n_z_beta = dat.K;
    if (dat.prior_dist == 7) { n_z_beta = 0; for (int k = 0; k < dat.K; ++k) n_z_beta += dat.num_normals[(size_t)k]; }
    n_mix = (dat.prior_dist == 5 || dat.prior_dist == 6) ? dat.K : 0;
    n_lambda = dat.prior_dist == 6 ? 1 : 0;
    int sum_p = 0;
    for (int i = 0; i < dat.t; ++i) {
      sum_p += dat.p[i];
      if (dat.p[i] > 1) for (int j = 0; j < dat.p[i]; ++j) delta.push_back(dat.concentration[j]);
      for (int j = 3; j <= dat.p[i]; ++j) len_z_T += dat.p[i] - 1;
    }
    len_rho = sum_p - dat.t;
    len_conc = (int)delta.size();
    D = n_z_beta + hs + hs * dat.K + (hs > 0 ? 1 : 0) + n_mix + n_lambda + dat.q + len_z_T + len_rho + len_conc + dat.t + (dat.is_binary ? 0 : 1);
    n_constrained = D + (dat.is_binary ? 0 : 1) + dat.K + dat.q + dat.len_theta_L;
    n_row = 7 + n_constrained;
    if ((int64_t)dat.offset_.size() != dat.N) dat.offset_.assign(dat.N, 0.0);
  }
  void set_offset(const double* o) { for (int64_t i = 0; i < dat.N; ++i) dat.offset_[i] = o[i]; }
  void set_response(const double* r) { for (int64_t i = 0; i < dat.N; ++i) dat.y[i] = r[i]; }

  // offsets inside the constrained part of a sample row (after the 7 sampler columns)
  int aux_pos() const { return D; }                               // aux.1 (continuous only)
  int beta_pos() const { return D + (dat.is_binary ? 0 : 1); }
  int b_pos() const { return beta_pos() + dat.K; }
  int theta_L_pos() const { return b_pos() + dat.q; }

  struct Transformed { Dual sigma; std::vector<Dual> beta, b, theta_L; std::vector<Dual> constrained; };

  // unconstrained -> constrained + transformed, accumulating the log-Jacobian
  void transform(const std::vector<double>& qv, Transformed& T, Dual& lp, bool jacobian) const {
    const size_t Dn = (size_t)D;
    auto var = [&](int idx) { Dual x(qv[(size_t)idx], Dn); x.d[(size_t)idx] = 1.0; return x; };
    int pos = 0;
    std::vector<Dual> z_beta, z_b, z_T, rho, zeta, tau;
    for (int k = 0; k < n_z_beta; ++k) z_beta.push_back(var(pos++));
    // lower-bounded blocks of the shrinkage priors (continuous.stan:266-270; arrays of vectors are read array-major)
    auto lb0e = [&](int idx) { Dual x = var(idx); if (jacobian) lp = lp + x; return exp(x); };
    std::vector<Dual> global, caux, one_over_lambda; std::vector<std::vector<Dual>> local((size_t)hs), mix(n_mix ? 1 : 0);
    for (int j = 0; j < hs; ++j) global.push_back(lb0e(pos++));
    for (int j = 0; j < hs; ++j) for (int k = 0; k < dat.K; ++k) local[(size_t)j].push_back(lb0e(pos++));
    if (hs > 0) caux.push_back(lb0e(pos++));
    for (int k = 0; k < n_mix; ++k) mix[0].push_back(lb0e(pos++));
    if (n_lambda) one_over_lambda.push_back(lb0e(pos++));
    for (int j = 0; j < dat.q; ++j) z_b.push_back(var(pos++));
    for (int j = 0; j < len_z_T; ++j) z_T.push_back(var(pos++));
    for (int j = 0; j < len_rho; ++j) {   // lub_constrain(x, 0, 1, lp)
      Dual x = var(pos++);
      double ax = std::fabs(x.v), sg = x.v >= 0 ? 1.0 : -1.0;
      double il = 1.0 / (1.0 + std::exp(-x.v));
      rho.push_back(unary(x, il, il * (1.0 - il)));
      if (jacobian) {   // -|x| - 2 log1p(exp(-|x|))
        double e = std::exp(-ax);
        lp = lp + unary(x, -ax - 2.0 * std::log1p(e), -sg + 2.0 * sg * e / (1.0 + e));
      }
    }
    auto lb0 = [&](int idx) { Dual x = var(idx); if (jacobian) lp = lp + x; return exp(x); };
    for (int j = 0; j < len_conc; ++j) zeta.push_back(lb0(pos++));
    for (int j = 0; j < dat.t; ++j) tau.push_back(lb0(pos++));
    Dual aux_unscaled = cst(0.0, Dn);
    if (!dat.is_binary) aux_unscaled = lb0(pos++);

    // transformed parameters (continuous.stan:270-346)
    Dual aux = cst(1.0, Dn);
    if (!dat.is_binary) {
      if (dat.prior_dist_for_aux == 0) aux = aux_unscaled;
      else {
        aux = aux_unscaled * dat.prior_scale_for_aux;
        if (dat.prior_dist_for_aux <= 2) aux = aux + dat.prior_mean_for_aux;
      }
    }
    T.sigma = aux;
    T.beta.clear();
    if (dat.prior_dist <= 2) for (int k = 0; k < dat.K; ++k) {
      if (dat.prior_dist == 0) T.beta.push_back(z_beta[k]);
      else if (dat.prior_dist == 1) T.beta.push_back(z_beta[k] * dat.prior_scale[k] + dat.prior_mean[k]);
      else T.beta.push_back(CFt(z_beta[k], dat.prior_df[k]) * dat.prior_scale[k] + dat.prior_mean[k]);
    } else if (hs > 0) {   // hs_prior / hsplus_prior (continuous.stan:124-144), error_scale = aux
      Dual c2 = caux[0] * (dat.slab_scale * dat.slab_scale);
      Dual tauG = global[0] * sqrt(global[1]) * dat.global_prior_scale * aux;
      for (int k = 0; k < dat.K; ++k) {
        Dual lam = local[0][(size_t)k] * sqrt(local[1][(size_t)k]);
        if (hs == 4) lam = lam * (local[2][(size_t)k] * sqrt(local[3][(size_t)k]));
        Dual lam2 = square(lam);
        Dual tilde = sqrt(c2 * lam2 / (c2 + square(tauG) * lam2));
        T.beta.push_back(z_beta[(size_t)k] * tilde * tauG);
      }
    } else if (dat.prior_dist == 5) {
      for (int k = 0; k < dat.K; ++k) T.beta.push_back(sqrt(mix[0][(size_t)k] * 2.0) * dat.prior_scale[(size_t)k] * z_beta[(size_t)k] + dat.prior_mean[(size_t)k]);
    } else if (dat.prior_dist == 6) {
      for (int k = 0; k < dat.K; ++k)
        T.beta.push_back(one_over_lambda[0] * dat.prior_scale[(size_t)k] * sqrt(mix[0][(size_t)k] * 2.0) * z_beta[(size_t)k] + dat.prior_mean[(size_t)k]);
    } else {               // product_normal
      int zp = 0;
      for (int k = 0; k < dat.K; ++k) {
        Dual bk = z_beta[(size_t)zp++];
        for (int n2 = 2; n2 <= dat.num_normals[(size_t)k]; ++n2) bk = bk * z_beta[(size_t)zp++];
        T.beta.push_back(bk * std::pow(dat.prior_scale[(size_t)k], (double)dat.num_normals[(size_t)k]) + dat.prior_mean[(size_t)k]);
      }
    }
    make_theta_L(aux, tau, zeta, rho, z_T, T.theta_L);
    make_b(z_b, T.theta_L, T.b);

    T.constrained.clear();
    for (auto& x : z_beta) T.constrained.push_back(x);
    for (auto& x : global) T.constrained.push_back(x);
    for (int k = 0; k < dat.K; ++k) for (int j = 0; j < hs; ++j) T.constrained.push_back(local[(size_t)j][(size_t)k]);   // written vector-index-major
    for (auto& x : caux) T.constrained.push_back(x);
    for (int k = 0; k < n_mix; ++k) T.constrained.push_back(mix[0][(size_t)k]);
    for (auto& x : one_over_lambda) T.constrained.push_back(x);
    for (auto& x : z_b) T.constrained.push_back(x);
    for (auto& x : z_T) T.constrained.push_back(x);
    for (auto& x : rho) T.constrained.push_back(x);
    for (auto& x : zeta) T.constrained.push_back(x);
    for (auto& x : tau) T.constrained.push_back(x);
    if (!dat.is_binary) T.constrained.push_back(aux_unscaled);

    // priors that only involve the O(D) parameters (continuous.stan:368-428)
    const double NEG_LOG_SQRT_TWO_PI = -0.91893853320467274178;
    if (!dat.is_binary && dat.prior_dist_for_aux > 0 && dat.prior_scale_for_aux > 0) {
      const double log_half = -0.693147180559945286;
      if (dat.prior_dist_for_aux == 1) lp = lp + (square(aux_unscaled) * -0.5 + NEG_LOG_SQRT_TWO_PI) - log_half;
      else if (dat.prior_dist_for_aux == 2) {
        double nu = dat.prior_df_for_aux;
        Dual t = log(square(aux_unscaled) / nu + 1.0) * (-(nu + 1.0) / 2.0);
        lp = lp + (t + (std::lgamma((nu + 1.0) / 2.0) - std::lgamma(nu / 2.0) - 0.5 * std::log(nu * M_PI))) - log_half;
      } else lp = lp - aux_unscaled;
    }
    if (dat.prior_dist >= 1)
      for (auto& z : z_beta) lp = lp + (square(z) * -0.5 + NEG_LOG_SQRT_TWO_PI);
    {
      const double log_half = -0.693147180559945286;
      auto half_normal = [&](const std::vector<Dual>& v) { for (auto& x : v) lp = lp + (square(x) * -0.5 + NEG_LOG_SQRT_TWO_PI); lp = lp - log_half; };
      auto inv_gamma = [&](const Dual& x, double al, double be) { lp = lp + (log(x) * (-(al + 1.0)) - cst(be, Dn) / x + (al * std::log(be) - std::lgamma(al))); };
      if (hs > 0) {
        half_normal(local[0]);
        for (int k = 0; k < dat.K; ++k) inv_gamma(local[1][(size_t)k], 0.5 * dat.prior_df[(size_t)k], 0.5 * dat.prior_df[(size_t)k]);
        if (hs == 4) {
          half_normal(local[2]);
          for (int k = 0; k < dat.K; ++k) inv_gamma(local[3][(size_t)k], 0.5 * dat.prior_scale[(size_t)k], 0.5 * dat.prior_scale[(size_t)k]);
        }
        lp = lp + (square(global[0]) * -0.5 + NEG_LOG_SQRT_TWO_PI) - log_half;
        inv_gamma(global[1], 0.5 * dat.global_prior_df, 0.5 * dat.global_prior_df);
        inv_gamma(caux[0], 0.5 * dat.slab_df, 0.5 * dat.slab_df);
      }
      for (int k = 0; k < n_mix; ++k) lp = lp - mix[0][(size_t)k];   // exponential_lpdf(mix | 1)
      if (n_lambda) {   // chi_square_lpdf(one_over_lambda | prior_df[1])
        double nu = dat.prior_df[0];
        lp = lp + (log(one_over_lambda[0]) * (0.5 * nu - 1.0) - one_over_lambda[0] * 0.5 - (0.5 * nu * std::log(2.0) + std::lgamma(0.5 * nu)));
      }
    }
    // decov_lp
    for (auto& z : z_b) lp = lp + (square(z) * -0.5 + NEG_LOG_SQRT_TWO_PI);
    for (auto& z : z_T) lp = lp + (square(z) * -0.5 + NEG_LOG_SQRT_TWO_PI);
    int pos_reg = 0, pos_rho = 0;
    for (int i = 0; i < dat.t; ++i) if (dat.p[i] > 1) {
      int m = dat.p[i] - 1;
      std::vector<double> s1(m), s2(m);
      double nu = dat.regularization[pos_reg++] + 0.5 * (dat.p[i] - 2);
      s1[0] = nu; s2[0] = nu;
      for (int j = 2; j <= m; ++j) { nu -= 0.5; s1[j - 1] = 0.5 * j; s2[j - 1] = nu; }
      for (int j = 0; j < m; ++j) {
        const Dual& r = rho[pos_rho + j];
        double lbeta = std::lgamma(s1[j]) + std::lgamma(s2[j]) - std::lgamma(s1[j] + s2[j]);
        lp = lp + (log(r) * (s1[j] - 1.0) + log1m(r) * (s2[j] - 1.0) - lbeta);
      }
      pos_rho += m;
    }
    for (int j = 0; j < len_conc; ++j) lp = lp + (log(zeta[j]) * (delta[j] - 1.0) - zeta[j] - std::lgamma(delta[j]));
    for (int j = 0; j < dat.t; ++j) lp = lp + (log(tau[j]) * (dat.shape[j] - 1.0) - tau[j] - std::lgamma(dat.shape[j]));
  }

  // log density (constants kept: every lpdf is <false>, continuous.hpp:2460-2631) and gradient
  double log_prob_grad(const std::vector<double>& qv, std::vector<double>& grad) const {
    ++gradEvals;
    const size_t Dn = (size_t)D;
    Dual lp = cst(0.0, Dn);
    Transformed T;
    transform(qv, T, lp, true);
    // O(N) part: eta = offset_ + X beta + Z b; normal_lpdf<false>(y | eta, sigma)
    const int64_t N = dat.N;
    std::vector<double> beta(dat.K), b(dat.q), gbeta(dat.K, 0.0), gb(dat.q, 0.0);
    for (int k = 0; k < dat.K; ++k) beta[k] = T.beta[k].v;
    for (int j = 0; j < dat.q; ++j) b[j] = T.b[j].v;
    const double sigma = T.sigma.v;
    double ss = 0.0;
    for (int64_t i = 0; i < N; ++i) {
      double eta = dat.offset_[i];
      for (int k = 0; k < dat.K; ++k) eta += dat.X[(size_t)k * N + i] * beta[k];
      if (dat.t > 0) for (int e = dat.u[i]; e < dat.u[i + 1]; ++e) eta += dat.w[e] * b[dat.v[e]];
      double r = dat.y[i] - eta;
      double wr = dat.has_weights ? dat.weights[i] * r : r;
      ss += wr * r;
      for (int k = 0; k < dat.K; ++k) gbeta[k] += dat.X[(size_t)k * N + i] * wr;
      if (dat.t > 0) for (int e = dat.u[i]; e < dat.u[i + 1]; ++e) gb[dat.v[e]] += dat.w[e] * wr;
    }
    const double s2 = sigma * sigma;
    double ll = -0.5 * ss / s2 - (double)N * std::log(sigma) + (double)N * -0.91893853320467274178;
    double gsigma = ss / (s2 * sigma) - (double)N / sigma;
    lp.v += ll;
    for (size_t j = 0; j < Dn; ++j) {
      double acc = gsigma * T.sigma.d[j];
      for (int k = 0; k < dat.K; ++k) acc += (gbeta[k] / s2) * T.beta[k].d[j];
      for (int m = 0; m < dat.q; ++m) acc += (gb[m] / s2) * T.b[m].d[j];
      lp.d[j] += acc;
    }
    grad = lp.d;
    return lp.v;
  }

  // write_array: constrained parameters then transformed parameters
  void write_array(const std::vector<double>& qv, double* out) const {
    Dual lp = cst(0.0, (size_t)D);
    Transformed T;
    transform(qv, T, lp, false);
    int o = 0;
    for (auto& x : T.constrained) out[o++] = x.v;
    if (!dat.is_binary) out[o++] = T.sigma.v;
    for (auto& x : T.beta) out[o++] = x.v;
    for (auto& x : T.b) out[o++] = x.v;
    for (auto& x : T.theta_L) out[o++] = x.v;
  }

  // get_parametric_mean (continuous.hpp:3662-3768): reads beta, b out of a sample row
  void parametric_mean(const double* constrainedRow, double* result, bool fixed, bool random) const {
    const double* beta = constrainedRow + beta_pos();
    const double* b = constrainedRow + b_pos();
    const int64_t N = dat.N;
    for (int64_t i = 0; i < N; ++i) {
      double eta = 0.0;
      if (fixed) for (int k = 0; k < dat.K; ++k) eta += dat.X[(size_t)k * N + i] * beta[k];
      if (random && dat.t > 0) for (int e = dat.u[i]; e < dat.u[i + 1]; ++e) eta += dat.w[e] * b[dat.v[e]];
      result[i] = eta;
    }
  }

 private:
  static Dual CFt(const Dual& z, double df) {
    Dual z2 = square(z), z3 = z2 * z, z5 = z2 * z3, z7 = z2 * z5, z9 = z2 * z7;
    double df2 = df * df, df3 = df2 * df, df4 = df2 * df2;
    return z + (z3 + z) / (4 * df) + (z5 * 5.0 + z3 * 16.0 + z * 3.0) / (96 * df2) +
           (z7 * 3.0 + z5 * 19.0 + z3 * 17.0 - z * 15.0) / (384 * df3) +
           (z9 * 79.0 + z7 * 776.0 + z5 * 1482.0 - z3 * 1920.0 - z * 945.0) / (92160 * df4);
  }
  void make_theta_L(const Dual& dispersion, const std::vector<Dual>& tau, const std::vector<Dual>& zeta,
                    const std::vector<Dual>& rho, const std::vector<Dual>& z_T, std::vector<Dual>& theta_L) const {
    const size_t Dn = (size_t)D;
    theta_L.clear();
    int zeta_mark = 0, rho_mark = 0, z_T_mark = 0;
    for (int i = 0; i < dat.t; ++i) {
      int nc = dat.p[i];
      if (nc == 1) { theta_L.push_back(tau[i] * dat.scale[i] * dispersion); continue; }
      std::vector<std::vector<Dual>> Ti(nc, std::vector<Dual>(nc, cst(0.0, Dn)));
      Dual trace = square(tau[i] * dat.scale[i] * dispersion) * (double)nc;
      std::vector<Dual> pi(nc);
      Dual sum_pi = cst(0.0, Dn);
      for (int j = 0; j < nc; ++j) { pi[j] = zeta[zeta_mark + j]; sum_pi = sum_pi + pi[j]; }
      for (int j = 0; j < nc; ++j) pi[j] = pi[j] / sum_pi;
      zeta_mark += nc;
      Dual std_dev = sqrt(pi[0] * trace);
      Ti[0][0] = std_dev;
      std_dev = sqrt(pi[1] * trace);
      Dual T21 = rho[rho_mark] * 2.0 - 1.0;
      rho_mark += 1;
      Ti[1][1] = std_dev * sqrt(1.0 - square(T21));
      Ti[1][0] = std_dev * T21;
      for (int r = 2; r <= nc - 1; ++r) {
        int rp1 = r + 1;
        Dual dot = cst(0.0, Dn);
        for (int c = 0; c < r; ++c) dot = dot + square(z_T[z_T_mark + c]);
        Dual scale_factor = sqrt(rho[rho_mark] / dot) * std_dev;
        for (int c = 0; c < r; ++c) Ti[rp1 - 1][c] = z_T[z_T_mark + c] * scale_factor;
        z_T_mark += r;
        std_dev = sqrt(pi[rp1 - 1] * trace);
        Ti[rp1 - 1][rp1 - 1] = sqrt(1.0 - rho[rho_mark]) * std_dev;
        rho_mark += 1;
      }
      for (int c = 0; c < nc; ++c) for (int r = c; r < nc; ++r) theta_L.push_back(Ti[r][c]);
    }
  }
  void make_b(const std::vector<Dual>& z_b, const std::vector<Dual>& theta_L, std::vector<Dual>& b) const {
    const size_t Dn = (size_t)D;
    b.assign((size_t)dat.q, cst(0.0, Dn));
    int b_mark = 0, th = 0;
    for (int i = 0; i < dat.t; ++i) {
      int nc = dat.p[i];
      if (nc == 1) {
        for (int s = b_mark; s < b_mark + dat.l[i]; ++s) b[s] = theta_L[th] * z_b[s];
        b_mark += dat.l[i]; th += 1;
      } else {
        std::vector<std::vector<Dual>> Ti(nc, std::vector<Dual>(nc, cst(0.0, Dn)));
        for (int c = 0; c < nc; ++c) { Ti[c][c] = theta_L[th++]; for (int r = c + 1; r < nc; ++r) Ti[r][c] = theta_L[th++]; }
        for (int j = 0; j < dat.l[i]; ++j) {
          for (int r = 0; r < nc; ++r) {
            Dual acc = cst(0.0, Dn);
            for (int c = 0; c <= r; ++c) acc = acc + Ti[r][c] * z_b[b_mark + c];
            b[b_mark + r] = acc;
          }
          b_mark += nc;
        }
      }
    }
  }
};

// ------------------------------------------------------------------ NUTS + adaptation
struct StanControl {
  uint32_t seed = 0; double init_radius = 2.0; int skip = 1;
  double adapt_gamma = 0.05, adapt_delta = 0.8, adapt_kappa = 0.75, adapt_t0 = 10.0;
  unsigned init_buffer = 75, term_buffer = 50, window = 25;
  double stepsize = 1.0, stepsize_jitter = 0.0; int max_treedepth = 10;
};

struct PsPoint { std::vector<double> q, p, g; double V = 0.0; };

class NutsSampler {
 public:
  StanModel& model;
  Ecuyer1988 rng;
  int D;
  PsPoint z;
  std::vector<double> inv_metric;
  double nom_epsilon = 0.1, epsilon = 0.1, epsilon_jitter = 0.0;
  int max_depth = 10; double max_deltaH = 1000.0;
  int depth_ = 0, n_leapfrog_ = 0; bool divergent_ = false; double energy_ = 0.0;
  bool adapt_flag = true;
  // stepsize adaptation
  double sa_mu = 0.5, sa_delta = 0.5, sa_gamma = 0.05, sa_kappa = 0.75, sa_t0 = 10, sa_counter = 0, sa_s_bar = 0, sa_x_bar = 0;
  // windowed variance adaptation
  unsigned num_warmup_ = 0, adapt_init_buffer_ = 0, adapt_term_buffer_ = 0, adapt_base_window_ = 0;
  unsigned adapt_window_counter_ = 0, adapt_next_window_ = 0, adapt_window_size_ = 0;
  double wf_n = 0; std::vector<double> wf_m, wf_m2;
  // current sample
  std::vector<double> cont_params; double lp_ = 0, accept_stat_ = 0;
  int num_skip;
  long tot_transitions = 0, tot_depth = 0, tot_leapfrog = 0, tot_divergent = 0;

  NutsSampler(StanModel& m, const StanControl& c, unsigned chain, int num_warmup)
      : model(m), D(m.D), num_skip(c.skip) {
    rng.create(c.seed, chain);
    cont_params = initialize(c.init_radius);
    inv_metric.assign((size_t)D, 1.0);
    z.q.assign((size_t)D, 0.0); z.p.assign((size_t)D, 0.0); z.g.assign((size_t)D, 0.0);
    wf_m.assign((size_t)D, 0.0); wf_m2.assign((size_t)D, 0.0);
    if (c.stepsize > 0) nom_epsilon = c.stepsize;
    if (c.stepsize_jitter > 0 && c.stepsize_jitter < 1) epsilon_jitter = c.stepsize_jitter;
    if (c.max_treedepth > 0) max_depth = c.max_treedepth;
    sa_mu = std::log(10 * c.stepsize);
    if (c.adapt_delta > 0 && c.adapt_delta < 1) sa_delta = c.adapt_delta;
    if (c.adapt_gamma > 0) sa_gamma = c.adapt_gamma;
    if (c.adapt_kappa > 0) sa_kappa = c.adapt_kappa;
    if (c.adapt_t0 > 0) sa_t0 = c.adapt_t0;
    window_restart();
    set_window_params((unsigned)(num_warmup * c.skip), c.init_buffer, c.term_buffer, c.window);
    adapt_flag = true;
    z.q = cont_params;
    init_stepsize();
  }

  // interruptable_sampler::run
  void run(double* row) {
    for (int s = 0; s < num_skip - 1; ++s) transition();
    transition();
    row[0] = lp_; row[1] = accept_stat_; row[2] = epsilon; row[3] = depth_; row[4] = n_leapfrog_;
    row[5] = divergent_ ? 1.0 : 0.0; row[6] = energy_;
    model.write_array(cont_params, row + 7);
  }
  void disengage_adaptation() { adapt_flag = false; nom_epsilon = std::exp(sa_x_bar); }

 private:
  std::vector<double> initialize(double radius) {
    const int MAX_TRIES = (radius == 0.0) ? 1 : 100;
    for (int tries = 0; tries < MAX_TRIES; ++tries) {
      std::vector<double> u((size_t)D);
      for (int i = 0; i < D; ++i) u[(size_t)i] = radius == 0.0 ? 0.0 : rng.uniform_real(-radius, radius);
      std::vector<double> g;
      double lp = model.log_prob_grad(u, g);          // log_prob<false,true> double check
      if (!std::isfinite(lp)) continue;
      lp = model.log_prob_grad(u, g);                 // log_prob_grad<true,true>
      double s = 0; for (double x : g) s += x;
      if (!std::isfinite(s)) continue;
      return u;
    }
    throw std::domain_error("Initialization failed.");
  }

  double T() const { double s = 0; for (int i = 0; i < D; ++i) s += z.p[i] * (inv_metric[i] * z.p[i]); return 0.5 * s; }
  double H() const { return T() + z.V; }
  void update_potential_gradient() {
    z.V = -model.log_prob_grad(z.q, z.g);
    for (double& x : z.g) x = -x;
  }
  void sample_p() { for (int i = 0; i < D; ++i) z.p[i] = boost_normal(rng) / std::sqrt(inv_metric[i]); }
  void evolve(double eps) {
    for (int i = 0; i < D; ++i) z.p[i] -= (0.5 * eps) * z.g[i];
    for (int i = 0; i < D; ++i) z.q[i] += eps * (inv_metric[i] * z.p[i]);
    update_potential_gradient();
    for (int i = 0; i < D; ++i) z.p[i] -= (0.5 * eps) * z.g[i];
  }
  std::vector<double> dtau_dp() const { std::vector<double> r((size_t)D); for (int i = 0; i < D; ++i) r[i] = inv_metric[i] * z.p[i]; return r; }

  void init_stepsize() {
    PsPoint z_init = z;
    if (nom_epsilon == 0 || nom_epsilon > 1e7 || std::isnan(nom_epsilon)) return;
    sample_p(); update_potential_gradient();
    double H0 = H();
    evolve(nom_epsilon);
    double h = H(); if (std::isnan(h)) h = std::numeric_limits<double>::infinity();
    double delta_H = H0 - h;
    int direction = delta_H > std::log(0.8) ? 1 : -1;
    while (1) {
      z = z_init;
      sample_p(); update_potential_gradient();
      double H0b = H();
      evolve(nom_epsilon);
      double hb = H(); if (std::isnan(hb)) hb = std::numeric_limits<double>::infinity();
      double dH = H0b - hb;
      if ((direction == 1) && !(dH > std::log(0.8))) break;
      else if ((direction == -1) && !(dH < std::log(0.8))) break;
      else nom_epsilon = direction == 1 ? 2.0 * nom_epsilon : 0.5 * nom_epsilon;
      if (nom_epsilon > 1e7) throw std::runtime_error("Posterior is improper. Please check your model.");
      if (nom_epsilon == 0) throw std::runtime_error("No acceptably small step size could be found.");
    }
    z = z_init;
  }

  static double log_sum_exp(double a, double b) {
    if (a == -std::numeric_limits<double>::infinity()) return b;
    if (a == std::numeric_limits<double>::infinity() && b == std::numeric_limits<double>::infinity()) return a;
    if (a > b) return a + std::log1p(std::exp(b - a));
    return b + std::log1p(std::exp(a - b));
  }
  static bool criterion(const std::vector<double>& p_sharp_minus, const std::vector<double>& p_sharp_plus, const std::vector<double>& rho) {
    double a = 0, b = 0;
    for (size_t i = 0; i < rho.size(); ++i) { a += p_sharp_plus[i] * rho[i]; b += p_sharp_minus[i] * rho[i]; }
    return a > 0 && b > 0;
  }
  typedef std::vector<double> Vec;
  static Vec add(const Vec& a, const Vec& b) { Vec r(a.size()); for (size_t i = 0; i < a.size(); ++i) r[i] = a[i] + b[i]; return r; }

  bool build_tree(int depth, PsPoint& z_propose, Vec& p_sharp_beg, Vec& p_sharp_end, Vec& rho, Vec& p_beg, Vec& p_end,
                  double H0, double sign, int& n_leapfrog, double& log_sum_weight, double& sum_metro_prob) {
    if (depth == 0) {
      evolve(sign * epsilon);
      ++n_leapfrog;
      double h = H(); if (std::isnan(h)) h = std::numeric_limits<double>::infinity();
      if ((h - H0) > max_deltaH) divergent_ = true;
      log_sum_weight = log_sum_exp(log_sum_weight, H0 - h);
      if (H0 - h > 0) sum_metro_prob += 1; else sum_metro_prob += std::exp(H0 - h);
      z_propose = z;
      p_sharp_beg = dtau_dp(); p_sharp_end = p_sharp_beg;
      for (int i = 0; i < D; ++i) rho[i] += z.p[i];
      p_beg = z.p; p_end = p_beg;
      return !divergent_;
    }
    double log_sum_weight_init = -std::numeric_limits<double>::infinity();
    Vec p_init_end((size_t)D), p_sharp_init_end((size_t)D), rho_init((size_t)D, 0.0);
    bool valid_init = build_tree(depth - 1, z_propose, p_sharp_beg, p_sharp_init_end, rho_init, p_beg, p_init_end, H0, sign,
                                 n_leapfrog, log_sum_weight_init, sum_metro_prob);
    if (!valid_init) return false;
    PsPoint z_propose_final = z;
    double log_sum_weight_final = -std::numeric_limits<double>::infinity();
    Vec p_final_beg((size_t)D), p_sharp_final_beg((size_t)D), rho_final((size_t)D, 0.0);
    bool valid_final = build_tree(depth - 1, z_propose_final, p_sharp_final_beg, p_sharp_end, rho_final, p_final_beg, p_end, H0,
                                  sign, n_leapfrog, log_sum_weight_final, sum_metro_prob);
    if (!valid_final) return false;
    double log_sum_weight_subtree = log_sum_exp(log_sum_weight_init, log_sum_weight_final);
    log_sum_weight = log_sum_exp(log_sum_weight, log_sum_weight_subtree);
    if (log_sum_weight_final > log_sum_weight_subtree) z_propose = z_propose_final;
    else {
      double accept_prob = std::exp(log_sum_weight_final - log_sum_weight_subtree);
      if (rng.uniform01() < accept_prob) z_propose = z_propose_final;
    }
    Vec rho_subtree = add(rho_init, rho_final);
    for (int i = 0; i < D; ++i) rho[i] += rho_subtree[i];
    bool persist = criterion(p_sharp_beg, p_sharp_end, rho_subtree);
    rho_subtree = add(rho_init, p_final_beg);
    persist &= criterion(p_sharp_beg, p_sharp_final_beg, rho_subtree);
    rho_subtree = add(rho_final, p_init_end);
    persist &= criterion(p_sharp_init_end, p_sharp_end, rho_subtree);
    return persist;
  }

  void base_transition() {
    epsilon = nom_epsilon;
    if (epsilon_jitter) epsilon *= 1.0 + epsilon_jitter * (2.0 * rng.uniform01() - 1.0);
    z.q = cont_params;
    sample_p(); update_potential_gradient();
    PsPoint z_fwd = z, z_bck = z, z_sample = z, z_propose = z;
    Vec p_fwd_fwd = z.p, p_sharp_fwd_fwd = dtau_dp();
    Vec p_fwd_bck = z.p, p_sharp_fwd_bck = p_sharp_fwd_fwd;
    Vec p_bck_fwd = z.p, p_sharp_bck_fwd = p_sharp_fwd_fwd;
    Vec p_bck_bck = z.p, p_sharp_bck_bck = p_sharp_fwd_fwd;
    Vec rho = z.p;
    double log_sum_weight = 0, H0 = H();
    int n_leapfrog = 0; double sum_metro_prob = 0;
    depth_ = 0; divergent_ = false;
    while (depth_ < max_depth) {
      Vec rho_fwd((size_t)D, 0.0), rho_bck((size_t)D, 0.0);
      bool valid_subtree = false;
      double log_sum_weight_subtree = -std::numeric_limits<double>::infinity();
      if (rng.uniform01() > 0.5) {
        z = z_fwd; rho_bck = rho; p_bck_fwd = p_fwd_fwd; p_sharp_bck_fwd = p_sharp_fwd_fwd;
        valid_subtree = build_tree(depth_, z_propose, p_sharp_fwd_bck, p_sharp_fwd_fwd, rho_fwd, p_fwd_bck, p_fwd_fwd, H0, 1,
                                   n_leapfrog, log_sum_weight_subtree, sum_metro_prob);
        z_fwd = z;
      } else {
        z = z_bck; rho_fwd = rho; p_fwd_bck = p_bck_bck; p_sharp_fwd_bck = p_sharp_bck_bck;
        valid_subtree = build_tree(depth_, z_propose, p_sharp_bck_fwd, p_sharp_bck_bck, rho_bck, p_bck_fwd, p_bck_bck, H0, -1,
                                   n_leapfrog, log_sum_weight_subtree, sum_metro_prob);
        z_bck = z;
      }
      if (!valid_subtree) break;
      ++depth_;
      if (log_sum_weight_subtree > log_sum_weight) z_sample = z_propose;
      else {
        double accept_prob = std::exp(log_sum_weight_subtree - log_sum_weight);
        if (rng.uniform01() < accept_prob) z_sample = z_propose;
      }
      log_sum_weight = log_sum_exp(log_sum_weight, log_sum_weight_subtree);
      rho = add(rho_bck, rho_fwd);
      bool persist = criterion(p_sharp_bck_bck, p_sharp_fwd_fwd, rho);
      Vec rho_extended = add(rho_bck, p_fwd_bck);
      persist &= criterion(p_sharp_bck_bck, p_sharp_fwd_bck, rho_extended);
      rho_extended = add(rho_fwd, p_bck_fwd);
      persist &= criterion(p_sharp_bck_fwd, p_sharp_fwd_fwd, rho_extended);
      if (!persist) break;
    }
    n_leapfrog_ = n_leapfrog;
    ++tot_transitions; tot_depth += depth_; tot_leapfrog += n_leapfrog; if (divergent_) ++tot_divergent;
    double accept_prob = sum_metro_prob / (double)n_leapfrog;
    z = z_sample;
    energy_ = H();
    cont_params = z.q; lp_ = -z.V; accept_stat_ = accept_prob;
  }

  // adapt_diag_e_nuts::transition
  void transition() {
    base_transition();
    if (adapt_flag) {
      learn_stepsize(accept_stat_);
      bool update = learn_variance();
      if (update) {
        init_stepsize();
        sa_mu = std::log(10 * nom_epsilon);
        sa_counter = 0; sa_s_bar = 0; sa_x_bar = 0;
      }
    }
  }
  void learn_stepsize(double adapt_stat) {
    ++sa_counter;
    adapt_stat = adapt_stat > 1 ? 1 : adapt_stat;
    const double eta = 1.0 / (sa_counter + sa_t0);
    sa_s_bar = (1.0 - eta) * sa_s_bar + eta * (sa_delta - adapt_stat);
    const double x = sa_mu - sa_s_bar * std::sqrt(sa_counter) / sa_gamma;
    const double x_eta = std::pow(sa_counter, -sa_kappa);
    sa_x_bar = (1.0 - x_eta) * sa_x_bar + x_eta * x;
    nom_epsilon = std::exp(x);
  }
  void window_restart() {
    adapt_window_counter_ = 0;
    adapt_window_size_ = adapt_base_window_;
    adapt_next_window_ = adapt_init_buffer_ + adapt_window_size_ - 1;
  }
  void set_window_params(unsigned num_warmup, unsigned init_buffer, unsigned term_buffer, unsigned base_window) {
    if (num_warmup < 20) return;
    if (init_buffer + base_window + term_buffer > num_warmup) {
      num_warmup_ = num_warmup;
      adapt_init_buffer_ = (unsigned)(0.15 * num_warmup);
      adapt_term_buffer_ = (unsigned)(0.10 * num_warmup);
      adapt_base_window_ = num_warmup - (adapt_init_buffer_ + adapt_term_buffer_);
      return;   // (no restart() here in the vendored Stan 2.28: windowed_adaptation.hpp:45-74)
    }
    num_warmup_ = num_warmup; adapt_init_buffer_ = init_buffer; adapt_term_buffer_ = term_buffer; adapt_base_window_ = base_window;
    window_restart();
  }
  bool adaptation_window() const {
    return (adapt_window_counter_ >= adapt_init_buffer_) && (adapt_window_counter_ < num_warmup_ - adapt_term_buffer_) &&
           (adapt_window_counter_ != num_warmup_);
  }
  bool end_adaptation_window() const { return (adapt_window_counter_ == adapt_next_window_) && (adapt_window_counter_ != num_warmup_); }
  void compute_next_window() {
    if (adapt_next_window_ == num_warmup_ - adapt_term_buffer_ - 1) return;
    adapt_window_size_ *= 2;
    adapt_next_window_ = adapt_window_counter_ + adapt_window_size_;
    if (adapt_next_window_ == num_warmup_ - adapt_term_buffer_ - 1) return;
    unsigned next_window_boundary = adapt_next_window_ + 2 * adapt_window_size_;
    if (next_window_boundary >= num_warmup_ - adapt_term_buffer_) adapt_next_window_ = num_warmup_ - adapt_term_buffer_ - 1;
  }
  bool learn_variance() {
    if (adaptation_window()) {
      ++wf_n;
      for (int i = 0; i < D; ++i) {
        double delta = z.q[i] - wf_m[i];
        wf_m[i] += delta / wf_n;
        wf_m2[i] += delta * (z.q[i] - wf_m[i]);
      }
    }
    if (end_adaptation_window()) {
      compute_next_window();
      double nn = wf_n;
      for (int i = 0; i < D; ++i) {
        double var = inv_metric[i];
        if (wf_n > 1) var = wf_m2[i] / (wf_n - 1.0);
        inv_metric[i] = (nn / (nn + 5.0)) * var + 1e-3 * (5.0 / (nn + 5.0));
      }
      wf_n = 0; std::fill(wf_m.begin(), wf_m.end(), 0.0); std::fill(wf_m2.begin(), wf_m2.end(), 0.0);
      ++adapt_window_counter_;
      return true;
    }
    ++adapt_window_counter_;
    return false;
  }
};

}  // namespace oracle
#endif
