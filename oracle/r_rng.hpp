// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into, imported by, or called from the
// product path (stan4bart_amd/). Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use anything under oracle/.
//
// CPU restatement of R's default random number generators, which is the stream the
// reference's BART block draws from in single-chain mode (reference
// src/init.cpp:259,298,750,919 GetRNGstate/PutRNGstate; R/stan4bart_fit.R:35-38
// set.seed + sample.int).  R itself is a third-party dependency that
// is absent from /root/reference; the algorithms restated here are R's published
// src/main/RNG.c (Mersenne-Twister, Randomize/FixupSeeds, R_unif_index "Rejection"),
// src/nmath/snorm.c (INVERSION), src/nmath/qnorm.c (Wichura AS241) and src/nmath/sexp.c.
// Pinned by the known-answer values of SURVEY.md Appendix B.1 (tests/test_oracle_rng.py).
#ifndef ORACLE_R_RNG_HPP
#define ORACLE_R_RNG_HPP

#include <cstdint>
#include <cmath>

namespace oracle {

struct RRng {
  uint32_t mt[624];
  int mti;  // R's dummy[0]; 624 => regenerate on next draw

  // R: set.seed(seed) with kind = Mersenne-Twister -> RNG_Init + FixupSeeds (RNG.c)
  void set_seed(uint32_t seed) {
    for (int j = 0; j < 50; ++j) seed = 69069u * seed + 1u;
    // i_seed[0] is dummy[0]=mti, i_seed[1..624] is mt[]
    seed = 69069u * seed + 1u;  // consumed by dummy[0], then overwritten by FixupSeeds
    for (int j = 0; j < 624; ++j) { seed = 69069u * seed + 1u; mt[j] = seed; }
    mti = 624;  // FixupSeeds: dummy[0] = 624
  }

  uint32_t next_u32() {
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    if (mti >= 624) {
      int kk;
      uint32_t y;
      for (kk = 0; kk < 624 - 397; ++kk) {
        y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
      }
      for (; kk < 623; ++kk) {
        y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
      }
      y = (mt[623] & UPPER) | (mt[0] & LOWER);
      mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
      mti = 0;
    }
    uint32_t y = mt[mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }

  // unif_rand(): MT_genrand() then fixup() into the open interval (0,1)
  double unif_rand() {
    const double i2_32m1 = 2.328306437080797e-10;
    double v = (double)next_u32() * 2.3283064365386963e-10;
    if (v <= 0.0) return 0.5 * i2_32m1;
    if ((1.0 - v) <= 0.0) return 1.0 - 0.5 * i2_32m1;
    return v;
  }

  // norm_rand(), N01_kind = INVERSION (snorm.c)
  double norm_rand() {
    const double BIG = 134217728.0;  // 2^27
    double u = unif_rand();
    u = (double)(int)(BIG * u) + unif_rand();
    return qnorm(u / BIG);
  }

  // exp_rand() (sexp.c, Ahrens & Dieter 1972)
  double exp_rand() {
    static const double q[] = {
        0.6931471805599453, 0.9333736875190459, 0.9888777961838675, 0.9984589039328340,
        0.9998292811061389, 0.9999833164100727, 0.9999985691438767, 0.9999998906925558,
        0.9999999924734159, 0.9999999995283275, 0.9999999999728814, 0.9999999999985598,
        0.9999999999999289, 0.9999999999999968, 0.9999999999999999, 1.0000000000000000};
    double a = 0.0;
    double u = unif_rand();
    while (u <= 0.0 || u >= 1.0) u = unif_rand();
    for (;;) {
      u += u;
      if (u > 1.0) break;
      a += q[0];
    }
    u -= 1.0;
    if (u <= q[0]) return a + u;
    int i = 0;
    double ustar = unif_rand(), umin = ustar;
    do {
      ustar = unif_rand();
      if (umin > ustar) umin = ustar;
      ++i;
    } while (u > q[i]);
    return a + umin * q[0];
  }

  // rgamma(a, scale) (nmath/rgamma.c: Ahrens & Dieter GD (1982) for a >= 1, GS (1974) for a < 1), the generator behind dbarts'
  // ext_rng_simulateGamma when the chain draws from R's native stream.  The static caches of the C source only save
  // recomputation; every call here recomputes what depends on `a`.
  double rgamma(double a, double scale) {
    const double sqrt32 = 5.656854, exp_m1 = 0.36787944117144233;
    const double q1 = 0.04166669, q2 = 0.02083148, q3 = 0.00801191, q4 = 0.00144121, q5 = -7.388e-5, q6 = 2.4511e-4, q7 = 2.424e-4;
    const double a1 = 0.3333333, a2 = -0.250003, a3 = 0.2000062, a4 = -0.1662921, a5 = 0.1423657, a6 = -0.1367177, a7 = 0.1233795;
    if (std::isnan(a) || std::isnan(scale)) return NAN;
    if (a <= 0.0 || scale <= 0.0) { if (scale == 0.0 || a == 0.0) return 0.0; return NAN; }
    if (!std::isfinite(a) || !std::isfinite(scale)) return INFINITY;
    double e, p, q, r, t, u, v, w, x, ret_val;
    if (a < 1) {  // GS
      e = 1.0 + exp_m1 * a;
      for (;;) {
        p = e * unif_rand();
        if (p >= 1.0) {
          x = -std::log((e - p) / a);
          if (exp_rand() >= (1.0 - a) * std::log(x)) break;
        } else {
          x = std::exp(std::log(p) / a);
          if (exp_rand() >= x) break;
        }
      }
      return scale * x;
    }
    // Step 1
    const double s2 = a - 0.5, s = std::sqrt(s2), d = sqrt32 - s * 12;
    // Step 2: immediate acceptance
    t = norm_rand();
    x = s + 0.5 * t;
    ret_val = x * x;
    if (t >= 0) return scale * ret_val;
    // Step 3: squeeze acceptance
    u = unif_rand();
    if (d * u <= t * t * t) return scale * ret_val;
    // Step 4
    r = 1 / a;
    const double q0 = ((((((q7 * r + q6) * r + q5) * r + q4) * r + q3) * r + q2) * r + q1) * r;
    double b, si, c;
    if (a <= 3.686) { b = 0.463 + s + 0.178 * s2; si = 1.235; c = 0.195 / s - 0.079 + 0.16 * s; }
    else if (a <= 13.022) { b = 1.654 + 0.0076 * s2; si = 1.68 / s + 0.275; c = 0.062 / s + 0.024; }
    else { b = 1.77; si = 0.75; c = 0.1515 / s; }
    // Step 5-7: quotient test
    if (x > 0.0) {
      v = t / (s + s);
      if (std::fabs(v) <= 0.25) q = q0 + 0.5 * t * t * ((((((a7 * v + a6) * v + a5) * v + a4) * v + a3) * v + a2) * v + a1) * v;
      else q = q0 - s * t + 0.25 * t * t + (s2 + s2) * std::log(1.0 + v);
      if (std::log(1.0 - u) <= q) return scale * ret_val;
    }
    for (;;) {
      // Step 8: double exponential sample
      e = exp_rand();
      u = unif_rand();
      u = u + u - 1.0;
      if (u < 0.0) t = b - si * e; else t = b + si * e;
      // Step 9
      if (t >= -0.71874483771719) {
        // Step 10
        v = t / (s + s);
        if (std::fabs(v) <= 0.25) q = q0 + 0.5 * t * t * ((((((a7 * v + a6) * v + a5) * v + a4) * v + a3) * v + a2) * v + a1) * v;
        else q = q0 - s * t + 0.25 * t * t + (s2 + s2) * std::log(1.0 + v);
        // Step 11: hat acceptance
        if (q > 0.0) {
          w = std::expm1(q);
          if (c * std::fabs(u) <= w * std::exp(e - 0.5 * t * t)) break;
        }
      }
    }
    x = s + 0.5 * t;
    return scale * x * x;
  }

  // R_unif_index(dn), sample.kind = "Rejection" (R >= 3.6)
  double unif_index(double dn) {
    if (dn <= 0) return 0.0;
    int bits = (int)std::ceil(std::log2(dn));
    double dv;
    do {
      int64_t v = 0;
      for (int n = 0; n <= bits; n += 16) {
        int v1 = (int)std::floor(unif_rand() * 65536);
        v = 65536 * v + v1;
      }
      if (bits < 64) v &= ((int64_t(1) << bits) - 1);
      dv = (double)v;
    } while (dn <= dv);
    return dv;
  }

  // qnorm5(p, 0, 1, lower_tail = TRUE, log_p = FALSE): Wichura (1988) AS241 PPND16
  static double qnorm(double p) {
    if (std::isnan(p)) return p;
    if (p <= 0.0) return -INFINITY;
    if (p >= 1.0) return INFINITY;
    double q = p - 0.5, r, val;
    if (std::fabs(q) <= 0.425) {
      r = .180625 - q * q;
      val = q * (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r +
                     45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r +
                  133.14166789178437745) * r + 3.387132872796366608) /
            (((((((r * 5226.495278852545925 + 28729.085735721942674) * r + 39307.89580009271061) * r +
                 21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r +
              42.313330701600911252) * r + 1.);
    } else {
      r = (q < 0) ? p : 1.0 - p;
      r = std::sqrt(-std::log(r));
      if (r <= 5.) {
        r += -1.6;
        val = (((((((r * 7.7454501427834140764e-4 + .0227238449892691845833) * r + .24178072517745061177) * r +
                   1.27045825245236838258) * r + 3.64784832476320460504) * r + 5.7694972214606914055) * r +
                4.6303378461565452959) * r + 1.42343711074968357734) /
              (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + .0151986665636164571966) * r +
                   .14810397642748007459) * r + .68976733498510000455) * r + 1.6763848301838038494) * r +
                2.05319162663775882187) * r + 1.);
      } else {
        r += -5.;
        val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r + .0012426609473880784386) * r +
                   .026532189526576123093) * r + .29656057182850489123) * r + 1.7848265399172913358) * r +
                5.4637849111641143699) * r + 6.6579046435011037772) /
              (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r +
                   7.868691311456132591e-4) * r + .0148753612908506148525) * r + .13692988092273580531) * r +
                .59983220655588793769) * r + 1.);
      }
      if (q < 0.0) val = -val;
    }
    return val;
  }
};

}  // namespace oracle
#endif
