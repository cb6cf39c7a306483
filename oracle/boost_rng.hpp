// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the Boost.Random pieces the reference's Stan block draws from:
//   boost::ecuyer1988 created by stan::services::util::create_rng(seed, chain)
//     (reference src/include/stan/services/util/create_rng.hpp:25-31, src/init.cpp:211-212),
//   boost::uniform_01 (src/include/stan/mcmc/hmc/base_hmc.hpp:192, base_nuts.hpp:131,167,331),
//   boost::random::uniform_real_distribution (src/include/stan/io/random_var_context.hpp:70-73),
//   boost::normal_distribution (src/include/stan/mcmc/hmc/hamiltonians/diag_e_metric.hpp:44-50).
// Boost (BH >= 1.72, reference DESCRIPTION:63) is NOT vendored in /root/reference and is
// absent from this image.  ecuyer1988, uniform_01 and uniform_real are fully specified by
// their published definitions (L'Ecuyer 1988 two-MLCG combination; pinned by Boost's own
// validation constant: 10001st output of a default-constructed engine == 2060321752).
// normal_distribution is restated as the 128-layer ziggurat with
// generate_int_float_pair<double, 8> digit slicing; its layer tables are recomputed from the
// Marsaglia-Tsang recurrence rather than copied, so the last bits may differ from Boost's
// literals.  PARITY UNPINNED for the normal stream (no golden values available here).
#ifndef ORACLE_BOOST_RNG_HPP
#define ORACLE_BOOST_RNG_HPP

#include <cstdint>
#include <cmath>

namespace oracle {

struct Ecuyer1988 {
  static constexpr uint64_t M1 = 2147483563ull, A1 = 40014ull;
  static constexpr uint64_t M2 = 2147483399ull, A2 = 40692ull;
  uint32_t x1, x2;

  static uint32_t seed_lcg(uint32_t s, uint64_t m) {
    uint32_t x = (uint32_t)(s % m);
    if (x == 0) x = 1;  // increment == 0 && x == 0 -> 1
    return x;
  }
  void seed(uint32_t s) { x1 = seed_lcg(s, M1); x2 = seed_lcg(s, M2); }

  static uint64_t powmod(uint64_t a, uint64_t e, uint64_t m) {
    uint64_t r = 1; a %= m;
    while (e) { if (e & 1) r = (r * a) % m; a = (a * a) % m; e >>= 1; }
    return r;
  }
  void discard(uint64_t z) {
    x1 = (uint32_t)((powmod(A1, z, M1) * x1) % M1);
    x2 = (uint32_t)((powmod(A2, z, M2) * x2) % M2);
  }
  // stan::services::util::create_rng
  void create(uint32_t s, uint32_t chain) { seed(s); discard((uint64_t(1) << 50) * chain); }

  static constexpr uint32_t min() { return 1; }
  static constexpr uint32_t max() { return 2147483562u; }

  uint32_t next() {
    x1 = (uint32_t)((A1 * x1) % M1);
    x2 = (uint32_t)((A2 * x2) % M2);
    if (x2 < x1) return x1 - x2;
    return x1 - x2 + (uint32_t)(M1 - 1);
  }

  // boost::uniform_01<Engine&>: (x - min) / (max - min + 1)
  double uniform01() { return (double)(next() - 1u) / 2147483562.0; }

  // boost::random::uniform_real_distribution<double>(a, b)
  double uniform_real(double a, double b) {
    for (;;) {
      double r = (double)(next() - 1u) / 2147483562.0 * (b - a) + a;
      if (r < b) return r;
    }
  }

  // generate_one_digit(eng, 30): rejection to a uniform 30-bit digit
  uint32_t digit30() {
    uint32_t u;
    do { u = next() - 1u; } while (u > (1u << 30) - 1u);
    return u;
  }
  // generate_int_float_pair<double, 8>: 8-bit bucket + 53-bit uniform in [0,1)
  void int_float_pair(double& r, int& bucket) {
    uint32_t u1 = digit30();
    bucket = (int)(u1 & 0xFFu);
    r = (double)(u1 >> 8) * (1.0 / 4194304.0);          // 2^-22
    uint32_t u2 = digit30();
    r += (double)u2;
    r *= (1.0 / 1073741824.0);                           // 2^-30
    uint32_t u3 = digit30();
    r += (double)(u3 & 1u);
    r *= 0.5;
  }
};

struct ZigguratTables {
  double nx[129], ny[129];   // normal, 128 layers
  double ex[257], ey[257];   // exponential, 256 layers
  ZigguratTables() {
    // normal: r = 3.442619855899, v = 9.91256303526217e-3 (Marsaglia & Tsang 2000)
    const double r = 3.442619855899, v = 9.91256303526217e-3;
    nx[1] = r; ny[1] = std::exp(-0.5 * r * r);
    nx[0] = v / ny[1]; ny[0] = 0.0;
    for (int i = 2; i < 128; ++i) {
      ny[i] = ny[i - 1] + v / nx[i - 1];
      nx[i] = std::sqrt(-2.0 * std::log(ny[i]));
    }
    nx[128] = 0.0; ny[128] = 1.0;
    // exponential: r = 7.69711747013104972, v = 3.949659822581572e-3
    const double re = 7.69711747013104972, ve = 3.949659822581572e-3;
    ex[1] = re; ey[1] = std::exp(-re);
    ex[0] = ve / ey[1]; ey[0] = 0.0;
    for (int i = 2; i < 256; ++i) {
      ey[i] = ey[i - 1] + ve / ex[i - 1];
      ex[i] = -std::log(ey[i]);
    }
    ex[256] = 0.0; ey[256] = 1.0;
  }
};

inline const ZigguratTables& zig_tables() { static ZigguratTables t; return t; }

// boost::random::exponential_distribution<double>(lambda) (ziggurat form)
inline double boost_exponential(Ecuyer1988& eng, double lambda) {
  const ZigguratTables& T = zig_tables();
  double shift = 0.0;
  for (;;) {
    double u; int i;
    eng.int_float_pair(u, i);
    double x = u * T.ex[i];
    if (x < T.ex[i + 1]) return (shift + x) / lambda;
    if (i == 0) { shift += T.ex[1]; continue; }
    double y01 = eng.uniform01();
    double y = T.ey[i] + y01 * (T.ey[i + 1] - T.ey[i]);
    if (y < std::exp(-x)) return (shift + x) / lambda;
  }
}

// boost::normal_distribution<double>(0, 1)
inline double boost_normal(Ecuyer1988& eng) {
  const ZigguratTables& T = zig_tables();
  for (;;) {
    double u; int b;
    eng.int_float_pair(u, b);
    int sign = (b & 1) * 2 - 1;
    int i = b >> 1;
    double x = u * T.nx[i];
    if (x < T.nx[i + 1]) return x * sign;
    if (i == 0) {
      const double tail_start = T.nx[1];
      for (;;) {
        double tx = boost_exponential(eng, tail_start);
        double ty = boost_exponential(eng, 1.0);
        if (2.0 * ty > tx * tx) return (tx + tail_start) * sign;
      }
    }
    double y01 = eng.uniform01();
    double y = T.ny[i] + y01 * (T.ny[i + 1] - T.ny[i]);
    if (y < std::exp(-0.5 * x * x)) return x * sign;
  }
}

}  // namespace oracle
#endif
