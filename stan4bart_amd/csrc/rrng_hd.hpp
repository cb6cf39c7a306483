// R-compatible random stream for the BART block, usable from device code.
//
// The reference's BART block consumes R's global generator (reference src/init.cpp:259,750
// GetRNGstate; dbarts draws unif_rand/norm_rand/exp_rand from it in single-chain mode).  On the
// MI355X path the Mersenne-Twister state lives in device memory and one lane of the per-tree
// `decide` kernel advances it, so accept/reject decisions never leave the GPU.  The same code
// compiles for the host (tree initialisation at create time, CPU unit tests of the host logic).
#ifndef S4B_RRNG_HD_HPP
#define S4B_RRNG_HD_HPP

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define S4B_HD __host__ __device__
#else
#define S4B_HD
#endif

// S4B_UNI(x): on the device the control code runs wave-uniformly (every lane computes the same scalar
// program); values that come out of memory are re-broadcast from lane 0 so that the compiler can prove it
// and emit scalar branches / SGPR loop counters instead of exec-mask divergence handling.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int s4b_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned s4b_uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ double s4b_uni(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
#define S4B_UNI(x) s4b_uni(x)
#else
#define S4B_UNI(x) (x)
#endif

namespace s4b {

struct MTState {
  uint32_t mt[624];
  int32_t mti;
  int32_t pad;   // device only: 1 while a full wave owns the state (lane-parallel block regeneration), else 0
};

#if defined(__HIPCC__)
// The block recurrence mt[k] <- mt[k+397 mod 624] ^ f(mt[k], mt[k+1 mod 624]) only reaches back 227 positions
// for its updated inputs, so 64 consecutive elements can be produced at once (reads of a chunk happen before
// its writes: one wave, lockstep).  Same values as the sequential loop.
__device__ inline void mt_regenerate_wave(MTState* s) {
  uint32_t* mt = s->mt;
  const int lane = (int)(threadIdx.x & 63);
  for (int base = 0; base < 624; base += 64) {
    const int k = base + lane;
    const bool in = k < 624;
    const int k1 = (k + 1 >= 624) ? k + 1 - 624 : k + 1, k397 = (k + 397 >= 624) ? k + 397 - 624 : k + 397;
    uint32_t a = in ? mt[k] : 0u, b = in ? mt[k1] : 0u, c = in ? mt[k397] : 0u;
    uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    uint32_t v = c ^ (y >> 1);
    if (y & 1u) v ^= 0x9908b0dfu;
    if (in) mt[k] = v;
    // the next chunk reads what other lanes have just written: order the LDS traffic wave-wide (the compiler
    // would otherwise be free to hoist the next chunk's loads, which never alias this lane's own store)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  s->mti = 0;
}
#endif

S4B_HD inline void mt_regenerate(MTState* s) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (S4B_UNI(s->pad) == 1) { mt_regenerate_wave(s); return; }
#endif
  uint32_t* mt = s->mt;
  for (int k = 0; k < 624; ++k) {
    uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
    uint32_t v = mt[(k + 397) % 624] ^ (y >> 1);
    if (y & 1u) v ^= 0x9908b0dfu;
    mt[k] = v;
  }
  s->mti = 0;
}

S4B_HD inline uint32_t mt_next(MTState* s) {
  int k = S4B_UNI((int)s->mti);
  if (k >= 624) { mt_regenerate(s); k = 0; }
  uint32_t y = S4B_UNI(s->mt[k]);
  s->mti = k + 1;
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

template <class RNG> S4B_HD inline double r_unif(RNG* s) {
  const double half_ulp = 0.5 * 2.328306437080797e-10;
  double v = (double)mt_next(s) * 2.3283064365386963e-10;
  if (v <= 0.0) return half_ulp;
  if (1.0 - v <= 0.0) return 1.0 - half_ulp;
  return v;
}

// quantile of N(0,1): AS241 (PPND16), as R's qnorm5
S4B_HD inline double r_qnorm(double p) {
  double q = p - 0.5;
  if (fabs(q) <= 0.425) {
    double r = 0.180625 - q * q;
    double num = 2509.0809287301226727;
    num = num * r + 33430.575583588128105; num = num * r + 67265.770927008700853;
    num = num * r + 45921.953931549871457; num = num * r + 13731.693765509461125;
    num = num * r + 1971.5909503065514427; num = num * r + 133.14166789178437745;
    num = num * r + 3.387132872796366608;
    double den = 5226.495278852545925;
    den = den * r + 28729.085735721942674; den = den * r + 39307.89580009271061;
    den = den * r + 21213.794301586595867; den = den * r + 5394.1960214247511077;
    den = den * r + 687.1870074920579083; den = den * r + 42.313330701600911252;
    den = den * r + 1.0;
    return q * num / den;
  }
  double r = q < 0.0 ? p : 1.0 - p;
  r = sqrt(-log(r));
  double val;
  if (r <= 5.0) {
    r -= 1.6;
    double num = 7.7454501427834140764e-4;
    num = num * r + 0.0227238449892691845833; num = num * r + 0.24178072517745061177;
    num = num * r + 1.27045825245236838258; num = num * r + 3.64784832476320460504;
    num = num * r + 5.7694972214606914055; num = num * r + 4.6303378461565452959;
    num = num * r + 1.42343711074968357734;
    double den = 1.05075007164441684324e-9;
    den = den * r + 5.475938084995344946e-4; den = den * r + 0.0151986665636164571966;
    den = den * r + 0.14810397642748007459; den = den * r + 0.68976733498510000455;
    den = den * r + 1.6763848301838038494; den = den * r + 2.05319162663775882187;
    den = den * r + 1.0;
    val = num / den;
  } else {
    r -= 5.0;
    double num = 2.01033439929228813265e-7;
    num = num * r + 2.71155556874348757815e-5; num = num * r + 0.0012426609473880784386;
    num = num * r + 0.026532189526576123093; num = num * r + 0.29656057182850489123;
    num = num * r + 1.7848265399172913358; num = num * r + 5.4637849111641143699;
    num = num * r + 6.6579046435011037772;
    double den = 2.04426310338993978564e-15;
    den = den * r + 1.4215117583164458887e-7; den = den * r + 1.8463183175100546818e-5;
    den = den * r + 7.868691311456132591e-4; den = den * r + 0.0148753612908506148525;
    den = den * r + 0.13692988092273580531; den = den * r + 0.59983220655588793769;
    den = den * r + 1.0;
    val = num / den;
  }
  return q < 0.0 ? -val : val;
}

// norm_rand() with N01_kind = INVERSION
template <class RNG> S4B_HD inline double r_norm(RNG* s) {
  const double BIG = 134217728.0;
  double u = r_unif(s);
  u = (double)(int)(BIG * u) + r_unif(s);
  return r_qnorm(u / BIG);
}

// exp_rand() (Ahrens & Dieter 1972 as in R's sexp.c).  q[k] = sum_{i=1}^{k+1} log(2)^i / i!  — through a switch, not a local
// array: a dynamically indexed local array lives in scratch (global) memory on the device, microseconds per access
S4B_HD inline double r_exp_q(int i) {
  switch (i) {
    case 0: return 0.6931471805599453; case 1: return 0.9333736875190459; case 2: return 0.9888777961838675;
    case 3: return 0.9984589039328340; case 4: return 0.9998292811061389; case 5: return 0.9999833164100727;
    case 6: return 0.9999985691438767; case 7: return 0.9999998906925558; case 8: return 0.9999999924734159;
    case 9: return 0.9999999995283275; case 10: return 0.9999999999728814; case 11: return 0.9999999999985598;
    case 12: return 0.9999999999999289; case 13: return 0.9999999999999968; case 14: return 0.9999999999999999;
    default: return 1.0000000000000000;
  }
}
template <class RNG> S4B_HD inline double r_exp(RNG* s) {
  const double q0 = 0.6931471805599453;
  double a = 0.0;
  double u = r_unif(s);
  while (u <= 0.0 || u >= 1.0) u = r_unif(s);
  for (;;) { u += u; if (u > 1.0) break; a += q0; }
  u -= 1.0;
  if (u <= q0) return a + u;
  int i = 0;
  double ustar = r_unif(s), umin = ustar;
  do { ustar = r_unif(s); if (umin > ustar) umin = ustar; ++i; } while (u > r_exp_q(i));
  return a + umin * q0;
}

// rgamma(shape, scale) from R's stream (nmath/rgamma.c; dbarts' gamma draws come from it when a chain uses R's native generator): used
// once per sweep by the k hyperprior (k_draw_k).  shape >= 1: Ahrens & Dieter's GD — a (s, 1/2)-normal deviate squared, accepted at
// once for t >= 0, by a squeeze, by the quotient test, else a double-exponential rejection loop; shape < 1: their GS.
struct GammaQuotient {          // the parts of GD that depend on the shape only
  double s2, s, q0, b, si, c;
  S4B_HD explicit GammaQuotient(double a) {
    s2 = a - 0.5; s = sqrt(s2);
    const double r = 1 / a;
    q0 = ((((((2.424e-4 * r + 2.4511e-4) * r + -7.388e-5) * r + 0.00144121) * r + 0.00801191) * r + 0.02083148) * r + 0.04166669) * r;
    if (a <= 3.686) { b = 0.463 + s + 0.178 * s2; si = 1.235; c = 0.195 / s - 0.079 + 0.16 * s; }
    else if (a <= 13.022) { b = 1.654 + 0.0076 * s2; si = 1.68 / s + 0.275; c = 0.062 / s + 0.024; }
    else { b = 1.77; si = 0.75; c = 0.1515 / s; }
  }
  S4B_HD double q(double t) const {
    const double v = t / (s + s);
    if (fabs(v) <= 0.25)
      return q0 + 0.5 * t * t * ((((((0.1233795 * v + -0.1367177) * v + 0.1423657) * v + -0.1662921) * v + 0.2000062) * v + -0.250003) * v + 0.3333333) * v;
    return q0 - s * t + 0.25 * t * t + (s2 + s2) * log(1.0 + v);
  }
};
template <class RNG> S4B_HD inline double r_gamma(RNG* g, double a, double scale) {
  if (a != a || scale != scale) return a + scale;
  if (a <= 0.0 || scale <= 0.0) return (scale == 0.0 || a == 0.0) ? 0.0 : (a - a) / (a - a);
  if (a - a != 0.0 || scale - scale != 0.0) return scale > a ? scale : a;      // (an infinite argument: +Inf)
  if (a < 1) {
    const double e = 1.0 + 0.36787944117144233 * a;
    double x;
    for (;;) {
      const double p = e * r_unif(g);
      if (p >= 1.0) { x = -log((e - p) / a); if (r_exp(g) >= (1.0 - a) * log(x)) break; }
      else { x = exp(log(p) / a); if (r_exp(g) >= x) break; }
    }
    return scale * x;
  }
  const GammaQuotient G(a);
  double t = r_norm(g);
  const double x0 = G.s + 0.5 * t, first = x0 * x0;
  if (t >= 0) return scale * first;
  double u = r_unif(g);
  if ((5.656854 - G.s * 12) * u <= t * t * t) return scale * first;
  if (x0 > 0.0 && log(1.0 - u) <= G.q(t)) return scale * first;
  for (;;) {
    const double e = r_exp(g);
    u = r_unif(g);
    u = u + u - 1.0;
    t = u < 0.0 ? G.b - G.si * e : G.b + G.si * e;
    if (t < -0.71874483771719) continue;
    const double q = G.q(t);
    if (q <= 0.0) continue;
    if (G.c * fabs(u) <= expm1(q) * exp(e - 0.5 * t * t)) break;
  }
  const double x = G.s + 0.5 * t;
  return scale * x * x;
}

// uniform integer in [lo, hi) the way dbarts' ext_rng does it: lo + (int64)(u * range)
template <class RNG> S4B_HD inline int r_unif_int(RNG* s, int lo, int hi_excl) {
  return lo + (int)(r_unif(s) * (double)(hi_excl - lo));
}

}  // namespace s4b
#endif
