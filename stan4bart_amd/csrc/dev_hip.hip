// HIP device layer of the stan4bart Gibbs hot path for MI355X (gfx950, wave64).
//
// Replaces, on the device, the O(N) work the reference does on one CPU thread per chain:
//   * dbarts' per-tree passes behind bartFunctions.runSamplerWithResults (reference src/init.cpp:824):
//     partial residual, per-leaf sufficient statistics, proposal statistics, leaf-value scatter
//       -> k_stats / k_control / k_apply          (algorithmic bytes 22 N per tree)
//   * dbarts setOffset / setSigma (reference src/init.cpp:799,817) -> k_param_mean / k_scale / k_rescale
//   * the residual hand-off (reference src/stan_files/continuous.hpp:3631-3768, src/init.cpp:828-842)
//     and the O(N) sums of the Stan log density (continuous.hpp:2438-2470)
//       -> k_stan_inputs / k_zt_chunks / k_leapfrog  (§8 a7, a11, a12)
// Design rules: all per-observation arrays are structure-of-arrays and read/written as
// 8–32 B per lane; the binned predictors are u16 with one column contiguous; every reduction has a
// fixed grid and a fixed combination order, so a chain is reproducible run to run; accept/reject and
// the R-compatible random stream stay on the device (one lane of k_control), so a sweep over all
// trees is a pure launch sequence with no host round trip.
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <chrono>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef S4B_CONTROL_TIMING
// (measurement build) birth proposals of candidate wave 1 in workgroup 0: time stamps inside propose()
__device__ unsigned long long g_prop[32];
#define S4B_PROP_T(i) do { if (blockIdx.x == 0 && (threadIdx.x >> 6) == 1 && (threadIdx.x & 63) == 0) { atomicAdd(&g_prop[i], (unsigned long long)wall_clock64()); if ((i) == 7) atomicAdd(&g_prop[15], 1ull); if ((i) <= 1) atomicAdd(&g_prop[8 + (i)], 1ull); } } while (0)
#endif
#ifdef S4B_CONTROL_TIMING
__device__ unsigned long long g_dec[16];
#define S4B_DEC_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { atomicAdd(&g_dec[i], (unsigned long long)wall_clock64()); if ((i) == 0) atomicAdd(&g_dec[15], 1ull); } } while (0)
#endif
#ifdef S4B_SWEEP_TIMING
// (measurement build `make sweeptiming`) time stamps inside decide() of the decider wave of workgroup 100 of k_sweep: sums of absolute clock
// values per slot ([15] = calls), printed as differences by profile_sweep_persistent; the kernel lives in the dev_sweep.hip translation unit,
// which owns (and reads back) its own copy of this symbol
static __device__ unsigned long long g_dec[16];
// ([i] = sum of (now - entry) over the calls that reach stamp i, [8 + i] = how many did; [0] holds the entry time of the call in flight)
#define S4B_DEC_T(i) do { if (blockIdx.x == 100 && threadIdx.x == 0) { if ((i) == 0) { g_dec[0] = (unsigned long long)wall_clock64(); atomicAdd(&g_dec[15], 1ull); } \
                                                                      else { atomicAdd(&g_dec[i], (unsigned long long)wall_clock64() - g_dec[0]); atomicAdd(&g_dec[7 + (i)], 1ull); } } } while (0)
#endif
#include "sampler_core.hpp"

namespace s4b {

#define HIP_OK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr);     \
  } while (0)

constexpr int BLOCK = 256;          // 4 waves
constexpr int GRID_MAX = 2048;      // upper bound on workgroups of the O(N) kernels (8 per CU); also the number of partials per bin
constexpr int NBMAX = 16;           // bins accumulated in registers per pass

// ------------------------------------------------------------------------------------------------
// wave / block reductions with a fixed order (deterministic)
// value of lane (l ^ O), O a power of two: register-to-register on gfx950 (DPP within rows of 16 lanes, the permlane swaps across
// them) instead of a trip through the LDS crossbar (ds_bpermute) — the dependent exchange chains of the reductions are several
// times shorter
template <int O>
__device__ __forceinline__ int wave_xor(int v) {
  static_assert(O == 1 || O == 2 || O == 4 || O == 8 || O == 16 || O == 32, "power of two below 64");
  if constexpr (O == 1) return __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false);           // quad_perm [1,0,3,2]
  else if constexpr (O == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
  else if constexpr (O == 4) {   // lanes with bit 2 clear read lane + 4 (row_shl:4), the others lane - 4 (row_shr:4)
    const int a = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0x5, false);
    return __builtin_amdgcn_update_dpp(a, v, 0x114, 0xf, 0xa, false);
  } else if constexpr (O == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);   // row_ror:8
  else if constexpr (O == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // {odd rows <- even rows of the copy, even rows <- odd rows}
    return ((threadIdx.x & 16) != 0) ? (int)r[0] : (int)r[1];
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return ((threadIdx.x & 32) != 0) ? (int)r[0] : (int)r[1];
  }
}
template <int O>
__device__ __forceinline__ double wave_xor(double v) {
  return __hiloint2double(wave_xor<O>(__double2hiint(v)), wave_xor<O>(__double2loint(v)));
}
__device__ __forceinline__ double wave_sum(double v) {   // same pairing and order as the xor butterfly 32, 16, ..., 1
  v += wave_xor<32>(v); v += wave_xor<16>(v); v += wave_xor<8>(v); v += wave_xor<4>(v); v += wave_xor<2>(v); v += wave_xor<1>(v);
  return v;
}
// NB independent sums over the 64 lanes at once: each halving step keeps one half of the values and hands the
// other half to the partner lane, so the cross-lane traffic is NB-1 + log2(64/NB) exchanges instead of 6 NB.
// Afterwards v[0] of lane l is the wave total of value wave_bin_of_lane<NB>(l).  Fixed order: deterministic.
template <int CNT, int O, int NB, class T>
__device__ __forceinline__ void wave_sum_bins_step(T (&v)[NB], int lane) {
  if constexpr (CNT > 1) {
    constexpr int h = CNT / 2;
    const bool upper = (lane & O) != 0;
#pragma unroll
    for (int j = 0; j < h; ++j) {
      const T send = upper ? v[j] : v[j + h];
      const T keep = upper ? v[j + h] : v[j];
      v[j] = keep + wave_xor<O>(send);
    }
    wave_sum_bins_step<h, O / 2, NB, T>(v, lane);
  } else if constexpr (O > 0) {
    v[0] += wave_xor<O>(v[0]);
    wave_sum_bins_step<1, O / 2, NB, T>(v, lane);
  }
}
template <int NB, class T>
__device__ __forceinline__ void wave_sum_bins(T (&v)[NB], int lane) { wave_sum_bins_step<NB, 32, NB, T>(v, lane); }
template <int NB>
__device__ __forceinline__ int wave_bin_of_lane(int lane) {
  // NB = 2^q: the q halving steps use lane bits 5, 4, ..., 6-q for value-index bits q-1, ..., 0
  constexpr int q = NB == 16 ? 4 : NB == 8 ? 3 : NB == 4 ? 2 : NB == 2 ? 1 : 0;
  return q == 0 ? 0 : (lane >> (6 - q)) & (NB - 1);
}
__device__ __forceinline__ double wave_min(double v) {
  v = fmin(v, wave_xor<32>(v)); v = fmin(v, wave_xor<16>(v)); v = fmin(v, wave_xor<8>(v)); v = fmin(v, wave_xor<4>(v)); v = fmin(v, wave_xor<2>(v)); v = fmin(v, wave_xor<1>(v));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, wave_xor<32>(v)); v = fmax(v, wave_xor<16>(v)); v = fmax(v, wave_xor<8>(v)); v = fmax(v, wave_xor<4>(v)); v = fmax(v, wave_xor<2>(v)); v = fmax(v, wave_xor<1>(v));
  return v;
}

#ifdef S4B_CONTROL_TIMING
__device__ long long g_dbg[40];
#define S4B_TICK(x) long long x = wall_clock64()
#define S4B_PTICK(x) __builtin_amdgcn_sched_barrier(0); long long x = wall_clock64(); __builtin_amdgcn_sched_barrier(0)
#else
#define S4B_TICK(x)
#define S4B_PTICK(x)
#endif
// ------------------------------------------------------------------------------------------------
// k_tree: the O(N) kernel of one tree update.  One pass over the observations:
//   apply half  (tree t-1, already decided):  R_i += mu_old[leaf] - mu_new[leaf'], relabel under the accepted move
//   stats half  (tree t, proposal pending):   (count, sum) of r_i = R_i + mu_t[leaf_t(i)] per A bin (current leaf)
//                                             and per B bin (leaf the proposal would create under its root)
// so the residual is read once and written once per tree update (22 B of algorithmic traffic per
// observation: R 8+8, leaf(t-1) 2, leaf(t) 2, binned predictor 2).
// Per-node tables live in LDS as packed 16-byte records (one ds_read_b128 per observation and half).
struct __attribute__((aligned(16))) NodeS { double mu; int16_t binA, binB; int16_t insub; int16_t pad; };   // stats half
struct __attribute__((aligned(8))) NodeP { int16_t var; uint16_t cut; int16_t left, right; };                // 8 B: routing
static_assert(sizeof(NodeS) == 16 && sizeof(NodeP) == 8, "LDS record sizes are part of the carve layout");
struct __attribute__((aligned(16))) NodeA { double muOld, muNew; };                                           // apply half
static_assert(sizeof(NodeA) == 16, "LDS record sizes are part of the carve layout");

typedef unsigned short us4_t __attribute__((ext_vector_type(4)));
static size_t apply_lds_bytes(int nc) { return (size_t)nc * 25 + 16; }

struct TreeLds {
  NodeS* S; NodeP* SP;     // stats: records + proposed-tree routing
  NodeA* A; NodeP* AP; uint8_t* Ain;   // apply: records + new-tree routing + re-route flags
  double* redS; int* redN; double* redW;
};
__host__ __device__ static inline size_t tree_lds_bytes(int nc) {
  return (size_t)nc * (16 + 8 + 16 + 8) + ((size_t)nc + 15) / 16 * 16 + 4 * 16 * 8 + 4 * 16 * 4 + 4 * 16 * 8;
}
__device__ __forceinline__ TreeLds carve_tree(unsigned char* base, int nc) {
  TreeLds L;
  L.S = (NodeS*)base; base += (size_t)nc * 16;
  L.A = (NodeA*)base; base += (size_t)nc * 16;
  L.SP = (NodeP*)base; base += (size_t)nc * 8;
  L.AP = (NodeP*)base; base += (size_t)nc * 8;
  L.Ain = (uint8_t*)base; base += ((size_t)nc + 15) / 16 * 16;
  L.redS = (double*)base; base += 4 * 16 * 8;
  L.redW = (double*)base; base += 4 * 16 * 8;
  L.redN = (int*)base;
  return L;
}

template <int NB, bool APPLY, bool WEIGHTED>
__device__ __forceinline__ void tree_pass(const BartArrays& a, int t, const TreeLds& L, int root, int prevRoot, int prevAcc, int base,
                                          int nbTotal) {
  // per-thread bin counts are small (a thread sees at most 4 * 255 observations, enforced by the launch geometry): two
  // 16-bit counters per register, unpacked before the block reduction
  double accS[NB]; unsigned accP[(NB + 1) / 2]; double accW[WEIGHTED ? NB : 1];
#pragma unroll
  for (int k = 0; k < NB; ++k) { accS[k] = 0.0; if (WEIGHTED) accW[k] = 0.0; }
#pragma unroll
  for (int k = 0; k < (NB + 1) / 2; ++k) accP[k] = 0u;
  const double* __restrict__ W = a.wts;
  const int64_t nQuads = (a.n + 3) >> 2;
  const uint16_t* __restrict__ leafPlane = a.leaf + (size_t)t * a.npad;
  uint16_t* __restrict__ prevPlane = a.leaf + (size_t)(t > 0 ? t - 1 : 0) * a.npad;
  double* __restrict__ R = a.R;
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  int64_t qd = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  // software pipeline, two quads ahead: the loads of quads q + stride and q + 2 stride are in flight while quad q is processed
  // the first routing level of both halves is known before the loop (the root of the proposed / accepted subtree): its
  // predictor values travel with the quad instead of being a dependent load per observation
  const int rootVarS = L.SP[root].var, rootCutS = L.SP[root].cut, rootLS = L.SP[root].left, rootRS = L.SP[root].right;
  const int rootVarA = (APPLY && prevAcc) ? (int)L.AP[prevRoot].var : -1;
  const int rootCutA = APPLY ? (int)L.AP[prevRoot].cut : 0, rootLA = APPLY ? (int)L.AP[prevRoot].left : 0, rootRA = APPLY ? (int)L.AP[prevRoot].right : 0;
  const uint16_t* __restrict__ colS = a.xbin + (size_t)(rootVarS >= 0 ? rootVarS : 0) * a.npad;
  const uint16_t* __restrict__ colA = a.xbin + (size_t)(rootVarA >= 0 ? rootVarA : 0) * a.npad;
  struct Quad { double2 r01, r23; us4_t lf4, pl4, xs4, xa4; double2 w01, w23; };
  auto fetch = [&](int64_t q, Quad& o) {
    if (q < nQuads) {
      const int64_t i0 = q << 2;
      if (rootVarS >= 0) o.xs4 = *reinterpret_cast<const us4_t*>(colS + i0);
      if (APPLY && rootVarA >= 0) o.xa4 = *reinterpret_cast<const us4_t*>(colA + i0);
      o.r01 = *reinterpret_cast<const double2*>(R + i0); o.r23 = *reinterpret_cast<const double2*>(R + i0 + 2);
      if (WEIGHTED) { o.w01 = *reinterpret_cast<const double2*>(W + i0); o.w23 = *reinterpret_cast<const double2*>(W + i0 + 2); }
      o.lf4 = __builtin_nontemporal_load(reinterpret_cast<const us4_t*>(leafPlane + i0));
      if (APPLY) o.pl4 = __builtin_nontemporal_load(reinterpret_cast<const us4_t*>(prevPlane + i0));
    }
  };
  Quad qa, qb;
  qa.w01 = qa.w23 = qb.w01 = qb.w23 = make_double2(1.0, 1.0);
  fetch(qd, qa); fetch(qd + stride, qb);
  for (; qd < nQuads; qd += stride) {
    const int64_t i0 = qd << 2;
    double rr[4] = {qa.r01.x, qa.r01.y, qa.r23.x, qa.r23.y};
    const double ww[4] = {qa.w01.x, qa.w01.y, qa.w23.x, qa.w23.y};
    const unsigned lf[4] = {qa.lf4.x, qa.lf4.y, qa.lf4.z, qa.lf4.w};
    unsigned pl[4] = {qa.pl4.x, qa.pl4.y, qa.pl4.z, qa.pl4.w};
    const unsigned xs[4] = {qa.xs4.x, qa.xs4.y, qa.xs4.z, qa.xs4.w};
    const unsigned xa[4] = {qa.xa4.x, qa.xa4.y, qa.xa4.z, qa.xa4.w};
    qa = qb;
    fetch(qd + 2 * stride, qb);
    const int valid = (a.n - i0) >= 4 ? 4 : (int)(a.n - i0);
    if (APPLY) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned l = e < valid ? pl[e] : 0u;
        const NodeA na = L.A[l];
        unsigned nl = l;
        double muNew = na.muNew;
        if (prevAcc && e < valid && L.Ain[l]) {
          int nd = prevRoot;
          if (rootVarA >= 0) nd = (xa[e] <= (unsigned)rootCutA) ? rootLA : rootRA;
          NodeP p = L.AP[nd];
          while (p.var >= 0) {
            const unsigned x = a.xbin[(size_t)p.var * a.npad + (size_t)(i0 + e)];
            nd = (x <= (unsigned)p.cut) ? p.left : p.right;
            p = L.AP[nd];
          }
          nl = (unsigned)nd;
          muNew = L.A[nl].muNew;
        }
        rr[e] = (rr[e] + na.muOld) - muNew;
        pl[e] = nl;
      }
      *reinterpret_cast<double2*>(R + i0) = make_double2(rr[0], rr[1]);
      *reinterpret_cast<double2*>(R + i0 + 2) = make_double2(rr[2], rr[3]);
      if (prevAcc) { us4_t o4; o4.x = (unsigned short)pl[0]; o4.y = (unsigned short)pl[1]; o4.z = (unsigned short)pl[2]; o4.w = (unsigned short)pl[3];
                     __builtin_nontemporal_store(o4, reinterpret_cast<us4_t*>(prevPlane + i0)); }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool ok = e < valid;
      const NodeS ns = L.S[ok ? lf[e] : 0u];
      const double r = WEIGHTED ? (rr[e] + ns.mu) * ww[e] : rr[e] + ns.mu;
      const int ba = ok ? (int)ns.binA - base : -1;
      int bb = -1 - base;
      if (ok && ns.insub) {
        int nd = root;
        if (rootVarS >= 0) nd = (xs[e] <= (unsigned)rootCutS) ? rootLS : rootRS;
        NodeP p = L.SP[nd];
        while (p.var >= 0) {
          const unsigned x = a.xbin[(size_t)p.var * a.npad + (size_t)(i0 + e)];
          nd = (x <= (unsigned)p.cut) ? p.left : p.right;
          p = L.SP[nd];
        }
        bb = (int)L.S[nd].binB - base;
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const bool m = (ba == k) | (bb == k);
        accS[k] += m ? r : 0.0;
        accP[k >> 1] += m ? ((k & 1) ? 65536u : 1u) : 0u;
        if (WEIGHTED) accW[k] += m ? ww[e] : 0.0;
      }
    }
  }
  // block reduction, fixed order: transposed halving inside each wave, then waves 0..3 in order
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int accN[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) accN[k] = (int)((accP[k >> 1] >> (16 * (k & 1))) & 0xffffu);
  wave_sum_bins<NB>(accS, lane);
  wave_sum_bins<NB>(accN, lane);
  if (WEIGHTED) wave_sum_bins<WEIGHTED ? NB : 1>(accW, lane);
  if ((lane & (64 / NB - 1)) == 0) { const int k = wave_bin_of_lane<NB>(lane); L.redS[wv * NBMAX + k] = accS[0]; L.redN[wv * NBMAX + k] = accN[0];
                                     if (WEIGHTED) L.redW[wv * NBMAX + k] = accW[0]; }
  __syncthreads();
  if ((int)threadIdx.x < NB && base + (int)threadIdx.x < nbTotal) {
    const int k = threadIdx.x;
    const double s = ((L.redS[k] + L.redS[NBMAX + k]) + L.redS[2 * NBMAX + k]) + L.redS[3 * NBMAX + k];
    const int c = L.redN[k] + L.redN[NBMAX + k] + L.redN[2 * NBMAX + k] + L.redN[3 * NBMAX + k];
    a.partSum[(size_t)(base + k) * a.grid + blockIdx.x] = s;
    a.partCnt[(size_t)(base + k) * a.grid + blockIdx.x] = (double)c;
    if (WEIGHTED) a.partWt[(size_t)(base + k) * a.grid + blockIdx.x] = ((L.redW[k] + L.redW[NBMAX + k]) + L.redW[2 * NBMAX + k]) + L.redW[3 * NBMAX + k];
  }
  __syncthreads();
}

// APPLY = false for the first tree of a sweep (nothing pending)
// WEIGHTED: observation weights (8 more bytes per observation; bins per pass capped at 8 to stay within the register budget)
template <bool APPLY, bool WEIGHTED>
__global__ __launch_bounds__(BLOCK) void k_tree(BartArrays a, int t) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  S4B_TICK(tq0);
  const StepScratch& c = a.sc[t & 1];
  const Proposal pr = *c.prop;
  const TreeLds L = carve_tree(smem, a.nc);
  const double* mu = a.mu + (size_t)t * a.nc;
  for (int i = threadIdx.x; i < pr.hwm; i += BLOCK) {
    NodeS s; s.mu = mu[i]; s.binA = c.binA[i]; s.binB = c.binB[i]; s.insub = c.insub[i]; s.pad = 0;
    NodeP p; p.var = c.pvar[i]; p.cut = c.pcut[i]; p.left = c.pleft[i]; p.right = c.pright[i];
    L.S[i] = s; L.SP[i] = p;
  }
  int prevAcc = 0, prevRoot = 0;
  if (APPLY) {
    const StepScratch& cp = a.sc[(t - 1) & 1];
    prevAcc = *cp.accepted; prevRoot = cp.prop->node;
    const int hp = cp.prop->hwm > a.hwm[t - 1] ? cp.prop->hwm : a.hwm[t - 1];
    const size_t o = (size_t)(t - 1) * a.nc;
    for (int i = threadIdx.x; i < hp; i += BLOCK) {
      NodeA q; q.muOld = cp.muOld[i]; q.muNew = a.mu[o + i];
      NodeP p; p.var = a.var[o + i]; p.cut = a.cut[o + i]; p.left = a.left[o + i]; p.right = a.right[o + i];
      L.A[i] = q; L.AP[i] = p; L.Ain[i] = cp.insub[i];
    }
  }
  __syncthreads();
  S4B_TICK(tq1);
  const int nb = pr.nbA + pr.nbB;
  if (nb <= 4) tree_pass<4, APPLY, WEIGHTED>(a, t, L, pr.node, prevRoot, prevAcc, 0, nb);
  else if (nb <= 8 || WEIGHTED) {
    tree_pass<8, APPLY, WEIGHTED>(a, t, L, pr.node, prevRoot, prevAcc, 0, nb);
    for (int base = 8; base < nb; base += 8) tree_pass<8, false, WEIGHTED>(a, t, L, pr.node, prevRoot, prevAcc, base, nb);
  } else {
    tree_pass<NBMAX, APPLY, WEIGHTED>(a, t, L, pr.node, prevRoot, prevAcc, 0, nb);
    // further bin passes read the residual this thread has just written
    for (int base = NBMAX; base < nb; base += NBMAX) tree_pass<NBMAX, false, WEIGHTED>(a, t, L, pr.node, prevRoot, prevAcc, base, nb);
  }
#ifdef S4B_CONTROL_TIMING
  { S4B_TICK(tq2);
    if (APPLY && threadIdx.x == 0) { atomicAdd((unsigned long long*)&g_dbg[8], (unsigned long long)(tq1 - tq0)); atomicAdd((unsigned long long*)&g_dbg[9], (unsigned long long)(tq2 - tq1));
      atomicAdd((unsigned long long*)&g_dbg[10], 1ull);
    } }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_control: combines the per-workgroup partials in a fixed order, then wave 0 runs the Metropolis-Hastings
// control code for tree t (decide + leaf draws) and draws the proposal of tree `next`.
//
// The control code is sequential and branchy; run from memory (even LDS) every dependent access costs
// 64+ cycles.  Here every small array of the step (tree structure, proposed tree, bin maps, leaf values) is
// held in ONE VGPR spread across the 64 lanes of the wave — element i lives in lane i — and read with
// v_readlane / written with v_writelane.  All 64 lanes execute the same (wave-uniform) scalar program, so an
// "array access" is a 1-instruction register access.  Trees with more than 64 node slots in use take the
// slower global-memory path (same source, pointer storage).
template <class T>
struct WaveArr {   // up to 64 elements of an integer type of <= 32 bits
  int r;
  __device__ __forceinline__ T get(int i) const { return (T)__builtin_amdgcn_readlane(r, i); }
  __device__ __forceinline__ void set(int i, T v) { r = ((int)(threadIdx.x & 63) == i) ? (int)v : r; }
};
struct WaveArrD {  // up to 64 doubles
  int lo, hi;
  __device__ __forceinline__ double get(int i) const {
    return __hiloint2double(__builtin_amdgcn_readlane(hi, i), __builtin_amdgcn_readlane(lo, i));
  }
  __device__ __forceinline__ void set(int i, double v) {
    const bool me = (int)(threadIdx.x & 63) == i;
    lo = me ? __double2loint(v) : lo;
    hi = me ? __double2hiint(v) : hi;
  }
  __device__ __forceinline__ void load(double v) { lo = __double2loint(v); hi = __double2hiint(v); }
  __device__ __forceinline__ double mine() const { return __hiloint2double(hi, lo); }
};
typedef TreeT<WaveArr<int16_t>, WaveArr<uint16_t>> WaveTree;
typedef StepTablesT<WaveTree, WaveArr<int16_t>, WaveArr<uint8_t>> WaveTables;

// whole-register copy (overload picked over the element-wise template)
__device__ __forceinline__ void tv_copy(const WaveTree& src, WaveTree& dst, int) {
  dst.var.r = src.var.r; dst.cut.r = src.cut.r; dst.left.r = src.left.r; dst.right.r = src.right.r; dst.parent.r = src.parent.r;
  dst.na.r = src.na.r; dst.dep.r = src.dep.r;
}
__device__ __forceinline__ void copy_leaf_values(const WaveArrD& mu, WaveArrD& muOld, int hwm, int) {
  const bool in = (int)(threadIdx.x & 63) < hwm;
  muOld.lo = in ? mu.lo : 0; muOld.hi = in ? mu.hi : 0;
}
// lane-parallel versions of the batched math of decide(): lane b / lane i owns bin b / leaf i
__device__ __forceinline__ void bins_loglik(const WaveArrD& binCnt, const WaveArrD& binSum, const WaveArrD& binWt, int, double sigma2, double prec, WaveArrD& out) {
  const double c = binCnt.mine();
  out.load(c == 0.0 ? 0.0 : leaf_loglik(binWt.mine(), binSum.mine(), sigma2, prec));
}
// value of one leaf from its statistics (weight w, weighted sum s) and the two uniforms of its draw (tree_hd.hpp leaves_draw)
// (in two halves: the standard normal deviate depends on the two uniforms alone — the persistent sweep has it ready before the statistics arrive)
__device__ __forceinline__ double leaf_deviate(double u1, double u2) {
  const double BIG = 134217728.0;
  return r_qnorm(((double)(int)(BIG * u1) + u2) / BIG);
}
// (... and of the posterior only the mean depends on the weighted sum: postPrec = w / sigma2, den = prec + postPrec, sd = 1 / sqrt(den))
__device__ __forceinline__ double leaf_value_parts(double postPrec, double den, double sd, double w, double s, double z) {
  const double mean = postPrec * (s / w) / den;
  return mean + sd * z;
}
__device__ __forceinline__ double leaf_value_z(double w, double s, double z, double sigma2, double prec) {
  const double postPrec = w / sigma2;
  const double den = prec + postPrec;
  const double sd = 1.0 / sqrt(den);
  return leaf_value_parts(postPrec, den, sd, w, s, z);
}
__device__ __forceinline__ double leaf_value(double w, double s, double u1, double u2, double sigma2, double prec) {
  return leaf_value_z(w, s, leaf_deviate(u1, u2), sigma2, prec);
}
__device__ __forceinline__ void leaves_draw(const WaveArrD& lc, const WaveArrD& ls, const WaveArrD& lw, const WaveArrD& u1, const WaveArrD& u2, int nl,
                                            double sigma2, double prec, WaveArrD& out) {
  const double c = lc.mine();
  double v = 0.0;
  if ((int)(threadIdx.x & 63) < nl && c != 0.0) v = leaf_value(lw.mine(), ls.mine(), u1.mine(), u2.mine(), sigma2, prec);
  out.load(v);
}

__device__ __forceinline__ void wave_tree_load(WaveTree& t, const int16_t* var, const uint16_t* cut, const int16_t* left, const int16_t* right,
                                               const int16_t* parent, int count, int nc, int lane) {
  const bool in = lane < count;
  t.var.r = in ? (int)var[lane] : (int)NODE_FREE; t.cut.r = in ? (int)cut[lane] : 0; t.left.r = in ? (int)left[lane] : -1;
  t.right.r = in ? (int)right[lane] : -1; t.parent.r = in ? (int)parent[lane] : -1; t.na.r = 0; t.dep.r = 0; t.nc = nc < 64 ? nc : 64;
}
__device__ __forceinline__ void wave_tree_store(const WaveTree& t, int16_t* var, uint16_t* cut, int16_t* left, int16_t* right, int16_t* parent,
                                                int count, int lane) {
  if (lane < count) { var[lane] = (int16_t)t.var.r; cut[lane] = (uint16_t)t.cut.r; left[lane] = (int16_t)t.left.r; right[lane] = (int16_t)t.right.r;
                      parent[lane] = (int16_t)t.parent.r; }
}

// slow path for trees with more than 64 node slots in use: the sequential code straight on the global arrays
__device__ __attribute__((noinline)) void control_global_path(BartArrays a, int t, int next, double* scratch) {
  a.model.scratch = scratch;
  if (t >= 0) control_step(a, t, next); else propose_step(a, next);
}

static size_t control_lds_bytes(int P, int logIntLen) { return ((size_t)P * 4 + 15) / 16 * 16 + (size_t)logIntLen * 8 + 64; }

typedef TreeCacheT<WaveArr<int16_t>> WaveCache;

// Generator as the control wave sees it: the state array stays in LDS, the position and the 64-word window
// around it live in registers (lane l = mt[wbase + l]), so a draw is a v_readlane plus the tempering.
struct WaveRng {
  MTState* st; int mti; int wbase; uint32_t win; int count; int regen;   // count: draws since open() / since last zeroed; regen: the block was regenerated
  __device__ __forceinline__ void open(MTState* s) { st = s; mti = S4B_UNI((int)s->mti); wbase = -64; win = 0u; count = 0; regen = 0; }
  __device__ __forceinline__ void close() { st->mti = mti; }
};
__device__ __forceinline__ uint32_t mt_next(WaveRng* r) {
  int k = r->mti;
  if (k >= 624) { mt_regenerate_wave(r->st); k = 0; r->wbase = -64; r->regen = 1; }
  const int wb = k & ~63;
  if (wb != r->wbase) {
    const int idx = wb + (int)(threadIdx.x & 63);
    r->win = idx < 624 ? r->st->mt[idx] : 0u;
    r->wbase = wb;
  }
  uint32_t y = (uint32_t)__builtin_amdgcn_readlane((int)r->win, k & 63);
  r->mti = k + 1; ++r->count;
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

// lane-parallel leaf statistics + draws of decide(): lane i owns leaf i of the cached DFS list.  The uniforms are the generator's
// next words in leaf order (two per leaf that holds observations): lane i reads the pair at its rank among those leaves.
__device__ __forceinline__ int wave_gather(int idx, int v) { return __builtin_amdgcn_ds_bpermute(idx << 2, v); }
__device__ __forceinline__ double wave_gather(int idx, const WaveArrD& a) { return __hiloint2double(wave_gather(idx, a.hi), wave_gather(idx, a.lo)); }
__device__ __forceinline__ double mt_word_to_unif(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  const double half_ulp = 0.5 * 2.328306437080797e-10;
  const double v = (double)y * 2.3283064365386963e-10;
  return v <= 0.0 ? half_ulp : (1.0 - v <= 0.0 ? 1.0 - half_ulp : v);
}
__device__ __forceinline__ void leaf_stats_draws(const WaveTables& tb, const WaveCache& ca, const WaveArrD& binCnt, const WaveArrD& binSum, const WaveArrD& binWt,
                                                 bool acc, bool deathAcc, int nd, double cDeath, double sDeath, double wDeath, DecideWork<WaveArrD>& wk, WaveRng* rng) {
  const int lane = (int)(threadIdx.x & 63);
  const int nl = ca.nl;
  const bool in = lane < nl;
  const int n = in ? ca.leaf.r : 0;
  const int bB = (int)(int16_t)wave_gather(n, tb.binB.r), bA = (int)(int16_t)wave_gather(n, tb.binA.r);
  const int b = ((acc && !deathAcc && bB >= 0) ? bB : bA) & 63;
  double lc = wave_gather(b, binCnt), ls = wave_gather(b, binSum), lw = wave_gather(b, binWt);
  const bool dn = deathAcc && n == nd;
  lc = dn ? cDeath : lc; ls = dn ? sDeath : ls; lw = dn ? wDeath : lw;
  const bool ne = in && lc != 0.0;
  const unsigned long long mask = __ballot(ne);
  const int cnt = __popcll(mask);
  if (rng->mti + 2 * cnt > 624) {   // the block of words runs out among these draws (about one step in a hundred): one draw at a time
    leaf_stats_draws<WaveTables, WaveCache, WaveArrD, WaveArrD, WaveRng>(tb, ca, binCnt, binSum, binWt, acc, deathAcc, nd, cDeath, sDeath, wDeath, wk, rng);
    return;
  }
  const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
  const int k = ne ? rng->mti + 2 * rank : 0;
  const uint32_t y1 = rng->st->mt[k], y2 = rng->st->mt[k + 1];
  if (in) { wk.lc.load(lc); wk.ls.load(ls); wk.lw.load(lw); }
  if (ne) { wk.u1.load(mt_word_to_unif(y1)); wk.u2.load(mt_word_to_unif(y2)); }
  rng->mti += 2 * cnt; rng->count += 2 * cnt; rng->wbase = -64;
}

// Model view of the control wave: the prior tables sit in registers (lane d / lane k), LDS only beyond 64 / 128
// (SPL: cgm(split.probs) — the weighted predictor choice — on the wave-register path: only the persistent sweep's kernels k_sweep_sp /
// k_sweep_few_sp are compiled with it; everywhere else the wave-register code is compiled without and such samplers take the
// pointer-storage control code)
template <bool SPL>
struct WaveModelT : ModelView {
  WaveArrD pg, lpg, l1pg, li0, li1; int nc0, nc1;
};
// (with the weights: spTab, on chip — [0, 128) the running sum of the weights through predictor v over the predictors that have cuts, in predictor order;
// [128, 256) log(weight of predictor v); [256] the sum over all of them, [257] its logarithm: what the sums below come to at a node where every predictor
// is still available, which is nearly every node — k_sweep_sp's prologue fills it)
template <>
struct WaveModelT<true> : ModelView {
  WaveArrD pg, lpg, l1pg, li0, li1; int nc0, nc1; const double* spTab;
};
constexpr int SP_TAB = 258;
typedef WaveModelT<false> WaveModel;
template <bool SPL> __device__ __forceinline__ double mv_pg_depth(const WaveModelT<SPL>& m, int d) { return d < 64 ? m.pg.get(d) : S4B_UNI(m.pgDepth[d]); }
template <bool SPL> __device__ __forceinline__ double mv_log_pg(const WaveModelT<SPL>& m, int d) { return d < 64 ? m.lpg.get(d) : S4B_UNI(m.logPg[d]); }
template <bool SPL> __device__ __forceinline__ double mv_log1m_pg(const WaveModelT<SPL>& m, int d) { return d < 64 ? m.l1pg.get(d) : S4B_UNI(m.log1mPg[d]); }
template <bool SPL> __device__ __forceinline__ double mv_log_int(const WaveModelT<SPL>& m, int k) {
  return k < 64 ? m.li0.get(k) : (k < 128 ? m.li1.get(k - 64) : S4B_UNI(m.logInt[k]));
}
template <bool SPL> __device__ __forceinline__ const double* mv_split_probs(const WaveModelT<SPL>& m) { return SPL ? m.splitProbs : nullptr; }
template <bool SPL> __device__ __forceinline__ int mv_num_cuts(const WaveModelT<SPL>& m, int v) {
  return v < 64 ? __builtin_amdgcn_readlane(m.nc0, v) : (v < 128 ? __builtin_amdgcn_readlane(m.nc1, v - 64) : S4B_UNI(m.numCuts[v]));
}

// cgm(split.probs) on the wave-register path.  Which predictors still have a free cut at node n: lane l answers for predictor 64 chunk + l — the walk up
// the ancestors is uniform (register reads), every lane narrows the interval of ITS predictor (tv_interval for 64 predictors at once).  The sums over the
// available predictors are then formed in increasing predictor order, one add per available predictor, exactly as the sequential tv_avail_prob_sum and
// tv_draw_var of tree_hd.hpp form them: the same doubles, the same draw.  (The sequential versions walk the ancestors once per predictor through register
// reads: 59 us per tree update at P = 49 against 8 without the weights; these: see DESIGN.md 8.)
__device__ __forceinline__ unsigned long long wave_avail_mask(const WaveTree& t, const WaveModelT<true>& m, int n, int chunk) {
  const int v = chunk * 64 + (int)(threadIdx.x & 63);
  const int ncv = chunk == 0 ? m.nc0 : (chunk == 1 ? m.nc1 : (v < m.P ? m.numCuts[v] : 0));
  int lo = 0, hi = ncv - 1;
  int child = n;
  for (int a = t.parent.get(n); a >= 0; child = a, a = t.parent.get(a)) {
    const int av = t.var.get(a), s = (int)t.cut.get(a);
    const bool isLeft = child == t.left.get(a);
    const bool hit = av == v;
    hi = (hit && isLeft && s - 1 < hi) ? s - 1 : hi;
    lo = (hit && !isLeft && s + 1 > lo) ? s + 1 : lo;
  }
  return __ballot(v < m.P && ncv > 0 && lo <= hi);
}
__device__ __forceinline__ double wave_lane_double(double x, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ double tv_avail_prob_sum(const WaveTree& t, const WaveModelT<true>& m, int n, const double* sp) {
  if ((int)t.na.get(n) == m.Pvalid) return S4B_UNI(m.spTab[256]);      // nothing exhausted on the way down to n: the sum over all, formed once
  double tot = 0.0;
  for (int c = 0; c * 64 < m.P; ++c) {
    unsigned long long mask = wave_avail_mask(t, m, n, c);
    const int v = c * 64 + (int)(threadIdx.x & 63);
    const double mine = v < m.P ? sp[v] : 0.0;
    while (mask) {
      const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)mask) - 1);
      mask &= mask - 1ull;
      tot += wave_lane_double(mine, b);
    }
  }
  return tot;
}
__device__ __forceinline__ double tv_log_var_prob(const WaveTree& t, const WaveModelT<true>& m, int n, int v, int na) {
  const double* sp = m.splitProbs;
  if (!sp) return -mv_log_int(m, na);
  const double lv = v < 128 ? S4B_UNI(m.spTab[128 + v]) : log(S4B_UNI(sp[v]));
  if (na == m.Pvalid) return lv - S4B_UNI(m.spTab[257]);
  return lv - log(tv_avail_prob_sum(t, m, n, sp));
}
// (k_sweep_sp's prologue, one wave: the table of the weights)
__device__ __forceinline__ void wave_fill_sp_tab(const BartArrays& a, double* spTab, int lane) {
  const double* sp = a.model.splitProbs;
  double run = 0.0;
  for (int c = 0; c * 64 < a.P; ++c) {
    const int v = c * 64 + lane;
    const bool ok = v < a.P && a.numCuts[v] > 0;
    const double mine = v < a.P ? sp[v] : 1.0;
    unsigned long long mask = __ballot(ok);
    double pre = run;
    while (mask) {
      const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)mask) - 1);
      mask &= mask - 1ull;
      run += wave_lane_double(mine, b);
      pre = lane >= b ? run : pre;
    }
    if (c < 2) { spTab[v] = pre; spTab[128 + v] = log(mine); }
  }
  if (lane == 0) { spTab[256] = run; spTab[257] = log(run); }
}
template <class RNG>
__device__ __forceinline__ int tv_draw_var(const WaveTree& t, const WaveModelT<true>& m, int n, RNG* rng) {
  const double* sp = m.splitProbs;
  if (!sp) { const int good = tv_num_avail(t, m, n); const int idx = r_unif_int(rng, 0, good); return tv_nth_avail_var(t, m, n, idx); }
  const double u = r_unif(rng) * tv_avail_prob_sum(t, m, n, sp);
  if ((int)t.na.get(n) == m.Pvalid && m.P <= 128) {
    // every predictor with cuts is available: the running sums are the table's — the first one beyond u, else the last predictor with cuts
    const int l = (int)(threadIdx.x & 63);
    const bool ok0 = l < m.P && m.nc0 > 0, ok1 = 64 + l < m.P && m.nc1 > 0;
    const unsigned long long v0 = __ballot(ok0), v1 = __ballot(ok1);
    const unsigned long long h0 = __ballot(ok0 && m.spTab[l] > u), h1 = __ballot(ok1 && m.spTab[64 + l] > u);
    if (h0) return __builtin_amdgcn_readfirstlane(__ffsll((long long)h0) - 1);
    if (h1) return __builtin_amdgcn_readfirstlane(64 + __ffsll((long long)h1) - 1);
    if (v1) return __builtin_amdgcn_readfirstlane(127 - __clzll((long long)v1));
    return v0 ? __builtin_amdgcn_readfirstlane(63 - __clzll((long long)v0)) : -1;
  }
  double run = 0.0; int last = -1;
  for (int c = 0; c * 64 < m.P; ++c) {
    unsigned long long mask = wave_avail_mask(t, m, n, c);
    const int v = c * 64 + (int)(threadIdx.x & 63);
    const double mine = v < m.P ? sp[v] : 0.0;
    while (mask) {
      const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)mask) - 1);
      mask &= mask - 1ull;
      run += wave_lane_double(mine, b); last = c * 64 + b;
      if (__builtin_amdgcn_readfirstlane((int)(run > u))) return last;
    }
  }
  return last;
}

// Workgroup of 8 waves with fixed roles, tied together by two LDS hand-shakes (no workgroup barrier after start-up):
//   wave 0        decide(t): waits for the bin totals, accept/reject, leaf draws, writes tree t, names the winner
//   waves 1, 2    candidates (waves are dealt round-robin to the 4 SIMDs: the three long-running roles sit on three SIMDs): draw the proposal of tree `next` from the generator position decide(t) will leave behind —
//                 that position is known up to one bit before the statistics arrive: decide consumes one uniform for the
//                 accept test plus two per leaf of the tree it ends with, i.e. d + 2 nl (reject) or d + 2 nl' (accept).
//                 Each candidate advances a private copy of the generator by its hypothesis and runs propose() while
//                 wave 0 is still waiting / deciding; the one whose hypothesis matches the draws actually consumed
//                 publishes its tables and generator state.  A leaf without observations (no draw) breaks both
//                 hypotheses: then wave 0 proposes itself, as it does at the start of a sweep.
//   waves 3-7     reducers: per-workgroup partials -> bin totals (fixed order)
constexpr int CBLOCK = 512;
constexpr int C_NRED = 5;
struct ControlShared {
  MTState rng[3];                        // slot 0: wave 0, slots 1, 2: candidates
  double scratch[3][S4B_MAX_DEPTH];
  double red[3][C_NRED][64];             // [sum | count | weight][reducer][bin]
  Proposal prT, prN[3];
  int arrived, verdict;
  long long tPost, tStart0;
};
#ifdef S4B_TUNING
__device__ int g_dbgSpin[4];      // (development) who waits for more than ~30 ms: source line, flag value | target << 16, block, thread
#endif
__device__ __forceinline__ void spin_until_at(int* flag, int target, int32_t* errFlag, int line) {
  int guard = 0;
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    __builtin_amdgcn_s_sleep(1);
#ifdef S4B_TUNING
    if (guard == (1 << 18) && (threadIdx.x & 63) == 0) { g_dbgSpin[0] = line; g_dbgSpin[1] = *flag | (target << 16); g_dbgSpin[2] = (int)blockIdx.x; g_dbgSpin[3] = (int)threadIdx.x; }
#endif
    if (++guard > (1 << 24)) { *errFlag |= S4B_ERR_INTERNAL; break; }   // never hang the device on a logic error
  }
  (void)line;
}
#define spin_until(flag, target, errFlag) spin_until_at(flag, target, errFlag, __LINE__)
__device__ __forceinline__ void rng_advance(WaveRng* r, int k) {
  int total = r->mti + k;
  while (total > 624) { mt_regenerate_wave(r->st); total -= 624; r->regen = 1; }
  r->mti = total; r->wbase = -64;
}

// the step's scalars in one gathered load: lane l < 20 holds dword l of the pending Proposal of tree tt, lanes 32.. / 40..
// the int32 per-tree scalars (row TI_x) of trees tt / tn; a second 8-byte gather brings the two log priors and sigma
static_assert(sizeof(Proposal) == 80, "gather layout");
struct StepScalars {
  int w; int dlo, dhi;
  __device__ __forceinline__ int i32(int l) const { return __builtin_amdgcn_readlane(w, l); }
  __device__ __forceinline__ double f64w(int l) const { return __hiloint2double(__builtin_amdgcn_readlane(w, l + 1), __builtin_amdgcn_readlane(w, l)); }
  __device__ __forceinline__ double f64(int l) const { return __hiloint2double(__builtin_amdgcn_readlane(dhi, l), __builtin_amdgcn_readlane(dlo, l)); }
  __device__ __forceinline__ void proposal(Proposal& p) const {
    p.type = i32(0); p.status = i32(1); p.node = i32(2); p.var = i32(3); p.split = i32(4); p.nbA = i32(5); p.nbB = i32(6); p.hwm = i32(7);
    p.newLeft = i32(8); p.newRight = i32(9); p.pad0 = i32(10); p.pad1 = i32(11);
    p.priorRatio = f64w(12); p.transRatio = f64w(14); p.XLogPi = f64w(16); p.YLogPi = f64w(18);
  }
};
__device__ __forceinline__ void step_scalars_load(StepScalars& g, const BartArrays& a, const Proposal* prop, int tt, int tn, int lane, bool withDoubles) {
  const int32_t* tS = a.treeI32; const size_t tT = (size_t)a.T;
  const int32_t* ap = (const int32_t*)prop + (lane < 20 ? lane : 0);
  const int f = lane & 7;
  if (lane >= 32 && lane < 48 && f < TI_COUNT) ap = tS + (size_t)f * tT + (lane < 40 ? tt : tn);
  g.w = (lane < 20 || (lane >= 32 && lane < 48 && f < TI_COUNT)) ? *ap : 0;
  g.dlo = 0; g.dhi = 0;
  if (withDoubles) {
    const double* dp = lane == 0 ? a.clogpi + tt : (lane == 1 ? a.clogpi + tn : &a.scale->sigma);
    const double d = lane < 3 ? *dp : 0.0;
    g.dlo = __double2loint(d); g.dhi = __double2hiint(d);
  }
}
// proposal record -> global, one dword per lane
__device__ __forceinline__ void proposal_store(const Proposal& p, Proposal* dst, int lane) {
  int w = 0;
  const int v[20] = {p.type, p.status, p.node, p.var, p.split, p.nbA, p.nbB, p.hwm, p.newLeft, p.newRight, p.pad0, p.pad1,
                     __double2loint(p.priorRatio), __double2hiint(p.priorRatio), __double2loint(p.transRatio), __double2hiint(p.transRatio),
                     __double2loint(p.XLogPi), __double2hiint(p.XLogPi), __double2loint(p.YLogPi), __double2hiint(p.YLogPi)};
#pragma unroll
  for (int i = 0; i < 20; ++i) w = lane == i ? v[i] : w;
  if (lane < 20) ((int*)dst)[lane] = w;
}

#ifndef S4B_SWEEP_TU   // (dev_sweep.hip compiles this file up to k_sweep only: see the end of the device section)
__global__ __launch_bounds__(CBLOCK) void k_control(BartArrays a, int t, int next) {
  __shared__ ControlShared S;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: the role branches are scalar branches
  S4B_TICK(tk0);
  if (threadIdx.x == 0) { S.arrived = 0; S.verdict = 0; }
  __syncthreads();
  const bool doDecide = t >= 0, doPropose = next >= 0;
  const int tt = doDecide ? t : 0, tn = doPropose ? next : 0;
  const StepScratch& cT = a.sc[tt & 1];
  const StepScratch& cN = a.sc[tn & 1];
  const int nc = a.nc;
  const bool isDecider = wv == 0, isCand = wv == 1 || wv == 2;

  // ================================================================ reducers (waves 3-7)
  if (!isDecider && !isCand) {
    if (!doDecide) return;
    const int redIdx = wv - 3;
    const int ridx = redIdx * 64 + lane;
    constexpr int RT = C_NRED * 64;
    // one hop: first 8 bins (sum, count) of two partials per lane + the scalars that select the path
    double pv[16], qv[16];
    {
      const int b0 = ridx, b1 = ridx + RT;
      const bool in0 = b0 < a.grid, in1 = b1 < a.grid;
      const size_t c0 = in0 ? b0 : 0, c1 = in1 ? b1 : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pv[j] = a.partSum[(size_t)j * a.grid + c0]; pv[8 + j] = a.partCnt[(size_t)j * a.grid + c0];
        qv[j] = a.partSum[(size_t)j * a.grid + c1]; qv[8 + j] = a.partCnt[(size_t)j * a.grid + c1];
      }
      StepScalars g; step_scalars_load(g, a, cT.prop, tt, tn, lane, false);
#pragma unroll
      for (int j = 0; j < 16; ++j) pv[j] = (in0 ? pv[j] : 0.0) + (in1 ? qv[j] : 0.0);
      const int prHwm = g.i32(7), nbAll = g.i32(5) + g.i32(6), hwmT = g.i32(32 + TI_HWM), hwmN = g.i32(40 + TI_HWM);
      int need = prHwm > hwmT ? prHwm : hwmT;
      if (doPropose && hwmN + 2 > need) need = hwmN + 2;
      const bool wavePath = need <= 64 && nbAll <= 64 && need <= nc + 2 && a.model.splitProbs == nullptr;
      if (wavePath) {
        for (int b = ridx + 2 * RT; b < a.grid; b += RT) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { pv[j] += a.partSum[(size_t)j * a.grid + b]; pv[8 + j] += a.partCnt[(size_t)j * a.grid + b]; }
        }
        wave_sum_bins<16>(pv, lane);
        if ((lane & 3) == 0) { const int k = wave_bin_of_lane<16>(lane); S.red[k >> 3][redIdx][k & 7] = pv[0]; }
        if (a.wts) {   // weight totals of the first 8 bins
          double wv8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) wv8[j] = 0.0;
          for (int b = ridx; b < a.grid; b += RT) {
#pragma unroll
            for (int j = 0; j < 8; ++j) wv8[j] += a.partWt[(size_t)j * a.grid + b];
          }
          wave_sum_bins<8>(wv8, lane);
          if ((lane & 7) == 0) S.red[2][redIdx][wave_bin_of_lane<8>(lane)] = wv8[0];
        }
        for (int k = 8; k < nbAll; ++k) {   // rare: more than 8 bins
          double sm = 0.0, c = 0.0, wt = 0.0;
          for (int b = ridx; b < a.grid; b += RT) { sm += a.partSum[(size_t)k * a.grid + b]; c += a.partCnt[(size_t)k * a.grid + b]; if (a.wts) wt += a.partWt[(size_t)k * a.grid + b]; }
          sm = wave_sum(sm); c = wave_sum(c); wt = wave_sum(wt);
          if (lane == 0) { S.red[0][redIdx][k] = sm; S.red[1][redIdx][k] = c; S.red[2][redIdx][k] = wt; }
        }
      } else {   // large tree: totals of every bin to global memory for the sequential path
        for (int k = redIdx; k < nbAll; k += C_NRED) {
          double sm = 0.0, c = 0.0, wt = 0.0;
          for (int b = lane; b < a.grid; b += 64) { sm += a.partSum[(size_t)k * a.grid + b]; c += a.partCnt[(size_t)k * a.grid + b]; if (a.wts) wt += a.partWt[(size_t)k * a.grid + b]; }
          sm = wave_sum(sm); c = wave_sum(c); wt = wave_sum(wt);
          if (lane == 0) { a.binSum[k] = sm; a.binCnt[k] = c; if (a.wts) a.binWt[k] = wt; }
        }
        __threadfence();
      }
    }
    if (lane == 0) __hip_atomic_fetch_add(&S.arrived, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
  }

  // ================================================================ decider (wave 0) and candidates (waves 1, 2)
  const int candIdx = wv - 1;
  const int slot = isCand ? 1 + candIdx : 0;
  const size_t oT = (size_t)tt * nc, oN = (size_t)tn * nc;
  const bool laneIn = lane < nc;
  const int li = laneIn ? lane : 0;
  // one base address per slab (dev_common.hpp): field f of tree x = tI + f * ts + x * nc
  int16_t* const tI = a.treeI16; const size_t ts = (size_t)a.T * nc; int32_t* const tS = a.treeI32; const size_t tT = (size_t)a.T;
  int16_t* const sT = cT.slab; int16_t* const sN = cN.slab;
  // ---- one hop: everything this wave needs
  WaveModel m;
  static_cast<ModelView&>(m) = a.model;           // overflow pointers (depth >= 64, tables beyond 128 entries) stay global
  m.scratch = S.scratch[slot];
  constexpr int MTW = (int)(sizeof(MTState) / 4), MTJ = (MTW + 63) / 64;
  uint32_t mtw[MTJ];
#pragma unroll
  for (int j = 0; j < MTJ; ++j) { const int i = lane + j * 64; mtw[j] = ((const uint32_t*)a.rng)[i < MTW ? i : 0]; }   // private generator copy
  m.pg.load(a.model.pgDepth[lane]); m.lpg.load(a.model.logPg[lane]); m.l1pg.load(a.model.log1mPg[lane]);
  m.li0.load(a.model.logInt[lane < m.logIntLen ? lane : 0]); m.li1.load(a.model.logInt[64 + lane < m.logIntLen ? 64 + lane : 0]);
  m.nc0 = a.numCuts[lane < a.P ? lane : 0]; m.nc1 = a.numCuts[64 + lane < a.P ? 64 + lane : 0];
  // tree `next`: the candidates fetch it in this hop; the decider only if it ends up proposing itself (sweep start, or no
  // candidate matched), one extra hop then
  WaveTree curN; WaveCache caN; WaveTables tbN;
  curN.nc = nc < 64 ? nc : 64;
  auto loadNext = [&]() {
    curN.var.r = laneIn ? (int)tI[TF_VAR * ts + oN + li] : (int)NODE_FREE; curN.cut.r = ((uint16_t*)tI)[TF_CUT * ts + oN + li]; curN.left.r = tI[TF_LEFT * ts + oN + li];
    curN.right.r = tI[TF_RIGHT * ts + oN + li]; curN.parent.r = tI[TF_PARENT * ts + oN + li]; curN.na.r = tI[TF_NA * ts + oN + li];
    curN.dep.r = tI[TF_DEP * ts + oN + li];
    caN.leaf.r = tI[TF_LEAF * ts + oN + li]; caN.pre.r = tI[TF_PRE * ts + oN + li]; caN.post.r = tI[TF_POST * ts + oN + li];
  };
  if (isCand) loadNext();
  WaveTree curT; WaveTables tbT; WaveCache caT; WaveArrD mu, muOld; WaveArr<int32_t> cnt;
  if (isDecider) {
    curT.var.r = tI[TF_VAR * ts + oT + li]; curT.cut.r = ((uint16_t*)tI)[TF_CUT * ts + oT + li]; curT.left.r = tI[TF_LEFT * ts + oT + li];
    curT.right.r = tI[TF_RIGHT * ts + oT + li]; curT.parent.r = tI[TF_PARENT * ts + oT + li]; curT.na.r = tI[TF_NA * ts + oT + li];
    curT.dep.r = tI[TF_DEP * ts + oT + li]; curT.nc = curN.nc;
    tbT.prop.var.r = sT[SF_VAR * nc + li]; tbT.prop.cut.r = ((uint16_t*)sT)[SF_CUT * nc + li]; tbT.prop.left.r = sT[SF_LEFT * nc + li];
    tbT.prop.right.r = sT[SF_RIGHT * nc + li]; tbT.prop.parent.r = sT[SF_PARENT * nc + li]; tbT.prop.na.r = sT[SF_NA * nc + li];
    tbT.prop.dep.r = sT[SF_DEP * nc + li]; tbT.prop.nc = curN.nc;
    tbT.binA.r = sT[SF_BINA * nc + li]; tbT.binB.r = sT[SF_BINB * nc + li]; tbT.insub.r = cT.insub[li]; tbT.list.r = 0;
    caT.leaf.r = tI[TF_LEAF * ts + oT + li]; caT.pre.r = tI[TF_PRE * ts + oT + li]; caT.post.r = tI[TF_POST * ts + oT + li];
    mu.load(a.mu[oT + li]); muOld.load(0.0); cnt.r = a.cnt[oT + li];
  }
  StepScalars g; step_scalars_load(g, a, cT.prop, tt, tn, lane, true);
  // ---- first uses
  m.li0.load(lane < m.logIntLen ? m.li0.mine() : 0.0); m.li1.load(64 + lane < m.logIntLen ? m.li1.mine() : 0.0);
  m.nc0 = lane < a.P ? m.nc0 : 0; m.nc1 = 64 + lane < a.P ? m.nc1 : 0;
#pragma unroll
  for (int j = 0; j < MTJ; ++j) { const int i = lane + j * 64; if (i < MTW) ((uint32_t*)&S.rng[slot])[i] = mtw[j]; }
  Proposal prT; g.proposal(prT);
  const int hwmT = g.i32(32 + TI_HWM), hwmN = g.i32(40 + TI_HWM);
  caN.nl = g.i32(40 + TI_NL); caN.ni = g.i32(40 + TI_NI); caN.g = g.i32(40 + TI_G); caN.gn = g.i32(40 + TI_GN); caN.valid = g.i32(40 + TI_VALID);
  caN.logPi = g.f64(1);
  int need = 0, nb = 0;
  if (doDecide) { need = prT.hwm > hwmT ? prT.hwm : hwmT; nb = prT.nbA + prT.nbB; }
  if (doPropose && hwmN + 2 > need) need = hwmN + 2;
  const bool wavePath = need <= 64 && nb <= 64 && need <= nc + 2 && a.model.splitProbs == nullptr;   // (cgm(split.probs): the pointer-storage control code)
  // generator positions decide(t) can end at; candidate 1 only exists when accepting changes the number of leaves
  const int drawsAccept = (doDecide && prT.status == 1) ? 1 : 0;
  const int nlNow = prT.nbA;
  const int nlAcc = prT.type == MOVE_BIRTH ? nlNow + 1 : (prT.type == MOVE_DEATH ? nlNow - 1 : nlNow);
  const int hyp0 = drawsAccept + 2 * nlNow, hyp1 = drawsAccept + 2 * nlAcc;
  const bool spec = doDecide && doPropose && next != t && wavePath;
  const bool cand1Exists = spec && drawsAccept == 1 && hyp1 != hyp0;
  S4B_TICK(tkA);
  if (!wavePath) {   // large tree: sequential code straight on the global arrays (one lane), no candidates
    if (isCand) return;
    if (doDecide) spin_until(&S.arrived, C_NRED, a.errFlag);
    if (threadIdx.x == 0) control_global_path(a, t, next, S.scratch[0]);
    return;
  }
  if (isCand && (!spec || (candIdx == 1 && !cand1Exists))) return;

  WaveRng rng; rng.open(&S.rng[slot]);
  S4B_PTICK(tc1);
  S4B_TICK(tkB);
  S4B_TICK(tk2);
  bool proposer = false;
  if (isCand) {
    rng_advance(&rng, candIdx == 0 ? hyp0 : hyp1);
    proposer = true;
    rng.count = 0;
  } else {
    int winner = 0;
    if (doDecide) {
      caT.nl = g.i32(32 + TI_NL); caT.ni = g.i32(32 + TI_NI); caT.g = g.i32(32 + TI_G); caT.gn = g.i32(32 + TI_GN); caT.valid = g.i32(32 + TI_VALID);
      caT.logPi = g.f64(0);
      const double sigma = g.f64(2);
      spin_until(&S.arrived, C_NRED, a.errFlag);
#ifdef S4B_CONTROL_TIMING
      tkB = wall_clock64();
#endif
      WaveArrD binSum, binCnt;
      {
        double sSum = 0.0, sCnt = 0.0;
        if (lane < nb) {
          sSum = S.red[0][0][lane]; sCnt = S.red[1][0][lane];
#pragma unroll
          for (int r = 1; r < C_NRED; ++r) { sSum += S.red[0][r][lane]; sCnt += S.red[1][r][lane]; }
        }
        binSum.load(sSum); binCnt.load(sCnt);
      }
      WaveArrD binWt = binCnt;   // without weights the precision comes from the counts
      if (a.wts) {
        double sWt = 0.0;
        if (lane < nb) {
          sWt = S.red[2][0][lane];
#pragma unroll
          for (int r = 1; r < C_NRED; ++r) sWt += S.red[2][r][lane];
        }
        binWt.load(sWt);
      }
      DecideWork<WaveArrD> wk;
      wk.ll.load(0.0); wk.lc.load(0.0); wk.ls.load(0.0); wk.u1.load(0.5); wk.u2.load(0.5); wk.val.load(0.0); wk.lw.load(0.0);
      StepRecord rec; int32_t accepted = 0;
      // as soon as the step's last random number is drawn: which candidate (if any) started from the position the
      // generator is at now?  (posted before the leaf arithmetic so that the winner publishes meanwhile)
      auto post = [&]() {
        const int used = rng.count;
        if (spec) winner = used == hyp0 ? 1 : ((cand1Exists && used == hyp1) ? 2 : 0);
        if (lane == 0) __hip_atomic_store(&S.verdict, winner ? winner : 3, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef S4B_CONTROL_TIMING
        if (lane == 0) { S.tPost = wall_clock64(); S.tStart0 = tk0; }
#endif
      };
      const int hwmNew = decide(curT, mu, cnt, muOld, hwmT, m, sigma, &rng, &prT, tbT, binCnt, binSum, binWt, wk, &accepted, &rec, caT, post);
#ifdef S4B_CONTROL_TIMING
      tk2 = wall_clock64();
#endif
      const int cntOut = prT.hwm > hwmNew ? prT.hwm : hwmNew;
      wave_tree_store(curT, tI + TF_VAR * ts + oT, (uint16_t*)(tI + TF_CUT * ts + oT), tI + TF_LEFT * ts + oT, tI + TF_RIGHT * ts + oT,
                      tI + TF_PARENT * ts + oT, cntOut, lane);
      if (lane < cntOut) { a.mu[oT + lane] = mu.mine(); a.cnt[oT + lane] = cnt.r; cT.muOld[lane] = muOld.mine(); cT.insub[lane] = (uint8_t)tbT.insub.r; }
      if (accepted) {   // the structure cache moved with the tree: keep the global copy current
        if (lane < cntOut) { tI[TF_NA * ts + oT + lane] = (int16_t)curT.na.r; tI[TF_DEP * ts + oT + lane] = (int16_t)curT.dep.r; }
        if (laneIn) { tI[TF_LEAF * ts + oT + lane] = (int16_t)caT.leaf.r; tI[TF_PRE * ts + oT + lane] = (int16_t)caT.pre.r; tI[TF_POST * ts + oT + lane] = (int16_t)caT.post.r; }
        if (lane == 0) { tS[TI_NL * tT + t] = caT.nl; tS[TI_NI * tT + t] = caT.ni; tS[TI_G * tT + t] = caT.g; tS[TI_GN * tT + t] = caT.gn; a.clogpi[t] = caT.logPi; }
      }
      if (lane == 0) {
        tS[TI_HWM * tT + t] = hwmNew; *cT.accepted = accepted;
        if (a.traceOn) push_trace(a, rec);
      }
    }
    proposer = doPropose && winner == 0;
    if (!doPropose) {   // last tree of the sweep: only the generator goes back
      if (rng.regen) {   // the 624-word block only changes when it was regenerated (every ~25 tree updates); else just the position
        rng.close(); S.rng[0].pad = 0;
#pragma unroll
        for (int j = 0; j < MTJ; ++j) { const int i = lane + j * 64; mtw[j] = ((const uint32_t*)&S.rng[0])[i < MTW ? i : 0]; }
#pragma unroll
        for (int j = 0; j < MTJ; ++j) { const int i = lane + j * 64; if (i < MTW) ((uint32_t*)a.rng)[i] = mtw[j]; }
      } else if (lane == 0) a.rng->mti = rng.mti;
    }
  }
  if (!proposer) {
#ifdef S4B_CONTROL_TIMING
    if (isDecider && lane == 0 && doDecide && doPropose) {
      S4B_TICK(tk4);
      atomicAdd((unsigned long long*)&g_dbg[0], (unsigned long long)(tkB - tk0)); atomicAdd((unsigned long long*)&g_dbg[1], (unsigned long long)(tk2 - tkB));
      atomicAdd((unsigned long long*)&g_dbg[3], (unsigned long long)(tk4 - tk2)); atomicAdd((unsigned long long*)&g_dbg[4], 1ull);
      atomicAdd((unsigned long long*)&g_dbg[5], (unsigned long long)(tkA - tk0)); atomicAdd((unsigned long long*)&g_dbg[6], (unsigned long long)(tkB - tkA));
      atomicAdd((unsigned long long*)&g_dbg[7], 1ull);
    }
#endif
    return;
  }

  // ================================================================ proposal of tree `next` (one call site for all roles)
  if (isDecider) loadNext();
  bool rebuilt = false;
  S4B_PTICK(tc2);
  if (!caN.valid) { tv_rebuild_cache(curN, m, caN); rebuilt = true; }
  S4B_PTICK(tc3);
  tbN.prop = curN;
  tbN.binA.r = -1; tbN.binB.r = -1; tbN.insub.r = 0; tbN.list.r = 0;
  Proposal prN;
  const int perr = propose(curN, hwmN, m, &rng, &prN, tbN, caN);
  S4B_PTICK(tc4);
  S4B_TICK(tk3);
  if (isCand) {
    spin_until(&S.verdict, 1, a.errFlag);
    if (__hip_atomic_load(&S.verdict, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 1 + candIdx) return;
  }
  S4B_PTICK(tc5);
  if (perr != 0 && lane == 0) *a.errFlag |= S4B_ERR_NODE_CAPACITY;
  if (rebuilt) {   // the rebuilt memo + lists + log prior are kept
    if (laneIn) { tI[TF_NA * ts + oN + lane] = (int16_t)curN.na.r; tI[TF_DEP * ts + oN + lane] = (int16_t)curN.dep.r; tI[TF_LEAF * ts + oN + lane] = (int16_t)caN.leaf.r;
                  tI[TF_PRE * ts + oN + lane] = (int16_t)caN.pre.r; tI[TF_POST * ts + oN + lane] = (int16_t)caN.post.r; }
    if (lane == 0) { tS[TI_NL * tT + next] = caN.nl; tS[TI_NI * tT + next] = caN.ni; tS[TI_G * tT + next] = caN.g; tS[TI_GN * tT + next] = caN.gn; a.clogpi[next] = caN.logPi; tS[TI_VALID * tT + next] = 1; }
  }
  {
    const int cntOut = prN.hwm;
    wave_tree_store(tbN.prop, sN + SF_VAR * nc, (uint16_t*)(sN + SF_CUT * nc), sN + SF_LEFT * nc, sN + SF_RIGHT * nc, sN + SF_PARENT * nc, cntOut, lane);
    if (lane < cntOut) { sN[SF_BINA * nc + lane] = (int16_t)tbN.binA.r; sN[SF_BINB * nc + lane] = (int16_t)tbN.binB.r; cN.insub[lane] = (uint8_t)tbN.insub.r;
                         sN[SF_NA * nc + lane] = (int16_t)tbN.prop.na.r; sN[SF_DEP * nc + lane] = (int16_t)tbN.prop.dep.r; }
    proposal_store(prN, cN.prop, lane);
  }
  // ---- generator state of the surviving stream back to global
  if (rng.regen) {   // (see above)
    rng.close(); S.rng[slot].pad = 0;
#pragma unroll
    for (int j = 0; j < MTJ; ++j) { const int i = lane + j * 64; mtw[j] = ((const uint32_t*)&S.rng[slot])[i < MTW ? i : 0]; }
#pragma unroll
    for (int j = 0; j < MTJ; ++j) { const int i = lane + j * 64; if (i < MTW) ((uint32_t*)a.rng)[i] = mtw[j]; }
  } else if (lane == 0) a.rng->mti = rng.mti;
#ifdef S4B_CONTROL_TIMING
  { S4B_TICK(tk4);
    if (lane == 0 && doDecide && doPropose) {
      if (isDecider) {
        atomicAdd((unsigned long long*)&g_dbg[0], (unsigned long long)(tkB - tk0)); atomicAdd((unsigned long long*)&g_dbg[1], (unsigned long long)(tk2 - tkB));
        atomicAdd((unsigned long long*)&g_dbg[2], (unsigned long long)(tk3 - tk2)); atomicAdd((unsigned long long*)&g_dbg[3], (unsigned long long)(tk4 - tk3));
        atomicAdd((unsigned long long*)&g_dbg[4], 1ull);
        atomicAdd((unsigned long long*)&g_dbg[5], (unsigned long long)(tkA - tk0)); atomicAdd((unsigned long long*)&g_dbg[6], (unsigned long long)(tkB - tkA));
      } else {
        atomicAdd((unsigned long long*)&g_dbg[14], (unsigned long long)(tk3 - tk0)); atomicAdd((unsigned long long*)&g_dbg[15], (unsigned long long)(tk4 - tk0));
        atomicAdd((unsigned long long*)&g_dbg[16], (unsigned long long)(tc1 - tk0)); atomicAdd((unsigned long long*)&g_dbg[17], (unsigned long long)(tc2 - tc1));
        atomicAdd((unsigned long long*)&g_dbg[18], (unsigned long long)(tc3 - tc2)); atomicAdd((unsigned long long*)&g_dbg[19], (unsigned long long)(tc4 - tc3));
        atomicAdd((unsigned long long*)&g_dbg[20], (unsigned long long)(tc5 - tc4)); atomicAdd((unsigned long long*)&g_dbg[21], (unsigned long long)(tk4 - tc5));
        atomicAdd((unsigned long long*)&g_dbg[22], (unsigned long long)rng.count);
        { const int ty = prN.type & 3; atomicAdd((unsigned long long*)&g_dbg[25 + ty], (unsigned long long)(tc4 - tc3)); atomicAdd((unsigned long long*)&g_dbg[32 + ty], 1ull); }
        atomicAdd((unsigned long long*)&g_dbg[23], (unsigned long long)(tc4 - S.tPost + 100000)); atomicAdd((unsigned long long*)&g_dbg[24], (unsigned long long)(tk0 - S.tStart0 + 100000));
      }
    } }
#endif
}
#endif   // S4B_SWEEP_TU

#include "dev_step.inc"
// The persistent sweep is compiled in a translation unit of its own (dev_sweep.hip includes this file with S4B_SWEEP_TU: the
// helpers above + dev_step.inc + the kernel), with -mllvm -disable-machine-licm: the kernel is one long loop over the trees, and
// hoisting every loop-invariant constant and mask out of it costs ~190 more spilled vector registers than it saves instructions.
// Here: its declarations only.
#include "dev_sweep.inc"
#ifndef S4B_SWEEP_TU

// ------------------------------------------------------------------------------------------------
// k_apply: R_i += mu_old[leaf] - mu_new[leaf'], relabel observations under the accepted move's root
__global__ __launch_bounds__(BLOCK) void k_apply(BartArrays a, int t) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const StepScratch& c = a.sc[t & 1];
  const int acc = *c.accepted;
  const int root = c.prop->node;
  const int hwm = c.prop->hwm > a.hwm[t] ? c.prop->hwm : a.hwm[t];
  const int nc = a.nc;
  double* muOld = (double*)smem; double* muNew = muOld + nc;
  int16_t* var = (int16_t*)(muNew + nc); uint16_t* cut = (uint16_t*)(var + nc); int16_t* left = (int16_t*)(cut + nc); int16_t* right = left + nc;
  uint8_t* insub = (uint8_t*)(right + nc);
  const size_t o = (size_t)t * nc;
  for (int i = threadIdx.x; i < hwm; i += BLOCK) {
    muOld[i] = c.muOld[i]; muNew[i] = a.mu[o + i]; var[i] = a.var[o + i]; cut[i] = a.cut[o + i]; left[i] = a.left[o + i]; right[i] = a.right[o + i];
    insub[i] = c.insub[i];
  }
  __syncthreads();
  const int64_t nQuads = (a.n + 3) >> 2;
  uint16_t* __restrict__ leafPlane = a.leaf + (size_t)t * a.npad;
  double* __restrict__ R = a.R;
  for (int64_t qd = (int64_t)blockIdx.x * BLOCK + threadIdx.x; qd < nQuads; qd += (int64_t)gridDim.x * BLOCK) {
    const int64_t i0 = qd << 2;
    double2 r01 = *reinterpret_cast<const double2*>(R + i0);
    double2 r23 = *reinterpret_cast<const double2*>(R + i0 + 2);
    ushort4 lf4 = *reinterpret_cast<const ushort4*>(leafPlane + i0);
    double rr[4] = {r01.x, r01.y, r23.x, r23.y};
    unsigned lf[4] = {lf4.x, lf4.y, lf4.z, lf4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (i0 + e >= a.n) break;
      const unsigned l = lf[e];
      unsigned nl = l;
      if (acc && insub[l]) {
        int nd = root;
        int v = var[nd];
        while (v >= 0) {
          const unsigned x = a.xbin[(size_t)v * a.npad + (size_t)(i0 + e)];
          nd = (x <= (unsigned)cut[nd]) ? left[nd] : right[nd];
          v = var[nd];
        }
        nl = (unsigned)nd;
      }
      rr[e] = (rr[e] + muOld[l]) - muNew[nl];
      lf[e] = nl;
    }
    *reinterpret_cast<double2*>(R + i0) = make_double2(rr[0], rr[1]);
    *reinterpret_cast<double2*>(R + i0 + 2) = make_double2(rr[2], rr[3]);
    if (acc) *reinterpret_cast<ushort4*>(leafPlane + i0) = make_ushort4((unsigned short)lf[0], (unsigned short)lf[1], (unsigned short)lf[2], (unsigned short)lf[3]);
  }
}

// ------------------------------------------------------------------------------------------------
// tree initialisation: full traversal for every tree, residual from scratch
// withResidual = 0 (set_state): only the leaf planes, the residual is given
__global__ __launch_bounds__(BLOCK) void k_assign_leaves(BartArrays a, int withResidual) {
  const ScaleState sc = *a.scale;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * BLOCK) {
    double r = a.binary ? a.lat[i] : (a.y[i] - a.off[i] - sc.min) / sc.range - 0.5;
    for (int t = 0; t < a.T; ++t) {
      const size_t o = (size_t)t * a.nc;
      int nd = 0;
      int v = a.var[o];
      while (v >= 0) {
        const unsigned x = a.xbin[(size_t)v * a.npad + (size_t)i];
        nd = (x <= (unsigned)a.cut[o + nd]) ? a.left[o + nd] : a.right[o + nd];
        v = a.var[o + nd];
      }
      a.leaf[(size_t)t * a.npad + (size_t)i] = (uint16_t)nd;
      r -= a.mu[o + nd];
    }
    if (withResidual) a.R[i] = r;
  }
}

// ------------------------------------------------------------------------------------------------
// offsets / response rescaling
struct StanArrays {
  int32_t K, q; int64_t nnz;
  const double* X;          // [K][n]
  const double* w; const int32_t* v; const int32_t* u;   // CSR of Z
  // CSC of Z in fixed chunks for the deterministic Z'e
  const int32_t* cscRow; const double* cscVal; const int32_t* chunkCol; const int64_t* chunkStart; const int32_t* chunkLen; int32_t numChunks;
  const int32_t* colChunkPtr;   // [q+1] chunk range of every column
  double* params;           // [K + q] beta, b (device copy)
  double* e0;               // [n]  y - stanOffset
  double* e;                // [n]  per-leapfrog residual (hmc_mode 1) / scratch
  double* train;            // [n]  BART fit on the data scale
  double* part;             // [(1 + K)][grid] partial sums
  double* chunkPart;        // [numChunks]
  double* out;              // [1 + K + q] reduced: ss, X'e, Z'e
  double* mmPart;           // [2][grid] min / max partials
};

__device__ __forceinline__ double param_mean_at(const StanArrays& s, int64_t n, int64_t i, int fixed, int random) {
  double eta = 0.0;
  if (fixed) for (int k = 0; k < s.K; ++k) eta += s.X[(size_t)k * n + i] * s.params[k];
  if (random && s.q) for (int e = s.u[i]; e < s.u[i + 1]; ++e) eta += s.w[e] * s.params[s.K + s.v[e]];
  return eta;
}

__global__ __launch_bounds__(BLOCK) void k_param_mean(BartArrays a, StanArrays s, int fixed, int random, int addUser, int fromHost, double* dst) {
  double mn = INFINITY, mx = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * BLOCK) {
    double eta;
    if (fromHost) eta = dst[i];
    else { eta = param_mean_at(s, a.n, i, fixed, random); if (addUser) eta += a.userOffset[i]; dst[i] = eta; }
    const double v = a.y[i] - eta;
    mn = fmin(mn, v); mx = fmax(mx, v);
  }
  __shared__ double smn[4], smx[4];
  mn = wave_min(mn); mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s.mmPart[blockIdx.x] = fmin(fmin(smn[0], smn[1]), fmin(smn[2], smn[3]));
    s.mmPart[a.grid + blockIdx.x] = fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3]));
  }
}

// one wave: new scale (when update) + sigma on the rescaled scale; then rescales every leaf value
__global__ void k_scale(BartArrays a, StanArrays s, int update, int gridUsed) {
  __shared__ ScaleState sh;
  __shared__ double smn[16], smx[16];
  if (update) {   // (min / max: any order gives the same result)
    double mn = INFINITY, mx = -INFINITY;
    for (int b = threadIdx.x; b < gridUsed; b += blockDim.x) { mn = fmin(mn, s.mmPart[b]); mx = fmax(mx, s.mmPart[a.grid + b]); }
    for (int o = 32; o; o >>= 1) { mn = fmin(mn, __shfl_xor(mn, o)); mx = fmax(mx, __shfl_xor(mx, o)); }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    ScaleState sc = *a.scale;
    sc.min0 = sc.min; sc.range0 = sc.range; sc.shiftPerTree = 0.0;
    if (update) {
      double mn = INFINITY, mx = -INFINITY;
      for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) { mn = fmin(mn, smn[w]); mx = fmax(mx, smx[w]); }
      sc.min = mn; sc.max = mx; sc.range = mx - mn;
      sc.shiftPerTree = (sc.min0 + 0.5 * sc.range0 - sc.min - 0.5 * sc.range) / (double)a.T;
    }
    sc.sigma = sc.sigmaData / sc.range;
    *a.scale = sc; sh = sc;
  }
  __syncthreads();
  if (update) {
    const ScaleState sc = sh;
    const size_t m = (size_t)a.T * a.nc;
    for (size_t k = threadIdx.x; k < m; k += blockDim.x) a.mu[k] = (sc.range0 * a.mu[k] + sc.shiftPerTree) / sc.range;
  }
}

// The k hyperprior's step (dev_common.hpp k_hyper_*): one workgroup; thread i sums the squared leaf values of trees i, i + 256, ..., thread 0
// adds the 256 partial sums in index order and draws k from R's stream where the sweep (and the latents) left it.  out (host-mapped):
// {k, leaf prior precision for that k} — the host puts the precision into the kernel arguments of the next sweep.
__global__ __launch_bounds__(256) void k_draw_k(BartArrays a, KHyper h, double kOld, double* out) {
  __shared__ double ss[256], mm[256];
  double s = 0.0, m = 0.0;
  for (int t = threadIdx.x; t < a.T; t += 256) { double s1, m1; k_hyper_tree_stats(a, t, s1, m1); s += s1; m += m1; }
  ss[threadIdx.x] = s; mm[threadIdx.x] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    double S = 0.0, M = 0.0;
    for (int i = 0; i < 256; ++i) { S += ss[i]; M += mm[i]; }
    const double k = k_hyper_draw(a.rng, h, a.T, S, M, kOld);
    out[0] = k; out[1] = leaf_precision(k, a.T, h.nodeScale);
    __threadfence_system();
  }
}

__global__ void k_set_sigma(BartArrays a, double sigmaData) {
  a.scale->sigmaData = sigmaData; a.scale->sigma = sigmaData / a.scale->range;
}

__global__ __launch_bounds__(BLOCK) void k_rescale(BartArrays a, int update) {
  const ScaleState sc = *a.scale;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * BLOCK) {
    const double y = a.y[i];
    const double yOld = (y - a.off[i] - sc.min0) / sc.range0 - 0.5;
    double F = yOld - a.R[i];
    if (update) F = (sc.range0 * F + (double)a.T * sc.shiftPerTree) / sc.range;
    const double yNew = (y - a.offNew[i] - sc.min) / sc.range - 0.5;
    a.R[i] = yNew - F;
  }
}

// The Stan -> BART hand-off of an iteration that does not update the response scale (every sampling-phase iteration, most warm-up
// ones) in ONE pass: k_param_mean + k_set_sigma + k_scale(update = 0) + k_rescale(update = 0) with the coefficients in the kernel
// arguments (K + q <= 64) — same arithmetic per observation, in the same order, as the four kernels.
struct OffsetArgs { int32_t fixed, random, addUser, pad; double sigmaData; double par[64]; };
__global__ __launch_bounds__(BLOCK) void k_offset_rescale(BartArrays a, StanArrays s, OffsetArgs o) {
  __shared__ double par[64];
  if (threadIdx.x < 64) par[threadIdx.x] = o.par[threadIdx.x];
  __syncthreads();
  const double mn = a.scale->min, range = a.scale->range;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * BLOCK) {
    double eta = 0.0;
    if (o.fixed) for (int k = 0; k < s.K; ++k) eta += s.X[(size_t)k * a.n + i] * par[k];
    if (o.random && s.q) for (int e = s.u[i]; e < s.u[i + 1]; ++e) eta += s.w[e] * par[s.K + s.v[e]];
    if (o.addUser) eta += a.userOffset[i];
    a.offNew[i] = eta;
    const double y = a.y[i];
    const double yOld = (y - a.off[i] - mn) / range - 0.5;
    const double F = yOld - a.R[i];
    const double yNew = (y - eta - mn) / range - 0.5;
    a.R[i] = yNew - F;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {   // (min and range keep their values: the other workgroups may read them meanwhile)
    ScaleState sc = *a.scale;
    sc.min0 = sc.min; sc.range0 = sc.range; sc.shiftPerTree = 0.0;
    sc.sigmaData = o.sigmaData; sc.sigma = o.sigmaData / sc.range;
    *a.scale = sc;
  }
}

// probit: latent response z_i ~ N(fit_i + offset_i, 1) truncated to (0, inf) if y_i = 1, (-inf, 0] otherwise
// (dbarts sampleProbitLatentVariables; reference consumes it through storeLatents, src/init.cpp:289,845).
// The draws come from R's sequential generator with a data-dependent number of uniforms per observation, so the
// stream order forces a serial loop: one workgroup stages chunks through LDS, lane 0 draws, all lanes store.
__device__ double lower_trunc_std_normal(MTState* rng, double lower) {
  double x;
  if (lower < 0.0) { x = r_norm(rng); while (x < lower) x = r_norm(rng); }
  else {
    const double aa = 0.5 * (lower + sqrt(lower * lower + 4.0));
    double u, r;
    do { x = r_exp(rng) / aa + lower; u = r_unif(rng); const double d = x - aa; r = exp(-0.5 * d * d); } while (u > r);
  }
  return x;
}
// Lane-parallel where the stream allows it.  The heavy arithmetic of norm_rand (the AS241 quantile) only depends on the
// stream position, not on the observation: wave 1 (producer) walks the generator ahead, window by window of 64 positions,
// and publishes for every position k the tempered output u_k and Z_k = the normal deviate a norm_rand() starting at k would
// return (it uses u_k, u_{k+1}).  Wave 0 (consumer) keeps the current window in registers and runs the sequential part —
// which observation consumes how many positions — with register reads: a rejection step of the common branch is a
// v_readlane and a compare.  The exponential-rejection branch (lower bound >= 0) reads the same uniforms through the
// window.  Same positions, same arithmetic as the serial loop => same latents, same generator state afterwards.
constexpr int LAT_RING = 8;
struct LatShared {
  uint32_t mt[2][626];          // generator blocks (MTState layout: 624 words + mti + pad): block b lives in mt[b & 1]
  uint32_t U[LAT_RING][64];     // tempered outputs of window w in slot w % LAT_RING
  double Z[LAT_RING][64];       // norm_rand() value for a draw starting at each position of the window
  int produced, consumed, stop;
};
struct WindowRng {              // the consumer's view of the stream (absolute output index g; block 0 = the incoming state)
  LatShared* S; int64_t g; int64_t wc; uint32_t ureg; int zlo, zhi; int32_t* errFlag;
  __device__ __forceinline__ void need() {
    const int64_t w = g >> 6;
    if (w != wc) {
      int guard = 0;
      while (__hip_atomic_load(&S->produced, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= (int)w) {
        __builtin_amdgcn_s_sleep(1);
        if (++guard > (1 << 24)) { *errFlag |= S4B_ERR_INTERNAL; break; }
      }
      const int lane = (int)(threadIdx.x & 63), slot = (int)(w % LAT_RING);
      ureg = S->U[slot][lane];
      const double z = S->Z[slot][lane];
      zlo = __double2loint(z); zhi = __double2hiint(z);
      wc = w;
      if (lane == 0) __hip_atomic_store(&S->consumed, (int)w, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // windows < w are free
    }
  }
  __device__ __forceinline__ double norm() {   // norm_rand(): two positions
    need();
    const int l = (int)(g & 63);
    const double z = __hiloint2double(__builtin_amdgcn_readlane(zhi, l), __builtin_amdgcn_readlane(zlo, l));
    g += 2;
    return z;
  }
};
__device__ __forceinline__ uint32_t mt_next(WindowRng* r) {
  r->need();
  const uint32_t y = (uint32_t)__builtin_amdgcn_readlane((int)r->ureg, (int)(r->g & 63));
  r->g += 1;
  return y;
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
  return y;
}
__device__ __forceinline__ double unif_fix(uint32_t raw) {   // unif_rand()'s mapping of a tempered output (rrng_hd.hpp r_unif)
  const double half_ulp = 0.5 * 2.328306437080797e-10;
  const double v = (double)raw * 2.3283064365386963e-10;
  if (v <= 0.0) return half_ulp;
  if (1.0 - v <= 0.0) return 1.0 - half_ulp;
  return v;
}

__global__ __launch_bounds__(BLOCK) void k_latents(BartArrays a) {
  __shared__ LatShared S;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int mti0 = a.rng->mti;                      // 0..624: absolute output index of the first draw
  const int64_t w0 = mti0 >> 6;
  if (threadIdx.x == 0) { S.produced = 0; S.consumed = (int)w0; S.stop = 0; }   // windows < produced are published, windows < consumed are free
  for (int i = threadIdx.x; i < 624; i += BLOCK) S.mt[0][i] = a.rng->mt[i];
  __syncthreads();
  if (wv >= 2) return;

  if (wv == 1) {   // ------------------------------------------------ producer
    int genBlocks = 1;
    // window w starts at absolute output index 64 w = 624 bw + iw (bw, iw advanced incrementally: no divisions)
    int bw = (int)((64 * w0) / 624), iw = (int)(64 * w0 - (int64_t)bw * 624);
    auto raw = [&]() -> uint32_t {   // tempered output of this lane's position in the window at (bw, iw); advances to the next window
      const int bMax = iw + 63 >= 624 ? bw + 1 : bw;
      while (genBlocks <= bMax) {   // next block = copy of the previous one, regenerated in place (whole wave)
        uint32_t* src = S.mt[(genBlocks - 1) & 1]; uint32_t* dst = S.mt[genBlocks & 1];
        for (int i = lane; i < 624; i += 64) dst[i] = src[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        mt_regenerate_wave((MTState*)dst);   // (also writes the mti word of the buffer, unused here)
        ++genBlocks;
      }
      int idx = iw + lane, b = bw;
      if (idx >= 624) { idx -= 624; b += 1; }
      const uint32_t v = mt_temper(S.mt[b & 1][idx]);
      iw += 64; if (iw >= 624) { iw -= 624; bw += 1; }
      return v;
    };
    int64_t w = w0;
    uint32_t unext = raw();
    for (;;) {
      int guard = 0; bool stop = false;
      for (;;) {   // ring space, or the consumer is done
        if (__hip_atomic_load(&S.stop, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) { stop = true; break; }
        if ((int)w - __hip_atomic_load(&S.consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < LAT_RING) break;
        __builtin_amdgcn_s_sleep(1);
        if (++guard > (1 << 24)) { *a.errFlag |= S4B_ERR_INTERNAL; stop = true; break; }
      }
      if (stop) break;
      const uint32_t ucur = unext;
      unext = raw();
      const uint32_t unb = (uint32_t)__shfl_down((int)ucur, 1, 64);
      const uint32_t u2raw = lane < 63 ? unb : (uint32_t)__builtin_amdgcn_readlane((int)unext, 0);
      const double BIG = 134217728.0;
      const double z = r_qnorm(((double)(int)(BIG * unif_fix(ucur)) + unif_fix(u2raw)) / BIG);
      const int slot = (int)(w % LAT_RING);
      S.U[slot][lane] = ucur; S.Z[slot][lane] = z;
      ++w;
      if (lane == 0) __hip_atomic_store(&S.produced, (int)w, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // windows < w are published
    }
    return;
  }

  // -------------------------------------------------------------------- consumer (wave 0)
  WindowRng wr; wr.S = &S; wr.g = mti0; wr.wc = -1; wr.ureg = 0u; wr.zlo = 0; wr.zhi = 0; wr.errFlag = a.errFlag;
  const int64_t nBatches = (a.n + 63) >> 6;
  auto load = [&](int64_t bi, double& mean, double& fOld, double& offv, int& pos) {
    const int64_t i = (bi << 6) + lane;
    mean = 0.0; fOld = 0.0; offv = 0.0; pos = 0;
    if (bi < nBatches && i < a.n) { fOld = a.lat[i] - a.R[i]; offv = a.off[i]; mean = fOld + offv; pos = a.y[i] > 0.0 ? 1 : 0; }
  };
  double meanN, fN, offN; int posN;
  load(0, meanN, fN, offN, posN);
  for (int64_t bi = 0; bi < nBatches; ++bi) {
    const double mean = meanN, fOld = fN, offv = offN; const int pos = posN;
    load(bi + 1, meanN, fN, offN, posN);
    const int cnt = (int)((a.n - (bi << 6)) < 64 ? (a.n - (bi << 6)) : 64);
    const int mlo = __double2loint(mean), mhi = __double2hiint(mean);
    double zout = 0.0;
    for (int j = 0; j < cnt; ++j) {
      const double m = __hiloint2double(__builtin_amdgcn_readlane(mhi, j), __builtin_amdgcn_readlane(mlo, j));
      const int yp = __builtin_amdgcn_readlane(pos, j);
      const double lower = yp ? 0.0 - m : m - 0.0;
      double x;
      if (lower < 0.0) { x = wr.norm(); while (x < lower) x = wr.norm(); }
      else {
        const double aa = 0.5 * (lower + sqrt(lower * lower + 4.0));
        double u, r;
        do { x = r_exp(&wr) / aa + lower; u = r_unif(&wr); const double d = x - aa; r = exp(-0.5 * d * d); } while (u > r);
      }
      const double z = yp ? m + x : m - x;
      zout = lane == j ? z : zout;
    }
    const int64_t i = (bi << 6) + lane;
    if (i < a.n) { const double nl = zout - offv; a.lat[i] = nl; a.R[i] = nl - fOld; }
  }
  // generator state after the last draw (lazy regeneration: a position on a block boundary stays in the old block, mti 624)
  {
    if (lane == 0) __hip_atomic_store(&S.stop, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int64_t g = wr.g;
    int b = (int)(g / 624), idx = (int)(g - (int64_t)b * 624);
    if (idx == 0 && b > 0) { b -= 1; idx = 624; }
    // the producer never runs more than LAT_RING + 1 windows (< 624 positions) ahead: block b is still in its buffer
    for (int k = lane; k < 624; k += 64) a.rng->mt[k] = S.mt[b & 1][k];
    if (lane == 0) { a.rng->mti = idx; a.rng->pad = 0; }
  }
}

// ------------------------------------------------------------------------------------------------
// k_latents2 + k_latents_finish: the same probit latents, same stream positions, same generator state as k_latents, ~20x faster.
// Which stream positions an observation consumes depends on where the previous observation stopped, but for a GIVEN start
// everything about an observation is independent of the others.  So, per batch of 32 observations:
//   build    (16 waves, lane-parallel): for observation i and every candidate start "slack" l = 0..63 (start position =
//            batch base + 2 i + l: every observation consumes at least two positions, so only the surplus is uncertain) run
//            the rejection logic on the position-indexed stream — table T[i][l] = slack the next observation starts with,
//            X[i][l] = the accepted deviate;
//   resolve  (one wave): the dependent chain collapses to o <- T[i][o]: one v_readlane per observation;
//   finish   (separate, fully parallel kernel): latent = mean +- x, residual update.
// The stream (tempered outputs U, and Z = the normal deviate a norm_rand() starting at each position returns) is produced
// ahead, whole Mersenne-Twister blocks at a time, by all waves; a short history of blocks is kept so that the generator state
// handed back is the one a sequential consumer would leave.
constexpr int LB2 = 1024;
#ifndef S4B_L_NB
#define S4B_L_NB 32
#endif
constexpr int L_RING = 4096, L_CH = 2048, L_NB = S4B_L_NB, L_BLK = 8, L_EMAX = 20;
struct Lat2Lds {
  uint32_t (*mt)[626]; uint32_t* U; double* Z; double* E; uint8_t* EL; double* lower; double (*X)[64]; uint8_t (*T)[64];
};
static size_t lat2_lds_bytes() {
  return (size_t)L_BLK * 626 * 4 + (size_t)L_RING * (4 + 8 + 8 + 1) + (size_t)L_CH * 8 + (size_t)L_NB * 64 * 8 + (size_t)L_NB * 64 + 64;
}
struct RingRng {   // position-indexed reader of the stream for the generic samplers (r_unif / r_exp)
  const uint32_t* U; long long p;
};
__device__ __forceinline__ uint32_t mt_next(RingRng* r) {
  const uint32_t y = r->U[r->p & (L_RING - 1)];
  r->p += 1;
  return y;
}
__global__ __launch_bounds__(LB2) void k_latents2(BartArrays a, double* xacc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Lat2Lds S;
  {
    unsigned char* b = smem;
    S.Z = (double*)b; b += (size_t)L_RING * 8;
    S.E = (double*)b; b += (size_t)L_RING * 8;
    S.lower = (double*)b; b += (size_t)L_CH * 8;
    S.X = (double (*)[64])b; b += (size_t)L_NB * 64 * 8;
    S.mt = (uint32_t (*)[626])b; b += (size_t)L_BLK * 626 * 4;
    S.U = (uint32_t*)b; b += (size_t)L_RING * 4;
    S.EL = (uint8_t*)b; b += (size_t)L_RING;
    S.T = (uint8_t (*)[64])b;
  }
  __shared__ long long shBase; __shared__ int shCnt, shStop;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block 0 = the incoming state; absolute stream position g = 624 * block + index
  for (int i = tid; i < 624; i += LB2) { const uint32_t v = a.rng->mt[i]; S.mt[0][i] = v; S.U[i] = mt_temper(v); }
  long long base = a.rng->mti;          // position of the next draw
  int nblk = 1;                         // blocks generated so far: positions [0, 624 nblk) exist
  long long zHi = 0;                    // Z valid for positions < zHi, E / EL for positions < zHi - L_EMAX
  long long eDone = 0;
  __syncthreads();
  const double BIG = 134217728.0;
  // What depends on the stream position only, for every position, lane-parallel: Z[p] = the deviate a norm_rand() starting at p
  // returns (2 positions), E[p] / EL[p] = the value an exp_rand() starting at p returns and the positions it consumes (<= 18)
  auto refill = [&]() {   // uniform control flow: every thread executes the same sequence of barriers
    bool any = false;
    while ((long long)624 * nblk - base < 1536 && (long long)624 * (nblk + 1) - base <= L_RING) {
      uint32_t* src = S.mt[(nblk - 1) % L_BLK]; uint32_t* dst = S.mt[nblk % L_BLK];
      if (wv == 0) {
        for (int i = lane; i < 624; i += 64) dst[i] = src[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        mt_regenerate_wave((MTState*)dst);
      }
      __syncthreads();
      for (int i = tid; i < 624; i += LB2) S.U[((long long)624 * nblk + i) & (L_RING - 1)] = mt_temper(dst[i]);
      ++nblk; any = true;
      __syncthreads();
    }
    if (!any && zHi != 0) return;
    const long long genHi = (long long)624 * nblk;
    const long long lo = zHi > base ? zHi : base;
    for (long long p = lo + tid; p < genHi - 1; p += LB2) {
      const double u1 = unif_fix(S.U[p & (L_RING - 1)]), u2 = unif_fix(S.U[(p + 1) & (L_RING - 1)]);
      S.Z[p & (L_RING - 1)] = r_qnorm(((double)(int)(BIG * u1) + u2) / BIG);
    }
    const long long elo = eDone > base ? eDone : base;
    for (long long p = elo + tid; p < genHi - L_EMAX; p += LB2) {
      RingRng r; r.U = S.U; r.p = p;
      S.E[p & (L_RING - 1)] = r_exp(&r);
      S.EL[p & (L_RING - 1)] = (uint8_t)(r.p - p);
    }
    zHi = genHi - 1; eDone = genHi - L_EMAX;
    __syncthreads();
  };
  for (int64_t c0 = 0; c0 < a.n; c0 += L_CH) {
    const int chN = (int)((a.n - c0) < L_CH ? (a.n - c0) : L_CH);
    __syncthreads();
    for (int i = tid; i < chN; i += LB2) {   // truncation bound of every observation of the chunk
      const int64_t g = c0 + i;
      const double fOld = a.lat[g] - a.R[g], mean = fOld + a.off[g];
      S.lower[i] = a.y[g] > 0.0 ? 0.0 - mean : mean - 0.0;
    }
    __syncthreads();
    int done = 0;
    while (done < chN) {
#ifdef S4B_CONTROL_TIMING
      const long long tl0 = wall_clock64();
#endif
      refill();
#ifdef S4B_CONTROL_TIMING
      const long long tl1 = wall_clock64();
#endif
      const int nbatch = (chN - done) < L_NB ? (chN - done) : L_NB;
      // ---- build: one observation per wave at a time, lane = candidate slack
      for (int i = wv; i < nbatch; i += LB2 / 64) {
        const double lower = S.lower[done + i];
        const long long p0 = base + 2 * i;
        long long next; double x = 0.0; int bad = 0;
        if (lower < 0.0) {
          // norm_rand() until the deviate is >= lower: the first accepted position at or after the start with the start's parity.
          // Acceptance of the 128 positions from p0 as two wave-wide bit masks; every lane then needs a shift and a count.
          const long long pa = p0 + lane, pb = p0 + 64 + lane;
          const double za = S.Z[pa & (L_RING - 1)], zb = S.Z[pb & (L_RING - 1)];
          const unsigned long long ma = __ballot(pa < zHi && !(za < lower)), mb = __ballot(pb < zHi && !(zb < lower));
          const unsigned long long win = lane == 0 ? ma : ((ma >> lane) | (mb << (64 - lane)));   // acceptance of positions pa, pa+1, ...
          const unsigned long long cand = win & 0x5555555555555555ull;
          if (cand != 0ull) {
            const int k = __builtin_ctzll(cand);
            x = S.Z[(pa + k) & (L_RING - 1)];
            next = pa + k + 2;
          } else {   // 32 rejections in a row (or the generated range ends): the plain loop
            long long p = pa + 64;
            for (;;) {
              if (p >= zHi) { bad = 1; break; }
              x = S.Z[p & (L_RING - 1)];
              p += 2;
              if (!(x < lower)) break;
            }
            next = p;
            if (pa + 64 > zHi) bad = 1;
          }
        } else {
          long long p = p0 + lane;
          const double aa = 0.5 * (lower + sqrt(lower * lower + 4.0));
          for (;;) {
            if (p >= eDone) { bad = 1; break; }
            const int len = S.EL[p & (L_RING - 1)];
            x = S.E[p & (L_RING - 1)] / aa + lower;
            const double u = unif_fix(S.U[(p + len) & (L_RING - 1)]);
            p += len + 1;
            const double d = x - aa;
            if (!(u > exp(-0.5 * d * d))) break;
          }
          next = p;
        }
        const long long sl = next - (p0 + lane) + lane - 2;      // slack of the next observation
        // (a slack beyond the 8-bit table — one observation consuming more than ~250 positions — is treated like a candidate that ran
        // out of generated positions: the chain stops before it; were it the first observation of a batch with the ring full, the
        // batch makes no progress and the kernel raises S4B_ERR_INTERNAL below instead of continuing from a wrong position)
        S.T[i][lane] = (bad || sl > 254) ? (uint8_t)255 : (uint8_t)sl;
        S.X[i][lane] = x;
      }
      __syncthreads();
#ifdef S4B_CONTROL_TIMING
      const long long tl2 = wall_clock64();
#endif
      // ---- resolve: o <- T[i][o]
      if (wv == 0) {
        int t[L_NB];
#pragma unroll
        for (int i = 0; i < L_NB; ++i) t[i] = (int)S.T[i][lane];      // unconditional: all reads in flight together (a read under a
#pragma unroll
        for (int i = 0; i < L_NB; ++i) t[i] = i < nbatch ? t[i] : 255;   // condition is a branch with its own wait)
#ifdef S4B_CONTROL_TIMING
        __builtin_amdgcn_s_waitcnt(0); const long long tr1 = wall_clock64();
#endif
        int o = 0, cnt = 0, alive = 1, mine = 0;
#pragma unroll
        for (int i = 0; i < L_NB; ++i) {
          // branch-free (selects on wave-uniform values): a branch per step costs more than the step
          const int nx = __builtin_amdgcn_readlane(t[i], o & 63);
          const int take = alive & (nx != 255 ? 1 : 0);     // observation i is resolved from slack o
          mine = (lane == i && take) ? o : mine;
          cnt += take;
          o = take ? nx : o;
          alive = take & (nx < 64 ? 1 : 0);
        }
#ifdef S4B_CONTROL_TIMING
        const long long tr2 = wall_clock64();
        if (lane == 0) { g_dbg[34] += tr1 - tl2; g_dbg[35] += tr2 - tr1; }
#endif
        if (lane < cnt) xacc[c0 + done + lane] = S.X[lane][mine];
        if (lane == 0) {
          shBase = base + 2 * cnt + o; shCnt = cnt;
          // no progress although the ring is as full as it can get: one observation would need thousands of positions
          shStop = (cnt == 0 && (long long)624 * (nblk + 1) - base > L_RING && (long long)624 * nblk - base >= 1536) ? 1 : 0;
        }
      }
      __syncthreads();
      base = shBase; done += shCnt;
      if (shStop) { if (tid == 0) *a.errFlag |= S4B_ERR_INTERNAL; done = chN; c0 = a.n; }
#ifdef S4B_CONTROL_TIMING
      if (tid == 0) { const long long tl3 = wall_clock64(); g_dbg[36] += tl1 - tl0; g_dbg[37] += tl2 - tl1; g_dbg[38] += tl3 - tl2; g_dbg[39] += 1; }
#endif
    }
  }
  __syncthreads();
  // generator state after the last draw (lazy regeneration: a position on a block boundary stays in the old block, mti 624)
  {
    int b = (int)(base / 624), idx = (int)(base - (long long)b * 624);
    if (idx == 0 && b > 0) { b -= 1; idx = 624; }
    for (int k = tid; k < 624; k += LB2) a.rng->mt[k] = S.mt[b % L_BLK][k];
    if (tid == 0) { a.rng->mti = idx; a.rng->pad = 0; }
  }
#ifdef S4B_CONTROL_TIMING
  if (tid == 0) { printf("DBG k_latents2: %lld batches (%.1f obs each), per batch us: refill %.2f build %.2f resolve+sync %.2f (table read %.2f, chain %.2f)\n", g_dbg[39], (double)a.n / (double)g_dbg[39],
                         g_dbg[36] / 100.0 / g_dbg[39], g_dbg[37] / 100.0 / g_dbg[39], g_dbg[38] / 100.0 / g_dbg[39], g_dbg[34] / 100.0 / g_dbg[39], g_dbg[35] / 100.0 / g_dbg[39]);
                         g_dbg[34] = g_dbg[35] = g_dbg[36] = g_dbg[37] = g_dbg[38] = g_dbg[39] = 0; }
#endif
}
__global__ __launch_bounds__(BLOCK) void k_latents_finish(BartArrays a, const double* xacc) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * BLOCK) {
    const double fOld = a.lat[i] - a.R[i], offv = a.off[i], mean = fOld + offv, x = xacc[i];
    const double z = a.y[i] > 0.0 ? mean + x : mean - x;
    const double nl = z - offv;
    a.lat[i] = nl; a.R[i] = nl - fOld;
  }
}

__global__ __launch_bounds__(BLOCK) void k_init_binary(BartArrays a) {   // latents 2y - 1, no tree fits yet
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ScaleState sc; sc.min = -0.5; sc.max = 0.5; sc.range = 1.0; sc.min0 = -0.5; sc.range0 = 1.0; sc.shiftPerTree = 0.0; sc.sigmaData = 1.0; sc.sigma = 1.0;
    *a.scale = sc;
  }
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * BLOCK) {
    const double z = 2.0 * a.y[i] - 1.0;
    a.lat[i] = z; a.R[i] = z;
  }
}

__global__ __launch_bounds__(BLOCK) void k_rescale_binary(BartArrays a) {   // keep latent + offset invariant
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * BLOCK) {
    const double old = a.lat[i];
    const double nl = old + (a.off[i] - a.offNew[i]);
    a.R[i] += nl - old;
    a.lat[i] = nl;
  }
}

__global__ __launch_bounds__(BLOCK) void k_init_residual(BartArrays a) {   // all tree fits zero: R = yRescaled
  const ScaleState sc = *a.scale;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * BLOCK)
    a.R[i] = (a.y[i] - a.off[i] - sc.min) / sc.range - 0.5;
}

// ------------------------------------------------------------------------------------------------
// Stan inputs: e0 = y - stanOffset, |e0|^2, X'e0 (fixed-order partials); optional BART fit output
template <int KC>
__device__ __forceinline__ void block_reduce_store(double (&acc)[KC], int kBase, int kCount, double* part, int grid) {
  __shared__ double red[4][KC];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < KC; ++k) { const double s = wave_sum(acc[k]); if (lane == 0) red[wv][k] = s; }
  __syncthreads();
  if ((int)threadIdx.x < KC && (int)threadIdx.x < kCount) {
    const int k = threadIdx.x;
    part[(size_t)(kBase + k) * grid + blockIdx.x] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
  }
  __syncthreads();
}

// mode 0: e = y; 1: y - fit; 2: y - user; 3: y - (fit + user).   direct = 1: e = e0 - X beta - Z b (leapfrog)
__global__ __launch_bounds__(BLOCK) void k_stan_inputs(BartArrays a, StanArrays s, int mode, int wantTrain, int direct) {
  const ScaleState sc = *a.scale;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};   // ss, X'e for the first 3 columns
  double* dstE = direct ? s.e : s.e0;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * BLOCK) {
    double e;
    if (direct) e = s.e0[i] - param_mean_at(s, a.n, i, 1, 1);
    else {
      double fit = 0.0, resp = a.y[i];
      if (mode != 0 || wantTrain) {
        if (a.binary) { const double z = a.lat[i]; fit = z - a.R[i]; if (mode != 0) resp = z + a.off[i]; }
        else { const double yr = (a.y[i] - a.off[i] - sc.min) / sc.range - 0.5; fit = ((yr - a.R[i]) + 0.5) * sc.range + sc.min; }
      }
      const double so = mode == 0 ? 0.0 : mode == 1 ? fit : mode == 2 ? a.userOffset[i] : fit + a.userOffset[i];
      e = resp - so;
      if (wantTrain) s.train[i] = fit;
    }
    // weighted likelihood (continuous.stan:358-366): sums of w e^2, X'(w e), Z'(w e); the followers read w e from s.e
    const double we = a.wts ? a.wts[i] * e : e;
    if (direct) dstE[i] = we; else { dstE[i] = e; if (a.wts) s.e[i] = we; }
    acc[0] += we * e;
#pragma unroll
    for (int k = 0; k < 3; ++k) if (k < s.K) acc[1 + k] += s.X[(size_t)k * a.n + i] * we;
  }
  block_reduce_store<4>(acc, 0, 1 + (s.K < 3 ? s.K : 3), s.part, a.grid);
}

// remaining columns of X'e (K > 3), four per launch
__global__ __launch_bounds__(BLOCK) void k_xt_e(BartArrays a, StanArrays s, int k0, int direct) {
  const double* src = direct ? s.e : s.e0;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * BLOCK) {
    const double e = src[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k0 + k < s.K) acc[k] += s.X[(size_t)(k0 + k) * a.n + i] * e;
  }
  block_reduce_store<4>(acc, 1 + k0, (s.K - k0) < 4 ? (s.K - k0) : 4, s.part, a.grid);
}

// Z'e: one workgroup per fixed chunk of one CSC column; rows of a column are ascending, so the gathers of e
// walk memory forward
__global__ __launch_bounds__(BLOCK) void k_zt_chunks(StanArrays s, int direct) {
  const double* src = direct ? s.e : s.e0;
  const int c = blockIdx.x;
  const int64_t st = s.chunkStart[c];
  const int len = s.chunkLen[c];
  double acc = 0.0;
  for (int j = threadIdx.x; j < len; j += BLOCK) acc += s.cscVal[st + j] * src[s.cscRow[st + j]];
  __shared__ double red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) s.chunkPart[c] = ((red[0] + red[1]) + red[2]) + red[3];
}

// final fixed-order combination: out[0] = ss, out[1..K] = X'e, out[1+K..] = Z'e
__global__ __launch_bounds__(BLOCK) void k_stan_finalize(BartArrays a, StanArrays s, int gridUsed) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int k = wv; k < 1 + s.K; k += BLOCK / 64) {
    double v = 0.0;
    for (int b = lane; b < gridUsed; b += 64) v += s.part[(size_t)k * a.grid + b];
    v = wave_sum(v);
    if (lane == 0) s.out[k] = v;
  }
  for (int j = threadIdx.x; j < s.q; j += BLOCK) {
    double v = 0.0;
    for (int c = s.colChunkPtr[j]; c < s.colChunkPtr[j + 1]; ++c) v += s.chunkPart[c];
    s.out[1 + s.K + j] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// k_stan_fused: every O(N) sum of one log-density / gradient evaluation in ONE pass and ONE launch
// (reference: the likelihood part of continuous_model::log_prob, src/stan_files/continuous.hpp:2438-2470, evaluated once per
// leapfrog through stan::math::gradient; SURVEY §8d B_lf = N (8K + 12z + 20) algorithmic bytes).
//   DIRECT   (a leapfrog, hmc_mode 1):        e_i = e0_i - x_i'beta - z_i'b, nothing O(N) is written
//   !DIRECT  (once per Gibbs iteration):      e_i = response_i - stanOffset_i from the BART state; e0 (and the fit) are written
// and in both: ss = sum w e^2, X'(w e) (K columns, register accumulators), Z'(w e) (q columns).
// Order-independent, hence deterministic, accumulation: every partial sum is multiplied by a power of two chosen per group of
// sums (|e|^2; X'e; Z'e — `scale`, from the magnitudes the previous evaluation saw), split into two 64-bit fixed-point limbs
// (2^-20 and 2^-56 units: together finer than the rounding of the double it came from) and added with INTEGER atomics — LDS
// histogram per wave for Z'e, 32 copies of the global accumulators for everything (workgroup b adds into copy b mod 32) — so there
// are no per-workgroup partial arrays and no finalize pass over them; k_stan_forward adds the copies (exact integer adds) and hands
// the result to the host.
// Range: beside the sums, every group accumulates the sum of the ABSOLUTE values of what went into it (coarse units of 2^10, rounded
// up).  While that total stays below 2^42 no partial sum anywhere (LDS slot, global copy, folded total) can leave the 64-bit limb;
// the host checks it and, when it does not hold (first evaluation, a response rescaled by orders of magnitude, a trajectory far
// outside the typical set), evaluates the same sums in plain doubles (reduce_pipeline) and re-centres the scales.
constexpr int SBLOCK = 256;
constexpr int S_COPIES = 32;            // copies of the global accumulators (workgroup b adds into copy b % 32: fewer atomics meet on one address)
constexpr int S_QMAX = 4096;          // Z'e histogram in LDS: 16 B per column (+ 8 B for b)
constexpr double S_FX_LIMIT = 4398046511104.0;   // 2^42
constexpr int S_PAR_INLINE = 64;
struct FxLimbs { long long hi, lo; };
__device__ __forceinline__ FxLimbs fx_split(double v) {
  const double SH = 1048576.0, SL = 72057594037927936.0;   // 2^20, 2^56
  const double h = rint(v * SH);
  FxLimbs r; r.hi = (long long)h; r.lo = (long long)rint((v - h / SH) * SL);
  return r;
}
__host__ __device__ static inline size_t fused_words(int M) { return (size_t)2 * M + 6; }   // limbs of the M sums + three magnitude words + three exponent words
struct StanFusedArgs {
  unsigned long long* acc;    // [2][S_COPIES][fused_words]: accumulators of this launch (parity) and of the next one (cleared here)
  int32_t* bad;               // [2]: something non-finite or beyond 2^42 after scaling went into a sum
  int32_t parity, mode, wantTrain;
  double scale[3];            // powers of two: |e|^2, X'e, Z'e
  // result hand-off without a copy command: k_stan_forward folds the copies into host memory the device can write
  // (hostOut: [fused_words] sums, [1] flag) and then publishes `seq` in hostOut[fused_words + 1]; the host polls that word
  unsigned long long* hostOut; uint32_t seq; int32_t zFixed;   // zFixed: every row of Z has exactly this many non-zeros (u[i] = zFixed i), -1: general CSR
  // DIRECT: beta, b travel in the kernel arguments when they are few (no host-to-device copy, no wait for the staging buffer)
  int32_t parInline, pad; double par[S_PAR_INLINE];
};
constexpr int S_ZMAX = 4;            // non-zeros per row the fixed-row-length path keeps in registers
template <int KMAX, bool DIRECT>
__global__ __launch_bounds__(SBLOCK) void k_stan_fused(BartArrays a, StanArrays s, StanFusedArgs f) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = s.K, q = s.q, M = 1 + K + q;
  // Z'e histogram: one copy per wave while that fits (fewer lanes meet on one address), [copy][q][2 limbs]
  const int nCopy = (size_t)q * 16 * (SBLOCK / 64) <= 32768 ? SBLOCK / 64 : 1;
  unsigned long long* zhAll = (unsigned long long*)smem;
  double* par = (double*)(smem + (size_t)q * 16 * nCopy);            // [K + q] beta, b (DIRECT)
  __shared__ double red[SBLOCK / 64][KMAX + 3];
  const size_t W = fused_words(M);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long* zh = zhAll + (size_t)(nCopy > 1 ? wv : 0) * 2 * q;
  for (int j = threadIdx.x; j < 2 * q * nCopy; j += SBLOCK) zhAll[j] = 0ull;
  if (DIRECT) for (int j = threadIdx.x; j < K + q; j += SBLOCK) par[j] = f.parInline ? f.par[j < S_PAR_INLINE ? j : 0] : s.params[j];
  {   // clear the accumulators the NEXT evaluation uses (the host has consumed them: it waits for every evaluation)
    unsigned long long* other = f.acc + (size_t)(1 - f.parity) * S_COPIES * W;
    const size_t tot = (size_t)S_COPIES * W;
    for (size_t j = (size_t)blockIdx.x * SBLOCK + threadIdx.x; j < tot; j += (size_t)gridDim.x * SBLOCK) other[j] = 0ull;
    if (blockIdx.x == 0 && threadIdx.x == 0) f.bad[1 - f.parity] = 0;
  }
  __syncthreads();
  const ScaleState sc = *a.scale;
  const double sS = f.scale[0], sX = f.scale[1], sZ = f.scale[2];
  double acc[KMAX + 1];
#pragma unroll
  for (int k = 0; k <= KMAX; ++k) acc[k] = 0.0;
  double mX = 0.0, mZ = 0.0;     // sums of absolute contributions (X'e unscaled, Z'e scaled)
  int bad = 0;
  const int64_t n = a.n;
  const int zf = f.zFixed;
  const bool fixedRows = zf >= 0 && zf <= S_ZMAX;
  // what one observation needs from memory; two in flight per thread (the loads of observation i + stride are requested before
  // observation i is processed)
  struct Obs { double x[KMAX]; double e0, y, off, R, lat, uo, wt; double zw[S_ZMAX]; int zv[S_ZMAX]; int r0, r1; };
  auto fetch = [&](int64_t i, Obs& o) {
    if (i >= n) return;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) o.x[k] = k < K ? s.X[(size_t)k * n + i] : 0.0;
    if (DIRECT) o.e0 = s.e0[i];
    else {
      o.y = a.y[i];
      if (f.mode != 0 || f.wantTrain) { o.R = a.R[i]; o.off = a.off[i]; if (a.binary) o.lat = a.lat[i]; }
      if (f.mode >= 2) o.uo = a.userOffset[i];
    }
    if (a.wts) o.wt = a.wts[i];
    if (q) {
      if (fixedRows) {
        const size_t b = (size_t)i * (size_t)zf;
#pragma unroll
        for (int z = 0; z < S_ZMAX; ++z) if (z < zf) { o.zw[z] = s.w[b + z]; o.zv[z] = s.v[b + z]; }
      } else { o.r0 = s.u[i]; o.r1 = s.u[i + 1]; }
    }
  };
  auto use = [&](int64_t i, const Obs& o) {
    double e;
    if (DIRECT) {
      double eta = 0.0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) if (k < K) eta += o.x[k] * par[k];
      if (q) {
        if (fixedRows) {
#pragma unroll
          for (int z = 0; z < S_ZMAX; ++z) if (z < zf) eta += o.zw[z] * par[K + o.zv[z]];
        } else for (int z = o.r0; z < o.r1; ++z) eta += s.w[z] * par[K + s.v[z]];
      }
      e = o.e0 - eta;
    } else {
      double fit = 0.0, resp = o.y;
      if (f.mode != 0 || f.wantTrain) {
        if (a.binary) { fit = o.lat - o.R; if (f.mode != 0) resp = o.lat + o.off; }
        else { const double yr = (o.y - o.off - sc.min) / sc.range - 0.5; fit = ((yr - o.R) + 0.5) * sc.range + sc.min; }
      }
      const double so = f.mode == 0 ? 0.0 : f.mode == 1 ? fit : f.mode == 2 ? o.uo : fit + o.uo;
      e = resp - so;
      if (f.wantTrain) s.train[i] = fit;
      s.e0[i] = e;
    }
    const double we = a.wts ? o.wt * e : e;
    acc[0] += we * e;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) if (k < K) { const double t = o.x[k] * we; acc[1 + k] += t; mX += fabs(t); }
    auto addZ = [&](double wz, int col) {
      const double c = (wz * we) * sZ;
      if (!(fabs(c) < S_FX_LIMIT)) { bad = 1; return; }
      mZ += fabs(c);
      const FxLimbs l = fx_split(c);
      unsigned long long* dst = zh + 2 * (size_t)col;
      atomicAdd(dst, (unsigned long long)l.hi); atomicAdd(dst + 1, (unsigned long long)l.lo);
    };
    if (q) {
      if (fixedRows) {
#pragma unroll
        for (int z = 0; z < S_ZMAX; ++z) if (z < zf) addZ(o.zw[z], o.zv[z]);
      } else for (int z = o.r0; z < o.r1; ++z) addZ(s.w[z], s.v[z]);
    }
  };
  {
    const int64_t stride = (int64_t)gridDim.x * SBLOCK;
    int64_t i = (int64_t)blockIdx.x * SBLOCK + threadIdx.x;
    Obs oa, ob;
    fetch(i, oa);
    for (; i < n; i += stride) {
      fetch(i + stride, ob);
      use(i, oa);
      oa = ob;
    }
  }
  // block reduction of |e|^2, X'e and the magnitudes in a fixed order, then the fixed-point hand-off
#pragma unroll
  for (int k = 0; k <= KMAX; ++k) { const double v = wave_sum(acc[k]); if (lane == 0) red[wv][k] = v; }
  { const double v = wave_sum(mX); if (lane == 0) red[wv][KMAX + 1] = v; }
  { const double v = wave_sum(mZ); if (lane == 0) red[wv][KMAX + 2] = v; }
  __syncthreads();
  unsigned long long* mine = f.acc + ((size_t)f.parity * S_COPIES + (blockIdx.x % S_COPIES)) * W;
  if ((int)threadIdx.x <= K) {
    const int k = threadIdx.x;
    const double v = (((red[0][k] + red[1][k]) + red[2][k]) + red[3][k]) * (k == 0 ? sS : sX);
    if (!(fabs(v) < S_FX_LIMIT)) bad = 1;
    else { const FxLimbs l = fx_split(v); atomicAdd(mine + 2 * k, (unsigned long long)l.hi); atomicAdd(mine + 2 * k + 1, (unsigned long long)l.lo); }
  }
  if ((int)threadIdx.x < 3) {   // magnitudes, units of 2^10, rounded up
    const int g = threadIdx.x;
    const int c = g == 0 ? 0 : KMAX + g;
    double m = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
    m = g == 0 ? fabs(m) * sS : (g == 1 ? m * sX : m);
    if (!(m < 4.0e18)) bad = 1;
    else {
      atomicAdd(mine + 2 * (size_t)M + g, (unsigned long long)(m * 0.0009765625) + 1ull);
      // binary exponent of the largest workgroup magnitude of the group (offset 4000; 0 = nothing but zeros went in): what the
      // host centres the scales on, and how it notices magnitudes that have sunk towards the resolution of the low limb
      if (m > 0.0) atomicMax(mine + 2 * (size_t)M + 3 + g, (unsigned long long)(ilogb(m) + 4000));
    }
  }
  for (int j = threadIdx.x; j < 2 * q; j += SBLOCK) {
    unsigned long long v = zhAll[j];
    for (int c = 1; c < nCopy; ++c) v += zhAll[(size_t)c * 2 * q + j];     // (integer adds: any order)
    if (v) atomicAdd(mine + 2 * (1 + K) + j, v);
  }
  if (bad) atomicOr(f.bad + f.parity, 1);
}

// folds the copies of the accumulators of one evaluation (exact integer adds; the exponent words take the maximum) and forwards the
// result to host memory the device can write, then publishes the sequence number the host polls (one small workgroup right behind
// k_stan_fused on the same stream: no copy command, no stream synchronisation).  hostOut: [fused_words] sums, [1] flag, [1] seq
__global__ __launch_bounds__(SBLOCK) void k_stan_forward(StanFusedArgs f, int M) {
  const size_t W = fused_words(M);
  const unsigned long long* src = f.acc + (size_t)f.parity * S_COPIES * W;
  for (size_t j = threadIdx.x; j < W; j += SBLOCK) {
    unsigned long long v = 0ull;
    const bool isMax = j >= 2 * (size_t)M + 3;
#pragma unroll 8
    for (int x = 0; x < S_COPIES; ++x) { const unsigned long long t = src[(size_t)x * W + j]; v = isMax ? (t > v ? t : v) : v + t; }
    f.hostOut[j] = v;
  }
  if (threadIdx.x == 0) f.hostOut[W] = (unsigned long long)f.bad[f.parity];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(f.hostOut + W + 1, (unsigned long long)f.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(BLOCK) void k_test_fits(BartArrays a, double* out) {
  const ScaleState sc = *a.scale;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < a.nTest; i += (int64_t)gridDim.x * BLOCK) {
    double f = 0.0;
    for (int t = 0; t < a.T; ++t) {
      const size_t o = (size_t)t * a.nc;
      int nd = 0, v = a.var[o];
      while (v >= 0) {
        const unsigned x = a.xbinTest[(size_t)v * a.nTestPad + (size_t)i];
        nd = (x <= (unsigned)a.cut[o + nd]) ? a.left[o + nd] : a.right[o + nd];
        v = a.var[o + nd];
      }
      f += a.mu[o + nd];
    }
    out[i] = a.binary ? f : (f + 0.5) * sc.range + sc.min;
  }
}

// The same for FEW test rows (BASELINE config 4: 747 counterfactual rows, 75 trees — three workgroups of k_test_fits walking 75 trees one after the other,
// 91 us of dependent loads per iteration): one thread per (row, tree), TF_ROWS rows per workgroup, a wave per row; the leaf values of a row meet in LDS
// and ONE lane adds them in the order of the trees — the sum k_test_fits forms, bit for bit.
constexpr int TF_ROWS = BLOCK / 64;
__global__ __launch_bounds__(BLOCK) void k_test_fits_few(BartArrays a, double* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tfSmem[];
  double* vals = (double*)tfSmem;      // [TF_ROWS][T]
  const ScaleState sc = *a.scale;
  const int lane = threadIdx.x & 63, rw = threadIdx.x >> 6;
  for (int64_t i0 = (int64_t)blockIdx.x * TF_ROWS; i0 < a.nTest; i0 += (int64_t)gridDim.x * TF_ROWS) {
    const int64_t i = i0 + rw;
    if (i < a.nTest) {
      for (int t = lane; t < a.T; t += 64) {
        const size_t o = (size_t)t * a.nc;
        int nd = 0, v = a.var[o];
        while (v >= 0) {
          const unsigned x = a.xbinTest[(size_t)v * a.nTestPad + (size_t)i];
          nd = (x <= (unsigned)a.cut[o + nd]) ? a.left[o + nd] : a.right[o + nd];
          v = a.var[o + nd];
        }
        vals[(size_t)rw * a.T + t] = a.mu[o + nd];
      }
    }
    __syncthreads();
    if (lane == 0 && i < a.nTest) {
      double f = 0.0;
      for (int t = 0; t < a.T; ++t) f += vals[(size_t)rw * a.T + t];
      out[i] = a.binary ? f : (f + 0.5) * sc.range + sc.min;
    }
    __syncthreads();
  }
}

// how often every predictor is used by a rule of the current trees (the reference's `varcount`, one row per kept draw): a thread per tree walks it from the
// root, as the host did on a download of all eight tree arrays per kept iteration.  One workgroup; integer adds: any order.
__global__ __launch_bounds__(BLOCK) void k_var_counts(BartArrays a, int32_t* out) {
  for (int j = threadIdx.x; j < a.P; j += BLOCK) out[j] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < a.T; t += BLOCK) {
    const TreeView tv = tree_view(a, t);
    int nd, k; Walker<TreeView> w(tv, 0);
    while (w.next(nd, k)) if (k == 1) atomicAdd(out + (int)tv.var.get(nd), 1);
  }
}

// stored-tree prediction (stan4bart_predictBART): one thread per (test row, kept draw)
__global__ __launch_bounds__(BLOCK) void k_predict(const uint16_t* xb, int64_t nT, const PackedNode* nodes, const int64_t* treeStart, int64_t S, int T,
                                                   const double* scale, int binary, double* out) {
  const int64_t total = nT * S;
  for (int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * BLOCK) {
    const int64_t k = idx / nT, i = idx - k * nT;
    double f = 0.0;
    for (int t = 0; t < T; ++t) {
      const PackedNode* base = nodes + treeStart[k * T + t];
      int nd = 0;
      PackedNode p = base[0];
      // (states are validated when they are loaded; the step cap is a second guard against a walk that never ends)
      for (int guard = 0; p.var >= 0 && guard < 32768; ++guard) { nd = (xb[(size_t)p.var * (size_t)nT + i] <= p.cut) ? p.left : p.right; p = base[nd]; }
      f += p.mu;
    }
    out[idx] = binary ? f : (f + 0.5) * scale[2 * k + 1] + scale[2 * k];
  }
}

// ------------------------------------------------------------------------------------------------
// device stream probe (measurement only): what HBM delivers to plain streaming kernels of this shape, to put the roofline
// fraction of the tree kernel next to a measured ceiling as well as the 8 TB/s specification
__global__ __launch_bounds__(BLOCK) void k_probe_read(const double2* __restrict__ x, int64_t n2, double* out) {
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n2; i += (int64_t)gridDim.x * BLOCK) { const double2 v = x[i]; acc += v.x + v.y; }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0 && acc == 123.456) out[0] = acc;   // keeps the loads alive
}
__global__ __launch_bounds__(BLOCK) void k_probe_update(double2* __restrict__ x, int64_t n2) {   // read 16 B + write 16 B, like R
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n2; i += (int64_t)gridDim.x * BLOCK) { double2 v = x[i]; v.x += 1.0; v.y -= 1.0; x[i] = v; }
}
static void stream_probe(int device, int64_t nDoubles, int reps, double out[4]) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) throw std::runtime_error("stan4bart_amd: no HIP device available");
  if (device < 0 || device >= count) throw std::runtime_error("stan4bart_amd: HIP device ordinal out of range");
  HIP_OK(hipSetDevice(device));
  if (nDoubles < 1024 || reps < 1) throw std::invalid_argument("stream probe: n >= 1024, reps >= 1");
  double2* x = nullptr; double* o = nullptr; hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
  const int64_t n2 = nDoubles / 2;
  HIP_OK(hipMalloc(&x, (size_t)n2 * 16)); HIP_OK(hipMalloc(&o, 64));
  HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipMemsetAsync(x, 0, (size_t)n2 * 16, st));
  auto timeit = [&](int which) {
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
      HIP_OK(hipEventRecord(e0, st));
      if (which == 0) hipLaunchKernelGGL(k_probe_read, dim3(2048), dim3(BLOCK), 0, st, x, n2, o);
      else hipLaunchKernelGGL(k_probe_update, dim3(2048), dim3(BLOCK), 0, st, x, n2);
      HIP_OK(hipEventRecord(e1, st)); HIP_OK(hipStreamSynchronize(st));
      float ms = 0; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0 && ms < best) best = ms;
    }
    return (double)best;
  };
  const double msR = timeit(0), msU = timeit(1);
  out[0] = (double)n2 * 16.0 / (msR * 1e-3) / 1e9;          // GB/s, read only
  out[1] = (double)n2 * 32.0 / (msU * 1e-3) / 1e9;          // GB/s, read + write in place
  out[2] = msR * 1e3; out[3] = msU * 1e3;
  (void)hipFree(x); (void)hipFree(o); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
}

// ------------------------------------------------------------------------------------------------
class DevHip {
 public:
  DevHip() {}
  static void probe_stream(int device, int64_t nDoubles, int reps, double out[4]) { stream_probe(device, nDoubles, reps, out); }
  // HIP's current device is per host thread: every entry point of the C-ABI binds the sampler's device first, so a sampler
  // may be driven from any thread (chains fitted concurrently, predict from another thread)
  void bind() { HIP_OK(hipSetDevice(device_)); }
  ~DevHip() {
    if (graphExec_) (void)hipGraphExecDestroy(graphExec_);
    if (graph_) (void)hipGraphDestroy(graph_);
    if (sweepStatus_) (void)hipHostFree(sweepStatus_);
    for (void* p : allocs_) (void)hipFree(p);
    if (pinned_) (void)hipHostFree(pinned_);
    if (testPinned_) (void)hipHostFree(testPinned_);
    if (varPinned_) (void)hipHostFree(varPinned_);
    if (pinnedAcc_) (void)hipHostFree(pinnedAcc_);
    if (kOut_) (void)hipHostFree(kOut_);
    if (stream_) (void)hipStreamDestroy(stream_);
    if (evStart_) { (void)hipEventDestroy(evStart_); (void)hipEventDestroy(evStop_); }
  }
  DevHip(const DevHip&) = delete;
  // ---- k hyperprior: k is redrawn after every sweep (k_draw_k) and the host waits for it — the leaf prior precision of the next sweep
  // travels in the kernel arguments (no graph replay then: its kernel nodes would hold the precision of the sweep that was captured)
  void set_k_hyper(double df, double scale, double nodeScale, double k0) {
    kModeled_ = true; kh_.df = df; kh_.invScale2 = std::isinf(scale) ? 0.0 : 1.0 / (scale * scale); kh_.nodeScale = nodeScale; kCur_ = k0;
    useGraph_ = false;
    if (!kOut_) { HIP_OK(hipHostMalloc(&kOut_, 64, hipHostMallocCoherent | hipHostMallocMapped)); void* dp = nullptr; HIP_OK(hipHostGetDevicePointer(&dp, kOut_, 0)); kOutDev_ = (double*)dp; }
  }
  double k_current() const { return kCur_; }
  void set_k(double k) { kCur_ = k; a_.model.leafPrec = leaf_precision(k, T_, kh_.nodeScale); }
  void draw_k() {
    hipLaunchKernelGGL(k_draw_k, dim3(1), dim3(256), 0, stream_, a_, kh_, kCur_, kOutDev_); ++launches_;
    sync();
    set_k(((volatile double*)kOut_)[0]);
  }
  bool kModeled_ = false; KHyper kh_{0, 0, 1}; double kCur_ = 2.0; double* kOut_ = nullptr; double* kOutDev_ = nullptr;

  // stored sampler: only a device, a stream and the predictor count (predict_stored needs nothing else)
  void init_stored(int device, int P) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
      throw std::runtime_error("stan4bart_amd: no HIP device available — the MI355X path has no CPU fallback");
    if (device < 0 || device >= count) throw std::runtime_error("stan4bart_amd: HIP device ordinal out of range");
    device_ = device;
    HIP_OK(hipSetDevice(device_));
    HIP_OK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    a_ = BartArrays{}; a_.P = P; P_ = P;
  }
  void init(const DevInit& d) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
      throw std::runtime_error("stan4bart_amd: no HIP device available — the MI355X path has no CPU fallback");
    if (d.device < 0 || d.device >= count) throw std::runtime_error("stan4bart_amd: HIP device ordinal out of range");
    device_ = d.device;
    HIP_OK(hipSetDevice(device_));
#ifdef S4B_TUNING
    // Environment switches of the TUNING build only (`make tuning`, libs4b_tuning.so): the release library's launch geometry — and
    // with it the summation order of the bin sums — depends on the problem alone, never on the caller's environment
    if (const char* g = getenv("S4B_GRAPH")) useGraph_ = atoi(g) != 0;
#endif
    HIP_OK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    HIP_OK(hipEventCreate(&evStart_)); HIP_OK(hipEventCreate(&evStop_));
    n_ = d.n; nTest_ = d.nTest; P_ = d.P; T_ = d.T; nc_ = d.nc; K_ = d.K; q_ = d.q;
    BartArrays& a = a_;
    a = BartArrays{};
    a.n = n_; a.npad = (n_ + 7) / 8 * 8; a.P = P_; a.T = T_; a.nc = nc_; a.nTest = nTest_; a.nTestPad = (nTest_ + 7) / 8 * 8;
    const int64_t nQuads = (n_ + 3) / 4;
    // fixed launch geometry (=> fixed reduction order): one quad per thread up to 1024 workgroups (4 per CU), more quads per
    // thread beyond; S4B_GRID overrides it (tuning experiments).  Measured on MI355X (profiles/r01_grid_sweep.txt): 1024
    // workgroups are best both at n = 1e6 (latency-bound: 11.5 us per launch) and at n = 1e7 (52 % of HBM peak)
    a.grid = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (nQuads + BLOCK - 1) / BLOCK));
#ifdef S4B_TUNING
    if (const char* g = getenv("S4B_GRID")) { int v = atoi(g); if (v >= 1 && v <= GRID_MAX) a.grid = v; }
#endif
    // the tree kernel keeps 16-bit per-thread bin counts: at most 255 quads per thread
    while ((nQuads + (int64_t)a.grid * BLOCK - 1) / ((int64_t)a.grid * BLOCK) > 255 && a.grid < GRID_MAX) a.grid *= 2;
    if ((nQuads + (int64_t)a.grid * BLOCK - 1) / ((int64_t)a.grid * BLOCK) > 255) throw std::invalid_argument("n is too large for one device (more than 255 quads per thread)");
    a.binCap = 2 * nc_; a.traceCap = d.traceCap;
    // ---- observation-length arrays
    uint16_t* xb = alloc<uint16_t>((size_t)P_ * a.npad);
    HIP_OK(hipMemsetAsync(xb, 0, (size_t)P_ * a.npad * 2, stream_));
    HIP_OK(hipMemcpy2DAsync(xb, (size_t)a.npad * 2, d.xbin, (size_t)n_ * 2, (size_t)n_ * 2, (size_t)P_, hipMemcpyHostToDevice, stream_));
    a.xbin = xb;
    if (nTest_) {
      uint16_t* xt = alloc<uint16_t>((size_t)P_ * a.nTestPad);
      HIP_OK(hipMemcpy2DAsync(xt, (size_t)a.nTestPad * 2, d.xbinTest, (size_t)nTest_ * 2, (size_t)nTest_ * 2, (size_t)P_, hipMemcpyHostToDevice, stream_));
      a.xbinTest = xt;
    }
    double* y = alloc<double>((size_t)a.npad); upload(y, d.y, (size_t)n_); a.y = y;
    a.R = alloc<double>((size_t)a.npad); HIP_OK(hipMemsetAsync(a.R, 0, (size_t)a.npad * 8, stream_));
    a.off = alloc<double>((size_t)a.npad); HIP_OK(hipMemsetAsync(a.off, 0, (size_t)a.npad * 8, stream_));
    a.offNew = alloc<double>((size_t)a.npad); HIP_OK(hipMemsetAsync(a.offNew, 0, (size_t)a.npad * 8, stream_));
    if (d.userOffset) { double* uo = alloc<double>((size_t)a.npad); upload(uo, d.userOffset, (size_t)n_); a.userOffset = uo; }
    a.binary = d.binary; binary_ = d.binary != 0;
    if (binary_) {
      a.lat = zalloc<double>((size_t)a.npad);
#ifdef S4B_TUNING
      const char* lv = getenv("S4B_LATENTS");   // 1: the serial-bookkeeping kernel (k_latents), default: tables + readlane chain
#else
      const char* lv = nullptr;
#endif
      if (!(lv && atoi(lv) == 1)) {
        latX_ = zalloc<double>((size_t)a.npad);
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_latents2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lat2_lds_bytes()));
      }
    }
    a.leaf = alloc<uint16_t>((size_t)T_ * a.npad); HIP_OK(hipMemsetAsync(a.leaf, 0, (size_t)T_ * a.npad * 2, stream_));
    // ---- trees
    const size_t m = (size_t)T_ * nc_;
    a.treeI16 = zalloc<int16_t>((size_t)TF_COUNT * m); a.treeI32 = zalloc<int32_t>((size_t)TI_COUNT * T_);
    a.var = a.treeI16 + TF_VAR * m; a.cut = (uint16_t*)(a.treeI16 + TF_CUT * m); a.left = a.treeI16 + TF_LEFT * m; a.right = a.treeI16 + TF_RIGHT * m;
    a.parent = a.treeI16 + TF_PARENT * m; a.cna = a.treeI16 + TF_NA * m; a.cdep = a.treeI16 + TF_DEP * m; a.cleaf = a.treeI16 + TF_LEAF * m;
    a.cpre = a.treeI16 + TF_PRE * m; a.cpost = a.treeI16 + TF_POST * m;
    a.hwm = a.treeI32 + TI_HWM * T_; a.cnl = a.treeI32 + TI_NL * T_; a.cni = a.treeI32 + TI_NI * T_; a.cg = a.treeI32 + TI_G * T_;
    a.cgn = a.treeI32 + TI_GN * T_; a.cvalid = a.treeI32 + TI_VALID * T_;
    a.mu = alloc<double>(m); a.cnt = alloc<int32_t>(m); a.clogpi = zalloc<double>((size_t)T_);
    {
      std::vector<int16_t> var(m, NODE_FREE), neg(m, -1); std::vector<int32_t> hwm((size_t)T_, 1);
      for (int t = 0; t < T_; ++t) var[(size_t)t * nc_] = NODE_LEAF;
      upload(a.var, var.data(), m); upload(a.left, neg.data(), m); upload(a.right, neg.data(), m); upload(a.parent, neg.data(), m);
      HIP_OK(hipMemsetAsync(a.cut, 0, m * 2, stream_)); HIP_OK(hipMemsetAsync(a.mu, 0, m * 8, stream_)); HIP_OK(hipMemsetAsync(a.cnt, 0, m * 4, stream_));
      upload(a.hwm, hwm.data(), (size_t)T_);
      HIP_OK(hipStreamSynchronize(stream_));
    }
    for (int s = 0; s < 2; ++s) {
      StepScratch& c = a.sc[s];
      c.slab = zalloc<int16_t>((size_t)SF_COUNT * nc_);
      c.pvar = c.slab + SF_VAR * nc_; c.pcut = (uint16_t*)(c.slab + SF_CUT * nc_); c.pleft = c.slab + SF_LEFT * nc_; c.pright = c.slab + SF_RIGHT * nc_;
      c.pparent = c.slab + SF_PARENT * nc_; c.pna = c.slab + SF_NA * nc_; c.pdep = c.slab + SF_DEP * nc_; c.binA = c.slab + SF_BINA * nc_; c.binB = c.slab + SF_BINB * nc_;
      c.list = zalloc<int16_t>(nc_); c.insub = zalloc<uint8_t>(nc_);
      c.muOld = zalloc<double>(nc_); c.head = zalloc<StepHeader>(1); c.prop = &c.head->pr; c.accepted = zalloc<int32_t>(1);
      c.snapMu = zalloc<double>(nc_); c.snapCnt = zalloc<int32_t>(nc_);
      c.work = zalloc<double>((size_t)14 * nc_);
    }
    a.partCnt = zalloc<double>((size_t)a.binCap * a.grid); a.partSum = zalloc<double>((size_t)a.binCap * a.grid);
    a.binCnt = zalloc<double>((size_t)a.binCap); a.binSum = zalloc<double>((size_t)a.binCap);
    if (d.weights) {
      double* wt = alloc<double>((size_t)a.npad); HIP_OK(hipMemsetAsync(wt, 0, (size_t)a.npad * 8, stream_)); upload(wt, d.weights, (size_t)n_); a.wts = wt;
      a.partWt = zalloc<double>((size_t)a.binCap * a.grid); a.binWt = zalloc<double>((size_t)a.binCap);
    }
    a.rngF = zalloc<MTState>(2); a.rng = a.rngF; a.scale = zalloc<ScaleState>(1);
    a.preDone = zalloc<int32_t>(2); a.ticket = zalloc<int32_t>(1);
    {   // fused path (one launch per tree update, dev_step.inc): one 512-thread workgroup per CU at most
      // gridF - 1 workgroups share the observations, the last one is the control workgroup (write-backs, proposals one launch ahead)
      a.gridF = 1 + (int)std::min<int64_t>(255, std::max<int64_t>(1, (nQuads + F_PT - 1) / F_PT));
#ifdef S4B_TUNING
      if (const char* g = getenv("S4B_GRIDF")) { int v = atoi(g); if (v >= 2 && v <= F_GRID_MAX) a.gridF = v; }
#endif
      const int64_t passThreads = (int64_t)(a.gridF - 1) * F_PT;
      const int64_t perThread = (nQuads + passThreads - 1) / passThreads;
      a.candStride = (int64_t)cand_bytes(nc_); a.candBase = zalloc<unsigned char>((size_t)4 * cand_bytes(nc_));   // k_step: 2 parities x 2 images
      ldsStep_ = step_lds_bytes(nc_, d.weights != nullptr);
      // automatic choice: the fused launch wins while a tree update is latency-bound; at large n the two-kernel path keeps
      // more waves streaming (4 per SIMD instead of 2)
      // (cgm(split.probs): the weighted predictor choice lives in the pointer-storage control code of k_control / k_step's tail and in the persistent sweep's
      // k_sweep_sp / k_sweep_few_sp; k_step's own wave-register code is compiled without it: no fused path of choice, k_step only finishes handed-over sweeps)
      const bool stepOk = perThread <= 255 && ldsStep_ + 24 * 1024 <= 160 * 1024;
      splitProbs_ = d.model.splitProbs != nullptr;
      fusedOk_ = stepOk && !splitProbs_;
      fusedAuto_ = perThread <= 8 && fusedOk_;
      a.partF = zalloc<double>((size_t)2 * 3 * a.binCap * a.gridF);
      // persistent sweep (dev_sweep.inc): ONE launch per sweep, the residual in registers, every workgroup deciding redundantly.
      // Needs every quad of a pass thread in registers (SW_PF of them, 4 pass waves per workgroup), all gridF workgroups resident at once (one per CU: they wait for
      // each other inside the launch) and no weights.
      {
        hipDeviceProp_t prop; HIP_OK(hipGetDeviceProperties(&prop, device_));
        // beyond SW_PF quads per pass thread the residual does not fit the registers.  The STREAMING variant of the same launch (k_sweep_stream:
        // the pass waves read and write the residual per tree, 22 B per observation and tree update like k_tree, everything else as k_sweep)
        // exists for every size but is only taken on request (choose_path); its workgroup counts must fit the 21-bit field of the exchange words
        // (observation weights: k_sweep_w keeps the workgroup's 4 096 weights in 32 KiB of LDS beside the tables; not together with split.probs,
        // not on the streaming variant)
        weighted_ = d.weights != nullptr;
        constexpr bool weightedSweepBuilt = !S4B_LINEAR && S4B_WAVERED;      // (the build variants with another reduction of the statistics have no k_sweep_w: weighted samplers take the per-tree kernels there)
        size_t staticLds = 40 * 1024;
        if (weighted_ && weightedSweepBuilt) { hipFuncAttributes fa; HIP_OK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_sweep_w))); staticLds = fa.sharedSizeBytes + 512; }      // (its static part: tables + 32 KiB of weights)
        const bool common = stepOk && !(weighted_ && (splitProbs_ || !weightedSweepBuilt)) && a.gridF >= 2 && a.gridF <= 256 && a.gridF <= prop.multiProcessorCount &&
                            sweep_lds_bytes() + staticLds <= 160 * 1024;
        if (weighted_) {      // the power of two that brings the largest weight into (0.5, 1]
          double mx = 0.0; for (int64_t i = 0; i < n_; ++i) mx = std::max(mx, d.weights[i]);
          int e = 0; if (mx > 0.0 && std::isfinite(mx)) (void)std::frexp(mx, &e);
          wScale_ = std::ldexp(1.0, -e); wUnscale_ = std::ldexp(1.0, e);
        }
        sweepRegsOk_ = common && nQuads <= (int64_t)(a.gridF - 1) * SW_PT * SW_PF;
        sweepStreamOk_ = common && !splitProbs_ && !weighted_ && (n_ + a.gridF - 2) / (a.gridF - 1) + 4 * SW_PT < (int64_t)1 << 21;
        sweepOk_ = sweepRegsOk_ || sweepStreamOk_;
        // at most 4096 observations: ONE workgroup holds them all and does the control duties too (no exchange: dev_sweep.inc "solo")
        sweepSolo_ = nQuads <= (int64_t)SW_PT * SW_PF;
#ifdef S4B_TUNING
        if (getenv("S4B_NOSOLO")) sweepSolo_ = false;
#endif
        // no pass thread owns all SW_PF quads (thread 0 of workgroup 0 owns the most): k_sweep_few
        sweepFew_ = nQuads <= (int64_t)(SW_PF - 1) * (sweepSolo_ ? 1 : a.gridF - 1) * SW_PT;
        if (sweepOk_) {
          xbuf_ = zalloc<unsigned long long>((size_t)2 * (XC_RING_WORDS + XC_ROLL_WORDS));   // two rings (+ the words of the roll call behind each): a launch uses one and clears the other for the next launch
          HIP_OK(hipHostMalloc(&sweepStatus_, 64, hipHostMallocCoherent | hipHostMallocMapped));
          sweepStatus_[0] = -1;
          { void* dp = nullptr; HIP_OK(hipHostGetDevicePointer(&dp, sweepStatus_, 0)); sweepStatusDev_ = (int32_t*)dp; }
          HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sweep_lds_bytes()));
          HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_stream), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sweep_lds_bytes()));
          HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_few), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sweep_lds_bytes()));
          HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_sp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sweep_lds_bytes()));
          HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_few_sp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sweep_lds_bytes()));
          HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_w), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sweep_lds_bytes()));
        }
      }
      choose_path();
    }
    int32_t* nc = alloc<int32_t>((size_t)P_); upload(nc, d.numCuts, (size_t)P_); a.numCuts = nc;
    a.trace = zalloc<StepRecord>((size_t)std::max(1, d.traceCap)); a.traceCount = zalloc<int32_t>(1); a.errFlag = zalloc<int32_t>(1);
    a.model = d.model; a.model.numCuts = nc; a.traceOn = 0; a.model.scratch = nullptr;
    if (d.model.splitProbs) { double* sp = alloc<double>((size_t)P_); upload(sp, d.model.splitProbs, (size_t)P_); a.model.splitProbs = sp; }
    { double* tb = alloc<double>(3 * S4B_MAX_DEPTH); upload(tb, d.model.pgDepth, (size_t)S4B_MAX_DEPTH); upload(tb + S4B_MAX_DEPTH, d.model.logPg, (size_t)S4B_MAX_DEPTH);
      upload(tb + 2 * S4B_MAX_DEPTH, d.model.log1mPg, (size_t)S4B_MAX_DEPTH);
      a.model.pgDepth = tb; a.model.logPg = tb + S4B_MAX_DEPTH; a.model.log1mPg = tb + 2 * S4B_MAX_DEPTH;
      double* li = alloc<double>((size_t)d.model.logIntLen); upload(li, d.model.logInt, (size_t)d.model.logIntLen); a.model.logInt = li; }
    // ---- Stan-side arrays
    StanArrays& s = s_;
    s = StanArrays{};
    s.K = K_; s.q = q_; s.nnz = d.nnz;
    if (K_) { double* X = alloc<double>((size_t)K_ * n_); upload(X, d.X, (size_t)K_ * n_); s.X = X; }
    { int32_t* u = alloc<int32_t>((size_t)n_ + 1);
      if (d.u) upload(u, d.u, (size_t)n_ + 1); else HIP_OK(hipMemsetAsync(u, 0, ((size_t)n_ + 1) * 4, stream_));
      s.u = u; }
    if (d.nnz) {
      double* w = alloc<double>((size_t)d.nnz); upload(w, d.w, (size_t)d.nnz); s.w = w;
      int32_t* v = alloc<int32_t>((size_t)d.nnz); upload(v, d.v, (size_t)d.nnz); s.v = v;
      build_csc(d);
    } else {
      std::vector<int32_t> ptr((size_t)q_ + 1, 0);
      int32_t* cp = alloc<int32_t>((size_t)q_ + 1); upload(cp, ptr.data(), (size_t)q_ + 1); s.colChunkPtr = cp;
      HIP_OK(hipStreamSynchronize(stream_));
    }
    s.params = zalloc<double>((size_t)std::max(1, K_ + q_));
    s.e0 = zalloc<double>((size_t)a.npad); s.e = zalloc<double>((size_t)a.npad); s.train = zalloc<double>((size_t)a.npad);
    s.part = zalloc<double>((size_t)(1 + K_) * a.grid);
    s.out = zalloc<double>((size_t)(1 + K_ + q_));
    s.mmPart = zalloc<double>((size_t)2 * a.grid);
    HIP_OK(hipHostMalloc(&pinned_, sizeof(double) * (size_t)(2 * (1 + K_ + q_) + 64), hipHostMallocDefault));
    {   // fused Stan sums: fixed-point accumulators (two parities, 32 copies), LDS histograms of Z'e
      const size_t M = (size_t)(1 + K_ + q_);
      fusedLds_ = (size_t)q_ * 16 * ((size_t)q_ * 16 * (SBLOCK / 64) <= 32768 ? SBLOCK / 64 : 1) + (size_t)(K_ + q_) * 8 + 16;
      stanFused_ = K_ <= 16 && q_ <= S_QMAX;
#ifdef S4B_TUNING
      if (const char* f = getenv("S4B_STAN_FUSED")) stanFused_ = stanFused_ && atoi(f) != 0;
#endif
      if (stanFused_) {
        fusedAcc_ = zalloc<unsigned long long>((size_t)2 * S_COPIES * fused_words((int)M)); fusedBad_ = zalloc<int32_t>(2);
        // (coherent: the device's stores reach host memory when they are released, not at the end of the kernel — the host polls this buffer)
        HIP_OK(hipHostMalloc(&pinnedAcc_, sizeof(unsigned long long) * (fused_words((int)M) + 8), hipHostMallocCoherent | hipHostMallocMapped));
        std::memset(pinnedAcc_, 0, sizeof(unsigned long long) * (fused_words((int)M) + 8));
        // every row of Z with the same number of non-zeros (the usual case: one per grouping-term coefficient): no row pointers needed
        zFixed_ = -1;
        if (q_ && d.u && n_ > 0) {
          const int64_t z = d.u[1] - d.u[0];
          bool same = z >= 0 && z <= S_ZMAX;
          for (int64_t i = 0; same && i <= n_; ++i) same = (int64_t)d.u[i] == z * i;
          if (same) zFixed_ = (int)z;
        }
        if (fusedLds_ > 48 * 1024) {
          const void* fns[8] = {reinterpret_cast<const void*>(k_stan_fused<2, true>), reinterpret_cast<const void*>(k_stan_fused<2, false>),
                                reinterpret_cast<const void*>(k_stan_fused<4, true>), reinterpret_cast<const void*>(k_stan_fused<4, false>),
                                reinterpret_cast<const void*>(k_stan_fused<8, true>), reinterpret_cast<const void*>(k_stan_fused<8, false>),
                                reinterpret_cast<const void*>(k_stan_fused<16, true>), reinterpret_cast<const void*>(k_stan_fused<16, false>)};
          for (const void* fn : fns) HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fusedLds_));
        }
      }
    }
    varDev_ = zalloc<int32_t>((size_t)P_); HIP_OK(hipHostMalloc(&varPinned_, (size_t)P_ * 4, hipHostMallocDefault));
    if (nTest_) { testOut_ = zalloc<double>((size_t)nTest_); if (nTest_ <= (int64_t)1 << 22) HIP_OK(hipHostMalloc(&testPinned_, (size_t)nTest_ * 8, hipHostMallocDefault)); }
    // ---- launch configuration
    gridN_ = a.grid;   // one launch geometry for every O(N) kernel: the partial buffers are sized by it
#ifdef S4B_CONTROL_TIMING
    { unsigned long long z[32] = {0}; HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_step), z, sizeof(z))); }
#endif
    ldsApply_ = apply_lds_bytes(nc_); ldsTree_ = tree_lds_bytes(nc_); ldsControl_ = control_lds_bytes(P_, d.model.logIntLen);
    if (ldsTree_ > 64 * 1024) {
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tree<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsTree_));
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tree<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsTree_));
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tree<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsTree_));
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tree<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsTree_));
    }
    if (ldsApply_ > 64 * 1024) HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_apply), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsApply_));
    if ((fusedOk_ || sweepOk_) && ldsStep_ > 32 * 1024) {
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_step<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsStep_));
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_step<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsStep_));
    }
    if (ldsTree_ > 160 * 1024) throw std::runtime_error("node_capacity too large for the 160 KiB LDS of a CU");
    // ---- initial scale from the raw response (offset 0), R = yRescaled
    if (binary_) { hipLaunchKernelGGL(k_init_binary, dim3(gridN_), dim3(BLOCK), 0, stream_, a_); ++launches_; }
    else {
      hipLaunchKernelGGL(k_param_mean, dim3(gridN_), dim3(BLOCK), 0, stream_, a_, s_, 0, 0, 0, 1, a_.offNew); ++launches_;
      hipLaunchKernelGGL(k_scale, dim3(1), dim3(BLOCK), 0, stream_, a_, s_, 1, gridN_); ++launches_;
      HIP_OK(hipMemsetAsync(a_.mu, 0, (size_t)T_ * nc_ * 8, stream_));   // the very first scale has no tree fits to carry over
      hipLaunchKernelGGL(k_init_residual, dim3(gridN_), dim3(BLOCK), 0, stream_, a_); ++launches_;
    }
    sync();
  }

  // ---- small transfers
  void upload_rng(const MTState& s) { HIP_OK(hipMemcpyAsync(a_.rng, &s, sizeof(MTState), hipMemcpyHostToDevice, stream_)); sync(); }
  void download_rng(MTState& s) { HIP_OK(hipMemcpyAsync(&s, a_.rng, sizeof(MTState), hipMemcpyDeviceToHost, stream_)); sync(); }
  void upload_trees(const int16_t* var, const uint16_t* cut, const int16_t* left, const int16_t* right, const int16_t* parent, const double* mu, const int32_t* hwm) {
    const size_t m = (size_t)T_ * nc_;
    upload(a_.var, var, m); upload(a_.cut, cut, m); upload(a_.left, left, m); upload(a_.right, right, m); upload(a_.parent, parent, m);
    upload(a_.mu, mu, m); upload(a_.hwm, hwm, (size_t)T_);
    HIP_OK(hipMemsetAsync(a_.cvalid, 0, (size_t)T_ * 4, stream_));   // new trees: structure caches are stale
    sync();
  }
  void download_trees(int16_t* var, uint16_t* cut, int16_t* left, int16_t* right, int16_t* parent, double* mu, int32_t* cnt, int32_t* hwm) {
    const size_t m = (size_t)T_ * nc_;
    download(var, a_.var, m); download(cut, a_.cut, m); download(left, a_.left, m); download(right, a_.right, m); download(parent, a_.parent, m);
    download(mu, a_.mu, m); download(cnt, a_.cnt, m); download(hwm, a_.hwm, (size_t)T_);
    sync();
  }
  void download_leaf_plane(int t, uint16_t* out) { download(out, a_.leaf + (size_t)t * a_.npad, (size_t)n_); sync(); }
  void get_scale(ScaleState& s) { flush_hand_off(); download(&s, a_.scale, 1); sync(); }
  int32_t error_flags() { int32_t e = 0; download(&e, a_.errFlag, 1); sync(); return e; }
  int64_t launches() const { return launches_; }
  void set_trace(bool on) { a_.traceOn = on ? 1 : 0; HIP_OK(hipMemsetAsync(a_.traceCount, 0, 4, stream_)); sync(); }
  int64_t get_trace(int64_t cap, int32_t* out) {
    int32_t m = 0; download(&m, a_.traceCount, 1); sync();
    std::vector<StepRecord> tmp((size_t)m);
    if (m) { download(tmp.data(), a_.trace, (size_t)m); sync(); }
    for (int64_t i = 0; i < m && i < cap; ++i) std::memcpy(out + 5 * i, &tmp[(size_t)i], 20);
    HIP_OK(hipMemsetAsync(a_.traceCount, 0, 4, stream_)); sync();
    return m;
  }

  // ---- offsets / scale
  void offset_from_host(const double* off) {
    upload(a_.offNew, off, (size_t)n_);
    hipLaunchKernelGGL(k_param_mean, dim3(gridN_), dim3(BLOCK), 0, stream_, a_, s_, 0, 0, 0, 1, a_.offNew); ++launches_;
  }
  // offset_from_params + set_sigma + rescale(false) of one Gibbs iteration are ONE launch (k_offset_rescale) when the coefficients fit
  // the kernel arguments: the first two calls are held until rescale() knows whether the scale is updated
  void offset_from_params(const double* beta, const double* b, int fixed, int random, int addUser) {
    flush_hand_off();
    if (!binary_ && K_ + q_ <= 64) {
      pend_.fixed = fixed; pend_.random = random; pend_.addUser = addUser; pend_.pad = 0; pend_.sigmaData = 0.0;
      for (int k = 0; k < K_; ++k) pend_.par[k] = beta[k];
      for (int j = 0; j < q_; ++j) pend_.par[K_ + j] = b[j];
      for (int j = K_ + q_; j < 64; ++j) pend_.par[j] = 0.0;
      pendOffset_ = true; pendSigma_ = false;
      return;
    }
    push_params(beta, b);
    hipLaunchKernelGGL(k_param_mean, dim3(gridN_), dim3(BLOCK), 0, stream_, a_, s_, fixed, random, addUser, 0, a_.offNew); ++launches_;
  }
  void flush_hand_off() {    // the held calls as their own kernels
    if (!pendOffset_) return;
    pendOffset_ = false;
    push_params(pend_.par, pend_.par + K_);
    hipLaunchKernelGGL(k_param_mean, dim3(gridN_), dim3(BLOCK), 0, stream_, a_, s_, pend_.fixed, pend_.random, pend_.addUser, 0, a_.offNew); ++launches_;
    if (pendSigma_) { pendSigma_ = false; hipLaunchKernelGGL(k_set_sigma, dim3(1), dim3(1), 0, stream_, a_, pend_.sigmaData); ++launches_; }
  }
  void param_mean_to_host(const double* beta, const double* b, double* out) {
    push_params(beta, b);
    hipLaunchKernelGGL(k_param_mean, dim3(gridN_), dim3(BLOCK), 0, stream_, a_, s_, 1, 1, 0, 0, s_.e); ++launches_;
    download(out, s_.e, (size_t)n_); sync();
  }
  void set_sigma(double s) {
    if (pendOffset_) { pend_.sigmaData = s; pendSigma_ = true; return; }
    hipLaunchKernelGGL(k_set_sigma, dim3(1), dim3(1), 0, stream_, a_, s); ++launches_;
  }
  void rescale(bool update) {
    if (pendOffset_ && pendSigma_ && !update) {
      pendOffset_ = false; pendSigma_ = false;
      hipLaunchKernelGGL(k_offset_rescale, dim3(gridN_), dim3(BLOCK), 0, stream_, a_, s_, pend_); ++launches_;
      std::swap(a_.off, a_.offNew);
      return;
    }
    flush_hand_off();
    if (binary_) {
      hipLaunchKernelGGL(k_rescale_binary, dim3(gridN_), dim3(BLOCK), 0, stream_, a_); ++launches_;
      std::swap(a_.off, a_.offNew);
      return;
    }
    hipLaunchKernelGGL(k_scale, dim3(1), dim3(BLOCK), 0, stream_, a_, s_, update ? 1 : 0, gridN_); ++launches_;
    hipLaunchKernelGGL(k_rescale, dim3(gridN_), dim3(BLOCK), 0, stream_, a_, update ? 1 : 0); ++launches_;
    std::swap(a_.off, a_.offNew);
  }

  // ---- trees
  void assign_leaves_and_residual() { hipLaunchKernelGGL(k_assign_leaves, dim3(gridN_), dim3(BLOCK), 0, stream_, a_, 1); ++launches_; }
  void assign_leaves_only() { hipLaunchKernelGGL(k_assign_leaves, dim3(gridN_), dim3(BLOCK), 0, stream_, a_, 0); ++launches_; }
  // ---- state injection / extraction (s4b_set_state / s4b_get_state)
  double* obs_array(int which) {
    switch (which) {
      case OBS_R: return a_.R;
      case OBS_OFF: return a_.off;
      case OBS_LAT: if (!a_.lat) throw std::invalid_argument("latents exist only for binary responses"); return a_.lat;
      case OBS_Y: return const_cast<double*>(a_.y);
    }
    throw std::invalid_argument("unknown observation array");
  }
  void download_obs(int which, double* out) { download(out, obs_array(which), (size_t)n_); sync(); }
  void upload_obs(int which, const double* in) {
    if (which == OBS_Y) throw std::invalid_argument("the response is fixed at creation");
    upload(obs_array(which), in, (size_t)n_); sync();
  }
  void upload_counts(const int32_t* cnt) { upload(a_.cnt, cnt, (size_t)T_ * nc_); sync(); }
  void set_scale(double mn, double mx, double range, double sigmaData) {
    ScaleState sc; sc.min = mn; sc.max = mx; sc.range = range; sc.min0 = mn; sc.range0 = range; sc.shiftPerTree = 0.0; sc.sigmaData = sigmaData; sc.sigma = sigmaData / range;
    upload(a_.scale, &sc, 1); sync();
  }
  // One sweep = 2 T + 2 launches with arguments that never change (per-tree state is reached through pointers), so
  // it is captured once into a hipGraph and replayed: the host then costs one call per sweep instead of ~4 us per launch.
  void sweep(int thin) {
#ifdef S4B_TUNING
    static const bool dbg = getenv("S4B_HOST_TIMING") != nullptr;
#else
    constexpr bool dbg = false;
#endif
    if (dbg) HIP_OK(hipEventRecord(evStart_, stream_));
    sweep_impl(thin);
    if (dbg) {
      HIP_OK(hipEventRecord(evStop_, stream_)); sync();
      float ms = 0; HIP_OK(hipEventElapsedTime(&ms, evStart_, evStop_)); dbgSweepMs_ += ms; ++dbgSweeps_;
      if (dbgSweeps_ % 20 == 0) fprintf(stderr, "S4B in-loop sweep GPU time: %.3f ms avg over %d\n", dbgSweepMs_ / dbgSweeps_, dbgSweeps_);
    }
  }
  // persistent path: one k_sweep launch per sweep; the status word (host-visible) says how far it got: T + 1 = the whole sweep,
  // t in 1..T = k_step launches t..T finish it (a tree outgrew the wave-register control path), 0 = nothing done (tree 0 did)
  // A persistent sweep needs every workgroup of its launch resident: two such launches interleaved on one device would wait for each
  // other's compute units.  Samplers of ONE process (threads: stan4bart(cores > 1) without the sharing hint, a C caller's own threads)
  // therefore take turns — the lock is held from the launch until the host has seen the launch end.  Other processes cannot be
  // kept out this way (s4b_set_device_sharing; the bounded waits turn such a collision into an error, not a hang).
  static std::mutex& sweep_mutex(int device) { static std::mutex m[64]; return m[device & 63]; }
  std::unique_lock<std::mutex> sweepLock_;
  // (scope guard: whatever ends a block that launched a persistent sweep — its normal end or an exception on the way — gives the device's turn back)
  struct TurnGuard { DevHip& d; explicit TurnGuard(DevHip& dd) : d(dd) {} ~TurnGuard() { if (d.sweepLock_.owns_lock()) d.sweepLock_.unlock(); }
                     TurnGuard(const TurnGuard&) = delete; TurnGuard& operator=(const TurnGuard&) = delete; };
  // The device was busy at the last persistent launch (its roll call failed: another process's kernels held compute units): the next
  // sweepBackoff_ sweeps run as k_step launches, then the persistent launch is tried again; every further failure doubles the pause.
  int64_t sweepBusyUntil_ = 0, sweepBusy_ = 0; int sweepBackoff_ = 16;
  bool persistent_now() const { return sweepCount_ >= sweepBusyUntil_; }
  // TEST HOOK (s4b_set_test_hook 1): every k-th persistent launch finds the decision word of its roll call already at BUSY — what a launch sees
  // that shares the device with somebody else's kernels — so that the busy fallback (the sweep rerun as per-tree launches, the back-off, the Stan
  // inputs formed again) runs under the parity tests without a second process.  0 = off.
  int64_t hookBusyEvery_ = 0, hookLaunchNo_ = 0;
  void set_test_hook(int hook, int64_t value) {
    if (hook != 1 || value < 0) throw std::invalid_argument("set_test_hook: hook 1 (every k-th persistent launch reports a busy device; 0 = off) is the only one");
    hookBusyEvery_ = value; hookLaunchNo_ = 0;
  }
  void hook_before_persistent_launch() {
    if (hookBusyEvery_ <= 0 || sweepGrid_ == 1) return;        // (a one-workgroup launch holds no roll call)
    if (++hookLaunchNo_ % hookBusyEvery_ != 0) return;
    unsigned long long* cur = xbuf_ + (size_t)xbufParity_ * (XC_RING_WORDS + XC_ROLL_WORDS);      // (the ring sweep_args() hands to the launch that follows)
    static const unsigned long long busy = SW_ROLL_BUSY;
    HIP_OK(hipMemcpyAsync(cur + XC_RING_WORDS + 1, &busy, 8, hipMemcpyHostToDevice, stream_));
  }
  void sweep_persistent_launch() {
    if (sweepLock_.owns_lock()) sweepLock_.unlock();      // (a previous launch whose end an exception kept us from seeing)
    sweepLock_ = std::unique_lock<std::mutex>(sweep_mutex(device_));
    for (int i = 0; i < 16; ++i) sweepStatus_[i] = 0;
    sweepStatus_[0] = -1;
    hook_before_persistent_launch();
    launch_sweep_kernel();
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sweepLock_.unlock(); HIP_OK(e); }
  }
  void launch_sweep_kernel() {
    if (sweepStream_) hipLaunchKernelGGL(k_sweep_stream, dim3(sweepGrid_), dim3(FBLOCK), sweep_lds_bytes(), stream_, a_, sweep_args());
    else if (weighted_) hipLaunchKernelGGL(k_sweep_w, dim3(sweepGrid_), dim3(FBLOCK), sweep_lds_bytes(), stream_, a_, sweep_args());
    else if (splitProbs_) hipLaunchKernelGGL(sweepFew_ ? k_sweep_few_sp : k_sweep_sp, dim3(sweepGrid_), dim3(FBLOCK), sweep_lds_bytes(), stream_, a_, sweep_args());
    else if (sweepFew_) hipLaunchKernelGGL(k_sweep_few, dim3(sweepGrid_), dim3(FBLOCK), sweep_lds_bytes(), stream_, a_, sweep_args());
    else hipLaunchKernelGGL(k_sweep, dim3(sweepGrid_), dim3(FBLOCK), sweep_lds_bytes(), stream_, a_, sweep_args());
    ++launches_;
  }
  bool sweep_streams() const { return sweepStream_; }
  SweepArgs sweep_args() {     // (the exchange ring of this launch, the one it clears for the next launch)
    unsigned long long* cur = xbuf_ + (size_t)xbufParity_ * (XC_RING_WORDS + XC_ROLL_WORDS);
    unsigned long long* nxt = xbuf_ + (size_t)(1 - xbufParity_) * (XC_RING_WORDS + XC_ROLL_WORDS);
    xbufParity_ ^= 1;
    return SweepArgs{cur, sweepStatusDev_, nxt, wScale_, wUnscale_};
  }
  // the launch has ended (the caller waited for it or for something behind it on the stream): true = the sweep is complete, false = the
  // rest of it (or all of it) has just been queued as k_step launches
  bool sweep_persistent_finish() {
    if (sweepLock_.owns_lock()) sweepLock_.unlock();
    const int st = sweepStatus_[0];
    ++sweepCount_;
    count_persistent_launch(st);
    if (st == T_ + 1) { if (sweepBackoff_ > 16 && sweepCount_ > sweepBusyUntil_ + 64) sweepBackoff_ = 16; return true; }
    if (st == -2) {       // roll call failed: the device is shared right now; nothing of the chain was touched
      ++sweepBusy_;
      sweepBusyUntil_ = sweepCount_ + sweepBackoff_;
      sweepBackoff_ = std::min(sweepBackoff_ * 2, 4096);
      sweep_fused_or_two_one();
      return false;
    }
    if (st < 0 || st > T_ + 1) {
      int32_t e = 0; (void)hipMemcpy(&e, a_.errFlag, 4, hipMemcpyDeviceToHost);
      throw std::runtime_error("persistent tree sweep: the launch did not complete (a workgroup timed out waiting for the others); status " +
                               std::to_string(st) + ", device error word " + std::to_string(e) + ", first wait that gave up: dev_sweep.inc:" + std::to_string(sweepStatus_[1])
#ifdef S4B_TUNING
                               + " | per workgroup (needBig << 24 | paGo << 8 | bailStep): " + std::to_string(sweepStatus_[2]) + " " + std::to_string(sweepStatus_[3]) + " " + std::to_string(sweepStatus_[4]) + " grid " + std::to_string(a_.gridF) + " tickets " + std::to_string(sweepStatus_[8]) + " " + std::to_string(sweepStatus_[9]) + " tail stage " + std::to_string(sweepStatus_[12])
#endif
                               );
    }
    ++sweepHandOvers_;
    if (st == 0) { if (splitProbs_) sweep_two_one(); else sweep_fused_one(); return false; }
    for (int t = st; t <= T_; ++t) { launch_step(t); ++launches_; }
    if (((T_ + 1) & 1) == 1) HIP_OK(hipMemcpyAsync(a_.rngF, a_.rngF + 1, sizeof(MTState), hipMemcpyDeviceToDevice, stream_));
    return false;
  }
  // The sweep of a persistent launch that found the device shared (and the sweeps of the back-off after it).  With three or more chains on the device
  // (s4b_set_device_sharing) the two-kernel tree update: the fused launch keeps every compute unit busy with one workgroup of 8 register-heavy
  // waves and leaves no room for the other chains' kernels (measured at n = 1e6, 4 chains: 512 against 425 iterations/s in aggregate) — the same
  // rule choose_path applies where the persistent sweep does not exist.  Both start from and leave the state every path shares between sweeps.
  void sweep_fused_or_two_one() { if (sharing_ >= 3 || splitProbs_) sweep_two_one(); else sweep_fused_one(); }
  void sweep_two_one() {
    hipLaunchKernelGGL(k_control, dim3(1), dim3(CBLOCK), 0, stream_, a_, -1, 0); ++launches_;
    for (int t = 0; t < T_; ++t) {
      launch_tree(t);
      hipLaunchKernelGGL(k_control, dim3(1), dim3(CBLOCK), 0, stream_, a_, t, t + 1 < T_ ? t + 1 : -1);
      launches_ += 2;
    }
    hipLaunchKernelGGL(k_apply, dim3(a_.grid), dim3(BLOCK), ldsApply_, stream_, a_, T_ - 1); ++launches_;
  }
  void sweep_persistent_one() {
    if (!persistent_now()) { ++sweepCount_; sweep_fused_or_two_one(); return; }
    TurnGuard turn(*this);
    sweep_persistent_launch();
    HIP_OK(hipStreamSynchronize(stream_));
    sweep_persistent_finish();
  }
  // The Gibbs iteration's sweep followed by the Stan block's inputs (sampler_core.hpp run()).  On the persistent path the host does not
  // wait for the sweep's status word before queueing the Stan kernels: it waits once, for their result, and looks at the status then.
  // A sweep that was not finished by the launch (a tree outgrew the wave-register control path: rare; the device was shared: rarer) is
  // finished with k_step launches and the Stan inputs are formed again — they only read the BART state and overwrite their own outputs;
  // what the discarded first evaluation did to the host-side state of the fixed-point sums (scales, counters) is put back first.
  void sweep_and_stan_inputs(int thin, int mode, bool wantTrain, double* cX, double* cZ, double* s0, double* trainOut) {
    if (!is_persistent() || binary_ || thin < 1 || !persistent_now() || kModeled_) { sweep(thin); stan_inputs(mode, wantTrain, cX, cZ, s0, trainOut); return; }
    for (int k = 0; k + 1 < thin; ++k) sweep_persistent_one();
    if (!persistent_now()) { ++sweepCount_; sweep_fused_or_two_one(); stan_inputs(mode, wantTrain, cX, cZ, s0, trainOut); return; }
    TurnGuard turn(*this);
    sweep_persistent_launch();
    const FxHostState saved = fx_host_state();
    stan_inputs(mode, wantTrain, cX, cZ, s0, trainOut);
    if (!sweep_persistent_finish()) { set_fx_host_state(saved); stan_inputs(mode, wantTrain, cX, cZ, s0, trainOut); }
  }
  struct FxHostState { int exp[3]; int64_t lastBad, evals, fallbacks; bool tiny; };
  FxHostState fx_host_state() const { FxHostState f; for (int g = 0; g < 3; ++g) f.exp[g] = fxExp_[g]; f.lastBad = fxLastBad_; f.evals = fusedEvals_; f.fallbacks = fusedFallbacks_; f.tiny = fxTinyFail_; return f; }
  void set_fx_host_state(const FxHostState& f) { for (int g = 0; g < 3; ++g) fxExp_[g] = f.exp[g]; fxLastBad_ = f.lastBad; fusedEvals_ = f.evals; fusedFallbacks_ = f.fallbacks; fxTinyFail_ = f.tiny; }
  void sweep_impl(int thin) {
    flush_hand_off();
    if (kModeled_) {      // one sweep at a time: trees, latents, k — then the host knows the precision the next sweep's launches carry
      for (int k = 0; k < thin; ++k) {
        if (is_persistent()) sweep_persistent_one(); else sweep_eager(1, false);
        if (binary_) launch_latents();
        draw_k();
      }
      return;
    }
    if (is_persistent()) {
      for (int k = 0; k < thin; ++k) { sweep_persistent_one(); if (binary_) launch_latents(); }
      return;
    }
    if (useGraph_) {
      if (!graphExec_ || graphTrace_ != a_.traceOn) capture_sweep();
      for (int k = 0; k < thin; ++k) {
        HIP_OK(hipGraphLaunch(graphExec_, stream_)); launches_ += useFused_ ? T_ + 2 : 2 * T_ + 2;
        if (binary_) launch_latents();
      }
      return;
    }
    sweep_eager(thin, true);
  }
  void capture_sweep() {
    if (graphExec_) { (void)hipGraphExecDestroy(graphExec_); graphExec_ = nullptr; }
    if (graph_) { (void)hipGraphDestroy(graph_); graph_ = nullptr; }
    HIP_OK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
    const int64_t before = launches_;
    sweep_eager(1, false);
    launches_ = before;
    HIP_OK(hipStreamEndCapture(stream_, &graph_));
    HIP_OK(hipGraphInstantiate(&graphExec_, graph_, nullptr, nullptr, 0));
    graphTrace_ = a_.traceOn;
  }
  void launch_tree(int t) {
    if (a_.wts) {
      if (t == 0) hipLaunchKernelGGL((k_tree<false, true>), dim3(a_.grid), dim3(BLOCK), ldsTree_, stream_, a_, t);
      else hipLaunchKernelGGL((k_tree<true, true>), dim3(a_.grid), dim3(BLOCK), ldsTree_, stream_, a_, t);
    } else {
      if (t == 0) hipLaunchKernelGGL((k_tree<false, false>), dim3(a_.grid), dim3(BLOCK), ldsTree_, stream_, a_, t);
      else hipLaunchKernelGGL((k_tree<true, false>), dim3(a_.grid), dim3(BLOCK), ldsTree_, stream_, a_, t);
    }
  }
  void launch_latents() {
    if (latX_) {
      hipLaunchKernelGGL(k_latents2, dim3(1), dim3(LB2), lat2_lds_bytes(), stream_, a_, latX_);
      hipLaunchKernelGGL(k_latents_finish, dim3(gridN_), dim3(BLOCK), 0, stream_, a_, (const double*)latX_);
      launches_ += 2;
    } else { hipLaunchKernelGGL(k_latents, dim3(1), dim3(BLOCK), 0, stream_, a_); ++launches_; }
  }
  void launch_step(int t) {
    if (a_.wts) hipLaunchKernelGGL((k_step<true>), dim3(a_.gridF), dim3(FBLOCK), ldsStep_, stream_, a_, t);
    else hipLaunchKernelGGL((k_step<false>), dim3(a_.gridF), dim3(FBLOCK), ldsStep_, stream_, a_, t);
  }
  // fused path: T + 1 launches per sweep (launch t: decide tree t-1, propose tree t, one O(N) pass), preceded by the
  // one-workgroup launch that handles an oversized first tree
  void sweep_fused_one() {
    hipLaunchKernelGGL(k_step_pre, dim3(1), dim3(FBLOCK), 0, stream_, a_); ++launches_;
    for (int t = 0; t <= T_; ++t) {
      launch_step(t); ++launches_;
#ifdef S4B_TUNING
      if (getenv("S4B_DEBUG_STEPS")) {      // (development: which launch of the sweep does not come back)
        const auto t0 = std::chrono::steady_clock::now();
        while (hipStreamQuery(stream_) != hipSuccess)
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 3.0) {
            int32_t hw = 0; hipStream_t s2; (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
            (void)hipMemcpyFromSymbolAsync(&hw, HIP_SYMBOL(g_dbgMark), 4, 0, hipMemcpyDeviceToHost, s2); (void)hipStreamSynchronize(s2);
            { int sp[4] = {0, 0, 0, 0}; (void)hipMemcpyFromSymbolAsync(sp, HIP_SYMBOL(g_dbgSpin), 16, 0, hipMemcpyDeviceToHost, s2); (void)hipStreamSynchronize(s2);
              fprintf(stderr, "S4B_DEBUG_STEPS: a wave waits at source line %d: flag %d, target %d, block %d, thread %d\n", sp[0], sp[1] & 0xffff, sp[1] >> 16, sp[2], sp[3]); }
            { std::vector<int> pg(640 * 8); (void)hipMemcpyFromSymbolAsync(pg.data(), HIP_SYMBOL(g_prog), pg.size() * 4, 0, hipMemcpyDeviceToHost, s2); (void)hipStreamSynchronize(s2);
              int hist[16] = {0}; for (int b = 0; b < a_.gridF; ++b) for (int w = 0; w < 8; ++w) { const int v = pg[(size_t)b * 8 + w]; if ((v >> 8) == t) ++hist[v & 15]; else ++hist[0]; }
              fprintf(stderr, "S4B_DEBUG_STEPS: waves by last checkpoint of this launch (0 = none): "); for (int i = 0; i < 8; ++i) fprintf(stderr, "%d:%d ", i, hist[i]); fprintf(stderr, "\n");
              int shown = 0; for (int b = 0; b < a_.gridF && shown < 12; ++b) for (int w = 0; w < 8 && shown < 12; ++w) { const int v = pg[(size_t)b * 8 + w]; if ((v >> 8) != t || (v & 15) < 4) { fprintf(stderr, "  block %d wave %d: launch %d checkpoint %d\n", b, w, v >> 8, v & 15); ++shown; } } }
            fprintf(stderr, "S4B_DEBUG_STEPS: launch t = %d of sweep %lld did not finish within 3 s; error / marker word 0x%x\n", t, (long long)dbgSweepNo_, hw); fflush(stderr); std::quick_exit(3); }
      }
#endif
    }
#ifdef S4B_TUNING
    ++dbgSweepNo_;
#endif
    // the generator alternates between two slots by launch parity; between sweeps it lives in slot 0 (a_.rng)
    if (((T_ + 1) & 1) == 1) HIP_OK(hipMemcpyAsync(a_.rngF, a_.rngF + 1, sizeof(MTState), hipMemcpyDeviceToDevice, stream_));
  }
  void sweep_eager(int thin, bool withLatents) {
    if (useFused_) {
      for (int k = 0; k < thin; ++k) {
        sweep_fused_one();
        if (binary_ && withLatents) launch_latents();
      }
      return;
    }
    for (int k = 0; k < thin; ++k) {
      // per tree: one fused O(N) kernel (finish tree t-1, statistics of tree t) + one control kernel
      sweep_two_one();
      if (binary_ && withLatents) launch_latents();
    }
  }
  // per-launch HIP-event timing of extra sweeps on the sampler's stream (bench.py roofline leg)
  void profile_sweep_fused(int nSweeps, int thin, double* out) {
    std::vector<hipEvent_t> ev((size_t)2 * (T_ + 1));
    for (auto& e : ev) HIP_OK(hipEventCreate(&e));
    double sum = 0, cnt = 0, sumLast = 0, cntLast = 0;
    for (int sIdx = 0; sIdx < nSweeps * thin; ++sIdx) {
      hipLaunchKernelGGL(k_step_pre, dim3(1), dim3(FBLOCK), 0, stream_, a_); ++launches_;
      for (int t = 0; t <= T_; ++t) {
        HIP_OK(hipEventRecord(ev[(size_t)2 * t], stream_));
        launch_step(t); ++launches_;
        HIP_OK(hipEventRecord(ev[(size_t)2 * t + 1], stream_));
      }
      if (((T_ + 1) & 1) == 1) HIP_OK(hipMemcpyAsync(a_.rngF, a_.rngF + 1, sizeof(MTState), hipMemcpyDeviceToDevice, stream_));
      if (binary_) launch_latents();
      if (kModeled_) draw_k();      // (a profiled sweep is a sweep of the chain: trees, latents, k — sweep_impl's order)
      sync();
      for (int t = 0; t <= T_; ++t) {
        float ms = 0; HIP_OK(hipEventElapsedTime(&ms, ev[(size_t)2 * t], ev[(size_t)2 * t + 1]));
        if (t >= 1 && t < T_) { sum += ms * 1000.0; cnt += 1; }            // launches with both halves
        if (t == T_) { sumLast += ms * 1000.0; cntLast += 1; }
      }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
#ifdef S4B_CONTROL_TIMING
    { unsigned long long h[32]; HIP_OK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_step), sizeof(h)));
      const double k = h[0] ? 1.0 / (100.0 * (double)h[0]) : 0.0;
      { unsigned long long q[32]; HIP_OK(hipMemcpyFromSymbol(q, HIP_SYMBOL(g_prop), sizeof(q)));
        for (int ty = 0; ty < 4; ++ty) fprintf(stderr, "DBG propose type %d: %llu proposals (%llu without a valid move), %.2f us each\n", ty, q[20 + ty], q[24 + ty], q[20 + ty] ? (double)q[16 + ty] / (100.0 * (double)q[20 + ty]) : 0.0);
        const double c = q[15] ? 1.0 / (100.0 * (double)q[15]) : 0.0;
        // stamps 0/1 are taken by every proposal, 2..7 by births only: report birth-path deltas from stamp 2 on
        fprintf(stderr, "DBG propose (births: %llu of %llu): first draw %.2f; select node .. draw var %.2f, interval + split %.2f, build %.2f, child memo %.2f, ratios %.2f us\n", q[15], q[8], q[8] ? (double)(q[1] - q[0]) / (100.0 * (double)q[8]) : 0.0,
                (double)(q[3] - q[2]) * c, (double)(q[4] - q[3]) * c, (double)(q[5] - q[4]) * c, (double)(q[6] - q[5]) * c, (double)(q[7] - q[6]) * c);
        unsigned long long z16[32] = {0}; HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_prop), z16, sizeof(z16))); }
      { static unsigned long long ws[F_GRID_MAX], we[F_GRID_MAX];
        HIP_OK(hipMemcpyFromSymbol(ws, HIP_SYMBOL(g_wgS), sizeof(ws))); HIP_OK(hipMemcpyFromSymbol(we, HIP_SYMBOL(g_wgE), sizeof(we)));
        const int G = a_.gridF; unsigned long long s0 = ~0ull; for (int b = 0; b < G; ++b) if (ws[b] < s0) s0 = ws[b];
        double lastEnd = 0, lastStart = 0; int bEnd = 0, bStart = 0;
        for (int b = 0; b < G; ++b) { const double st = (double)(ws[b] - s0) * k, en = (double)(we[b] - s0) * k; if (en > lastEnd) { lastEnd = en; bEnd = b; } if (st > lastStart) { lastStart = st; bStart = b; } }
        fprintf(stderr, "DBG workgroups (average us after the earliest start): last start %.2f (block %d), last end %.2f (block %d); ends of blocks 0, G/2, G-2, G-1: %.2f %.2f %.2f %.2f; starts: %.2f %.2f %.2f %.2f\n",
                lastStart, bStart, lastEnd, bEnd, (double)(we[0] - s0) * k, (double)(we[G / 2] - s0) * k, (double)(we[G - 2] - s0) * k, (double)(we[G - 1] - s0) * k,
                (double)(ws[0] - s0) * k, (double)(ws[G / 2] - s0) * k, (double)(ws[G - 2] - s0) * k, (double)(ws[G - 1] - s0) * k);
        memset(ws, 0, sizeof(ws)); HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_wgS), ws, sizeof(ws))); HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_wgE), ws, sizeof(ws))); }
      { unsigned long long q[16]; HIP_OK(hipMemcpyFromSymbol(q, HIP_SYMBOL(g_dec), sizeof(q)));
        const double c = q[15] ? 1.0 / (100.0 * (double)q[15]) : 0.0;
        // (stamps 1, 2 are only taken for valid proposals: their deltas are approximate)
        fprintf(stderr, "DBG decide (%llu): stamps 0..7 deltas: total %.2f | leaf loop (draws) %.2f, quantiles %.2f, write-out %.2f | entry->accept test done %.2f, accepted-tree update %.2f\n", q[15],
                (double)(q[7] - q[0]) * c, (double)(q[5] - q[4]) * c, (double)(q[6] - q[5]) * c, (double)(q[7] - q[6]) * c, (double)(q[3] - q[0]) * c, (double)(q[4] - q[3]) * c);
        unsigned long long z16[16] = {0}; HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_dec), z16, sizeof(z16))); }
      fprintf(stderr, "DBG control workgroup (us from its start): proposal settled %.2f, draw ahead starts %.2f, drawn %.2f, image written %.2f | launches with both halves %llu, of which wave 0 had to propose itself %llu | wave 0 enters its role %.2f, all loads issued %.2f\n",
              h[24] * k, h[25] * k, h[26] * k, h[27] * k, h[29], h[28], h[30] * k, h[31] * k);
      fprintf(stderr, "DBG k_step us from the start of one workgroup (avg over %llu launches): reducers done %.2f | decider loads %.2f totals %.2f verdict posted %.2f decide %.2f stores %.2f | cand0: loads %.2f ready %.2f proposed %.2f verdict %.2f | arrival at the barrier: loaders %.2f cand0 %.2f cand1 %.2f | (iter %.0f) write-backs done %.2f | barrier %.2f pass done %.2f | pass (first bin pass): routing columns arrived +%.2f, prefetched quads done +%.2f (routing init +%.2f, deeper levels +%.2f [%.2f iterations], arithmetic +%.2f), block reduction + partials +%.2f\n",
              h[0], h[1] * k, h[2] * k, h[3] * k, h[23] * k, h[4] * k, h[5] * k, h[19] * k, h[20] * k, h[21] * k, h[22] * k, h[16] * k, h[17] * k, h[18] * k, (double)h[6], h[7] * k, h[8] * k, h[9] * k, (double)(h[11] - h[10]) * k, (double)(h[12] - h[11]) * k, (double)(h[14] - h[11]) * k, (double)(h[15] - h[14]) * k, h[0] ? (double)h[6] / (double)h[0] : 0.0,
              (double)(h[12] - h[15]) * k, (double)(h[13] - h[12]) * k);
      unsigned long long z[32] = {0}; HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_step), z, sizeof(z))); }
#endif
#ifdef S4B_LIGHT_TIMING
    { unsigned long long h[16]; HIP_OK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_arr), sizeof(h)));
      const double k = h[9] ? 1.0 / (100.0 * (double)h[9]) : 0.0;
      fprintf(stderr, "LIGHT k_step, workgroup 100, us from its start (avg over %llu launches): arrival at the barrier of waves 0..7: %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f | pass done %.2f\n",
              h[9], h[0] * k, h[1] * k, h[2] * k, h[3] * k, h[4] * k, h[5] * k, h[6] * k, h[7] * k, h[8] * k);
      unsigned long long z[16] = {0}; HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_arr), z, sizeof(z))); }
#endif
    out[0] = cnt ? sum / cnt : 0.0; out[3] = cnt;        // the fused launch (statistics + control + apply)
    out[1] = 0.0; out[4] = 0.0;                          // no separate control kernel
    out[2] = cntLast ? sumLast / cntLast : 0.0; out[5] = cntLast;
    HIP_OK(hipEventRecord(evStart_, stream_));
    for (int sIdx = 0; sIdx < nSweeps; ++sIdx) sweep(thin);
    HIP_OK(hipEventRecord(evStop_, stream_));
    sync();
    float ms = 0; HIP_OK(hipEventElapsedTime(&ms, evStart_, evStop_));
    out[6] = ms * 1000.0 / nSweeps;
  }
  bool fused() const { return useFused_; }
  // tree-update path: 0 automatic, 1 two kernels per tree (k_tree + k_control), 2 fused (k_step), 4 persistent (k_sweep: residual in the
  // registers of the pass waves), 5 persistent with the streaming pass (k_sweep_stream: on request only, at any size); 3 was the lagged
  // launch (k_lag) of round 3, removed
  void set_tree_path(int path) {
    if (path < 0 || path > 5 || path == 3) throw std::invalid_argument("tree path must be 0 (automatic), 1 (two-kernel), 2 (fused), 4 (persistent) or 5 (persistent, streaming pass)");
    pathReq_ = path; choose_path();
  }
  void get_tree_path(int32_t out[2]) const { out[0] = pathReq_; out[1] = path_; }
  bool is_persistent() const { return path_ == PATH_SWEEP || path_ == PATH_STREAM; }
  void choose_path() {
    int want = pathReq_;
    // automatic: the persistent sweep wherever it applies; else the fused launch while a tree update is latency-bound (few quads per thread),
    // two kernels per tree beyond and when three or more chains share the device.
    // (The streaming variant of the persistent launch is never the automatic choice: from the four register-heavy pass waves of a compute
    // unit it streams at 3.1 TB/s — 70.8 us per tree update at n = 1e7 against 62.7 for k_tree + k_control, 24.0 against 24.1 for k_step at
    // n = 2e6: DESIGN.md 8, round 5.)
    // (Round 5: the persistent sweep also when chains share the device.  Samplers of one process take turns through the per-device lock, launches of
    // other processes are sorted out by the roll call, and a chain's host phase runs under another chain's sweep: tools/multi_chain_probe.py at
    // n = 1e6, aggregate iterations/s with 2 / 3 / 4 / 6 chains: 532 / 461 / 521 / 539 against 393 / 430 / 390 / 444 on the per-tree kernels the hint
    // used to select.  The hint still picks between those where the persistent sweep does not apply.)
    if (want == 0) want = sweepRegsOk_ ? PATH_SWEEP : (sharing_ >= 3 ? PATH_TWO : (fusedAuto_ ? PATH_FUSED : PATH_TWO));
    if (want == PATH_SWEEP && !sweepRegsOk_) want = fusedOk_ ? PATH_FUSED : PATH_TWO;
    if (want == PATH_STREAM && !sweepStreamOk_) want = fusedOk_ ? PATH_FUSED : PATH_TWO;
    if (want == PATH_FUSED && !fusedOk_) want = PATH_TWO;
    if (want == path_) return;
    path_ = want; useFused_ = path_ == PATH_FUSED;
    sweepStream_ = path_ == PATH_STREAM;
    sweepGrid_ = (path_ == PATH_SWEEP && sweepSolo_) ? 1 : a_.gridF;
    if (graphExec_) { (void)hipGraphExecDestroy(graphExec_); graphExec_ = nullptr; }
  }
  // Hint: `chains` samplers share this device.  The fused launch keeps every CU busy with one workgroup of 8 register-heavy waves,
  // which leaves no room for another chain's kernels: with three or more chains per device the two-kernel tree update gives the
  // higher aggregate rate (measured at n = 1e6: 4 chains 512 vs 425 iterations/s; 2 chains 379 vs 403).  Between sweeps both
  // paths start from the same state (main tree arrays, generator slot 0), so the switch is safe at any call boundary.
  void set_device_sharing(int chains) { sharing_ = chains; choose_path(); }
  // persistent path: the sweep IS one launch; out[0] = its average duration (HIP events on the sampler's stream), out[1] = sweeps that
  // were handed over to k_step so far / all persistent sweeps so far (out[4]), out[6] = wall time per sweep without events
  void profile_sweep_persistent(int nSweeps, int thin, double* out) {
    double sum = 0; int cnt = 0;
#ifdef S4B_SWEEP_WG
    { sync(); static unsigned long long dropW[256 * 16]; sweep_wg_fetch(dropW); static unsigned long long dropD[256 * 64]; sweep_wd_fetch(dropD); }
#endif
#ifdef S4B_SWEEP_TIMING
    { sync(); unsigned long long drop[128]; sweep_timing_fetch(drop); }     // (only the sweeps profiled here: not the chain's first ones, which rebuild every structure cache)
#endif
    for (int sIdx = 0; sIdx < nSweeps * thin; ++sIdx) {
      const int64_t ho = sweepHandOvers_;
      {
        std::lock_guard<std::mutex> turn(sweep_mutex(device_));
        HIP_OK(hipEventRecord(evStart_, stream_));
        sweepStatus_[0] = -1; sweepStatus_[13] = 0; sweepStatus_[14] = 0;
        launch_sweep_kernel();
        HIP_OK(hipEventRecord(evStop_, stream_));
        sync();
      }
      const int st = sweepStatus_[0];
      ++sweepCount_;
      count_persistent_launch(st);
      if (st == -2) { ++sweepBusy_; sweep_fused_or_two_one(); if (binary_) launch_latents(); if (kModeled_) draw_k(); sync(); continue; }     // (roll call failed: the device is shared; not a sample of the persistent launch)
      if (st < 0 || st > T_ + 1) throw std::runtime_error("persistent tree sweep: the launch did not complete");
      if (st != T_ + 1) {
        ++sweepHandOvers_;
        if (st == 0) { if (splitProbs_) sweep_two_one(); else sweep_fused_one(); }
        else { for (int t = st; t <= T_; ++t) { launch_step(t); ++launches_; }
               if (((T_ + 1) & 1) == 1) HIP_OK(hipMemcpyAsync(a_.rngF, a_.rngF + 1, sizeof(MTState), hipMemcpyDeviceToDevice, stream_)); }
      }
      if (binary_) launch_latents();
      if (kModeled_) draw_k();
      sync();
      if (ho == sweepHandOvers_) { float ms = 0; HIP_OK(hipEventElapsedTime(&ms, evStart_, evStop_)); sum += ms * 1000.0; ++cnt; }
    }
#ifdef S4B_SWEEP_WG
    { static unsigned long long w[256 * 16]; sweep_wg_fetch(w);
      double mn = 1e30, mx = -1e30, sm = 0.0, smn = 1e30, smx = -1e30, xmn = 1e30, xmx = -1e30, xsm = 0.0; int cntW = 0, cntX = 0, argmx = -1, argsmx = -1, argsmn = -1;
      double refSum = 0.0; int refCnt = 0;
      for (int b = 0; b < 256; ++b) if (w[b * 16 + 3] && w[b * 16 + 3] == w[3]) { refSum += (double)w[b * 16 + 2] / (double)w[b * 16 + 3]; ++refCnt; }
      const double ref = refCnt ? refSum / refCnt : 0.0;
      for (int b = 0; b < 256; ++b) {
        if (w[b * 16 + 1]) { const double v = (double)w[b * 16] / (100.0 * (double)w[b * 16 + 1]); if (v < mn) mn = v; if (v > mx) { mx = v; argmx = b; } sm += v; ++cntW; }
        if (w[b * 16 + 3] && w[b * 16 + 3] == w[3]) { const double sk = ((double)w[b * 16 + 2] / (double)w[b * 16 + 3] - ref) / 100.0; if (sk < smn) { smn = sk; argsmn = b; } if (sk > smx) { smx = sk; argsmx = b; } }
        if (w[b * 16 + 5]) { const double v = (double)w[b * 16 + 4] / (100.0 * (double)w[b * 16 + 5]); if (v < xmn) xmn = v; if (v > xmx) xmx = v; xsm += v; ++cntX; }
      }
      fprintf(stderr, "SWEEP exchange, per workgroup (steps whose statistics were published speculatively: %llu): last publish of anybody -> totals seen: min %.2f mean %.2f max %.2f us; first -> last publish %.2f us\n",
              w[5], xmn, cntX ? xsm / cntX : 0.0, xmx, w[5] ? (double)w[6] / (100.0 * (double)w[5]) : 0.0);
      { double a8 = 0, a9 = 0, a11 = 0; int c8 = 0;
        for (int b = 0; b < 255; ++b) if (w[b * 16 + 1]) { const double kq = 1.0 / (100.0 * (double)w[b * 16 + 1]); a8 += (double)w[b * 16 + 8] * kq; a9 += (double)w[b * 16 + 9] * kq; a11 += (double)w[b * 16 + 11] * kq; ++c8; }
        if (c8) fprintf(stderr, "SWEEP per workgroup (mean over the workgroups), us after the totals were seen, steps that speculate: wave 3's leaf values out %.2f, wave 5 starts the statistics %.2f, wave 4 has published (below), wave 5 through with the step (steps borne out, per speculating step) %.2f\n", a8 / c8, a9 / c8, a11 / c8); }
#ifdef S4B_SWEEP_WGD
      { static unsigned long long wd[256 * 64]; sweep_wd_fetch(wd);
        double d[8], c[8], big[8], mx[8]; for (int k = 0; k < 8; ++k) { d[k] = 0; c[k] = 0; big[k] = 0; mx[k] = 0; }
        for (int b = 0; b < 255; ++b) for (int k = 0; k < 8; ++k) { d[k] += (double)wd[b * 64 + k]; c[k] += (double)wd[b * 64 + 8 + k]; big[k] += (double)wd[b * 64 + 16 + k]; mx[k] = std::max(mx[k], (double)wd[b * 64 + 24 + k] / 100.0); }
        auto A = [&](int k) { return c[k] > 0 ? d[k] / (100.0 * c[k]) : 0.0; };
        fprintf(stderr, "SWEEP pass wave 5, durations (mean over workgroups and steps; counts %.0f / %.0f / %.0f), us: end of the previous step -> next tree's leaf ids requested %.2f, wait for the image drawn ahead %.2f, routing under it %.2f, "
                        "wait for wave 3's leaf values %.2f, ahead pass %.2f, wait for the verdict + fold %.2f | a step done the old way with its ahead pass behind it %.2f | wave 3's wait for the ahead pass of the step before %.2f\n",
                c[1], c[4], c[6], A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7));
        { // the workgroups whose wave 3 waits longest for its pass waves: their own phase means
          int worst[3] = {-1, -1, -1}; double wv[3] = {-1, -1, -1};
          for (int b = 0; b < 255; ++b) { const double v = wd[b * 64 + 8 + 7] ? (double)wd[b * 64 + 7] / (double)wd[b * 64 + 8 + 7] : 0.0;
            for (int q = 0; q < 3; ++q) if (v > wv[q]) { for (int r = 2; r > q; --r) { wv[r] = wv[r - 1]; worst[r] = worst[r - 1]; } wv[q] = v; worst[q] = b; break; } }
          for (int q = 0; q < 3; ++q) if (worst[q] >= 0) { const int b = worst[q]; fprintf(stderr, "SWEEP workgroup %d (wave 3 waits longest for its pass waves), its own means of the eight phases:", b);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %.2f", wd[b * 64 + 8 + k] ? (double)wd[b * 64 + k] / (100.0 * (double)wd[b * 64 + 8 + k]) : 0.0); fprintf(stderr, "\n"); }
          double lo = 1e9, hi = 0; int blo = -1, bhi = -1;
          for (int b = 0; b < 255; ++b) { const double v = wd[b * 64 + 8 + 4] ? (double)wd[b * 64 + 4] / (100.0 * (double)wd[b * 64 + 8 + 4]) : 0.0; if (v < lo) { lo = v; blo = b; } if (v > hi) { hi = v; bhi = b; } }
          fprintf(stderr, "SWEEP ahead pass, per-workgroup means: %.2f (workgroup %d) .. %.2f us (workgroup %d)\n", lo, blo, hi, bhi);
          lo = 1e9; hi = 0;
          for (int b = 0; b < 255; ++b) { const double v = wd[b * 64 + 8 + 2] ? (double)wd[b * 64 + 2] / (100.0 * (double)wd[b * 64 + 8 + 2]) : 0.0; if (v < lo) { lo = v; blo = b; } if (v > hi) { hi = v; bhi = b; } }
          fprintf(stderr, "SWEEP routing, per-workgroup means: %.2f (workgroup %d) .. %.2f us (workgroup %d)\n", lo, blo, hi, bhi); }
        { double as[8], ac[8]; for (int k = 0; k < 8; ++k) { as[k] = 0; ac[k] = 0; }
          for (int b = 0; b < 255; ++b) for (int k = 0; k < 8; ++k) { as[k] += (double)(long long)wd[b * 64 + 32 + k]; ac[k] += (double)wd[b * 64 + 48 + k]; }
          auto M = [&](int k) { return ac[k] > 0 ? as[k] / (100.0 * ac[k]) : 0.0; };
          fprintf(stderr, "SWEEP moments of a step, us after this workgroup saw its totals (means over workgroups and steps): decider past the totals %.2f, has wave 3's leaf values %.2f, past its wait for the pass waves %.2f, verdict out %.2f, "
                          "decider's step ends %.2f | proposal settled %.2f, wave 1 starts drawing the ahead image %.2f, has drawn it %.2f\n", M(0), M(1), M(2), M(3), M(4), M(7), M(5), M(6)); }
        fprintf(stderr, "SWEEP the same phases: share of the steps in which the phase took more than 3 us (maximum, us):");
        for (int k = 0; k < 8; ++k) fprintf(stderr, " [%d] %.4f (%.1f)", k, c[k] > 0 ? big[k] / c[k] : 0.0, mx[k]);
        fprintf(stderr, "\n"); }
#endif
      fprintf(stderr, "SWEEP per workgroup: totals seen -> speculative statistics published, us: min %.2f mean %.2f max %.2f (workgroup %d) over %d workgroups; the moment the totals are seen, relative to the mean: %.2f (workgroup %d) .. %.2f us (workgroup %d)\n",
              mn, cntW ? sm / cntW : 0.0, mx, argmx, cntW, smn, argsmn, smx, argsmx); }
#endif
#ifdef S4B_SWEEP_TIMING
    { unsigned long long h[128]; sweep_timing_fetch(h);
      const double k = h[0] ? 1.0 / (100.0 * (double)h[0]) : 0.0;
      fprintf(stderr, "SWEEP workgroup 100 (avg over %llu steps, %.2f bins, %llu routed after the decision) | pass waves, us after the previous publish: totals gathered (wave 3) %.2f; wave 4: images there %.2f, routed %.2f, tables + proposal there %.2f, arithmetic done %.2f; published (wave 3) = step %.2f | decider, us after its previous step: totals seen %.2f, verdict %.2f, tables out %.2f, step end %.2f\n",
              h[0], h[0] ? (double)h[7] / (double)h[0] : 0.0, h[8], h[1] * k, h[2] * k, h[3] * k, h[4] * k, h[5] * k, h[6] * k, h[9] * k, h[10] * k, h[11] * k, h[12] * k);
      fprintf(stderr, "SWEEP steps whose leaf values went out with the verdict (move not accepted, wave 3's values): %llu of %llu\n", h[35], h[0]);
      { auto A = [&](int i) { return (double)(long long)h[i] * k; };
        fprintf(stderr, "SWEEP timeline of a step, us after this workgroup saw the totals complete: wave 3's leaf values %.2f | decider: sees totals %.2f, verdict + old values out %.2f, new values out %.2f | proposal settled %.2f (x%.2f) | wave 5: (routed %.2f) past foldReady %.2f, old values folded %.2f, tables + proposal seen %.2f, new values folded %.2f, statistics reduced %.2f | wave 4: published %.2f | next totals complete = step %.2f\n",
                A(51), A(41), A(42), A(43), A(52), 1.0, A(44) - A(40), A(45), A(46), A(47), A(48), A(49), A(50), A(40));
        { const double ks = h[89] ? 1.0 / (100.0 * (double)h[89]) : 0.0;
          auto G = [&](int i) { return (double)(long long)h[i] * ks; };
          fprintf(stderr, "SWEEP speculation: %llu steps published before their verdict, %llu of them borne out | timeline of those steps, us after the totals: wave 5 starts the statistics %.2f, wave 4 has published %.2f, wave 5 past foldReady %.2f, through with the step (borne out) %.2f | all steps: wave 3 past paGo %.2f, decider's decDone %.2f\n",
                  h[89], h[88], G(90), G(91), G(92), h[88] ? (double)(long long)h[93] / (100.0 * (double)h[88]) : 0.0, (double)(long long)h[94] * k, (double)(long long)h[95] * k); }
        fprintf(stderr, "SWEEP foldReady: stored by wave 0 -> wave 4 / wave 5 past their wait: %.2f / %.2f us; wave 5 still routing when it was stored: %llu steps of %llu, by %.2f us on average\n", A(53), A(54), h[55], h[0], h[55] ? (double)h[56] / (100.0 * (double)h[55]) : 0.0);
        const double kl0 = h[55] ? 1.0 / (100.0 * (double)h[55]) : 0.0;
        const double kl = h[55] ? 1.0 / (100.0 * (double)h[55]) : 0.0, ke = (h[0] - h[55]) ? 1.0 / (100.0 * (double)(h[0] - h[55])) : 0.0;
        fprintf(stderr, "SWEEP late routing steps: wave 5 waited %.2f us for the images\n", (double)(long long)h[28] * kl0);
        fprintf(stderr, "SWEEP late routing steps by the move type of image 0: birth %llu, death %llu, swap %llu, change %llu\n", h[20], h[21], h[22], h[23]);
        fprintf(stderr, "SWEEP late routing steps: routing took %.2f us, images seen %.2f us after the totals of the step, %llu of them early-verdict steps | other steps: routing took %.2f us, images seen %.2f us after the totals of the PREVIOUS step\n",
                (double)(long long)h[57] * kl, (double)(long long)h[58] * kl, h[59], (double)(long long)h[60] * ke, (double)(long long)h[61] * ke); }
      fprintf(stderr, "SWEEP folded in, us after the wave's own previous pass: waves 4..7: %.2f %.2f %.2f %.2f\n", h[16] * k, h[17] * k, h[18] * k, h[19] * k);
      fprintf(stderr, "SWEEP image drawings (both waves) that took more than 8 us from the settled proposal to imgReady: %llu; generator copied %.2f, model + wait for the staged tree %.2f, tree loaded + generator advanced %.2f, propose() %.2f, image stored %.2f us\n",
              h[15], h[15] ? h[13] / (100.0 * h[15]) : 0.0, h[15] ? h[14] / (100.0 * h[15]) : 0.0, h[15] ? h[29] / (100.0 * h[15]) : 0.0, h[15] ? h[34] / (100.0 * h[15]) : 0.0, h[15] ? h[70] / (100.0 * h[15]) : 0.0);
      fprintf(stderr, "SWEEP all image drawings (%llu): cache invalid %llu times, generator block renewed %llu times; tree loaded %.2f, cache %.2f, generator opened + advanced %.2f us\n",
              h[69], h[64], h[65], h[69] ? h[66] / (100.0 * h[69]) : 0.0, h[69] ? h[67] / (100.0 * h[69]) : 0.0, h[69] ? h[68] / (100.0 * h[69]) : 0.0);
      fprintf(stderr, "SWEEP image wave 1, propose() alone by move type (count, us): birth %llu %.2f, death %llu %.2f, swap %llu %.2f, change %llu %.2f; without a valid move %llu; from the start of the drawing step to propose() %.2f us\n",
              h[36], h[36] ? h[24] / (100.0 * h[36]) : 0.0, h[37], h[37] ? h[25] / (100.0 * h[37]) : 0.0, h[38], h[38] ? h[26] / (100.0 * h[38]) : 0.0, h[39], h[39] ? h[27] / (100.0 * h[39]) : 0.0, h[62],
              (h[36] + h[37] + h[38] + h[39]) ? h[63] / (100.0 * (h[36] + h[37] + h[38] + h[39])) : 0.0);
      { unsigned long long d[16]; sweep_decide_fetch(d);
        if (d[15]) { auto A = [&](int i) { return d[7 + i] ? (double)d[i] / (100.0 * (double)d[7 + i]) : 0.0; };
          fprintf(stderr, "SWEEP decide() of the decider wave (%llu calls; %llu with a pending move), us after its entry: bin log-likelihoods %.2f, ratio ready %.2f | all calls: accept test out (generator open, uniform; hooks: early verdict) %.2f, tree updated %.2f, leaf statistics + uniforms %.2f, leaf values %.2f, stored %.2f\n",
                  d[15], d[8], A(1), A(2), A(3), A(4), A(5), A(6), A(7)); } }
      fprintf(stderr, "SWEEP bins per step (histogram 0..15+):"); for (int i = 0; i < 16; ++i) fprintf(stderr, " %llu", h[72 + i]); fprintf(stderr, "\n");
      fprintf(stderr, "SWEEP statistics phase: wave 5 accumulated %.2f, wave-reduced + slots written %.2f, barrier passed %.2f; wave 3 barrier passed %.2f\n", h[30] * k, h[31] * k, h[32] * k, h[33] * k); }
#endif
    out[0] = cnt ? sum / cnt : 0.0; out[3] = cnt;
    out[1] = (double)sweepHandOvers_; out[4] = (double)sweepCount_;
    out[2] = 0.0; out[5] = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int sIdx = 0; sIdx < nSweeps; ++sIdx) sweep(thin);
    sync();
    out[6] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / nSweeps;
  }
  void profile_sweep(int nSweeps, int thin, double* out) {
    if (is_persistent()) { profile_sweep_persistent(nSweeps, thin, out); return; }
    if (useFused_) { profile_sweep_fused(nSweeps, thin, out); return; }
    const int perSweep = 2 * T_ * thin + thin;
    std::vector<hipEvent_t> ev((size_t)perSweep * 2 + 4);
    for (auto& e : ev) HIP_OK(hipEventCreate(&e));
    double sum[3] = {0, 0, 0}, cnt[3] = {0, 0, 0};
    for (int sIdx = 0; sIdx < nSweeps; ++sIdx) {
      size_t e = 0;
      for (int k = 0; k < thin; ++k) {
        hipLaunchKernelGGL(k_control, dim3(1), dim3(CBLOCK), 0, stream_, a_, -1, 0); ++launches_;
        for (int t = 0; t < T_; ++t) {
          HIP_OK(hipEventRecord(ev[e++], stream_));
          launch_tree(t);
          HIP_OK(hipEventRecord(ev[e++], stream_));
          HIP_OK(hipEventRecord(ev[e++], stream_));
          hipLaunchKernelGGL(k_control, dim3(1), dim3(CBLOCK), 0, stream_, a_, t, t + 1 < T_ ? t + 1 : -1);
          HIP_OK(hipEventRecord(ev[e++], stream_));
          launches_ += 2;
        }
        HIP_OK(hipEventRecord(ev[e++], stream_));
        hipLaunchKernelGGL(k_apply, dim3(a_.grid), dim3(BLOCK), ldsApply_, stream_, a_, T_ - 1); ++launches_;
        HIP_OK(hipEventRecord(ev[e++], stream_));
        if (binary_) launch_latents();
        if (kModeled_) draw_k();
      }
      sync();
      for (size_t i = 0; i + 1 < e; i += 2) {
        float ms = 0; HIP_OK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        const size_t pair = i / 2, perThin = (size_t)2 * T_ + 1, pos = pair % perThin;
        const bool first = pos == 0;                       // k_tree<false>: no apply half, not representative
        const int c = pos == perThin - 1 ? 2 : (int)(pos % 2);
        if (c == 0 && first) continue;
        sum[c] += ms * 1000.0; cnt[c] += 1;
      }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    for (int c = 0; c < 3; ++c) { out[c] = cnt[c] ? sum[c] / cnt[c] : 0.0; out[3 + c] = cnt[c]; }
    // wall time per sweep without per-launch events
    HIP_OK(hipEventRecord(evStart_, stream_));
    for (int sIdx = 0; sIdx < nSweeps; ++sIdx) sweep(thin);
    HIP_OK(hipEventRecord(evStop_, stream_));
    sync();
    float ms = 0; HIP_OK(hipEventElapsedTime(&ms, evStart_, evStop_));
    out[6] = ms * 1000.0 / nSweeps;
#ifdef S4B_CONTROL_TIMING
    { long long h[40]; HIP_OK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg), sizeof(h)));
      fprintf(stderr, "DBG winner propose us by move type (count): birth %.2f (%lld) death %.2f (%lld) swap %.2f (%lld) change %.2f (%lld)\n",
              h[32] ? h[25]/100.0/h[32] : 0.0, h[32], h[33] ? h[26]/100.0/h[33] : 0.0, h[33], h[34] ? h[27]/100.0/h[34] : 0.0, h[34], h[35] ? h[28]/100.0/h[35] : 0.0, h[35]);
      if (h[7]) fprintf(stderr, "DBG candidate timeline us: loads %.2f advance %.2f rebuild %.2f propose %.2f wait-verdict %.2f stores %.2f | draws/propose %.2f\n", h[16]/100.0/h[7], h[17]/100.0/h[7], h[18]/100.0/h[7], h[19]/100.0/h[7], h[20]/100.0/h[7], h[21]/100.0/h[7], (double)h[22]/h[7]);
      if (h[7]) fprintf(stderr, "DBG candidate reached the verdict wait %.2f us after wave 0 posted it; candidate wave started %.2f us after wave 0\n", (h[23]/(double)h[7] - 100000)/100.0, (h[24]/(double)h[7] - 100000)/100.0);
      fprintf(stderr, "DBG k_tree per-WG us: stage %.2f pass %.2f (n=%lld)\n", h[8]/100.0/h[10], h[9]/100.0/h[10], h[10]);       fprintf(stderr, "DBG control per-call us: stage %.2f (loads %.2f wait-reduce %.2f) decide %.2f own-propose %.2f out %.2f n=%lld | candidate hit %lld, winner propose-done at %.2f, end at %.2f\n", h[0]/100.0/h[4], h[5]/100.0/h[4], h[6]/100.0/h[4], h[1]/100.0/h[4], h[2]/100.0/h[4], h[3]/100.0/h[4], h[4], h[7], h[7] ? h[14]/100.0/h[7] : 0.0, h[7] ? h[15]/100.0/h[7] : 0.0); }
#endif
  }
  void launch_test_fits() {
    const size_t lds = (size_t)TF_ROWS * (size_t)T_ * 8;
    if (nTest_ <= 65536 && lds <= 48 * 1024) {      // few rows: a thread per (row, tree)
      const int g = (int)std::min<int64_t>(GRID_MAX, std::max<int64_t>(1, (nTest_ + TF_ROWS - 1) / TF_ROWS));
      hipLaunchKernelGGL(k_test_fits_few, dim3(g), dim3(BLOCK), lds, stream_, a_, testOut_);
    } else {
      const int g = (int)std::min<int64_t>(GRID_MAX, std::max<int64_t>(1, (nTest_ + BLOCK - 1) / BLOCK));
      hipLaunchKernelGGL(k_test_fits, dim3(g), dim3(BLOCK), 0, stream_, a_, testOut_);
    }
    ++launches_;
  }
  // The fits of the test rows of an iteration only need the trees of its sweep: run() asks for them BEFORE the sweep (request_test_fits), stan_inputs() queues
  // the kernel and the copy to pinned host memory behind the Stan block's input kernels — in front of its wait for their result — and test_fits() finds them
  // done: no launch + wait of their own per iteration.  A sweep that was handed over forms the Stan inputs again, and these with them.
  void request_test_fits() { testWanted_ = nTest_ > 0 && testPinned_ != nullptr; testQueued_ = false; }
  void request_var_counts() { varWanted_ = true; varQueued_ = false; }      // (the same for the predictors' use counts)
  void queue_test_fits_if_wanted() {
    if (testWanted_) {
      launch_test_fits();
      HIP_OK(hipMemcpyAsync(testPinned_, testOut_, (size_t)nTest_ * 8, hipMemcpyDeviceToHost, stream_));
      testQueued_ = true;
    }
    if (varWanted_) { launch_var_counts(); varQueued_ = true; }
  }
  void launch_var_counts() {
    hipLaunchKernelGGL(k_var_counts, dim3(1), dim3(BLOCK), 0, stream_, a_, varDev_); ++launches_;
    HIP_OK(hipMemcpyAsync(varPinned_, varDev_, (size_t)P_ * 4, hipMemcpyDeviceToHost, stream_));
  }
  void var_counts(int32_t* out) {
    if (!(varWanted_ && varQueued_)) launch_var_counts();
    varWanted_ = false; varQueued_ = false;
    sync();
    std::memcpy(out, varPinned_, (size_t)P_ * 4);
  }
  void test_fits(double* out) {
    if (testWanted_ && testQueued_) { sync(); std::memcpy(out, testPinned_, (size_t)nTest_ * 8); testWanted_ = false; testQueued_ = false; return; }
    testWanted_ = false; testQueued_ = false;
    launch_test_fits();
    download(out, testOut_, (size_t)nTest_); sync();
  }

  void predict_stored(const uint16_t* xb, int64_t nT, const PackedNode* nodes, size_t numNodes, const int64_t* treeStart, int64_t S, int T,
                      const double* scale, int binary, double* out) {
    uint16_t* dx = nullptr; PackedNode* dn = nullptr; int64_t* dt = nullptr; double* ds = nullptr; double* dout = nullptr;
    auto freeAll = [&] { (void)hipFree(dx); (void)hipFree(dn); (void)hipFree(dt); (void)hipFree(ds); (void)hipFree(dout); };
    try {
      HIP_OK(hipMalloc(&dx, (size_t)a_.P * (size_t)nT * 2)); HIP_OK(hipMalloc(&dn, std::max<size_t>(16, numNodes * sizeof(PackedNode))));
      HIP_OK(hipMalloc(&dt, (size_t)S * T * 8)); HIP_OK(hipMalloc(&ds, (size_t)S * 16)); HIP_OK(hipMalloc(&dout, (size_t)S * (size_t)nT * 8));
      upload(dx, xb, (size_t)a_.P * (size_t)nT); upload(dn, nodes, numNodes); upload(dt, treeStart, (size_t)S * T); upload(ds, scale, (size_t)S * 2);
      const int64_t total = nT * S;
      const int g = (int)std::min<int64_t>(4096, std::max<int64_t>(1, (total + BLOCK - 1) / BLOCK));
      hipLaunchKernelGGL(k_predict, dim3(g), dim3(BLOCK), 0, stream_, dx, nT, dn, dt, S, T, ds, binary, dout); ++launches_;
      download(out, dout, (size_t)total); sync();
    } catch (...) { freeAll(); throw; }
    freeAll();
  }

  // ---- Stan inputs
  void stan_inputs(int mode, bool wantTrain, double* cX, double* cZ, double* s0, double* trainOut) {
    flush_hand_off();
    if (stanFused_) {
      launch_stan_fused(mode, wantTrain ? 1 : 0, false);
      queue_test_fits_if_wanted();
      if (!fetch_fused(cX, cZ, s0)) { reduce_pipeline(mode, wantTrain ? 1 : 0, 0); fetch_out(cX, cZ, s0); recentre_scales(cX, cZ, *s0); }
    } else { reduce_pipeline(mode, wantTrain ? 1 : 0, 0); fetch_out(cX, cZ, s0); }
    if (wantTrain && trainOut) { download(trainOut, s_.train, (size_t)n_); sync(); }
  }
  double leapfrog_sums(const double* beta, const double* b, double* gX, double* gZ) {
    double ss;
    inlineBeta_ = beta; inlineB_ = b;
    if (!(stanFused_ && K_ + q_ <= S_PAR_INLINE)) push_params(beta, b);
    if (stanFused_) {
      launch_stan_fused(0, 0, true);
      const bool fusedOk = fetch_fused(gX, gZ, &ss);
      if (!fusedOk) {
        if (K_ + q_ <= S_PAR_INLINE) push_params(beta, b);     // (the plain-double kernels read beta, b from device memory)
        reduce_pipeline(0, 0, 1); fetch_out(gX, gZ, &ss); recentre_scales(gX, gZ, ss);
      }
#ifdef S4B_FX_DEBUG
      if (fusedOk) {   // (debug build: every fixed-point evaluation is repeated in plain doubles and compared)
        std::vector<double> rX((size_t)K_ + 1), rZ((size_t)q_ + 1); double rs;
        if (K_ + q_ <= S_PAR_INLINE) push_params(beta, b);
        reduce_pipeline(0, 0, 1); fetch_out(rX.data(), rZ.data(), &rs);
        double worst = std::fabs(rs - ss) / (std::fabs(rs) + 1e-300);
        for (int k = 0; k < K_; ++k) worst = std::max(worst, std::fabs(rX[k] - gX[k]) / (std::fabs(rX[k]) + 1e-12));
        for (int j = 0; j < q_; ++j) worst = std::max(worst, std::fabs(rZ[j] - gZ[j]) / (std::fabs(rZ[j]) + 1e-12));
        if (worst > 1e-9) fprintf(stderr, "FXDBG eval %lld: worst rel diff %.3e (ss %.17g vs %.17g) exps %d %d %d\n", (long long)fusedEvals_, worst, ss, rs, fxExp_[0], fxExp_[1], fxExp_[2]);
      }
#endif
      inlineBeta_ = nullptr; inlineB_ = nullptr;
    } else { reduce_pipeline(0, 0, 1); fetch_out(gX, gZ, &ss); }
    return ss;
  }
  // one launch per evaluation (k_stan_fused)
  template <int KMAX> void launch_stan_fused_k(const StanFusedArgs& f, bool direct, size_t lds, int grid) {
    if (direct) hipLaunchKernelGGL((k_stan_fused<KMAX, true>), dim3(grid), dim3(SBLOCK), lds, stream_, a_, s_, f);
    else hipLaunchKernelGGL((k_stan_fused<KMAX, false>), dim3(grid), dim3(SBLOCK), lds, stream_, a_, s_, f);
  }
  void launch_stan_fused(int mode, int wantTrain, bool direct) {
    StanFusedArgs f; f.acc = fusedAcc_; f.bad = fusedBad_; f.parity = fusedParity_; f.mode = mode; f.wantTrain = wantTrain;
    for (int g = 0; g < 3; ++g) f.scale[g] = std::ldexp(1.0, fxExp_[g]);
    f.hostOut = pinnedAcc_; f.seq = ++fusedSeq_; f.zFixed = zFixed_;
    f.parInline = (direct && K_ + q_ <= S_PAR_INLINE && inlineBeta_) ? 1 : 0; f.pad = 0;
    if (f.parInline) { for (int k = 0; k < K_; ++k) f.par[k] = inlineBeta_[k]; for (int j = 0; j < q_; ++j) f.par[K_ + j] = inlineB_[j]; }
    const int grid = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (n_ + SBLOCK - 1) / SBLOCK));
    if (K_ <= 2) launch_stan_fused_k<2>(f, direct, fusedLds_, grid);
    else if (K_ <= 4) launch_stan_fused_k<4>(f, direct, fusedLds_, grid);
    else if (K_ <= 8) launch_stan_fused_k<8>(f, direct, fusedLds_, grid);
    else launch_stan_fused_k<16>(f, direct, fusedLds_, grid);
    hipLaunchKernelGGL(k_stan_forward, dim3(1), dim3(SBLOCK), 0, stream_, f, 1 + K_ + q_);
    launches_ += 2;
  }
  // false: the evaluation cannot be trusted (see k_stan_fused) — the caller repeats it in plain doubles
  bool fetch_fused(double* cX, double* cZ, double* s0) {
    const size_t M = (size_t)(1 + K_ + q_), W = fused_words((int)M);
    {
      // the kernel's last workgroup wrote the accumulators into this (device-writable) host buffer and then the sequence number:
      // poll it instead of issuing a copy and synchronising the stream (~15 us less per evaluation)
      volatile unsigned long long* flag = pinnedAcc_ + W + 1;
      const auto t0 = std::chrono::steady_clock::now();
      long spins = 0;
      while (*flag != (unsigned long long)fusedSeq_) {
        if ((++spins & 0xfff) == 0) {
          if (hipStreamQuery(stream_) == hipSuccess && *flag != (unsigned long long)fusedSeq_) { sync(); break; }   // (kernel done: the store must be here)
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0) throw std::runtime_error("fused Stan sums: no result from the device within 30 s");
        }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      if (*flag != (unsigned long long)fusedSeq_) throw std::runtime_error("fused Stan sums: the result hand-off did not arrive");
    }
    fusedParity_ ^= 1;
    bool bad = (*(const int32_t*)(pinnedAcc_ + W)) != 0;
    // the magnitudes: sums of absolute contributions in units of 2^10 (after scaling); below 2^32 no limb can have wrapped
    unsigned long long mag[3] = {0, 0, 0};
    for (int g = 0; g < 3; ++g) mag[g] = pinnedAcc_[2 * M + g];
    for (int g = 0; g < 3; ++g) if (mag[g] >= (1ull << 32)) bad = true;
    int ex[3]; bool seen[3];
    for (int g = 0; g < 3; ++g) {
      const unsigned long long m = pinnedAcc_[2 * M + 3 + g];
      seen[g] = m != 0; ex[g] = (int)m - 4000;
      if (seen[g] && ex[g] < 4) bad = true;         // the scale is far too small for what is being summed now: resolution lost
    }
    ++fusedEvals_;
#ifdef S4B_FX_DEBUG
    if (bad) fprintf(stderr, "FXDBG eval %lld bad: flag %d mag %llu %llu %llu ex %d %d %d seen %d%d%d exps %d %d %d\n", (long long)fusedEvals_, (int)*(const int32_t*)(pinnedAcc_ + W),
                     mag[0], mag[1], mag[2], ex[0], ex[1], ex[2], (int)seen[0], (int)seen[1], (int)seen[2], fxExp_[0], fxExp_[1], fxExp_[2]);
#endif
    if (bad) {
      ++fusedFallbacks_;
      const bool overflow = (*(const int32_t*)(pinnedAcc_ + W)) != 0 || mag[0] >= (1ull << 32) || mag[1] >= (1ull << 32) || mag[2] >= (1ull << 32);
      // only the resolution check failed: the exponents are a valid measurement — centre on them, the caller repeats this one in doubles
      if (!overflow) { for (int g = 0; g < 3; ++g) if (seen[g]) fxExp_[g] = std::max(-900, std::min(900, fxExp_[g] + (22 - ex[g]))); fxTinyFail_ = true; }
      else fxTinyFail_ = false;
      return false;
    }
    const double inv[3] = {std::ldexp(1.0, -fxExp_[0]), std::ldexp(1.0, -fxExp_[1]), std::ldexp(1.0, -fxExp_[2])};
    auto val = [&](size_t k, int g) {
      const long long hi = (long long)pinnedAcc_[2 * k], lo = (long long)pinnedAcc_[2 * k + 1];
      return ((double)hi / 1048576.0 + (double)lo / 72057594037927936.0) * inv[g];
    };
    *s0 = val(0, 0);
    for (int k = 0; k < K_; ++k) cX[k] = val((size_t)1 + k, 1);
    for (int j = 0; j < q_; ++j) cZ[j] = val((size_t)1 + K_ + j, 2);
    // keep the largest workgroup magnitude of every group near 2^22 after scaling: with at most 2^11 workgroups the magnitude total
    // stays below 2^33, far inside the 2^42 bound, and the low limb resolves 2^-78 of a workgroup's contribution
    for (int g = 0; g < 3; ++g) if (seen[g]) fxExp_[g] = std::max(-900, std::min(900, fxExp_[g] + (22 - ex[g])));
    return true;
  }
  // after an evaluation in plain doubles that an OVERFLOW of the fixed-point range made necessary: smaller scales, from the totals
  // themselves where they are finite (a first guess; the next fused evaluation that succeeds centres them on the magnitudes).
  // Consecutive failures shrink them further.
  void recentre_scales(const double* cX, const double* cZ, double ss) {
    if (fxTinyFail_) return;        // (already centred on the measured exponents)
    double mx = 0.0, mz = 0.0;
    for (int k = 0; k < K_; ++k) if (std::isfinite(cX[k])) mx = std::max(mx, std::fabs(cX[k]));
    for (int j = 0; j < q_; ++j) if (std::isfinite(cZ[j])) mz = std::max(mz, std::fabs(cZ[j]));
    const double tot[3] = {std::isfinite(ss) ? std::fabs(ss) : 0.0, mx, mz};
    // a single evaluation out of range (one point of a diverging trajectory, the random initial values) leaves the scales alone: the
    // next one is usually back in range; only a second failure in a row moves them
    if (fxLastBad_ == fusedEvals_ - 1)
      for (int g = 0; g < 3; ++g) {
        int want = fxExp_[g] - 10;
        if (tot[g] > 0.0) want = std::min(want, 20 - std::ilogb(tot[g]));
        fxExp_[g] = std::max(-900, std::min(900, want));
      }
    fxLastBad_ = fusedEvals_;
  }
  void fused_stats(int64_t out[2]) const { out[0] = fusedEvals_; out[1] = fusedFallbacks_; }
  // The power-of-two scales of the fixed-point Stan sums follow the evaluation history.  They are put back to their initial values at
  // every run() and set_state(), so that a chain continued in another sampler (get_state / set_state) sums with the same scales as
  // the chain that was never interrupted: bit-identical, not just equal to 2^-67.
  void reset_fused_scales() { fxExp_[0] = fxExp_[1] = fxExp_[2] = 0; fxLastBad_ = -2; fxTinyFail_ = false; }
  void sweep_stats(int64_t out[2]) const { out[0] = sweepCount_; out[1] = sweepHandOvers_; }
  // {persistent launches that ran (roll call passed), tree updates decided inside them, steps whose statistics went out before their verdict, steps borne out}
  void sweep_spec(int64_t out[4]) const { out[0] = sweepLaunchesRun_; out[1] = sweepTreesInside_; out[2] = sweepSpecSteps_; out[3] = sweepSpecOk_; }
  void count_persistent_launch(int st) {
    if (st < 0 || st > T_ + 1) return;      // (busy or failed: nothing of the chain was touched / an error follows)
    ++sweepLaunchesRun_; sweepTreesInside_ += st == T_ + 1 ? T_ : std::max(st - 1, 0);      // (status t <= T: steps 0 .. t-1 ran inside, i.e. trees 0 .. t-2 were decided there)
    sweepSpecSteps_ += sweepStatus_[13]; sweepSpecOk_ += sweepStatus_[14];
  }
  int64_t sweep_busy() const { return sweepBusy_; }

  // HIP-event timing of the per-leapfrog O(N) sums (hmc_mode 1 path: e = e0 - X beta - Z b, |e|^2, X'e, Z'e) on the
  // sampler's stream.  out: [0] us per evaluation, kernels only; [1] us per evaluation including the result fetch;
  // [2] kernel launches per evaluation
  void profile_leapfrog(int nEvals, const double* beta, const double* b, double* out) {
    inlineBeta_ = nullptr; inlineB_ = nullptr;
    push_params(beta, b);
    if (stanFused_) { launch_stan_fused(0, 0, true); fusedParity_ ^= 1; } else reduce_pipeline(0, 0, 1);
    sync();                                               // warm
    const int64_t l0 = launches_;
    HIP_OK(hipEventRecord(evStart_, stream_));
    for (int i = 0; i < nEvals; ++i) { if (stanFused_) { launch_stan_fused(0, 0, true); fusedParity_ ^= 1; } else reduce_pipeline(0, 0, 1); }
    HIP_OK(hipEventRecord(evStop_, stream_));
    sync();
    float ms = 0; HIP_OK(hipEventElapsedTime(&ms, evStart_, evStop_));
    out[0] = ms * 1000.0 / nEvals; out[2] = (double)(launches_ - l0) / nEvals;
    std::vector<double> gX((size_t)K_ + 1), gZ((size_t)q_ + 1);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < nEvals; ++i) (void)leapfrog_sums(beta, b, gX.data(), gZ.data());
    out[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / nEvals;
  }
  // timing hook for bench.py: events on the stream the kernels are launched on
  hipStream_t stream() const { return stream_; }
  const BartArrays& arrays() const { return a_; }
  void sync() { HIP_OK(hipStreamSynchronize(stream_)); }

 private:
  // small arrays (everything the control kernel touches) are carved from one arena so that they share a few pages;
  // observation-length arrays get their own allocation
  template <class T> T* alloc(size_t count) {
    const size_t bytes = std::max<size_t>(16, count * sizeof(T));
    if (bytes <= ((size_t)2 << 20)) {
      const size_t need = (bytes + 255) / 256 * 256;
      if (!arena_ || arenaUsed_ + need > arenaSize_) {
        arenaSize_ = (size_t)32 << 20; arenaUsed_ = 0;
        void* q = nullptr; HIP_OK(hipMalloc(&q, arenaSize_)); allocs_.push_back(q); arena_ = (char*)q;
      }
      T* r = (T*)(arena_ + arenaUsed_); arenaUsed_ += need; return r;
    }
    void* p = nullptr; HIP_OK(hipMalloc(&p, bytes)); allocs_.push_back(p); return (T*)p;
  }
  template <class T> T* zalloc(size_t count) { T* p = alloc<T>(count); HIP_OK(hipMemsetAsync(p, 0, std::max<size_t>(16, count * sizeof(T)), stream_)); return p; }
  template <class T> void upload(T* dst, const T* src, size_t count) { if (count) HIP_OK(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, stream_)); }
  template <class T> void download(T* dst, const T* src, size_t count) { if (count) HIP_OK(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, stream_)); }

  void push_params(const double* beta, const double* b) {
    double* h = pinned_ + (1 + K_ + q_);
    // the previous async copy out of this staging area must have completed before we overwrite it
    sync();
    for (int k = 0; k < K_; ++k) h[k] = beta[k];
    for (int j = 0; j < q_; ++j) h[K_ + j] = b[j];
    if (K_ + q_) HIP_OK(hipMemcpyAsync(s_.params, h, (size_t)(K_ + q_) * 8, hipMemcpyHostToDevice, stream_));
  }
  void reduce_pipeline(int mode, int wantTrain, int direct) {
    hipLaunchKernelGGL(k_stan_inputs, dim3(gridN_), dim3(BLOCK), 0, stream_, a_, s_, mode, wantTrain, direct); ++launches_;
    const int fromE = (direct || a_.wts) ? 1 : 0;   // with weights the weighted residual lives in s.e in both modes
    for (int k0 = 3; k0 < K_; k0 += 4) { hipLaunchKernelGGL(k_xt_e, dim3(gridN_), dim3(BLOCK), 0, stream_, a_, s_, k0, fromE); ++launches_; }
    if (s_.numChunks) { hipLaunchKernelGGL(k_zt_chunks, dim3(s_.numChunks), dim3(BLOCK), 0, stream_, s_, fromE); ++launches_; }
    hipLaunchKernelGGL(k_stan_finalize, dim3(1), dim3(BLOCK), 0, stream_, a_, s_, gridN_); ++launches_;
  }
  void fetch_out(double* cX, double* cZ, double* s0) {
    HIP_OK(hipMemcpyAsync(pinned_, s_.out, (size_t)(1 + K_ + q_) * 8, hipMemcpyDeviceToHost, stream_));
    sync();
    *s0 = pinned_[0];
    for (int k = 0; k < K_; ++k) cX[k] = pinned_[1 + k];
    for (int j = 0; j < q_; ++j) cZ[j] = pinned_[1 + K_ + j];
  }
  void build_csc(const DevInit& d) {
    // column-major copy of Z with ascending rows inside a column, cut into fixed chunks of <= CHUNK entries
    const int CHUNK = 4096;
    std::vector<int64_t> colCount((size_t)q_ + 1, 0);
    for (int64_t e = 0; e < d.nnz; ++e) ++colCount[(size_t)d.v[e] + 1];
    for (int j = 0; j < q_; ++j) colCount[(size_t)j + 1] += colCount[(size_t)j];
    std::vector<int32_t> row((size_t)d.nnz); std::vector<double> val((size_t)d.nnz);
    std::vector<int64_t> fill(colCount.begin(), colCount.end() - 1);
    for (int64_t i = 0; i < n_; ++i) for (int e = d.u[i]; e < d.u[i + 1]; ++e) { int64_t pos = fill[(size_t)d.v[e]]++; row[(size_t)pos] = (int32_t)i; val[(size_t)pos] = d.w[e]; }
    std::vector<int32_t> chunkCol, chunkLen, colChunkPtr((size_t)q_ + 1, 0); std::vector<int64_t> chunkStart;
    for (int j = 0; j < q_; ++j) {
      colChunkPtr[(size_t)j] = (int32_t)chunkCol.size();
      for (int64_t st = colCount[(size_t)j]; st < colCount[(size_t)j + 1]; st += CHUNK) {
        chunkCol.push_back(j); chunkStart.push_back(st); chunkLen.push_back((int32_t)std::min<int64_t>(CHUNK, colCount[(size_t)j + 1] - st));
      }
    }
    colChunkPtr[(size_t)q_] = (int32_t)chunkCol.size();
    StanArrays& s = s_;
    s.numChunks = (int32_t)chunkCol.size();
    int32_t* dr = alloc<int32_t>(row.size()); upload(dr, row.data(), row.size()); s.cscRow = dr;
    double* dv = alloc<double>(val.size()); upload(dv, val.data(), val.size()); s.cscVal = dv;
    int32_t* cc = alloc<int32_t>(chunkCol.size()); upload(cc, chunkCol.data(), chunkCol.size()); s.chunkCol = cc;
    int64_t* cs = alloc<int64_t>(chunkStart.size()); upload(cs, chunkStart.data(), chunkStart.size()); s.chunkStart = cs;
    int32_t* cl = alloc<int32_t>(chunkLen.size()); upload(cl, chunkLen.data(), chunkLen.size()); s.chunkLen = cl;
    int32_t* cp = alloc<int32_t>(colChunkPtr.size()); upload(cp, colChunkPtr.data(), colChunkPtr.size()); s.colChunkPtr = cp;
    s.chunkPart = zalloc<double>((size_t)std::max(1, s.numChunks));
    HIP_OK(hipStreamSynchronize(stream_));   // host vectors go out of scope
  }

  int device_ = 0; hipStream_t stream_ = nullptr; hipEvent_t evStart_ = nullptr, evStop_ = nullptr;
  int64_t n_ = 0, nTest_ = 0; int P_ = 0, T_ = 0, nc_ = 0, K_ = 0, q_ = 0, gridN_ = 1; bool binary_ = false;
  size_t ldsApply_ = 0, ldsControl_ = 0, ldsTree_ = 0, ldsStep_ = 0; bool useFused_ = false, fusedAuto_ = false, fusedOk_ = false;
  enum { PATH_TWO = 1, PATH_FUSED = 2, PATH_SWEEP = 4, PATH_STREAM = 5 };
  bool splitProbs_ = false, weighted_ = false; double wScale_ = 1.0, wUnscale_ = 1.0; bool sweepOk_ = false, sweepRegsOk_ = false, sweepStreamOk_ = false, sweepSolo_ = false, sweepFew_ = false, sweepStream_ = false; unsigned long long* xbuf_ = nullptr; int32_t* sweepStatus_ = nullptr; int32_t* sweepStatusDev_ = nullptr;
  int64_t sweepCount_ = 0, sweepHandOvers_ = 0;
  int64_t sweepLaunchesRun_ = 0, sweepTreesInside_ = 0, sweepSpecSteps_ = 0, sweepSpecOk_ = 0;
  int xbufParity_ = 0; long long dbgSweepNo_ = 0; int sweepGrid_ = 0;
  OffsetArgs pend_; bool pendOffset_ = false, pendSigma_ = false;   // Stan -> BART hand-off calls held for the one-launch form (offset_from_params)
  int pathReq_ = 0, path_ = 0, sharing_ = 1;
  double dbgSweepMs_ = 0; int dbgSweeps_ = 0;
  hipGraph_t graph_ = nullptr; hipGraphExec_t graphExec_ = nullptr; int graphTrace_ = -1; bool useGraph_ = true;
  BartArrays a_; StanArrays s_;
  std::vector<void*> allocs_;
  char* arena_ = nullptr; size_t arenaSize_ = 0, arenaUsed_ = 0;
  double* pinned_ = nullptr; double* testOut_ = nullptr; double* latX_ = nullptr; double* testPinned_ = nullptr; bool testWanted_ = false, testQueued_ = false;
  int32_t* varDev_ = nullptr; int32_t* varPinned_ = nullptr; bool varWanted_ = false, varQueued_ = false;
  unsigned long long* fusedAcc_ = nullptr; int32_t* fusedBad_ = nullptr; unsigned long long* pinnedAcc_ = nullptr;
  size_t fusedLds_ = 0; int fusedParity_ = 0; bool stanFused_ = false;
  uint32_t fusedSeq_ = 0; int zFixed_ = -1; const double* inlineBeta_ = nullptr; const double* inlineB_ = nullptr;
  int fxExp_[3] = {0, 0, 0}; int64_t fusedEvals_ = 0, fusedFallbacks_ = 0, fxLastBad_ = -2; bool fxTinyFail_ = false;
  int64_t launches_ = 0;
};

#endif   // S4B_SWEEP_TU
}  // namespace s4b

#ifndef S4B_SWEEP_TU
#define S4B_DEV s4b::DevHip
#include "c_api.inc"
#endif
