// Development probe (round 4): the exchange step of a PERSISTENT tree sweep in which every workgroup decides redundantly.
//   G workgroups of 512 threads stay resident for `steps` steps.  Per step every workgroup
//     (pass)    all 8 waves busy for tPass ticks of the 100 MHz clock (stands for the O(N) pass over register-resident residuals),
//     (publish) lanes 0..W-1 of wave 3 store the workgroup's W partial words (f64) with sc1 stores into buffer step % 3,
//     (gather)  waves 3-7: lane r < G polls column r (W words, sc1 loads) until none is the sentinel, then the columns are
//               combined in a fixed order (transposed wave reduction, then waves in order) -> W totals in LDS,
//     (decide)  wave 0 waits for the totals, checks them against the closed form, stays busy for tDec ticks,
//     (barrier) __syncthreads.
//   No tags: a word is "there" when it is not the sentinel NaN; every workgroup re-arms its own words of buffer (step + 1) % 3 right
//   after a completed gather (everybody has left that buffer by then: a workgroup publishes step s only after gathering step s - 1)
//   and drains the store counter before its next publish.
//   variant 0: one hop, all-to-all.   variant 1: two hops — groups of GS workgroups gather each other, the group's first
//   workgroup publishes the group sum, everybody gathers the G / GS group sums.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
typedef unsigned long long u64;
constexpr u64 SENT = 0xFFF8DEADBEEF1234ull;
constexpr int GMAX = 256, WMAX = 32;
__device__ __forceinline__ u64 ld_agent(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void busy(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1); }
__device__ __forceinline__ double wsum(double v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o); return v; }

// buf: [3][WMAX][GMAX] words (column r = workgroup r); grp: [3][WMAX][GMAX / 2]
template <int W, int VARIANT>
__global__ __launch_bounds__(512) void k_sweep(u64* buf, u64* grp, u64* stat, unsigned* err, int steps, int tPass, int tDec, int GS) {
  extern __shared__ unsigned char dyn[];
  __shared__ double red[5][WMAX];
  __shared__ double tot[WMAX];
  __shared__ int arrived, bad;
  const int b = blockIdx.x, G = gridDim.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = (int)threadIdx.x - 192;     // reducer lane index (waves 3-7)
  if (threadIdx.x == 0) { arrived = 0; bad = 0; }
  __syncthreads();
  u64 tExch = 0;
  for (int st = 1; st <= steps; ++st) {
    u64* cur = buf + (size_t)(st % 3) * WMAX * GMAX;
    u64* nxt = buf + (size_t)((st + 1) % 3) * WMAX * GMAX;
    u64* gcur = grp + (size_t)(st % 3) * WMAX * (GMAX / 2);
    u64* gnxt = grp + (size_t)((st + 1) % 3) * WMAX * (GMAX / 2);
    busy(tPass);
    __syncthreads();                           // end of the pass (the block reduction's barrier)
    const long long t0 = wall_clock64();
    if (wv >= 3) {
      if (r < W) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-arming stores of two steps ago are out
        st_agent(cur + (size_t)r * GMAX + b, (u64)__double_as_longlong((double)(st % 1000) * 4.0 + (double)r + (double)b * 0.0078125));
      }
      double v[W];
      bool fine = true;
      if (VARIANT == 0) {
        const bool in = r < G;
        for (int g = 0; g < (1 << 16); ++g) {
          u64 w[W]; bool all = true;
#pragma unroll
          for (int k = 0; k < W; ++k) w[k] = in ? ld_agent(cur + (size_t)k * GMAX + r) : 0ull;
#pragma unroll
          for (int k = 0; k < W; ++k) { all = all && w[k] != SENT; v[k] = in ? __longlong_as_double((long long)w[k]) : 0.0; }
          if (all) break;
          if (g == (1 << 16) - 1) fine = false;
          __builtin_amdgcn_s_sleep(2);
        }
      } else {
        // hop 1: the GS columns of this workgroup's group, by the first GS reducer lanes; then the group sum
        const int g0 = (b / GS) * GS;
        const bool in = r < GS && g0 + r < G;
        for (int g = 0; g < (1 << 16); ++g) {
          u64 w[W]; bool all = true;
#pragma unroll
          for (int k = 0; k < W; ++k) w[k] = in ? ld_agent(cur + (size_t)k * GMAX + g0 + r) : 0ull;
#pragma unroll
          for (int k = 0; k < W; ++k) { all = all && w[k] != SENT; v[k] = in ? __longlong_as_double((long long)w[k]) : 0.0; }
          if (all) break;
          if (g == (1 << 16) - 1) fine = false;
          __builtin_amdgcn_s_sleep(2);
        }
        if (wv == 3) {
#pragma unroll
          for (int k = 0; k < W; ++k) v[k] = wsum(v[k]);
          if (b == g0 && lane == 0) {
#pragma unroll
            for (int k = 0; k < W; ++k) st_agent(gcur + (size_t)k * (GMAX / 2) + b / GS, (u64)__double_as_longlong(v[k]));
          }
        }
        // hop 2: all group sums
        const int NG = (G + GS - 1) / GS;
        const bool in2 = r < NG;
        for (int g = 0; g < (1 << 16); ++g) {
          u64 w[W]; bool all = true;
#pragma unroll
          for (int k = 0; k < W; ++k) w[k] = in2 ? ld_agent(gcur + (size_t)k * (GMAX / 2) + r) : 0ull;
#pragma unroll
          for (int k = 0; k < W; ++k) { all = all && w[k] != SENT; v[k] = in2 ? __longlong_as_double((long long)w[k]) : 0.0; }
          if (all) break;
          if (g == (1 << 16) - 1) fine = false;
          __builtin_amdgcn_s_sleep(2);
        }
      }
      if (!fine) { bad = 1; *err = 1; }
      // re-arm this workgroup's words of the buffer after next (everybody has left it)
      if (r < W) st_agent(nxt + (size_t)r * GMAX + b, SENT);
      if (VARIANT == 1 && b % GS == 0 && r < W) st_agent(gnxt + (size_t)r * (GMAX / 2) + b / GS, SENT);
#pragma unroll
      for (int k = 0; k < W; ++k) v[k] = wsum(v[k]);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < W; ++k) red[wv - 3][k] = v[k];
        __hip_atomic_fetch_add(&arrived, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else if (wv == 0) {
      for (int g = 0; g < (1 << 22); ++g) { if (__hip_atomic_load(&arrived, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= 5 * st) break; __builtin_amdgcn_s_sleep(1); }
      if (lane < W) {
        double s = red[0][lane]; for (int q = 1; q < 5; ++q) s += red[q][lane];
        const double want = (double)G * ((double)(st % 1000) * 4.0 + (double)lane) + 0.0078125 * (double)G * (double)(G - 1) * 0.5;
        if (s != want) *err = 2;
        tot[lane] = s;
      }
      if (lane == 0) tExch += (u64)(wall_clock64() - t0);
      busy(tDec);
    }
    __syncthreads();
    if (bad) return;
  }
  if (threadIdx.x == 0) stat[b] = tExch;
}


// variant 2: ORDER-FREE integer accumulation.  Every workgroup adds its W words (fixed-point fields + an arrival count in the top
// bits: field + (1 << 58)) into copy b % C of buffer step % 4 with agent-scope atomic adds that return nothing; a word is complete
// when its arrival field shows the number of workgroups that feed the copy.  Readers (ONE wave) poll C * W words — a few hundred
// instead of 255 * W.  Buffer (step + 2) % 4 is cleared by the first C workgroups after their gather of step `step` (dead since
// step - 1 was published by everybody; nobody adds to it before it has seen this workgroup's publish of step + 1, which waits for
// the clearing stores).
template <int W>
__global__ __launch_bounds__(512) void k_sweep_atomic(u64* buf, u64* stat, unsigned* err, int steps, int tPass, int tDec, int C, int strideWords, int pollSleep, int ws) {
  extern __shared__ unsigned char dyn[];
  __shared__ double tot[WMAX];
  __shared__ int arrived, bad;
  const int b = blockIdx.x, G = gridDim.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) { arrived = 0; bad = 0; }
  __syncthreads();
  u64 tExch = 0;
  const size_t bufWords = (size_t)C * strideWords;
  for (int st = 1; st <= steps; ++st) {
    u64* cur = buf + (size_t)(st & 3) * bufWords;
    u64* clr = buf + (size_t)((st + 2) & 3) * bufWords;
    busy(tPass);
    __syncthreads();
    const long long t0 = wall_clock64();
    if (wv == 3) {
      if (lane < W) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u64 field = (u64)((st % 1000) * 4 + lane) + (u64)b;
        __hip_atomic_fetch_add(cur + (size_t)(b % C) * strideWords + (size_t)lane * ws, field + (1ull << 58), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // poll: C * W words over 64 lanes
      constexpr int PER = (32 * W + 63) / 64;    // up to C = 32 copies
      u64 w[PER]; bool fine = false;
      for (int g = 0; g < (1 << 16); ++g) {
        bool all = true;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          const int e = lane + j * 64; const int c = e / W, k = e % W;
          const bool in = c < C;
          w[j] = in ? ld_agent(cur + (size_t)c * strideWords + (size_t)k * ws) : 0ull;
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          const int e = lane + j * 64; const int c = e / W;
          const bool in = c < C;
          const unsigned want = in ? (unsigned)((G - 1 - c) / C + 1) : 0u;     // workgroups b with b % C == c, b < G
          all = all && (!in || (unsigned)(w[j] >> 58) == want);
        }
        all = __all(all);
        if (all) { fine = true; break; }
        if (pollSleep == 2) __builtin_amdgcn_s_sleep(2); else if (pollSleep == 8) __builtin_amdgcn_s_sleep(8);
      }
      if (!fine) { bad = 1; *err = 1; }
      if (b < C && lane < W) st_agent(clr + (size_t)b * strideWords + (size_t)lane * ws, 0ull);
      // totals: word k = sum over copies
      double sums[PER];
#pragma unroll
      for (int j = 0; j < PER; ++j) { const int e = lane + j * 64; sums[j] = (e / W) < C ? (double)(w[j] & ((1ull << 58) - 1)) : 0.0; }
      // (LDS accumulate, order-free for integers that fit a double exactly)
      if (lane < W) tot[lane] = 0.0;
      __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
      for (int j = 0; j < PER; ++j) { const int e = lane + j * 64; if ((e / W) < C) atomicAdd(&tot[e % W], sums[j]); }
      __builtin_amdgcn_s_waitcnt(0);
      if (lane == 0) __hip_atomic_fetch_add(&arrived, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (wv == 0) {
      for (int g = 0; g < (1 << 22); ++g) { if (__hip_atomic_load(&arrived, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= st) break; __builtin_amdgcn_s_sleep(1); }
      if (lane < W) {
        const double want = (double)G * (double)((st % 1000) * 4 + lane) + (double)G * (double)(G - 1) * 0.5;
        if (tot[lane] != want) *err = 2;
      }
      if (lane == 0) tExch += (u64)(wall_clock64() - t0);
      busy(tDec);
    }
    __syncthreads();
    if (bad) return;
  }
  if (threadIdx.x == 0) stat[b] = tExch;
}
template <int W>
static int run_atomic(u64* buf, u64* stat, unsigned* err, int G, int steps, int tPass, int tDec, int C, int strideWords, int pollSleep, int ws = 1) {
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  OK(hipMemset(buf, 0, (size_t)4 * C * strideWords * 8));
  OK(hipMemset(stat, 0, GMAX * 8)); OK(hipMemset(err, 0, 64));
  OK(hipFuncSetAttribute((const void*)k_sweep_atomic<W>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  OK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_sweep_atomic<W>), dim3(G), dim3(512), 100 * 1024, 0, buf, stat, err, steps, tPass, tDec, C, strideWords, pollSleep, ws);
  OK(hipGetLastError());
  OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
  float ms = 0; OK(hipEventElapsedTime(&ms, e0, e1));
  u64 s[GMAX]; unsigned er; OK(hipMemcpy(s, stat, sizeof s, hipMemcpyDeviceToHost)); OK(hipMemcpy(&er, err, 4, hipMemcpyDeviceToHost));
  double ex = 0, mx = 0; for (int i = 0; i < G; ++i) { ex += (double)s[i]; if ((double)s[i] > mx) mx = (double)s[i]; }
  const double busyUs = (tPass + tDec) / 100.0;
  printf("variant 2 (integer atomics), W = %2d words, %2d copies, copy stride %5d words, word stride %2d, poll sleep %2d, busy %.1f + %.1f us: %.3f us per step = busy + %.3f;  pass-end -> totals in wave 0: mean %.2f, slowest %.2f us;  err %u\n",
         W, C, strideWords, ws, pollSleep, tPass / 100.0, tDec / 100.0, 1e3 * ms / steps, 1e3 * ms / steps - busyUs, ex / G / steps / 100.0, mx / steps / 100.0, er);
  return 0;
}

template <int W, int V>
static int run(u64* buf, u64* grp, u64* stat, unsigned* err, int G, int steps, int tPass, int tDec, int GS) {
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  const size_t nb = (size_t)3 * WMAX * GMAX, ng = (size_t)3 * WMAX * (GMAX / 2);
  u64* h = (u64*)malloc(nb * 8); for (size_t i = 0; i < nb; ++i) h[i] = SENT;
  OK(hipMemcpy(buf, h, nb * 8, hipMemcpyHostToDevice)); OK(hipMemcpy(grp, h, ng * 8, hipMemcpyHostToDevice)); free(h);
  OK(hipMemset(stat, 0, GMAX * 8)); OK(hipMemset(err, 0, 64));
  OK(hipFuncSetAttribute((const void*)k_sweep<W, V>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  OK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_sweep<W, V>), dim3(G), dim3(512), 100 * 1024, 0, buf, grp, stat, err, steps, tPass, tDec, GS);
  OK(hipGetLastError());
  OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
  float ms = 0; OK(hipEventElapsedTime(&ms, e0, e1));
  u64 s[GMAX]; unsigned er; OK(hipMemcpy(s, stat, sizeof s, hipMemcpyDeviceToHost)); OK(hipMemcpy(&er, err, 4, hipMemcpyDeviceToHost));
  double ex = 0, mx = 0; for (int i = 0; i < G; ++i) { ex += (double)s[i]; if ((double)s[i] > mx) mx = (double)s[i]; }
  const double busyUs = (tPass + tDec) / 100.0;
  printf("variant %d (%s), W = %2d words, GS = %2d, busy %.1f + %.1f us: %.3f us per step = busy + %.3f;  pass-end -> totals in wave 0: mean %.2f, slowest workgroup %.2f us;  err %u\n",
         V, V == 0 ? "one hop all-to-all" : "two hops", W, GS, tPass / 100.0, tDec / 100.0, 1e3 * ms / steps, 1e3 * ms / steps - busyUs,
         ex / G / steps / 100.0, mx / steps / 100.0, er);
  return 0;
}

int main(int argc, char** argv) {
  u64 *buf, *grp, *stat; unsigned* err;
  OK(hipMalloc(&buf, (size_t)8 << 20)); OK(hipMalloc(&grp, (size_t)3 * WMAX * (GMAX / 2) * 8)); OK(hipMalloc(&stat, GMAX * 8)); OK(hipMalloc(&err, 64));
  const int steps = 3000, G = 256;
  const bool all = argc > 1 && argv[1][0] == 'a';
  if (argc > 1 && argv[1][0] == 'w') {     // words of one copy on separate cache lines (word stride 16 = 128 B, 32 = 256 B)
    for (int rep = 0; rep < 2; ++rep) for (int busyOn = 0; busyOn < 2; ++busyOn) {
      const int tp = busyOn ? 230 : 0, td = busyOn ? 400 : 0;
      for (int C : {4, 8, 16}) for (int ws : {1, 16, 32}) {
        if (run_atomic<6>(buf, stat, err, G, steps, tp, td, C, 32 * ws + 16, 2, ws)) return 1;
        if (run_atomic<12>(buf, stat, err, G, steps, tp, td, C, 32 * ws + 16, 2, ws)) return 1;
        if (run_atomic<24>(buf, stat, err, G, steps, tp, td, C, 32 * ws + 16, 2, ws)) return 1;
      }
    }
    return 0;
  }
  for (int rep = 0; rep < 2; ++rep) {
    for (int busyOn = 0; busyOn < 2; ++busyOn) {
      const int tp = busyOn ? 230 : 0, td = busyOn ? 400 : 0;
      if (all) {
        if (run<8, 0>(buf, grp, stat, err, G, steps, tp, td, 16)) return 1;
        if (run<16, 0>(buf, grp, stat, err, G, steps, tp, td, 16)) return 1;
        if (run<32, 0>(buf, grp, stat, err, G, steps, tp, td, 16)) return 1;
        if (run<8, 1>(buf, grp, stat, err, G, steps, tp, td, 16)) return 1;
        if (run<16, 1>(buf, grp, stat, err, G, steps, tp, td, 32)) return 1;
      }
      for (int C : {8, 16, 32}) for (int stride : {64, 512, 8192}) {
        if (run_atomic<12>(buf, stat, err, G, steps, tp, td, C, stride, 2)) return 1;
        if (run_atomic<24>(buf, stat, err, G, steps, tp, td, C, stride, 2)) return 1;
      }
      if (run_atomic<24>(buf, stat, err, G, steps, tp, td, 8, 512, 8)) return 1;
      if (run_atomic<24>(buf, stat, err, G, steps, tp, td, 8, 512, 0)) return 1;
      if (run_atomic<32>(buf, stat, err, G, steps, tp, td, 8, 512, 2)) return 1;
    }
  }
  return 0;
}
