// Microbenchmark (development probe, not part of the product): the two hops of a grid-wide step inside a persistent kernel,
// timed separately with the device-wide 100 MHz clock, with ONE CONTROL WORKGROUP PER XCD (blocks 0..7; workgroups are dealt
// round-robin to the 8 XCDs, checked with XCC_ID) and 248 pass workgroups.
//   broadcast: control c publishes (step, its clock); a pass workgroup polls ONE control and notes  now - published
//   gather:    every pass workgroup writes 16 doubles and publishes (step, its clock) on its own line; every control polls all
//              248 lines, reads the 248 x 16 doubles, and notes  now - latest published
// modes: 0 pass listens to the control of its own XCD, agent-scope atomics (sc1)
//        1 pass listens to the control of the next XCD, agent-scope atomics
//        2 own XCD, the poll is  global_load sc0  (miss the CU's cache, hit the XCD's L2), the publish a plain store
//        3 everybody listens to control 0 (the layout of handshake_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
typedef unsigned long long u64;
struct Sh { u64* pub; u64* flags; double* part; unsigned* err; u64* stat; unsigned* xcc; };
constexpr int TS = 40;
constexpr u64 TMASK = (1ull << TS) - 1;

__device__ __forceinline__ u64 ld_agent(u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_l2(u64* p) { u64 v; asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void st_l2(u64* p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory"); }

template <int MODE>
__global__ __launch_bounds__(512) void k_probe(Sh s, int steps, int NC, int W, int S) {
  const int b = blockIdx.x, P = gridDim.x - NC;
  const bool ctl = b < NC;
  __shared__ int ok; __shared__ u64 tmax;
  if (threadIdx.x == 0) { ok = 1; unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); s.xcc[b] = x & 15; }
  __syncthreads();
  double acc = 0.0; u64 latSum = 0;
  const int src = MODE == 3 ? 0 : MODE == 1 ? ((b + 1) % NC) : (b % NC);
  for (int st = 1; st <= steps; ++st) {
    if (ctl) {
      if (st > 1 && MODE == 4) {
        // tagged words: word k of workgroup g = (payload32 | step << 32); 32 words per workgroup; all threads poll their share
        if (threadIdx.x == 0) tmax = 0;
        __syncthreads();
        const int total = P * S; bool fine = true; unsigned mx = 0;
        u64* base = s.flags + (size_t)((st - 1) & 1) * 256 * 32;
        for (int k0 = threadIdx.x; k0 < total; k0 += 512 * 8) {
          u64 w[8]; unsigned need = 0;
          #pragma unroll
          for (int k = 0; k < 8; ++k) { const int e = k0 + k * 512; if (e < total && e % S < W) need |= 1u << k; }
          for (int guard = 0; need && guard < (1 << 20); ++guard) {
            #pragma unroll
            for (int k = 0; k < 8; ++k) if (need >> k & 1) w[k] = ld_agent(base + NC * S + k0 + k * 512);
            #pragma unroll
            for (int k = 0; k < 8; ++k) if ((need >> k & 1) && (unsigned)(w[k] >> 32) == (unsigned)(st - 1)) { need &= ~(1u << k); acc += (double)(unsigned)w[k]; if (((k0 + k * 512) % S) == 0) mx = max(mx, (unsigned)w[k]); }
          }
          if (need) { fine = false; *s.err = 1; }
        }
        atomicMax(&tmax, (u64)mx);
        if (!__syncthreads_and(fine ? 1 : 0)) return;
        if (threadIdx.x == 0) latSum += (unsigned)((unsigned)wall_clock64() - (unsigned)tmax);
      } else
      if (st > 1) {
        if (threadIdx.x == 0) tmax = 0;
        __syncthreads();
        bool fine = true;
        if ((int)threadIdx.x < P) {
          fine = false;
          for (int guard = 0; guard < (1 << 22); ++guard) {
            const u64 w = ld_agent(s.flags + (size_t)(NC + threadIdx.x) * 16);
            if ((w >> TS) >= (u64)(st - 1)) { fine = true; atomicMax(&tmax, w & TMASK); break; }
          }
          if (!fine) *s.err = 1;
        }
        if (!__syncthreads_and(fine ? 1 : 0)) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if ((int)threadIdx.x < P) for (int k = 0; k < 16; ++k) acc += ld_agent((u64*)(s.part + ((size_t)((st - 1) & 1) * 256 + NC + threadIdx.x) * 16 + k)) ? 1.0 : 0.0;
        __syncthreads();
        if (threadIdx.x == 0) latSum += (wall_clock64() & TMASK) - tmax;
      }
      if (threadIdx.x == 0) { const u64 w = ((u64)st << TS) | (wall_clock64() & TMASK); if (MODE == 2) st_l2(s.pub + b * 16, w); else st_agent(s.pub + b * 16, w); }
    } else {
      if (threadIdx.x == 0) {
        ok = 0;
        for (int guard = 0; guard < (1 << 22); ++guard) {
          const u64 w = MODE == 2 ? ld_l2(s.pub + src * 16) : ld_agent(s.pub + src * 16);
          if ((w >> TS) >= (u64)st) { ok = 1; latSum += (wall_clock64() & TMASK) - (w & TMASK); break; }
        }
        if (!ok) *s.err = 1;
      }
      __syncthreads();
      if (!ok) return;
      if (MODE == 4) {
        if ((int)threadIdx.x < W) st_agent(s.flags + (size_t)(st & 1) * 256 * 32 + (size_t)b * S + threadIdx.x, ((u64)(unsigned)st << 32) | (unsigned)wall_clock64());
        continue;
      }
      if (threadIdx.x < 16) st_agent((u64*)(s.part + ((size_t)(st & 1) * 256 + b) * 16 + threadIdx.x), (u64)__double_as_longlong((double)(st + threadIdx.x)));
      __syncthreads();
      if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); st_agent(s.flags + (size_t)b * 16, ((u64)st << TS) | (wall_clock64() & TMASK)); }
    }
  }
  if (threadIdx.x == 0) s.stat[b] = latSum;
  if (ctl && threadIdx.x == 0) s.part[b] = acc;
}

int main() {
  Sh s; OK(hipMalloc(&s.pub, 8 * 128)); OK(hipMalloc(&s.flags, 2 * 256 * 32 * 8 + 256 * 128)); OK(hipMalloc(&s.part, 2 * 256 * 16 * 8)); OK(hipMalloc(&s.err, 256)); OK(hipMalloc(&s.stat, 256 * 8)); OK(hipMalloc(&s.xcc, 256 * 4));
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  for (int mode = 11; mode < 19; ++mode) for (int rep = 0; rep < 2; ++rep) {
    OK(hipMemset(s.pub, 0, 8 * 128)); OK(hipMemset(s.flags, 0, 2 * 256 * 32 * 8 + 256 * 128)); OK(hipMemset(s.err, 0, 256)); OK(hipMemset(s.part, 0, 2 * 256 * 16 * 8)); OK(hipMemset(s.stat, 0, 256 * 8));
    const int steps = 2000;
    const int NC = mode == 3 || mode == 5 || mode >= 7 ? 1 : 8; const int S = mode >= 11 ? 16 : 0; const int GR = mode == 11 ? 256 : mode == 12 ? 129 : mode == 13 ? 65 : mode == 14 ? 33 : mode == 15 ? 9 : mode == 16 ? 2 : 256; const int W0 = mode >= 11 && mode <= 16 ? 1 : mode == 17 ? 8 : mode == 18 ? 16 : 0; const int W = W0 ? W0 : mode == 7 ? 1 : mode == 8 ? 4 : mode == 9 ? 16 : mode == 10 ? 2 : 32;
    OK(hipEventRecord(e0, 0));
    if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(256), dim3(512), 0, 0, s, steps, NC, W, S ? S : W);
    else if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(256), dim3(512), 0, 0, s, steps, NC, W, S ? S : W);
    else if (mode == 2) continue;
    else if (mode == 3) hipLaunchKernelGGL(k_probe<3>, dim3(256), dim3(512), 0, 0, s, steps, NC, W, S ? S : W);
    else if (mode == 6) hipLaunchKernelGGL(k_probe<4>, dim3(128), dim3(512), 0, 0, s, steps, NC, W, S ? S : W);
    else hipLaunchKernelGGL(k_probe<4>, dim3(GR), dim3(512), 0, 0, s, steps, NC, W, S ? S : W);
    OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
    float ms = 0; OK(hipEventElapsedTime(&ms, e0, e1));
    unsigned err = 0; OK(hipMemcpy(&err, s.err, 4, hipMemcpyDeviceToHost));
    u64 stat[256]; unsigned xcc[256]; OK(hipMemcpy(stat, s.stat, sizeof stat, hipMemcpyDeviceToHost)); OK(hipMemcpy(xcc, s.xcc, sizeof xcc, hipMemcpyDeviceToHost));
    double g = 0, bc = 0; const int G = mode == 6 ? 128 : GR; for (int i = 0; i < NC; ++i) g += (double)stat[i]; for (int i = NC; i < G; ++i) bc += (double)stat[i];
    int same = 0; for (int i = 8; i < 256; ++i) same += xcc[i] == xcc[i & 7];
    printf("mode %d (%d control, %s gather, %d words, grid %d): %.3f us per step, broadcast hop %.2f us, gather hop %.2f us (poll + read 248 x 16 doubles), err %u; pass workgroups on the XCD of control (b & 7): %d of 248; XCDs of blocks 0..9:", mode, NC, mode >= 4 ? "tagged-word" : "flag + fence + 16 loads", W, G,
           1e3 * ms / steps, bc / (G - NC) / steps / 100.0, g / NC / (steps - 1) / 100.0, err, same);
    for (int i = 0; i < 10; ++i) printf(" %u", xcc[i]);
    printf("\n");
  }
  return 0;
}
