// Development probe: symmetric ping-pong between workgroup 0 and workgroup `peer` through two 64-bit words in device memory
// (agent-scope relaxed atomics, no fences), both hops timed with the device-wide clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
typedef unsigned long long u64;
__device__ __forceinline__ u64 ld_agent(u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// variant 0: thread 0 only; 1: a __syncthreads() between seeing and answering (both sides); 2: the answer is stored by thread 1 (another lane of wave 0) after the barrier; 3: by thread 64 (another wave)
__global__ __launch_bounds__(512) void k_pp(u64* wa, u64* wb, u64* stat, int steps, int peer, int variant) {
  const int b = blockIdx.x;
  if (b != 0 && b != peer) return;
  __shared__ int seen;
  u64 lat = 0;
  const int answerer = variant == 2 ? 1 : variant == 3 ? 64 : 0;
  for (int st = 1; st <= steps; ++st) {
    const int off = variant == 4 ? 2048 : variant == 5 ? 16 : 0;   // 4 / 5: the words alternate between two places (16 KB / 128 B apart) by step parity
    u64* mine = (b == 0 ? wa : wb) + (st & 1) * off; u64* theirs = (b == 0 ? wb : wa) + ((b == 0 ? st - 1 : st) & 1) * off;
    const unsigned want = b == 0 ? st - 1 : st;       // block 0 waits for the answer to the previous step, then publishes step st
    if (threadIdx.x == 0) {
      seen = 1;
      if (want) { seen = 0; for (int g = 0; g < (1 << 22); ++g) { const u64 w = ld_agent(theirs); if ((unsigned)(w >> 32) >= want) { lat += (unsigned)((unsigned)wall_clock64() - (unsigned)w); seen = 1; break; } } }
    }
    if (variant && variant < 4) __syncthreads();
    if (variant && variant < 4 && !seen) return;
    if ((int)threadIdx.x == answerer) st_agent(mine, ((u64)(unsigned)st << 32) | (unsigned)wall_clock64());
  }
  if (threadIdx.x == 0) stat[b == 0 ? 0 : 1] = lat;
}
int main() {
  u64 *w, *stat; OK(hipMalloc(&w, 65536)); OK(hipMalloc(&stat, 64));
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  const int steps = 2000;
  for (int variant = 0; variant < 6; ++variant) for (int peer : {1, 8, 9, 255}) {
    OK(hipMemset(w, 0, 65536)); OK(hipMemset(stat, 0, 64));
    OK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_pp, dim3(256), dim3(512), 0, 0, w, w + 4096, stat, steps, peer, variant);
    OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
    float ms = 0; OK(hipEventElapsedTime(&ms, e0, e1));
    u64 s[2]; OK(hipMemcpy(s, stat, 16, hipMemcpyDeviceToHost));
    printf("variant %d, workgroup 0 <-> workgroup %3d: %.3f us per round trip; hop to 0: %.2f us, hop to peer: %.2f us\n", variant, peer, 1e3 * ms / steps, s[0] / 100.0 / (steps - 1), s[1] / 100.0 / steps);
  }
  return 0;
}
