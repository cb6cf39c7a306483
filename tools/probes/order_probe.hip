// Development probe (litmus test): flag-after-data without a fence.  Writer workgroups store 256 words with agent-scope (sc1) relaxed
// stores, wait for their store counter (s_waitcnt vmcnt(0)), then publish a flag with another sc1 store.  Reader workgroups on other
// XCDs poll the flag and then read the 256 words with sc1 loads: every word must show the step the flag shows.  Counts violations.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
__device__ __forceinline__ unsigned ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// pair k: writer = workgroup 2k, reader = workgroup 2k + 1 (adjacent workgroups sit on different XCDs); NOWAIT: no s_waitcnt before the flag
template <bool NOWAIT>
__global__ __launch_bounds__(256) void k_order(unsigned* data, unsigned* flag, unsigned* back, unsigned* bad, int steps) {
  const int pair = blockIdx.x >> 1; const bool writer = (blockIdx.x & 1) == 0;
  unsigned* d = data + (size_t)pair * 4096; unsigned* f = flag + pair * 64; unsigned* b = back + pair * 64;
  __shared__ int ok;
  for (int s = 1; s <= steps; ++s) {
    if (writer) {
      // wait until the reader has consumed step s - 1
      if (threadIdx.x == 0) { ok = 0; for (int g = 0; g < (1 << 24); ++g) if (ld(b) >= (unsigned)(s - 1)) { ok = 1; break; } }
      __syncthreads();
      if (!ok) { if (threadIdx.x == 0) atomicAdd(bad + 1, 1u); return; }
      st(d + threadIdx.x * 16, (unsigned)s);          // 256 words on 256 different 64-byte lines
      if (!NOWAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) st(f, (unsigned)s);
    } else {
      if (threadIdx.x == 0) { ok = 0; for (int g = 0; g < (1 << 24); ++g) if (ld(f) >= (unsigned)s) { ok = 1; break; } }
      __syncthreads();
      if (!ok) { if (threadIdx.x == 0) atomicAdd(bad + 1, 1u); return; }
      const unsigned v = ld(d + threadIdx.x * 16);
      if (v != (unsigned)s) atomicAdd(bad, 1u);
      __syncthreads();
      if (threadIdx.x == 0) st(b, (unsigned)s);
    }
  }
}
int main() {
  unsigned *data, *flag, *back, *bad; OK(hipMalloc(&data, 128 * 4096 * 4)); OK(hipMalloc(&flag, 128 * 256)); OK(hipMalloc(&back, 128 * 256)); OK(hipMalloc(&bad, 64));
  for (int v = 0; v < 2; ++v) {
    OK(hipMemset(data, 0, 128 * 4096 * 4)); OK(hipMemset(flag, 0, 128 * 256)); OK(hipMemset(back, 0, 128 * 256)); OK(hipMemset(bad, 0, 64));
    const int steps = 100000;
    if (v == 0) hipLaunchKernelGGL(k_order<false>, dim3(256), dim3(256), 0, 0, data, flag, back, bad, steps);
    else hipLaunchKernelGGL(k_order<true>, dim3(256), dim3(256), 0, 0, data, flag, back, bad, steps);
    OK(hipDeviceSynchronize());
    unsigned h[2]; OK(hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost));
    printf("%s: 128 writer/reader pairs x %d steps x 256 words: %u stale words, %u timeouts\n", v == 0 ? "s_waitcnt vmcnt(0) before the flag" : "no wait before the flag (barrier only)", steps, h[0], h[1]);
  }
  return 0;
}
