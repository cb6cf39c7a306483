// Development probe: one grid-wide step of a persistent kernel with a tight gather.
//   control (workgroup 0) publishes step s (one tagged word); the P = grid-1 pass workgroups poll it, then each stores W tagged
//   64-bit words (payload: the device clock | step << 32); the control polls all P*W words with coalesced agent-scope loads,
//   every thread W*P/512 words, until every tag shows step s, and notes  now - latest payload clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
typedef unsigned long long u64;
__device__ __forceinline__ u64 ld_agent(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int PER>    // words per control thread
__global__ __launch_bounds__(512) void k_gather(u64* pub, u64* words, u64* stat, unsigned* err, int steps, int W) {
  const int b = blockIdx.x, P = gridDim.x - 1, total = P * W;
  __shared__ unsigned red[8]; __shared__ int ok;
  u64 lat = 0, lat2 = 0;
  for (int st = 1; st <= steps; ++st) {
    if (b == 0) {
      if (threadIdx.x == 0) st_agent(pub, ((u64)(unsigned)st << 32) | (unsigned)wall_clock64());
      const u64* base = words + (size_t)(st & 1) * 65536;
      unsigned mx = 0; bool all = false;
      for (int g = 0; g < (1 << 18) && !all; ++g) {
        u64 w[PER];
        #pragma unroll
        for (int k = 0; k < PER; ++k) { const int e = threadIdx.x + k * 512; w[k] = e < total ? ld_agent(base + e) : ((u64)(unsigned)st << 32); }
        all = true; mx = 0;
        #pragma unroll
        for (int k = 0; k < PER; ++k) { all = all && (unsigned)(w[k] >> 32) == (unsigned)st; mx = max(mx, (unsigned)w[k]); }
      }
      if (!all) *err = 1;
      const unsigned t1 = (unsigned)wall_clock64();
      for (int o = 32; o; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
      __syncthreads();
      if (threadIdx.x == 0) { unsigned m = 0; for (int i = 0; i < 8; ++i) m = max(m, red[i]); lat += (unsigned)((unsigned)wall_clock64() - m); lat2 += (unsigned)(t1 - m); }
      __syncthreads();
    } else {
      if (threadIdx.x == 0) { ok = 0; for (int g = 0; g < (1 << 22); ++g) { const u64 w = ld_agent(pub); if ((unsigned)(w >> 32) >= (unsigned)st) { ok = 1; lat += (unsigned)((unsigned)wall_clock64() - (unsigned)w); break; } } if (!ok) *err = 1; }
      __syncthreads();
      if (!ok) return;
      if ((int)threadIdx.x < W) st_agent(words + (size_t)(st & 1) * 65536 + (size_t)(b - 1) * W + threadIdx.x, ((u64)(unsigned)st << 32) | (unsigned)wall_clock64());
    }
  }
  if (threadIdx.x == 0) { stat[b * 2] = lat; stat[b * 2 + 1] = lat2; }
}
int main() {
  u64 *pub, *words, *stat; unsigned* err; OK(hipMalloc(&pub, 256)); OK(hipMalloc(&words, 2 * 65536 * 8)); OK(hipMalloc(&stat, 256 * 16)); OK(hipMalloc(&err, 64));
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  const int steps = 2000;
  for (int W : {1, 16, 32, 48}) for (int G : {256, 128}) {
    OK(hipMemset(pub, 0, 256)); OK(hipMemset(words, 0, 2 * 65536 * 8)); OK(hipMemset(stat, 0, 256 * 16)); OK(hipMemset(err, 0, 64));
    OK(hipEventRecord(e0, 0));
    if (W == 1) hipLaunchKernelGGL(k_gather<1>, dim3(G), dim3(512), 0, 0, pub, words, stat, err, steps, W);
    else if (W == 16) hipLaunchKernelGGL(k_gather<8>, dim3(G), dim3(512), 0, 0, pub, words, stat, err, steps, W);
    else if (W == 32) hipLaunchKernelGGL(k_gather<16>, dim3(G), dim3(512), 0, 0, pub, words, stat, err, steps, W);
    else hipLaunchKernelGGL(k_gather<24>, dim3(G), dim3(512), 0, 0, pub, words, stat, err, steps, W);
    OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
    float ms = 0; OK(hipEventElapsedTime(&ms, e0, e1));
    u64 s[512]; unsigned er; OK(hipMemcpy(s, stat, sizeof s, hipMemcpyDeviceToHost)); OK(hipMemcpy(&er, err, 4, hipMemcpyDeviceToHost));
    double bc = 0; for (int i = 1; i < G; ++i) bc += (double)s[2 * i];
    printf("grid %3d, %2d words per pass workgroup: %.3f us per step; broadcast hop %.2f us; gather hop %.2f us (%.2f to the last tag seen); err %u\n", G, W, 1e3 * ms / steps, bc / (G - 1) / steps / 100.0, s[0] / 100.0 / steps, s[1] / 100.0 / steps, er);
  }
  return 0;
}
