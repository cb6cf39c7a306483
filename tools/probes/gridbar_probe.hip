// Development probe: a grid barrier inside a persistent kernel that is also a coherence point (what a kernel boundary gives):
//   every workgroup writes 32 doubles with plain stores, release fence (write back the XCD's L2), arrives; waits for everybody;
//   acquire fence (invalidate L1 / L2); then EVERY workgroup reads everybody's 32 doubles with plain loads (64 KB) and checks them.
// variants: 0 arrivals polled all-to-all (each workgroup polls the 256 flag words itself); 1 arrivals gathered by workgroup 0 which
// then publishes one word everybody polls; 2 like 0 without the data (barrier + fences only); 3 like 0 without fences and data
#include <hip/hip_runtime.h>
#include <cstdio>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
typedef unsigned long long u64;
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int V>
__global__ __launch_bounds__(512) void k_bar(unsigned* flags, unsigned* pub, double* data, unsigned* err, int steps) {
  const int b = blockIdx.x, G = gridDim.x;
  __shared__ int ok;
  double acc = 0.0;
  for (int st = 1; st <= steps; ++st) {
    double* mine = data + (size_t)(st & 1) * 256 * 32;
    if (V < 2 && threadIdx.x < 32) mine[b * 32 + threadIdx.x] = (double)(st * 1000 + b + threadIdx.x);
    __syncthreads();
    if (threadIdx.x < 64) {
      if (V != 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (threadIdx.x == 0) st_agent(flags + b, (unsigned)st);
      bool fine = true;
      if (V == 1) {
        if (b == 0) {
          for (int k = threadIdx.x; k < G; k += 64) { bool f = false; for (int g = 0; g < (1 << 20); ++g) if (ld_agent(flags + k) >= (unsigned)st) { f = true; break; } fine = fine && f; }
          fine = __all(fine);
          if (threadIdx.x == 0) st_agent(pub, (unsigned)st);
        } else if (threadIdx.x == 0) { fine = false; for (int g = 0; g < (1 << 22); ++g) if (ld_agent(pub) >= (unsigned)st) { fine = true; break; } }
      } else {
        for (int k = threadIdx.x; k < G; k += 64) { bool f = false; for (int g = 0; g < (1 << 20); ++g) if (ld_agent(flags + k) >= (unsigned)st) { f = true; break; } fine = fine && f; }
      }
      fine = __all(fine);
      if (V != 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (threadIdx.x == 0) { ok = fine ? 1 : 0; if (!fine) *err = 1; }
    }
    __syncthreads();
    if (!ok) return;
    if (V < 2) {
      // every workgroup reads everybody's 32 doubles: G * 32 doubles over 512 threads
      for (int i = threadIdx.x; i < G * 32; i += 512) { const double v = mine[i]; if (v != (double)(st * 1000 + (i >> 5) + (i & 31))) *err = 2; acc += v; }
    }
  }
  if (acc == 1.2345) *err = 3;
}
int main() {
  unsigned *flags, *pub, *err; double* data; OK(hipMalloc(&flags, 4096)); OK(hipMalloc(&pub, 256)); OK(hipMalloc(&err, 64)); OK(hipMalloc(&data, 2 * 256 * 32 * 8));
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  const int steps = 2000;
  for (int v = 0; v < 4; ++v) for (int rep = 0; rep < 2; ++rep) {
    OK(hipMemset(flags, 0, 4096)); OK(hipMemset(pub, 0, 256)); OK(hipMemset(err, 0, 64));
    OK(hipEventRecord(e0, 0));
    if (v == 0) hipLaunchKernelGGL(k_bar<0>, dim3(256), dim3(512), 0, 0, flags, pub, data, err, steps);
    else if (v == 1) hipLaunchKernelGGL(k_bar<1>, dim3(256), dim3(512), 0, 0, flags, pub, data, err, steps);
    else if (v == 2) hipLaunchKernelGGL(k_bar<2>, dim3(256), dim3(512), 0, 0, flags, pub, data, err, steps);
    else hipLaunchKernelGGL(k_bar<3>, dim3(256), dim3(512), 0, 0, flags, pub, data, err, steps);
    OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
    float ms = 0; OK(hipEventElapsedTime(&ms, e0, e1));
    unsigned er; OK(hipMemcpy(&er, err, 4, hipMemcpyDeviceToHost));
    printf("variant %d: %.3f us per step, err %u\n", v, 1e3 * ms / steps, er);
  }
  return 0;
}
