// Does a ping-pong residual (read A, write B) stream faster than the in-place update k_tree / k_step / k_apply do today?  (round 6; MI355X)
//   hipcc --offload-arch=gfx950 -O3 -o copy_probe copy_probe.hip && ./copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int BLOCK = 256;
__global__ __launch_bounds__(BLOCK) void k_inplace(double2* __restrict__ x, long n2) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n2; i += (long)gridDim.x * BLOCK) { double2 v = x[i]; v.x += 1.0; v.y -= 1.0; x[i] = v; }
}
__global__ __launch_bounds__(BLOCK) void k_copy(const double2* __restrict__ x, double2* __restrict__ y, long n2) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n2; i += (long)gridDim.x * BLOCK) { double2 v = x[i]; v.x += 1.0; v.y -= 1.0; y[i] = v; }
}
__global__ __launch_bounds__(BLOCK) void k_copy_nt(const double2* __restrict__ x, double2* __restrict__ y, long n2) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n2; i += (long)gridDim.x * BLOCK) {
    double2 v = x[i]; v.x += 1.0; v.y -= 1.0; __builtin_nontemporal_store(v.x, &y[i].x); __builtin_nontemporal_store(v.y, &y[i].y); }
}
// the tree kernel's mix: 8 B residual read + written, 2 + 2 B leaf ids read, 2 B predictor read (22 B per observation)
__global__ __launch_bounds__(BLOCK) void k_mix(double2* __restrict__ r, const double2* __restrict__ rin, const unsigned* __restrict__ a, const unsigned* __restrict__ b, const unsigned* __restrict__ c, long n2, int inplace) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n2; i += (long)gridDim.x * BLOCK) {
    double2 v = inplace ? r[i] : rin[i]; const unsigned u = a[i] + b[i] + c[i]; v.x += (double)(u & 1u); v.y -= 1.0; r[i] = v; }
}
int main() {
  const long n2 = 1L << 26;   // 1 GiB of double2
  double2 *x, *y; unsigned *a, *b, *c;
  OK(hipMalloc(&x, n2 * 16)); OK(hipMalloc(&y, n2 * 16)); OK(hipMalloc(&a, n2 * 4)); OK(hipMalloc(&b, n2 * 4)); OK(hipMalloc(&c, n2 * 4));
  OK(hipMemset(x, 0, n2 * 16)); OK(hipMemset(y, 0, n2 * 16)); OK(hipMemset(a, 0, n2 * 4)); OK(hipMemset(b, 0, n2 * 4)); OK(hipMemset(c, 0, n2 * 4));
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  for (int grid : {1024, 2048, 4096}) {
    for (int which = 0; which < 5; ++which) {
      float best = 1e30f;
      for (int r = 0; r < 6; ++r) {
        OK(hipEventRecord(e0));
        if (which == 0) hipLaunchKernelGGL(k_inplace, dim3(grid), dim3(BLOCK), 0, 0, x, n2);
        else if (which == 1) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(BLOCK), 0, 0, (r & 1) ? y : x, (r & 1) ? x : y, n2);
        else if (which == 2) hipLaunchKernelGGL(k_copy_nt, dim3(grid), dim3(BLOCK), 0, 0, (r & 1) ? y : x, (r & 1) ? x : y, n2);
        else if (which == 3) hipLaunchKernelGGL(k_mix, dim3(grid), dim3(BLOCK), 0, 0, x, x, a, b, c, n2, 1);
        else hipLaunchKernelGGL(k_mix, dim3(grid), dim3(BLOCK), 0, 0, (r & 1) ? x : y, (r & 1) ? y : x, a, b, c, n2, 0);
        OK(hipEventRecord(e1)); OK(hipEventSynchronize(e1));
        float ms; OK(hipEventElapsedTime(&ms, e0, e1)); if (r > 0 && ms < best) best = ms;
      }
      const double bytes = which < 3 ? n2 * 32.0 : n2 * 44.0;
      const char* nm[5] = {"in place (read + write one buffer)", "ping-pong (read A, write B)", "ping-pong, non-temporal stores", "tree-kernel mix, residual in place", "tree-kernel mix, residual ping-pong"};
      printf("grid %4d  %-38s %7.3f ms  %7.1f GB/s\n", grid, nm[which], best, bytes / (best * 1e-3) / 1e9);
    }
  }
  return 0;
}
