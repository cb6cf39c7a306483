// Microbenchmark (development probe): a hipGraph of two kernel chains with cross dependencies —
//   P_j (255 workgroups, busy ~tp us) and C_j (1 workgroup, busy ~tc us);  edges P_{j-1}->P_j, C_{j-1}->C_j, P_{j-1}->C_j, C_{j-1}->P_j
// against ONE chain of 256-workgroup launches busy for max(tp, tc).  Question: what does a step cost when the O(N) pass and the
// control code of a tree update are separate kernels that run side by side?
#include <hip/hip_runtime.h>
#include <cstdio>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
__global__ void k_busy(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1);
}
int main() {
  int* sink; OK(hipMalloc(&sink, 4)); OK(hipMemset(sink, 0, 4));
  hipStream_t sa, sb; OK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); OK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const int steps = 200;
  for (int variant = 0; variant < 3; ++variant) {
    const double tp = variant == 2 ? 50.0 : 7.0, tc = 8.0;   // us
    hipGraph_t g; hipGraphExec_t ge;
    OK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
    if (variant == 0) {
      for (int j = 0; j < steps; ++j) hipLaunchKernelGGL(k_busy, dim3(256), dim3(512), 0, sa, (long long)(100 * (tp > tc ? tp : tc)), sink);
    } else {
      hipEvent_t eFork; OK(hipEventCreate(&eFork));
      OK(hipEventRecord(eFork, sa)); OK(hipStreamWaitEvent(sb, eFork, 0));
      hipEvent_t ep, ec;
      for (int j = 0; j < steps; ++j) {
        hipLaunchKernelGGL(k_busy, dim3(255), dim3(512), 0, sa, (long long)(100 * tp), sink);
        hipLaunchKernelGGL(k_busy, dim3(1), dim3(512), 0, sb, (long long)(100 * tc), sink);
        OK(hipEventCreate(&ep)); OK(hipEventCreate(&ec));
        OK(hipEventRecord(ep, sa)); OK(hipEventRecord(ec, sb));
        OK(hipStreamWaitEvent(sb, ep, 0)); OK(hipStreamWaitEvent(sa, ec, 0));
      }
    }
    OK(hipStreamEndCapture(sa, &g));
    OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    OK(hipGraphLaunch(ge, sa)); OK(hipStreamSynchronize(sa));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
      OK(hipEventRecord(e0, sa));
      OK(hipGraphLaunch(ge, sa));
      OK(hipEventRecord(e1, sa)); OK(hipStreamSynchronize(sa));
      float ms = 0; OK(hipEventElapsedTime(&ms, e0, e1));
      printf("variant %d (%s, pass busy %.0f us, control busy %.0f us): %.2f us per step\n", variant,
             variant == 0 ? "one chain of 256-workgroup launches" : "two chains with cross edges", tp, tc, 1e3 * ms / steps);
    }
  }
  return 0;
}
