// Microbenchmark (development probe, not part of the product): cost of one grid-wide producer/consumer step inside a persistent
// kernel on MI355X.  256 workgroups of 512 threads (one per CU); the last one is the "control" workgroup.
//   control: wait until all pass workgroups have arrived for step s-1, read their 16-double partials, publish step s
//   pass:    wait for step s, write 16 doubles, release, arrive
// Every wait is bounded (no hang on a logic error).  Prints microseconds per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

struct Sh { unsigned* pub; unsigned* arrive; double* part; unsigned* err; double* out; unsigned* flags; };

__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned target, unsigned* err) {
  for (int guard = 0; guard < (1 << 22); ++guard) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); return true; }
    __builtin_amdgcn_s_sleep(1);
  }
  *err = 1; return false;
}

template <int MODE>   // 2: per-workgroup arrival flags (own 128-byte line each), polled lane-parallel by the control workgroup; 0: fences (wbl2 / inv) around plain loads and stores; 1: every shared word moved with agent-scope atomics (no fences)
__global__ __launch_bounds__(512) void k_probe(Sh s, int steps) {
  const int G = gridDim.x, P = G - 1;
  const bool ctl = (int)blockIdx.x == P;
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  double acc = 0.0;
  for (int st = 1; st <= steps; ++st) {
    if (ctl) {
      if (st > 1) {
        if (MODE == 2) {
          // thread t < P polls the flag of workgroup t until it shows step st-1
          bool fine = true;
          if ((int)threadIdx.x < P) {
            fine = false;
            for (int guard = 0; guard < (1 << 22); ++guard) { if (__hip_atomic_load(s.flags + (size_t)threadIdx.x * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(st - 1)) { fine = true; break; } }
            if (!fine) *s.err = 1;
          }
          if (!__syncthreads_and(fine ? 1 : 0)) return;
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else {
        if (threadIdx.x == 0) ok = wait_ge(s.arrive + ((st - 1) & 1), (unsigned)P, s.err) ? 1 : 0;
        __syncthreads();
        if (!ok) return;
        }
        // read the partials of step st-1: thread t < P reads workgroup t's 16 doubles
        if ((int)threadIdx.x < P) {
          for (int k = 0; k < 16; ++k) {
            const double* q = s.part + ((size_t)((st - 1) & 1) * 256 + threadIdx.x) * 16 + k;
            acc += MODE >= 1 ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
          }
        }
        __syncthreads();
        if (MODE != 2 && threadIdx.x == 0) __hip_atomic_store(s.arrive + ((st - 1) & 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __hip_atomic_store(s.pub, (unsigned)st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    } else {
      if (threadIdx.x == 0) {
        if (MODE == 2) { ok = 0; for (int guard = 0; guard < (1 << 22); ++guard) if (__hip_atomic_load(s.pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)st) { ok = 1; break; } if (!ok) *s.err = 1; }
        else ok = wait_ge(s.pub, (unsigned)st, s.err) ? 1 : 0;
      }
      __syncthreads();
      if (!ok) return;
      if (threadIdx.x < 16) {
        double* q = s.part + ((size_t)(st & 1) * 256 + blockIdx.x) * 16 + threadIdx.x;
        const double v = (double)(st + threadIdx.x);
        if (MODE >= 1) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *q = v;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (MODE == 2) __hip_atomic_store(s.flags + (size_t)blockIdx.x * 32, (unsigned)st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(s.arrive + (st & 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (ctl && threadIdx.x == 0) *s.out = acc;
}

int main() {
  Sh s; OK(hipMalloc(&s.pub, 256)); OK(hipMalloc(&s.arrive, 256)); OK(hipMalloc(&s.part, 2 * 256 * 16 * 8)); OK(hipMalloc(&s.err, 256)); OK(hipMalloc(&s.out, 256)); OK(hipMalloc(&s.flags, 256 * 128));
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode) for (int rep = 0; rep < 3; ++rep) {
    OK(hipMemset(s.pub, 0, 256)); OK(hipMemset(s.flags, 0, 256 * 128)); OK(hipMemset(s.arrive, 0, 256)); OK(hipMemset(s.err, 0, 256)); OK(hipMemset(s.part, 0, 2 * 256 * 16 * 8));
    const int steps = 2000;
    OK(hipEventRecord(e0, 0));
    if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(256), dim3(512), 0, 0, s, steps); else if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(256), dim3(512), 0, 0, s, steps); else hipLaunchKernelGGL(k_probe<2>, dim3(256), dim3(512), 0, 0, s, steps);
    OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
    float ms = 0; OK(hipEventElapsedTime(&ms, e0, e1));
    unsigned err = 0; OK(hipMemcpy(&err, s.err, 4, hipMemcpyDeviceToHost));
    printf("mode %d (%s): %d steps, %.3f us per step, err %u\n", mode, mode == 2 ? "per-workgroup flags, no sleep, atomics only" : (mode ? "atomics only" : "release/acquire fences + plain accesses"), steps, 1e3 * ms / steps, err);
  }
  return 0;
}
