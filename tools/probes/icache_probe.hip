// Diagnostic: what does COLD straight-line code cost?  Kernels of N dependent-free v_fma instructions executed once per wave
// (no loop), 24 workgroups of 512 threads, timed back to back.  If instruction fetch is the bound, time grows with code bytes and
// not with work.   hipcc --offload-arch=gfx950 -O3 tools/probes/icache_probe.hip -o /tmp/icache_probe && /tmp/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N>
__global__ __launch_bounds__(512) void straight(float* out, float a, float b) {
  float x0 = threadIdx.x, x1 = a, x2 = b, x3 = a + b;
#pragma unroll
  for (int i = 0; i < N; ++i) {       // 4 independent chains: 8 bytes per v_fma_f32 (VOP3)
    x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, b, a); x2 = __builtin_fmaf(x2, a, a); x3 = __builtin_fmaf(x3, b, b);
  }
  out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3;
}

template <int N>
__global__ __launch_bounds__(512) void looped(float* out, float a, float b, int reps) {
  float x0 = threadIdx.x, x1 = a, x2 = b, x3 = a + b;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, b, a); x2 = __builtin_fmaf(x2, a, a); x3 = __builtin_fmaf(x3, b, b);
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3;
}

template <typename F>
float time_it(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();
  hipEventRecord(e0);
  for (int i = 0; i < 200; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / 200;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
#define RUN(N) printf("straight-line %6d fma (%4d KB of code)  24 WGs: %7.1f us   256 WGs: %7.1f us\n", 4 * N, 4 * N * 8 / 1024, \
    time_it([&] { hipLaunchKernelGGL(straight<N>, dim3(24), dim3(512), 0, 0, out, 1.0001f, 0.5f); }),                       \
    time_it([&] { hipLaunchKernelGGL(straight<N>, dim3(256), dim3(512), 0, 0, out, 1.0001f, 0.5f); }))
  RUN(32); RUN(128); RUN(512); RUN(1024); RUN(2048);
  printf("same work as a loop over a 1 KB body (2048 x 4 fma = 16 x 128 x 4): 24 WGs %7.1f us\n",
         time_it([&] { hipLaunchKernelGGL(looped<32>, dim3(24), dim3(512), 0, 0, out, 1.0001f, 0.5f, 64); }));
  return 0;
}
