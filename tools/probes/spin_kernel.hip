// Diagnostic (GPU box, tools/step_timeline.py --dummy spin:...): a kernel that only occupies resources -- `grid` workgroups of 256
// threads, `lds` bytes of dynamic LDS each, spinning for `us` microseconds on the constant 100 MHz s_memtime clock -- to find out
// what it is about the head's kernels that stretches the backbone forward running beside them.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probes/libspin.so tools/probes/spin_kernel.hip
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void spin_kernel(int ticks, float* out) {
  extern __shared__ float smem[];
  if (threadIdx.x == 0) smem[0] = (float)ticks;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)ticks) __builtin_amdgcn_s_sleep(8);
  if (out != nullptr && threadIdx.x == 0 && blockIdx.x == 0) out[0] = smem[0];
}
extern "C" int spin_launch(int grid, int lds, int us, float* out, hipStream_t st) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), lds, st, us * 100, out);
  return (int)hipGetLastError();
}
