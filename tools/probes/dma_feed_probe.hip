// Diagnostic (GPU box): how fast does one CU take in L2-resident bytes by LDS-DMA as a function of the bytes it keeps in
// flight?  One 512-thread workgroup per CU (the GEMM's shape); every wave keeps DEPTH `global_load_lds_dwordx4` (1 KiB each)
// outstanding with a counted vmcnt wait and walks a footprint that all workgroups share (default 2 MiB: L2-resident in every
// XCD after the first pass).  Output: bytes / clk / CU for DEPTH = 2 .. 16 (16 .. 128 KiB in flight per CU).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/dma_feed_probe tools/probes/dma_feed_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int DEPTH>
__global__ __launch_bounds__(512) void feed_kernel(const char* __restrict__ src, size_t footprint, int iters,
                                                   unsigned long long* __restrict__ cycles) {
  extern __shared__ char smem[];   // 8 waves x DEPTH KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* ring = smem + wave * DEPTH * 1024;
  // every workgroup starts elsewhere in the footprint; a wave's pieces are 1 KiB = 8 rows x 128 B of a 1536-byte-pitch matrix
  size_t pos = ((size_t)blockIdx.x * 131 + wave * 17) * 12288 % footprint;
  const size_t lane_off = (size_t)(lane >> 3) * 1536 + (lane & 7) * 16;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      size_t a = pos + lane_off;
      if (a >= footprint) a -= footprint;
      __builtin_amdgcn_global_load_lds(GLB_PTR(src + a), LDS_PTR(ring + s * 1024), 16, 0, 0);
      pos += 12288;
      if (pos >= footprint) pos -= footprint;
      // keep DEPTH - 1 younger loads in flight: wait for the oldest before its slot is issued again
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int DEPTH>
double run(const char* src, size_t footprint, int iters, unsigned long long* dcyc, int nwg) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(feed_kernel<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DEPTH * 1024);
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL(feed_kernel<DEPTH>, dim3(nwg), dim3(512), 8 * DEPTH * 1024, 0, src, footprint, iters, dcyc);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(feed_kernel<DEPTH>, dim3(nwg), dim3(512), 8 * DEPTH * 1024, 0, src, footprint, iters, dcyc);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwg);
  hipMemcpy(h.data(), dcyc, nwg * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto c : h) sum += (double)c;
  const double bytes_per_wg = (double)iters * DEPTH * 8 * 1024;
  printf("  depth %2d (%3d KiB in flight / CU): %6.1f B/clk/CU (s_memtime clock), %6.2f TB/s chip, kernel %.1f us\n", DEPTH,
         DEPTH * 8, bytes_per_wg / (sum / nwg), bytes_per_wg * nwg / (ms * 1e-3) / 1e12, ms * 1e3);
  return ms;
}

int main(int argc, char** argv) {
  const size_t footprint_mb = argc > 1 ? atoi(argv[1]) : 2;
  const size_t footprint = footprint_mb << 20;
  const int nwg = argc > 2 ? atoi(argv[2]) : 256;
  char* src;
  hipMalloc(&src, footprint + (1 << 20));
  hipMemset(src, 1, footprint + (1 << 20));
  unsigned long long* dcyc;
  hipMalloc(&dcyc, nwg * 8);
  printf("footprint %zu MiB shared by %d workgroups of 512 threads\n", footprint_mb, nwg);
  const int total_kib = 48 * 1024;   // per workgroup
  run<2>(src, footprint, total_kib / (2 * 8), dcyc, nwg);
  run<4>(src, footprint, total_kib / (4 * 8), dcyc, nwg);
  run<6>(src, footprint, total_kib / (6 * 8), dcyc, nwg);
  run<8>(src, footprint, total_kib / (8 * 8), dcyc, nwg);
  run<12>(src, footprint, total_kib / (12 * 8), dcyc, nwg);
  run<16>(src, footprint, total_kib / (16 * 8), dcyc, nwg);
  return 0;
}
