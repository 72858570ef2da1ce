// Diagnostic (GPU box): the matrix pipe fed from REGISTERS only (no LDS, no global traffic in the loop), random bf16 operands,
// v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16 -- the same 512 MACs per SIMD and clock, but the 32 x 32 form reads
// its A / B registers half as often per MAC.  Does the clock the chip holds under its power management differ?
// Output per shape: ticks per 8192-MAC-equivalent, TFLOP/s over the chip, clock.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_shape_probe tools/probes/mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int SHAPE>   // 16 or 32
__global__ __launch_bounds__(256) void mfma_kernel(const bf16x8_t* __restrict__ src, int iters, float* __restrict__ out,
                                                   unsigned long long* __restrict__ cycles) {
  bf16x8_t a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = src[(blockIdx.x * 256 + threadIdx.x) * 16 + i];
    b[i] = src[(blockIdx.x * 256 + threadIdx.x) * 16 + 8 + i];
  }
  float s = 0.f;
  unsigned long long t0, t1;
  if constexpr (SHAPE == 16) {
    f32x4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(b[j]), "v"(a[i]));
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  } else {
    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      // the same MACs per iteration as the 16 x 16 form: 16 tiles x 2 k-steps of 32 x 32 x 16 = 64 x 8192 MACs
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(b[j + 4 * ks]), "v"(a[i + 4 * ks]));
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  }
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE>
void run(const bf16x8_t* src, int iters, float* out, unsigned long long* dcyc, int nwg) {
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(mfma_kernel<SHAPE>, dim3(nwg), dim3(256), 0, 0, src, iters, out, dcyc);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mfma_kernel<SHAPE>, dim3(nwg), dim3(256), 0, 0, src, iters, out, dcyc);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwg);
  hipMemcpy(h.data(), dcyc, nwg * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto c : h) sum += (double)c;
  const double ticks = sum / nwg, flop = 2.0 * 64 * 8192 * (double)iters * 4 * nwg;
  printf("  %2d x %2d: %7.1f ticks / 64 x 8192 MACs (ideal 1024)  %7.1f TFLOP/s  kernel %.1f us  => clock %.2f GHz\n", SHAPE, SHAPE,
         ticks / iters, flop / (ms * 1e-3) / 1e12, ms * 1e3, ticks / (ms * 1e3) / 1e3);
}

int main(int argc, char** argv) {
  const int nwg = 256, iters = argc > 1 ? atoi(argv[1]) : 2048, mode = argc > 2 ? atoi(argv[2]) : 0;
  std::vector<unsigned short> h((size_t)nwg * 256 * 16 * 8);
  unsigned x = 999;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = mode == 1 ? 0 : (unsigned short)(((x >> 16) & 0x807f) | 0x3f80); }
  bf16x8_t* src; hipMalloc(&src, h.size() * 2); hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  float* out; hipMalloc(&out, nwg * 256 * 4);
  unsigned long long* dcyc; hipMalloc(&dcyc, nwg * 8);
  printf("matrix pipe from registers, %s operands, %d iterations x 64 MFMA-equivalents per wave, one wave per SIMD\n",
         mode == 1 ? "all-zero" : "random", iters);
  run<16>(src, iters, out, dcyc, nwg);
  run<32>(src, iters, out, dcyc, nwg);
  run<16>(src, iters, out, dcyc, nwg);
  run<32>(src, iters, out, dcyc, nwg);
  return 0;
}
