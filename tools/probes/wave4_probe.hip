// Diagnostic (GPU box): what does the K loop of a 256 x 256 tile cost when FOUR waves (one per SIMD, 512 registers each, the
// 128 x 128 accumulator block of a wave in AGPRs) run it instead of the product kernel's eight (two per SIMD, 256 registers)?
// DESIGN.md 4b(a) rejected that structure by arithmetic (an in-order wave has to issue its own LDS-DMA pieces and fragment
// reads between its MFMAs, with no partner wave on the SIMD to cover for it); this probe measures it.  Loop only: no epilogue,
// operands L2-resident, results written once so that nothing is optimised away (and NOT checked: the schedule is what is timed).
//   stage = 32 k: A 256 rows x 64 B + B 256 rows x 64 B = 32 KiB, ring of four stages (128 KiB), LDS-DMA two stages ahead,
//   per stage and wave: 64 MFMA 16x16x32 (1 024 matrix cycles), 16 ds_read_b128 (fragments of the NEXT stage, double-buffered
//   in registers), 8 global_load_lds_dwordx4, one counted vmcnt wait + one barrier in the middle of the stage.
// Variants: full | no DMAs | no fragment reads | MFMAs + barriers only.  Output: s_memtime ticks per stage (ideal 1 024) and
// TFLOP/s over the chip by events.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/wave4_probe tools/probes/wave4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int STAGE_BYTES = 32 * 1024, NSTAGE = 4;

// the accumulators are pinned to AGPRs ("+a"): left to itself hipcc keeps part of a 256-register accumulator block in VGPRs
// and shuffles it through v_accvgpr_read / _write (600 moves per four stages).  Every accumulator is touched once per stage
// (64 MFMAs apart), so no MFMA -> MFMA wait state is hidden from the hazard pass; the reads after the loop get explicit s_nops
__device__ __forceinline__ void mfma_a(f32x4_t& c, const bf16x8_t& a, const bf16x8_t& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <bool DMA, bool DSR, bool RND = true>
__global__ __launch_bounds__(256) void wave4_kernel(const char* __restrict__ src, size_t footprint, int nstage_iters,
                                                    float* __restrict__ out, unsigned long long* __restrict__ cycles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int frow = lane & 15, fgrp = lane >> 4;
  // fragment i of this wave's 128 rows: 16 rows x 64 B = 1 KiB contiguous (lane reads row frow, 16-byte chunk fgrp)
  const int a_rd = (wr * 128 + frow) * 64 + fgrp * 16;
  const int b_rd = 16384 + (wc * 128 + frow) * 64 + fgrp * 16;
  // this wave's 8 of a stage's 32 pieces (1 KiB = 16 rows x 64 B each): rows (wave * 8 + p) * 16 + (lane >> 2), chunk lane & 3
  // source of a piece = uniform base (SGPRs) + one 32-bit lane offset; rows of a 1 536-byte-pitch matrix (K = 768 bf16)
  unsigned pos = (unsigned)(((size_t)blockIdx.x * 131) * 32768 % footprint);   // uniform; the allocation has 1 MiB of slack
  const unsigned lane_off = (unsigned)((wave * 128 + (lane >> 2)) * 1536 + (lane & 3) * 16);

  f32x4_t acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[2][8], bfr[2][8];

  auto issue_dma = [&](int stage_slot, int p, bool always = false) {
    if (DMA || always) {
      const char* base = src + pos + (size_t)p * 16 * 1536;     // uniform
      __builtin_amdgcn_global_load_lds(GLB_PTR(base + lane_off), LDS_PTR(smem + stage_slot * STAGE_BYTES + (wave * 8 + p) * 1024), 16, 0, 0);
    }
  };
  auto read_frags = [&](int stage_slot, int fb, int i, bool always = false) {   // fragment pair i (one A, one B) of a stage into register set fb
    if (DSR || always) {
      const char* s = smem + stage_slot * STAGE_BYTES;
      af[fb][i] = *reinterpret_cast<const bf16x8_t*>(s + a_rd + i * 1024);
      bfr[fb][i] = *reinterpret_cast<const bf16x8_t*>(s + b_rd + i * 1024);
    }
  };
  // prologue: stages 0 and 1 in flight, stage 0 landed, its fragments read
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int p = 0; p < 8; ++p) issue_dma(s, p, true);
    pos += 64;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // (without fragment reads in the loop: both register sets hold stage 0 / 1 operands for good -- random like the rest --
  // or, RND = false, one constant pattern in every lane)
#pragma unroll
  for (int i = 0; i < 8; ++i) read_frags(0, 0, i, true);
  if constexpr (!DSR) {
#pragma unroll
    for (int i = 0; i < 8; ++i) read_frags(1, 1, i, true);
    if constexpr (!RND) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        af[0][i] = af[1][i] = (bf16x8_t){1, 2, 3, 4, 5, 6, 7, 8};
        bfr[0][i] = bfr[1][i] = (bf16x8_t){1, 2, 3, 4, 5, 6, 7, 8};
      }
    }
  }
  if constexpr (!DMA) __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int st = 0; st < nstage_iters; st += 4) {
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) {       // stage st + uu: fragments in register set uu & 1, slot uu
      const int u = uu & 1, slot = uu, nslot = (uu + 1) & 3, dslot = (uu + 2) & 3;
      // first half: rows i = 0..3 (32 MFMAs) with the 8 DMAs of stage + 2
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        issue_dma(dslot, 2 * i);
        issue_dma(dslot, 2 * i + 1);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          mfma_a(acc[i][j], bfr[u][j], af[u][i]);
      }
      pos += 64;
      if (pos >= (unsigned)footprint) pos -= (unsigned)footprint;
      // stage + 1 landed (this wave's pieces: all but the 8 just issued), then everybody's
      if constexpr (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // second half: rows 4..7 with the 16 fragment reads of stage + 1
#pragma unroll
      for (int i = 4; i < 8; ++i) {
        read_frags(nslot, u ^ 1, 2 * (i - 4));
        read_frags(nslot, u ^ 1, 2 * (i - 4) + 1);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          mfma_a(acc[i][j], bfr[u][j], af[u][i]);
      }
      (void)slot;
    }
  }
  if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool DMA, bool DSR, bool RND = true>
void run(const char* name, const char* src, size_t footprint, int stages, float* out, unsigned long long* dcyc, int nwg) {
  auto k = wave4_kernel<DMA, DSR, RND>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_BYTES);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), NSTAGE * STAGE_BYTES, 0, src, footprint, stages, out, dcyc);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k, dim3(nwg), dim3(256), NSTAGE * STAGE_BYTES, 0, src, footprint, stages, out, dcyc);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwg);
  hipMemcpy(h.data(), dcyc, nwg * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto c : h) sum += (double)c;
  const double flop = 2.0 * 256 * 256 * 32 * (double)stages * nwg;
  const double tps = sum / nwg / stages;
  printf("  %-28s %7.1f ticks / stage (ideal 1024 matrix cycles)  %7.1f TFLOP/s  kernel %.1f us  => clock %.2f GHz\n", name, tps,
         flop / (ms * 1e-3) / 1e12, ms * 1e3, tps * stages / (ms * 1e3) / 1e3);
}

int main(int argc, char** argv) {
  const size_t footprint = (size_t)(argc > 1 ? atoi(argv[1]) : 2) << 20;
  const int nwg = argc > 2 ? atoi(argv[2]) : 256;
  const int stages = argc > 3 ? atoi(argv[3]) : 2048;
  const int mode = argc > 4 ? atoi(argv[4]) : 0;   // 0 random sign / mantissa in +-[1, 2), 1 all zero, 2 N(0, 1) rounded to bf16
  const bool zeros = mode == 1;
  char* src;
  hipMalloc(&src, footprint + (4 << 20));
  std::vector<unsigned short> h((footprint + (4 << 20)) / 2);
  unsigned x = 12345;
  auto rnd = [&] { x = x * 1664525u + 1013904223u; return x; };
  for (auto& v : h) {            // (the clock the chip holds depends on the operand bits)
    if (mode == 2) {
      const float u1 = ((rnd() >> 8) + 1) * (1.0f / 16777217.0f), u2 = (rnd() >> 8) * (1.0f / 16777216.0f);
      const float g = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
      unsigned b;
      memcpy(&b, &g, 4);
      v = (unsigned short)((b + 0x7fff + ((b >> 16) & 1)) >> 16);
    } else {
      const unsigned r = rnd();
      v = zeros ? 0 : (unsigned short)(((r >> 16) & 0x807f) | 0x3f80);
    }
  }
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  float* out;
  hipMalloc(&out, (size_t)nwg * 256 * 4);
  unsigned long long* dcyc;
  hipMalloc(&dcyc, nwg * 8);
  printf("4 waves x (128 x 128) per 256 x 256 tile, %d workgroups, %d stages of 32 k, operand footprint %zu MiB%s\n", nwg, stages,
         footprint >> 20, mode == 1 ? ", all-zero operands" : mode == 2 ? ", N(0,1) operands" : "");
  const int sustain = argc > 5 ? atoi(argv[5]) : 0;   // > 0: only the full loop, `sustain` batches of 200 back-to-back launches
  if (sustain > 0) {
    auto k = wave4_kernel<true, true, true>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int b = 0; b < sustain; ++b) {
      hipEventRecord(e0, 0);
      for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), NSTAGE * STAGE_BYTES, 0, src, footprint, stages, out, dcyc);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      printf("   %.1f TFLOP/s\n", 2.0 * 256 * 256 * 32 * (double)stages * nwg * 200 / (ms * 1e-3) / 1e12);
      fflush(stdout);
    }
    return 0;
  }
  run<true, true>("full loop", src, footprint, stages, out, dcyc, nwg);
  run<false, true>("no LDS-DMA", src, footprint, stages, out, dcyc, nwg);
  run<true, false>("no fragment reads", src, footprint, stages, out, dcyc, nwg);
  run<false, false>("MFMAs + barriers only", src, footprint, stages, out, dcyc, nwg);
  run<false, false, false>("  same, constant operands", src, footprint, stages, out, dcyc, nwg);
  return 0;
}
