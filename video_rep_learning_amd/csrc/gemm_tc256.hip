// 256x256-tile bf16 MFMA GEMM for the frozen ViT backbone (the dominant kernel of the training step):
//     C[M,N] = epi(A[M,K] . W[N,K]^T + bias[N]),  K % 128 == 0
// Same contract and epilogues as gemm_tc.hip (which stays the fp32 parity kernel and the odd-K fallback); replaces
// the ATen/cuBLAS GEMMs behind timm's nn.Linear / Conv2d(patch) calls reached from CARL_MVF/models/transformer.py:188.
//
// gfx950 design (DESIGN.md "gemm_tc256"):
//  * 512 threads = 8 waves as 2(M) x 4(N); a wave owns 128 x 64 outputs = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16
//    (128 accumulator registers); two waves share each SIMD.
//  * K tile = 64 bf16 (128-B rows).  A K tile is FOUR 16-KiB half-tiles in LDS: A0/A1 hold the m-quadrant 0/1 rows
//    of both wave rows, B0/B1 the n-quadrant 0/1 rows of all four wave columns; two K-tile buffers = 128 KiB.
//  * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, lane-linear image, XOR swizzle applied on the
//    per-lane SOURCE address and on the ds_read_b128 address) -- never through VGPRs.
//  * 4 phases per K tile, one output quadrant (64 x 32 per wave, 16 MFMAs) and one half-tile prefetch each:
//        P0: read B0(4) + A0(8)   MFMA (A0,B0)   issue (t+1).A1
//        P1: read B1(4)           MFMA (A0,B1)   issue (t+2).B0
//        P2: read A1(8)           MFMA (A1,B1)   issue (t+2).A0
//        P3: --                   MFMA (A1,B0)   issue (t+2).B1 ; s_waitcnt vmcnt(6)
//    Each phase is  {ds_reads, LDS-DMA issue} s_barrier {MFMAs} s_barrier.  The wave row wr = 1 runs ONE barrier
//    behind wr = 0, so on every SIMD one wave is in its MFMA segment while its partner loads (matrix pipe kept busy).
//  * hazards (slots = intervals between workgroup barriers; wr=0 loads in slot 2k, computes in 2k+1; wr=1 one later):
//      RAW  a tile's last half-tile (A1) is issued 4 phases before the tile's P3 wait; vmcnt(6) leaves exactly the 3
//           younger half-tiles (2 LDS-DMAs per wave each) in flight; every wave waits BEFORE the first barrier of P3
//           and the first read of the new tile comes after that barrier in both wave rows.
//      WAR  B0 is re-staged one phase after its reads: they are retired by lgkmcnt(8) before P0's first barrier
//           (B reads are issued first; order pinned by sched_barrier).  A0, B1, A1 are re-staged two phases after
//           their reads, whose lgkmcnt(0) precedes the reading phase's second barrier in both wave rows.
//  * no vmcnt(0) / __syncthreads inside the loop; all LDS is one dynamic array (a second __shared__ object makes
//    hipcc drain the DMA queue before every ds_read).
//  * operands swapped (W fragment as MFMA-A) so a lane owns 4 consecutive output columns -> 8/16-byte stores.
//  * 1-D grid, bijective XCD-aware remap: the tiles of one A row-panel run on one XCD and share its L2.
#include "common.h"
#include "mvf_hip_internal.h"
#include "gemm_tc_epi.h"

namespace {
using namespace gemm_tc;

constexpr int BM = 256, BN = 256, ROWB = 128;
constexpr int HALF_BYTES = 128 * ROWB;      // 16 KiB: 128 rows x 64 bf16
constexpr int BUF_BYTES = 4 * HALF_BYTES;   // A0 A1 B0 B1
constexpr int LDS_BYTES = 2 * BUF_BYTES;    // 128 KiB -> one workgroup per CU
constexpr int OFF_A0 = 0, OFF_A1 = HALF_BYTES, OFF_B0 = 2 * HALF_BYTES, OFF_B1 = 3 * HALF_BYTES;

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define WG_BARRIER()              \
  do {                            \
    SCHED_FENCE();                \
    __builtin_amdgcn_s_barrier(); \
    SCHED_FENCE();                \
  } while (0)

// DBG (diagnostic build only, selected with mvf_gemm_tc_select(3); never on the product path): lane 0 of waves 0 and 4
// stamps s_memtime at kernel start, after the prologue wait, after K tiles 0/1/nk-1 and after the epilogue into
// a.dbg[block][2][8] (a buffer of its own; no output depends on it).
template <int EPI, bool DBG>
__global__ __launch_bounds__(512, 2) void gemm_tc256_kernel(GemmTcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  unsigned long long stamps[8];
  int nstamp = 0;
#define STAMP()                                                                              \
  if constexpr (DBG) {                                                                       \
    SCHED_FENCE();                                                                           \
    unsigned long long t_;                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
    SCHED_FENCE();                                                                           \
    stamps[nstamp++] = t_;                                                                   \
  }
  STAMP();

  // ---- XCD-aware, bijective block remap (blocks b and b+8 share an XCD) ----
  const int nbn = (a.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int xcd = blockIdx.x & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int m0 = (lid / nbn) * BM;
  const int n0 = (lid % nbn) * BN;
  const int nk = a.K >> 6;  // K tiles of 64 (even: K % 128 == 0)

  // ---- LDS-DMA source pointers: per half-tile every wave issues 2 pieces of 1 KiB = 8 rows x 128 B ----
  // piece p = i*8 + wave covers half-tile rows p*8 .. p*8+7; lane -> (row = p*8 + lane/8, physical chunk = lane%8)
  const int prow = lane >> 3;
  const int lchunk = (lane & 7) ^ prow;  // logical 16-B chunk fetched into physical chunk lane%8 (row & 7 == prow)
  const char* asrc[2][2];                // [half][piece]
  const char* wsrc[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (i * 8 + wave) * 8 + prow;                 // row inside the half-tile, 0..127
      const int am = (r >> 6) * 128 + h * 64 + (r & 63);       // A: wave row r>>6, m-quadrant h
      const int wn = (r >> 5) * 64 + h * 32 + (r & 31);        // W: wave column r>>5, n-quadrant h
      const int gm = min(m0 + am, a.M - 1);
      const int gn = min(n0 + wn, a.N - 1);
      asrc[h][i] = a.A + (size_t)gm * a.lda * 2 + lchunk * 16;
      wsrc[h][i] = a.W + (size_t)gn * a.ldw * 2 + lchunk * 16;
    }
  const int piece0 = wave * 1024, piece1 = (8 + wave) * 1024;
  auto stage = [&](int buf, int off, const char* const (&src)[2], int kt) {
    char* dst = smem + buf * BUF_BYTES + off;
    const size_t koff = (size_t)kt * ROWB;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src[0] + koff), LDS_PTR(dst + piece0), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GLB_PTR(src[1] + koff), LDS_PTR(dst + piece1), 16, 0, 0);
  };

  // ---- fragment read offsets: lane (frow, fgrp) reads row frow of a 16-row tile, 16-B chunk (ks*4 + fgrp) ^ (frow&7)
  const int frow = lane & 15, fgrp = lane >> 4;
  const int choff = (fgrp ^ (frow & 7)) << 4;                   // k-step 0; k-step 1 is choff ^ 64
  const int a_rd = (wr * 64 + frow) * ROWB + choff;
  const int b_rd = (wc * 32 + frow) * ROWB + choff;

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: tile 0 complete, tile 1 except A1 (issued in P0 of tile 0) ----
  stage(0, OFF_B0, wsrc[0], 0);
  stage(0, OFF_A0, asrc[0], 0);
  stage(0, OFF_B1, wsrc[1], 0);
  stage(0, OFF_A1, asrc[1], 0);
  stage(1, OFF_B0, wsrc[0], 1);
  stage(1, OFF_A0, asrc[0], 1);
  stage(1, OFF_B1, wsrc[1], 1);
  WAIT_VMCNT(6);
  WG_BARRIER();
  STAMP();
  if (wr == 1) WG_BARRIER();  // stagger: wave row 1 runs one barrier behind wave row 0

  bf16x8_t af[4][2], bf0[2][2], bf1[2][2];

#define LOAD_A(OFF)                                                                                          \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                             \
    af[i][0] = *reinterpret_cast<const bf16x8_t*>(base + (OFF) + a_rd + i * 16 * ROWB);                      \
    af[i][1] = *reinterpret_cast<const bf16x8_t*>(base + (OFF) + (a_rd ^ 64) + i * 16 * ROWB);               \
  }
#define LOAD_B(BF, OFF)                                                                                      \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                             \
    BF[j][0] = *reinterpret_cast<const bf16x8_t*>(base + (OFF) + b_rd + j * 16 * ROWB);                      \
    BF[j][1] = *reinterpret_cast<const bf16x8_t*>(base + (OFF) + (b_rd ^ 64) + j * 16 * ROWB);               \
  }
#define MFMA_QUAD(MQ, NQ, BF)                                                                                \
  __builtin_amdgcn_s_setprio(1);                                                                             \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                            \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                               \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                               \
    acc[(MQ) * 4 + i][(NQ) * 2 + j] =                                                                        \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[j][ks], af[i][ks], acc[(MQ) * 4 + i][(NQ) * 2 + j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);

#define K_TILE(BUF, T)                                                                                       \
  {                                                                                                          \
    const char* base = smem + (BUF) * BUF_BYTES;                                                             \
    const int t = (T);                                                                                       \
    /* P0 */                                                                                                 \
    LOAD_B(bf0, OFF_B0);                                                                                     \
    SCHED_FENCE();                                                                                           \
    LOAD_A(OFF_A0);                                                                                          \
    if (t + 1 < nk) stage((BUF) ^ 1, OFF_A1, asrc[1], t + 1);                                                \
    SCHED_FENCE();                                                                                           \
    WAIT_LGKM(8);                                                                                            \
    WG_BARRIER();                                                                                            \
    WAIT_LGKM(0);                                                                                            \
    SCHED_FENCE();                                                                                           \
    MFMA_QUAD(0, 0, bf0);                                                                                    \
    WG_BARRIER();                                                                                            \
    /* P1 */                                                                                                 \
    LOAD_B(bf1, OFF_B1);                                                                                     \
    if (t + 2 < nk) stage((BUF), OFF_B0, wsrc[0], t + 2);                                                    \
    WG_BARRIER();                                                                                            \
    WAIT_LGKM(0);                                                                                            \
    SCHED_FENCE();                                                                                           \
    MFMA_QUAD(0, 1, bf1);                                                                                    \
    WG_BARRIER();                                                                                            \
    /* P2 */                                                                                                 \
    LOAD_A(OFF_A1);                                                                                          \
    if (t + 2 < nk) stage((BUF), OFF_A0, asrc[0], t + 2);                                                    \
    WG_BARRIER();                                                                                            \
    WAIT_LGKM(0);                                                                                            \
    SCHED_FENCE();                                                                                           \
    MFMA_QUAD(1, 1, bf1);                                                                                    \
    WG_BARRIER();                                                                                            \
    /* P3 */                                                                                                 \
    if (t + 2 < nk) {                                                                                        \
      stage((BUF), OFF_B1, wsrc[1], t + 2);                                                                  \
      WAIT_VMCNT(6);                                                                                         \
    } else {                                                                                                 \
      WAIT_VMCNT(0);                                                                                         \
    }                                                                                                        \
    WG_BARRIER();                                                                                            \
    MFMA_QUAD(1, 0, bf0);                                                                                    \
    WG_BARRIER();                                                                                            \
  }

  for (int kt = 0; kt < nk; kt += 2) {
    K_TILE(0, kt);
    if constexpr (DBG) { if (kt == 0) { STAMP(); } }
    K_TILE(1, kt + 1);
    if constexpr (DBG) { if (kt == 0 || kt == 2 || kt + 2 >= nk) { STAMP(); } }
  }
  if (wr == 0) WG_BARRIER();  // balance the stagger barrier of wave row 1
  STAMP();

#undef K_TILE
#undef MFMA_QUAD
#undef LOAD_A
#undef LOAD_B

  // ---- epilogue: two adjacent 16x16 tiles at a time (gemm_tc_epi.h epilogue_pair_bf16); bias loaded once ----
  float4 bj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wc * 64 + j * 16 + fgrp * 4;
    bj[j] = (a.bias != nullptr && n < a.N) ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wr * 128 + (i >> 2) * 64 + (i & 3) * 16 + frow;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      const int nb = n0 + wc * 64 + jp * 32;
      // N % 32 == 0: a tile pair is in range or out as a whole
      epilogue_pair_bf16<EPI>(a, m, m < a.M && nb < a.N, nb, fgrp, acc[i][2 * jp], acc[i][2 * jp + 1], bj[2 * jp],
                              bj[2 * jp + 1]);
    }
  }
  if constexpr (DBG) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP();
    if (a.dbg != nullptr && lane == 0 && (wave == 0 || wave == 4))
      for (int q = 0; q < 8; ++q) a.dbg[((size_t)blockIdx.x * 2 + (wave >> 2)) * 8 + q] = q < nstamp ? stamps[q] : 0ull;
  }
#undef STAMP
}

template <int EPI, bool DBG = false>
int launch(const GemmTcArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tc256_kernel<EPI, DBG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm_tc256_kernel<EPI, DBG>), dim3(nbm * nbn), dim3(512), LDS_BYTES, st, a);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

}  // namespace

int mvf_gemm_tc256_launch(int epi, const gemm_tc::GemmTcArgs& a, hipStream_t st) {
  if (a.K % 128 != 0 || a.K < 128 || a.N % 32 != 0) return MVF_ERR_ARG;
  if (a.dbg != nullptr) return epi == EPI_STORE ? launch<EPI_STORE, true>(a, st) : MVF_ERR_UNSUPPORTED;
  switch (epi) {
    case EPI_STORE: return launch<EPI_STORE>(a, st);
    case EPI_GELU: return launch<EPI_GELU>(a, st);
    case EPI_RESID: return launch<EPI_RESID>(a, st);
    case EPI_PATCH: return launch<EPI_PATCH>(a, st);
  }
  return MVF_ERR_ARG;
}
