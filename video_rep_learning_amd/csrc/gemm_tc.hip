// Tiled MFMA GEMM for the frozen ViT backbone:  C[M,N] = epi(A[M,K] . W[N,K]^T + bias[N])
//
// Replaces the ATen/cuBLAS GEMMs behind timm's nn.Linear / Conv2d(patch) calls that the
// reference reaches through CARL_MVF/models/transformer.py:188 (qkv, proj, fc1, fc2, patch_embed).
//
// gfx950 design (see DESIGN.md "gemm_tc"):
//  * 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 4x4 MFMA tiles)
//  * K tile = 128 BYTES per row for both dtypes (64 bf16 / 32 f32) so one LDS geometry serves both
//  * operands stream HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip); the LDS
//    image is lane-linear, the XOR bank swizzle is applied on the per-lane SOURCE address and again on
//    the ds_read_b128 address (same involution both sides)
//  * 2-stage ring: loads of tile k+1 fly under the MFMAs of tile k, one barrier per K tile
//  * bf16: v_mfma_f32_16x16x32_bf16; f32 (parity mode): v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain)
//  * operands are swapped (W fragment as MFMA-A) so each lane ends up with 4 CONSECUTIVE output columns
//    of one row -> 8/16-byte epilogue stores and float4 bias / residual / pos-embed loads
//  * 1-D grid with a bijective XCD-aware remap: workgroups that share an A row-panel land on one XCD's L2
#include "common.h"
#include "mvf_hip_internal.h"
#include "gemm_tc_epi.h"

#include <cstdlib>

namespace {
using namespace gemm_tc;
int g_variant = 0;
// (a "tail split" -- the last, partly empty round of 256x256 tiles on this 128x128 kernel -- was measured in round 2:
// -15 % on a lone N = 768 launch, +5 % on the training step, where the other backbone lane already fills that tail; removed)
unsigned long long* g_dbg = nullptr;
unsigned g_dbg_rowmask = 0x7fffffffu;
int g_dbg_kt = -1;
int g_dbg_abl = 0;

constexpr int BM = 128, BN = 128, ROWB = 128;
constexpr int TILE_BYTES = BM * ROWB;          // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + W
constexpr int LDS_BYTES = 2 * STAGE_BYTES;     // 64 KiB -> 2 workgroups / CU

template <typename T, int EPI, bool LN>
__global__ __launch_bounds__(256, 2) void gemm_tc_kernel(GemmTcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KE = ROWB / (int)sizeof(T);  // K elements per tile
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // ---- XCD-aware, bijective block remap (blocks b and b+8 share an XCD) ----
  const int nbn = (a.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int xcd = blockIdx.x & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int m0 = (lid / nbn) * BM;
  const int n0 = (lid % nbn) * BN;
  const int nk = a.K / KE;

  // ---- per-thread staging coordinates (4 LDS-DMA pieces per operand per tile) ----
  // piece i of this wave covers tile rows i*32 + wave*8 .. +8; lane -> (row, physical 16-B chunk)
  const int srow = tid >> 3;        // + i*32
  const int pchunk = tid & 7;
  const int lchunk = pchunk ^ (srow & 7);  // (i*32 keeps row&7)
  const char* asrc[4];
  const char* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = i * 32 + srow;
    int gm = min(m0 + r, a.M - 1);
    int gn = min(n0 + r, a.N - 1);
    asrc[i] = a.A + (size_t)gm * a.lda * sizeof(T) + lchunk * 16;
    wsrc[i] = a.W + (size_t)gn * a.ldw * sizeof(T) + lchunk * 16;
  }
  auto stage = [&](int s, int kt) {
    char* sa = smem + s * STAGE_BYTES;
    char* sb = sa + TILE_BYTES;
    const size_t koff = (size_t)kt * ROWB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = (i * 256 + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds(GLB_PTR(asrc[i] + koff), LDS_PTR(sa + piece), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLB_PTR(wsrc[i] + koff), LDS_PTR(sb + piece), 16, 0, 0);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;  // fragment row inside a 16-row MFMA tile
  const int fgrp = lane >> 4;  // k-group 0..3
  const int fsw = frow & 7;

  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // tile kt landed (own vmcnt drained by the compiler) and everyone left tile kt-1
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* sa = smem + (kt & 1) * STAGE_BYTES + (wr * 64 + frow) * ROWB;
    const char* sb = smem + (kt & 1) * STAGE_BYTES + TILE_BYTES + (wc * 64 + frow) * ROWB;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int coff = ((h * 4 + fgrp) ^ fsw) << 4;
      if constexpr (sizeof(T) == 2) {
        bf16x8_t af[4], wf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(sa + i * 16 * ROWB + coff);
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(sb + j * 16 * ROWB + coff);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
      } else {
        f32x4_t af[4], wf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const f32x4_t*>(sa + i * 16 * ROWB + coff);
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const f32x4_t*>(sb + j * 16 * ROWB + coff);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][s], af[i][s], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane owns C[m][n..n+3], m = tile row (lane&15), n = 4*(lane>>4) + r ----
  if constexpr (!LN) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wr * 64 + i * 16 + frow;
      if (m >= a.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + fgrp * 4;
        if (n >= a.N) continue;
        epilogue4<T, EPI>(a, m, n, acc[i][j]);
      }
    }
  } else {
    // LN-fold extras (same contract as gemm_tc256's): N % 64 == 0 here, so a wave's 64 columns are in range or out as a whole;
    // rows beyond M only skip their memory accesses -- every lane takes part in the cross-lane row sums
    const bool cols = n0 + wc * 64 < a.N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wr * 64 + i * 16 + frow;
      const bool ok = m < a.M && cols;
      float2 mr = make_float2(0.f, 1.f);
      if constexpr (EPI == EPI_STORE || EPI == EPI_GELU) {
        if (ok) mr = *reinterpret_cast<const float2*>(a.ln_mr + (size_t)m * 2);
      }
      float s1 = 0.f, s2 = 0.f;
      if (ok) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          float v0[4] = {0.f, 0.f, 0.f, 0.f}, v1[4] = {0.f, 0.f, 0.f, 0.f};
          epilogue4<T, EPI, true>(a, m, n0 + wc * 64 + (2 * jp) * 16 + fgrp * 4, acc[i][2 * jp], mr, v0);
          epilogue4<T, EPI, true>(a, m, n0 + wc * 64 + (2 * jp + 1) * 16 + fgrp * 4, acc[i][2 * jp + 1], mr, v1);
#pragma unroll
          for (int r = 0; r < 4; ++r) {     // the summation order of epilogue_pair_bf16_ln: the two kernels agree bit for bit
            s1 += v0[r] + v1[r];
            s2 = fmaf(v0[r], v0[r], fmaf(v1[r], v1[r], s2));
          }
        }
      }
      if constexpr (EPI == EPI_RESID) {
        if (a.stats != nullptr) {
          s1 = row_quad_sum(s1);
          s2 = row_quad_sum(s2);
          if (fgrp == 0 && ok)
            *reinterpret_cast<float2*>(a.stats + ((size_t)((n0 >> 6) + wc) * a.M + m) * 2) = make_float2(s1, s2);
        }
      }
    }
  }
}

template <typename T, int EPI, bool LN = false>
int launch(const GemmTcArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tc_kernel<T, EPI, LN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm_tc_kernel<T, EPI, LN>), dim3(nbm * nbn), dim3(256), LDS_BYTES, st, a);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

template <typename T>
int dispatch(int epi, const GemmTcArgs& a, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    if (a.ln_mr != nullptr || a.xb != nullptr || a.stats != nullptr) {   // LN-fold extras: bf16 only
      switch (epi) {
        case EPI_STORE: return launch<T, EPI_STORE, true>(a, st);
        case EPI_GELU: return launch<T, EPI_GELU, true>(a, st);
        case EPI_RESID: return launch<T, EPI_RESID, true>(a, st);
      }
      return MVF_ERR_ARG;
    }
  }
  switch (epi) {
    case EPI_STORE: return launch<T, EPI_STORE>(a, st);
    case EPI_GELU: return launch<T, EPI_GELU>(a, st);
    case EPI_RESID: return launch<T, EPI_RESID>(a, st);
    case EPI_PATCH: return launch<T, EPI_PATCH>(a, st);
  }
  return MVF_ERR_ARG;
}

}  // namespace

// Internal entry used by the ViT driver and exported through the C ABI (mvf_gemm_tc in mvf_hip.h).
int mvf_gemm_tc_impl(int dtype, int epi, const void* A, int lda, const void* W, int ldw, const float* bias, void* C,
                     int ldc, float* resid, int ldr, void* tap, int ldt, const float* pos, const float* ls, int tpf, int M,
                     int N, int K, hipStream_t st, int batch_rows, int w_batch_rows, const MvfGemmLn* ln) {
  const bool f16 = dtype == MVF_F16;     // fp16 operands: the 256x256 kernel's F16 instantiations, nothing else
  if (f16) dtype = MVF_BF16;             // (same element size, alignment rules and epilogues as bf16 from here on)
  const int esz = dtype == MVF_BF16 ? 2 : 4;
  const int ke = ROWB / esz;
  MVF_CHECK_ARG(dtype == MVF_BF16 || dtype == MVF_F32);
  MVF_CHECK_ARG(A && W && M > 0 && N > 0 && K > 0);
  MVF_CHECK_ARG(K % ke == 0 && N % 4 == 0);
  MVF_CHECK_ARG(((size_t)lda * esz) % 16 == 0 && ((size_t)ldw * esz) % 16 == 0);
  MVF_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0);
  MVF_CHECK_ARG(bias == nullptr || ((uintptr_t)bias % 16) == 0);
  if (epi == EPI_STORE || epi == EPI_GELU) MVF_CHECK_ARG(C && ldc % 4 == 0 && ((uintptr_t)C % 16) == 0);
  if (epi == EPI_RESID) MVF_CHECK_ARG(resid && ldr % 4 == 0 && (tap == nullptr || (ldt % 4 == 0 && tpf > 1)));
  if (epi == EPI_PATCH) MVF_CHECK_ARG(resid && pos && tpf > 1 && ldr % 4 == 0 && M % (tpf - 1) == 0);
  GemmTcArgs a;
  a.A = (const char*)A; a.W = (const char*)W; a.bias = bias; a.C = (char*)C; a.resid = resid; a.tap = (char*)tap;
  a.pos = pos; a.ls = ls; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.ldt = ldt; a.M = M; a.N = N; a.K = K; a.tpf = tpf;
  a.dbg = g_dbg; a.dbg_rowmask = g_dbg_rowmask; a.dbg_kt = g_dbg_kt; a.dbg_abl = g_dbg_abl;
  a.sched = nullptr;   // set by the persistent gemm_tc256 launch
  a.batch_rows = batch_rows; a.w_batch_rows = w_batch_rows;
  a.xb = nullptr; a.ldxb = 0; a.stats = nullptr; a.ln_mr = nullptr; a.ln_c = nullptr;
  a.ln_part = nullptr; a.ln_ns = 0; a.ln_inv_d = 0.f; a.ln_eps = 0.f; a.ngroup = 0; a.f16 = 0;
  a.sa = nullptr; a.sw = nullptr; a.csc = nullptr;
  a.radd = resid;
  a.radd2 = nullptr; a.ldr2 = 0;
  a.f16 = f16 ? 1 : 0;
  if (ln != nullptr && ln->addend2 != nullptr) {
    MVF_CHECK_ARG(epi == EPI_RESID && dtype == MVF_BF16 && ((uintptr_t)ln->addend2 % 8) == 0 && ln->ld2 % 4 == 0 && ln->ld2 >= N);
    a.radd2 = (const bf16_t*)ln->addend2; a.ldr2 = ln->ld2;
  }
  if (ln != nullptr && ln->addend_mode != 0) {
    MVF_CHECK_ARG(epi == EPI_RESID && (ln->addend_mode == 2 || (ln->addend_mode == 1 && ln->addend && ((uintptr_t)ln->addend % 16) == 0)));
    a.radd = ln->addend_mode == 1 ? ln->addend : nullptr;
  }
  const bool fold = ln != nullptr && (ln->xb || ln->stats || ln->ln_mr || ln->ln_c || ln->ln_part);
  if (fold) {   // LN fold: epilogue extras of the 256x256 bf16 kernel
    if (ln->xb || ln->stats) MVF_CHECK_ARG(epi == EPI_RESID);
    if (ln->xb) MVF_CHECK_ARG(ln->ldxb % 8 == 0 && ln->ldxb >= N && ((uintptr_t)ln->xb % 16) == 0);
    if (ln->stats) MVF_CHECK_ARG(N % 64 == 0 && ((uintptr_t)ln->stats % 8) == 0);
    if (ln->ln_mr || ln->ln_c || ln->ln_part)
      MVF_CHECK_ARG((epi == EPI_STORE || epi == EPI_GELU) && ln->ln_c && (ln->ln_mr != nullptr) != (ln->ln_part != nullptr) &&
                    ((uintptr_t)ln->ln_mr % 8) == 0 && ((uintptr_t)ln->ln_part % 16) == 0 && (!ln->ln_part || ln->ln_ns > 0));
    if (!(dtype == MVF_BF16 && N % 64 == 0 && batch_rows == 0)) return MVF_ERR_UNSUPPORTED;
    // partial sums instead of (mean, rstd): the persistent 256x256 kernel only (it refuses shapes it cannot stage)
    if (ln->ln_part && !(g_variant != 1 && g_variant != 3 && K % 128 == 0 && N % 32 == 0)) return MVF_ERR_UNSUPPORTED;
    a.xb = (char*)ln->xb; a.ldxb = ln->ldxb; a.stats = ln->stats; a.ln_mr = ln->ln_mr; a.ln_c = ln->ln_c;
    a.ln_part = ln->ln_part; a.ln_ns = ln->ln_ns; a.ln_inv_d = 1.0f / (float)K; a.ln_eps = ln->ln_eps;
  }
  if (batch_rows != 0) {   // stacked batches: the 256x256 kernel only
    MVF_CHECK_ARG(batch_rows > 0 && batch_rows % 256 == 0 && M % batch_rows == 0 && w_batch_rows >= N);
    if (!(dtype == MVF_BF16 && g_variant != 1 && K % 128 == 0 && N % 32 == 0)) return MVF_ERR_UNSUPPORTED;
  }
  // bf16 with K a multiple of 128: the 256x256 8-phase kernel (gemm_tc256.hip); g_variant 1 pins the 128x128 kernel
  if (f16 && !(g_variant != 1 && K % 128 == 0 && N % 32 == 0 && batch_rows == 0)) return MVF_ERR_UNSUPPORTED;
  if (dtype == MVF_BF16 && g_variant != 1 && K % 128 == 0 && N % 32 == 0) {
    const int rc = mvf_gemm_tc256_launch(epi, a, /*persistent=*/g_variant != 3, st);
    if (f16) return rc;
    // operands of 4 GiB or more are beyond the 256x256 kernel's 32-bit offsets: the 128x128 kernel (64-bit addressing)
    // takes over unless the caller pinned the kernel or asked for stacked batches
    if (rc != MVF_ERR_UNSUPPORTED || g_variant >= 2 || batch_rows != 0) return rc;
  }
  if (g_variant >= 2 || a.ln_part != nullptr) return MVF_ERR_UNSUPPORTED;   // (partial sums: the 256x256 kernel or nothing)
  return dtype == MVF_BF16 ? dispatch<bf16_t>(epi, a, st) : dispatch<float>(epi, a, st);
}

// MX-fp8 operands: the 256x256 kernel's FP8 variants only (no 128x128 fallback: sizes beyond its 32-bit offsets are refused)
int mvf_gemm_fp8_impl(int epi, const void* A, int lda, const unsigned* sa, const void* W, int ldw, const unsigned* sw,
                      const float* bias, void* C, int ldc, unsigned* c_scales, float* resid, int ldr, void* tap, int ldt,
                      const float* ls, int tpf, int M, int N, int K, hipStream_t st, const void* addend2, int ld2,
                      const MvfGemmLn* ln) {
  MVF_CHECK_ARG(A && W && sa && sw && M > 0 && N > 0 && K > 0 && K % 256 == 0 && N % 32 == 0);
  if (c_scales != nullptr) {   // epi 1 with an MX-fp8 result: C = e4m3 bytes, c_scales [N/128][M]
    MVF_CHECK_ARG(epi == EPI_GELU && C && N % 128 == 0 && ldc % 8 == 0 && ((uintptr_t)C % 8) == 0 && ((uintptr_t)c_scales % 4) == 0);
    epi = EPI_GELU_Q;
  }
  MVF_CHECK_ARG(lda % 16 == 0 && ldw % 16 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0);
  MVF_CHECK_ARG(((uintptr_t)sa % 4) == 0 && ((uintptr_t)sw % 4) == 0 && (bias == nullptr || ((uintptr_t)bias % 16) == 0));
  if (epi == EPI_STORE || epi == EPI_GELU) MVF_CHECK_ARG(C && ldc % 4 == 0 && ((uintptr_t)C % 16) == 0);
  else if (epi == EPI_RESID) MVF_CHECK_ARG(resid && ldr % 4 == 0 && (tap == nullptr || (ldt % 4 == 0 && tpf > 1)));
  else if (epi != EPI_GELU_Q) return MVF_ERR_ARG;
  GemmTcArgs a;
  a.A = (const char*)A; a.W = (const char*)W; a.bias = bias; a.C = (char*)C; a.resid = resid; a.tap = (char*)tap;
  a.pos = nullptr; a.ls = ls; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.ldt = ldt; a.M = M; a.N = N; a.K = K; a.tpf = tpf;
  a.dbg = nullptr; a.dbg_rowmask = 0x7fffffffu; a.dbg_kt = -1; a.dbg_abl = 0; a.sched = nullptr; a.batch_rows = 0; a.w_batch_rows = 0;
  a.xb = nullptr; a.ldxb = 0; a.stats = nullptr; a.ln_mr = nullptr; a.ln_c = nullptr;
  a.ln_part = nullptr; a.ln_ns = 0; a.ln_inv_d = 0.f; a.ln_eps = 0.f; a.ngroup = 0; a.f16 = 0;
  a.sa = sa; a.sw = sw; a.csc = c_scales; a.radd = resid; a.radd2 = nullptr; a.ldr2 = 0;
  if (addend2 != nullptr) {
    MVF_CHECK_ARG(epi == EPI_RESID && ((uintptr_t)addend2 % 8) == 0 && ld2 % 4 == 0 && ld2 >= N);
    a.radd2 = (const bf16_t*)addend2; a.ldr2 = ld2;
  }
  if (ln != nullptr) {   // LayerNorm folded into the fp8 GEMMs (include/mvf_hip.h: qkv_c in fp8 mode)
    if (ln->ln_mr != nullptr) {          // consumer: A = MX-fp8 of the UN-normalised residual stream, W = MX-fp8(gamma (.) W)
      MVF_CHECK_ARG(epi == EPI_STORE && ln->ln_c != nullptr && ((uintptr_t)ln->ln_mr % 8) == 0 && ((uintptr_t)ln->ln_c % 16) == 0);
      a.ln_mr = ln->ln_mr; a.ln_c = ln->ln_c;
    } else {                             // producer: the new residual row also as MX-fp8 + its row sums
      MVF_CHECK_ARG(epi == EPI_RESID && ln->xb != nullptr && ln->stats != nullptr && ln->xb_scales != nullptr && N % 128 == 0 &&
                    ln->ldxb % 8 == 0 && ln->ldxb >= N && ((uintptr_t)ln->xb % 8) == 0 && ((uintptr_t)ln->xb_scales % 4) == 0 &&
                    ((uintptr_t)ln->stats % 8) == 0);
      a.xb = (char*)ln->xb; a.ldxb = ln->ldxb; a.stats = ln->stats; a.csc = ln->xb_scales;
    }
  }
  return mvf_gemm_tc256_launch(epi, a, /*persistent=*/true, st);
}

// 0 = automatic choice, 1 = always the 128x128 kernel, 2 = only the 256x256 kernel (error where it does not apply),
// 3 = the 256x256 kernel launched one workgroup per tile instead of persistent (A/B measurements)
// diagnostic: stamps buffer [blocks][2][8] u64 for the gemm_tc256 DBG build (null = product kernels)
extern "C" int mvf_gemm_tc_debug_stamps(unsigned long long* buf) {
  g_dbg = buf;
  return MVF_OK;
}
// diagnostic (stamped build only): A rows are read as row & mask -- A's footprint shrinks to mask + 1 rows (L2-resident)
// diagnostic (stamped build only): kt >= 0 also stamps every workgroup barrier of K tile kt in each workgroup's second tile;
// the stamps buffer then needs [blocks][2][8] + [blocks][2][16] entries (blocks = workgroups launched)
extern "C" int mvf_gemm_tc_debug_ktile(int kt) {
  g_dbg_kt = kt;
  return MVF_OK;
}
// diagnostic (stamped build only), timing ablations whose results are garbage: bit 0 no MFMAs, bit 1 no LDS fragment reads,
// bit 2 no operand DMAs -- what is left of the K loop's time when one of its three streams is removed
extern "C" int mvf_gemm_tc_debug_ablate(int bits) {
  g_dbg_abl = bits;
  return MVF_OK;
}
extern "C" int mvf_gemm_tc_debug_rowmask(int mask) {
  g_dbg_rowmask = (unsigned)mask;
  return MVF_OK;
}

// C[b*R + m, n] (+)= A[b*R + m, :] . W[b*Wr + n, :]^T for b = 0 .. M/R - 1: the split-K form of a weight gradient (each batch
// one chunk of the reduction, fp32 partial sums through the read-modify epilogue)
extern "C" int mvf_gemm_tc_batched(int epi, const void* A, int lda, const void* W, int ldw, void* C, int ldc, float* resid,
                                   int ldr, int M, int N, int K, int batch_rows, int w_batch_rows, hipStream_t st) {
  MVF_CHECK_ARG(epi == EPI_STORE || epi == EPI_RESID);
  return mvf_gemm_tc_impl(MVF_BF16, epi, A, lda, W, ldw, nullptr, C, ldc, resid, ldr, nullptr, 0, nullptr, nullptr, 1, M, N,
                          K, st, batch_rows, w_batch_rows);
}

// the same with plain fp32 results (no addend: `out` need not be initialised)
extern "C" int mvf_gemm_tc_batched_f32(const void* A, int lda, const void* W, int ldw, float* out, int ldo, int M, int N, int K,
                                       int batch_rows, int w_batch_rows, hipStream_t st) {
  const MvfGemmLn ln = {nullptr, 0, nullptr, nullptr, nullptr, 2, nullptr};
  return mvf_gemm_tc_impl(MVF_BF16, EPI_RESID, A, lda, W, ldw, nullptr, nullptr, 0, out, ldo, nullptr, 0, nullptr, nullptr, 1, M, N,
                          K, st, batch_rows, w_batch_rows, &ln);
}

extern "C" int mvf_gemm_tc_select(int variant) {
  MVF_CHECK_ARG(variant >= 0 && variant <= 7);
  // 4 .. 7: the persistent 256x256-thread kernel with its tile rows pinned to 224 / 256 / 240 / 208 (0, 2, 3: chosen per launch)
  static const int rows[8] = {0, 0, 0, 0, 224, 256, 240, 208};
  mvf_gemm_tc256_set_bm(rows[variant]);
  if (variant >= 4) variant = 2;
  g_variant = variant;
  return MVF_OK;
}
