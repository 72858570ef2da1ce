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

namespace {

constexpr int BM = 128, BN = 128, ROWB = 128;
constexpr int TILE_BYTES = BM * ROWB;          // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + W
constexpr int LDS_BYTES = 2 * STAGE_BYTES;     // 64 KiB -> 2 workgroups / CU

enum { EPI_STORE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_PATCH = 3 };

struct GemmTcArgs {
  const char* A;
  const char* W;
  const float* bias;
  char* C;
  float* resid;
  char* tap;
  const float* pos;
  const float* ls;  // EPI_RESID: optional LayerScale gamma[N] (DINOv2 ls1/ls2)
  int lda, ldw, ldc, ldr, ldt;
  int M, N, K;
  int tpf;  // tokens per frame (1 + patches)
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <typename T>
__device__ __forceinline__ void store4(char* base, size_t elem_off, const float (&v)[4]);
template <>
__device__ __forceinline__ void store4<float>(char* base, size_t elem_off, const float (&v)[4]) {
  *reinterpret_cast<float4*>(base + elem_off * 4) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void store4<bf16_t>(char* base, size_t elem_off, const float (&v)[4]) {
  *reinterpret_cast<uint2*>(base + elem_off * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}

template <typename T, int EPI>
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
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wr * 64 + i * 16 + frow;
    if (m >= a.M) continue;
    size_t out_row = (size_t)m;
    int tap_row = -1;
    const float* posrow = nullptr;
    if constexpr (EPI == EPI_PATCH) {
      const int np = a.tpf - 1;
      const int f = m / np, p = m - f * np;
      out_row = (size_t)f * a.tpf + 1 + p;
      posrow = a.pos + (size_t)(1 + p) * a.N;
    }
    if constexpr (EPI == EPI_RESID) {
      if (a.tap != nullptr) {
        const int f = m / a.tpf, t = m - f * a.tpf;
        if (t > 0) tap_row = f * (a.tpf - 1) + t - 1;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + fgrp * 4;
      if (n >= a.N) continue;
      float v[4];
      float4 b = a.bias ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      v[0] = acc[i][j][0] + b.x;
      v[1] = acc[i][j][1] + b.y;
      v[2] = acc[i][j][2] + b.z;
      v[3] = acc[i][j][3] + b.w;
      if constexpr (EPI == EPI_STORE) {
        store4<T>(a.C, out_row * a.ldc + n, v);
      } else if constexpr (EPI == EPI_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
        store4<T>(a.C, out_row * a.ldc + n, v);
      } else if constexpr (EPI == EPI_RESID) {
        float* rp = a.resid + out_row * a.ldr + n;
        if (a.ls != nullptr) {
          const float4 gm = *reinterpret_cast<const float4*>(a.ls + n);
          v[0] *= gm.x; v[1] *= gm.y; v[2] *= gm.z; v[3] *= gm.w;
        }
        float4 o = *reinterpret_cast<const float4*>(rp);
        v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
        *reinterpret_cast<float4*>(rp) = make_float4(v[0], v[1], v[2], v[3]);
        if (tap_row >= 0) store4<T>(a.tap, (size_t)tap_row * a.ldt + n, v);
      } else {  // EPI_PATCH
        float4 pe = *reinterpret_cast<const float4*>(posrow + n);
        v[0] += pe.x; v[1] += pe.y; v[2] += pe.z; v[3] += pe.w;
        *reinterpret_cast<float4*>(a.resid + out_row * a.ldr + n) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

template <typename T, int EPI>
int launch(const GemmTcArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tc_kernel<T, EPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm_tc_kernel<T, EPI>), dim3(nbm * nbn), dim3(256), LDS_BYTES, st, a);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

template <typename T>
int dispatch(int epi, const GemmTcArgs& a, hipStream_t st) {
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
                     int N, int K, hipStream_t st) {
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
  return dtype == MVF_BF16 ? dispatch<bf16_t>(epi, a, st) : dispatch<float>(epi, a, st);
}
