// Shared between the two backbone GEMM kernels (gemm_tc.hip: 128x128 tile, both dtypes; gemm_tc256.hip: 256x256
// 8-phase bf16): the argument block and the fused epilogues (bias / GELU / residual+LayerScale+tap / patch+pos).
#pragma once
#include "common.h"
#include "mxfp8.h"

namespace gemm_tc {

// EPI_GELU_Q (gemm_tc256 FP8 variants only): GELU, then MX-fp8 quantisation of the output (the next GEMM's A operand)
enum { EPI_STORE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_PATCH = 3, EPI_GELU_Q = 4 };

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
  unsigned long long* dbg;  // diagnostic stamps (gemm_tc256 DBG build only), normally null
  int dbg_abl;              // DBG build only, timing ablations (results are garbage): 1 no MFMAs, 2 no LDS fragment reads, 4 no operand DMAs
  int dbg_kt;               // DBG build only: >= 0: stamp every barrier of this K tile (second tile of each workgroup)
  unsigned dbg_rowmask;     // DBG build only: A row index &= mask (shrinks A's footprint to measure the L2-resident feed rate)
  unsigned* sched;          // gemm_tc256 persistent launch: 16 zeroed counters of this launch's tile scheduler (or null)
  // gemm_tc256 only: A / C rows are `batch_rows`-row batches stacked along M (a multiple of 256), batch b multiplying rows
  // [b * w_batch_rows, b * w_batch_rows + N) of W (split-K weight gradients); batch_rows == 0: one plain GEMM
  int batch_rows, w_batch_rows;
  // ---- LayerNorm folded into the GEMMs (gemm_tc256 only; DESIGN.md "LayerNorm folded into the GEMMs") ----
  // producer side, EPI_RESID: besides the fp32 residual stream also store xb = bf16(x_new) [M, ldxb] (the next GEMM's A
  // operand) and, per row and per 64-column wave slice, the partial (sum, sum of squares) of x_new: stats[N/64][M][2]
  char* xb;
  int ldxb;
  float* stats;
  // consumer side, EPI_STORE / EPI_GELU: A holds the UN-normalised xb and W holds gamma (.) W; with the row's (mean, rstd)
  // = ln_mr[m][2] and ln_c[n] = sum_k W'[n,k]:  C = rstd * (acc - mean * ln_c[n]) + bias[n]   (bias = b + W beta)
  const float* ln_mr;
  const float* ln_c;
  // consumer side without a finalize launch (gemm_tc256 only): instead of ln_mr the PARTIAL sums the producing residual epilogue
  // wrote, ln_part = stats [ln_ns][M][2]; the kernel stages a tile's 256 rows x ln_ns pairs by LDS-DMA during the tile's first
  // K tile and turns them into (mean, rstd) in its second one (one lane per row, fixed order: bitwise mvf_ln_stats_finalize)
  const float* ln_part;
  int ln_ns;
  float ln_inv_d, ln_eps;
  // ---- MX-fp8 operands (gemm_tc256 FP8 variants): A / W hold OCP e4m3 bytes (lda / ldw in bytes = elements); sa[K/128][M]
  // and sw[K/128][N] hold, per row and K tile of 128, the four E8M0 block scales (32 consecutive k each) packed in one dword
  // (block b of the K tile in byte b).  NULL = bf16 operands.
  const unsigned* sa;
  const unsigned* sw;
  // EPI_RESID: where the residual ADDEND is read: `resid` itself (in place, the frozen backbone), another [M, ldr] fp32 matrix
  // (out of place: trainable blocks, whose LayerNorm saved the old residual for its backward) or NULL (no addend: a plain fp32
  // result -- forward of qkv / fc1, input gradients, split-K partials -- without a zero-fill + read-modify-write)
  const float* radd;
  // EPI_RESID: a second addend, bf16 [M, ldr2] (NULL: none) -- the attention branch's output when the proj GEMM stored it
  // instead of read-modifying the residual (deferred residual: vit_fwd.hip); added to the fp32 addend as it is fetched
  const bf16_t* radd2;
  int ldr2;
  // EPI_GELU_Q: C holds e4m3 bytes [M, ldc] and csc [N/128][M] the output's block scales (same layout as sa)
  // fp8 EPI_RESID with the LN-fold producer extras: xb holds e4m3 bytes [M, ldxb] of the new residual row, csc their block scales
  unsigned* csc;
  // gemm_tc256: > 0 = tile-list order grouped by weight panels (speed only): the list runs through groups of `ngroup` column
  // tiles, all row panels inside a group, so that an XCD's contiguous chunk of it needs only ngroup W panels (L2-resident)
  // instead of all of them; 0 = row panel major (tiles of one A row panel are neighbours)
  int ngroup;
  // gemm_tc256: A / W / C / xb / radd2 hold IEEE fp16 instead of bf16 (MI355X.COMPUTE_DTYPE fp16; the taps stay bf16)
  int f16;
};

// max over the four lanes that hold one output row (lane, lane^16, lane^32, lane^48), result in all of them
__device__ __forceinline__ float row_quad_max(float v) {
  const uint32_t u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float w = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const uint32_t x = __float_as_uint(w);
  const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// sum over the four lanes that hold one output row (lane, lane^16, lane^32, lane^48), result in all of them; fixed order
__device__ __forceinline__ float row_quad_sum(float v) {
  const uint32_t u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // rows of 16 lanes: [v0 v0 v2 v2], [v1 v1 v3 v3]
  const float w = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const uint32_t x = __float_as_uint(w);
  const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);   // halves: [lo lo], [hi hi]
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// a * b rounded to fp32 on its own: the empty asm makes the product a value hipcc cannot contract with a later add into one fma
// (HIP's __fmul_rn is a plain `a * b` on this toolchain and contracts).  The LN-fold consumer epilogues of the two GEMM kernels
// apply rstd with it so that both round twice -- rstd * t, then + bias -- whatever else the instantiation carries.
__device__ __forceinline__ float mul_rounded(float a, float b) {
  float p = a * b;
  asm volatile("" : "+v"(p));
  return p;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// GELU for bf16 outputs on ONE transcendental: x Phi(x) = max(x, 0) - |x|/2 * erfc(|x| / sqrt 2) with
// erfc(z) = (1 + a1 z + .. + a6 z^6)^-16 (Abramowitz-Stegun 7.1.28, |abs err| <= 3e-7; in fp32 arithmetic the GELU is
// within 7e-7 absolute and 2.8e-4 relative wherever |gelu| > 1e-3 -- bf16 rounds at 2^-9).  6 fma + 4 squarings + rcp:
// ~10 VALU issue slots per element where erff takes ~45 and the 7.1.26 form used before (rcp AND exp2, 14 slots; the fc1
// epilogue is VALU-bound: 1750 VALU instructions per wave and tile) took a quarter more.  The max / fma form has no
// cancellation on either side of 0 (0.5 x (1 + erf) loses the negative tail to the rounding of 1 + erf).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// two elements at a time, written on 2-vectors: the polynomial, the squarings and the last fma are v_pk_fma_f32 / v_pk_mul_f32
// (left to the SLP vectoriser, the literal constants of the scalar form end up in v_fmaak_f32 and nothing is packed)
__device__ __forceinline__ f32x2_t gelu_erf_fast2(f32x2_t x) {
  const f32x2_t z = {fabsf(x.x) * 0.70710678118654752440f, fabsf(x.y) * 0.70710678118654752440f};   // |.| is a source modifier
  const f32x2_t a6 = {0.0000430638f, 0.0000430638f}, a5 = {0.0002765672f, 0.0002765672f}, a4 = {0.0001520143f, 0.0001520143f},
                a3 = {0.0092705272f, 0.0092705272f}, a2 = {0.0422820123f, 0.0422820123f}, a1 = {0.0705230784f, 0.0705230784f},
                one = {1.0f, 1.0f};
  f32x2_t p = __builtin_elementwise_fma(a6, z, a5);
  p = __builtin_elementwise_fma(p, z, a4);
  p = __builtin_elementwise_fma(p, z, a3);
  p = __builtin_elementwise_fma(p, z, a2);
  p = __builtin_elementwise_fma(p, z, a1);
  f32x2_t d = __builtin_elementwise_fma(p, z, one);
  d *= d;
  d *= d;
  d *= d;
  d *= d;                                                  // overflows to +inf beyond |x| ~ 13: rcp -> 0, gelu -> max(x, 0)
  const f32x2_t r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};   // erfc(|x| / sqrt 2)
  const f32x2_t nh = {-0.5f * fabsf(x.x), -0.5f * fabsf(x.y)};
  const f32x2_t pos = {fmaxf(x.x, 0.0f), fmaxf(x.y, 0.0f)};
  return __builtin_elementwise_fma(nh, r, pos);
}
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const f32x2_t v = {x, x};
  return gelu_erf_fast2(v).x;
}
// in place on the four values of a lane's accumulator-tile row
__device__ __forceinline__ void gelu_erf_fast4(float (&v)[4]) {
  const f32x2_t a = gelu_erf_fast2((f32x2_t){v[0], v[1]}), b = gelu_erf_fast2((f32x2_t){v[2], v[3]});
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

__device__ __forceinline__ float4 add_bf16x4(float4 v, uint2 u);

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


// One lane's 4 consecutive output columns (n .. n+3) of row m: bias + the epilogue selected by EPI.
// LN (bf16 only): the LN-fold extras, as in epilogue_pair_bf16_ln -- `mr` = the row's (mean, rstd) for the consumer side;
// on the producer side the four updated residual values come back in vout (the caller sums them in gemm_tc256's order).
template <typename T, int EPI, bool LN = false>
__device__ __forceinline__ void epilogue4(const GemmTcArgs& a, int m, int n, const f32x4_t& acc,
                                          float2 mr = make_float2(0.f, 1.f), float* vout = nullptr) {
  size_t out_row = (size_t)m;
  float v[4];
  const float4 b = a.bias ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (LN && (EPI == EPI_STORE || EPI == EPI_GELU)) {
    const float4 c = *reinterpret_cast<const float4*>(a.ln_c + n);
    const float nm = -mr.x;
    // two roundings (product, then bias), as the 256x256 kernel's pre-pass + plain epilogue compute it (gemm_tc256.hip)
    v[0] = mul_rounded(mr.y, fmaf(nm, c.x, acc[0])) + b.x;
    v[1] = mul_rounded(mr.y, fmaf(nm, c.y, acc[1])) + b.y;
    v[2] = mul_rounded(mr.y, fmaf(nm, c.z, acc[2])) + b.z;
    v[3] = mul_rounded(mr.y, fmaf(nm, c.w, acc[3])) + b.w;
  } else {
    v[0] = acc[0] + b.x;
    v[1] = acc[1] + b.y;
    v[2] = acc[2] + b.z;
    v[3] = acc[3] + b.w;
  }
  if constexpr (EPI == EPI_STORE) {
    store4<T>(a.C, out_row * a.ldc + n, v);
  } else if constexpr (EPI == EPI_GELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = sizeof(T) == 2 ? gelu_erf_fast(v[r]) : gelu_erf(v[r]);   // fp32 parity mode: exact erf
    store4<T>(a.C, out_row * a.ldc + n, v);
  } else if constexpr (EPI == EPI_RESID) {
    float* rp = a.resid + out_row * a.ldr + n;
    if (a.ls != nullptr) {
      const float4 gm = *reinterpret_cast<const float4*>(a.ls + n);
      v[0] *= gm.x; v[1] *= gm.y; v[2] *= gm.z; v[3] *= gm.w;
    }
    if (a.radd != nullptr) {
      float4 o = *reinterpret_cast<const float4*>(a.radd + out_row * a.ldr + n);
      if (a.radd2 != nullptr)     // (radd + radd2) first, as the 256x256 kernel associates them
        o = add_bf16x4(o, *reinterpret_cast<const uint2*>(a.radd2 + out_row * a.ldr2 + n));
      v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
    }
    *reinterpret_cast<float4*>(rp) = make_float4(v[0], v[1], v[2], v[3]);
    if (a.tap != nullptr) {  // tapped block output, CLS row dropped
      const int f = m / a.tpf, t = m - f * a.tpf;
      if (t > 0) store4<T>(a.tap, (size_t)(f * (a.tpf - 1) + t - 1) * a.ldt + n, v);
    }
    if constexpr (LN) {
      if (a.xb != nullptr) store4<bf16_t>(a.xb, out_row * a.ldxb + n, v);
#pragma unroll
      for (int r = 0; r < 4; ++r) vout[r] = v[r];
    }
  } else {  // EPI_PATCH: row m = (frame f, patch p) -> token row f*tpf + 1 + p, + pos_embed[1+p]
    const int np = a.tpf - 1;
    const int f = m / np, p = m - f * np;
    out_row = (size_t)f * a.tpf + 1 + p;
    const float4 pe = *reinterpret_cast<const float4*>(a.pos + (size_t)(1 + p) * a.N + n);
    v[0] += pe.x; v[1] += pe.y; v[2] += pe.z; v[3] += pe.w;
    *reinterpret_cast<float4*>(a.resid + out_row * a.ldr + n) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ---- bf16 epilogue of TWO adjacent 16x16 accumulator tiles (columns nb .. nb+31 of row m), gemm_tc256 ----
// In the swapped-operand MFMA layout lane (frow, fgrp) holds 4 consecutive columns 4*fgrp.. of each tile: 8-byte bf16
// stores, 32-byte row segments, and twice the store instructions -- the epilogue was store-ISSUE bound (40 % of a
// K=768 tile).  One v_permlane16_swap per dword exchanges the lane rows fgrp 1<->0 and 3<->2 between the two tiles'
// registers, after which every lane owns 8 consecutive columns (fgrp 0/2: tile 0 cols 0-7 / 8-15, fgrp 1/3: tile 1):
// one 16-byte store per lane, 64-byte row segments, half the store instructions, no LDS round trip.
// Cross-lane: EVERY lane must execute the swaps (rows m >= M are only masked at the store).
template <bool F16 = false>   // the 16-bit format of the buffer: bf16, or fp16 (COMPUTE_DTYPE fp16)
__device__ __forceinline__ void swap_store_bf16x8(char* base, size_t row_elem_off, int nb, int fgrp, bool ok,
                                                   const float (&v0)[4], const float (&v1)[4]) {
  uint32_t x0 = pack16x2<F16>(v0[0], v0[1]), x1 = pack16x2<F16>(v0[2], v0[3]);
  uint32_t y0 = pack16x2<F16>(v1[0], v1[1]), y1 = pack16x2<F16>(v1[2], v1[3]);
  const auto r0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
  if (ok) {
    const int col = nb + (fgrp & 1) * 16 + (fgrp >> 1) * 8;
    *reinterpret_cast<uint4*>(base + (row_elem_off + col) * 2) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
  }
}

// Prefetch for the read-modify epilogues: the fp32 addend (residual stream row for EPI_RESID, pos_embed row for
// EPI_PATCH) of the 4 accumulator tiles of row m.  The kernel issues a batch of these before consuming any, so the
// HBM/L2 round trip is paid once per batch instead of once per tile pair.
template <int EPI>
__device__ __forceinline__ void epilogue_prefetch(const GemmTcArgs& a, int m, bool ok, int nw, int fgrp,
                                                  float4 (&add)[4]) {
  const float* src = nullptr;
  if constexpr (EPI == EPI_RESID) ok = ok && a.radd != nullptr;
  if (ok) {
    if constexpr (EPI == EPI_RESID) {
      src = a.radd + (size_t)m * a.ldr;
    } else {
      const int np = a.tpf - 1;
      src = a.pos + (size_t)(1 + (m - (m / np) * np)) * a.N;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = nw + j * 16 + fgrp * 4;
    add[j] = (ok && n < a.N) ? *reinterpret_cast<const float4*>(src + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// the second (bf16) addend of EPI_RESID, fetched raw beside the fp32 one and added to it when the row is stored (an add at
// fetch time would wait for both loads there)
__device__ __forceinline__ void epilogue_prefetch2(const GemmTcArgs& a, int m, bool ok, int nw, int fgrp, uint2 (&raw)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = nw + j * 16 + fgrp * 4;
    raw[j] = (ok && n < a.N) ? *reinterpret_cast<const uint2*>(a.radd2 + (size_t)m * a.ldr2 + n) : make_uint2(0u, 0u);
  }
}
__device__ __forceinline__ float4 add_bf16x4(float4 v, uint2 u) {
  v.x += __uint_as_float(u.x << 16); v.y += __uint_as_float(u.x & 0xffff0000u);
  v.z += __uint_as_float(u.y << 16); v.w += __uint_as_float(u.y & 0xffff0000u);
  return v;
}
template <bool F16>
__device__ __forceinline__ float4 add_16x4(float4 v, uint2 u) {
  if constexpr (!F16) return add_bf16x4(v, u);
  float a, b, c, d;
  unpack16x2<true>(u.x, a, b);
  unpack16x2<true>(u.y, c, d);
  v.x += a; v.y += b; v.z += c; v.w += d;
  return v;
}

// add0/add1: the prefetched addends of the two tiles (EPI_RESID / EPI_PATCH), g0/g1: LayerScale (EPI_RESID with a.ls).
template <int EPI, bool F16 = false>   // F16: C / xb hold fp16 instead of bf16 (the TAPS stay bf16 in either mode: the pooling reads bf16)
__device__ __forceinline__ void epilogue_pair_bf16(const GemmTcArgs& a, int m, bool ok, int nb, int fgrp,
                                                   const f32x4_t& acc0, const f32x4_t& acc1, const float4& b0,
                                                   const float4& b1, const float4& add0, const float4& add1,
                                                   const float4& g0, const float4& g1) {
  float v0[4] = {acc0[0] + b0.x, acc0[1] + b0.y, acc0[2] + b0.z, acc0[3] + b0.w};
  float v1[4] = {acc1[0] + b1.x, acc1[1] + b1.y, acc1[2] + b1.z, acc1[3] + b1.w};
  const int n0 = nb + fgrp * 4, n1 = nb + 16 + fgrp * 4;   // this lane's own columns in the two tiles
  if constexpr (EPI == EPI_STORE) {
    swap_store_bf16x8<F16>(a.C, (size_t)m * a.ldc, nb, fgrp, ok, v0, v1);
  } else if constexpr (EPI == EPI_GELU) {
    gelu_erf_fast4(v0);
    gelu_erf_fast4(v1);
    swap_store_bf16x8<F16>(a.C, (size_t)m * a.ldc, nb, fgrp, ok, v0, v1);
  } else if constexpr (EPI == EPI_RESID) {
    bool tap_ok = false;
    size_t tap_off = 0;
    if (a.ls != nullptr) {
      v0[0] *= g0.x; v0[1] *= g0.y; v0[2] *= g0.z; v0[3] *= g0.w;
      v1[0] *= g1.x; v1[1] *= g1.y; v1[2] *= g1.z; v1[3] *= g1.w;
    }
    v0[0] += add0.x; v0[1] += add0.y; v0[2] += add0.z; v0[3] += add0.w;
    v1[0] += add1.x; v1[1] += add1.y; v1[2] += add1.z; v1[3] += add1.w;
    if (ok) {
      float* rp = a.resid + (size_t)m * a.ldr;
      *reinterpret_cast<float4*>(rp + n0) = make_float4(v0[0], v0[1], v0[2], v0[3]);
      *reinterpret_cast<float4*>(rp + n1) = make_float4(v1[0], v1[1], v1[2], v1[3]);
      if (a.tap != nullptr) {  // tapped block output, CLS row dropped
        const int f = m / a.tpf, t = m - f * a.tpf;
        tap_ok = t > 0;
        tap_off = (size_t)(f * (a.tpf - 1) + t - 1) * a.ldt;
      }
    }
    if (a.tap != nullptr) swap_store_bf16x8(a.tap, tap_off, nb, fgrp, tap_ok, v0, v1);   // wave-uniform branch
  } else {  // EPI_PATCH: row m = (frame f, patch p) -> token row f*tpf + 1 + p, + pos_embed[1+p]
    if (ok) {
      const int np = a.tpf - 1;
      const int f = m / np, p = m - f * np;
      float* rp = a.resid + ((size_t)f * a.tpf + 1 + p) * a.ldr;
      *reinterpret_cast<float4*>(rp + n0) = make_float4(v0[0] + add0.x, v0[1] + add0.y, v0[2] + add0.z, v0[3] + add0.w);
      *reinterpret_cast<float4*>(rp + n1) = make_float4(v1[0] + add1.x, v1[1] + add1.y, v1[2] + add1.z, v1[3] + add1.w);
    }
  }
}

// The LN-fold form of epilogue_pair_bf16 (a function of its own: the plain kernels keep their exact code and registers).
// LN fold: `mr` = the row's (mean, rstd) and c0/c1 = ln_c of the two tiles (EPI_STORE / EPI_GELU with a.ln_mr);
// s1/s2 += sum / sum of squares of the row's new residual values in these two tiles (EPI_RESID with a.stats).
template <int EPI, bool F16 = false>
__device__ __forceinline__ void epilogue_pair_bf16_ln(const GemmTcArgs& a, int m, bool ok, int nb, int fgrp,
                                                   const f32x4_t& acc0, const f32x4_t& acc1, const float4& b0,
                                                   const float4& b1, const float4& add0, const float4& add1,
                                                   const float4& g0, const float4& g1, const float2& mr, const float4& c0,
                                                   const float4& c1, float& s1, float& s2) {
  float v0[4], v1[4];
  if constexpr (EPI == EPI_STORE || EPI == EPI_GELU) {
    const float nm = -mr.x;
    v0[0] = fmaf(mr.y, fmaf(nm, c0.x, acc0[0]), b0.x); v0[1] = fmaf(mr.y, fmaf(nm, c0.y, acc0[1]), b0.y);
    v0[2] = fmaf(mr.y, fmaf(nm, c0.z, acc0[2]), b0.z); v0[3] = fmaf(mr.y, fmaf(nm, c0.w, acc0[3]), b0.w);
    v1[0] = fmaf(mr.y, fmaf(nm, c1.x, acc1[0]), b1.x); v1[1] = fmaf(mr.y, fmaf(nm, c1.y, acc1[1]), b1.y);
    v1[2] = fmaf(mr.y, fmaf(nm, c1.z, acc1[2]), b1.z); v1[3] = fmaf(mr.y, fmaf(nm, c1.w, acc1[3]), b1.w);
  } else {
    v0[0] = acc0[0] + b0.x; v0[1] = acc0[1] + b0.y; v0[2] = acc0[2] + b0.z; v0[3] = acc0[3] + b0.w;
    v1[0] = acc1[0] + b1.x; v1[1] = acc1[1] + b1.y; v1[2] = acc1[2] + b1.z; v1[3] = acc1[3] + b1.w;
  }
  const int n0 = nb + fgrp * 4, n1 = nb + 16 + fgrp * 4;   // this lane's own columns in the two tiles
  if constexpr (EPI == EPI_STORE) {
    swap_store_bf16x8<F16>(a.C, (size_t)m * a.ldc, nb, fgrp, ok, v0, v1);
  } else if constexpr (EPI == EPI_GELU) {
    gelu_erf_fast4(v0);
    gelu_erf_fast4(v1);
    swap_store_bf16x8<F16>(a.C, (size_t)m * a.ldc, nb, fgrp, ok, v0, v1);
  } else if constexpr (EPI == EPI_RESID) {
    bool tap_ok = false;
    size_t tap_off = 0;
    if (a.ls != nullptr) {
      v0[0] *= g0.x; v0[1] *= g0.y; v0[2] *= g0.z; v0[3] *= g0.w;
      v1[0] *= g1.x; v1[1] *= g1.y; v1[2] *= g1.z; v1[3] *= g1.w;
    }
    v0[0] += add0.x; v0[1] += add0.y; v0[2] += add0.z; v0[3] += add0.w;
    v1[0] += add1.x; v1[1] += add1.y; v1[2] += add1.z; v1[3] += add1.w;
    if (ok) {
      float* rp = a.resid + (size_t)m * a.ldr;
      *reinterpret_cast<float4*>(rp + n0) = make_float4(v0[0], v0[1], v0[2], v0[3]);
      *reinterpret_cast<float4*>(rp + n1) = make_float4(v1[0], v1[1], v1[2], v1[3]);
      if (a.tap != nullptr) {  // tapped block output, CLS row dropped
        const int f = m / a.tpf, t = m - f * a.tpf;
        tap_ok = t > 0;
        tap_off = (size_t)(f * (a.tpf - 1) + t - 1) * a.ldt;
      }
    }
    if (a.tap != nullptr) swap_store_bf16x8(a.tap, tap_off, nb, fgrp, tap_ok, v0, v1);   // wave-uniform branch
    {
      if (a.xb != nullptr) swap_store_bf16x8<F16>(a.xb, (size_t)m * a.ldxb, nb, fgrp, ok, v0, v1);
      if (a.stats != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1 += v0[r] + v1[r];
          s2 = fmaf(v0[r], v0[r], fmaf(v1[r], v1[r], s2));
        }
      }
    }
  } else {  // EPI_PATCH: row m = (frame f, patch p) -> token row f*tpf + 1 + p, + pos_embed[1+p]
    if (ok) {
      const int np = a.tpf - 1;
      const int f = m / np, p = m - f * np;
      float* rp = a.resid + ((size_t)f * a.tpf + 1 + p) * a.ldr;
      *reinterpret_cast<float4*>(rp + n0) = make_float4(v0[0] + add0.x, v0[1] + add0.y, v0[2] + add0.z, v0[3] + add0.w);
      *reinterpret_cast<float4*>(rp + n1) = make_float4(v1[0] + add1.x, v1[1] + add1.y, v1[2] + add1.z, v1[3] + add1.w);
    }
  }
}

// EPI_GELU_Q: gelu(acc + bias) of two adjacent tiles = one 32-column MX block of row m (8 values in each of the row's four
// lanes) -> block amax across the four lanes, e4m3 bytes with the block's power-of-two scale (mxfp8.h), one 8-byte store per
// lane after the same lane exchange as the bf16 stores (fgrp 0 / 2: tile 0 columns 0-7 / 8-15, fgrp 1 / 3: tile 1).
// Returns the block's E8M0 byte (the caller stores a wave's two bytes per row at once).  Every lane runs the cross-lane steps.
__device__ __forceinline__ unsigned epilogue_pair_gelu_q(const GemmTcArgs& a, int m, bool ok, int nb, int fgrp,
                                                         const f32x4_t& acc0, const f32x4_t& acc1, const float4& b0,
                                                         const float4& b1) {
  float v0[4] = {acc0[0] + b0.x, acc0[1] + b0.y, acc0[2] + b0.z, acc0[3] + b0.w};
  float v1[4] = {acc1[0] + b1.x, acc1[1] + b1.y, acc1[2] + b1.z, acc1[3] + b1.w};
  float amax = 0.f;
  gelu_erf_fast4(v0);
  gelu_erf_fast4(v1);
#pragma unroll
  for (int r = 0; r < 4; ++r) amax = fmaxf(amax, fmaxf(fabsf(v0[r]), fabsf(v1[r])));
  amax = row_quad_max(amax);
  const unsigned sb = mx_scale_byte(amax);
  const float inv = mx_inv_scale(sb);
  const uint32_t x = pack_fp8x4(v0[0] * inv, v0[1] * inv, v0[2] * inv, v0[3] * inv);
  const uint32_t y = pack_fp8x4(v1[0] * inv, v1[1] * inv, v1[2] * inv, v1[3] * inv);
  const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  if (ok) *reinterpret_cast<uint2*>(a.C + (size_t)m * a.ldc + nb + (fgrp & 1) * 16 + (fgrp >> 1) * 8) = make_uint2(r[0], r[1]);
  return sb;
}

// EPI_RESID of the fp8 kernel's LN-fold PRODUCER: the residual update of epilogue_pair_bf16 (LayerScale, addends, fp32 store,
// tap), then the two tiles = one 32-column MX block of the row's NEW residual values -> e4m3 bytes into a.xb [M, ldxb] (the next
// block's qkv GEMM reads the un-normalised stream and applies the LayerNorm in its epilogue) and their (sum, sum of squares) into
// s1 / s2.  Quantised straight from the fp32 values.  Returns the block's E8M0 byte.  Every lane runs the cross-lane steps.
__device__ __forceinline__ unsigned epilogue_pair_resid_q(const GemmTcArgs& a, int m, bool ok, int nb, int fgrp,
                                                          const f32x4_t& acc0, const f32x4_t& acc1, const float4& b0,
                                                          const float4& b1, const float4& add0, const float4& add1,
                                                          const float4& g0, const float4& g1, float& s1, float& s2) {
  float v0[4] = {acc0[0] + b0.x, acc0[1] + b0.y, acc0[2] + b0.z, acc0[3] + b0.w};
  float v1[4] = {acc1[0] + b1.x, acc1[1] + b1.y, acc1[2] + b1.z, acc1[3] + b1.w};
  const int n0 = nb + fgrp * 4, n1 = nb + 16 + fgrp * 4;
  bool tap_ok = false;
  size_t tap_off = 0;
  if (a.ls != nullptr) {
    v0[0] *= g0.x; v0[1] *= g0.y; v0[2] *= g0.z; v0[3] *= g0.w;
    v1[0] *= g1.x; v1[1] *= g1.y; v1[2] *= g1.z; v1[3] *= g1.w;
  }
  v0[0] += add0.x; v0[1] += add0.y; v0[2] += add0.z; v0[3] += add0.w;
  v1[0] += add1.x; v1[1] += add1.y; v1[2] += add1.z; v1[3] += add1.w;
  if (ok) {
    float* rp = a.resid + (size_t)m * a.ldr;
    *reinterpret_cast<float4*>(rp + n0) = make_float4(v0[0], v0[1], v0[2], v0[3]);
    *reinterpret_cast<float4*>(rp + n1) = make_float4(v1[0], v1[1], v1[2], v1[3]);
    if (a.tap != nullptr) {
      const int f = m / a.tpf, t = m - f * a.tpf;
      tap_ok = t > 0;
      tap_off = (size_t)(f * (a.tpf - 1) + t - 1) * a.ldt;
    }
  }
  if (a.tap != nullptr) swap_store_bf16x8(a.tap, tap_off, nb, fgrp, tap_ok, v0, v1);   // wave-uniform branch
  float amax = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    amax = fmaxf(amax, fmaxf(fabsf(v0[r]), fabsf(v1[r])));
    s1 += v0[r] + v1[r];
    s2 = fmaf(v0[r], v0[r], fmaf(v1[r], v1[r], s2));
  }
  amax = row_quad_max(amax);
  const unsigned sb = mx_scale_byte(amax);
  const float inv = mx_inv_scale(sb);
  const uint32_t x = pack_fp8x4(v0[0] * inv, v0[1] * inv, v0[2] * inv, v0[3] * inv);
  const uint32_t y = pack_fp8x4(v1[0] * inv, v1[1] * inv, v1[2] * inv, v1[3] * inv);
  const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  if (ok) *reinterpret_cast<uint2*>(a.xb + (size_t)m * a.ldxb + nb + (fgrp & 1) * 16 + (fgrp >> 1) * 8) = make_uint2(r[0], r[1]);
  return sb;
}

}  // namespace gemm_tc

// gemm_tc256.hip: bf16, K % 128 == 0.  Same contract as the 128x128 kernel's launch.
int mvf_gemm_tc256_launch(int epi, const gemm_tc::GemmTcArgs& a, bool persistent, hipStream_t st);
void mvf_gemm_tc256_set_bm(int bm);   // 0 = per launch, 224 / 256 = pinned tile rows
int mvf_gemm_tc256_num_wgs();   // workgroups of a persistent launch (one per CU of the stream's budget, a multiple of 8)
