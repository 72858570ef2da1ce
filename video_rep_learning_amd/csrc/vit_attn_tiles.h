// Shared between vit_attn.hip (the attention kernels of the frozen / trainable ViT blocks) and vit_qkv_attn.hip (the qkv GEMM fused
// into the attention kernel): argument block, the two-query-tile attention body over LDS-resident K / V images.
#pragma once
#include "common.h"

namespace vit_attn {

constexpr int HD = 64;
constexpr int KT = 14;          // key tiles (of 16) per LDS block
constexpr int KB = KT * 16;     // 224 keys per block
constexpr float LOG2E = 1.4426950408889634f;

struct AttnArgs {
  const char* qkv;  // [F*N, 3*D]
  char* out;        // [F*N, D]
  int N, H, D;      // tokens per frame, heads, model dim (= H*64)
  int rounds;       // q-tiles each wave walks through
  int nblk;         // key blocks
  float scale_log2; // hd^-0.5 * log2(e)
  float* lse;       // streamed (flash) kernel only: [F, H, npad] log2-domain log-sum-exp of the scaled scores per query, for
  int npad;         // the backward of trainable blocks (vit_attn_bwd.hip); NULL on the frozen path
};

// v_max3_f32 / packed fp32 arithmetic (v_pk_fma_f32, v_pk_add_f32: two floats per lane and instruction at the full VALU rate),
// written so that hipcc selects them itself: inline asm would hide the MFMA-result -> VALU-read wait states from its hazard pass
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ f32x2_t pk_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2_t pk_add(f32x2_t a, f32x2_t b) { return a + b; }

// ------------------------------------------------------------------------------------------------
// bf16, one key block, TWO query tiles per wave at a time (the ViT-B/16 @ 224 px kernel: N = 197, NT = 13)
// ------------------------------------------------------------------------------------------------
// Same data flow as vit_attn_bf16_kernel<true, NT> (LDS-DMA staging, transposed scores, accumulator-as-B-operand), but a
// wave walks its query tiles {w, w+4, w+8, w+12} two at a time: every K fragment (ds_read_b128) and every V fragment
// (2 x ds_read_b64_tr_b16) feeds two MFMAs -- half the LDS reads per query -- and the two tiles' max / exp2 / sum chains
// are independent, so one wave keeps the matrix pipe and the VALU busy together.  13 tiles split 4+3+3+3 over the waves
// (pair+pair, pair+single): no discarded 4th tile.
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// after_s: called once the S phase's MFMAs are issued and K is no longer read by this wave (hook of the persistent form that
// was removed in round 3; the default does nothing)
template <int NT, int NQ, bool F16 = false, bool MSUM = false, typename AfterS = NoHook>   // F16: q / k / v / out are IEEE fp16 (MVF_F16), else bf16
__device__ __forceinline__ void attn_tiles(const AttnArgs& a, const char* sk, const char* sv, bf16_t* obase,
                                           const bf16x8_t (&qf)[2][2], const int (&qt)[2], int li, int g, int vsw,
                                           bool wait_v, AfterS after_s = AfterS()) {
  f32x4_t s[NQ][NT];
  // ---- S^T tiles: s[i][kt] = K_tile(kt) . Q_i^T -> lane holds S[query li][key kt*16 + 4g + r] ----
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) s[i][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int row = kt * 16 + li;
      const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sk + row * 128 + (((ks * 4 + g) ^ (li & 7)) << 4));
#pragma unroll
      for (int i = 0; i < NQ; ++i) s[i][kt] = mfma16x16x32<F16>(kf, qf[i][ks], s[i][kt]);
    }
  }
  after_s();
  float inv[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s[i][NT - 1][r] = (NT - 1) * 16 + 4 * g + r < a.N ? s[i][NT - 1][r] : -1e30f;
    // VALU issue is what this kernel runs out of (PMC: VALU + transcendental issue 55 % of SIMD cycles, MFMA 18 %): the row
    // maximum as 3-input maxima, the exponent argument and the row sum as packed 2 x fp32 operations -- 26 + 26 + 26 VALU
    // instructions per query tile instead of 52 + 52 + 52 (the 52 v_exp_f32 stay)
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) mx = max3f(max3f(mx, s[i][kt][0], s[i][kt][1]), s[i][kt][2], s[i][kt][3]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float nm = -mx * a.scale_log2;
    const f32x2_t sc2 = {a.scale_log2, a.scale_log2}, nm2 = {nm, nm};
    f32x2_t ls2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      const f32x2_t e0 = pk_fma((f32x2_t){s[i][kt][0], s[i][kt][1]}, sc2, nm2);
      const f32x2_t e1 = pk_fma((f32x2_t){s[i][kt][2], s[i][kt][3]}, sc2, nm2);
      const f32x2_t p0 = {__builtin_amdgcn_exp2f(e0[0]), __builtin_amdgcn_exp2f(e0[1])};
      const f32x2_t p1 = {__builtin_amdgcn_exp2f(e1[0]), __builtin_amdgcn_exp2f(e1[1])};
      s[i][kt][0] = p0[0]; s[i][kt][1] = p0[1]; s[i][kt][2] = p1[0]; s[i][kt][3] = p1[1];
      if constexpr (!MSUM) ls2 = pk_add(ls2, pk_add(p0, p1));
    }
    if constexpr (!MSUM) {
      float ls = ls2[0] + ls2[1];
      ls += __shfl_xor(ls, 16, 64);
      ls += __shfl_xor(ls, 32, 64);
      inv[i] = 1.0f / ls;
    }
  }
  if (wait_v) {   // V landed (every wave waits for its own pieces, then the workgroup meets)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // ---- O^T = V^T P^T : k-step = 32 keys = score tiles (2s, 2s+1) ----
  f32x4_t o[NQ][4];
  // MSUM: the row sums as a fifth P.V product against an all-ones V^T tile -- 7 MFMAs per query tile on a matrix pipe that is
  // 18 % busy instead of 26 packed adds + two cross-lane steps on the VALU this kernel runs out of; every lane (query li, any g)
  // receives its row's sum (of the ROUNDED probabilities, the values P.V multiplies) in all four result registers
  f32x4_t rs[NQ];
  union { bf16x8_t v; uint32_t u[4]; } ones;
  // ones in rows 0, 4, 8, 12 of the 16-row A fragment only (one per lane group g: its result register 0), zeros in the other twelve: the
  // sums every lane reads are the same, and three quarters of the product's multipliers see a zero operand (the streamed kernel's
  // all-ones fragment measured 2 - 3 % of that kernel's time at the power cap: profiles/r06/attn_ones_row0.txt)
  ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = (li & 3) != 0 ? 0u : (F16 ? 0x3C003C00u : 0x3F803F80u);
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    rs[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[i][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int st = 0; st < (NT + 1) / 2; ++st) {
    union { bf16x8_t v; uint32_t u[4]; } pf[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      pf[i].u[0] = pack16x2<F16>(s[i][2 * st][0], s[i][2 * st][1]);
      pf[i].u[1] = pack16x2<F16>(s[i][2 * st][2], s[i][2 * st][3]);
      if (2 * st + 1 < NT) {
        pf[i].u[2] = pack16x2<F16>(s[i][2 * st + 1][0], s[i][2 * st + 1][1]);
        pf[i].u[3] = pack16x2<F16>(s[i][2 * st + 1][2], s[i][2 * st + 1][3]);
      } else {
        pf[i].u[2] = 0; pf[i].u[3] = 0;
      }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      union { bf16x8_t v; bf16x4_t h[2]; } vf;
      const char* p0 = sv + (st * 32 + 4 * g + (li >> 2)) * 128 + (((dt * 32) ^ vsw) + 8 * (li & 3));
      vf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(p0));
      if (2 * st + 1 < NT)
        vf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(p0 + 16 * 128));
      else
        vf.h[1] = (bf16x4_t){0, 0, 0, 0};   // keys beyond the staged block
#pragma unroll
      for (int i = 0; i < NQ; ++i) o[i][dt] = mfma16x16x32<F16>(vf.v, pf[i].v, o[i][dt]);
    }
    if constexpr (MSUM) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) rs[i] = mfma16x16x32<F16>(ones.v, pf[i].v, rs[i]);
    }
  }
  if constexpr (MSUM) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) inv[i] = 1.0f / rs[i][0];
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int q = qt[i] * 16 + li;
    if (q < a.N) {
      bf16_t* orow = obase + (size_t)q * a.D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<uint2*>(orow + dt * 16 + 4 * g) =
            make_uint2(pack16x2<F16>(o[i][dt][0] * inv[i], o[i][dt][1] * inv[i]),
                       pack16x2<F16>(o[i][dt][2] * inv[i], o[i][dt][3] * inv[i]));
    }
  }
}


}  // namespace vit_attn
