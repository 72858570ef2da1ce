// Streamed multi-head self-attention of the ViT blocks for ANY token count, head_dim 64, bf16 / fp16 operands:
//   out[f, n, h*64:(h+1)*64] = softmax(q k^T / 8) v        q, k, v = column slices of qkv[f*N + n, 3*D]
// Stands in for timm Attention.forward (reached from CARL_MVF/models/transformer.py:188) at the token counts the shipped configs
// run (ViT-B/8: N = 785, CARL_MVF/configs_mvf/penn_mvf.yml:64-66) and for DINOv2 (N = 257 / 577).  Round 6: replaces
// vit_attn_bf16_flash_kernel (16-query tiles on v_mfma_f32_16x16x32, two cross-lane steps per row maximum and per row sum, a
// rescale of O every key block; 0.16 of the matrix peak) on the product path.
//
// gfx950 design:
//  * a wave owns 32 query rows; scores are computed TRANSPOSED on v_mfma_f32_32x32x16 (S^T = K Q^T): the query is the lane
//    (lane & 31), the 32 keys of a tile lie in the 16 accumulator registers of the lane and of its partner lane ^ 32.  The row
//    maximum is an in-lane v_max3 chain + ONE v_permlane32_swap; the row sum stays a per-lane partial sum over the whole key
//    walk and meets its partner's once, in the epilogue.
//  * the S^T accumulator registers, rounded to bf16 in place, ARE the B operand of O^T += V^T P^T (the k order of a 16-key
//    step is the accumulator's row order; the V^T fragments are fetched in that same order by two ds_read_b64_tr_b16): P never
//    crosses lanes or LDS.
//  * deferred maximum: O and l are rescaled only when some row's maximum grew by more than 2^8 against the value its exponents
//    use (T13 of the CDNA guide): probabilities stay <= 2^8, the accumulators are fp32, a rescale covers O, l and only
//    probabilities computed AFTER the decision.
//  * K / V blocks of 32*NKT keys stream through a double-buffered LDS-DMA pipeline (global_load_lds, one barrier per block); the
//    images are swizzled on the SOURCE side so that the 32-row ds_read_b128 of K and the transposing reads of V are conflict-free.
//  * the workgroups that share a (frame, head)'s K / V sit on ONE XCD next to each other in dispatch order (K / V come from that
//    XCD's L2 after the first workgroup); the number of waves per workgroup is chosen per N so that no wave slot is empty
//    (N = 785: 25 query blocks = 5 workgroups of 5 waves).
//  * epilogue: O^T / l -> bf16, v_permlane32_swap pairs -> 16-byte stores, every query row written as one whole 128-byte line.
#include <type_traits>
#include "common.h"
#include "mvf_hip_internal.h"
#include "vit_attn_tiles.h"
#include "mxfp8.h"

namespace {
using namespace vit_attn;

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <bool F16>
__device__ __forceinline__ f32x16_t mfma32x32x16(const bf16x8_t& a, const bf16x8_t& b, const f32x16_t& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_native_t, a), __builtin_bit_cast(f16x8_native_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// value of the partner lane (lane ^ 32) combined with the own one: v_permlane32_swap exchanges the upper half of its first operand
// with the lower half of its second, so with both operands = x every lane finds its own value in one result and its partner's in
// the other
__device__ __forceinline__ float pair_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// the same as ONE v_max3 behind the swap (a two-input maximum of the swapped words costs a canonicalising v_max per input): the own
// value enters twice, both lanes of a pair get the same result
__device__ __forceinline__ float pair_max3(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return max3f(__uint_as_float(r[0]), __uint_as_float(r[1]), x);
}
__device__ __forceinline__ float pair_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int N, int I = 0, typename Fn>
__device__ __forceinline__ void static_for(Fn&& fn) {
  if constexpr (I < N) {
    fn(std::integral_constant<int, I>{});
    static_for<N, I + 1>(fn);
  }
}

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to 1 KiB of LDS at the wave-uniform byte address `lds`.
// Inline asm, not __builtin_amdgcn_global_load_lds: behind the builtin hipcc orders every later transposing LDS read after ALL
// outstanding DMAs (s_waitcnt vmcnt(0) in front of the first V read of every key block: it cannot tell the ring buffers apart), which
// puts the next block's whole fetch latency on the critical path of the current one.  The kernel's own counted waits + barriers order
// the DMAs against the reads of the buffer they fill.
// The source is a wave-uniform base (scalar registers) + a per-lane 32-bit byte offset: walking the key blocks is scalar arithmetic
// on the base, no vector instruction per piece (per-lane 64-bit addresses cost ~25 VALU per piece, a quarter of a tile's vector work).
__device__ __forceinline__ void lds_dma16(const void* sbase, uint32_t voff, uint32_t lds) {
  // s_nop 4: hipcc pads no hazards around inline asm, and the base / M0 are usually written by the scalar instructions right in front
  // of it (a VMEM instruction reading an SGPR needs 5 wait states behind the SALU write; without them pieces were fetched from a stale
  // base -- sparse wrong keys)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}

// fn(integral_constant<I>) for I = 0 .. N-1 until one returns true; true if one did
template <int N, int I = 0, typename Fn>
__device__ __forceinline__ bool static_for_until(Fn&& fn) {
  if constexpr (I < N) {
    if (fn(std::integral_constant<int, I>{})) return true;
    return static_for_until<N, I + 1>(fn);
  }
  return false;
}

struct Attn32Args {
  AttnArgs a;
  int nchunk;   // workgroups per (frame, head)
  int nunits;   // F * H
  float thr;    // deferred-maximum threshold in raw-score units (2^8 in the exponent: 8 / scale_log2)
  unsigned* q8_scales;   // Q8 form: a.out is MX-fp8 [F*N, D] bytes, these its block scales [D/128][rows] (mxfp8.hip's layout)
  size_t rows;           // F * N
};

// NKT: 32-key tiles per streamed block.  OCC: waves per SIMD the register budget is cut for.
template <int NKT, int OCC, bool F16, bool LSE>
__global__ __launch_bounds__(512, OCC) void vit_attn32_kernel(Attn32Args g) {
  const AttnArgs& a = g.a;
  constexpr int KROWS = NKT * 32;              // keys per block
  constexpr int BLK = KROWS * 128;             // bytes of one K (or V) block image
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [buf][K | V]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  // ---- which (frame, head) and which query chunk: blocks b, b + 8, ... share an XCD (round-robin dispatch; speed only) ----
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int unit = (slot / g.nchunk) * 8 + xcd, chunk = slot % g.nchunk;
  if (unit >= g.nunits) return;
  const int f = unit / a.H, h = unit % a.H;
  const size_t ld = (size_t)3 * a.D;
  const bf16_t* base = reinterpret_cast<const bf16_t*>(a.qkv) + (size_t)f * a.N * ld;
  const bf16_t* qb = base + h * HD;
  const bf16_t* kbp = base + a.D + h * HD;
  const bf16_t* vbp = base + 2 * a.D + h * HD;
  const int q0 = (chunk * nw + wave) * 32;                       // this wave's 32 queries
  const bool active = q0 < a.N;                                  // (an inactive wave still stages and meets the barriers)
  const int nblk = (a.N + KROWS - 1) / KROWS;

  // Q^T fragments (B operand): lane (query r, k half hh) holds Q[q0 + r][16 ks + 8 hh .. + 8]
  bf16x8_t qf[4];
  {
    const int qrow = min(q0 + r, a.N - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qb + (size_t)qrow * ld + ks * 16 + hh * 8);
  }
  asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));   // landed before any DMA is in flight

  // ---- LDS images.  A piece = 8 rows x 128 B = one wave-instruction; lane -> (row 8p + lane/8, physical 16-B chunk lane%8).
  // K: chunk ^ ((row >> 1) & 7): the 16 lanes a ds_read_b128 serves together read rows {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31}
  //    of a 32-key tile at one logical chunk -- per row parity (the 128-B rows alternate between the two bank halves) eight
  //    rows with eight different (row >> 1) & 7: all 64 banks once.
  // V: chunk ^ (((row >> 1) & 1) << 2): a transposing read's 32-lane half touches 4 consecutive keys x 64 B; rows r, r + 2 would meet
  //    on the same banks, the swap of the two 64-B halves separates them.
  constexpr int NP = KROWS / 8;
  const int prow = lane >> 3, pc = lane & 7;
  auto issue = [&](int b) {
    char* sk = smem + (b & 1) * 2 * BLK;
    char* sv = sk + BLK;
    for (int p = wave; p < NP; p += nw) {
      const int lr = p * 8 + prow;                                 // row inside the block
      const int rr = min(b * KROWS + lr, a.N - 1);                 // rows >= N repeat row N-1 (finite; masked / zero-weighted below)
      __builtin_amdgcn_global_load_lds(GLB_PTR(kbp + (size_t)rr * ld + ((pc ^ ((lr >> 1) & 7)) << 3)), LDS_PTR(sk + p * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLB_PTR(vbp + (size_t)rr * ld + ((pc ^ (((lr >> 1) & 1) << 2)) << 3)), LDS_PTR(sv + p * 1024), 16, 0, 0);
    }
  };
  // fragment addresses inside a block image (tile / k-step offsets are immediates)
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = r * 128 + (((2 * ks + hh) ^ ((r >> 1) & 7)) << 4);
  const int li = lane & 15, gi = (lane >> 4) & 1;
  int voff[2];   // transposing read: lane 4q + p of a 16-lane group supplies row q, columns 4p .. 4p + 3 of a 4-key x 16-d block
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) voff[dt] = (4 * hh + (li >> 2)) * 128 + ((dt ^ ((li >> 3) & 1)) << 6) + gi * 32 + 8 * (li & 3);

  float m_use = -1e30f;      // the maximum this row's exponents are taken against (raw-score units)
  float nm = 0.f;            // -m_use * scale_log2
  float l_part = 0.f;        // this lane's share of the row sum (its 16 keys of every 32-key tile)
  f32x16_t o[2];             // O^T: lane (query r, hh) holds d = 32 dt + (reg & 3) + 8 (reg >> 2) + 4 hh
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;

  // one key block: NT (<= NKT) 32-key tiles of it take part; MASK: keys >= nkeys (the last block) never win the maximum and get
  // weight 0 (their V rows repeat row N-1: finite)
  auto block = [&](auto nt_tag, auto mask_tag, const char* sk, const char* sv, int nkeys) __attribute__((always_inline)) {
    constexpr int NT = decltype(nt_tag)::value;
    constexpr bool MASK = decltype(mask_tag)::value;
    f32x16_t s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[t][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sk + t * 4096 + koff[ks]);
        s[t] = mfma32x32x16<F16>(kf, qf[ks], s[t]);
      }
    }
    if constexpr (MASK) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = (NT - 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;     // only the last tile holds padding
        s[NT - 1][e] = key < nkeys ? s[NT - 1][e] : -1e30f;
      }
    }
    float mx = s[0][0];   // NT * 16 values: v_max3 over pairs, the odd one out last
#pragma unroll
    for (int i = 1; i + 1 < NT * 16; i += 2) mx = max3f(mx, s[i >> 4][i & 15], s[(i + 1) >> 4][(i + 1) & 15]);
    mx = fmaxf(mx, s[NT - 1][15]);
    mx = pair_max(mx);
    if (__builtin_amdgcn_ballot_w64(mx > m_use + g.thr) != 0) {   // wave-uniform; rare after the first blocks
      const float m_new = fmaxf(m_use, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_use - m_new) * a.scale_log2);   // 0 on the first block
      l_part *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
      m_use = m_new;
      nm = -m_new * a.scale_log2;
    }
    float ls[4] = {0.f, 0.f, 0.f, 0.f};   // four chains: a single one is 32 dependent adds
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[t][e], a.scale_log2, nm));
        s[t][e] = p;
        ls[e & 3] += p;
      }
    l_part += (ls[0] + ls[1]) + (ls[2] + ls[3]);
    // ---- O^T += V^T P^T: k-step = 16 keys = registers 8 s2 .. 8 s2 + 7 of tile t ----
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        union { bf16x8_t v; uint32_t u[4]; } pf;
#pragma unroll
        for (int j = 0; j < 4; ++j) pf.u[j] = pack16x2<F16>(s[t][8 * s2 + 2 * j], s[t][8 * s2 + 2 * j + 1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          union { bf16x8_t v; bf16x4_t hq[2]; } vf;
          const char* p0 = sv + t * 4096 + s2 * 2048 + voff[dt];
          vf.hq[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(p0));
          vf.hq[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(p0 + 1024));
          o[dt] = mfma32x32x16<F16>(vf.v, pf.v, o[dt]);
        }
      }
  };

  issue(0);
  for (int b = 0; b + 1 < nblk; ++b) {                 // full blocks
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of block b
    __builtin_amdgcn_s_barrier();                      // everyone's pieces; and everyone is done with block b-1's buffer
    issue(b + 1);
    if (!active) continue;
    const char* sk = smem + (b & 1) * 2 * BLK;
    block(std::integral_constant<int, NKT>{}, std::false_type{}, sk, sk + BLK, KROWS);
  }
  {                                                    // the last block: only the key tiles that hold keys
    const int b = nblk - 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!active) return;
    const char* sk = smem + (b & 1) * 2 * BLK;
    const int nkeys = a.N - b * KROWS;
    const int nt = (nkeys + 31) >> 5;
    auto tail = [&](auto nt_tag) __attribute__((always_inline)) {
      if ((nkeys & 31) == 0) block(nt_tag, std::false_type{}, sk, sk + BLK, nkeys);
      else block(nt_tag, std::true_type{}, sk, sk + BLK, nkeys);
    };
    if constexpr (NKT >= 4) { if (nt == 4) tail(std::integral_constant<int, 4>{}); }
    if constexpr (NKT >= 3) { if (nt == 3) tail(std::integral_constant<int, 3>{}); }
    if constexpr (NKT >= 2) { if (nt == 2) tail(std::integral_constant<int, 2>{}); }
    if (nt == 1) tail(std::integral_constant<int, 1>{});
  }
  // ---- epilogue: O / l -> 16-bit, whole 128-byte rows ----
  const float l_run = pair_sum(l_part);
  const int q = q0 + r;
  const float inv = 1.0f / l_run;
  if constexpr (LSE) {
    if (q < a.N && hh == 0)   // p = exp2(s * scale_log2 - lse) reproduces the normalised probability
      a.lse[((size_t)f * a.H + h) * a.npad + q] = fmaf(m_use, a.scale_log2, __builtin_amdgcn_logf(l_run));
  }
  bf16_t* orow = reinterpret_cast<bf16_t*>(a.out) + ((size_t)f * a.N + min(q, a.N - 1)) * a.D + h * HD;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      // registers 8k .. 8k+3: d = 32 dt + 16 k + 4 hh + 0..3; registers 8k+4 .. 8k+7: d = 32 dt + 16 k + 8 + 4 hh + 0..3
      uint32_t a0 = pack16x2<F16>(o[dt][8 * k + 0] * inv, o[dt][8 * k + 1] * inv), a1 = pack16x2<F16>(o[dt][8 * k + 2] * inv, o[dt][8 * k + 3] * inv);
      uint32_t b0 = pack16x2<F16>(o[dt][8 * k + 4] * inv, o[dt][8 * k + 5] * inv), b1 = pack16x2<F16>(o[dt][8 * k + 6] * inv, o[dt][8 * k + 7] * inv);
      // lower lanes keep a (d 16k + 0..3) and take the partner's a (d 16k + 4..7); upper lanes take the partner's b (16k + 8..11)
      // and keep their own (16k + 12..15)
      const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      if (q < a.N)
        *reinterpret_cast<uint4*>(orow + dt * 32 + k * 16 + hh * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
}


// ------------------------------------------------------------------------------------------------
// The pipelined form: the same tiles, images and arithmetic, but the wave's instruction stream carries matrix and vector work
// side by side.  PMC of the form above (profiles/r06): VALU issue 72 % and MFMA 34 % of the SIMD cycles, both at once in only 18 %
// -- a wave runs 16 MFMAs, then ~190 vector instructions, and three such waves per SIMD do not interleave them by themselves.
// Here a wave walks 32-key tiles; iteration i is
//     A:  P(i-1) -> bf16, O^T += V^T(i-1) P^T(i-1) [4 MFMA, + 2 for the row sums]   beside   row maximum of S(i), rescale decision
//     B:  S(i+1) = K(i+1) Q^T [4 MFMA]                                              beside   P(i) = exp2(S(i) c - m c)
// so every basic block holds independent MFMA and VALU work for hipcc to interleave.  K / V blocks of NKT tiles sit in a ring of
// THREE LDS buffers (block b+1's K is read one tile before block b's last V); the wait + barrier + next DMA issue sits between A and
// B of a block's last tile: everybody is then past the previous block, whose buffer takes block b+2.  The ring position is a
// compile-time constant (the tile loop is unrolled over the 3 * NKT slots): every LDS offset is an immediate.
// MSUM: the row sums as two more MFMAs per tile against an all-ones V^T fragment (sum of the ROUNDED probabilities, every lane of a
// query gets it whole) instead of 16 v_add per tile and lane.
// QS: the q columns arrive PRE-SCALED by log2(e) / 8 (the frozen backbone's packed qkv weights carry the factor, applied in fp32 before
// their one rounding: ops.PackedViT), so a score is already the exponent.  The reference maximum then enters the score tiles as the
// MFMA's initial accumulator (-m in every register: S' = K Q^T - m) and a probability is exp2(S') -- no v_fma per score.
// Q8: the output leaves as MX-fp8 (e4m3 bytes + one E8M0 scale per 32 channels = per 32-row half of this head's O^T) instead of 16-bit
// rows -- what mvf_quant_mxfp8 would make of the 16-bit output, bit for bit (the values are rounded to bf16 first), without the
// [F*N, D] bf16 tensor's round trip through HBM in front of the fp8 proj GEMM (BASELINE configs[4]).
template <int NKT, int OCC, bool F16, bool LSE, bool MSUM, bool QS = false, bool Q8 = false>
__global__ __launch_bounds__(512, OCC) void vit_attn32p_kernel(Attn32Args g) {
  const AttnArgs& a = g.a;
  constexpr int KROWS = NKT * 32;
  constexpr int BLK = KROWS * 128;             // bytes of one K (or V) block image
  constexpr int NS = 3 * NKT;                  // tile slots of the ring
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3][K | V]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int unit = (slot / g.nchunk) * 8 + xcd, chunk = slot % g.nchunk;
  if (unit >= g.nunits) return;
  const int f = unit / a.H, h = unit % a.H;
  const size_t ld = (size_t)3 * a.D;
  const bf16_t* base = reinterpret_cast<const bf16_t*>(a.qkv) + (size_t)f * a.N * ld;
  const bf16_t* qb = base + h * HD;
  const bf16_t* kbp = base + a.D + h * HD;
  const bf16_t* vbp = base + 2 * a.D + h * HD;
  const int q0 = (chunk * nw + wave) * 32;
  const int nblk = (a.N + KROWS - 1) / KROWS;
  const int T = (a.N + 31) >> 5;               // 32-key tiles

  constexpr int NP = KROWS / 8;
  const int prow = lane >> 3, pc = lane & 7;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
  // piece p of a block = its rows 8p .. 8p+7; lane -> (row 8p + lane/8, physical 16-B chunk lane%8); the K swizzle of a row depends
  // on the piece only through its parity ((row >> 1) & 7 = 4 (p & 1) + (lane >> 4)), the V swizzle not at all
  const uint32_t ld2 = (uint32_t)ld * 2;      // row stride in bytes
  const uint32_t vrow = (uint32_t)prow * ld2;
  const uint32_t voK0 = vrow + ((pc ^ (prow >> 1)) << 4), voK1 = vrow + ((pc ^ (4 + (prow >> 1))) << 4);
  const uint32_t voV = vrow + ((pc ^ (((prow >> 1) & 1) << 2)) << 4);
  auto issue = [&](int b, int rb) {            // block b into ring buffer rb
    const uint32_t ldsk = lds0 + rb * 2 * BLK;
    if ((b + 1) * KROWS <= a.N) {              // a block of real rows: scalar base per piece, constant lane offsets
      for (int p = wave; p < NP; p += nw) {
        const size_t row0 = (size_t)(b * KROWS + p * 8) * ld;
        lds_dma16(kbp + row0, (p & 1) ? voK1 : voK0, ldsk + p * 1024);
        lds_dma16(vbp + row0, voV, ldsk + BLK + p * 1024);
      }
    } else {                                   // the last block: rows >= N repeat row N-1 (finite; masked / zero-weighted below)
      for (int p = wave; p < NP; p += nw) {
        const int lr = p * 8 + prow;
        const uint32_t rr = (uint32_t)min(b * KROWS + lr, a.N - 1) * ld2;
        lds_dma16(kbp, rr + ((pc ^ ((lr >> 1) & 7)) << 4), ldsk + p * 1024);
        lds_dma16(vbp, rr + ((pc ^ (((lr >> 1) & 1) << 2)) << 4), ldsk + BLK + p * 1024);
      }
    }
  };
  if (q0 >= a.N) {
    // a wave without query rows (the last workgroup of a (frame, head)): its share of the DMA issues and the barriers, nothing else
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nblk > 1) issue(1, 1);
    for (int b = 0; b + 1 < nblk; ++b) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (b + 2 < nblk) issue(b + 2, (b + 2) % 3);
    }
    return;
  }

  bf16x8_t qf[4];
  {
    const int qrow = min(q0 + r, a.N - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qb + (size_t)qrow * ld + ks * 16 + hh * 8);
  }
  const int koff0 = r * 128;
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = koff0 + (((2 * ks + hh) ^ ((r >> 1) & 7)) << 4);
  const int li = lane & 15, gi = (lane >> 4) & 1;
  int voff[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) voff[dt] = BLK + (4 * hh + (li >> 2)) * 128 + ((dt ^ ((li >> 3) & 1)) << 6) + gi * 32 + 8 * (li & 3);
  // byte offset of tile slot s (0 .. NS-1) inside the ring, for K (V: the same + BLK, folded into voff)
  auto slot_off = [](int s) { return (s / NKT) * 2 * BLK + (s % NKT) * 4096; };

  float m_use = -1e30f, m_lim = 0.f, nm = 0.f, l_part = 0.f;
  uint64_t trig = 0;                 // lanes whose row maximum outgrew the threshold (sticky)
  f32x16_t o[2], lacc, sA, sB;       // O^T, row sums (MSUM), S(i) / S(i+1): even ring slots keep S(i) in sA, odd ones in sB
  uint32_t pp[8];                    // P(i-1), rounded to the operand format: the two 16-key k-steps' B fragments
#pragma unroll
  for (int e = 0; e < 16; ++e) { o[0][e] = 0.f; o[1][e] = 0.f; lacc[e] = 0.f; }
#pragma unroll
  for (int e = 0; e < 8; ++e) pp[e] = 0;
  union { bf16x8_t v; uint32_t u[4]; } ones;
#ifndef MVF_ATTN_ONES_ROW0
#define MVF_ATTN_ONES_ROW0 1
#endif
  // the row-sum product's A operand: ones in ROW 0 only (lanes 0 and 32), zeros in the other 31 rows -- the sum lands in register 0 of the
  // lower lane half and 31 of the 32 result rows multiply by zero (what an all-ones fragment spends on 32 identical rows is toggling)
  ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = (MVF_ATTN_ONES_ROW0 && r != 0) ? 0u : (F16 ? 0x3C003C00u : 0x3F803F80u);

  f32x16_t negm;                     // QS: -m_use in every register, the score tiles' initial accumulator (zero until tile 0 is known)
#pragma unroll
  for (int e = 0; e < 16; ++e) negm[e] = 0.f;
  auto qk = [&](int soff) __attribute__((always_inline)) {
    f32x16_t s;
    if constexpr (QS) {
      s = negm;
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(smem + koff[ks] + soff);
      s = mfma32x32x16<F16>(kf, qf[ks], s);
    }
    return s;
  };
  // O^T += V^T(tile at ring offset soff) P^T, P = pp
  auto pv = [&](int soff) __attribute__((always_inline)) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      union { bf16x8_t v; uint32_t u[4]; } pf;
#pragma unroll
      for (int j = 0; j < 4; ++j) pf.u[j] = pp[4 * s2 + j];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        union { bf16x8_t v; bf16x4_t hq[2]; } vf;
        const char* p0 = smem + voff[dt] + soff + s2 * 2048;
        vf.hq[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(p0));
        vf.hq[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(p0 + 1024));
        o[dt] = mfma32x32x16<F16>(vf.v, pf.v, o[dt]);
      }
      if constexpr (MSUM) lacc = mfma32x32x16<F16>(ones.v, pf.v, lacc);
    }
  };

  // One tile step.  sc = S(i), sn receives S(i+1); so_prev / so_cur / so_next: ring byte offsets of tiles i-1, i, i+1 (compile-time
  // constants in the unrolled rounds -> LDS immediates; run-time values in the remainder loop -> one v_add per address register);
  // sync: the next tile opens a new block; last: no S(i+1), padded keys masked, P(i) V(i) right away.
  // A row maximum that outgrows the first tile's by more than the threshold only raises the sticky flag `trig`: the wave finishes
  // its walk (its results are then void, possibly not finite) and redoes its rows in careful() -- no exit from the fast loop, whose
  // control flow stays a plain chain of blocks (exits with live accumulators cost register copies on the common path).
  auto step = [&](f32x16_t& sc, f32x16_t& sn, int i, int so_prev, int so_cur, int so_next, auto sync_tag, auto last_tag, int rb_issue)
      __attribute__((always_inline)) {
    constexpr bool SYNC = decltype(sync_tag)::value, LAST = decltype(last_tag)::value;
    // ---- A ----
    pv(so_prev);
    if constexpr (LAST) {
      const int nkeys = a.N - i * 32;
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[e] = (e & 3) + 8 * (e >> 2) + 4 * hh < nkeys ? sc[e] : -1e30f;
    }
    float mx = max3f(sc[0], sc[1], sc[2]);   // v_max3 only: a two-input maximum of matrix results costs a canonicalising v_max per input
#pragma unroll
    for (int e = 3; e + 1 < 16; e += 2) mx = max3f(mx, sc[e], sc[e + 1]);
    mx = pair_max3(max3f(mx, sc[15], sc[15]));   // (every value of the lane before the swap: both lanes of a pair must agree)
    trig |= __builtin_amdgcn_ballot_w64(mx > m_lim);
    if constexpr (SYNC) {
      // the next tile opens block b+1: its pieces have landed (every wave waits for its own, then the workgroup meets); everybody is
      // past A of this tile, i.e. done with block b-1, whose buffer takes block b+2
      const int b = i / NKT;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (b + 2 < nblk) issue(b + 2, rb_issue);
    }
    // ---- B ----
    if constexpr (!LAST) sn = qk(so_next);
    float ls[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      const float p0 = __builtin_amdgcn_exp2f(QS ? sc[e] : fmaf(sc[e], a.scale_log2, nm));
      const float p1 = __builtin_amdgcn_exp2f(QS ? sc[e + 1] : fmaf(sc[e + 1], a.scale_log2, nm));
      pp[e >> 1] = pack16x2<F16>(p0, p1);
      if constexpr (!MSUM) { ls[e & 3] += p0; ls[(e + 1) & 3] += p1; }
    }
    if constexpr (!MSUM) l_part += (ls[0] + ls[1]) + (ls[2] + ls[3]);
    if constexpr (LAST) pv(so_cur);
  };

  // The careful walk: this wave's 32 query rows against all keys again, from scratch, with the textbook online softmax (reference
  // maximum and rescale of O and l EVERY tile) and K / V fragments fetched straight from global memory -- no LDS ring, no barrier,
  // no state taken over from the fast loop.  Entered when a row's maximum outgrows the first tile's by more than the threshold (2^30
  // in the exponent for bf16 probabilities, 2^15 for fp16): on real attention scores practically never, but the result must not
  // depend on that.  Slow (2-byte gathers for V^T) and self-contained by design: the fast loop's exits carry no register state.
  auto careful = [&]() __attribute__((always_inline)) {
    m_use = -1e30f; l_part = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o[0][e] = 0.f; o[1][e] = 0.f; lacc[e] = 0.f; }
    for (int t = 0; t < T; ++t) {
      f32x16_t s;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = 0.f;
      const int krow = min(t * 32 + r, a.N - 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kbp + (size_t)krow * ld + ks * 16 + hh * 8);
        s = mfma32x32x16<F16>(kf, qf[ks], s);
      }
      float mx = -1e30f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s[e] = t * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh < a.N ? s[e] : -1e30f;
        mx = fmaxf(mx, s[e]);
      }
      mx = pair_max(mx);
      const float csc = QS ? 1.0f : a.scale_log2;
      const float m_new = fmaxf(m_use, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_use - m_new) * csc);   // 0 on the first tile
      m_use = m_new;
      const float nmc = -m_new * csc;
      l_part *= alpha;
#pragma unroll
      for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; lacc[e] *= alpha; }
      uint32_t pq[8];
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(s[e], csc, nmc));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(s[e + 1], csc, nmc));
        pq[e >> 1] = pack16x2<F16>(p0, p1);
        l_part += p0 + p1;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        union { bf16x8_t v; uint32_t u[4]; } pf;
#pragma unroll
        for (int j = 0; j < 4; ++j) pf.u[j] = pq[4 * s2 + j];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          union { bf16x8_t v; bf16_t e[8]; } vf;   // V^T fragment: element j <-> key 32 t + 16 s2 + 8 (j >> 2) + 4 hh + (j & 3), d = 32 dt + r
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int key = min(t * 32 + 16 * s2 + 8 * (j >> 2) + 4 * hh + (j & 3), a.N - 1);
            vf.e[j] = vbp[(size_t)key * ld + dt * 32 + r];
          }
          o[dt] = mfma32x32x16<F16>(vf.v, pf.v, o[dt]);
        }
        if constexpr (MSUM) lacc = mfma32x32x16<F16>(ones.v, pf.v, lacc);
      }
    }
  };

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the Q fragments too)
  __builtin_amdgcn_s_barrier();
  if (nblk > 1) issue(1, 1);
  sA = qk(slot_off(0));
  {
    // the first tile's row maximum is the reference of every exponent of the walk (O and l start at zero: nothing to rescale)
    // (padded keys of a short tile repeat key N-1: their scores equal a real one's, the maximum needs no mask)
    float mx = max3f(sA[0], sA[1], sA[2]);
#pragma unroll
    for (int e = 3; e + 1 < 16; e += 2) mx = max3f(mx, sA[e], sA[e + 1]);
    mx = pair_max3(max3f(mx, sA[15], sA[15]));
    m_use = mx;
    if constexpr (QS) {
      // from here on every score tile is born relative to the reference (tile 0 itself: one subtraction per score, once)
      m_lim = g.thr;
#pragma unroll
      for (int e = 0; e < 16; ++e) { negm[e] = -mx; sA[e] -= mx; }
    } else {
      m_lim = mx + g.thr;
      nm = -mx * a.scale_log2;
    }
  }
  static_assert(NS % 2 == 0, "the score tiles alternate between two register blocks with the tile's parity");
  int i0 = 0;
  for (; i0 + NS < T; i0 += NS) {      // whole rounds of the ring, none of them the last tile: every LDS offset an immediate
    // tile 0 has no predecessor: P = 0 meets tile 0's own (landed, finite) V rows instead of the ring's unwritten last slot
    const int so_first_prev = i0 == 0 ? 0 : slot_off(NS - 1);
    static_for<NS>([&](auto s_tag) __attribute__((always_inline)) {
      constexpr int S = decltype(s_tag)::value;
      constexpr bool SYNC = S % NKT == NKT - 1;
      step((S & 1) ? sB : sA, (S & 1) ? sA : sB, i0 + S, S == 0 ? so_first_prev : slot_off(S - 1), slot_off(S), slot_off((S + 1) % NS),
           std::integral_constant<bool, SYNC>{}, std::false_type{}, (S / NKT + 2) % 3);
    });
  }
  // the remaining 1 .. NS tiles: ring offsets at run time; i0 is even, so tile i keeps its parity's register block
  auto rt_off = [&](int i) { return ((i / NKT) % 3) * 2 * BLK + (i % NKT) * 4096; };
  auto rt_step = [&](f32x16_t& sc, f32x16_t& sn, int i) __attribute__((always_inline)) {
    const int sp = i == 0 ? 0 : rt_off(i - 1);
    if (i % NKT == NKT - 1) step(sc, sn, i, sp, rt_off(i), rt_off(i + 1), std::true_type{}, std::false_type{}, (i / NKT + 2) % 3);
    else step(sc, sn, i, sp, rt_off(i), rt_off(i + 1), std::false_type{}, std::false_type{}, 0);
  };
  int i = i0;
  for (; i + 2 < T; i += 2) {
    rt_step(sA, sB, i);
    rt_step(sB, sA, i + 1);
  }
  if (i + 1 < T) {                     // two tiles left: a middle one, then the last one in sB
    rt_step(sA, sB, i);
    ++i;
#pragma unroll
    for (int e = 0; e < 16; ++e) sA[e] = sB[e];
  }
  step(sA, sB, i, i == 0 ? 0 : rt_off(i - 1), rt_off(i), 0, std::false_type{}, std::true_type{}, 0);
  if (trig != 0) careful();
  // ---- epilogue: O / l -> 16-bit, whole 128-byte rows ----
  const float l_run = MSUM ? (MVF_ATTN_ONES_ROW0 ? pair_sum(lacc[0]) : lacc[0]) : pair_sum(l_part);   // (row 0 = register 0 of the lane half hh = 0; the other half holds 0)
  const int q = q0 + r;
  const float inv = 1.0f / l_run;
  if constexpr (LSE) {
    if (q < a.N && hh == 0)   // p = exp2(s * scale_log2 - lse) reproduces the normalised probability
      a.lse[((size_t)f * a.H + h) * a.npad + q] = fmaf(m_use, QS ? 1.0f : a.scale_log2, __builtin_amdgcn_logf(l_run));
  }
  if constexpr (Q8) {
    // The 32 channels of d-tile dt (one MX block) are this lane's 16 registers and its partner's: block maximum = in-lane chain + one
    // swap.  Values are rounded to bf16 FIRST (what the 16-bit form stores and mvf_quant_mxfp8 reads), then scaled and converted.
    static_assert(!F16 && !LSE, "MX-fp8 output: bf16 rounding, frozen path");
    unsigned char* qrow = reinterpret_cast<unsigned char*>(a.out) + ((size_t)f * a.N + min(q, a.N - 1)) * a.D + h * HD;
    unsigned sb2 = 0;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      float v[16];
      float amax = 0.f;
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const uint32_t w = pack16x2<false>(o[dt][e] * inv, o[dt][e + 1] * inv);
        v[e] = __uint_as_float(w << 16);
        v[e + 1] = __uint_as_float(w & 0xffff0000u);
        amax = max3f(amax, fabsf(v[e]), fabsf(v[e + 1]));
      }
      amax = pair_max(amax);
      const unsigned sb = mx_scale_byte(amax);
      const float is = mx_inv_scale(sb);
      sb2 |= sb << (8 * dt);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const uint32_t A = pack_fp8x4(v[8 * k + 0] * is, v[8 * k + 1] * is, v[8 * k + 2] * is, v[8 * k + 3] * is);
        const uint32_t B = pack_fp8x4(v[8 * k + 4] * is, v[8 * k + 5] * is, v[8 * k + 6] * is, v[8 * k + 7] * is);
        const auto rr = __builtin_amdgcn_permlane32_swap(A, B, false, false);
        if (q < a.N) *reinterpret_cast<uint2*>(qrow + dt * 32 + k * 16 + hh * 8) = make_uint2(rr[0], rr[1]);
      }
    }
    // the head's two block scales: bytes 2 (h & 1), 2 (h & 1) + 1 of the row's dword of K tile h / 2
    if (q < a.N && hh == 0)
      *reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(g.q8_scales + (size_t)(h >> 1) * g.rows + (size_t)f * a.N + q) + (h & 1) * 2) =
          (unsigned short)sb2;
    return;
  }
  bf16_t* orow = reinterpret_cast<bf16_t*>(a.out) + ((size_t)f * a.N + min(q, a.N - 1)) * a.D + h * HD;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      uint32_t a0 = pack16x2<F16>(o[dt][8 * k + 0] * inv, o[dt][8 * k + 1] * inv), a1 = pack16x2<F16>(o[dt][8 * k + 2] * inv, o[dt][8 * k + 3] * inv);
      uint32_t b0 = pack16x2<F16>(o[dt][8 * k + 4] * inv, o[dt][8 * k + 5] * inv), b1 = pack16x2<F16>(o[dt][8 * k + 6] * inv, o[dt][8 * k + 7] * inv);
      const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      if (q < a.N)
        *reinterpret_cast<uint4*>(orow + dt * 32 + k * 16 + hh * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
}

template <int NKT, int OCC, bool F16, bool LSE, bool MSUM, bool QS = false, bool Q8 = false>
int launch32p(const Attn32Args& g, int nw, hipStream_t st) {
  constexpr size_t LDS = (size_t)6 * NKT * 32 * 128;
  static uint64_t done = 0;
  const void* fn = reinterpret_cast<const void*>(vit_attn32p_kernel<NKT, OCC, F16, LSE, MSUM, QS, Q8>);
  if (LDS > 48 * 1024) {
    const int rc = mvf_ensure_lds(fn, LDS, done);
    if (rc != MVF_OK) return rc;
  }
  const int grid = ceil_div(g.nunits, 8) * 8 * g.nchunk;
  hipLaunchKernelGGL((vit_attn32p_kernel<NKT, OCC, F16, LSE, MSUM, QS, Q8>), dim3(grid), dim3(nw * 64), LDS, st, g);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

template <int NKT, int OCC, bool F16, bool LSE>
int launch32(const Attn32Args& g, int nw, hipStream_t st) {
  constexpr size_t LDS = (size_t)4 * NKT * 32 * 128;
  static uint64_t done = 0;
  const void* fn = reinterpret_cast<const void*>(vit_attn32_kernel<NKT, OCC, F16, LSE>);
  if (LDS > 48 * 1024) {
    const int rc = mvf_ensure_lds(fn, LDS, done);
    if (rc != MVF_OK) return rc;
  }
  const int grid = ceil_div(g.nunits, 8) * 8 * g.nchunk;
  hipLaunchKernelGGL((vit_attn32_kernel<NKT, OCC, F16, LSE>), dim3(grid), dim3(nw * 64), LDS, st, g);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

}  // namespace

// form: 5 = the product path (pipelined walk, 64-key blocks, row sums on the matrix pipe); 4 = the same with VALU row sums; 6 / 7 = 128-key
// blocks (8 waves); 0 .. 3 = the unpipelined walk (64 / 96 / 128-key blocks, 64-key at 3 waves per SIMD), kept for A/B runs and as an
// independent implementation the tests compare with.  nw_force > 0: waves per workgroup
int mvf_vit_attn32_impl(int dtype, const void* qkv, void* out, float* lse, int F, int N, int H, int D, int form, int nw_force,
                        hipStream_t st, unsigned* q8_scales) {
  MVF_CHECK_ARG(qkv && out && F > 0 && N > 0 && H > 0 && D == H * vit_attn::HD);
  MVF_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0);
  MVF_CHECK_ARG(dtype == MVF_BF16 || dtype == MVF_F16);
  MVF_CHECK_ARG(nw_force >= 0 && nw_force <= 8 && (lse == nullptr || form < 16));
  Attn32Args g;
  g.a.qkv = (const char*)qkv; g.a.out = (char*)out; g.a.N = N; g.a.H = H; g.a.D = D;
  g.a.nblk = 0; g.a.rounds = 0;
  g.a.scale_log2 = vit_attn::LOG2E / 8.0f;  // 64^-0.5 * log2(e)
  g.a.lse = lse;
  g.a.npad = ceil_div(N, 16) * 16;
  // deferred-maximum threshold: probabilities reach at most 2^30 (bf16: fp32's exponent range) / 2^15 (fp16: max 65504)
  g.thr = (dtype == MVF_F16 ? 15.0f : 30.0f) / g.a.scale_log2;
  const int nqb = ceil_div(N, 32);
  const int nw = nw_force > 0 ? nw_force : (nqb < 4 ? nqb : (form == 6 || form == 7) ? 8 : 4);   // whole waves-per-SIMD rounds: 5 .. 7 waves leave SIMDs uneven (measured slower)
  g.nchunk = ceil_div(nqb, nw);
  g.nunits = F * H;
  const bool f16 = dtype == MVF_F16;
  g.q8_scales = q8_scales;
  g.rows = (size_t)F * N;
  if (q8_scales != nullptr) {   // MX-fp8 output (out: [F*N, D] bytes): the product form only, bf16 arithmetic, head pairs share a scale dword
    MVF_CHECK_ARG(!f16 && lse == nullptr && H % 2 == 0 && ((uintptr_t)out % 8) == 0 && ((uintptr_t)q8_scales % 4) == 0);
    if (form == 5 + 16) {   // pre-scaled q (below)
      g.thr = 30.0f;
      return launch32p<2, 2, false, false, true, true, true>(g, nw, st);
    }
    return launch32p<2, 2, false, false, true, false, true>(g, nw, st);
  }
  if (lse) {
    if (f16) return MVF_ERR_ARG;
    return launch32p<2, 2, false, true, true>(g, nw, st);
  }
  if (form == 5 + 16) {   // the product path of a backbone whose q columns are pre-scaled by log2(e) / 8: scores are exponents
    g.thr = dtype == MVF_F16 ? 15.0f : 30.0f;
    return f16 ? launch32p<2, 3, true, false, true, true>(g, nw, st) : launch32p<2, 3, false, false, true, true>(g, nw, st);   // (three waves per SIMD asked for: left alone hipcc takes 172 registers)
  }
  switch (form) {
    case 1: return f16 ? launch32<3, 2, true, false>(g, nw, st) : launch32<3, 2, false, false>(g, nw, st);
    case 2: return f16 ? launch32<4, 2, true, false>(g, nw, st) : launch32<4, 2, false, false>(g, nw, st);
    case 3: return f16 ? launch32<2, 3, true, false>(g, nw, st) : launch32<2, 3, false, false>(g, nw, st);
    case 4: return f16 ? launch32p<2, 2, true, false, false>(g, nw, st) : launch32p<2, 2, false, false, false>(g, nw, st);
    case 5: return f16 ? launch32p<2, 2, true, false, true>(g, nw, st) : launch32p<2, 2, false, false, true>(g, nw, st);
    case 6: return f16 ? launch32p<4, 2, true, false, false>(g, nw, st) : launch32p<4, 2, false, false, false>(g, nw, st);
    case 7: return f16 ? launch32p<4, 2, true, false, true>(g, nw, st) : launch32p<4, 2, false, false, true>(g, nw, st);
    default: return f16 ? launch32<2, 2, true, false>(g, nw, st) : launch32<2, 2, false, false>(g, nw, st);
  }
}
