// Fused multi-head self-attention of the ViT blocks (no mask, head_dim 64):
//   out[f, n, h*64:(h+1)*64] = softmax(q k^T / sqrt(64)) v       q,k,v = column slices of qkv[f*N+n, 3*D]
// Stands in for timm Attention.forward (reached from CARL_MVF/models/transformer.py:188).  The [N,N] score
// matrix never leaves registers.
//
// gfx950 design:
//  * one workgroup (4 waves) per (frame, head) [x query chunk when N > 224]; K and V of a 224-key block are
//    staged ONCE in LDS (K XOR-swizzled for conflict-free ds_read_b128 MFMA fragments)
//  * scores are computed TRANSPOSED, S^T = K Q^T, so a lane owns one query column: the softmax row
//    reduction is in-lane + two cross-lane steps (xor 16, 32), and the S^T accumulator registers are
//    directly the B operand of the next MFMA (O^T = V^T P^T) -- no LDS round trip for P
//  * bf16: v_mfma_f32_16x16x32_bf16, V fragments fetched with the ds_read_b64_tr_b16 transposing read
//    (or 2-byte gathers, selectable, used to cross-check the transposing read on hardware)
//  * f32 (parity mode): v_mfma_f32_16x16x4_f32, exact fp32
//  * online softmax across key blocks (only one block for N = 197)
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "mvf_hip_internal.h"
#include "vit_attn_tiles.h"

namespace {
using namespace vit_attn;

// ------------------------------------------------------------------------------------------------
// bf16
// ------------------------------------------------------------------------------------------------
// NT = key tiles (of 16) that are computed per key block.  NT == KT is the general form (any N, every tile masked against
// the valid key count).  NT < KT is the specialisation for ONE key block whose last NT-th tile is the only padded one
// (ViT-B/16 at 224 px: N = 197 -> NT = 13): all-padding tiles are not computed at all and only tile NT-1 is masked.
// Compile-time loop bounds matter here: per-tile run-time branches serialise read -> MFMA -> max chains.
template <bool TR_READ, int NT>
__global__ __launch_bounds__(256, NT < KT ? 3 : 2) void vit_attn_bf16_kernel(AttnArgs a) {
  // only the NT*16 keys that are computed are staged: at NT = 13 the block needs 52 KiB, so THREE workgroups (12 waves)
  // share a CU instead of two -- the per-wave chain read -> MFMA -> softmax -> MFMA is serial, more waves hide it
  constexpr int KROWS = NT * 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * KROWS * 128];
  char* sk = smem;                // K block: [keys][64] bf16, 128-B rows, 16-B chunks XOR-swizzled by (row & 7)
  char* sv = smem + KROWS * 128;  // V block: [keys][64] bf16, 128-B rows, 32-B chunks XOR-swizzled by ((row >> 1) & 3):
                               // the transposing reads of one 32-lane half touch rows r, r+2, r+4, r+6 at the same
                               // column -- 128-B rows put those on one set of banks (4-way conflict, measured 56 % of
                               // all LDS cycles) unless the chunk index is rotated with the row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int f = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const size_t ld = (size_t)3 * a.D;                       // qkv row stride (elements)
  const bf16_t* base = reinterpret_cast<const bf16_t*>(a.qkv) + (size_t)f * a.N * ld;
  const bf16_t* qb = base + h * HD;
  const bf16_t* kbp = base + a.D + h * HD;
  const bf16_t* vbp = base + 2 * a.D + h * HD;
  // row inside a 32-key step that this lane addresses in the transposing V read: 4g + (li >> 2) (+16)
  const int vsw = ((2 * g + (li >> 3)) & 3) << 5;          // its chunk rotation, in bytes (same for row + 16)

  for (int rd = 0; rd < a.rounds; ++rd) {
    const int qt = (blockIdx.y * a.rounds + rd) * 4 + wave;  // query tile of this wave
    // rounds > 1 only with a single key block, staged (with its barriers) in round 0: a wave without a query tile in
    // a later round is done (N = 197: 13 tiles on 4 waves, three of them would compute a discarded 4th tile)
    if (rd > 0 && qt * 16 >= a.N) break;
    const int qrow = min(qt * 16 + li, a.N - 1);
    // Q fragments (MFMA B operand): lane (query li, k-group g) holds Q[q][ks*32 + 8g .. +8]
    bf16x8_t qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[ks] = *reinterpret_cast<const bf16x8_t*>(qb + (size_t)qrow * ld + ks * 32 + g * 8);

    float m_run = -1e30f, l_run = 0.f;   // m_run in raw-score units (scores are scaled inside the exp2 argument)
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    for (int kb = 0; kb < a.nblk; ++kb) {
      const int nkeys = min(a.N - kb * KB, KB);            // valid keys of this block
      if constexpr (NT < KT) {
        // ---- one key block, staged once per workgroup by LDS-DMA (no VGPR round trip): K pieces first, then V; the
        // S^T MFMAs and the softmax only need K, so V's HBM latency hides under them (ablation: the synchronous
        // load -> ds_write staging cost 35 % of the kernel).  A piece = 8 rows x 128 B = one wave-instruction; lane ->
        // (row 8p + lane/8, physical chunk lane%8); both swizzles are applied on the SOURCE chunk index.
        if (rd == 0) {
          constexpr int NP = KROWS / 8;
          asm volatile("" : "+v"(qf[0]), "+v"(qf[1]));   // Q loads are waited for here, before any DMA is in flight
          const int prow = lane >> 3, pc = lane & 7;
          for (int p = wave; p < NP; p += 4) {
            const int r = p * 8 + prow;
            const bf16_t* src = kbp + (size_t)min(r, a.N - 1) * ld + ((pc ^ (r & 7)) << 3);
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sk + p * 1024), 16, 0, 0);
          }
          for (int p = wave; p < NP; p += 4) {
            const int r = p * 8 + prow;    // rows >= N repeat row N-1: finite values, their probabilities are 0
            const bf16_t* src = vbp + (size_t)min(r, a.N - 1) * ld + ((pc ^ (((r >> 1) & 3) << 1)) << 3);
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sv + p * 1024), 16, 0, 0);
          }
          // K landed <=> all but this wave's V pieces retired (waves < NP % 4 issued one piece more)
          if (wave < (NP & 3)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NP + 3) / 4) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP / 4) : "memory");
          __builtin_amdgcn_s_barrier();
        }
      } else if (a.nblk > 1 || rd == 0) {
        if (kb > 0 || rd > 0) __syncthreads();  // everyone done with the previous block
        // ---- stage K, V block: thread -> (row = tid/8 + 32*i, chunk = tid%8) ----
#pragma unroll
        for (int i = 0; i < (KROWS + 31) / 32; ++i) {
          const int r = (tid >> 3) + 32 * i, c = tid & 7;
          const int key = kb * KB + r;
          uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
          if (key < a.N) {
            kv = *reinterpret_cast<const uint4*>(kbp + (size_t)key * ld + c * 8);
            vv = *reinterpret_cast<const uint4*>(vbp + (size_t)key * ld + c * 8);
          }
          if (KROWS % 32 == 0 || r < KROWS) {
            *reinterpret_cast<uint4*>(sk + r * 128 + ((c ^ (r & 7)) << 4)) = kv;
            *reinterpret_cast<uint4*>(sv + r * 128 + ((c ^ (((r >> 1) & 3) << 1)) << 4)) = vv;
          }
        }
        __syncthreads();
      }
      // ---- S^T tiles: s[kt] = K_tile(kt) . Q^T  -> lane holds S[query li][key kt*16 + 4g + r] ----
      f32x4_t s[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        s[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (kt < NT) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int row = kt * 16 + li;
            const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sk + row * 128 + (((ks * 4 + g) ^ (li & 7)) << 4));
            s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kt], 0, 0, 0);
          }
        }
      }
      float mx = -1e30f;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (NT == KT || kt == NT - 1) {   // padded keys -> -inf (general form: every tile; specialised: the last one)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kt][r] = kt * 16 + 4 * g + r < nkeys ? s[kt][r] : -1e30f;
        }
        mx = fmaxf(fmaxf(mx, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
      }
      // ---- online softmax: p = exp2(scale_log2 * (s - m)), the scale folded into one fma per score ----
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float nm = -m_new * a.scale_log2;
      float ls = 0.f;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], a.scale_log2, nm));
          s[kt][r] = p;
          ls += p;
        }
      }
      ls += __shfl_xor(ls, 16, 64);
      ls += __shfl_xor(ls, 32, 64);
      if (kb > 0) {            // a single key block (N <= 224, every ViT-B/16 config) never rescales
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * a.scale_log2);
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
      }
      l_run += ls;
      m_run = m_new;
      if constexpr (NT < KT) {
        if (rd == 0) {   // V landed (every wave waits for its own pieces, then the workgroup meets)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
      }
      // ---- O^T += V^T P^T : k-step = 32 keys = score tiles (2s, 2s+1) ----
#pragma unroll
      for (int st = 0; st < (NT + 1) / 2; ++st) {
        {
          union { bf16x8_t v; uint32_t u[4]; } pf;
          pf.u[0] = pack_bf16x2(s[2 * st][0], s[2 * st][1]);
          pf.u[1] = pack_bf16x2(s[2 * st][2], s[2 * st][3]);
          pf.u[2] = pack_bf16x2(s[2 * st + 1][0], s[2 * st + 1][1]);
          pf.u[3] = pack_bf16x2(s[2 * st + 1][2], s[2 * st + 1][3]);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            union { bf16x8_t v; bf16x4_t h[2]; bf16_t e[8]; } vf;
            if constexpr (TR_READ) {
              // group of 16 lanes reads a 4-key x 16-d block transposed: lane supplies &V[key0+4g+(li>>2)][d0+4*(li&3)],
              // receives V[key0+4g+0..3][d0+li]
              const char* p0 = sv + (st * 32 + 4 * g + (li >> 2)) * 128 + (((dt * 32) ^ vsw) + 8 * (li & 3));
              vf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  (__attribute__((address_space(3))) bf16x4_t*)(p0));
              if (2 * st + 1 < NT)
                vf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) bf16x4_t*)(p0 + 16 * 128));
              else
                vf.h[1] = (bf16x4_t){0, 0, 0, 0};   // keys beyond the staged block: their probabilities are 0 too
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const int key = st * 32 + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
                vf.e[j] = key < KROWS ? *reinterpret_cast<const bf16_t*>(sv + key * 128 + (((dt * 32) ^ (((key >> 1) & 3) << 5)) + li * 2))
                                      : (bf16_t)0;
              }
            }
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, o[dt], 0, 0, 0);
          }
        }
      }
    }
    // ---- write O[query li][dt*16 + 4g + r] / l ----
    const int q = qt * 16 + li;
    if (q < a.N) {
      const float inv = 1.0f / l_run;
      bf16_t* orow = reinterpret_cast<bf16_t*>(a.out) + ((size_t)f * a.N + q) * a.D + h * HD;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<uint2*>(orow + dt * 16 + 4 * g) =
            make_uint2(pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv));
    }
  }
}

template <int NT, int OCC, bool F16 = false, bool MSUM = false>
__global__ __launch_bounds__(256, OCC) void vit_attn_bf16_pair_kernel(AttnArgs a) {
  constexpr int KROWS = NT * 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * KROWS * 128];
  char* sk = smem;                // layouts and swizzles as in vit_attn_bf16_kernel
  char* sv = smem + KROWS * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int f = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const size_t ld = (size_t)3 * a.D;
  const bf16_t* base = reinterpret_cast<const bf16_t*>(a.qkv) + (size_t)f * a.N * ld;
  const bf16_t* qb = base + h * HD;
  const bf16_t* kbp = base + a.D + h * HD;
  const bf16_t* vbp = base + 2 * a.D + h * HD;
  bf16_t* obase = reinterpret_cast<bf16_t*>(a.out) + (size_t)f * a.N * a.D + h * HD;
  const int vsw = ((2 * g + (li >> 3)) & 3) << 5;
  const int qtiles = (a.N + 15) >> 4;
  const int cnt = wave < qtiles ? (qtiles - wave + 3) >> 2 : 0;   // query tiles wave, wave+4, ... of this wave

  auto load_q = [&](bf16x8_t (&qf)[2][2], const int (&qt)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int qrow = min(qt[i] * 16 + li, a.N - 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        qf[i][ks] = *reinterpret_cast<const bf16x8_t*>(qb + (size_t)qrow * ld + ks * 32 + g * 8);
    }
  };
  bf16x8_t qf[2][2];
  int qt[2] = {wave, wave + 4};
  load_q(qf, qt);
  // waited for before any DMA is in flight.  (Round 3: as untracked inline-asm loads that the counted K wait covers -- the Q
  // round trip then runs under the DMAs' -- 76.4 us against 77-79: the CU's other two workgroups already hide it.  Without any
  // fence hipcc drains the whole queue, V included, with vmcnt(0) in front of the first MFMA: the DMA loops have run-time trips.)
  asm volatile("" : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[1][0]), "+v"(qf[1][1]));
  {
    constexpr int NP = KROWS / 8;
    const int prow = lane >> 3, pc = lane & 7;
    for (int p = wave; p < NP; p += 4) {
      const int r = p * 8 + prow;
      const bf16_t* src = kbp + (size_t)min(r, a.N - 1) * ld + ((pc ^ (r & 7)) << 3);
      __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sk + p * 1024), 16, 0, 0);
    }
    for (int p = wave; p < NP; p += 4) {
      const int r = p * 8 + prow;    // rows >= N repeat row N-1: finite values, their probabilities are 0
      const bf16_t* src = vbp + (size_t)min(r, a.N - 1) * ld + ((pc ^ (((r >> 1) & 3) << 1)) << 3);
      __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sv + p * 1024), 16, 0, 0);
    }
    // K landed <=> all but this wave's V pieces retired (waves < NP % 4 issued one piece more)
    if (wave < (NP & 3)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NP + 3) / 4) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP / 4) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (cnt >= 2) {
    attn_tiles<NT, 2, F16, MSUM>(a, sk, sv, obase, qf, qt, li, g, vsw, true);
  } else if (cnt == 1) {
    attn_tiles<NT, 1, F16, MSUM>(a, sk, sv, obase, qf, qt, li, g, vsw, true);
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    return;
  }
  if (cnt > 2) {
    qt[0] = wave + 8; qt[1] = wave + 12;
    load_q(qf, qt);
    if (cnt >= 4) attn_tiles<NT, 2, F16, MSUM>(a, sk, sv, obase, qf, qt, li, g, vsw, false);
    else attn_tiles<NT, 1, F16, MSUM>(a, sk, sv, obase, qf, qt, li, g, vsw, false);
  }
}

// (A persistent form -- 3 x 256 workgroups walking the (frame, head) units with one K / V image pair, the next unit's K pieces
// issued as soon as every wave is past the current unit's last S phase -- was built in round 2, was bit-identical and measured
// SLOWER: 105.9 us against 78.2 us (two more workgroup barriers per unit, 24 spilled registers at the 168-VGPR budget; what it
// saves is what the CU's other two workgroups already hide).  Removed in round 3 together with the 2-waves-per-SIMD form (114 us).)

// ------------------------------------------------------------------------------------------------
// bf16, any N: key blocks of KT*16 keys streamed through a double-buffered LDS-DMA pipeline, two query tiles per wave
// ------------------------------------------------------------------------------------------------
// The long-sequence form (ViT-B/8 of every shipped configs_mvf/*.yml: N = 785; DINOv2 at 224/336 px: 257 / 577).
// A workgroup (4 waves) owns 8 query tiles = 128 queries of one (frame, head) and walks all key blocks:
//     wait for block b (vmcnt(0)) -> barrier -> issue block b+1 into the other buffer -> S^T, online softmax, O^T on block b
// so block b+1's HBM/L2 latency hides under block b's arithmetic and there is ONE barrier per block (it also guarantees
// every wave is done with the buffer the new DMAs overwrite).  Same LDS images / swizzles / accumulator-as-B-operand
// chaining as the single-block kernels; online softmax state (m, l) per query tile; keys >= N are masked in the last
// block only.  The earlier general kernel staged 224 keys synchronously through VGPRs for 64 queries per workgroup
// (280 TFLOP/s at N = 785, 160 at N = 257); this one keeps 2 x KT*16 keys in flight per workgroup.  Measured, F = 80,
// N = 785: KT = 6 (96 keys, 168 VGPRs, 3 waves/SIMD) 295 us = 514 TFLOP/s; KT = 4 (4 waves/SIMD) 345; KT = 2 344; KT = 8
// (2 waves/SIMD) 523; 8 waves per workgroup with KT = 4: 365.  N = 257 (F = 256): 149 / 166 / 192 / 355 / 240 us.
// Round 5: the last key block walks only the key tiles that hold keys (N = 16 k + 1 with the CLS token leaves ONE key in the last
// 96-key block at N = 577, 17 at N = 785): 652.8 -> 613.0 us (F = 256, N = 577), 313.8 -> 290.7 us (F = 80, N = 785), bit-identical.
template <int KT, int OCC, int NW = 4, bool F16 = false>
__global__ __launch_bounds__(NW * 64, OCC) void vit_attn_bf16_flash_kernel(AttnArgs a) {
  constexpr int KROWS = KT * 16;               // keys per block
  constexpr int BLK = KROWS * 128;             // bytes of one K (or V) block
  __shared__ __attribute__((aligned(16))) char smem[4 * BLK];   // [buf][K | V]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int f = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const size_t ld = (size_t)3 * a.D;
  const bf16_t* base = reinterpret_cast<const bf16_t*>(a.qkv) + (size_t)f * a.N * ld;
  const bf16_t* qb = base + h * HD;
  const bf16_t* kbp = base + a.D + h * HD;
  const bf16_t* vbp = base + 2 * a.D + h * HD;
  bf16_t* obase = reinterpret_cast<bf16_t*>(a.out) + (size_t)f * a.N * a.D + h * HD;
  const int vsw = ((2 * g + (li >> 3)) & 3) << 5;
  const int nblk = (a.N + KROWS - 1) / KROWS;
  const int qt0 = (blockIdx.y * NW + wave) * 2;                 // this wave's two query tiles: qt0, qt0 + 1
  const bool active = qt0 * 16 < a.N;                            // (an inactive wave still stages and meets the barriers)

  bf16x8_t qf[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int qrow = min((qt0 + i) * 16 + li, a.N - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[i][ks] = *reinterpret_cast<const bf16x8_t*>(qb + (size_t)qrow * ld + ks * 32 + g * 8);
  }
  asm volatile("" : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[1][0]), "+v"(qf[1][1]));   // landed before any DMA is in flight

  constexpr int NP = KROWS / 8;                 // 1-KiB pieces per K (or V) block; NP % 4 == 0 for KT even
  const int prow = lane >> 3, pc = lane & 7;
  auto issue = [&](int b) {
    char* sk = smem + (b & 1) * 2 * BLK;
    char* sv = sk + BLK;
    for (int p = wave; p < NP; p += NW) {
      const int r = b * KROWS + p * 8 + prow;   // rows >= N repeat row N-1 (finite; masked / zero-weighted below)
      const int rr = min(r, a.N - 1), lr = p * 8 + prow;
      __builtin_amdgcn_global_load_lds(GLB_PTR(kbp + (size_t)rr * ld + ((pc ^ (lr & 7)) << 3)), LDS_PTR(sk + p * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLB_PTR(vbp + (size_t)rr * ld + ((pc ^ (((lr >> 1) & 3) << 1)) << 3)),
                                       LDS_PTR(sv + p * 1024), 16, 0, 0);
    }
  };

  float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
  f32x4_t o[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[i][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // one key block: KTB (<= KT, even) key tiles of it take part -- full blocks run KT, the last block only the tiles that hold keys
  // (N = 16 k + 1 with the CLS token: one key in a 96-key block at N = 577, 17 at N = 785).  Skipped tiles would contribute exact zeros
  // (masked scores -> weight 0), so the result is bit-identical to walking all KT.
  auto block = [&](auto ktb_tag, const char* sk, const char* sv, int nkeys) __attribute__((always_inline)) {
    constexpr int KTB = decltype(ktb_tag)::value;
    f32x4_t s[2][KTB];
#pragma unroll
    for (int kt = 0; kt < KTB; ++kt) {
#pragma unroll
      for (int i = 0; i < 2; ++i) s[i][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sk + (kt * 16 + li) * 128 + (((ks * 4 + g) ^ (li & 7)) << 4));
#pragma unroll
        for (int i = 0; i < 2; ++i) s[i][kt] = mfma16x16x32<F16>(kf, qf[i][ks], s[i][kt]);
      }
    }
    if (nkeys < KTB * 16) {                            // last block: keys beyond N never win the max and get weight 0
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kt = 0; kt < KTB; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[i][kt][r] = kt * 16 + 4 * g + r < nkeys ? s[i][kt][r] : -1e30f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float mx = -1e30f;     // 3-input maxima and packed fp32 arithmetic: see attn_tiles
#pragma unroll
      for (int kt = 0; kt < KTB; ++kt) mx = max3f(max3f(mx, s[i][kt][0], s[i][kt][1]), s[i][kt][2], s[i][kt][3]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[i], mx);
      const float nm = -m_new * a.scale_log2;
      const f32x2_t sc2 = {a.scale_log2, a.scale_log2}, nm2 = {nm, nm};
      f32x2_t ls2 = {0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < KTB; ++kt) {
        const f32x2_t e0 = pk_fma((f32x2_t){s[i][kt][0], s[i][kt][1]}, sc2, nm2);
        const f32x2_t e1 = pk_fma((f32x2_t){s[i][kt][2], s[i][kt][3]}, sc2, nm2);
        const f32x2_t p0 = {__builtin_amdgcn_exp2f(e0[0]), __builtin_amdgcn_exp2f(e0[1])};
        const f32x2_t p1 = {__builtin_amdgcn_exp2f(e1[0]), __builtin_amdgcn_exp2f(e1[1])};
        s[i][kt][0] = p0[0]; s[i][kt][1] = p0[1]; s[i][kt][2] = p1[0]; s[i][kt][3] = p1[1];
        ls2 = pk_add(ls2, pk_add(p0, p1));
      }
      float ls = ls2[0] + ls2[1];
      ls += __shfl_xor(ls, 16, 64);
      ls += __shfl_xor(ls, 32, 64);
      const float alpha = __builtin_amdgcn_exp2f((m_run[i] - m_new) * a.scale_log2);   // 0 on the first block
      l_run[i] = l_run[i] * alpha + ls;
      m_run[i] = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[i][dt][r] *= alpha;
    }
#pragma unroll
    for (int st = 0; st < KTB / 2; ++st) {
      union { bf16x8_t v; uint32_t u[4]; } pf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        pf[i].u[0] = pack16x2<F16>(s[i][2 * st][0], s[i][2 * st][1]);
        pf[i].u[1] = pack16x2<F16>(s[i][2 * st][2], s[i][2 * st][3]);
        pf[i].u[2] = pack16x2<F16>(s[i][2 * st + 1][0], s[i][2 * st + 1][1]);
        pf[i].u[3] = pack16x2<F16>(s[i][2 * st + 1][2], s[i][2 * st + 1][3]);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        union { bf16x8_t v; bf16x4_t hh[2]; } vf;
        const char* p0 = sv + (st * 32 + 4 * g + (li >> 2)) * 128 + (((dt * 32) ^ vsw) + 8 * (li & 3));
        vf.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(p0));
        vf.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(p0 + 16 * 128));
#pragma unroll
        for (int i = 0; i < 2; ++i) o[i][dt] = mfma16x16x32<F16>(vf.v, pf[i].v, o[i][dt]);
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
    block(std::integral_constant<int, KT>{}, sk, sk + BLK, KROWS);
  }
  {                                                    // the last block: only the key tiles that hold keys
    const int b = nblk - 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!active) return;
    const char* sk = smem + (b & 1) * 2 * BLK;
    const int nkeys = a.N - b * KROWS;
    if (KT >= 6 && nkeys <= 32) block(std::integral_constant<int, 2>{}, sk, sk + BLK, nkeys);
    else block(std::integral_constant<int, KT>{}, sk, sk + BLK, nkeys);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = (qt0 + i) * 16 + li;
    if (q < a.N) {
      const float inv = 1.0f / l_run[i];
      if (a.lse != nullptr && g == 0)   // p = exp2(s * scale_log2 - lse) reproduces the normalised probability
        a.lse[((size_t)f * a.H + h) * a.npad + q] = fmaf(m_run[i], a.scale_log2, __builtin_amdgcn_logf(l_run[i]));
      bf16_t* orow = obase + (size_t)q * a.D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<uint2*>(orow + dt * 16 + 4 * g) =
            make_uint2(pack16x2<F16>(o[i][dt][0] * inv, o[i][dt][1] * inv), pack16x2<F16>(o[i][dt][2] * inv, o[i][dt][3] * inv));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fp32 (parity mode)
// ------------------------------------------------------------------------------------------------
constexpr int VROW = (HD + 4) * 4;  // 272-B V rows: keys 4 apart fall on different banks for ds_read_b32
constexpr int F32_LDS = KB * 256 + KB * VROW;

__global__ __launch_bounds__(256, 1) void vit_attn_f32_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sk = smem;             // K block [224][64] f32, 256-B rows, 16-B chunks swizzled by (row & 15)
  char* sv = smem + KB * 256;  // V block [224][68] f32
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int f = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const size_t ld = (size_t)3 * a.D;
  const float* base = reinterpret_cast<const float*>(a.qkv) + (size_t)f * a.N * ld;
  const float* qb = base + h * HD;
  const float* kbp = base + a.D + h * HD;
  const float* vbp = base + 2 * a.D + h * HD;

  for (int rd = 0; rd < a.rounds; ++rd) {
    const int qt = (blockIdx.y * a.rounds + rd) * 4 + wave;
    const int qrow = min(qt * 16 + li, a.N - 1);
    f32x4_t qf[4];  // lane (query li, g): Q[q][4*(g+4h) .. +4], h = 0..3
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) qf[hh] = *reinterpret_cast<const f32x4_t*>(qb + (size_t)qrow * ld + 4 * (g + 4 * hh));

    float m_run = -1e30f, l_run = 0.f;
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    for (int kb = 0; kb < a.nblk; ++kb) {
      if (a.nblk > 1 || rd == 0) {
        if (kb > 0 || rd > 0) __syncthreads();
        // thread -> (row = tid/16 + 16*i, chunk = tid%16)
#pragma unroll
        for (int i = 0; i < KB / 16; ++i) {
          const int r = (tid >> 4) + 16 * i, c = tid & 15;
          const int key = kb * KB + r;
          float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
          if (key < a.N) {
            kv = *reinterpret_cast<const float4*>(kbp + (size_t)key * ld + c * 4);
            vv = *reinterpret_cast<const float4*>(vbp + (size_t)key * ld + c * 4);
          }
          *reinterpret_cast<float4*>(sk + r * 256 + ((c ^ (r & 15)) << 4)) = kv;
          *reinterpret_cast<float4*>(sv + r * VROW + (c << 4)) = vv;
        }
        __syncthreads();
      }
      f32x4_t s[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        s[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
          const int row = kt * 16 + li;
          const f32x4_t kf = *reinterpret_cast<const f32x4_t*>(sk + row * 256 + (((g + 4 * hh) ^ li) << 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[hh][e], s[kt], 0, 0, 0);
        }
      }
      float mx = -1e30f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb * KB + kt * 16 + 4 * g + r;
          float v = s[kt][r] * a.scale_log2;
          v = key < a.N ? v : -1e30f;
          s[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = exp2f(m_run - m_new);
      float ls = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = exp2f(s[kt][r] - m_new);
          s[kt][r] = p;
          ls += p;
        }
      ls += __shfl_xor(ls, 16, 64);
      ls += __shfl_xor(ls, 32, 64);
      l_run = l_run * alpha + ls;
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
      // O^T[d][q] += sum_key V[key][d] P[q][key]; MFMA step (kt, r): k-lane g <-> key kt*16 + 4g + r
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const char* vr = sv + (kt * 16 + 4 * g + r) * VROW + li * 4;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const float vv = *reinterpret_cast<const float*>(vr + dt * 64);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, s[kt][r], o[dt], 0, 0, 0);
          }
        }
    }
    const int q = qt * 16 + li;
    if (q < a.N) {
      const float inv = 1.0f / l_run;
      float* orow = reinterpret_cast<float*>(a.out) + ((size_t)f * a.N + q) * a.D + h * HD;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<float4*>(orow + dt * 16 + 4 * g) =
            make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
    }
  }
}

}  // namespace

// variant: 0 = default (transposing LDS read for V; two query tiles per wave when N = 193..208), 1 = 2-byte gather reads
// (cross-check path), 2 = one query tile per wave / synchronously staged 224-key blocks (the earlier kernels, kept for A/B
// runs), 3 = the two-tile kernel at 2 waves per SIMD, 4 = the streamed kernel with 64-key blocks, 5 = the streamed kernel for any N
// Forward of a TRAINABLE block in bf16 mode: the streamed kernel (any N) with the per-query log-sum-exp kept for
// mvf_vit_attn_bwd; lse is [F, H, 16 * ceil(N / 16)] floats (rows beyond N are never written)
extern "C" int mvf_vit_attn_fwd_lse(const void* qkv, void* out, float* lse, int F, int N, int H, int D, hipStream_t st) {
  MVF_CHECK_ARG(qkv && out && lse && F > 0 && N > 0 && H > 0 && D == H * HD);
  MVF_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0);
  return mvf_vit_attn32_impl(MVF_BF16, qkv, out, lse, F, N, H, D, 5, 0, st);
}

extern "C" int mvf_vit_attn_fwd_mxfp8(const void* qkv, void* q, unsigned* scales, int F, int N, int H, int D, hipStream_t st) {
  MVF_CHECK_ARG(qkv && q && scales && F > 0 && N > 0 && H > 0 && H % 2 == 0 && D == H * HD);
  return mvf_vit_attn32_impl(MVF_BF16, qkv, q, nullptr, F, N, H, D, 5, 0, st, scales);
}

// 1 when the default (variant 0) 16-bit kernel for N tokens normalises by the row sum of the ROUNDED probabilities (taken on the
// matrix pipe with the P.V product), 0 when by the fp32 sum of the unrounded ones.  The one statement of that convention: the
// dispatch below and vit_qkv_attn.hip's fused kernel follow it, the emulating oracle is tested against it (tests/test_abi.py).
// (Round 6: every 16-bit product kernel takes the row sums on the matrix pipe -- the one-block kernels of N = 193 .. 208 since round 4,
// the streamed 32-query-row kernel of every other N now -- so the answer no longer depends on N.)
extern "C" int mvf_vit_attn_rowsum_rounded(int dtype, int N) { return (dtype == MVF_BF16 || dtype == MVF_F16) && N > 0 ? 1 : 0; }
// N served by the one-key-block kernels (13 key tiles of 16: ViT-B/16 at 224 px and its neighbours)
static bool attn_one_block(int N) { return ceil_div(N, KB) == 1 && ceil_div(N, 16) == 13; }
// frozen backbones whose attention runs on the streamed 32-query-row kernel get q rows pre-scaled by log2(e) / 8 (no v_fma per score:
// vit_attn32.hip QS); the one-block kernels and the fused qkv + attention kernel (193 .. 208 tokens) fold the factor into their packed
// exponent fma anyway and keep plain q
extern "C" int mvf_vit_attn_q_prescaled(int dtype, int N) {
  static const bool off = [] { const char* e = getenv("MVF_ATTN_QS"); return e != nullptr && e[0] == '0'; }();
  return !off && (dtype == MVF_BF16 || dtype == MVF_F16 || dtype == MVF_FP8) && N > 0 && !attn_one_block(N) ? 1 : 0;
}

int mvf_vit_attn_impl(int dtype, const void* qkv, void* out, int F, int N, int H, int D, int variant, hipStream_t st) {
  MVF_CHECK_ARG(qkv && out && F > 0 && N > 0 && H > 0 && D == H * HD);
  MVF_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0);
  AttnArgs a;
  a.lse = nullptr; a.npad = 0;
  a.qkv = (const char*)qkv; a.out = (char*)out; a.N = N; a.H = H; a.D = D;
  a.nblk = ceil_div(N, KB);
  const int qtiles = ceil_div(N, 16);
  int chunks;
  if (a.nblk == 1) { a.rounds = ceil_div(qtiles, 4); chunks = 1; }
  else { a.rounds = 1; chunks = ceil_div(qtiles, 4); }
  a.scale_log2 = LOG2E / 8.0f;  // 64^-0.5 * log2(e)
  dim3 grid(F * H, chunks);
  // the streamed 32-query-row kernel (vit_attn32.hip): the product path of every N outside the one-block specialisation; variants
  // 8 .. 23 select its forms for any N, 32 + form + 16 * waves also the workgroup size (A/B runs, tests)
  // + 0x1000: the q columns are pre-scaled by log2(e) / 8 (MVF_ATTN_Q_PRESCALED: the frozen backbone's packed weights carry the factor)
  const bool qs = (variant & MVF_ATTN_Q_PRESCALED) != 0;
  variant &= ~MVF_ATTN_Q_PRESCALED;
  if (qs) {
    MVF_CHECK_ARG(dtype == MVF_BF16 || dtype == MVF_F16);
    a.scale_log2 = 1.0f;       // the one-block kernels: the same arithmetic with the factor already in q
    if (variant == 0 && !attn_one_block(N)) return mvf_vit_attn32_impl(dtype, qkv, out, nullptr, F, N, H, D, 5 + 16, 0, st);
    if (variant == 8 + 5) return mvf_vit_attn32_impl(dtype, qkv, out, nullptr, F, N, H, D, 5 + 16, 0, st);
    MVF_CHECK_ARG(variant == 0 || variant == 6);
  }
  if ((dtype == MVF_BF16 || dtype == MVF_F16) && ((variant == 0 && !attn_one_block(N)) || (variant >= 8 && variant < 24)))
    return mvf_vit_attn32_impl(dtype, qkv, out, nullptr, F, N, H, D, variant == 0 ? 5 : variant - 8, 0, st);
  if ((dtype == MVF_BF16 || dtype == MVF_F16) && variant >= 32 && variant < 32 + 144)
    return mvf_vit_attn32_impl(dtype, qkv, out, nullptr, F, N, H, D, (variant - 32) & 15, (variant - 32) >> 4, st);
  if (dtype == MVF_BF16) {
    const int ntile = ceil_div(N, 16);
    const dim3 fg(F * H, ceil_div(ntile, 8));   // streamed kernels: 8 query tiles per workgroup
    if (variant == 1) hipLaunchKernelGGL((vit_attn_bf16_kernel<false, KT>), grid, dim3(256), 0, st, a);
    else if (attn_one_block(N) && variant == 0)          // row sums on the matrix pipe: 73.8 -> 72.3 us, 1 358 -> 1 325 W sustained
      hipLaunchKernelGGL((vit_attn_bf16_pair_kernel<13, 3, false, true>), grid, dim3(256), 0, st, a);
    else if (a.nblk == 1 && ntile == 13 && variant == 6)     // row sums on the VALU from the unrounded probabilities (the earlier form)
      hipLaunchKernelGGL((vit_attn_bf16_pair_kernel<13, 3>), grid, dim3(256), 0, st, a);
    else if (a.nblk == 1 && ntile == 13) hipLaunchKernelGGL((vit_attn_bf16_kernel<true, 13>), grid, dim3(256), 0, st, a);
    else if (variant == 7) hipLaunchKernelGGL((vit_attn_bf16_flash_kernel<6, 3>), fg, dim3(256), 0, st, a);   // 96-key blocks: the streamed kernel of rounds 2-5
    else if (variant == 4) hipLaunchKernelGGL((vit_attn_bf16_flash_kernel<4, 4>), fg, dim3(256), 0, st, a);   // 64-key blocks
    else hipLaunchKernelGGL((vit_attn_bf16_kernel<true, KT>), grid, dim3(256), 0, st, a);
  } else if (dtype == MVF_F16) {   // fp16 q / k / v / out: the two-tile kernel (N = 193 .. 208) or the streamed kernel (any N)
    const int ntile = ceil_div(N, 16);
    const dim3 fg(F * H, ceil_div(ntile, 8));
    if (attn_one_block(N)) hipLaunchKernelGGL((vit_attn_bf16_pair_kernel<13, 3, true, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((vit_attn_bf16_flash_kernel<6, 3, 4, true>), fg, dim3(256), 0, st, a);   // variant 7 (variant 0 left above)
  } else if (dtype == MVF_F32) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vit_attn_f32_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, F32_LDS);
      attr = true;
    }
    hipLaunchKernelGGL(vit_attn_f32_kernel, grid, dim3(256), F32_LDS, st, a);
  } else {
    return MVF_ERR_ARG;
  }
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}
