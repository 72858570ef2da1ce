// Row-chain kernels of the trainable MV-Former head on the 16-bit matrix cores: a workgroup owns 32 rows of the
// [B*S, d] activation matrix and walks a whole CHAIN of row-wise operators on them -- LayerNorm, Linear (+bias, ReLU,
// dropout, residual), the next LayerNorm, the next Linear ... -- with the activations in LDS and the weights streamed
// from L2 straight into MFMA fragments.  One launch stands for what used to be 5..8 launches of head_gemm.hip /
// head_misc.hip kernels (the reference: one ATen kernel per operator and more).
//
//   mvf_enc_layer_fwd  temporal EncoderLayer (CARL_MVF/models/utils.py:196-226): [attention output -> linear_d2Q + dropout
//                      + residual -> LayerNorm -> fc1 + ReLU -> fc2 + dropout + residual] and, for the NEXT layer,
//                      [LayerNorm -> Q|K|V projection] (utils.py:75-108,147-194); the attention core itself stays in
//                      head_attn_mfma.hip (it mixes the rows of a clip)
//   mvf_enc_layer_bwd  the same chain backwards: [dQKV -> input gradient of the Q|K|V projection -> LayerNorm backward +
//                      residual gradient] and [dropout mask -> fc2^T -> ReLU mask -> fc1^T -> LayerNorm backward + residual
//                      -> dropout mask -> linear_d2Q^T]; every Linear's output gradient is also written TRANSPOSED (bf16)
//                      for the weight-gradient kernel
//   mvf_head_dw        the weight / bias gradients of up to 16 Linears in ONE launch (dW (+)= g^T x over the rows, both
//                      operands row-transposed bf16, so fragments are 16-byte global loads)
//   mvf_head_pack_weights  fp32 master weights -> bf16 [N, K] and bf16 transposed [K, N] copies (once per optimizer step)
//
// Numerics: GEMM operands bf16 (what fp16 autocast does to these layers in the reference, train.py:113-117, with bf16's
// range instead of a loss scaler), fp32 accumulation; LayerNorm, bias, dropout, residual stream and every saved
// statistic fp32.  oracle/head.py `emulate='bf16'` rounds at the same points.  MI355X.COMPUTE_DTYPE fp32 keeps the fp32
// kernels of head_gemm.hip / head_misc.hip.
//
// gfx950 design: 256 threads = 4 waves, one workgroup per CU (up to 158 KB of LDS panels).  A GEMM stage computes
// out[32, N] = A[32, K] W[N, K]^T with v_mfma_f32_16x16x32_bf16: A fragments by ds_read_b128 from the bf16 LDS panel (row
// pad 16 B), W fragments as 16-byte global loads four k-steps ahead -- a wave owns 64 output columns at a time, so no W
// element is used by two waves and staging it in LDS would only add a barrier.  The operands are swapped in the MFMA
// (D^T = W A^T) so that a lane owns 4 consecutive output columns of one row: float4 epilogue traffic.  M = 768 rows are
// 24 workgroups: latency-bound by design (the step is bound by the frozen backbone beside it; what the head costs it is
// launches), each streaming <= 1.6 MB of bf16 weights from L2.
#include "common.h"
#include "mvf_hip_internal.h"

namespace {

// Workgroup barrier for LDS traffic only.  __syncthreads() also drains vmcnt(0): here that would wait for every global store of the
// stage before (row saves, transposed saves) and for the weight fragments already requested for the next GEMM -- nothing another
// thread of the workgroup reads from global memory inside these kernels.
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

constexpr int TM = 32;                 // rows per workgroup
constexpr int NTH = 512;                // chain kernels: 8 waves (the weight-gradient and pack kernels: NTH2 = 256)
constexpr int NTH2 = 256;
constexpr int NW = NTH / 64;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

struct Drop { uint32_t thresh; float scale; uint64_t seed, offset; };

Drop make_drop(const MvfDrop& d) {
  Drop r{};
  r.thresh = d.p > 0.f ? (uint32_t)std::min<double>(4294967295.0, (double)d.p * 4294967296.0) : 0u;
  r.scale = d.p > 0.f ? 1.0f / (1.0f - d.p) : 1.0f;
  r.seed = d.seed; r.offset = d.offset;
  return r;
}

__device__ __forceinline__ float drop_apply(const Drop& d, float v, uint64_t idx) {
  return d.thresh == 0u ? v : (drop_keep(d.seed, d.offset, idx, d.thresh) ? v * d.scale : 0.f);
}

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// ---------------------------------------------------------------------------------------------------------------
// out[32, N] = A[32, K] . W[N, K]^T      A: bf16 LDS panel (row stride lda elements, lda % 8 == 0), W: bf16 global, rows of
// ldw elements, k contiguous.  N % (16 NT) == 0.  epi(m, n, v): lane's four results out[m][n .. n+3].
// ---------------------------------------------------------------------------------------------------------------
// pre(m, n) -> Aux: whatever the epilogue needs from global memory for out[m][n .. n+3] (bias, residual); requested BEFORE the k loop
// of the chunk, so its latency hides under the loop.  (Loads inside the epilogue serialise behind the epilogue's own global stores
// -- the compiler must assume they alias -- one L2 round trip per tile: 70 of the 108 us of a layer launch.)
// weight-fragment load: non-temporal.  A layer launch streams 1.5 MB of weights through each of its 24 workgroups; with the default
// policy that stream displaces the operands of the backbone GEMMs running beside the head in the same L2s (measured with
// tools/stretch_parts.py: the encoder's launches cost the pipelined step 0.55 ms with plain loads, 0.37 ms with these).
// MVF_NT_OFF (build flag): plain loads, for A/B measurements.
#ifdef MVF_NT_OFF
#define WLOAD(p) (*reinterpret_cast<const bf16x8_t*>(p))
#else
#define WLOAD(p) __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(p))
#endif
struct NoAux {};
struct Aux1 { float4 b; };            // bias
struct Aux2 { float4 b, r; };         // bias + residual row

// FRAGMENT-MAJOR operand layout ("FM") of a bf16 matrix X[rows][red] (red = the reduction index of the GEMM it feeds):
//     FM[rows / 16][red / 32][64 lanes][8]      lane = row % 16 + 16 * ((red % 32) / 8),  element = red % 8
// i.e. every 16 x 32 block is stored exactly as the 64 lanes of v_mfma_f32_16x16x32_bf16 hold it, so a wave fetches a fragment with
// ONE fully coalesced 1 KB load (16 B per lane, consecutive lanes consecutive addresses).  The plain row-major form makes the same
// load touch 16 rows x 64 B: the texture-address unit then serves about one lane per cycle -- measured 15-26 GB/s per CU for the
// weight stream of a layer launch (1.5 MB: 90 us), independent of L2 warmth and of the number of loads in flight.
// rows are padded to a multiple of 64, red to a multiple of 128 (zeros).  fm_elems() = elements of the padded image.
__host__ __device__ inline size_t fm_elems(int rows, int red) { return (size_t)((rows + 63) & ~63) * ((red + 127) & ~127); }
__host__ __device__ inline int fm_steps(int red) { return ((red + 127) & ~127) >> 5; }     // 32-wide reduction steps of the padded image

// out[32, N] = A[32, K] . W^T    A: bf16 LDS panel (row stride lda elements, lda % 8 == 0, columns >= K up to the padded K hold
// zeros), W: FM image of [N, K].  N % (16 NT) == 0.  pre(m, n) -> Aux: what the epilogue needs from global memory for
// out[m][n .. n+3] (bias, residual), requested BEFORE the k loop of the chunk (loads inside the epilogue serialise behind the
// epilogue's own global stores -- the compiler must assume they alias).  epi(m, n, v, aux): the lane's four results.
// NT: 16-column tiles per wave and chunk; PF: k-steps in flight (NT * PF fragment loads per wave); steps % PF == 0 (PF <= 4).
template <int NT, int PF, typename Pre, typename Epi>
__device__ __forceinline__ void chain_gemm(const bf16_t* A, int lda, int K, const bf16_t* __restrict__ W, int N, Pre pre, Epi epi) {
  // Software pipeline: the W fragments of PF k-steps are in flight in a ring of registers that is refilled in place right after
  // the MFMAs that consumed a slot -- straight-line code, no branch around a load (a conditional refill made hipcc load into
  // temporaries and wait for them in the same step), the tail refills re-read the last step (clamped address), and sched_barriers
  // pin every request where it is written: left alone the scheduler sinks the refills to just before their use (register
  // pressure heuristic: one load in flight) and reorders the prologue (the loop's static vmcnt then has to be 0).
  const int lane = threadIdx.x & 63, wave = wave_id(), c = lane & 15, g = lane >> 4;
  const bf16_t* a0p = A + c * lda + 8 * g;
  const bf16_t* a1p = a0p + 16 * lda;
  const int nsteps = fm_steps(K);
  for (int n0 = wave * 16 * NT; n0 < N; n0 += NW * 16 * NT) {
    f32x4_t acc[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { acc[0][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[1][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
    decltype(pre(0, 0)) aux[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) aux[mt][nt] = pre(mt * 16 + c, n0 + nt * 16 + 4 * g);
    const bf16_t* wp = W + ((size_t)(n0 >> 4) * nsteps * 64 + lane) * 8;      // tile nt, step s: + (nt * nsteps + s) * 512
    bf16x8_t bq[PF][NT];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bq[p][nt] = WLOAD(wp + (size_t)(nt * nsteps + p) * 512);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(a0p), a1 = *reinterpret_cast<const bf16x8_t*>(a1p);
    for (int s0 = 0; s0 < nsteps; s0 += PF) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        const int st = s0 + p;
        const int kn = min(st + 1, nsteps - 1) << 5;
        const bf16x8_t a0n = *reinterpret_cast<const bf16x8_t*>(a0p + kn), a1n = *reinterpret_cast<const bf16x8_t*>(a1p + kn);
        __builtin_amdgcn_sched_barrier(0);     // the next step's A fragments are requested BEFORE this step's MFMAs
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[p][nt], a0, acc[0][nt], 0, 0, 0);
          acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[p][nt], a1, acc[1][nt], 0, 0, 0);
        }
        const int sr = min(st + PF, nsteps - 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq[p][nt] = WLOAD(wp + (size_t)(nt * nsteps + sr) * 512);
        a0 = a0n; a1 = a1n;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) epi(mt * 16 + c, n0 + nt * 16 + 4 * g, acc[mt][nt], aux[mt][nt]);
  }
}

// ---- panel helpers (all 256 threads; the caller places the barriers) ----

// fp32 global rows [m0, m0 + 32) x [0, C) (row stride ld) -> bf16 panel; rows >= M read as 0.  C % 4 == 0.
// (batches of 8 loads per thread are issued before the first is used: a run-time loop of load -> convert -> store pays one
// memory round trip per iteration)
__device__ __forceinline__ void load_rows_bf16(const float* __restrict__ src, long ld, int m0, int M, int C, bf16_t* P, int ldp) {
  const int c4 = C >> 2, total = TM * c4;
  constexpr int U = 8;
  for (int i0 = threadIdx.x; i0 < total; i0 += NTH * U) {
    f32x4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * NTH, r = i / c4, q = i - r * c4;
      v[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (i < total && m0 + r < M) v[u] = *reinterpret_cast<const f32x4_t*>(src + (size_t)(m0 + r) * ld + 4 * q);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * NTH, r = i / c4, q = i - r * c4;
      if (i < total) *reinterpret_cast<u32x2_t*>(P + r * ldp + 4 * q) = (u32x2_t){pack_bf16x2(v[u][0], v[u][1]), pack_bf16x2(v[u][2], v[u][3])};
    }
  }
}

// fp32 global rows -> fp32 panel
__device__ __forceinline__ void load_rows_f32(const float* __restrict__ src, long ld, int m0, int M, int C, float* P, int ldp) {
  const int c4 = C >> 2, total = TM * c4;
  constexpr int U = 8;
  for (int i0 = threadIdx.x; i0 < total; i0 += NTH * U) {
    f32x4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * NTH, r = i / c4, q = i - r * c4;
      v[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (i < total && m0 + r < M) v[u] = *reinterpret_cast<const f32x4_t*>(src + (size_t)(m0 + r) * ld + 4 * q);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * NTH, r = i / c4, q = i - r * c4;
      if (i < total) *reinterpret_cast<f32x4_t*>(P + r * ldp + 4 * q) = v[u];
    }
  }
}

// bf16 panel [32 rows m][C features] -> the FM image of the TRANSPOSE X^T[C][Mp] (rows = features, reduction = m): the operand
// form of the weight-gradient kernel.  Row block m0 is reduction step m0 / 32 of every feature tile; thread = one feature = the four
// lanes c + 16 g of that fragment.  Rows >= M are written as 0 (they would otherwise carry bias / LayerNorm-beta values into the
// weight gradients); the last row block also clears the steps up to the padded Mp.  Features >= C up to the padded 64 are never
// read (the gradient kernel clamps its tile rows).
__device__ __forceinline__ void store_T(const bf16_t* P, int ldp, int C, bf16_t* __restrict__ dst, int Mp, int m0, int M) {
  if (dst == nullptr) return;
  const int valid = min(TM, M - m0);
  const int msteps = Mp >> 5, ms = m0 >> 5;
  for (int cc = threadIdx.x; cc < C; cc += NTH) {
    unsigned w[TM / 2];
#pragma unroll
    for (int r = 0; r < TM; r += 2) {
      const unsigned lo = r < valid ? P[r * ldp + cc] : 0u, hi = r + 1 < valid ? P[(r + 1) * ldp + cc] : 0u;
      w[r >> 1] = lo | (hi << 16);
    }
    bf16_t* o = dst + (((size_t)(cc >> 4) * msteps + ms) * 64 + (cc & 15)) * 8;
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<u32x4_t*>(o + g * 128) = (u32x4_t){w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]};
    if (blockIdx.x == gridDim.x - 1)
      for (int q = ms + 1; q < msteps; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<u32x4_t*>(o + (size_t)(q - ms) * 512 + g * 128) = (u32x4_t){0u, 0u, 0u, 0u};
  }
}

// bf16 panel rows -> global bf16 [M, C] (16-byte stores)
__device__ __forceinline__ void store_rows_bf16(const bf16_t* P, int ldp, int C, bf16_t* __restrict__ dst, int m0, int M) {
  if (dst == nullptr) return;
  const int c8 = C >> 3;
  for (int i = threadIdx.x; i < TM * c8; i += NTH) {
    const int r = i / c8, q = i - r * c8;
    if (m0 + r < M)
      *reinterpret_cast<u32x4_t*>(dst + (size_t)(m0 + r) * C + 8 * q) = *reinterpret_cast<const u32x4_t*>(P + r * ldp + 8 * q);
  }
}

// LayerNorm of the fp32 panel rows -> bf16 panel; (mean, rstd) -> global.  16 lanes per row (512 threads = 32 rows): a lane sums
// D / 16 elements, four xor-shuffles finish a row -- every row of the panel at once.  (One wave per row spends its time in the
// cross-lane reductions: 12 ds_bpermute round trips per row, 10 us for the panel.)
__device__ __forceinline__ float sum16(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}
__device__ __forceinline__ void ln_panel(const float* X, int ldx, int D, const float* __restrict__ gam, const float* __restrict__ bet,
                                         float eps, bf16_t* H, int ldh, float* __restrict__ mean, float* __restrict__ rstd, int m0,
                                         int M) {
  constexpr int NC = 32;                // D <= 512: columns per lane
  const int r = threadIdx.x >> 4, sub = threadIdx.x & 15;
  const int nc = D >> 4;
  const float* xr = X + r * ldx;
  float xv[NC], gv[NC], bv[NC];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < NC; ++q)
    if (q < nc) {
      const int cc = sub + 16 * q;
      xv[q] = xr[cc]; gv[q] = gam[cc]; bv[q] = bet[cc];
      s += xv[q];
    }
  const float mu = sum16(s) / D;
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < NC; ++q)
    if (q < nc) { const float d = xv[q] - mu; ss += d * d; }
  const float rs = rsqrtf(sum16(ss) / D + eps);
#pragma unroll
  for (int q = 0; q < NC; ++q)
    if (q < nc) H[r * ldh + sub + 16 * q] = f32_to_bf16((xv[q] - mu) * rs * gv[q] + bv[q]);
  if (sub == 0 && m0 + r < M) {
    if (mean) mean[m0 + r] = mu;
    if (rstd) rstd[m0 + r] = rs;
  }
}

// LayerNorm backward on panels: DX += d LN(x) / dx applied to DH, i.e. DX[r][c] += rs (dh g - c1 - xh c2) (DX holds the
// residual-path gradient on entry); x rows and the statistics are read from global.  16 lanes per row as in ln_panel.
// dbeta[c] = sum_r dh, dgamma[c] = sum_r dh xh: column sums over the panel's rows (rows >= M hold zeros: their operands were
// zero-filled), taken from DH before and after it is overwritten in place with dh xh; one float atomic per column and workgroup.
__device__ __forceinline__ void ln_bwd_panel(float* DH, int lddh, float* DX, int lddx, int D, const float* __restrict__ x,
                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                             const float* __restrict__ gam, float* __restrict__ dg, float* __restrict__ db, int m0,
                                             int M) {
  constexpr int NC = 32;
  const int r = threadIdx.x >> 4, sub = threadIdx.x & 15;
  const int nc = D >> 4;
  const int gm = min(m0 + r, M - 1);
  const bool live = m0 + r < M;
  float xh[NC], gv[NC];
  const float mu = mean[gm], rs = rstd[gm];
#pragma unroll
  for (int q = 0; q < NC; ++q)
    if (q < nc) {
      const int cc = sub + 16 * q;
      xh[q] = x[(size_t)gm * D + cc];
      gv[q] = gam[cc];
    }
  if (db != nullptr)
    for (int cc = threadIdx.x; cc < D; cc += NTH) {
      float a = 0.f;
#pragma unroll 8
      for (int rr = 0; rr < TM; ++rr) a += DH[rr * lddh + cc];
      atomicAdd(db + cc, a);
    }
  LDS_BARRIER();
  float c1 = 0.f, c2 = 0.f, dgv[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q)
    if (q < nc) {
      const float d = live ? DH[r * lddh + sub + 16 * q] : 0.f;
      xh[q] = (xh[q] - mu) * rs;
      dgv[q] = d * gv[q];
      c1 += dgv[q];
      c2 += dgv[q] * xh[q];
      DH[r * lddh + sub + 16 * q] = d * xh[q];
    }
  c1 = sum16(c1) / D;
  c2 = sum16(c2) / D;
  if (live) {
#pragma unroll
    for (int q = 0; q < NC; ++q)
      if (q < nc) DX[r * lddx + sub + 16 * q] += rs * (dgv[q] - c1 - xh[q] * c2);
  }
  LDS_BARRIER();
  if (dg != nullptr)
    for (int cc = threadIdx.x; cc < D; cc += NTH) {
      float a = 0.f;
#pragma unroll 8
      for (int rr = 0; rr < TM; ++rr) a += DH[rr * lddh + cc];
      atomicAdd(dg + cc, a);
    }
}

// bf16 panel <- dropout-masked fp32 panel (the operand of a Linear's backward whose forward ended in dropout):
// G[r][c] = bf16(mask(m*C + c) * X[r][c]); rows >= M -> 0
__device__ __forceinline__ void mask_to_bf16(const float* X, int ldx, int C, const Drop& d, bf16_t* G, int ldg, int m0, int M) {
  const int c4 = C >> 2;
  for (int i = threadIdx.x; i < TM * c4; i += NTH) {
    const int r = i / c4, q = i - r * c4;
    float4 v = *reinterpret_cast<const float4*>(X + r * ldx + 4 * q);
    if (m0 + r >= M) v = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint64_t idx = (uint64_t)(m0 + r) * C + 4 * q;
    v.x = drop_apply(d, v.x, idx); v.y = drop_apply(d, v.y, idx + 1); v.z = drop_apply(d, v.z, idx + 2); v.w = drop_apply(d, v.w, idx + 3);
    *reinterpret_cast<u32x2_t*>(G + r * ldg + 4 * q) = (u32x2_t){pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
  }
}

// LDS carve-up shared by the two encoder kernels (bytes); D, DFF multiples of 64
struct EncLds {
  int ldf, ldb0, ldb1;      // element strides
  size_t pf0, pf1, pb0, pb1, total;
};
__host__ __device__ inline EncLds enc_lds(int D, int DFF, bool bwd) {
  EncLds l;
  l.ldf = D + 4; l.ldb0 = D + 8;
  const int wide = DFF > 3 * D ? DFF : 3 * D;       // the wide bf16 panel also takes the dQKV rows in the backward
  l.ldb1 = (bwd ? wide : DFF) + 8;
  size_t o = 0;
  l.pf0 = o; o += (size_t)TM * l.ldf * 4;
  l.pf1 = o; if (bwd) o += (size_t)TM * l.ldf * 4;
  l.pb0 = o; o += (size_t)TM * l.ldb0 * 2;
  l.pb1 = o; o += (size_t)TM * l.ldb1 * 2;
  l.total = o;
  return l;
}

// measurement knob (tools/chain_probe.py): bit 0 skips the transposed saves, bit 1 cuts every GEMM's k loop to one round, bit 2
// skips the row saves of `a`; results are wrong with any bit set
int g_chain_dbg = 0;
long long* g_chain_stamps = nullptr;   // diagnostic builds of the probe only: wave 0 of workgroup 0 stores s_memrealtime (100 MHz) at stage boundaries
#define STAMP(i) do { if (k.stamps != nullptr && threadIdx.x == 0 && blockIdx.x == 0) k.stamps[i] = wall_clock64(); } while (0)

struct EncFwdK {
  int M, D, DFF, Mp, dbg;
  float eps;
  const float *o, *x_in;
  const bf16_t *wo, *w1, *w2, *wqkv;
  const float *bo, *b1, *b2, *bqkv, *g1, *be1, *g0, *be0;
  Drop da, df;
  float *x1, *mean1, *rstd1, *x2, *qkv, *mean0, *rstd0;
  bf16_t *a, *oT, *h1T, *aT, *h0T;
  long long* stamps;
};

__global__ __launch_bounds__(NTH) void enc_fwd_kernel(EncFwdK k) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const EncLds L = enc_lds(k.D, k.DFF, false);
  float* Pf = reinterpret_cast<float*>(sm + L.pf0);
  bf16_t* Pb0 = reinterpret_cast<bf16_t*>(sm + L.pb0);
  bf16_t* Pb1 = reinterpret_cast<bf16_t*>(sm + L.pb1);
  const int m0 = blockIdx.x * TM, M = k.M, D = k.D, DFF = k.DFF;
  if (k.dbg & 1) { k.oT = k.h1T = k.aT = k.h0T = nullptr; }
  if (k.dbg & 4) k.a = nullptr;
  const int KD = (k.dbg & 2) ? 128 : D, KF = (k.dbg & 2) ? 128 : DFF;
  STAMP(0);
  if (k.o != nullptr) {
    // ---- x1 = x + drop(o Wo^T + bo) ----
    load_rows_bf16(k.o, D, m0, M, D, Pb0, L.ldb0);
    LDS_BARRIER();
    STAMP(1);
    store_T(Pb0, L.ldb0, D, k.oT, k.Mp, m0, M);
    STAMP(2);
    chain_gemm<2, 4>(Pb0, L.ldb0, KD, k.wo, D, [&](int m, int n) {
      Aux2 a;
      a.b = *reinterpret_cast<const float4*>(k.bo + n);
      a.r = *reinterpret_cast<const float4*>(k.x_in + (size_t)min(m0 + m, M - 1) * D + n);
      return a;
    }, [&](int m, int n, const f32x4_t& v, const Aux2& ax) {
      const int gm = m0 + m;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gm < M) {
        const float4 bb = ax.b, xi = ax.r;
        const uint64_t idx = (uint64_t)gm * D + n;
        r.x = xi.x + drop_apply(k.da, v[0] + bb.x, idx);
        r.y = xi.y + drop_apply(k.da, v[1] + bb.y, idx + 1);
        r.z = xi.z + drop_apply(k.da, v[2] + bb.z, idx + 2);
        r.w = xi.w + drop_apply(k.da, v[3] + bb.w, idx + 3);
        if (k.x1) *reinterpret_cast<float4*>(k.x1 + (size_t)gm * D + n) = r;
      }
      *reinterpret_cast<float4*>(Pf + m * L.ldf + n) = r;
    });
    LDS_BARRIER();
    STAMP(3);
    // ---- h1 = LN(x1);  a = relu(h1 W1^T + b1) ----
    ln_panel(Pf, L.ldf, D, k.g1, k.be1, k.eps, Pb0, L.ldb0, k.mean1, k.rstd1, m0, M);
    LDS_BARRIER();
    STAMP(4);
    store_T(Pb0, L.ldb0, D, k.h1T, k.Mp, m0, M);
    STAMP(5);
    chain_gemm<4, 4>(Pb0, L.ldb0, KD, k.w1, DFF, [&](int, int n) {
      Aux1 a;
      a.b = *reinterpret_cast<const float4*>(k.b1 + n);
      return a;
    }, [&](int m, int n, const f32x4_t& v, const Aux1& ax) {
      const float4 bb = ax.b;
      *reinterpret_cast<u32x2_t*>(Pb1 + m * L.ldb1 + n) =
          (u32x2_t){pack_bf16x2(fmaxf(v[0] + bb.x, 0.f), fmaxf(v[1] + bb.y, 0.f)), pack_bf16x2(fmaxf(v[2] + bb.z, 0.f), fmaxf(v[3] + bb.w, 0.f))};
    });
    LDS_BARRIER();
    STAMP(6);
    store_rows_bf16(Pb1, L.ldb1, DFF, k.a, m0, M);
    store_T(Pb1, L.ldb1, DFF, k.aT, k.Mp, m0, M);
    STAMP(7);
    // ---- x2 = x1 + drop(a W2^T + b2) ----
    chain_gemm<2, 4>(Pb1, L.ldb1, KF, k.w2, D, [&](int, int n) {
      Aux1 a;
      a.b = *reinterpret_cast<const float4*>(k.b2 + n);
      return a;
    }, [&](int m, int n, const f32x4_t& v, const Aux1& ax) {
      const int gm = m0 + m;
      float4 r = *reinterpret_cast<const float4*>(Pf + m * L.ldf + n);
      if (gm < M) {
        const float4 bb = ax.b;
        const uint64_t idx = (uint64_t)gm * D + n;
        r.x += drop_apply(k.df, v[0] + bb.x, idx);
        r.y += drop_apply(k.df, v[1] + bb.y, idx + 1);
        r.z += drop_apply(k.df, v[2] + bb.z, idx + 2);
        r.w += drop_apply(k.df, v[3] + bb.w, idx + 3);
        if (k.x2) *reinterpret_cast<float4*>(k.x2 + (size_t)gm * D + n) = r;
      }
      *reinterpret_cast<float4*>(Pf + m * L.ldf + n) = r;
    });
    LDS_BARRIER();
    STAMP(8);
  } else if (k.wqkv != nullptr) {
    load_rows_f32(k.x_in, D, m0, M, D, Pf, L.ldf);
    LDS_BARRIER();
  }
  if (k.wqkv != nullptr) {
    // ---- the next layer's h0 = LN(x);  qkv = h0 Wqkv^T + bqkv ----
    ln_panel(Pf, L.ldf, D, k.g0, k.be0, k.eps, Pb0, L.ldb0, k.mean0, k.rstd0, m0, M);
    LDS_BARRIER();
    STAMP(9);
    store_T(Pb0, L.ldb0, D, k.h0T, k.Mp, m0, M);
    STAMP(10);
    chain_gemm<2, 4>(Pb0, L.ldb0, KD, k.wqkv, 3 * D, [&](int, int n) {
      Aux1 a;
      a.b = *reinterpret_cast<const float4*>(k.bqkv + n);
      return a;
    }, [&](int m, int n, const f32x4_t& v, const Aux1& ax) {
      const int gm = m0 + m;
      if (gm < M) {
        const float4 bb = ax.b;
        *reinterpret_cast<float4*>(k.qkv + (size_t)gm * 3 * D + n) = make_float4(v[0] + bb.x, v[1] + bb.y, v[2] + bb.z, v[3] + bb.w);
      }
    });
    STAMP(11);
  }
}

struct EncBwdK {
  int M, D, DFF, Mp;
  const float *dqkv, *x_in, *mean0, *rstd0, *g0, *dres;
  const bf16_t *wqkvT, *w2T, *w1T, *woT, *a;
  float *dg0, *db0, *dx_out;
  bf16_t *dqkvT, *g2T, *duT, *goT;
  Drop df, da;
  const float *x1, *mean1, *rstd1, *g1;
  float *dg1, *db1, *dx1_out, *d_o;
};

__global__ __launch_bounds__(NTH) void enc_bwd_kernel(EncBwdK k) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const EncLds L = enc_lds(k.D, k.DFF, true);
  float* Pf0 = reinterpret_cast<float*>(sm + L.pf0);      // the gradient on the residual stream
  float* Pf1 = reinterpret_cast<float*>(sm + L.pf1);      // a LayerNorm's output gradient
  bf16_t* Pb0 = reinterpret_cast<bf16_t*>(sm + L.pb0);
  bf16_t* Pb1 = reinterpret_cast<bf16_t*>(sm + L.pb1);
  const int m0 = blockIdx.x * TM, M = k.M, D = k.D, DFF = k.DFF;
  load_rows_f32(k.dres, D, m0, M, D, Pf0, L.ldf);
  if (k.dqkv != nullptr) {
    // ---- dh0 = dqkv Wqkv;  dx = dres + LN0'(dh0) ----
    load_rows_bf16(k.dqkv, 3 * D, m0, M, 3 * D, Pb1, L.ldb1);
    LDS_BARRIER();
    store_T(Pb1, L.ldb1, 3 * D, k.dqkvT, k.Mp, m0, M);
    chain_gemm<2, 4>(Pb1, L.ldb1, 3 * D, k.wqkvT, D, [](int, int) { return NoAux{}; }, [&](int m, int n, const f32x4_t& v, const NoAux&) {
      *reinterpret_cast<float4*>(Pf1 + m * L.ldf + n) = make_float4(v[0], v[1], v[2], v[3]);
    });
    LDS_BARRIER();
    ln_bwd_panel(Pf1, L.ldf, Pf0, L.ldf, D, k.x_in, k.mean0, k.rstd0, k.g0, k.dg0, k.db0, m0, M);
    LDS_BARRIER();
    if (k.dx_out != nullptr) {
      const int c4 = D >> 2;
      for (int i = threadIdx.x; i < TM * c4; i += NTH) {
        const int r = i / c4, q = i - r * c4;
        if (m0 + r < M) *reinterpret_cast<float4*>(k.dx_out + (size_t)(m0 + r) * D + 4 * q) = *reinterpret_cast<const float4*>(Pf0 + r * L.ldf + 4 * q);
      }
    }
  } else {
    LDS_BARRIER();
  }
  if (k.w2T == nullptr) return;
  // ---- g2 = mask_f(dy);  du = (g2 W2) [a > 0];  dh1 = du W1;  dx1 = dy + LN1'(dh1) ----
  mask_to_bf16(Pf0, L.ldf, D, k.df, Pb0, L.ldb0, m0, M);
  LDS_BARRIER();
  store_T(Pb0, L.ldb0, D, k.g2T, k.Mp, m0, M);
  struct AuxA { u32x2_t a; };
  chain_gemm<4, 4>(Pb0, L.ldb0, D, k.w2T, DFF, [&](int m, int n) {
    AuxA x;
    x.a = *reinterpret_cast<const u32x2_t*>(k.a + (size_t)min(m0 + m, M - 1) * DFF + n);
    return x;
  }, [&](int m, int n, const f32x4_t& v, const AuxA& ax) {
    const u32x2_t av = ax.a;
    const float d0 = (av[0] & 0xffffu) != 0u && !(av[0] & 0x8000u) ? v[0] : 0.f;
    const float d1 = (av[0] >> 16) != 0u && !(av[0] & 0x80000000u) ? v[1] : 0.f;
    const float d2 = (av[1] & 0xffffu) != 0u && !(av[1] & 0x8000u) ? v[2] : 0.f;
    const float d3 = (av[1] >> 16) != 0u && !(av[1] & 0x80000000u) ? v[3] : 0.f;
    *reinterpret_cast<u32x2_t*>(Pb1 + m * L.ldb1 + n) = (u32x2_t){pack_bf16x2(d0, d1), pack_bf16x2(d2, d3)};
  });
  LDS_BARRIER();
  store_T(Pb1, L.ldb1, DFF, k.duT, k.Mp, m0, M);
  chain_gemm<2, 4>(Pb1, L.ldb1, DFF, k.w1T, D, [](int, int) { return NoAux{}; }, [&](int m, int n, const f32x4_t& v, const NoAux&) {
    *reinterpret_cast<float4*>(Pf1 + m * L.ldf + n) = make_float4(v[0], v[1], v[2], v[3]);
  });
  LDS_BARRIER();
  ln_bwd_panel(Pf1, L.ldf, Pf0, L.ldf, D, k.x1, k.mean1, k.rstd1, k.g1, k.dg1, k.db1, m0, M);
  LDS_BARRIER();
  // ---- dx1 out;  g_o = mask_a(dx1);  d_o = g_o Wo ----
  {
    const int c4 = D >> 2;
    for (int i = threadIdx.x; i < TM * c4; i += NTH) {
      const int r = i / c4, q = i - r * c4;
      if (m0 + r < M) *reinterpret_cast<float4*>(k.dx1_out + (size_t)(m0 + r) * D + 4 * q) = *reinterpret_cast<const float4*>(Pf0 + r * L.ldf + 4 * q);
    }
  }
  mask_to_bf16(Pf0, L.ldf, D, k.da, Pb0, L.ldb0, m0, M);
  LDS_BARRIER();
  store_T(Pb0, L.ldb0, D, k.goT, k.Mp, m0, M);
  chain_gemm<2, 4>(Pb0, L.ldb0, D, k.woT, D, [](int, int) { return NoAux{}; }, [&](int m, int n, const f32x4_t& v, const NoAux&) {
    const int gm = m0 + m;
    if (gm < M) *reinterpret_cast<float4*>(k.d_o + (size_t)gm * D + n) = make_float4(v[0], v[1], v[2], v[3]);
  });
}

// ---------------------------------------------------------------------------------------------------------------
// weight / bias gradients of several Linears in one launch.  Problem p: dW[n][k] (+)= sum_m gT[n][m] xT[k][m],
// db[n] (+)= sum_m gT[n][m]; gT, xT: FM images (rows n resp. k, reduction m) written by store_T.  One workgroup = one
// 64 (n) x 64 (k) tile, a wave = 32 x 32 of it; both operands come straight from global memory as 1 KB fragment loads, 4 m-steps
// ahead (ring refilled in place, see chain_gemm).
// ---------------------------------------------------------------------------------------------------------------
constexpr int DW_MAX = 16;
struct DwProb { const bf16_t* gT; const bf16_t* xT; float* dw; long lddw; float* db; int N, K, tiles_k, tile0; };
struct DwArgs { DwProb p[DW_MAX]; int n, Mp, accumulate; };

__global__ __launch_bounds__(NTH2) void head_dw_kernel(DwArgs a) {
  int pi = 0;
  const int b = blockIdx.x;
#pragma unroll 1
  for (int i = 1; i < a.n; ++i)
    if (b >= a.p[i].tile0) pi = i;
  const DwProb& P = a.p[pi];
  const int t = b - P.tile0, tn = t / P.tiles_k, tk = t - tn * P.tiles_k;
  const int lane = threadIdx.x & 63, wave = wave_id(), c = lane & 15, g = lane >> 4;
  const int n0 = tn * 64 + (wave >> 1) * 32, k0 = tk * 64 + (wave & 1) * 32;
  const int nsteps = a.Mp >> 5;         // Mp % 128 == 0: nsteps % PF == 0
  constexpr int PF = 4;
  // fragment bases of the wave's two n tiles and two k tiles (tile rows past N / K exist in the padded images: computed, dropped)
  const bf16_t* gp[2];
  const bf16_t* xp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    gp[i] = P.gT + ((size_t)((n0 >> 4) + i) * nsteps * 64 + lane) * 8;
    xp[i] = P.xT + ((size_t)((k0 >> 4) + i) * nsteps * 64 + lane) * 8;
  }
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float bs[2] = {0.f, 0.f};
  const bool want_b = P.db != nullptr && tk == 0 && (wave & 1) == 0;
  bf16x8_t gq[PF][2], xq[PF][2];
#pragma unroll
  for (int p = 0; p < PF; ++p) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      gq[p][i] = *reinterpret_cast<const bf16x8_t*>(gp[i] + (size_t)p * 512);
      xq[p][i] = *reinterpret_cast<const bf16x8_t*>(xp[i] + (size_t)p * 512);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  for (int s0 = 0; s0 < nsteps; s0 += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      // D[k-row 4g + r][n-col c] : lane owns dW[n = c][k .. k+3]
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xq[p][j], gq[p][i], acc[i][j], 0, 0, 0);
      if (want_b) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) bs[i] += bf16_to_f32((bf16_t)gq[p][i][e]);
      }
      const int mr = min(s0 + p + PF, nsteps - 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        gq[p][i] = *reinterpret_cast<const bf16x8_t*>(gp[i] + (size_t)mr * 512);
        xq[p][i] = *reinterpret_cast<const bf16x8_t*>(xp[i] + (size_t)mr * 512);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + 16 * i + c;
    if (want_b) {
      float v = bs[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0 && n < P.N) P.db[n] = a.accumulate ? P.db[n] + v : v;
    }
    if (n >= P.N) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kk = k0 + 16 * j + 4 * g;
      float* dp = P.dw + (size_t)n * P.lddw + kk;
      if (kk + 3 < P.K && (P.lddw & 3) == 0 && ((uintptr_t)P.dw & 15) == 0) {
        float4 o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if (a.accumulate) { const float4 q = *reinterpret_cast<const float4*>(dp); o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
        *reinterpret_cast<float4*>(dp) = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kk + r < P.K) dp[r] = a.accumulate ? dp[r] + acc[i][j][r] : acc[i][j][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 master weights -> bf16 FM operands.  Entry e: w [N, K] (row stride ld) -> w16 = FM image of W (rows n, reduction k: the
// forward operand) and w16t = FM image of W^T (rows k, reduction n: the input-gradient operand).  One thread per 16-byte piece.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PACK_MAX = 32;
struct PackEnt { const float* w; long ld; int N, K; unsigned piece0, pieces16; bf16_t* w16; bf16_t* w16t; };
struct PackArgs { PackEnt e[PACK_MAX]; int n; };

__global__ __launch_bounds__(NTH2) void head_pack_kernel(PackArgs a, unsigned total) {
  for (unsigned pid = blockIdx.x * NTH2 + threadIdx.x; pid < total; pid += gridDim.x * NTH2) {
    int ei = 0;
#pragma unroll 1
    for (int i = 1; i < a.n; ++i)
      if (pid >= a.e[i].piece0) ei = i;
    const PackEnt& E = a.e[ei];
    unsigned q = pid - E.piece0;
    const bool tr = q >= E.pieces16;            // second half of the entry's pieces: the transposed image
    if (tr) q -= E.pieces16;
    bf16_t* dst = tr ? E.w16t : E.w16;
    if (dst == nullptr) continue;
    const int rows = tr ? E.K : E.N, red = tr ? E.N : E.K;
    const int steps = fm_steps(red);
    const int lane = q & 63, blk = q >> 6, rt = blk / steps, stp = blk - rt * steps;
    const int r = rt * 16 + (lane & 15), c0 = stp * 32 + (lane >> 4) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cc = c0 + e;
      const bool in = r < rows && cc < red;
      v[e] = in ? (tr ? E.w[(size_t)cc * E.ld + r] : E.w[(size_t)r * E.ld + cc]) : 0.f;
    }
    *reinterpret_cast<u32x4_t*>(dst + (size_t)q * 8) =
        (u32x4_t){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  }
}

// Reads `bytes` of read-only data into every XCD's L2: workgroup b takes slice b / 8 of grid / 8 slices (workgroups are dealt
// round-robin over the 8 XCDs, so the 32 workgroups of an XCD cover the whole range; placement is a speed matter only).
__global__ __launch_bounds__(256) void l2_warm_kernel(const u32x4_t* __restrict__ p, size_t n16, unsigned* sink) {
  const size_t slices = gridDim.x / 8, sl = blockIdx.x / 8;
  const size_t per = (n16 + slices - 1) / slices, beg = sl * per, end = beg + per < n16 ? beg + per : n16;
  unsigned acc = 0;
  for (size_t i = beg + threadIdx.x; i < end; i += 256) {
    const u32x4_t v = __builtin_nontemporal_load(p + i);
    acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x9e3779b9u && sink != nullptr) *sink = acc;     // keeps the loads alive; practically never true
}

bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int mvf_head_chain_debug(int bits) {
  g_chain_dbg = bits;
  return MVF_OK;
}
extern "C" int mvf_head_chain_debug_stamps(long long* stamps16) {
  g_chain_stamps = stamps16;
  return MVF_OK;
}

extern "C" int mvf_head_l2_warm(const void* p, size_t bytes, hipStream_t st) {
  MVF_CHECK_ARG(p && bytes >= 16 && al16(p));
  hipLaunchKernelGGL(l2_warm_kernel, dim3(256), dim3(256), 0, st, reinterpret_cast<const u32x4_t*>(p), bytes / 16, (unsigned*)nullptr);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_head_pack_weights(const MvfPackEntry* entries_host, int n, hipStream_t st) {
  MVF_CHECK_ARG(entries_host && n > 0 && n <= PACK_MAX);
  PackArgs a{};
  size_t pieces = 0;
  for (int i = 0; i < n; ++i) {
    const MvfPackEntry& s = entries_host[i];
    MVF_CHECK_ARG(s.w && s.N > 0 && s.K > 0 && s.ld >= s.K && (s.w16 || s.w16t) && al16(s.w16) && al16(s.w16t));
    PackEnt& e = a.e[i];
    e.w = s.w; e.ld = s.ld; e.N = s.N; e.K = s.K; e.w16 = (bf16_t*)s.w16; e.w16t = (bf16_t*)s.w16t;
    e.piece0 = (unsigned)pieces;
    e.pieces16 = (unsigned)(fm_elems(s.N, s.K) / 8);
    pieces += fm_elems(s.N, s.K) / 8 + fm_elems(s.K, s.N) / 8;
  }
  MVF_CHECK_ARG(pieces < (1ull << 31));
  a.n = n;
  const int grid = (int)std::min<size_t>((pieces + NTH2 - 1) / NTH2, 2048);
  hipLaunchKernelGGL(head_pack_kernel, dim3(grid), dim3(NTH2), 0, st, a, (unsigned)pieces);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" size_t mvf_head_pack_elems(int N, int K, int transposed) {
  return transposed ? fm_elems(K, N) : fm_elems(N, K);
}

extern "C" int mvf_enc_layer_fwd(const MvfEncFwd* s, hipStream_t st) {
  MVF_CHECK_ARG(s && s->M > 0 && s->D > 0 && s->D % 256 == 0 && s->DFF % 256 == 0 && s->D <= 512 && s->x_in);
  MVF_CHECK_ARG(s->o != nullptr || s->wqkv != nullptr);
  MVF_CHECK_ARG(s->Mp % 128 == 0 && s->Mp >= s->M);
  if (s->o) MVF_CHECK_ARG(s->wo && s->w1 && s->w2 && s->bo && s->b1 && s->b2 && s->ln1_g && s->ln1_b && al16(s->o) && al16(s->x_in));
  if (s->wqkv) MVF_CHECK_ARG(s->bqkv && s->ln0_g && s->ln0_b && s->qkv && al16(s->qkv));
  const EncLds L = enc_lds(s->D, s->DFF, false);
  if (L.total > 160 * 1024) return MVF_ERR_UNSUPPORTED;
  EncFwdK k{};
  k.M = s->M; k.D = s->D; k.DFF = s->DFF; k.Mp = s->Mp; k.eps = s->ln_eps; k.dbg = g_chain_dbg; k.stamps = g_chain_stamps;
  k.o = s->o; k.x_in = s->x_in;
  k.wo = (const bf16_t*)s->wo; k.w1 = (const bf16_t*)s->w1; k.w2 = (const bf16_t*)s->w2; k.wqkv = (const bf16_t*)s->wqkv;
  k.bo = s->bo; k.b1 = s->b1; k.b2 = s->b2; k.bqkv = s->bqkv; k.g1 = s->ln1_g; k.be1 = s->ln1_b; k.g0 = s->ln0_g; k.be0 = s->ln0_b;
  k.da = make_drop(s->drop_attn); k.df = make_drop(s->drop_ffn);
  k.x1 = s->x1; k.mean1 = s->mean1; k.rstd1 = s->rstd1; k.x2 = s->x2; k.qkv = s->qkv; k.mean0 = s->mean0; k.rstd0 = s->rstd0;
  k.a = (bf16_t*)s->a; k.oT = (bf16_t*)s->oT; k.h1T = (bf16_t*)s->h1T; k.aT = (bf16_t*)s->aT; k.h0T = (bf16_t*)s->h0T;
  static uint64_t attr = 0;
  if (mvf_ensure_lds(reinterpret_cast<const void*>(enc_fwd_kernel), 160 * 1024, attr) != MVF_OK) return MVF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(enc_fwd_kernel, dim3(ceil_div(s->M, TM)), dim3(NTH), L.total, st, k);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_enc_layer_bwd(const MvfEncBwd* s, hipStream_t st) {
  MVF_CHECK_ARG(s && s->M > 0 && s->D > 0 && s->D % 256 == 0 && s->DFF % 256 == 0 && s->D <= 512 && s->dres && al16(s->dres));
  MVF_CHECK_ARG(s->dqkv != nullptr || s->w2T != nullptr);
  MVF_CHECK_ARG(s->Mp % 128 == 0 && s->Mp >= s->M);
  if (s->dqkv) MVF_CHECK_ARG(s->wqkvT && s->x_in && s->mean0 && s->rstd0 && s->ln0_g && al16(s->dqkv) && (s->dx_out || s->w2T) &&
                             ((s->dln0_g == nullptr) == (s->dln0_b == nullptr)));
  if (s->w2T) MVF_CHECK_ARG(s->w1T && s->woT && s->a && s->x1 && s->mean1 && s->rstd1 && s->ln1_g && s->dx1_out && s->d_o &&
                            al16(s->d_o) && al16(s->dx1_out) && ((s->dln1_g == nullptr) == (s->dln1_b == nullptr)));
  const EncLds L = enc_lds(s->D, s->DFF, true);
  if (L.total > 160 * 1024) return MVF_ERR_UNSUPPORTED;
  EncBwdK k{};
  k.M = s->M; k.D = s->D; k.DFF = s->DFF; k.Mp = s->Mp;
  k.dqkv = s->dqkv; k.x_in = s->x_in; k.mean0 = s->mean0; k.rstd0 = s->rstd0; k.g0 = s->ln0_g; k.dres = s->dres;
  k.wqkvT = (const bf16_t*)s->wqkvT; k.w2T = (const bf16_t*)s->w2T; k.w1T = (const bf16_t*)s->w1T; k.woT = (const bf16_t*)s->woT;
  k.a = (const bf16_t*)s->a;
  k.dg0 = s->dln0_g; k.db0 = s->dln0_b; k.dx_out = s->dx_out;
  k.dqkvT = (bf16_t*)s->dqkvT; k.g2T = (bf16_t*)s->g2T; k.duT = (bf16_t*)s->duT; k.goT = (bf16_t*)s->goT;
  k.df = make_drop(s->drop_ffn); k.da = make_drop(s->drop_attn);
  k.x1 = s->x1; k.mean1 = s->mean1; k.rstd1 = s->rstd1; k.g1 = s->ln1_g; k.dg1 = s->dln1_g; k.db1 = s->dln1_b;
  k.dx1_out = s->dx1_out; k.d_o = s->d_o;
  static uint64_t attr = 0;
  if (mvf_ensure_lds(reinterpret_cast<const void*>(enc_bwd_kernel), 160 * 1024, attr) != MVF_OK) return MVF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(enc_bwd_kernel, dim3(ceil_div(s->M, TM)), dim3(NTH), L.total, st, k);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_head_dw(const MvfDwProblem* probs_host, int n, int Mp, int accumulate, hipStream_t st) {
  MVF_CHECK_ARG(probs_host && n > 0 && n <= DW_MAX && Mp > 0 && Mp % 128 == 0);
  DwArgs a{};
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    const MvfDwProblem& s = probs_host[i];
    MVF_CHECK_ARG(s.gT && s.xT && s.dw && s.N > 0 && s.K > 0 && s.lddw >= s.K && al16(s.gT) && al16(s.xT));
    DwProb& p = a.p[i];
    p.gT = (const bf16_t*)s.gT; p.xT = (const bf16_t*)s.xT; p.dw = s.dw; p.lddw = s.lddw; p.db = s.db; p.N = s.N; p.K = s.K;
    p.tiles_k = ceil_div(s.K, 64); p.tile0 = tiles;
    tiles += ceil_div(s.N, 64) * p.tiles_k;
  }
  a.n = n; a.Mp = Mp; a.accumulate = accumulate;
  hipLaunchKernelGGL(head_dw_kernel, dim3(tiles), dim3(NTH2), 0, st, a);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}
