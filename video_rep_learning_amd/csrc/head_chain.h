// Building blocks of the row-chain kernels of the trainable head (head_chain.hip: temporal encoder; head_rowlin.hip: FC stack, embedding
// layer, projection head): 32-row panels in LDS, 512-thread workgroups, GEMM stages on v_mfma_f32_16x16x32_bf16 with fragment-major
// ("FM") bf16 weights streamed straight from L2 / HBM.  See head_chain.hip for the design notes.
#pragma once
#include "common.h"
#include "mvf_hip_internal.h"

namespace chain {

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

inline Drop make_drop(const MvfDrop& d) {
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

// weight-fragment load: plain (default cache policy).  Every workgroup of a launch streams the same weights (1.5 MB per encoder layer),
// three workgroups per XCD: with the non-temporal hint the fragments were not kept in the L2 for the neighbours and every wave paid the
// Infinity-Cache latency -- a 512 x 512 GEMM stage 14.0 us against 7.0 with plain loads, the encoder-layer launches 46 / 52 -> 33 / 41 us
// in the serial step, the pipelined step 10.747 -> 10.726 ms (three alternating runs each, profiles/r05/lib_ab_ntoff_chain.txt).
// MVF_NT_WEIGHTS (build flag): the non-temporal form, for A/B measurements.
#ifdef MVF_NT_WEIGHTS
#define WLOAD(p) __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(p))
#else
#define WLOAD(p) (*reinterpret_cast<const bf16x8_t*>(p))
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
// F16: both operands IEEE fp16 (the FORWARD GEMMs of MI355X.HEAD_DTYPE fp16; the backward GEMMs stay bf16: head_chain.hip header)
template <int NT, int PF, bool F16 = false, typename Pre, typename Epi>
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
          acc[0][nt] = mfma16x16x32<F16>(bq[p][nt], a0, acc[0][nt]);
          acc[1][nt] = mfma16x16x32<F16>(bq[p][nt], a1, acc[1][nt]);
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

// one fp32 -> the panel's 16-bit format (bf16, or IEEE fp16 in the forward kernels of the fp16 head)
template <bool F16>
__device__ __forceinline__ bf16_t f32_to_16(float v) {
  if constexpr (F16) return f32_to_f16(v);
  else return f32_to_bf16(v);
}
// a 16-bit panel value as the bf16 the weight-gradient launch multiplies (identity for a bf16 panel)
template <bool F16>
__device__ __forceinline__ unsigned to_bf16_bits(unsigned h) {
  if constexpr (F16) return f32_to_bf16(f16_to_f32((uint16_t)h));
  else return h;
}

// ---- panel helpers (all 256 threads; the caller places the barriers) ----

// fp32 global rows [m0, m0 + 32) x [0, C) (row stride ld) -> bf16 panel; rows >= M read as 0.  C % 4 == 0.
// (batches of 8 loads per thread are issued before the first is used: a run-time loop of load -> convert -> store pays one
// memory round trip per iteration)
template <bool F16 = false>
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
      if (i < total) *reinterpret_cast<u32x2_t*>(P + r * ldp + 4 * q) = (u32x2_t){pack16x2<F16>(v[u][0], v[u][1]), pack16x2<F16>(v[u][2], v[u][3])};
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
// F16: the panel holds fp16 (forward of the fp16 head); the image is written as bf16, the weight-gradient launch's operand format
template <bool F16 = false>
__device__ __forceinline__ void store_T(const bf16_t* P, int ldp, int C, bf16_t* __restrict__ dst, int Mp, int m0, int M) {
  if (dst == nullptr) return;
  const int valid = min(TM, M - m0);
  const int msteps = Mp >> 5, ms = m0 >> 5;
  for (int cc = threadIdx.x; cc < C; cc += NTH) {
    unsigned w[TM / 2];
#pragma unroll
    for (int r = 0; r < TM; r += 2) {
      const unsigned lo = r < valid ? to_bf16_bits<F16>(P[r * ldp + cc]) : 0u, hi = r + 1 < valid ? to_bf16_bits<F16>(P[(r + 1) * ldp + cc]) : 0u;
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
template <bool F16 = false>
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
    if (q < nc) H[r * ldh + sub + 16 * q] = f32_to_16<F16>((xv[q] - mu) * rs * gv[q] + bv[q]);
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


inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace chain
