// Learned-query spatial token pooling (LSTP cross-attention), forward and backward.
// Reference: LSTPCrossAtt.forward / LearnableTokenPooling.forward, CARL_MVF/models/mvformer.py:243-266, 352-414
// with `attention` from models/utils.py:11-44.
//
// The reference projects K = x W_K^T + b_K and V = x W_V^T + b_V for all F*N tokens (2 x 0.35 GFLOP/frame,
// K and V round-trip HBM) and then attends with nq (= 3) queries.  Because softmax rows sum to one and the
// queries do not depend on the token, the same numbers are
//     scores[f,n,j] = x[f,n,:] . wq[f,j,:]  (+ q_j.b_K, constant over n -> cancels in softmax)
//     P = softmax_n(scores / sqrt(d)),   pooled[f,j,:] = sum_n P[f,j,n] x[f,n,:],   out = pooled W_V^T + b_V
// with wq = q W_K (tiny GEMM).  What is left on the big [F, N, C] tap tensor is two streaming passes
// (scores, weighted sum) forward and two backward: HBM-bound, no GEMM-shaped work.  These kernels ARE those
// passes; the tap tensors are read as separate [F*N, D] buffers per tapped block (no channel concat copy).
//
// Layouts: taps[t] [F*N, D] (bf16 or f32);  vec [G, nq, C] with C = n_taps*D, G = 1 (shared queries) or one
// per frame stored as [Bc, nq, T, C];  scores / dP [F*N, nq];  P / dS [F, nq, N];  pooled [Bc, nq, T, C]
// (rows already in the (clip, entity, frame) order the temporal encoder wants).
#include "common.h"
#include "mvf_hip_internal.h"

namespace {

constexpr int MAXQ = 8;
constexpr int MAXTAPS = 8;

struct PoolArgs {
  const void* taps[MAXTAPS];
  int n_taps, D, F, N, T, nq;
  const float* vec; int per_frame;   // scores: the query-side vectors
  float* scores;                      // [F*N, nq]
  const float* w;                     // wsum: weights [F, nq, N]
  float* out;                         // wsum: [Bc, nq, T, C]
};

template <typename T> struct Ld16;
template <> struct Ld16<float> {
  static constexpr int E = 4;
  static __device__ __forceinline__ void ld(const float* p, float (&v)[8]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
};
template <> struct Ld16<bf16_t> {
  static constexpr int E = 8;
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
  }
};

// ---- pass 1: scores[f*N+n, j] = sum_c x[f,n,c] * vec[f|0, j, c] --------------------------------------------
// grid (F, ysplit); a wave walks tokens of its frame two at a time; vec of the frame sits in LDS ([nq][C] f32).
template <typename T>
__global__ __launch_bounds__(256) void lstp_scores_kernel(PoolArgs a) {
  extern __shared__ __attribute__((aligned(16))) float svec[];  // [nq][C]
  constexpr int E = Ld16<T>::E;
  const int C = a.n_taps * a.D;
  const int f = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    const int b = f / a.T, t = f % a.T;
    for (int i = threadIdx.x; i < a.nq * C; i += 256) {
      const int j = i / C, c = i % C;
      svec[i] = a.per_frame ? a.vec[(((size_t)b * a.nq + j) * a.T + t) * C + c] : a.vec[(size_t)j * C + c];
    }
  }
  __syncthreads();
  const int wid = blockIdx.y * 4 + wave, nw = gridDim.y * 4;
  for (int n0 = wid * 2; n0 < a.N; n0 += nw * 2) {
    const int n1 = min(n0 + 1, a.N - 1);
    float acc0[MAXQ], acc1[MAXQ];
#pragma unroll
    for (int j = 0; j < MAXQ; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
    for (int tp = 0; tp < a.n_taps; ++tp) {
      const T* x0 = reinterpret_cast<const T*>(a.taps[tp]) + ((size_t)f * a.N + n0) * a.D;
      const T* x1 = reinterpret_cast<const T*>(a.taps[tp]) + ((size_t)f * a.N + n1) * a.D;
      for (int c = lane * E; c < a.D; c += 64 * E) {
        float v0[8], v1[8];
        Ld16<T>::ld(x0 + c, v0);
        Ld16<T>::ld(x1 + c, v1);
#pragma unroll
        for (int j = 0; j < MAXQ; ++j) {
          if (j < a.nq) {
            const float* wv = svec + j * C + tp * a.D + c;
#pragma unroll
            for (int e = 0; e < E; e += 4) {
              const float4 w4 = *reinterpret_cast<const float4*>(wv + e);
              acc0[j] += v0[e] * w4.x + v0[e + 1] * w4.y + v0[e + 2] * w4.z + v0[e + 3] * w4.w;
              acc1[j] += v1[e] * w4.x + v1[e + 1] * w4.y + v1[e + 2] * w4.z + v1[e + 3] * w4.w;
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < MAXQ; ++j) {
      if (j < a.nq) {
        const float s0 = wave_sum(acc0[j]), s1 = wave_sum(acc1[j]);
        if (lane == 0) {
          a.scores[((size_t)f * a.N + n0) * a.nq + j] = s0;
          if (n0 + 1 < a.N) a.scores[((size_t)f * a.N + n0 + 1) * a.nq + j] = s1;
        }
      }
    }
  }
}

// ---- pass 2: out[b, j, t, tap*D + c] = sum_n w[f, j, n] * x[f, n, c] -----------------------------------------
// grid (F, n_taps); thread = one 16-byte channel chunk x one token phase; phases combined through LDS.
template <typename T>
__global__ __launch_bounds__(256) void lstp_wsum_kernel(PoolArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [nq][N] weights, then reduction scratch
  constexpr int E = Ld16<T>::E;
  const int f = blockIdx.x, tp = blockIdx.y;
  const int ncols = a.D / E;
  const int phases = blockDim.x / ncols;
  const int col = threadIdx.x % ncols, ph = threadIdx.x / ncols;
  for (int i = threadIdx.x; i < a.nq * a.N; i += blockDim.x) sw[i] = a.w[(size_t)f * a.nq * a.N + i];
  __syncthreads();
  float acc[MAXQ][E];
#pragma unroll
  for (int j = 0; j < MAXQ; ++j)
#pragma unroll
    for (int e = 0; e < E; ++e) acc[j][e] = 0.f;
  const T* x = reinterpret_cast<const T*>(a.taps[tp]) + (size_t)f * a.N * a.D + col * E;
  if (ph < phases) {
    for (int n = ph; n < a.N; n += phases) {
      float v[8];
      Ld16<T>::ld(x + (size_t)n * a.D, v);
#pragma unroll
      for (int j = 0; j < MAXQ; ++j) {
        if (j < a.nq) {
          const float wj = sw[j * a.N + n];
#pragma unroll
          for (int e = 0; e < E; ++e) acc[j][e] += wj * v[e];
        }
      }
    }
  }
  __syncthreads();
  float* red = sw + a.nq * a.N;  // [phases][nq][D]
  if (ph < phases) {
#pragma unroll
    for (int j = 0; j < MAXQ; ++j)
      if (j < a.nq)
#pragma unroll
        for (int e = 0; e < E; ++e) red[((size_t)ph * a.nq + j) * a.D + col * E + e] = acc[j][e];
  }
  __syncthreads();
  const int C = a.n_taps * a.D;
  const int b = f / a.T, t = f % a.T;
  for (int i = threadIdx.x; i < a.nq * a.D; i += blockDim.x) {
    const int j = i / a.D, c = i % a.D;
    float s = 0.f;
    for (int p = 0; p < phases; ++p) s += red[((size_t)p * a.nq + j) * a.D + c];
    a.out[(((size_t)b * a.nq + j) * a.T + t) * C + tp * a.D + c] = s;
  }
}

// ---- softmax over tokens (per frame, per query) and its backward ------------------------------------------------
// scores [F*N, nq] -> P [F, nq, N] = softmax_n(scores * inv_sqrt_d); disjoint: Pm = P * [j == argmax_j P[:, n]]
__global__ __launch_bounds__(256) void lstp_softmax_kernel(const float* __restrict__ scores, float* __restrict__ P,
                                                           float* __restrict__ Pm, float* __restrict__ rowsum, int N,
                                                           int nq, float inv_sqrt_d, int disjoint) {
  extern __shared__ float sp[];  // [nq][N]
  const int f = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < nq; j += 4) {
    float mx = -1e30f;
    for (int n = lane; n < N; n += 64) mx = fmaxf(mx, scores[((size_t)f * N + n) * nq + j] * inv_sqrt_d);
    mx = wave_max(mx);
    float s = 0.f;
    for (int n = lane; n < N; n += 64) {
      const float e = expf(scores[((size_t)f * N + n) * nq + j] * inv_sqrt_d - mx);
      sp[j * N + n] = e;
      s += e;
    }
    s = wave_sum(s);
    const float inv = 1.f / s;
    for (int n = lane; n < N; n += 64) {
      const float p = sp[j * N + n] * inv;
      sp[j * N + n] = p;
      P[((size_t)f * nq + j) * N + n] = p;
    }
  }
  if (!disjoint) return;
  __syncthreads();
  // torch.argmax picks the FIRST maximal index
  for (int n = threadIdx.x; n < N; n += 256) {
    int am = 0;
    float m = sp[n];
    for (int j = 1; j < nq; ++j)
      if (sp[j * N + n] > m) { m = sp[j * N + n]; am = j; }
    for (int j = 0; j < nq; ++j) {
      const float v = j == am ? sp[j * N + n] : 0.f;
      Pm[((size_t)f * nq + j) * N + n] = v;
      sp[j * N + n] = v;
    }
  }
  __syncthreads();
  for (int j = wave; j < nq; j += 4) {
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += sp[j * N + n];
    s = wave_sum(s);
    if (lane == 0) rowsum[(size_t)f * nq + j] = s;
  }
}

// dS[f, j, n] = inv_sqrt_d * P * (g - sum_n' P g),  g = dP[f*N+n, j] * [Pm != 0 if disjoint]
// (+ drow[f, j] added to g where selected: gradient of rowsum(Pm) used by the b_V term)
__global__ __launch_bounds__(256) void lstp_softmax_bwd_kernel(const float* __restrict__ P, const float* __restrict__ Pm,
                                                               const float* __restrict__ dP, const float* __restrict__ drow,
                                                               float* __restrict__ dS, int N, int nq, float inv_sqrt_d) {
  const int f = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < nq; j += 4) {
    const float* p = P + ((size_t)f * nq + j) * N;
    const float* pm = Pm ? Pm + ((size_t)f * nq + j) * N : nullptr;
    const float dr = drow ? drow[(size_t)f * nq + j] : 0.f;
    float s = 0.f;
    for (int n = lane; n < N; n += 64) {
      float g = dP[((size_t)f * N + n) * nq + j] + dr;
      if (pm && pm[n] == 0.f) g = 0.f;
      s += p[n] * g;
    }
    s = wave_sum(s);
    for (int n = lane; n < N; n += 64) {
      float g = dP[((size_t)f * N + n) * nq + j] + dr;
      if (pm && pm[n] == 0.f) g = 0.f;
      dS[((size_t)f * nq + j) * N + n] = inv_sqrt_d * p[n] * (g - s);
    }
  }
}

// out[j, c] = sum_{b,t} G[b, j, t, c]     (query-vector gradient of the shared-query case)
__global__ __launch_bounds__(256) void lstp_reduce_frames_kernel(const float* __restrict__ G, float* __restrict__ out,
                                                                 int Bc, int nq, int T, int C) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, j = blockIdx.y;
  float s = 0.f;
  if (c < C)
    for (int r = rl; r < Bc * T; r += 4) {
      const int b = r / T, t = r % T;
      s += G[(((size_t)b * nq + j) * T + t) * C + c];
    }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) out[(size_t)j * C + c] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
}

// ------------------------------------------------------------------------------------------------------------------
// One-pass forms (static or per-frame queries, no SMART_DISJOINT, frozen taps): the tap tensors are read ONCE forward and
// ONCE backward instead of twice each.
//   forward : online softmax over a frame's tokens.  A wave owns tokens n = w, w + 8, ..; per token it has the whole
//             channel vector x_n in registers, takes the NQ scores s_j = x_n . vec_j (vec in LDS), updates its running
//             (m_j, l_j) and accumulates p_j x_n; the eight waves' partial results are merged in a fixed order through LDS.
//             Writes pooled [Bc, nq, T, C] and P [F, nq, N] (the normalised weights: LSTPCrossAtt.attn_matrix and the
//             backward's input).
//   backward: with g_jn = dpooled_j . x_n and gbar_j = sum_n P_jn g_jn, the raw-score gradient is dS_jn = c P_jn (g_jn - gbar_j)
//             (c = 1 / sqrt(d)), so the query-vector gradient  sum_n dS_jn x_n = c (sum_n P_jn g_jn x_n - gbar_j pooled_j)
//             needs ONE pass that accumulates A_j = sum_n P_jn g_jn x_n and gbar_j (pooled_j is the forward's output).
// Lane layout: 16-byte (bf16: 8 values, f32: 4 values x 2) channel chunks; chunk q of tap t belongs to lane q % 64, slot q / 64.
// ------------------------------------------------------------------------------------------------------------------
constexpr int FQ = 3;            // queries the one-pass kernels hold accumulators for
constexpr int FSLOT = 2;         // chunk slots per lane and tap: D <= 8 * 64 * FSLOT = 1024
constexpr int FTAPS = 3;
constexpr int FWAVES = 8;

struct FusedArgs {
  const void* taps[FTAPS];
  int n_taps, D, N, T, nq, per_frame;
  const float* vec;       // fwd: query-side vectors; bwd: dpooled [Bc, nq, T, C]
  float* P;               // [F, nq, N]  (fwd: out, bwd: in)
  float* pooled;          // [Bc, nq, T, C]  (fwd: out, bwd: in)
  float* G;               // bwd: [Bc, nq, T, C]
  float inv_sqrt_d;
};

template <typename T>
__device__ __forceinline__ void load_chunk(const T* p, float (&v)[8]);
template <>
__device__ __forceinline__ void load_chunk<bf16_t>(const bf16_t* p, float (&v)[8]) { Ld16<bf16_t>::ld(p, v); }
template <>
__device__ __forceinline__ void load_chunk<float>(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// dot products of one token's channel vector with the NQ vectors in LDS: d[j] = sum over this lane's chunks; the caller reduces
template <typename T>
__device__ __forceinline__ void token_dots(const FusedArgs& a, const float* svec, int C, size_t row, int lane, float (&d)[FQ]) {
#pragma unroll
  for (int j = 0; j < FQ; ++j) d[j] = 0.f;
  const int nch = a.D >> 3;
#pragma unroll
  for (int tp = 0; tp < FTAPS; ++tp) {
    if (tp >= a.n_taps) break;
    const T* x = reinterpret_cast<const T*>(a.taps[tp]) + row * a.D;
#pragma unroll
    for (int sl = 0; sl < FSLOT; ++sl) {
      const int q = lane + 64 * sl;
      if (q < nch) {
        float v[8];
        load_chunk<T>(x + q * 8, v);
#pragma unroll
        for (int j = 0; j < FQ; ++j) {
          if (j < a.nq) {
            const float4 w0 = *reinterpret_cast<const float4*>(svec + j * C + tp * a.D + q * 8);
            const float4 w1 = *reinterpret_cast<const float4*>(svec + j * C + tp * a.D + q * 8 + 4);
            d[j] += v[0] * w0.x + v[1] * w0.y + v[2] * w0.z + v[3] * w0.w + v[4] * w1.x + v[5] * w1.y + v[6] * w1.z + v[7] * w1.w;
          }
        }
      }
    }
  }
}

// acc[j][tp][sl][:] = acc * scale[j] + wgt[j] * x  for this lane's chunks of one token
template <typename T>
__device__ __forceinline__ void token_axpy(const FusedArgs& a, size_t row, int lane, const float (&wgt)[FQ],
                                           float (&acc)[FQ][FTAPS][FSLOT][8]) {
  const int nch = a.D >> 3;
#pragma unroll
  for (int tp = 0; tp < FTAPS; ++tp) {
    if (tp >= a.n_taps) break;
    const T* x = reinterpret_cast<const T*>(a.taps[tp]) + row * a.D;
#pragma unroll
    for (int sl = 0; sl < FSLOT; ++sl) {
      const int q = lane + 64 * sl;
      if (q < nch) {
        float v[8];
        load_chunk<T>(x + q * 8, v);        // second read of the token: served by L1 / L2 (4.6 KB read microseconds earlier)
#pragma unroll
        for (int j = 0; j < FQ; ++j)
          if (j < a.nq)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j][tp][sl][e] = fmaf(wgt[j], v[e], acc[j][tp][sl][e]);
      }
    }
  }
}

// the eight waves' accumulators -> red[nq][C] in LDS, each scaled by its wave's factor, added in wave order (deterministic)
__device__ __forceinline__ void merge_waves(const FusedArgs& a, float* red, int C, int lane, int wave, const float (&f)[FQ],
                                            const float (&acc)[FQ][FTAPS][FSLOT][8]) {
  const int nch = a.D >> 3;
  for (int w = 0; w < FWAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < FQ; ++j) {
        if (j >= a.nq) break;
#pragma unroll
        for (int tp = 0; tp < FTAPS; ++tp) {
          if (tp >= a.n_taps) break;
#pragma unroll
          for (int sl = 0; sl < FSLOT; ++sl) {
            const int q = lane + 64 * sl;
            if (q < nch) {
              float* r = red + j * C + tp * a.D + q * 8;
#pragma unroll
              for (int e = 0; e < 8; ++e) r[e] = (w == 0 ? 0.f : r[e]) + f[j] * acc[j][tp][sl][e];
            }
          }
        }
      }
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(512) void lstp_fused_fwd_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = a.n_taps * a.D;
  float* svec = sm;                         // [nq][C]; reused as the merge buffer
  float* ssc = sm + a.nq * C;               // [nq][N] raw scores (scaled by 1 / sqrt(d))
  float* sst = ssc + a.nq * a.N;            // [FWAVES][nq][2] (m, l) per wave
  const int f = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = f / a.T, t = f % a.T;
  for (int i = threadIdx.x; i < a.nq * C; i += 512) {
    const int j = i / C, c = i % C;
    svec[i] = a.per_frame ? a.vec[(((size_t)b * a.nq + j) * a.T + t) * C + c] : a.vec[(size_t)j * C + c];
  }
  __syncthreads();
  float acc[FQ][FTAPS][FSLOT][8];
#pragma unroll
  for (int j = 0; j < FQ; ++j)
#pragma unroll
    for (int tp = 0; tp < FTAPS; ++tp)
#pragma unroll
      for (int sl = 0; sl < FSLOT; ++sl)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][tp][sl][e] = 0.f;
  float m[FQ], l[FQ];
#pragma unroll
  for (int j = 0; j < FQ; ++j) { m[j] = -1e30f; l[j] = 0.f; }
  for (int n = wave; n < a.N; n += FWAVES) {
    const size_t row = (size_t)f * a.N + n;
    float d[FQ], wgt[FQ];
    token_dots<T>(a, svec, C, row, lane, d);
    bool grow = false;
    float al[FQ];
#pragma unroll
    for (int j = 0; j < FQ; ++j) {
      al[j] = 1.f;
      wgt[j] = 0.f;
      if (j < a.nq) {
        const float s = wave_sum(d[j]) * a.inv_sqrt_d;
        if (lane == 0) ssc[j * a.N + n] = s;
        const float mn = fmaxf(m[j], s);
        al[j] = __expf(m[j] - mn);
        grow = grow || (mn > m[j]);
        wgt[j] = __expf(s - mn);
        l[j] = l[j] * al[j] + wgt[j];
        m[j] = mn;
      }
    }
    if (grow) {       // wave-uniform: a running maximum moved (rare after the first few tokens)
#pragma unroll
      for (int j = 0; j < FQ; ++j)
#pragma unroll
        for (int tp = 0; tp < FTAPS; ++tp)
#pragma unroll
          for (int sl = 0; sl < FSLOT; ++sl)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j][tp][sl][e] *= al[j];
    }
    token_axpy<T>(a, row, lane, wgt, acc);
  }
  if (lane == 0)
    for (int j = 0; j < a.nq; ++j) { sst[(wave * a.nq + j) * 2] = m[j]; sst[(wave * a.nq + j) * 2 + 1] = l[j]; }
  __syncthreads();        // every wave is done with svec; (m, l) of all waves and all raw scores are visible
  float fac[FQ], M[FQ], L[FQ];
#pragma unroll
  for (int j = 0; j < FQ; ++j) {
    fac[j] = 0.f; M[j] = 0.f; L[j] = 1.f;
    if (j < a.nq) {
      float mm = -1e30f;
      for (int w = 0; w < FWAVES; ++w) mm = fmaxf(mm, sst[(w * a.nq + j) * 2]);
      float ll = 0.f;
      for (int w = 0; w < FWAVES; ++w) ll += sst[(w * a.nq + j) * 2 + 1] * __expf(sst[(w * a.nq + j) * 2] - mm);
      M[j] = mm; L[j] = ll;
      fac[j] = __expf(m[j] - mm) / ll;
    }
  }
  merge_waves(a, svec, C, lane, wave, fac, acc);
  for (int i = threadIdx.x; i < a.nq * C; i += 512) {
    const int j = i / C, c = i % C;
    a.pooled[(((size_t)b * a.nq + j) * a.T + t) * C + c] = svec[i];
  }
  for (int i = threadIdx.x; i < a.nq * a.N; i += 512) {
    const int j = i / a.N;
    float Mj = M[0], Lj = L[0];
#pragma unroll
    for (int q = 1; q < FQ; ++q) if (j == q) { Mj = M[q]; Lj = L[q]; }
    a.P[(size_t)f * a.nq * a.N + i] = __expf(ssc[i] - Mj) / Lj;
  }
}

template <typename T>
__global__ __launch_bounds__(512) void lstp_fused_bwd_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = a.n_taps * a.D;
  float* sdp = sm;                          // [nq][C] dpooled of this frame; reused as the merge buffer
  float* sP = sm + a.nq * C;                // [nq][N]
  float* sgb = sP + a.nq * a.N;             // [FWAVES][nq] partial gbar
  const int f = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = f / a.T, t = f % a.T;
  for (int i = threadIdx.x; i < a.nq * C; i += 512) {
    const int j = i / C, c = i % C;
    sdp[i] = a.vec[(((size_t)b * a.nq + j) * a.T + t) * C + c];
  }
  for (int i = threadIdx.x; i < a.nq * a.N; i += 512) sP[i] = a.P[(size_t)f * a.nq * a.N + i];
  __syncthreads();
  float acc[FQ][FTAPS][FSLOT][8];
#pragma unroll
  for (int j = 0; j < FQ; ++j)
#pragma unroll
    for (int tp = 0; tp < FTAPS; ++tp)
#pragma unroll
      for (int sl = 0; sl < FSLOT; ++sl)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][tp][sl][e] = 0.f;
  float gb[FQ] = {0.f, 0.f, 0.f};
  for (int n = wave; n < a.N; n += FWAVES) {
    const size_t row = (size_t)f * a.N + n;
    float d[FQ], wgt[FQ];
    token_dots<T>(a, sdp, C, row, lane, d);
#pragma unroll
    for (int j = 0; j < FQ; ++j) {
      wgt[j] = 0.f;
      if (j < a.nq) {
        wgt[j] = sP[j * a.N + n] * wave_sum(d[j]);      // P_jn g_jn
        gb[j] += wgt[j];
      }
    }
    token_axpy<T>(a, row, lane, wgt, acc);
  }
  if (lane == 0)
    for (int j = 0; j < a.nq; ++j) sgb[wave * a.nq + j] = gb[j];
  __syncthreads();
  const float one[FQ] = {1.f, 1.f, 1.f};
  merge_waves(a, sdp, C, lane, wave, one, acc);
  for (int i = threadIdx.x; i < a.nq * C; i += 512) {
    const int j = i / C, c = i % C;
    float g = 0.f;
    for (int w = 0; w < FWAVES; ++w) g += sgb[w * a.nq + j];
    const size_t o = (((size_t)b * a.nq + j) * a.T + t) * C + c;
    a.G[o] = a.inv_sqrt_d * (sdp[i] - g * a.pooled[o]);
  }
}

int fused_fill(FusedArgs& a, const void* const* taps, int n_taps, int dtype, int D, int F, int N, int T, int nq) {
  MVF_CHECK_ARG(taps && F > 0 && N > 0 && T > 0 && F % T == 0 && (dtype == MVF_F32 || dtype == MVF_BF16));
  if (!(n_taps >= 1 && n_taps <= FTAPS && nq >= 1 && nq <= FQ && D % 8 == 0 && D <= 8 * 64 * FSLOT)) return MVF_ERR_UNSUPPORTED;
  const size_t lds = ((size_t)nq * n_taps * D + (size_t)nq * N + FWAVES * nq * 2) * sizeof(float);
  if (lds > 96 * 1024) return MVF_ERR_UNSUPPORTED;
  for (int i = 0; i < n_taps; ++i) {
    MVF_CHECK_ARG(taps[i] && ((uintptr_t)taps[i] & 15) == 0);
    a.taps[i] = taps[i];
  }
  a.n_taps = n_taps; a.D = D; a.N = N; a.T = T; a.nq = nq;
  return MVF_OK;
}

template <typename K>
int fused_launch(K kernel, const FusedArgs& a, int F, hipStream_t st) {
  const size_t lds = ((size_t)a.nq * a.n_taps * a.D + (size_t)a.nq * a.N + FWAVES * a.nq * 2) * sizeof(float);
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return MVF_ERR_ARG;
  hipLaunchKernelGGL(kernel, dim3(F), dim3(512), lds, st, a);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

int fill_args(PoolArgs& a, const void* const* taps, int n_taps, int dtype, int D, int F, int N, int T, int nq) {
  MVF_CHECK_ARG(taps && n_taps > 0 && n_taps <= MAXTAPS && nq > 0 && nq <= MAXQ && F > 0 && N > 0 && T > 0 && F % T == 0);
  MVF_CHECK_ARG(dtype == MVF_F32 || dtype == MVF_BF16);
  const int e = dtype == MVF_BF16 ? 8 : 4;
  MVF_CHECK_ARG(D % e == 0 && D / e <= 256);
  for (int i = 0; i < n_taps; ++i) {
    MVF_CHECK_ARG(taps[i] && ((uintptr_t)taps[i] & 15) == 0);
    a.taps[i] = taps[i];
  }
  a.n_taps = n_taps; a.D = D; a.F = F; a.N = N; a.T = T; a.nq = nq;
  return MVF_OK;
}

}  // namespace

extern "C" int mvf_lstp_scores(const void* const* taps, int n_taps, int dtype, int D, int F, int N, int T, int nq,
                               const float* vec, int per_frame, float* scores, hipStream_t st) {
  PoolArgs a{};
  int rc = fill_args(a, taps, n_taps, dtype, D, F, N, T, nq);
  if (rc != MVF_OK) return rc;
  MVF_CHECK_ARG(vec && scores);
  a.vec = vec; a.per_frame = per_frame; a.scores = scores;
  const size_t lds = (size_t)nq * n_taps * D * sizeof(float);
  MVF_CHECK_ARG(lds <= 64 * 1024);
  const int ysplit = F >= 512 ? 1 : (F >= 128 ? 4 : 8);
  dim3 grid(F, ysplit);
  if (dtype == MVF_BF16) hipLaunchKernelGGL(lstp_scores_kernel<bf16_t>, grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL(lstp_scores_kernel<float>, grid, dim3(256), lds, st, a);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_lstp_wsum(const void* const* taps, int n_taps, int dtype, int D, int F, int N, int T, int nq,
                             const float* w, float* out, hipStream_t st) {
  PoolArgs a{};
  int rc = fill_args(a, taps, n_taps, dtype, D, F, N, T, nq);
  if (rc != MVF_OK) return rc;
  MVF_CHECK_ARG(w && out);
  a.w = w; a.out = out;
  const int e = dtype == MVF_BF16 ? 8 : 4;
  const int ncols = D / e;
  const int phases = std::max(1, 256 / ncols);
  const int threads = ncols * phases;
  const size_t lds = ((size_t)nq * N + (size_t)phases * nq * D) * sizeof(float);
  MVF_CHECK_ARG(lds <= 64 * 1024);
  dim3 grid(F, n_taps);
  if (dtype == MVF_BF16) hipLaunchKernelGGL(lstp_wsum_kernel<bf16_t>, grid, dim3(threads), lds, st, a);
  else hipLaunchKernelGGL(lstp_wsum_kernel<float>, grid, dim3(threads), lds, st, a);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_lstp_softmax_fwd(const float* scores, float* P, float* Pm, float* rowsum, int F, int N, int nq,
                                    float inv_sqrt_d, int disjoint, hipStream_t st) {
  MVF_CHECK_ARG(scores && P && F > 0 && N > 0 && nq > 0 && nq <= MAXQ && (!disjoint || (Pm && rowsum)));
  hipLaunchKernelGGL(lstp_softmax_kernel, dim3(F), dim3(256), (size_t)nq * N * sizeof(float), st, scores, P, Pm, rowsum, N,
                     nq, inv_sqrt_d, disjoint);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_lstp_softmax_bwd(const float* P, const float* Pm, const float* dP, const float* drow, float* dS, int F,
                                    int N, int nq, float inv_sqrt_d, hipStream_t st) {
  MVF_CHECK_ARG(P && dP && dS && F > 0 && N > 0 && nq > 0 && nq <= MAXQ);
  hipLaunchKernelGGL(lstp_softmax_bwd_kernel, dim3(F), dim3(256), 0, st, P, Pm, dP, drow, dS, N, nq, inv_sqrt_d);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// (entry points of the one-pass kernels: end of file)
// ---- late fusion: AdaptiveMaxPool2d(1) / AdaptiveAvgPool2d(1) over a frame's tokens (models/transformer.py:258-262) ----
// out[f, tap*D + c] = max_n | mean_n x[tap][f*N+n, c]; grid (F, n_taps), a thread owns channels, tokens in sequence
template <typename T>
__global__ __launch_bounds__(256) void token_pool_kernel(PoolArgs a, int mode) {
  const int f = blockIdx.x, tap = blockIdx.y;
  const T* x = reinterpret_cast<const T*>(a.taps[tap]) + (size_t)f * a.N * a.D;
  for (int c = threadIdx.x; c < a.D; c += 256) {
    float acc = mode == 0 ? -3.402823466e38f : 0.0f;
    for (int n = 0; n < a.N; ++n) {
      float v;
      if constexpr (sizeof(T) == 2) v = __uint_as_float((uint32_t)x[(size_t)n * a.D + c] << 16);
      else v = x[(size_t)n * a.D + c];
      acc = mode == 0 ? fmaxf(acc, v) : acc + v;
    }
    a.out[(size_t)f * a.n_taps * a.D + tap * a.D + c] = mode == 0 ? acc : acc / (float)a.N;
  }
}

// ---- gradient w.r.t. the tokens (trainable backbone blocks only; the frozen path never needs it) ----------------
//   dx[f,n,c] = sum_j ( W[f,j,n] * dpooled[b,j,t,c]  +  dS[f,j,n] * vec[f|0, j, c] )        (f = b*T + t)
// W = the weights the forward summed with (P, or the masked Pm under SMART_DISJOINT), dS = d(raw scores) from
// mvf_lstp_softmax_bwd.  One thread per (token, 4 channels): 2*nq small-vector FMAs, writes fp32 [F*N, D] per tap.
struct DxArgs {
  float* dx[MAXTAPS];
  int n_taps, D, F, N, T, nq, per_frame;
  const float* w; const float* ds; const float* dpooled; const float* vec;
};

__global__ __launch_bounds__(256) void lstp_dx_kernel(DxArgs a) {
  const int C = a.n_taps * a.D;
  const int c4n = C >> 2;
  const int f = blockIdx.y;
  const int b = f / a.T, t = f - b * a.T;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.N * c4n; i += gridDim.x * 256) {
    const int n = i / c4n, c = (i - n * c4n) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < a.nq; ++j) {
      if (a.dpooled != nullptr) {     // value side (uniform branches: either term may be absent)
        const float wv = a.w[((size_t)f * a.nq + j) * a.N + n];
        const float4 dp = *reinterpret_cast<const float4*>(a.dpooled + (((size_t)b * a.nq + j) * a.T + t) * C + c);
        acc.x += wv * dp.x; acc.y += wv * dp.y; acc.z += wv * dp.z; acc.w += wv * dp.w;
      }
      if (a.ds != nullptr) {          // score side
        const float dv = a.ds[((size_t)f * a.nq + j) * a.N + n];
        const float4 vq = *reinterpret_cast<const float4*>(
            a.vec + (a.per_frame ? (((size_t)b * a.nq + j) * a.T + t) * C : (size_t)j * C) + c);
        acc.x += dv * vq.x; acc.y += dv * vq.y; acc.z += dv * vq.z; acc.w += dv * vq.w;
      }
    }
    const int tap = c / a.D, cc = c - tap * a.D;
    *reinterpret_cast<float4*>(a.dx[tap] + ((size_t)f * a.N + n) * a.D + cc) = acc;
  }
}

extern "C" int mvf_token_pool(const void* const* taps, int n_taps, int dtype, int D, int F, int N, int mode, float* out,
                              hipStream_t st) {
  PoolArgs a{};
  int rc = fill_args(a, taps, n_taps, dtype, D, F, N, 1, 1);
  if (rc != MVF_OK) return rc;
  MVF_CHECK_ARG(out && (mode == 0 || mode == 1));
  a.out = out;
  if (dtype == MVF_BF16) hipLaunchKernelGGL(token_pool_kernel<bf16_t>, dim3(F, n_taps), dim3(256), 0, st, a, mode);
  else hipLaunchKernelGGL(token_pool_kernel<float>, dim3(F, n_taps), dim3(256), 0, st, a, mode);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_lstp_dx(float* const* dx_host, int n_taps, int D, int F, int N, int T, int nq, const float* w,
                           const float* ds, const float* dpooled, const float* vec, int per_frame, hipStream_t st) {
  MVF_CHECK_ARG(dx_host && ((w && dpooled) || (ds && vec)) && (!dpooled || w) && (!ds || vec) && n_taps > 0 && n_taps <= MAXTAPS && D > 0 && D % 4 == 0 && F > 0 &&
                N > 0 && T > 0 && F % T == 0 && nq > 0 && nq <= MAXQ);
  DxArgs a{};
  for (int i = 0; i < n_taps; ++i) {
    MVF_CHECK_ARG(dx_host[i] != nullptr && ((uintptr_t)dx_host[i] & 15) == 0);
    a.dx[i] = dx_host[i];
  }
  a.n_taps = n_taps; a.D = D; a.F = F; a.N = N; a.T = T; a.nq = nq; a.per_frame = per_frame;
  a.w = w; a.ds = ds; a.dpooled = dpooled; a.vec = vec;
  const int work = N * (n_taps * D / 4);
  hipLaunchKernelGGL(lstp_dx_kernel, dim3(std::min(ceil_div(work, 256), 64), F), dim3(256), 0, st, a);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_lstp_reduce_frames(const float* G, float* out, int Bc, int nq, int T, int C, hipStream_t st) {
  MVF_CHECK_ARG(G && out && Bc > 0 && nq > 0 && T > 0 && C > 0);
  hipLaunchKernelGGL(lstp_reduce_frames_kernel, dim3(ceil_div(C, 64), nq), dim3(256), 0, st, G, out, Bc, nq, T, C);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// One-pass pooling forward (see the kernel comment): pooled [Bc, nq, T, C] and P [F, nq, N] in one read of the taps.
// MVF_ERR_UNSUPPORTED when the shape is outside the kernel's register budget (nq > 3, more than 3 taps, D > 1024): the caller
// then runs the three-launch form (scores / softmax / weighted sum).
extern "C" int mvf_lstp_fused_fwd(const void* const* taps, int n_taps, int dtype, int D, int F, int N, int T, int nq,
                                  const float* vec, int per_frame, float inv_sqrt_d, float* P, float* pooled,
                                  hipStream_t st) {
  FusedArgs a{};
  const int rc = fused_fill(a, taps, n_taps, dtype, D, F, N, T, nq);
  if (rc != MVF_OK) return rc;
  MVF_CHECK_ARG(vec && P && pooled);
  a.vec = vec; a.per_frame = per_frame; a.inv_sqrt_d = inv_sqrt_d; a.P = P; a.pooled = pooled;
  return dtype == MVF_BF16 ? fused_launch(lstp_fused_fwd_kernel<bf16_t>, a, F, st) : fused_launch(lstp_fused_fwd_kernel<float>, a, F, st);
}

// One-pass backward: G [Bc, nq, T, C] = d loss / d vec of every frame (rows (clip, query, frame)) from dpooled, the forward's
// P and pooled; shared queries: sum G over frames with mvf_lstp_reduce_frames.
extern "C" int mvf_lstp_fused_bwd(const void* const* taps, int n_taps, int dtype, int D, int F, int N, int T, int nq,
                                  const float* dpooled, const float* P, const float* pooled, float inv_sqrt_d, float* G,
                                  hipStream_t st) {
  FusedArgs a{};
  const int rc = fused_fill(a, taps, n_taps, dtype, D, F, N, T, nq);
  if (rc != MVF_OK) return rc;
  MVF_CHECK_ARG(dpooled && P && pooled && G);
  a.vec = dpooled; a.per_frame = 1; a.inv_sqrt_d = inv_sqrt_d; a.P = const_cast<float*>(P); a.pooled = const_cast<float*>(pooled);
  a.G = G;
  return dtype == MVF_BF16 ? fused_launch(lstp_fused_bwd_kernel<bf16_t>, a, F, st) : fused_launch(lstp_fused_bwd_kernel<float>, a, F, st);
}
