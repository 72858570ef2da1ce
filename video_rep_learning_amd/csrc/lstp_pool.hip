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
#include <cstdlib>

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
// workgroup = 64 columns x 16 row slices (rows r = s, s + 16, ..); a thread's loads go out in batches of 8 before the first is used
// (as a run-time loop of load -> add, 64 rows per thread were 64 dependent round trips: 24 us for 7 MB); fixed order, no atomics
constexpr int RF_SL = 16;
__global__ __launch_bounds__(64 * RF_SL) void lstp_reduce_frames_kernel(const float* __restrict__ G, float* __restrict__ out,
                                                                        int Bc, int nq, int T, int C) {
  __shared__ float red[RF_SL][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, j = blockIdx.y, R = Bc * T;
  float s = 0.f;
  if (c < C)
    for (int r0 = rl; r0 < R; r0 += RF_SL * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = min(r0 + u * RF_SL, R - 1), b = r / T, t = r - b * T;
        v[u] = G[(((size_t)b * nq + j) * T + t) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + u * RF_SL < R) s += v[u];
    }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < RF_SL; ++w) a += red[w][cl];
    out[(size_t)j * C + c] = a;
  }
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
// Everything a lane touches is fixed at compile time -- NQ queries, NT taps, NG groups of 4 channels per lane and tap
// (D = 256 NG: 768 -> 3, 1024 -> 4; group q of a tap belongs to lane q % 64) -- so the token loop is straight-line code:
// with run-time trip counts hipcc branched around every load and drained the queue (vmcnt(0)) before each use, which undid the
// register double buffering (the next token's loads in flight while this one is reduced: 2 tokens x 8 waves = 74 KB per CU).
// ------------------------------------------------------------------------------------------------------------------
constexpr int FWAVES = 8;

struct FusedArgs {
  const void* taps[MAXTAPS];
  int D, N, T, per_frame;
  const float* vec;       // fwd: query-side vectors; bwd: dpooled [Bc, nq, T, C]
  float* P;               // [F, nq, N]  (fwd: out, bwd: in)
  float* pooled;          // [Bc, nq, T, C]  (fwd: out, bwd: in)
  float* G;               // bwd: [Bc, nq, T, C]
  float inv_sqrt_d;
};

template <typename T> struct Raw4;
template <> struct Raw4<bf16_t> {
  uint2 r;
  __device__ __forceinline__ void ld(const bf16_t* p) { r = *reinterpret_cast<const uint2*>(p); }
  __device__ __forceinline__ void get(float (&v)[4]) const {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  }
};
template <> struct Raw4<float> {
  float4 r;
  __device__ __forceinline__ void ld(const float* p) { r = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void get(float (&v)[4]) const { v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }
};

template <typename T, int NT, int NG>
struct Token {
  Raw4<T> g[NT][NG];
  __device__ __forceinline__ void load(const FusedArgs& a, size_t row, int lane) {
#pragma unroll
    for (int tp = 0; tp < NT; ++tp) {
      const T* x = reinterpret_cast<const T*>(a.taps[tp]) + row * a.D + lane * 4;
#pragma unroll
      for (int q = 0; q < NG; ++q) g[tp][q].ld(x + q * 256);
    }
  }
};

// d[j] = this lane's share of x . vec_j (vec in LDS, [NQ][C]); the caller reduces over the wave
template <typename T, int NQ, int NT, int NG>
__device__ __forceinline__ void token_dots(const Token<T, NT, NG>& tk, const float* svec, int D, int lane, float (&d)[NQ]) {
  const int C = NT * D;
#pragma unroll
  for (int j = 0; j < NQ; ++j) d[j] = 0.f;
#pragma unroll
  for (int tp = 0; tp < NT; ++tp) {
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      float v[4];
      tk.g[tp][q].get(v);
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const float4 w = *reinterpret_cast<const float4*>(svec + j * C + tp * D + q * 256 + lane * 4);
        d[j] = fmaf(v[0], w.x, fmaf(v[1], w.y, fmaf(v[2], w.z, fmaf(v[3], w.w, d[j]))));
      }
    }
    // one tap's LDS reads at a time: hoisted together, the NQ x NT x NG float4 operands alone are 100+ registers
    if constexpr (NQ * NT * NG > 12) __builtin_amdgcn_sched_barrier(0);
  }
}

// acc = acc * scale + wgt * x (SCALE: the online softmax's rescaling of the running sums, 1 almost always -- applied
// unconditionally: a wave-uniform branch around it made hipcc keep a second copy of the 108 accumulators alive and spill)
template <typename T, int NQ, int NT, int NG, bool SCALE>
__device__ __forceinline__ void token_axpy(const Token<T, NT, NG>& tk, const float (&wgt)[NQ], const float (&scale)[NQ],
                                           float (&acc)[NQ][NT][NG][4]) {
#pragma unroll
  for (int tp = 0; tp < NT; ++tp) {
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      float v[4];
      tk.g[tp][q].get(v);
#pragma unroll
      for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[j][tp][q][e] = SCALE ? fmaf(acc[j][tp][q][e], scale[j], wgt[j] * v[e]) : fmaf(wgt[j], v[e], acc[j][tp][q][e]);
    }
    if constexpr (SCALE && NQ * NT * NG > 12) __builtin_amdgcn_sched_barrier(0);     // (keeps the products of one tap together)
  }
}

// the eight waves' accumulators -> red[NQ][C] in LDS, each scaled by its wave's factor, added in wave order (deterministic)
template <int NQ, int NT, int NG>
__device__ __forceinline__ void merge_waves(float* red, int D, int lane, int wave, const float (&f)[NQ],
                                            const float (&acc)[NQ][NT][NG][4]) {
  const int C = NT * D;
  for (int w = 0; w < FWAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int tp = 0; tp < NT; ++tp)
#pragma unroll
          for (int q = 0; q < NG; ++q) {
            float4* r = reinterpret_cast<float4*>(red + j * C + tp * D + q * 256 + lane * 4);
            float4 o = w == 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : *r;
            o.x = fmaf(f[j], acc[j][tp][q][0], o.x); o.y = fmaf(f[j], acc[j][tp][q][1], o.y);
            o.z = fmaf(f[j], acc[j][tp][q][2], o.z); o.w = fmaf(f[j], acc[j][tp][q][3], o.w);
            *r = o;
          }
    }
    __syncthreads();
  }
}

template <typename T, int NQ, int NT, int NG>
__global__ __launch_bounds__(512) void lstp_fused_fwd_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = a.D, C = NT * D;
  float* svec = sm;                         // [NQ][C]; reused as the merge buffer
  float* ssc = sm + NQ * C;                 // [NQ][N] raw scores (scaled by 1 / sqrt(d))
  float* sst = ssc + NQ * a.N;              // [FWAVES][NQ][2] (m, l) per wave
  const int f = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = f / a.T, t = f % a.T;
  for (int i = threadIdx.x; i < NQ * C; i += 512) {
    const int j = i / C, c = i % C;
    svec[i] = a.per_frame ? a.vec[(((size_t)b * NQ + j) * a.T + t) * C + c] : a.vec[(size_t)j * C + c];
  }
  __syncthreads();
  float acc[NQ][NT][NG][4];
#pragma unroll
  for (int j = 0; j < NQ; ++j)
#pragma unroll
    for (int tp = 0; tp < NT; ++tp)
#pragma unroll
      for (int q = 0; q < NG; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][tp][q][e] = 0.f;
  float m[NQ], l[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) { m[j] = -1e30f; l[j] = 0.f; }
  // One token of look-ahead in a second register set.  Measured at configs[1] size (256 frames x 196 tokens x 2304 channels, bf16,
  // 3 queries): 157 us forward / 108 us backward = 1.5 / 2.1 TB/s against 162 / 184 us for the three-launch chain; the step gains
  // 0.1 ms.  It is NOT memory-bound: neither a two-token look-ahead nor an L2 prefetch three tokens ahead (LDS-DMA touches of
  // every 128-byte line) changed the time -- with 2 waves per SIMD the per-token chain (27 LDS operand reads, 3 x 6 bpermute
  // steps of the wave reductions, dependent FMA chains) is exposed.  Unrolling the buffer rotation made hipcc hoist the unrolled
  // steps' loads and spill 90 - 170 registers.  The next step would be one QUERY per wave (36 accumulators instead of 108: 12 - 16
  // waves per CU), tap rows re-read through L1 by the three query waves.
  Token<T, NT, NG> cur, nxt;
  const size_t row0 = (size_t)f * a.N;
  cur.load(a, row0 + min(wave, a.N - 1), lane);
  for (int n = wave; n < a.N; n += FWAVES) {
    nxt.load(a, row0 + min(n + FWAVES, a.N - 1), lane);      // the next token (a harmless re-read of the last row at the end)
    float d[NQ], wgt[NQ], al[NQ];
    token_dots<T, NQ, NT, NG>(cur, svec, D, lane, d);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const float s = wave_sum(d[j]) * a.inv_sqrt_d;
      if (lane == 0) ssc[j * a.N + n] = s;
      const float mn = fmaxf(m[j], s);
      al[j] = __expf(m[j] - mn);
      wgt[j] = __expf(s - mn);
      l[j] = l[j] * al[j] + wgt[j];
      m[j] = mn;
    }
    token_axpy<T, NQ, NT, NG, true>(cur, wgt, al, acc);
    cur = nxt;
  }
  if (lane == 0)
#pragma unroll
    for (int j = 0; j < NQ; ++j) { sst[(wave * NQ + j) * 2] = m[j]; sst[(wave * NQ + j) * 2 + 1] = l[j]; }
  __syncthreads();        // every wave is done with svec; (m, l) of all waves and all raw scores are visible
  float fac[NQ], M[NQ], L[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    float mm = -1e30f;
    for (int w = 0; w < FWAVES; ++w) mm = fmaxf(mm, sst[(w * NQ + j) * 2]);
    float ll = 0.f;
    for (int w = 0; w < FWAVES; ++w) ll += sst[(w * NQ + j) * 2 + 1] * __expf(sst[(w * NQ + j) * 2] - mm);
    M[j] = mm; L[j] = ll;
    fac[j] = __expf(m[j] - mm) / ll;       // a wave without tokens (N < 8) has l = 0 and contributes nothing
  }
  merge_waves<NQ, NT, NG>(svec, D, lane, wave, fac, acc);
  for (int i = threadIdx.x; i < NQ * C; i += 512) {
    const int j = i / C, c = i % C;
    a.pooled[(((size_t)b * NQ + j) * a.T + t) * C + c] = svec[i];
  }
  for (int i = threadIdx.x; i < NQ * a.N; i += 512) {
    const int j = i / a.N;
    float Mj = M[0], Lj = L[0];
#pragma unroll
    for (int q = 1; q < NQ; ++q) if (j == q) { Mj = M[q]; Lj = L[q]; }
    a.P[(size_t)f * NQ * a.N + i] = __expf(ssc[i] - Mj) / Lj;
  }
}

template <typename T, int NQ, int NT, int NG>
__global__ __launch_bounds__(512) void lstp_fused_bwd_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = a.D, C = NT * D;
  float* sdp = sm;                          // [NQ][C] dpooled of this frame; reused as the merge buffer
  float* sP = sm + NQ * C;                  // [NQ][N]
  float* sgb = sP + NQ * a.N;               // [FWAVES][NQ] partial gbar
  const int f = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = f / a.T, t = f % a.T;
  for (int i = threadIdx.x; i < NQ * C; i += 512) {
    const int j = i / C, c = i % C;
    sdp[i] = a.vec[(((size_t)b * NQ + j) * a.T + t) * C + c];
  }
  for (int i = threadIdx.x; i < NQ * a.N; i += 512) sP[i] = a.P[(size_t)f * NQ * a.N + i];
  __syncthreads();
  float acc[NQ][NT][NG][4];
#pragma unroll
  for (int j = 0; j < NQ; ++j)
#pragma unroll
    for (int tp = 0; tp < NT; ++tp)
#pragma unroll
      for (int q = 0; q < NG; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][tp][q][e] = 0.f;
  float gb[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) gb[j] = 0.f;
  Token<T, NT, NG> cur, nxt;
  const size_t row0 = (size_t)f * a.N;
  cur.load(a, row0 + min(wave, a.N - 1), lane);
  for (int n = wave; n < a.N; n += FWAVES) {
    nxt.load(a, row0 + min(n + FWAVES, a.N - 1), lane);
    float d[NQ], wgt[NQ];
    token_dots<T, NQ, NT, NG>(cur, sdp, D, lane, d);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      wgt[j] = sP[j * a.N + n] * wave_sum(d[j]);      // P_jn g_jn
      gb[j] += wgt[j];
    }
    token_axpy<T, NQ, NT, NG, false>(cur, wgt, wgt, acc);
    cur = nxt;
  }
  if (lane == 0)
#pragma unroll
    for (int j = 0; j < NQ; ++j) sgb[wave * NQ + j] = gb[j];
  __syncthreads();
  float one[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) one[j] = 1.f;
  merge_waves<NQ, NT, NG>(sdp, D, lane, wave, one, acc);
  for (int i = threadIdx.x; i < NQ * C; i += 512) {
    const int j = i / C, c = i % C;
    float g = 0.f;
    for (int w = 0; w < FWAVES; ++w) g += sgb[w * NQ + j];
    const size_t o = (((size_t)b * NQ + j) * a.T + t) * C + c;
    a.G[o] = a.inv_sqrt_d * (sdp[i] - g * a.pooled[o]);
  }
}

// [nq][C] vectors / merge buffer, [nq][N] scores or weights, per-wave statistics
size_t fused_lds(int nq, int n_taps, int D, int N) { return ((size_t)nq * n_taps * D + (size_t)nq * N + FWAVES * nq * 2) * sizeof(float); }

// shapes with an instantiation: nq 1..3, 1 or 3 taps, D = 768 or 1024 (ViT-B / ViT-L) -- except 3 queries on 3 taps of 1024
// channels, whose 144 accumulators + two tokens do not fit 256 registers (160+ spills: the three-launch form is faster)
bool fused_supported(int n_taps, int D, int N, int nq) {
  return nq >= 1 && nq <= 3 && (n_taps == 1 || n_taps == 3) && (D == 768 || D == 1024) && N >= 1 &&
         nq * n_taps * (D / 256) <= 27 && fused_lds(nq, n_taps, D, N) <= 96 * 1024;
}

template <typename K>
int fused_go(K kernel, const FusedArgs& a, size_t lds, int F, hipStream_t st) {
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return MVF_ERR_ARG;
  hipLaunchKernelGGL(kernel, dim3(F), dim3(512), lds, st, a);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

template <bool BWD, typename T, int NQ, int NT>
int fused_pick_ng(const FusedArgs& a, size_t lds, int F, hipStream_t st) {
  if (a.D == 768) return BWD ? fused_go(lstp_fused_bwd_kernel<T, NQ, NT, 3>, a, lds, F, st) : fused_go(lstp_fused_fwd_kernel<T, NQ, NT, 3>, a, lds, F, st);
  return BWD ? fused_go(lstp_fused_bwd_kernel<T, NQ, NT, 4>, a, lds, F, st) : fused_go(lstp_fused_fwd_kernel<T, NQ, NT, 4>, a, lds, F, st);
}
template <bool BWD, typename T, int NQ>
int fused_pick_nt(const FusedArgs& a, int n_taps, size_t lds, int F, hipStream_t st) {
  return n_taps == 1 ? fused_pick_ng<BWD, T, NQ, 1>(a, lds, F, st) : fused_pick_ng<BWD, T, NQ, 3>(a, lds, F, st);
}
template <bool BWD, typename T>
int fused_pick(const FusedArgs& a, int n_taps, int nq, int F, hipStream_t st) {
  const size_t lds = fused_lds(nq, n_taps, a.D, a.N);
  switch (nq) {
    case 1: return fused_pick_nt<BWD, T, 1>(a, n_taps, lds, F, st);
    case 2: return fused_pick_nt<BWD, T, 2>(a, n_taps, lds, F, st);
    default: return fused_pick_nt<BWD, T, 3>(a, n_taps, lds, F, st);
  }
}

// argument checks common to both one-pass forms; the VALU form's own shape limits (fused_supported) are applied by the entry
// points only AFTER the matrix-core form has had its turn -- 3 queries on 3 taps of 1024 channels (ViT-L, BASELINE configs[4])
// exist only there
int fused_fill(FusedArgs& a, const void* const* taps, int n_taps, int dtype, int D, int F, int N, int T, int nq) {
  MVF_CHECK_ARG(taps && F > 0 && N > 0 && T > 0 && F % T == 0 && (dtype == MVF_F32 || dtype == MVF_BF16));
  if (n_taps < 1 || n_taps > 3 || nq < 1) return MVF_ERR_UNSUPPORTED;
  for (int i = 0; i < n_taps; ++i) {
    MVF_CHECK_ARG(taps[i] && ((uintptr_t)taps[i] & 15) == 0);
    a.taps[i] = taps[i];
  }
  a.D = D; a.N = N; a.T = T;
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
  hipLaunchKernelGGL(lstp_reduce_frames_kernel, dim3(ceil_div(C, 64), nq), dim3(64 * RF_SL), 0, st, G, out, Bc, nq, T, C);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// One-pass pooling forward (see the kernel comment): pooled [Bc, nq, T, C] and P [F, nq, N] in one read of the taps.
// MVF_ERR_UNSUPPORTED when the shape is outside the kernel's register budget (nq > 3, more than 3 taps, D > 1024): the caller
// then runs the three-launch form (scores / softmax / weighted sum).
// (lstp_mfma.hip) the matrix-core form for bf16 taps; MVF_ERR_UNSUPPORTED where it has no instantiation
int mvf_lstp_mfma_impl(bool bwd, const void* const* taps, int n_taps, int D, int F, int N, int T, int nq, const float* vec,
                       int per_frame, float inv_sqrt_d, float* P, float* pooled, float* G, hipStream_t st);
static int g_lstp_form = [] { const char* e = getenv("MVF_LSTP_FORM"); return e ? atoi(e) : 0; }();

// which one-pass kernel mvf_lstp_fused_fwd / _bwd run (tests, A/B measurements): 0 = matrix-core form for bf16 taps where it has
// an instantiation, else the VALU form; 1 = always the VALU form
extern "C" int mvf_lstp_select(int form) {
  MVF_CHECK_ARG(form == 0 || form == 1);
  g_lstp_form = form;
  return MVF_OK;
}

extern "C" int mvf_lstp_fused_fwd(const void* const* taps, int n_taps, int dtype, int D, int F, int N, int T, int nq,
                                  const float* vec, int per_frame, float inv_sqrt_d, float* P, float* pooled,
                                  hipStream_t st) {
  FusedArgs a{};
  const int rc = fused_fill(a, taps, n_taps, dtype, D, F, N, T, nq);
  if (rc != MVF_OK) return rc;
  MVF_CHECK_ARG(vec && P && pooled);
  if (dtype == MVF_BF16 && g_lstp_form == 0) {
    const int r2 = mvf_lstp_mfma_impl(false, taps, n_taps, D, F, N, T, nq, vec, per_frame, inv_sqrt_d, P, pooled, nullptr, st);
    if (r2 != MVF_ERR_UNSUPPORTED) return r2;
  }
  if (!fused_supported(n_taps, D, N, nq)) return MVF_ERR_UNSUPPORTED;
  a.vec = vec; a.per_frame = per_frame; a.inv_sqrt_d = inv_sqrt_d; a.P = P; a.pooled = pooled;
  return dtype == MVF_BF16 ? fused_pick<false, bf16_t>(a, n_taps, nq, F, st) : fused_pick<false, float>(a, n_taps, nq, F, st);
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
  if (dtype == MVF_BF16 && g_lstp_form == 0) {
    const int r2 = mvf_lstp_mfma_impl(true, taps, n_taps, D, F, N, T, nq, dpooled, 1, inv_sqrt_d, const_cast<float*>(P),
                                      const_cast<float*>(pooled), G, st);
    if (r2 != MVF_ERR_UNSUPPORTED) return r2;
  }
  if (!fused_supported(n_taps, D, N, nq)) return MVF_ERR_UNSUPPORTED;
  a.vec = dpooled; a.per_frame = 1; a.inv_sqrt_d = inv_sqrt_d; a.P = const_cast<float*>(P); a.pooled = const_cast<float*>(pooled);
  a.G = G;
  return dtype == MVF_BF16 ? fused_pick<true, bf16_t>(a, n_taps, nq, F, st) : fused_pick<true, float>(a, n_taps, nq, F, st);
}

// ---- static queries folded through W_K (LSTPCrossAtt with num_dynamic = 0, mvformer.py:383: Q = Q_s + Q_s_b) ----
//     wq[j, c] = sum_k (qs[j, k] + qb[k]) wk[k, c]            qs [nq, d], qb [d], wk [d, C] (row stride ldw), wq [nq, C]
// and its backward, every parameter gradient ACCUMULATED in place (flat gradient buffer):
//     gqs[j, k] += sum_c dv[j, c] wk[k, c];   gqb[k] += sum_j (that);   gwk[k, c] += sum_j (qs[j, k] + qb[k]) dv[j, c]
// nq = 3 rows against a 384 x 2 304 matrix: through the 64-row MFMA GEMM these were six launches with K-long serial chains in a
// handful of workgroups (49 us each for the two [nq | 1] x 2 304 -> 384 products); here one launch each way, a few microseconds,
// plain fp32 FMAs in a fixed order (deterministic; no atomics).
//   forward : workgroup = 64 columns x 16 slices of k (k = s, s + 16, ..), 1 024 threads; q + b staged in LDS once; a thread's
//             weight loads are issued in batches of 8 before the first is used (as a run-time loop of load -> fma the 96 loads per
//             thread were 96 dependent round trips: 71 us for a 3 x 384 x 2 304 product), LDS combine in a fixed order
//   backward: workgroup = one row k of wk: threads stride over c in batches of 4 columns (loads first); gwk's row updated in
//             place, the nq dot products block-reduced
namespace {

constexpr int SQ_SL = 16;        // k slices per workgroup (forward)

__global__ __launch_bounds__(1024) void static_query_fwd_kernel(const float* __restrict__ qs, const float* __restrict__ qb,
                                                                const float* __restrict__ wk, long ldw, float* __restrict__ wq,
                                                                int nq, int d, int C) {
  extern __shared__ float sq_lds[];
  float* aq = sq_lds;                          // [nq][d]: q + b
  float* red = sq_lds + (size_t)nq * d;        // [SQ_SL][MAXQ][64]
  for (int i = threadIdx.x; i < nq * d; i += 1024) aq[i] = qs[i] + qb[i % d];
  __syncthreads();
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int cc = min(c, C - 1);
  float acc[MAXQ];
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) acc[j] = 0.f;
  constexpr int U = 8;
  for (int k0 = sl; k0 < d; k0 += SQ_SL * U) {
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + SQ_SL * u;
      w[u] = k < d ? wk[(long)k * ldw + cc] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(k0 + SQ_SL * u, d - 1);
#pragma unroll
      for (int j = 0; j < MAXQ; ++j)
        if (j < nq) acc[j] = fmaf(aq[j * d + k], w[u], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) red[(sl * MAXQ + j) * 64 + cl] = acc[j];
  __syncthreads();
  if (sl < nq && c < C) {       // wave j combines query j
    float v = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < SQ_SL; ++s2) v += red[(s2 * MAXQ + sl) * 64 + cl];
    wq[(long)sl * C + c] = v;
  }
}

__global__ __launch_bounds__(256) void static_query_bwd_kernel(const float* __restrict__ dv, long ldv, const float* __restrict__ qs,
                                                               const float* __restrict__ qb, const float* __restrict__ wk, long ldw,
                                                               float* __restrict__ gqs, float* __restrict__ gqb,
                                                               float* __restrict__ gwk, long ldgw, int nq, int d, int C) {
  __shared__ float red[4][MAXQ];
  const int k = blockIdx.x;
  float a[MAXQ], s[MAXQ];
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) {
    a[j] = j < nq ? qs[(long)j * d + k] + qb[k] : 0.f;
    s[j] = 0.f;
  }
  constexpr int U = 4;
  for (int c0 = threadIdx.x; c0 < C; c0 += 256 * U) {
    float w[U], go[U], v[U][MAXQ];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 256 * u;
      const bool in = c < C;
      w[u] = in ? wk[(long)k * ldw + c] : 0.f;
      go[u] = in ? gwk[(long)k * ldgw + c] : 0.f;
#pragma unroll
      for (int j = 0; j < MAXQ; ++j) v[u][j] = (in && j < nq) ? dv[(long)j * ldv + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 256 * u;
      float g = 0.f;
#pragma unroll
      for (int j = 0; j < MAXQ; ++j)
        if (j < nq) {
          s[j] = fmaf(v[u][j], w[u], s[j]);
          g = fmaf(a[j], v[u][j], g);
        }
      if (c < C) gwk[(long)k * ldgw + c] = go[u] + g;
    }
  }
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) {
    float v = s[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int j = 0; j < nq; ++j) {
      const float v = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
      gqs[(long)j * d + k] += v;
      tot += v;
    }
    gqb[k] += tot;
  }
}

}  // namespace

extern "C" int mvf_static_query_fwd(const float* qs, const float* qb, const float* wk, long ldw, float* wq, int nq, int d, int C,
                                    hipStream_t st) {
  MVF_CHECK_ARG(qs && qb && wk && wq && nq >= 1 && nq <= MAXQ && d >= 1 && C >= 1 && ldw >= C);
  const size_t lds = ((size_t)nq * d + (size_t)SQ_SL * MAXQ * 64) * 4;
  if (lds > 64 * 1024) return MVF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(static_query_fwd_kernel, dim3((C + 63) / 64), dim3(1024), lds, st, qs, qb, wk, ldw, wq, nq, d, C);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_static_query_bwd(const float* dv, long ldv, const float* qs, const float* qb, const float* wk, long ldw,
                                    float* gqs, float* gqb, float* gwk, long ldgw, int nq, int d, int C, hipStream_t st) {
  MVF_CHECK_ARG(dv && qs && qb && wk && gqs && gqb && gwk && nq >= 1 && nq <= MAXQ && d >= 1 && C >= 1 && ldw >= C && ldgw >= C &&
                ldv >= C);
  hipLaunchKernelGGL(static_query_bwd_kernel, dim3(d), dim3(256), 0, st, dv, ldv, qs, qb, wk, ldw, gqs, gqb, gwk, ldgw, nq, d, C);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}
