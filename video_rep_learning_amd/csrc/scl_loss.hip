// Sequence-contrastive loss (SCL), fused forward + backward.
// Reference: SCL.compute_sequence_loss, CARL_MVF/algos/scl.py:52-105 (safe_div :13-16).
//
// The reference materialises six dense [M,M] fp32 tensors (logits, distance, weight, label, exp_logits,
// pair mask) and loops over the batch in Python.  Here every pair quantity is recomputed in registers from
// per-row metadata (step, seq_len, mask; video/view ids follow from the row index), the similarity tile
// E_i . E_k / tau lives in LDS only, and nothing of size M^2 touches HBM:
//   scl_stats : per row i   S_i = sum_k w_ik exp(l_ik),  R_i = sum_{k in other view} pos_ik,
//                           c_i = sum_k m_ik y_ik p_ik/(p_ik+eps),  loss_i = sum_k m_ik kl(y_ik, p_ik)
//   scl_grad  : dE_i = g/(tau Z) sum_k (G_ik + G_ki) E_k,  G_ik = w_ik p_ik c_i - m_ik y_ik p_ik/(p_ik+eps)
// (p = exp(l)/S, y = pos/R, Z = sum of masks).  The transposed term G_ki is evaluated from the stats of
// row k, so no atomics are needed and the gradient is bit-reproducible.  `row0/rows` restrict the output
// rows so that, with cross-GPU gathered embeddings, a rank only produces the gradient of its own slice.
#include "common.h"
#include "mvf_hip_internal.h"

namespace {

constexpr int RB = 16;   // rows per workgroup
constexpr int CB = 64;   // columns per tile

struct SclArgs {
  const float* emb;     // [M, E]
  const float* step;    // [M]
  const float* len;     // [M]
  const float* mask;    // [M]
  float* S; float* R; float* c; float* lossrow;   // [M] each
  float* dE;            // [rows, E]
  const float* gout;    // upstream scalar gradient (device) or null (= 1)
  int M, E, T;          // rows, channels, frames per view (2 views)
  int single, noself;
  float inv_tau, inv_2var;
  int row0, rows;
};

struct Meta { float step, len, mask; int vid, view; };

__device__ __forceinline__ Meta load_meta(const SclArgs& a, int r) {
  Meta m;
  m.step = a.step[r]; m.len = a.len[r]; m.mask = a.mask[r];
  m.vid = r / (2 * a.T); m.view = (r / a.T) & 1;
  return m;
}

// weight, pair mask and (un-normalised) positive weight of ordered pair (i -> k)   scl.py:59-96
__device__ __forceinline__ void pair_terms(const SclArgs& a, const Meta& i, const Meta& k, float& w, float& pm, float& pos) {
  pm = i.mask * k.mask;
  const bool same_vid = i.vid == k.vid, same_view = i.view == k.view;
  w = 1.f;
  if (a.single && !same_vid) w = 0.f;
  if (a.noself && same_vid && same_view) w = 0.f;
  if (pm == 0.f) w = 1e-6f;
  pos = 0.f;
  if (same_vid && !same_view) {
    float d = fabsf(i.step / i.len * k.len - k.step);
    if (pm == 0.f) d = 1e6f;
    pos = expf(-d * d * a.inv_2var);
  }
}

__device__ __forceinline__ float sum16(float v) {  // over the 16 lanes that share a row
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}

// LDS carve: er [RB][E+1], ec [CB][E+1], coef [RB][CB], row metas, col metas
struct Smem {
  float* er; float* ec; float* coef; Meta* mr; Meta* mc;
  __device__ Smem(char* base, int E) {
    er = (float*)base;
    ec = er + RB * (E + 1);
    coef = ec + CB * (E + 1);
    mr = (Meta*)(coef + RB * CB);
    mc = mr + RB;
  }
};
size_t smem_bytes(int E) { return ((size_t)(RB + CB) * (E + 1) + RB * CB) * 4 + (RB + CB) * sizeof(Meta); }

__device__ __forceinline__ void load_rows(const SclArgs& a, float* dst, Meta* md, int r0, int nrows) {
  for (int i = threadIdx.x; i < nrows * a.E; i += 256) {
    const int r = i / a.E, e = i % a.E;
    const int gr = min(r0 + r, a.M - 1);
    dst[r * (a.E + 1) + e] = a.emb[(size_t)gr * a.E + e];
  }
  for (int r = threadIdx.x; r < nrows; r += 256) {
    const int gr = r0 + r;
    Meta m = load_meta(a, min(gr, a.M - 1));
    if (gr >= a.M) { m.vid = -1 - r; m.mask = 0.f; }  // out of range: matches nothing
    md[r] = m;
  }
}

// Columns that can interact with rows i0 .. i0+RB-1.  With 'single' in NEGATIVE_TYPE every pair outside the row's own
// video has weight 0 and no label mass in BOTH directions (scl.py:74-77), so only the columns of the videos these rows
// belong to are visited: 2T of M columns -- 4x less work at B = 4, 32x with 8 ranks' embeddings gathered.
__device__ __forceinline__ void col_range(const SclArgs& a, int i0, int& kbeg, int& kend) {
  kbeg = 0;
  kend = a.M;
  if (a.single) {
    const int per = 2 * a.T;
    kbeg = (min(i0, a.M - 1) / per) * per;
    kend = (min(i0 + RB - 1, a.M - 1) / per + 1) * per;
  }
}

__global__ __launch_bounds__(256) void scl_stats_kernel(SclArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Smem sm(smem_raw, a.E);
  const int r = threadIdx.x >> 4, cl = threadIdx.x & 15;
  const int i0 = blockIdx.x * RB;
  const int E1 = a.E + 1;
  load_rows(a, sm.er, sm.mr, i0, RB);
  float Ssum = 0.f, Rsum = 0.f;
  int kbeg, kend;
  col_range(a, i0, kbeg, kend);
  for (int k0 = kbeg; k0 < kend; k0 += CB) {
    __syncthreads();
    load_rows(a, sm.ec, sm.mc, k0, CB);
    __syncthreads();
    const Meta mi = sm.mr[r];
#pragma unroll
    for (int q = 0; q < CB / 16; ++q) {
      const int kc = cl + 16 * q;
      if (k0 + kc >= a.M) continue;
      float dot = 0.f;
      for (int e = 0; e < a.E; ++e) dot += sm.er[r * E1 + e] * sm.ec[kc * E1 + e];
      float w, pm, pos;
      pair_terms(a, mi, sm.mc[kc], w, pm, pos);
      Ssum += w * expf(dot * a.inv_tau);
      Rsum += pos;
    }
  }
  Ssum = sum16(Ssum);
  Rsum = sum16(Rsum);
  // second sweep over the positive block only (other view of the same video): c_i and loss_i
  const int gi = i0 + r;
  float csum = 0.f, lsum = 0.f;
  if (gi < a.M) {
    const Meta mi = sm.mr[r];
    const int kstart = mi.vid * 2 * a.T + (1 - mi.view) * a.T;
    for (int kk = cl; kk < a.T; kk += 16) {
      const int k = kstart + kk;
      const Meta mk = load_meta(a, k);
      float dot = 0.f;
      for (int e = 0; e < a.E; ++e) dot += sm.er[r * E1 + e] * a.emb[(size_t)k * a.E + e];
      float w, pm, pos;
      pair_terms(a, mi, mk, w, pm, pos);
      const float y = Rsum > 0.f ? pos / Rsum : 0.f;      // safe_div: 0/0 -> 0
      const float p = expf(dot * a.inv_tau) / Ssum;
      if (y > 0.f) {
        csum += pm * y * p / (p + 1e-6f);
        lsum += pm * (y * logf(y) - y * logf(p + 1e-6f));
      }
    }
  }
  csum = sum16(csum);
  lsum = sum16(lsum);
  if (cl == 0 && gi < a.M) { a.S[gi] = Ssum; a.R[gi] = Rsum; a.c[gi] = csum; a.lossrow[gi] = lsum; }
}

// loss = sum(lossrow) / sum(mask)
__global__ __launch_bounds__(256) void scl_finalize_kernel(const float* __restrict__ lossrow, const float* __restrict__ mask,
                                                           int M, float* __restrict__ loss) {
  __shared__ float s1[4], s2[4];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < M; i += 256) { a += lossrow[i]; b += mask[i]; }
  a = wave_sum(a); b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = a; s2[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (s1[0] + s1[1] + s1[2] + s1[3]) / (s2[0] + s2[1] + s2[2] + s2[3]);
}

__global__ __launch_bounds__(256) void scl_grad_kernel(SclArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Smem sm(smem_raw, a.E);
  __shared__ float zs[4];
  const int r = threadIdx.x >> 4, cl = threadIdx.x & 15;
  const int i0 = a.row0 + blockIdx.x * RB;
  const int E1 = a.E + 1;
  // Z = sum of masks
  float z = 0.f;
  for (int i = threadIdx.x; i < a.M; i += 256) z += a.mask[i];
  z = wave_sum(z);
  if ((threadIdx.x & 63) == 0) zs[threadIdx.x >> 6] = z;
  load_rows(a, sm.er, sm.mr, i0, RB);
  __syncthreads();
  const float g = (a.gout ? a.gout[0] : 1.f) * a.inv_tau / (zs[0] + zs[1] + zs[2] + zs[3]);
  const int gi = min(i0 + r, a.M - 1);
  const float Si = a.S[gi], Ri = a.R[gi], ci = a.c[gi];
  float acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  int kbeg, kend;
  col_range(a, i0, kbeg, kend);
  for (int k0 = kbeg; k0 < kend; k0 += CB) {
    __syncthreads();
    load_rows(a, sm.ec, sm.mc, k0, CB);
    __syncthreads();
    const Meta mi = sm.mr[r];
#pragma unroll
    for (int q = 0; q < CB / 16; ++q) {
      const int kc = cl + 16 * q;
      const int gk = k0 + kc;
      float coef = 0.f;
      if (gk < a.M) {
        float dot = 0.f;
        for (int e = 0; e < a.E; ++e) dot += sm.er[r * E1 + e] * sm.ec[kc * E1 + e];
        const float ex = expf(dot * a.inv_tau);
        const Meta mk = sm.mc[kc];
        float w, pm, pos;
        pair_terms(a, mi, mk, w, pm, pos);                       // (i -> k)
        float p = ex / Si;
        float y = Ri > 0.f ? pos / Ri : 0.f;
        coef = w * p * ci - (y > 0.f ? pm * y * p / (p + 1e-6f) : 0.f);
        pair_terms(a, mk, mi, w, pm, pos);                       // (k -> i)
        const float Rk = a.R[gk];
        p = ex / a.S[gk];
        y = Rk > 0.f ? pos / Rk : 0.f;
        coef += w * p * a.c[gk] - (y > 0.f ? pm * y * p / (p + 1e-6f) : 0.f);
      }
      sm.coef[r * CB + kc] = coef;
    }
    __syncthreads();
    for (int k = 0; k < CB; ++k) {
      const float cf = sm.coef[r * CB + k];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int e = cl + 16 * q;
        if (e < a.E) acc[q] += cf * sm.ec[k * E1 + e];
      }
    }
  }
  const int li = blockIdx.x * RB + r;
  if (li < a.rows && i0 + r < a.M) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = cl + 16 * q;
      if (e < a.E) a.dE[(size_t)li * a.E + e] = g * acc[q];
    }
  }
}

int fill(SclArgs& a, const float* emb, const float* step, const float* len, const float* mask, float* S, float* R, float* c,
         float* lossrow, int M, int E, int T, int negative_flags, float temperature, float label_variance) {
  MVF_CHECK_ARG(emb && step && len && mask && S && R && c && lossrow);
  MVF_CHECK_ARG(M > 0 && T > 0 && M % (2 * T) == 0 && E > 0 && E <= 256);
  a.emb = emb; a.step = step; a.len = len; a.mask = mask; a.S = S; a.R = R; a.c = c; a.lossrow = lossrow;
  a.M = M; a.E = E; a.T = T; a.single = negative_flags & 1; a.noself = (negative_flags >> 1) & 1;
  a.inv_tau = 1.0f / temperature; a.inv_2var = 1.0f / (2.0f * label_variance);
  return MVF_OK;
}

template <typename K>
void set_lds(K kern, size_t bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// the loader's bookkeeping tensors -> the three per-row float vectors the loss kernels read (rows ordered (video, view, frame))
__global__ void scl_rows_kernel(const long long* __restrict__ steps, const long long* __restrict__ lens,
                                const float* __restrict__ masks, float* __restrict__ out, int M, int T) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= M) return;
  out[r] = (float)steps[r];
  out[M + r] = (float)lens[r / T];
  out[2 * M + r] = masks != nullptr ? masks[r] : 1.0f;
}

}  // namespace

// rows[0 | 1 | 2][M]: chosen_steps [clips, T] int64, seq_lens [clips] int64 (one per clip, repeated over its T frames) and
// video_masks [clips, T] fp32 (NULL: all ones) as floats per embedding row -- what SCL.compute_sequence_loss derives with
// reshape / expand / .float() before the loss proper (algos/scl.py:52-64)
extern "C" int mvf_scl_rows(const long long* steps, const long long* seq_lens, const float* masks, float* rows, int clips,
                            int T, hipStream_t st) {
  MVF_CHECK_ARG(steps && seq_lens && rows && clips > 0 && T > 0);
  const int M = clips * T;
  hipLaunchKernelGGL(scl_rows_kernel, dim3((M + 255) / 256), dim3(256), 0, st, steps, seq_lens, masks, rows, M, T);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// negative_flags: bit0 = 'single' in NEGATIVE_TYPE, bit1 = 'noself' in NEGATIVE_TYPE
extern "C" int mvf_scl_fwd(const float* emb, const float* step, const float* len, const float* mask, float* S, float* R,
                           float* c, float* lossrow, float* loss, int M, int E, int T, int negative_flags,
                           float temperature, float label_variance, hipStream_t st) {
  SclArgs a{};
  int rc = fill(a, emb, step, len, mask, S, R, c, lossrow, M, E, T, negative_flags, temperature, label_variance);
  if (rc != MVF_OK) return rc;
  MVF_CHECK_ARG(loss);
  const size_t lds = smem_bytes(E);
  set_lds(scl_stats_kernel, lds);
  hipLaunchKernelGGL(scl_stats_kernel, dim3(ceil_div(M, RB)), dim3(256), lds, st, a);
  hipLaunchKernelGGL(scl_finalize_kernel, dim3(1), dim3(256), 0, st, lossrow, mask, M, loss);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// dE [rows, E] = d loss / d emb[row0 : row0+rows] * gout
extern "C" int mvf_scl_bwd(const float* emb, const float* step, const float* len, const float* mask, const float* S,
                           const float* R, const float* c, const float* gout, float* dE, int M, int E, int T, int row0,
                           int rows, int negative_flags, float temperature, float label_variance, hipStream_t st) {
  SclArgs a{};
  int rc = fill(a, emb, step, len, mask, const_cast<float*>(S), const_cast<float*>(R), const_cast<float*>(c),
                const_cast<float*>(S), M, E, T, negative_flags, temperature, label_variance);
  if (rc != MVF_OK) return rc;
  MVF_CHECK_ARG(dE && row0 >= 0 && rows > 0 && row0 + rows <= M && row0 % RB == 0);
  a.dE = dE; a.gout = gout; a.row0 = row0; a.rows = rows;
  const size_t lds = smem_bytes(E);
  set_lds(scl_grad_kernel, lds);
  hipLaunchKernelGGL(scl_grad_kernel, dim3(ceil_div(rows, RB)), dim3(256), lds, st, a);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}
