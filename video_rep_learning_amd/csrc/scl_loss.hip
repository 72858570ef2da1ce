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
  float* loss;          // forward: the scalar, written by the workgroup that arrives last (common.h last_arriver)
  int ticket;           // its ticket slot (mvf_hip_internal.h TicketRing)
};

// loss = sum(lossrow) / sum(mask), in the launch that produced lossrow: the last workgroup to arrive adds the rows in a fixed order
__device__ unsigned g_scl_ticket[TICKET_SLOTS];
TicketRing g_scl_ring;
__device__ __forceinline__ void scl_finalize(const SclArgs& a) {
  if (a.loss == nullptr || !last_arriver(&g_scl_ticket[a.ticket], gridDim.x)) return;
  __shared__ float s1[16], s2[16];
  float x = 0.f, y = 0.f;
  for (int i = threadIdx.x; i < a.M; i += blockDim.x) { x += a.lossrow[i]; y += a.mask[i]; }
  x = wave_sum(x); y = wave_sum(y);
  if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = x; s2[threadIdx.x >> 6] = y; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float u = 0.f, v = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { u += s1[w]; v += s2[w]; }
    a.loss[0] = u / v;
  }
}

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
  scl_finalize(a);
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

// ------------------------------------------------------------------------------------------------
// Matrix-core forms (E = 64 | 128 | 256): the pair similarities E_i . E_k on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains).
// With cross-GPU gathered embeddings (BASELINE configs[2]: 8 ranks x 256 rows, M = 2 048) the scalar kernels above take
// 0.35 ms forward and 4.5 ms backward (16 workgroups walking 2 048 columns with two LDS reads per FMA); these take the same
// walk at one MFMA per 1 024 FMAs.  A workgroup = 16 rows x all their columns, NW waves x 16 columns per step:
//   D^T tile = E_cols (A operand: lane (column li, k chunk g) holds E/4 consecutive channels straight from global memory --
//   L2-resident, no LDS staging) x E_rows^T (B operand, loop-invariant, in registers): lane (li, g) receives the dots of ROW li
//   with columns 4 g .. 4 g + 3 -- one row per lane, so the row sums are in-lane + two cross-lane steps, and in the backward
//   those four coefficients ARE the B operand of dE^T[e][i] = sum_k E_k[e] coef[i][k] (k slot g, step s <-> column 4 g + s):
//   no transpose of the coefficient tile; the E_k[e] operand comes from a wave-private LDS copy of the 16 columns the wave has
//   just loaded.  Everything is summed in a fixed order (lanes, then waves): bit-reproducible, no atomics.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ Meta meta_of(const SclArgs& a, int r, int q /* = r / T */) {
  Meta m;
  m.step = a.step[r]; m.len = a.len[r]; m.mask = a.mask[r];
  m.vid = q >> 1; m.view = q & 1;
  return m;
}

template <int E>
__device__ __forceinline__ void load_chunk(const float* __restrict__ p, float (&v)[E / 4]) {
#pragma unroll
  for (int s = 0; s < E / 16; ++s) {
    const float4 x = reinterpret_cast<const float4*>(p)[s];
    v[4 * s] = x.x; v[4 * s + 1] = x.y; v[4 * s + 2] = x.z; v[4 * s + 3] = x.w;
  }
}

// per-column scalars of the four columns kb .. kb + 3 a lane handles in one step, fetched together at the top of the step (one
// 16-byte load per array where all four are in range and the address is 16-byte aligned; element loads otherwise)
struct Col4 { float step[4], len[4], mask[4]; };
__device__ __forceinline__ void load4(const float* __restrict__ p, int kb, int M, float (&v)[4]) {
  if (kb + 4 <= M && ((uintptr_t)(p + kb) & 15) == 0) {      // ('single' negatives with odd T, or row vectors sliced out of a
    const float4 x = *reinterpret_cast<const float4*>(p + kb);   //  [3, M] tensor with M % 4 != 0, start columns off a 16-byte boundary)
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = p[min(kb + r, M - 1)];
  }
}
__device__ __forceinline__ Col4 load_cols(const SclArgs& a, int kb) {
  Col4 c;
  load4(a.step, kb, a.M, c.step); load4(a.len, kb, a.M, c.len); load4(a.mask, kb, a.M, c.mask);
  return c;
}
__device__ __forceinline__ Meta meta_c(const Col4& c, int r, int q) {
  Meta m;
  m.step = c.step[r]; m.len = c.len[r]; m.mask = c.mask[r];
  m.vid = q >> 1; m.view = q & 1;
  return m;
}

// dots of row li with columns k0 + 4 g .. + 3 (k0 = the wave's first column): d[r]
template <int E>
__device__ __forceinline__ f32x4_t dots16(const float (&cf)[E / 4], const float (&rf)[E / 4]) {
  f32x4_t d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < E / 4; ++s) d = __builtin_amdgcn_mfma_f32_16x16x4f32(cf[s], rf[s], d, 0, 0, 0);
  return d;
}

// NW waves per workgroup (16 rows x all columns), wave w taking columns 16 w .. of every 16 NW-column step: with 4 waves the
// gathered size was latency-bound (one wave per SIMD, its exp / division VALU work and its MFMAs strictly in sequence: 62 + 100 us
// at M = 2 048); 8 waves share a SIMD two at a time and walk half the columns each (16 waves: 128 registers per lane, spills).
// LDS of the forward: pos[16 rows][T][3] (exp(l), pos, pair mask of the row's positive block) + red[NW waves][16][2] + red2
template <int NW>
size_t stats_mfma_lds(int T) { return ((size_t)16 * T * 3 + 2 * NW * 16 * 2) * sizeof(float); }

template <int E, int NW, bool PF>   // PF: request the next step's operands one step ahead (costs E/4 + 12 registers)
__global__ __launch_bounds__(64 * NW) void scl_stats_mfma_kernel(SclArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* posb = reinterpret_cast<float*>(smem_raw);            // [16][T][3]
  float* red = posb + 16 * a.T * 3;                            // [NW][16][2]
  float* red2 = red + NW * 16 * 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int i0 = blockIdx.x * RB;
  const int gi = min(i0 + li, a.M - 1);
  float rf[E / 4];
  load_chunk<E>(a.emb + (size_t)gi * E + g * (E / 4), rf);
  Meta mi = meta_of(a, gi, gi / a.T);
  if (i0 + li >= a.M) { mi.vid = -1 - li; mi.mask = 0.f; }
  const int kstart = mi.vid * 2 * a.T + (1 - mi.view) * a.T;   // first column of the row's positive block (other view, same video)
  float Ssum = 0.f, Rsum = 0.f;
  int kbeg, kend;
  col_range(a, i0, kbeg, kend);
  kend = min(kend, a.M);
  // one step ahead: the next step's column chunk and scalars are requested before this step's MFMAs (a step is ~1 k cycles of
  // matrix work behind ~2 k cycles of L2 latency: un-prefetched, 16 workgroups walking 2 048 columns spent 2/3 of the time waiting)
  float cf[E / 4], nf[E / 4];
  Col4 cc, nc;
  int k0 = kbeg + wave * 16;
  if (k0 < kend) {
    load_chunk<E>(a.emb + (size_t)min(k0 + li, a.M - 1) * E + g * (E / 4), cf);
    cc = load_cols(a, k0 + 4 * g);
  }
  constexpr int STEP = 16 * NW;
  for (; k0 < kend; k0 += STEP) {
    if constexpr (PF) {
      if (k0 + STEP < kend) {
        load_chunk<E>(a.emb + (size_t)min(k0 + STEP + li, a.M - 1) * E + g * (E / 4), nf);
        nc = load_cols(a, k0 + STEP + 4 * g);
      }
    }
    const f32x4_t d = dots16<E>(cf, rf);
    const int kb = k0 + 4 * g;
    int q = kb / a.T, rem = kb - q * a.T;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kb + r;
      if (k < kend) {
        const Meta mk = meta_c(cc, r, q);
        float w, pm, pos;
        pair_terms(a, mi, mk, w, pm, pos);
        const float ex = expf(d[r] * a.inv_tau);
        Ssum += w * ex;
        Rsum += pos;
        if (mi.vid == mk.vid && mi.view != mk.view) {
          float* pb = posb + ((size_t)li * a.T + (k - kstart)) * 3;
          pb[0] = ex; pb[1] = pos; pb[2] = pm;
        }
      }
      if (++rem == a.T) { rem = 0; ++q; }
    }
    if constexpr (PF) {
#pragma unroll
      for (int s2 = 0; s2 < E / 4; ++s2) cf[s2] = nf[s2];
      cc = nc;
    } else if (k0 + STEP < kend) {
      load_chunk<E>(a.emb + (size_t)min(k0 + STEP + li, a.M - 1) * E + g * (E / 4), cf);
      cc = load_cols(a, k0 + STEP + 4 * g);
    }
  }
  Ssum += __shfl_xor(Ssum, 16, 64); Ssum += __shfl_xor(Ssum, 32, 64);
  Rsum += __shfl_xor(Rsum, 16, 64); Rsum += __shfl_xor(Rsum, 32, 64);
  if (g == 0) { red[(wave * 16 + li) * 2] = Ssum; red[(wave * 16 + li) * 2 + 1] = Rsum; }
  __syncthreads();
  Ssum = 0.f; Rsum = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) { Ssum += red[(w * 16 + li) * 2]; Rsum += red[(w * 16 + li) * 2 + 1]; }   // wave order: fixed
  // c_i and loss_i over the row's positive block: column kk handled by lane group (wave, g) = kk mod 4 NW
  float csum = 0.f, lsum = 0.f;
  if (i0 + li < a.M) {
    for (int kk = wave * 4 + g; kk < a.T; kk += 4 * NW) {
      const float* pb = posb + ((size_t)li * a.T + kk) * 3;
      const float ex = pb[0], pos = pb[1], pm = pb[2];
      const float y = Rsum > 0.f ? pos / Rsum : 0.f;      // safe_div: 0/0 -> 0
      const float p = ex / Ssum;
      if (y > 0.f) {
        csum += pm * y * p / (p + 1e-6f);
        lsum += pm * (y * logf(y) - y * logf(p + 1e-6f));
      }
    }
  }
  csum += __shfl_xor(csum, 16, 64); csum += __shfl_xor(csum, 32, 64);
  lsum += __shfl_xor(lsum, 16, 64); lsum += __shfl_xor(lsum, 32, 64);
  if (g == 0) { red2[(wave * 16 + li) * 2] = csum; red2[(wave * 16 + li) * 2 + 1] = lsum; }
  __syncthreads();
  if (wave == 0 && g == 0 && i0 + li < a.M) {
    const int r = i0 + li;
    a.S[r] = Ssum; a.R[r] = Rsum;
    float cs = 0.f, ls = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { cs += red2[(w * 16 + li) * 2]; ls += red2[(w * 16 + li) * 2 + 1]; }
    a.c[r] = cs;
    a.lossrow[r] = ls;
  }
  scl_finalize(a);
}

// LDS of the backward: NW wave-private column copies [16][E + 4] (re-used as the waves' partial gradients [16 rows][E + 4])
template <int E, int NW>
constexpr size_t grad_mfma_lds() { return (size_t)NW * 16 * (E + 4) * sizeof(float); }

template <int E, int NW, bool PF>
__global__ __launch_bounds__(64 * NW) void scl_grad_mfma_kernel(SclArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ float zs[NW];
  constexpr int LD = E + 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  float* wec = reinterpret_cast<float*>(smem_raw) + wave * 16 * LD;
  const int i0 = a.row0 + blockIdx.x * RB;
  float z = 0.f;
  for (int i = threadIdx.x; i < a.M; i += 64 * NW) z += a.mask[i];
  z = wave_sum(z);
  if (lane == 0) zs[wave] = z;
  const int gi = min(i0 + li, a.M - 1);
  float rf[E / 4];
  load_chunk<E>(a.emb + (size_t)gi * E + g * (E / 4), rf);
  Meta mi = meta_of(a, gi, gi / a.T);
  if (i0 + li >= a.M) { mi.vid = -1 - li; mi.mask = 0.f; }
  const float Si = a.S[gi], Ri = a.R[gi], ci = a.c[gi];
  f32x4_t acc[E / 16];
#pragma unroll
  for (int et = 0; et < E / 16; ++et) acc[et] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  int kbeg, kend;
  col_range(a, i0, kbeg, kend);
  kend = min(kend, a.M);
  float cf[E / 4], nf[E / 4];
  Col4 cc, nc;
  float cS[4], cR[4], cC[4], nS[4], nR[4], nC[4];
  int k0 = kbeg + wave * 16;
  if (k0 < kend) {
    load_chunk<E>(a.emb + (size_t)min(k0 + li, a.M - 1) * E + g * (E / 4), cf);
    cc = load_cols(a, k0 + 4 * g);
    load4(a.S, k0 + 4 * g, a.M, cS); load4(a.R, k0 + 4 * g, a.M, cR); load4(a.c, k0 + 4 * g, a.M, cC);
  }
  constexpr int STEP = 16 * NW;
  for (; k0 < kend; k0 += STEP) {
    if constexpr (PF) {
      if (k0 + STEP < kend) {   // one step ahead, as in the forward
        load_chunk<E>(a.emb + (size_t)min(k0 + STEP + li, a.M - 1) * E + g * (E / 4), nf);
        nc = load_cols(a, k0 + STEP + 4 * g);
        load4(a.S, k0 + STEP + 4 * g, a.M, nS); load4(a.R, k0 + STEP + 4 * g, a.M, nR); load4(a.c, k0 + STEP + 4 * g, a.M, nC);
      }
    }
    // the wave's 16 columns -> its LDS copy [column][channel] (lane (li, g) holds channels g E/4 .. of column k0 + li)
#pragma unroll
    for (int s = 0; s < E / 16; ++s)
      *reinterpret_cast<float4*>(wec + li * LD + g * (E / 4) + 4 * s) = make_float4(cf[4 * s], cf[4 * s + 1], cf[4 * s + 2], cf[4 * s + 3]);
    const f32x4_t d = dots16<E>(cf, rf);
    const int kb = k0 + 4 * g;
    int q = kb / a.T, rem = kb - q * a.T;
    float coef[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kb + r;
      float cv = 0.f;
      if (k < kend) {
        const Meta mk = meta_c(cc, r, q);
        const float ex = expf(d[r] * a.inv_tau);
        float w, pm, pos;
        pair_terms(a, mi, mk, w, pm, pos);                       // (i -> k)
        float p = ex / Si;
        float y = Ri > 0.f ? pos / Ri : 0.f;
        cv = w * p * ci - (y > 0.f ? pm * y * p / (p + 1e-6f) : 0.f);
        pair_terms(a, mk, mi, w, pm, pos);                       // (k -> i)
        const float Rk = cR[r];
        p = ex / cS[r];
        y = Rk > 0.f ? pos / Rk : 0.f;
        cv += w * p * cC[r] - (y > 0.f ? pm * y * p / (p + 1e-6f) : 0.f);
      }
      coef[r] = cv;
      if (++rem == a.T) { rem = 0; ++q; }
    }
    // dE^T[e][i] += sum over the wave's 16 columns: A = E_k[e] (row e = li, k slot g -> column 4 g + s), B = coef[i = li][4 g + s]
#pragma unroll
    for (int et = 0; et < E / 16; ++et)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(wec[(4 * g + s) * LD + et * 16 + li], coef[s], acc[et], 0, 0, 0);
    if constexpr (PF) {
#pragma unroll
      for (int s2 = 0; s2 < E / 4; ++s2) cf[s2] = nf[s2];
      cc = nc;
#pragma unroll
      for (int r = 0; r < 4; ++r) { cS[r] = nS[r]; cR[r] = nR[r]; cC[r] = nC[r]; }
    } else if (k0 + STEP < kend) {
      load_chunk<E>(a.emb + (size_t)min(k0 + STEP + li, a.M - 1) * E + g * (E / 4), cf);
      cc = load_cols(a, k0 + STEP + 4 * g);
      load4(a.S, k0 + STEP + 4 * g, a.M, cS); load4(a.R, k0 + STEP + 4 * g, a.M, cR); load4(a.c, k0 + STEP + 4 * g, a.M, cC);
    }
  }
  // the four waves' partial gradients -> LDS [row i][channel] (lane (i = li, g) holds channels et 16 + 4 g .. + 3), summed in wave order
  __syncthreads();     // (also: zs complete; every wave done with its column copy)
#pragma unroll
  for (int et = 0; et < E / 16; ++et)
    *reinterpret_cast<float4*>(wec + li * LD + et * 16 + 4 * g) = make_float4(acc[et][0], acc[et][1], acc[et][2], acc[et][3]);
  __syncthreads();
  float zt = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) zt += zs[w];
  const float gs = (a.gout ? a.gout[0] : 1.f) * a.inv_tau / zt;
  const float* all = reinterpret_cast<const float*>(smem_raw);
  for (int idx = threadIdx.x; idx < 16 * (E / 4); idx += 64 * NW) {
    const int r = idx / (E / 4), e4 = (idx - r * (E / 4)) * 4;
    const int lrow = blockIdx.x * RB + r;
    if (lrow < a.rows && i0 + r < a.M) {
      float4 v = *reinterpret_cast<const float4*>(all + r * LD + e4);
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const float4 u = *reinterpret_cast<const float4*>(all + (w * 16 + r) * LD + e4);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      *reinterpret_cast<float4*>(a.dE + (size_t)lrow * E + e4) = make_float4(gs * v.x, gs * v.y, gs * v.z, gs * v.w);
    }
  }
}

// 0 = matrix-core kernels where they have an instantiation (E = 64 | 128 | 256, T <= 512), 1 = always the scalar kernels (tests, A/B)
int g_scl_form = [] { const char* e = getenv("MVF_SCL_FORM"); return e ? atoi(e) : 0; }();
bool scl_mfma_ok(int E, int T) { return g_scl_form == 0 && (E == 64 || E == 128 || E == 256) && T <= 512; }

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
  a.loss = loss;
  a.ticket = g_scl_ring.take();
  if (scl_mfma_ok(E, T)) {
    MVF_CHECK_ARG(((uintptr_t)emb % 16) == 0);
    constexpr int NW = 8;     // two waves per SIMD at up to 256 registers: room for the look-ahead copies (E = 256: without them)
    const size_t lds = stats_mfma_lds<NW>(T);
    const dim3 grid(ceil_div(M, RB)), block(64 * NW);
    if (E == 64) { set_lds(scl_stats_mfma_kernel<64, NW, true>, lds); hipLaunchKernelGGL((scl_stats_mfma_kernel<64, NW, true>), grid, block, lds, st, a); }
    else if (E == 128) { set_lds(scl_stats_mfma_kernel<128, NW, true>, lds); hipLaunchKernelGGL((scl_stats_mfma_kernel<128, NW, true>), grid, block, lds, st, a); }
    else { set_lds(scl_stats_mfma_kernel<256, NW, true>, lds); hipLaunchKernelGGL((scl_stats_mfma_kernel<256, NW, true>), grid, block, lds, st, a); }
  } else {
    const size_t lds = smem_bytes(E);
    set_lds(scl_stats_kernel, lds);
    hipLaunchKernelGGL(scl_stats_kernel, dim3(ceil_div(M, RB)), dim3(256), lds, st, a);
  }
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
  if (scl_mfma_ok(E, T)) {
    MVF_CHECK_ARG(((uintptr_t)emb % 16) == 0 && ((uintptr_t)dE % 16) == 0);
    const dim3 grid(ceil_div(rows, RB));
    constexpr size_t l64 = grad_mfma_lds<64, 8>(), l128 = grad_mfma_lds<128, 8>(), l256 = grad_mfma_lds<256, 8>();
    if (E == 64) { set_lds(scl_grad_mfma_kernel<64, 8, true>, l64); hipLaunchKernelGGL((scl_grad_mfma_kernel<64, 8, true>), grid, dim3(512), l64, st, a); }
    else if (E == 128) { set_lds(scl_grad_mfma_kernel<128, 8, true>, l128); hipLaunchKernelGGL((scl_grad_mfma_kernel<128, 8, true>), grid, dim3(512), l128, st, a); }
    else { set_lds(scl_grad_mfma_kernel<256, 8, false>, l256); hipLaunchKernelGGL((scl_grad_mfma_kernel<256, 8, false>), grid, dim3(512), l256, st, a); }
  } else {
    const size_t lds = smem_bytes(E);
    set_lds(scl_grad_kernel, lds);
    hipLaunchKernelGGL(scl_grad_kernel, dim3(ceil_div(rows, RB)), dim3(256), lds, st, a);
  }
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// which kernels mvf_scl_fwd / _bwd run (tests, A/B measurements): 0 = matrix-core form where it has an instantiation, 1 = scalar
extern "C" int mvf_scl_select(int form) {
  MVF_CHECK_ARG(form == 0 || form == 1);
  g_scl_form = form;
  return MVF_OK;
}
