// Temporal multi-head self-attention of the MV-Former encoder on the fp32 matrix cores, forward and backward.
//   o = softmax(q k^T / sqrt(dk) + key_mask) v      per (clip, head), q/k/v = column slices of qkv[B*S, 3*Dm]
// Same contract as the scalar kernels of head_attn.hip (which remain the fallback for dk % 16 != 0): replaces
// `attention` + the head split/merge copies of MultiheadedAttention (CARL_MVF/models/utils.py:11-44, 88-104); the
// [B,H,S,S] score matrix never exists; forward stores only the log2-domain log-sum-exp per query, backward recomputes
// the probabilities from it; dQ and dK/dV are separate kernels (owner-computes, no atomics, bit-reproducible).
//
// gfx950 design: v_mfma_f32_16x16x4_f32 (exact fp32).  A workgroup = 4 waves = 64 rows of the OWNED index (queries in
// forward / dQ, keys in dK/dV), one 16-row tile per wave; the STREAMED index (keys, resp. queries) passes through LDS
// in blocks of 64 rows.  With the owned index on the MFMA column (lane & 15) and the streamed index on its rows, every
// product the backward needs chains without a transpose:
//   forward / dQ (owned = query q, streamed = key k):
//        S^T[k][q]  = K . Q^T           A = K rows (LDS row fragment),  B = Q rows (registers)
//        dP^T[k][q] = V . dO^T          A = V rows,                     B = dO rows (registers)
//        O^T[d][q]  += V^T . P^T        A = V[k][d] (LDS scalars),      B = P^T accumulator as it stands
//        dQ^T[d][q] += K^T . dS^T       A = K[k][d] (LDS scalars),      B = dS^T accumulator as it stands
//   dK/dV (owned = key k, streamed = query q):
//        S[q][k]    = Q . K^T           A = Q rows (LDS),               B = K rows (registers)
//        dP[q][k]   = dO . V^T          A = dO rows (LDS),              B = V rows (registers)
//        dV^T[d][k] += dO^T . P         A = dO[q][d] (LDS scalars),     B = P accumulator
//        dK^T[d][k] += Q^T . dS         A = Q[q][d] (LDS scalars),      B = dS accumulator
// (the 16x16x4 layout: lane (c = lane & 15, g = lane >> 4) holds A[row c][k-slot g], B[k-slot g][col c] and
// D[row 4g + r][col c] in register r -- so a D tile is directly a B operand whose k index is D's row index.)
// Each streamed block lives in LDS twice: rows of dk floats with the 16-byte chunks XOR-swizzled by the row (row
// fragments, ds_read_b128, conflict-free) and rows of dk + 4 floats (scalar column reads, conflict-free).
#include "common.h"
#include "mvf_hip_internal.h"

namespace {

constexpr int BLK = 64;   // rows of the streamed operand per LDS block; also rows owned per workgroup (4 waves x 16)
constexpr float LOG2E = 1.4426950408889634f;

struct TAttnArgs {
  const float* qkv;    // [B*S, 3*Dm]
  const float* mask;   // [B, Sm] (1 keep / 0 masked) or null; key s reads column s % Sm (Sm = S: a plain key mask;
  int Sm;              // Sm = T for the joint entity x frame sequence, whose keys (j, t) share the frame mask [B, T])
  float* o;            // [B*S, Dm]
  float* lse;          // [B, H, S]  log2-domain log-sum-exp of scaled scores
  const float* d_o;    // [B*S, Dm]
  float* dqkv;         // [B*S, 3*Dm]
  int B, S, H, Dm;
  float scale_log2;    // dk^-0.5 * log2(e)
  float scale;         // dk^-0.5
};

// LDS images of one [64][DK] block
template <int DK>
struct Img {
  static constexpr int NCH = DK / 4;            // 16-byte chunks per row
  static constexpr int FROW = DK;               // floats per row of the fragment image
  static constexpr int SROW = DK + 4;           // floats per row of the scalar image
  static constexpr int FRAG_F = BLK * FROW, SCAL_F = BLK * SROW;
  // float offset of chunk ch of row r in the fragment image
  static __device__ __forceinline__ int frag(int r, int ch) { return r * FROW + ((ch ^ (r & (NCH - 1))) << 2); }
  static __device__ __forceinline__ int scal(int r, int d) { return r * SROW + d; }
};

// stage rows [r0, r0+64) of a [*, ld] matrix (column offset already applied to `src`) into both images; rows >= R -> 0
template <int DK>
__device__ __forceinline__ void stage_block(const float* src, size_t ld, int r0, int R, float* fimg, float* simg) {
  using I = Img<DK>;
  for (int i = threadIdx.x; i < BLK * I::NCH; i += 256) {
    const int r = i / I::NCH, ch = i % I::NCH;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < R) v = *reinterpret_cast<const float4*>(src + (size_t)(r0 + r) * ld + ch * 4);
    *reinterpret_cast<float4*>(fimg + I::frag(r, ch)) = v;
    *reinterpret_cast<float4*>(simg + I::scal(r, ch * 4)) = v;
  }
}

// row fragment of the OWNED operand straight from global memory: lane (c, g) holds X[row c][4*(g + 4h) .. +3], h < DK/16
template <int DK>
__device__ __forceinline__ void load_rowfrag(const float* src, size_t ld, int row, int g, f32x4_t (&f)[DK / 16]) {
#pragma unroll
  for (int h = 0; h < DK / 16; ++h) f[h] = *reinterpret_cast<const f32x4_t*>(src + (size_t)row * ld + 4 * (g + 4 * h));
}

// D[rows = streamed tile t][cols = owned] += X_lds(tile t rows) . Y_reg^T over the DK-long inner index
template <int DK>
__device__ __forceinline__ f32x4_t tile_dot(const float* fimg, int t, int c, int g, const f32x4_t (&y)[DK / 16]) {
  using I = Img<DK>;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int h = 0; h < DK / 16; ++h) {
    const f32x4_t x = *reinterpret_cast<const f32x4_t*>(fimg + I::frag(t * 16 + c, g + 4 * h));
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], y[h][e], acc, 0, 0, 0);
  }
  return acc;
}

// ------------------------------------------------------------------------------------------------ forward
template <int DK>
__global__ __launch_bounds__(256) void tattn_mfma_fwd_kernel(TAttnArgs a) {
  using I = Img<DK>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* kf = lds;                       // K fragment image
  float* vs = kf + I::FRAG_F;            // V scalar image
  float* vdummy = vs + I::SCAL_F;        // (K scalar / V fragment images are not needed in the forward)
  float* sm = vdummy;                    // [64] key mask of the block
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const size_t ld = (size_t)3 * a.Dm;
  const float* base = a.qkv + (size_t)b * a.S * ld + h * DK;
  const int q = blockIdx.x * BLK + wave * 16 + c;          // this lane's query (MFMA column)
  const int qc = min(q, a.S - 1);
  f32x4_t qf[DK / 16];
  load_rowfrag<DK>(base, ld, qc, g, qf);
  f32x4_t o[DK / 16];
#pragma unroll
  for (int dt = 0; dt < DK / 16; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;

  for (int k0 = 0; k0 < a.S; k0 += BLK) {
    __syncthreads();
    // K: fragment image only; V: scalar image only
    for (int i = threadIdx.x; i < BLK * I::NCH; i += 256) {
      const int r = i / I::NCH, ch = i % I::NCH;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (k0 + r < a.S) {
        kv = *reinterpret_cast<const float4*>(base + (size_t)(k0 + r) * ld + a.Dm + ch * 4);
        vv = *reinterpret_cast<const float4*>(base + (size_t)(k0 + r) * ld + 2 * a.Dm + ch * 4);
      }
      *reinterpret_cast<float4*>(kf + I::frag(r, ch)) = kv;
      *reinterpret_cast<float4*>(vs + I::scal(r, ch * 4)) = vv;
    }
    if (threadIdx.x < BLK) {
      const int key = k0 + threadIdx.x;
      sm[threadIdx.x] = (key < a.S && (a.mask == nullptr || a.mask[(size_t)b * a.Sm + key % a.Sm] != 0.f)) ? 1.f : 0.f;
    }
    __syncthreads();
    f32x4_t s[4];
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = tile_dot<DK>(kf, t, c, g, qf);                // S^T[key 16t + 4g + r][query c]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[t][r] = sm[t * 16 + 4 * g + r] != 0.f ? s[t][r] * a.scale_log2 : -1e30f;
        mx = fmaxf(mx, s[t][r]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);
    float ls = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // a fully masked prefix keeps m_new = -1e30: exp2(0) = 1 would count masked keys, so mask explicitly
        const float p = s[t][r] > -1e29f ? exp2f(s[t][r] - m_new) : 0.f;
        s[t][r] = p;
        ls += p;
      }
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * alpha + ls;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < DK / 16; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
    // O^T[d][q] += sum_key V[key][d] P[q][key]; MFMA (t, r): k-slot g <-> key 16t + 4g + r
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* vr = vs + I::scal(t * 16 + 4 * g + r, c);
#pragma unroll
        for (int dt = 0; dt < DK / 16; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[dt * 16], s[t][r], o[dt], 0, 0, 0);
      }
  }
  if (q < a.S) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* op = a.o + ((size_t)b * a.S + q) * a.Dm + h * DK;
#pragma unroll
    for (int dt = 0; dt < DK / 16; ++dt)
      *reinterpret_cast<float4*>(op + dt * 16 + 4 * g) =
          make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
    if (g == 0) a.lse[((size_t)b * a.H + h) * a.S + q] = m_run + log2f(l_run);
  }
}

// ------------------------------------------------------------------------------------------------ dQ
template <int DK>
__device__ __forceinline__ void tattn_dq_body(const TAttnArgs& a, int h, float* lds) {
  using I = Img<DK>;
  float* kf = lds;                       // K fragment image (S^T)
  float* ks = kf + I::FRAG_F;            // K scalar image (dQ^T)
  float* vf = ks + I::SCAL_F;            // V fragment image (dP^T)
  float* sm = vf + I::FRAG_F;            // [64] key mask
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int b = blockIdx.z;
  const size_t ld = (size_t)3 * a.Dm;
  const float* base = a.qkv + (size_t)b * a.S * ld + h * DK;
  const int q = blockIdx.x * BLK + wave * 16 + c;
  const int qc = min(q, a.S - 1);
  f32x4_t qf[DK / 16], gf[DK / 16], of[DK / 16];
  load_rowfrag<DK>(base, ld, qc, g, qf);
  load_rowfrag<DK>(a.d_o + (size_t)b * a.S * a.Dm + h * DK, a.Dm, qc, g, gf);
  load_rowfrag<DK>(a.o + (size_t)b * a.S * a.Dm + h * DK, a.Dm, qc, g, of);
  float delta = 0.f;                     // sum_d dO[q][d] O[q][d]: this lane holds DK/4 of the d's, the k-groups the rest
#pragma unroll
  for (int hh = 0; hh < DK / 16; ++hh)
#pragma unroll
    for (int e = 0; e < 4; ++e) delta += gf[hh][e] * of[hh][e];
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);
  const float lse = a.lse[((size_t)b * a.H + h) * a.S + qc];
  f32x4_t dq[DK / 16];
#pragma unroll
  for (int dt = 0; dt < DK / 16; ++dt) dq[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < a.S; k0 += BLK) {
    __syncthreads();
    stage_block<DK>(base + a.Dm, ld, k0, a.S, kf, ks);
    for (int i = threadIdx.x; i < BLK * I::NCH; i += 256) {
      const int r = i / I::NCH, ch = i % I::NCH;
      float4 vv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + r < a.S) vv = *reinterpret_cast<const float4*>(base + (size_t)(k0 + r) * ld + 2 * a.Dm + ch * 4);
      *reinterpret_cast<float4*>(vf + I::frag(r, ch)) = vv;
    }
    if (threadIdx.x < BLK) {
      const int key = k0 + threadIdx.x;
      sm[threadIdx.x] = (key < a.S && (a.mask == nullptr || a.mask[(size_t)b * a.Sm + key % a.Sm] != 0.f)) ? 1.f : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4_t s = tile_dot<DK>(kf, t, c, g, qf);     // S^T[key][q]
      const f32x4_t dp = tile_dot<DK>(vf, t, c, g, gf);    // dP^T[key][q]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = sm[t * 16 + 4 * g + r] != 0.f ? exp2f(s[r] * a.scale_log2 - lse) : 0.f;
        const float ds = p * (dp[r] - delta) * a.scale;
        const float* kr = ks + I::scal(t * 16 + 4 * g + r, c);
#pragma unroll
        for (int dt = 0; dt < DK / 16; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[dt * 16], ds, dq[dt], 0, 0, 0);
      }
    }
  }
  if (q < a.S) {
    float* dp = a.dqkv + ((size_t)b * a.S + q) * ld + h * DK;
#pragma unroll
    for (int dt = 0; dt < DK / 16; ++dt)
      *reinterpret_cast<float4*>(dp + dt * 16 + 4 * g) = make_float4(dq[dt][0], dq[dt][1], dq[dt][2], dq[dt][3]);
  }
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <int DK>
__device__ __forceinline__ void tattn_dkv_body(const TAttnArgs& a, int h, float* lds) {
  using I = Img<DK>;
  float* qfi = lds;                      // Q fragment image (S)
  float* qsi = qfi + I::FRAG_F;          // Q scalar image (dK^T)
  float* gfi = qsi + I::SCAL_F;          // dO fragment image (dP)
  float* gsi = gfi + I::FRAG_F;          // dO scalar image (dV^T)
  float* slse = gsi + I::SCAL_F;         // [64] lse of the block's queries (+big for rows >= S -> p = 0)
  float* sdel = slse + BLK;              // [64] delta of the block's queries
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int b = blockIdx.z;
  const size_t ld = (size_t)3 * a.Dm;
  const float* base = a.qkv + (size_t)b * a.S * ld + h * DK;
  const float* gbase = a.d_o + (size_t)b * a.S * a.Dm + h * DK;
  const float* obase = a.o + (size_t)b * a.S * a.Dm + h * DK;
  const int key = blockIdx.x * BLK + wave * 16 + c;        // this lane's key (MFMA column)
  const int kc = min(key, a.S - 1);
  f32x4_t kfr[DK / 16], vfr[DK / 16];
  load_rowfrag<DK>(base + a.Dm, ld, kc, g, kfr);
  load_rowfrag<DK>(base + 2 * a.Dm, ld, kc, g, vfr);
  const bool keep = key < a.S && (a.mask == nullptr || a.mask[(size_t)b * a.Sm + kc % a.Sm] != 0.f);
  f32x4_t dk[DK / 16], dv[DK / 16];
#pragma unroll
  for (int dt = 0; dt < DK / 16; ++dt) { dk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }

  for (int q0 = 0; q0 < a.S; q0 += BLK) {
    __syncthreads();
    stage_block<DK>(base, ld, q0, a.S, qfi, qsi);
    stage_block<DK>(gbase, a.Dm, q0, a.S, gfi, gsi);
    if (threadIdx.x < BLK) {
      const int qi = q0 + threadIdx.x;
      float del = 0.f, l = 1e30f;
      if (qi < a.S) {
        const float* gp = gbase + (size_t)qi * a.Dm;
        const float* op = obase + (size_t)qi * a.Dm;
#pragma unroll
        for (int d = 0; d < DK; d += 4) {
          const float4 gv = *reinterpret_cast<const float4*>(gp + d), ov = *reinterpret_cast<const float4*>(op + d);
          del += gv.x * ov.x + gv.y * ov.y + gv.z * ov.z + gv.w * ov.w;
        }
        l = a.lse[((size_t)b * a.H + h) * a.S + qi];
      }
      sdel[threadIdx.x] = del;
      slse[threadIdx.x] = l;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4_t s = tile_dot<DK>(qfi, t, c, g, kfr);   // S[query 16t + 4g + r][key c]
      const f32x4_t dp = tile_dot<DK>(gfi, t, c, g, vfr);  // dP[query][key]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = t * 16 + 4 * g + r;
        const float p = keep ? exp2f(s[r] * a.scale_log2 - slse[qq]) : 0.f;
        const float ds = p * (dp[r] - sdel[qq]) * a.scale;
        const float* gr = gsi + I::scal(qq, c);
        const float* qr = qsi + I::scal(qq, c);
#pragma unroll
        for (int dt = 0; dt < DK / 16; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gr[dt * 16], p, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qr[dt * 16], ds, dk[dt], 0, 0, 0);
        }
      }
    }
  }
  if (key < a.S) {
    float* dkp = a.dqkv + ((size_t)b * a.S + key) * ld + a.Dm + h * DK;
    float* dvp = a.dqkv + ((size_t)b * a.S + key) * ld + 2 * a.Dm + h * DK;
#pragma unroll
    for (int dt = 0; dt < DK / 16; ++dt) {
      *reinterpret_cast<float4*>(dkp + dt * 16 + 4 * g) = make_float4(dk[dt][0], dk[dt][1], dk[dt][2], dk[dt][3]);
      *reinterpret_cast<float4*>(dvp + dt * 16 + 4 * g) = make_float4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]);
    }
  }
}

// dQ and dK/dV of one attention in ONE launch: the two are independent (both read q, k, v, o, dO, lse), blockIdx.y < H takes the
// query-owning form, the rest the key-owning form
template <int DK>
__global__ __launch_bounds__(256) void tattn_mfma_bwd_kernel(TAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.y < a.H) tattn_dq_body<DK>(a, blockIdx.y, lds);
  else tattn_dkv_body<DK>(a, blockIdx.y - a.H, lds);
}

template <int DK>
int run(int which, const TAttnArgs& a, hipStream_t st) {
  using I = Img<DK>;
  dim3 grid(ceil_div(a.S, BLK), a.H, a.B);
  const size_t lds_f = (size_t)(I::FRAG_F + I::SCAL_F + BLK) * 4;
  const size_t lds_q = (size_t)(2 * I::FRAG_F + I::SCAL_F + BLK) * 4;
  const size_t lds_kv = (size_t)(2 * I::FRAG_F + 2 * I::SCAL_F + 2 * BLK) * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tattn_mfma_bwd_kernel<DK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
    attr = true;
  }
  if (which == 0) {
    hipLaunchKernelGGL(tattn_mfma_fwd_kernel<DK>, grid, dim3(256), lds_f, st, a);
  } else {
    (void)lds_q;
    hipLaunchKernelGGL(tattn_mfma_bwd_kernel<DK>, dim3(grid.x, 2 * a.H, grid.z), dim3(256), lds_kv, st, a);
  }
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

}  // namespace

// which: 0 forward, 1 backward; returns MVF_ERR_UNSUPPORTED when dk is not 16, 32 or 64 (scalar kernels take over)
int mvf_tattn_mfma(int which, const float* qkv, const float* mask, int mask_len, float* o, float* lse, const float* d_o,
                   float* dqkv, int B, int S, int H, int Dm, hipStream_t st) {
  TAttnArgs a{};
  a.qkv = qkv; a.mask = mask; a.Sm = mask != nullptr ? mask_len : S; a.o = o; a.lse = lse; a.d_o = d_o; a.dqkv = dqkv; a.B = B; a.S = S; a.H = H; a.Dm = Dm;
  const int dk = Dm / H;
  a.scale = 1.0f / sqrtf((float)dk);
  a.scale_log2 = a.scale * LOG2E;
  switch (dk) {
    case 16: return run<16>(which, a, st);
    case 32: return run<32>(which, a, st);
    case 64: return run<64>(which, a, st);
  }
  return MVF_ERR_UNSUPPORTED;
}
