// Temporal multi-head self-attention of the MV-Former encoder, forward and backward, fp32.
//   o = softmax(q k^T / sqrt(dk) + key_mask) v      per (clip, head), q/k/v = column slices of qkv[B*S, 3*Dm]
// Replaces `attention` + the head split/merge copies of MultiheadedAttention
// (CARL_MVF/models/utils.py:11-44, 88-104).  The [B,H,S,S] score matrix is never written to HBM:
// forward keeps a running (max, sum) per query (online softmax) and stores only the log-sum-exp; backward
// recomputes the probabilities from it (flash-style), two kernels: dQ (thread = query) and dK/dV
// (thread = key), so no atomics and bit-reproducible gradients.
//
// Round-1 structure: S = nst*T is 96 (config #2) .. 1440 and dk = 32, i.e. ~0.3 GFLOP per layer f+b: the
// kernels are latency-bound, so they use plain fp32 FMAs with the K/V (or Q/dO) tile broadcast from LDS
// (every lane reads the same address = conflict-free broadcast).  An MFMA version only pays at S >= ~512.
#include "common.h"
#include "mvf_hip_internal.h"

namespace {

constexpr int TQ = 128;   // threads per workgroup = rows owned (queries or keys); 64 measured no faster (dK/dV slower)
constexpr int TK = 64;    // rows of the streamed operand per LDS tile
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

template <int DK>
__global__ __launch_bounds__(TQ) void tattn_fwd_kernel(TAttnArgs a) {
  __shared__ __attribute__((aligned(16))) float sk[TK][DK];
  __shared__ __attribute__((aligned(16))) float sv[TK][DK];
  __shared__ float sm[TK];
  const int b = blockIdx.z, h = blockIdx.y;
  const int qi = blockIdx.x * TQ + threadIdx.x;
  const bool qvalid = qi < a.S;
  const size_t ld = (size_t)3 * a.Dm;
  const float* base = a.qkv + (size_t)b * a.S * ld + h * DK;
  float q[DK], o[DK];
  {
    const float* qp = base + (size_t)min(qi, a.S - 1) * ld;
#pragma unroll
    for (int d = 0; d < DK; d += 4) {
      const float4 v = *reinterpret_cast<const float4*>(qp + d);
      q[d] = v.x * a.scale_log2; q[d + 1] = v.y * a.scale_log2; q[d + 2] = v.z * a.scale_log2; q[d + 3] = v.w * a.scale_log2;
    }
  }
#pragma unroll
  for (int d = 0; d < DK; ++d) o[d] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  for (int k0 = 0; k0 < a.S; k0 += TK) {
    __syncthreads();
    for (int i = threadIdx.x; i < TK * (DK / 4); i += TQ) {
      const int r = i / (DK / 4), c = (i % (DK / 4)) * 4;
      const int key = k0 + r;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (key < a.S) {
        kv = *reinterpret_cast<const float4*>(base + (size_t)key * ld + a.Dm + c);
        vv = *reinterpret_cast<const float4*>(base + (size_t)key * ld + 2 * a.Dm + c);
      }
      *reinterpret_cast<float4*>(&sk[r][c]) = kv;
      *reinterpret_cast<float4*>(&sv[r][c]) = vv;
    }
    if (threadIdx.x < TK) {
      const int key = k0 + threadIdx.x;
      sm[threadIdx.x] = (key < a.S && (a.mask == nullptr || a.mask[(size_t)b * a.Sm + key % a.Sm] != 0.f)) ? 1.f : 0.f;
    }
    __syncthreads();
    for (int c0 = 0; c0 < TK; c0 += 8) {
      float s[8];
      float mx = -1e30f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < DK; ++d) acc += q[d] * sk[c0 + j][d];
        s[j] = sm[c0 + j] != 0.f ? acc : -1e30f;
        mx = fmaxf(mx, s[j]);
      }
      const float m_new = fmaxf(m_run, mx);
      const float alpha = exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DK; ++d) o[d] *= alpha;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float p = exp2f(s[j] - m_new);
        l_run += p;
#pragma unroll
        for (int d = 0; d < DK; ++d) o[d] += p * sv[c0 + j][d];
      }
      m_run = m_new;
    }
  }
  if (qvalid) {
    const float inv = 1.f / l_run;
    float* op = a.o + ((size_t)b * a.S + qi) * a.Dm + h * DK;
#pragma unroll
    for (int d = 0; d < DK; d += 4)
      *reinterpret_cast<float4*>(op + d) = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
    a.lse[((size_t)b * a.H + h) * a.S + qi] = m_run + log2f(l_run);
  }
}

// dQ: thread = query, streams K/V tiles
template <int DK>
__global__ __launch_bounds__(TQ) void tattn_bwd_dq_kernel(TAttnArgs a) {
  __shared__ __attribute__((aligned(16))) float sk[TK][DK];
  __shared__ __attribute__((aligned(16))) float sv[TK][DK];
  __shared__ float sm[TK];
  const int b = blockIdx.z, h = blockIdx.y;
  const int qi = blockIdx.x * TQ + threadIdx.x;
  const bool qvalid = qi < a.S;
  const int qc = min(qi, a.S - 1);
  const size_t ld = (size_t)3 * a.Dm;
  const float* base = a.qkv + (size_t)b * a.S * ld + h * DK;
  float q[DK], go[DK], dq[DK];
  float delta = 0.f;
  {
    const float* qp = base + (size_t)qc * ld;
    const float* gp = a.d_o + ((size_t)b * a.S + qc) * a.Dm + h * DK;
    const float* op = a.o + ((size_t)b * a.S + qc) * a.Dm + h * DK;
#pragma unroll
    for (int d = 0; d < DK; ++d) {
      q[d] = qp[d] * a.scale_log2;
      go[d] = gp[d];
      delta += gp[d] * op[d];
      dq[d] = 0.f;
    }
  }
  const float lse = a.lse[((size_t)b * a.H + h) * a.S + qc];
  for (int k0 = 0; k0 < a.S; k0 += TK) {
    __syncthreads();
    for (int i = threadIdx.x; i < TK * (DK / 4); i += TQ) {
      const int r = i / (DK / 4), c = (i % (DK / 4)) * 4;
      const int key = k0 + r;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (key < a.S) {
        kv = *reinterpret_cast<const float4*>(base + (size_t)key * ld + a.Dm + c);
        vv = *reinterpret_cast<const float4*>(base + (size_t)key * ld + 2 * a.Dm + c);
      }
      *reinterpret_cast<float4*>(&sk[r][c]) = kv;
      *reinterpret_cast<float4*>(&sv[r][c]) = vv;
    }
    if (threadIdx.x < TK) {
      const int key = k0 + threadIdx.x;
      sm[threadIdx.x] = (key < a.S && (a.mask == nullptr || a.mask[(size_t)b * a.Sm + key % a.Sm] != 0.f)) ? 1.f : 0.f;
    }
    __syncthreads();
    for (int j = 0; j < TK; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < DK; ++d) { s += q[d] * sk[j][d]; dp += go[d] * sv[j][d]; }
      const float p = sm[j] != 0.f ? exp2f(s - lse) : 0.f;
      const float ds = p * (dp - delta) * a.scale;
#pragma unroll
      for (int d = 0; d < DK; ++d) dq[d] += ds * sk[j][d];
    }
  }
  if (qvalid) {
    float* dp = a.dqkv + ((size_t)b * a.S + qi) * ld + h * DK;
#pragma unroll
    for (int d = 0; d < DK; d += 4) *reinterpret_cast<float4*>(dp + d) = make_float4(dq[d], dq[d + 1], dq[d + 2], dq[d + 3]);
  }
}

// dK, dV: thread = key, streams Q / dO tiles
template <int DK>
__global__ __launch_bounds__(TQ) void tattn_bwd_dkv_kernel(TAttnArgs a) {
  __shared__ __attribute__((aligned(16))) float sq[TK][DK];
  __shared__ __attribute__((aligned(16))) float sg[TK][DK];
  __shared__ float slse[TK], sdel[TK];
  const int b = blockIdx.z, h = blockIdx.y;
  const int ki = blockIdx.x * TQ + threadIdx.x;
  const bool kvalid = ki < a.S;
  const int kc = min(ki, a.S - 1);
  const size_t ld = (size_t)3 * a.Dm;
  const float* base = a.qkv + (size_t)b * a.S * ld + h * DK;
  float k[DK], v[DK], dk[DK], dv[DK];
  {
    const float* kp = base + (size_t)kc * ld + a.Dm;
    const float* vp = base + (size_t)kc * ld + 2 * a.Dm;
#pragma unroll
    for (int d = 0; d < DK; ++d) { k[d] = kp[d] * a.scale_log2; v[d] = vp[d]; dk[d] = 0.f; dv[d] = 0.f; }
  }
  const bool keep = kvalid && (a.mask == nullptr || a.mask[(size_t)b * a.Sm + kc % a.Sm] != 0.f);
  for (int q0 = 0; q0 < a.S; q0 += TK) {
    __syncthreads();
    for (int i = threadIdx.x; i < TK * (DK / 4); i += TQ) {
      const int r = i / (DK / 4), c = (i % (DK / 4)) * 4;
      const int qi = q0 + r;
      float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), gv = qv;
      if (qi < a.S) {
        qv = *reinterpret_cast<const float4*>(base + (size_t)qi * ld + c);
        gv = *reinterpret_cast<const float4*>(a.d_o + ((size_t)b * a.S + qi) * a.Dm + h * DK + c);
      }
      *reinterpret_cast<float4*>(&sq[r][c]) = qv;
      *reinterpret_cast<float4*>(&sg[r][c]) = gv;
    }
    if (threadIdx.x < TK) {
      const int qi = q0 + threadIdx.x;
      float del = 0.f, l = 1e30f;  // lse = +big -> p = 0 for out-of-range queries
      if (qi < a.S) {
        const float* gp = a.d_o + ((size_t)b * a.S + qi) * a.Dm + h * DK;
        const float* op = a.o + ((size_t)b * a.S + qi) * a.Dm + h * DK;
#pragma unroll
        for (int d = 0; d < DK; ++d) del += gp[d] * op[d];
        l = a.lse[((size_t)b * a.H + h) * a.S + qi];
      }
      sdel[threadIdx.x] = del;
      slse[threadIdx.x] = l;
    }
    __syncthreads();
    if (keep) {
      for (int j = 0; j < TK; ++j) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < DK; ++d) { s += sq[j][d] * k[d]; dp += sg[j][d] * v[d]; }
        const float p = exp2f(s - slse[j]);
        const float ds = p * (dp - sdel[j]) * a.scale;
#pragma unroll
        for (int d = 0; d < DK; ++d) { dv[d] += p * sg[j][d]; dk[d] += ds * sq[j][d]; }
      }
    }
  }
  if (kvalid) {
    float* dkp = a.dqkv + ((size_t)b * a.S + ki) * ld + a.Dm + h * DK;
    float* dvp = a.dqkv + ((size_t)b * a.S + ki) * ld + 2 * a.Dm + h * DK;
#pragma unroll
    for (int d = 0; d < DK; d += 4) {
      *reinterpret_cast<float4*>(dkp + d) = make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]);
      *reinterpret_cast<float4*>(dvp + d) = make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]);
    }
  }
}

template <int DK>
int run(int which, const TAttnArgs& a, hipStream_t st) {
  dim3 grid(ceil_div(a.S, TQ), a.H, a.B);
  if (which == 0) {
    hipLaunchKernelGGL(tattn_fwd_kernel<DK>, grid, dim3(TQ), 0, st, a);
  } else {
    hipLaunchKernelGGL(tattn_bwd_dq_kernel<DK>, grid, dim3(TQ), 0, st, a);
    hipLaunchKernelGGL(tattn_bwd_dkv_kernel<DK>, grid, dim3(TQ), 0, st, a);
  }
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

int dispatch(int which, int dk, const TAttnArgs& a, hipStream_t st) {
  switch (dk) {
    case 8: return run<8>(which, a, st);
    case 16: return run<16>(which, a, st);
    case 32: return run<32>(which, a, st);
    case 64: return run<64>(which, a, st);
  }
  return MVF_ERR_UNSUPPORTED;
}

}  // namespace

int g_tattn_scalar = 0;
// 1 = always the scalar-FMA kernels of this file (cross-check / A-B measurements), 0 = matrix-core kernels when dk % 16 == 0
extern "C" int mvf_tattn_select(int scalar_only) {
  g_tattn_scalar = scalar_only != 0;
  return MVF_OK;
}

extern "C" int mvf_tattn_fwd(const float* qkv, const float* mask, int mask_len, float* o, float* lse, int B, int S, int H,
                             int Dm, hipStream_t st) {
  MVF_CHECK_ARG(qkv && o && lse && B > 0 && S > 0 && H > 0 && Dm % H == 0 && Dm % 4 == 0);
  MVF_CHECK_ARG(mask == nullptr || (mask_len > 0 && S % mask_len == 0));
  MVF_CHECK_ARG(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)o & 15) == 0);
  if ((Dm / H) % 16 == 0 && g_tattn_scalar == 0) {   // matrix-core path (head_attn_mfma.hip)
    const int rc = mvf_tattn_mfma(0, qkv, mask, mask_len, o, lse, nullptr, nullptr, B, S, H, Dm, st);
    if (rc != MVF_ERR_UNSUPPORTED) return rc;
  }
  TAttnArgs a{};
  a.qkv = qkv; a.mask = mask; a.Sm = mask != nullptr ? mask_len : S; a.o = o; a.lse = lse; a.B = B; a.S = S; a.H = H; a.Dm = Dm;
  const int dk = Dm / H;
  a.scale = 1.0f / sqrtf((float)dk);
  a.scale_log2 = a.scale * LOG2E;
  return dispatch(0, dk, a, st);
}

extern "C" int mvf_tattn_bwd(const float* qkv, const float* mask, int mask_len, const float* o, const float* lse,
                             const float* d_o, float* dqkv, int B, int S, int H, int Dm, hipStream_t st) {
  MVF_CHECK_ARG(qkv && o && lse && d_o && dqkv && B > 0 && S > 0 && H > 0 && Dm % H == 0 && Dm % 4 == 0);
  MVF_CHECK_ARG(mask == nullptr || (mask_len > 0 && S % mask_len == 0));
  MVF_CHECK_ARG(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)dqkv & 15) == 0 && ((uintptr_t)d_o & 15) == 0);
  if ((Dm / H) % 16 == 0 && g_tattn_scalar == 0) {
    const int rc = mvf_tattn_mfma(1, qkv, mask, mask_len, const_cast<float*>(o), const_cast<float*>(lse), d_o, dqkv, B, S, H, Dm, st);
    if (rc != MVF_ERR_UNSUPPORTED) return rc;
  }
  TAttnArgs a{};
  a.qkv = qkv; a.mask = mask; a.Sm = mask != nullptr ? mask_len : S; a.o = const_cast<float*>(o); a.lse = const_cast<float*>(lse);
  a.d_o = d_o; a.dqkv = dqkv; a.B = B; a.S = S; a.H = H; a.Dm = Dm;
  const int dk = Dm / H;
  a.scale = 1.0f / sqrtf((float)dk);
  a.scale_log2 = a.scale * LOG2E;
  return dispatch(1, dk, a, st);
}
