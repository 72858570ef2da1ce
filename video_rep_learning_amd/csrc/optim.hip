// Fused global-norm gradient clip + Adam (L2 weight decay) over ONE flat fp32 parameter buffer.
// Reference: torch.nn.utils.clip_grad_norm_(model.parameters(), GRAD_CLIP) followed by
// torch.optim.Adam(lr, betas=(0.9, 0.999), weight_decay) -- CARL_MVF/train.py:124-133,147-149,
// utils/optimizer.py:60-66.  The head's 4.8 M trainable parameters live in one contiguous buffer (the DDP
// gradient bucket is the matching flat gradient buffer), so the whole optimizer is two launches instead
// of ~10 per parameter tensor.  HBM-bound: 4 streams read + 3 written per element, float4 accesses.
#include "common.h"
#include "mvf_hip_internal.h"

namespace {

#ifdef MVF_NT_OFF      // build flag, A/B measurements only
#define NT_LD(p) (*(p))
#define NT_ST(p, v) (*(p) = (v))
#else
#define NT_LD(p) __builtin_nontemporal_load(p)
#define NT_ST(p, v) __builtin_nontemporal_store((v), (p))
#endif

// ticket of the one-launch norm (zero-initialised with the code object, reset by the last arriver; one optimizer step at a time)
__device__ unsigned g_sqnorm_ticket[TICKET_SLOTS];
TicketRing g_sqnorm_ring;

// norm_out[0] = sqrt(sum g^2 + extra_sq[0]): every workgroup stores its partial sum, the last one to arrive adds them in a fixed
// order (reproducible) -- one launch instead of a partial and a final one.
// norm_out[1] += 1 when that norm is not finite: the count of optimizer steps adam_kernel has SKIPPED (what
// torch.cuda.amp.GradScaler.step does for the reference's fp16 path, train.py:127-133: no update, no step count)
constexpr int SQ_TH = 1024;
__global__ __launch_bounds__(SQ_TH) void sqnorm_kernel(const float* __restrict__ g, size_t n, float* __restrict__ part,
                                                       const float* __restrict__ extra_sq, float* __restrict__ norm_out, int ticket) {
  __shared__ float red[SQ_TH / 64];
  float s = 0.f;
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * SQ_TH;
  // a thread's loads go out four at a time (as a load -> use loop the ~18 float4 per thread were as many dependent round trips)
  for (size_t i0 = (size_t)blockIdx.x * SQ_TH + threadIdx.x; i0 < n4; i0 += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t i = i0 + u * stride;
      v[u] = reinterpret_cast<const float4*>(g)[i < n4 ? i : n4 - 1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * stride < n4) s += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += SQ_TH) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < SQ_TH / 64; ++w) a += red[w];
    part[blockIdx.x] = a;
  }
  if (!last_arriver(&g_sqnorm_ticket[ticket], gridDim.x)) return;
  float t = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += SQ_TH) t += part[i];
  t = wave_sum(t);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < SQ_TH / 64; ++w) a += red[w];
    const float nrm = sqrtf(a + (extra_sq ? extra_sq[0] : 0.f));
    norm_out[0] = nrm;
    if (!isfinite(nrm)) norm_out[1] += 1.f;
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt, float clip,
                                                   const float* __restrict__ norm, float gscale, int step_no, int zero_grad) {
  float coef = gscale;
  if (norm != nullptr) {
    // a NaN / Inf gradient norm poisons parameters and both moments for good (fminf(1, clip / NaN) = 1): skip the step,
    // and leave it out of the bias-correction step count like a GradScaler-skipped step
    if (!isfinite(norm[0])) {
      if (zero_grad)      // the rejected gradient is dropped all the same
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) g[i] = 0.f;
      return;
    }
    const float skipped = norm[1];
    if (skipped > 0.f) {
      const double eff = fmax((double)step_no - (double)skipped, 1.0);
      bc1 = (float)(1.0 - pow((double)b1, eff));
      bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, eff));
    }
    if (clip > 0.f) coef *= fminf(1.f, clip / (norm[0] * gscale + 1e-6f));
  }
  const float step = lr / bc1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    // moments and gradients are touched once per step: non-temporal, so that 134 MB of optimizer traffic do not push the backbone
    // GEMMs' operands (the next batch's forward runs beside this kernel) out of the L2; the parameters are re-read by the next
    // forward's kernels and keep the default policy
    const float pi = p[i];
    const float gi = NT_LD(g + i) * coef + wd * pi;
    const float mi = b1 * NT_LD(m + i) + (1.f - b1) * gi;
    const float vi = b2 * NT_LD(v + i) + (1.f - b2) * gi * gi;
    NT_ST(m + i, mi);
    NT_ST(v + i, vi);
    p[i] = pi - step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    if (zero_grad) NT_ST(g + i, 0.f);      // the next step's zero_grad() in the pass that has the gradient in registers anyway
  }
}

// The same update on FEW compute units: 1 024-thread workgroups that each take a CU to themselves (launched with a large dynamic LDS
// request), float4 accesses, two batches of loads in flight per thread.  The wide form puts 2 048 small workgroups on every CU that
// falls free; beside the frozen backbone's persistent GEMM (one 512-register workgroup per CU) each of those CUs is then lost to the
// GEMM until its last optimizer workgroup has gone -- the optimizer's 37 us cost the pipelined step 0.10 - 0.13 ms.  Narrow, it
// holds `width` CUs for longer instead (mvf_optim_set_width).  All four pointers 16-byte aligned, n4 = n / 4 vectors.
constexpr int AD_TH = 1024;
__global__ __launch_bounds__(AD_TH) void adam_narrow_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                            float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                            float wd, float bc1, float bc2_sqrt, float clip,
                                                            const float* __restrict__ norm, float gscale, int step_no, int zero_grad) {
  float coef = gscale;
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * AD_TH, t0 = (size_t)blockIdx.x * AD_TH + threadIdx.x;
  if (norm != nullptr) {
    if (!isfinite(norm[0])) {
      if (zero_grad)
        for (size_t i = t0; i < n; i += stride) g[i] = 0.f;
      return;
    }
    const float skipped = norm[1];
    if (skipped > 0.f) {
      const double eff = fmax((double)step_no - (double)skipped, 1.0);
      bc1 = (float)(1.0 - pow((double)b1, eff));
      bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, eff));
    }
    if (clip > 0.f) coef *= fminf(1.f, clip / (norm[0] * gscale + 1e-6f));
  }
  const float step = lr / bc1;
  auto upd = [&](float pi, float gr, float mo, float vo, float& po, float& mn, float& vn) {
    const float gi = gr * coef + wd * pi;
    mn = b1 * mo + (1.f - b1) * gi;
    vn = b2 * vo + (1.f - b2) * gi * gi;
    po = pi - step * mn / (sqrtf(vn) / bc2_sqrt + eps);
  };
  f32x4_t* p4 = reinterpret_cast<f32x4_t*>(p);
  f32x4_t* g4 = reinterpret_cast<f32x4_t*>(g);
  f32x4_t* m4 = reinterpret_cast<f32x4_t*>(m);
  f32x4_t* v4 = reinterpret_cast<f32x4_t*>(v);
  constexpr int U = 2;
  for (size_t i0 = t0; i0 < n4; i0 += U * stride) {
    f32x4_t pv[U], gv[U], mv[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride < n4 ? i0 + u * stride : n4 - 1;
      pv[u] = p4[i];
      gv[u] = NT_LD(g4 + i);
      mv[u] = NT_LD(m4 + i);
      vv[u] = NT_LD(v4 + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      if (i >= n4) break;
      f32x4_t po, mn, vn;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a, b, c;
        upd(pv[u][e], gv[u][e], mv[u][e], vv[u][e], a, b, c);
        po[e] = a; mn[e] = b; vn[e] = c;
      }
      NT_ST(m4 + i, mn);
      NT_ST(v4 + i, vn);
      p4[i] = po;
      if (zero_grad) NT_ST(g4 + i, ((f32x4_t){0.f, 0.f, 0.f, 0.f}));
    }
  }
  if (blockIdx.x == 0)       // the n % 4 tail
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += AD_TH) {
      float po, mn, vn;
      upd(p[i], g[i], m[i], v[i], po, mn, vn);
      m[i] = mn; v[i] = vn; p[i] = po;
      if (zero_grad) g[i] = 0.f;
    }
}

// compute units the optimizer's two launches may hold (0 = the wide forms); env MVF_OPT_WIDTH.  Measured beside the backbone forwards
// (tools/stretch_parts.py --parts none,opt; ms the optimizer adds to the step): wide 0.055, 16 CUs 0.053, 32: 0.043, 64: 0.030,
// 128: 0.045, 256: 0.07 -- profiles/r05/opt_width.txt
int g_opt_width = [] { const char* e = getenv("MVF_OPT_WIDTH"); return e ? atoi(e) : 64; }();
constexpr size_t OPT_LDS_HOLD = 96 * 1024;      // dynamic LDS request that keeps a second workgroup off the CU

}  // namespace

extern "C" int mvf_optim_set_width(int cus) {
  MVF_CHECK_ARG(cus >= 0 && cus <= 256);
  g_opt_width = cus;
  return MVF_OK;
}

// norm_out[0] = || g ||_2 (optionally sqrt(||g||^2 + extra_sq[0])); norm_out[1] (zeroed ONCE by the caller) counts the
// calls whose norm was not finite; scratch: >= 1024 floats
extern "C" int mvf_grad_norm(const float* g, size_t n, const float* extra_sq, float* scratch, float* norm_out,
                             hipStream_t st) {
  MVF_CHECK_ARG(g && scratch && norm_out && n > 0 && ((uintptr_t)g & 15) == 0);
  // 256 workgroups at most: every arrival is one atomic on the same ticket word (~11 ns each, serialised)
  int nblk = (int)std::min<size_t>(256, (n / 4 + SQ_TH - 1) / SQ_TH + 1);
  size_t lds = 0;
  if (g_opt_width > 0) {       // narrow form: `width` workgroups, each alone on its CU
    static uint64_t attr = 0;
    if (mvf_ensure_lds(reinterpret_cast<const void*>(sqnorm_kernel), OPT_LDS_HOLD, attr) != MVF_OK) return MVF_ERR_UNSUPPORTED;
    nblk = std::min(nblk, g_opt_width);
    lds = OPT_LDS_HOLD;
  }
  hipLaunchKernelGGL(sqnorm_kernel, dim3(nblk), dim3(SQ_TH), lds, st, g, n, scratch, extra_sq, norm_out, g_sqnorm_ring.take());
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// One Adam step on flat buffers. step >= 1 (1-based, like torch). clip <= 0 disables clipping.
// gscale multiplies every gradient first (e.g. 1/world_size when the all-reduce summed instead of averaged).
// norm (optional) = the two floats of mvf_grad_norm: a non-finite norm[0] turns the call into a no-op, and norm[1] steps
// skipped that way so far are taken off `step` in the bias correction.  zero_grad != 0: g is zeroed as it is consumed (also
// when the step is skipped) -- optimizer.zero_grad() of the next iteration (train.py:113) without its own pass over g.
extern "C" int mvf_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, float clip, const float* norm, float gscale,
                             int zero_grad, hipStream_t st) {
  MVF_CHECK_ARG(p && g && m && v && n > 0 && step >= 1);
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  if (g_opt_width > 0 && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && n >= 4) {
    static uint64_t attr = 0;
    if (mvf_ensure_lds(reinterpret_cast<const void*>(adam_narrow_kernel), OPT_LDS_HOLD, attr) != MVF_OK) return MVF_ERR_UNSUPPORTED;
    const int nb = (int)std::min<size_t>(g_opt_width, (n / 4 + AD_TH - 1) / AD_TH);
    hipLaunchKernelGGL(adam_narrow_kernel, dim3(nb), dim3(AD_TH), OPT_LDS_HOLD, st, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                       bc1, bc2s, clip, norm, gscale, step, zero_grad);
    MVF_LAUNCH_CHECK();
    return MVF_OK;
  }
  const int nblk = (int)std::min<size_t>(2048, (n + 255) / 256);
  hipLaunchKernelGGL(adam_kernel, dim3(nblk), dim3(256), 0, st, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1,
                     bc2s, clip, norm, gscale, step, zero_grad);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}
