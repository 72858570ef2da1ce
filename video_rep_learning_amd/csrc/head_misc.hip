// Row-/column-wise fp32 kernels of the trainable head (forward and backward):
//   LayerNorm              models/utils.py:147-159 (ResidualConnection.norm, eps 1e-5)
//   BatchNorm1d (+ReLU)    models/mvformer.py:78-79, resnet_c2d.py:118-119 (train: batch stats; eval: running)
//   entity one-hot concat  models/mvformer.py:144-149
//   entity reduction       models/mvformer.py:181-195 (SMART_FINAL one|avg|max)
//   F.normalize            models/transformer.py:228,230
// All are HBM/L2-bound on <= 1.6 MB tensors: one wave per row (row ops) or 64 columns x 4 row-lanes per
// workgroup (column reductions); no atomics, so results are run-to-run reproducible.
#include "common.h"
#include "mvf_hip_internal.h"

namespace {

// ---------------- LayerNorm ----------------
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                     const float* __restrict__ b, float* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows, int D,
                                                     float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * D;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += xr[c];
  const float mu = wave_sum(s) / D;
  float ss = 0.f;
  for (int c = lane; c < D; c += 64) { const float d = xr[c] - mu; ss += d * d; }
  const float rs = rsqrtf(wave_sum(ss) / D + eps);
  for (int c = lane; c < D; c += 64) y[(size_t)row * D + c] = (xr[c] - mu) * rs * g[c] + b[c];
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// One kernel: dx for 16 rows per workgroup (one wave = 4 rows) and the workgroup's partial dgamma/dbeta added with
// float atomics (rows/16 adders per address) into dg/db, which the caller zeroed or which already hold this step's
// gradient (flat gradient buffer).  Replaces a dx kernel + a 4-workgroup column-reduction kernel (50 us at 768 rows).
constexpr int LN_ROWS_PER_WAVE = 4;
template <int NCMAX>   // columns per lane: D <= 64 * NCMAX (8: the head's widths, 24: ViT widths up to 1536)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ g, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* dres, float* dx,
                                                     float* __restrict__ dg, float* __restrict__ db, int rows, int D) {
  // dres (may be NULL, may alias dx): a gradient that reaches x on a second path -- the residual connection around the
  // LayerNorm -- added here instead of by an elementwise kernel of the autograd engine
  extern __shared__ float red[];  // [4 waves][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nc = (D + 63) / 64;   // columns per lane (<= NCMAX)
  float pg[NCMAX], pb[NCMAX];
#pragma unroll
  for (int q = 0; q < NCMAX; ++q) { pg[q] = 0.f; pb[q] = 0.f; }
  for (int rr = 0; rr < LN_ROWS_PER_WAVE; ++rr) {
    const int row = (blockIdx.x * 4 + wave) * LN_ROWS_PER_WAVE + rr;
    if (row >= rows) break;
    const size_t o = (size_t)row * D;
    const float mu = mean[row], rs = rstd[row];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int q = 0; q < NCMAX; ++q) {
      const int c = lane + q * 64;
      if (q < nc && c < D) {
        const float d = dy[o + c], xh = (x[o + c] - mu) * rs;
        const float dgv = d * g[c];
        c1 += dgv;
        c2 += dgv * xh;
        pg[q] += d * xh;
        pb[q] += d;
      }
    }
    c1 = wave_sum(c1) / D;
    c2 = wave_sum(c2) / D;
#pragma unroll
    for (int q = 0; q < NCMAX; ++q) {
      const int c = lane + q * 64;
      if (q < nc && c < D) {
        const float v = rs * (dy[o + c] * g[c] - c1 - (x[o + c] - mu) * rs * c2);
        dx[o + c] = dres != nullptr ? dres[o + c] + v : v;
      }
    }
  }
  if (dg == nullptr) return;
#pragma unroll
  for (int q = 0; q < NCMAX; ++q) {
    const int c = lane + q * 64;
    if (q < nc && c < D) { red[(wave * 2 + 0) * D + c] = pg[q]; red[(wave * 2 + 1) * D + c] = pb[q]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    atomicAdd(dg + c, red[0 * D + c] + red[2 * D + c] + red[4 * D + c] + red[6 * D + c]);
    atomicAdd(db + c, red[1 * D + c] + red[3 * D + c] + red[5 * D + c] + red[7 * D + c]);
  }
}

// ---------------- BatchNorm1d ----------------
// Column reductions over the rows of a [rows, C] matrix (BN statistics, BN backward sums) in two stages so that a
// 768-row problem uses (C/64) x RS workgroups instead of C/64: stage 1 writes per-row-split partial sums to a caller
// workspace ws[RS][2][C], stage 2 (one thread per channel) adds them in a fixed order -- deterministic, no atomics.
// (The one-stage form with 8 workgroups took 47-53 us per call.)
__host__ __device__ inline int bn_splits(int rows) { return rows >= 1024 ? 32 : (rows >= 64 ? rows / 32 : 1); }

// stage 1 of the statistics: shifted sums  A = sum(x - p), B = sum((x - p)^2)  with the pivot p[c] = x[0, c]
// (plain E[x^2] - E[x]^2 would cancel catastrophically in fp32 when |mean| >> std)
// (stage 2 runs in the same launch: the last of a column block's RS workgroups to arrive adds the partial sums -- common.h
// last_arriver, one ticket per column block)
__device__ unsigned g_bn_ticket[TICKET_SLOTS][512];
TicketRing g_bn_ring;

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int rows, int C, int RS, float* __restrict__ ws,
                                                       float* __restrict__ mean, float* __restrict__ var, float* __restrict__ rmean,
                                                       float* __restrict__ rvar, float momentum, int ticket) {
  __shared__ float r1[4][64], r2[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, rs = blockIdx.y;
  const int chunk = (rows + RS - 1) / RS, rbeg = rs * chunk, rend = min(rows, rbeg + chunk);
  float a = 0.f, b = 0.f;
  if (c < C) {
    const float p = x[c];
    // (loads in batches of 8 rows, same addition order: a run-time loop of load -> add pays a memory round trip per row)
    for (int r0 = rbeg + rl; r0 < rend; r0 += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[(size_t)min(r0 + 4 * u, rend - 1) * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + 4 * u < rend) { const float d = v[u] - p; a += d; b += d * d; }
    }
  }
  r1[rl][cl] = a; r2[rl][cl] = b;
  __syncthreads();
  if (rl == 0 && c < C) {
    ws[((size_t)rs * 2 + 0) * C + c] = r1[0][cl] + r1[1][cl] + r1[2][cl] + r1[3][cl];
    ws[((size_t)rs * 2 + 1) * C + c] = r2[0][cl] + r2[1][cl] + r2[2][cl] + r2[3][cl];
  }
  // stage 2: mean, biased variance; optionally the running statistics (nn.BatchNorm1d: momentum, unbiased variance)
  if (!last_arriver(&g_bn_ticket[ticket][blockIdx.x], (unsigned)RS)) return;
  if (rl != 0 || c >= C) return;
  a = 0.f; b = 0.f;
  for (int q0 = 0; q0 < RS; q0 += 8) {
    float pa[8], pb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = min(q0 + u, RS - 1);
      pa[u] = ws[((size_t)q * 2 + 0) * C + c];
      pb[u] = ws[((size_t)q * 2 + 1) * C + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (q0 + u < RS) { a += pa[u]; b += pb[u]; }
  }
  const float inv = 1.f / rows, d = a * inv;
  const float mu = x[c] + d, v = fmaxf(b * inv - d * d, 0.f);
  mean[c] = mu;
  var[c] = v;
  if (rmean != nullptr) {
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * v * ((float)rows / fmaxf((float)rows - 1.f, 1.f));
  }
}

__global__ __launch_bounds__(256) void bn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                     const float* __restrict__ var, const float* __restrict__ g,
                                                     const float* __restrict__ b, float* __restrict__ y, size_t n, int C,
                                                     float eps, int relu) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float v = (x[i] - mean[c]) * rsqrtf(var[c] + eps) * g[c] + b[c];
    y[i] = relu ? fmaxf(v, 0.f) : v;
  }
}

// s1[c] = sum_r dy_eff ; s2[c] = sum_r dy_eff * xhat     (dy_eff = dy * [bn(x) > 0] when relu); stage 1: partial sums
// stage 2 (the last of a column block's RS workgroups): s1, s2 and (optionally) the parameter gradients dbeta (+)= s1,
// dgamma (+)= s2 of THIS rank's rows
__device__ unsigned g_bn_bwd_ticket[TICKET_SLOTS][512];
TicketRing g_bn_bwd_ring;

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ var,
                                                            const float* __restrict__ g, const float* __restrict__ b,
                                                            float* __restrict__ ws, int rows, int C, int RS, float eps,
                                                            int relu, float* __restrict__ s1, float* __restrict__ s2,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                            int ticket) {
  __shared__ float r1[4][64], r2[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, rs = blockIdx.y;
  const int chunk = (rows + RS - 1) / RS, rbeg = rs * chunk, rend = min(rows, rbeg + chunk);
  float a = 0.f, bb = 0.f;
  if (c < C) {
    const float mu = mean[c], rsd = rsqrtf(var[c] + eps), gg = g[c], be = b[c];
    for (int r0 = rbeg + rl; r0 < rend; r0 += 16) {      // batches of 4 rows x 2 loads, same addition order
      float xv[4], dv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t o = (size_t)min(r0 + 4 * u, rend - 1) * C + c;
        xv[u] = x[o];
        dv[u] = dy[o];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r0 + 4 * u >= rend) break;
        const float xh = (xv[u] - mu) * rsd;
        float d = dv[u];
        if (relu && !(xh * gg + be > 0.f)) d = 0.f;
        a += d;
        bb += d * xh;
      }
    }
  }
  r1[rl][cl] = a; r2[rl][cl] = bb;
  __syncthreads();
  if (rl == 0 && c < C) {
    ws[((size_t)rs * 2 + 0) * C + c] = r1[0][cl] + r1[1][cl] + r1[2][cl] + r1[3][cl];
    ws[((size_t)rs * 2 + 1) * C + c] = r2[0][cl] + r2[1][cl] + r2[2][cl] + r2[3][cl];
  }
  if (!last_arriver(&g_bn_bwd_ticket[ticket][blockIdx.x], (unsigned)RS)) return;
  if (rl != 0 || c >= C) return;
  float sa = 0.f, sb = 0.f;
  for (int q0 = 0; q0 < RS; q0 += 8) {
    float pa[8], pb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = min(q0 + u, RS - 1);
      pa[u] = ws[((size_t)q * 2 + 0) * C + c];
      pb[u] = ws[((size_t)q * 2 + 1) * C + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (q0 + u < RS) { sa += pa[u]; sb += pb[u]; }
  }
  s1[c] = sa;
  s2[c] = sb;
  if (dgamma != nullptr) {
    dgamma[c] = accumulate ? dgamma[c] + sb : sb;
    dbeta[c] = accumulate ? dbeta[c] + sa : sa;
  }
}

// train: dx = g*rstd*(dy_eff - s1/count - xhat*s2/count) ; eval (count <= 0): dx = g*rstd*dy_eff
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ var,
                                                           const float* __restrict__ g, const float* __restrict__ b,
                                                           const float* __restrict__ s1, const float* __restrict__ s2,
                                                           float* __restrict__ dx, size_t n, int C, float eps, int relu,
                                                           float inv_count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const float rs = rsqrtf(var[c] + eps);
    const float xh = (x[i] - mean[c]) * rs;
    float d = dy[i];
    if (relu && !(xh * g[c] + b[c] > 0.f)) d = 0.f;
    dx[i] = g[c] * rs * (d - s1[c] * inv_count - xh * s2[c] * inv_count);
  }
}

// ---------------- one-hot concat ----------------
// out[r, 0:cin] = x[r, :]; out[r, cin + j] = [tok(r) == j], tok(r) = (r / div) % ntok
__global__ void concat_onehot_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int cin, int ntok,
                                     int div) {
  const int cout = cin + ntok;
  const size_t n = (size_t)rows * cout;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cout), c = (int)(i % cout);
    out[i] = c < cin ? x[(size_t)r * cin + c] : ((r / div) % ntok == c - cin ? 1.f : 0.f);
  }
}

// ---------------- entity reduction: x [B, ntok, T, D] -> y [B, T, D] ----------------
// mode 0 'one' (token 0), 1 'avg', 2 'max' (argmax saved for backward)
__global__ void final_reduce_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int* __restrict__ arg, int B,
                                        int ntok, int T, int D, int mode) {
  const size_t n = (size_t)B * T * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int t = (int)((i / D) % T);
    const int bb = (int)(i / ((size_t)D * T));
    const float* p = x + (((size_t)bb * ntok) * T + t) * D + d;
    const size_t ts = (size_t)T * D;
    if (mode == 0) {
      y[i] = p[0];
    } else if (mode == 1) {
      float s = 0.f;
      for (int k = 0; k < ntok; ++k) s += p[k * ts];
      y[i] = s / ntok;
    } else {
      float m = p[0];
      int am = 0;
      for (int k = 1; k < ntok; ++k)
        if (p[k * ts] > m) { m = p[k * ts]; am = k; }
      y[i] = m;
      arg[i] = am;
    }
  }
}

__global__ void final_reduce_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ arg, float* __restrict__ dx,
                                        int B, int ntok, int T, int D, int mode) {
  const size_t n = (size_t)B * ntok * T * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int t = (int)((i / D) % T);
    const int k = (int)((i / ((size_t)D * T)) % ntok);
    const int bb = (int)(i / ((size_t)D * T * ntok));
    const size_t j = ((size_t)bb * T + t) * D + d;
    float v;
    if (mode == 0) v = k == 0 ? dy[j] : 0.f;
    else if (mode == 1) v = dy[j] / ntok;
    else v = arg[j] == k ? dy[j] : 0.f;
    dx[i] = v;
  }
}

// ---------------- F.normalize(dim=-1) ----------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ nrm, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t o = (size_t)row * D;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += x[o + c] * x[o + c];
  const float nn = fmaxf(sqrtf(wave_sum(s)), eps);
  for (int c = lane; c < D; c += 64) y[o + c] = x[o + c] / nn;
  if (lane == 0) nrm[row] = nn;
}

// y = x / n, n = max(||x||, eps):  dx = (dy - y (y.dy)) / n   (clamped branch: dx = dy / eps)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ nrm, float* __restrict__ dx, int rows,
                                                         int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t o = (size_t)row * D;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += y[o + c] * dy[o + c];
  s = wave_sum(s);
  const float nn = nrm[row];
  const bool clamped = !(nn > eps);
  for (int c = lane; c < D; c += 64) dx[o + c] = clamped ? dy[o + c] / nn : (dy[o + c] - y[o + c] * s) / nn;
}

// dx = dy * [y > 0]   (backward of the FFN's ReLU, models/utils.py:190)
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// exact-erf GELU of the trainable ViT blocks (timm Mlp act_layer = nn.GELU): y = x Phi(x), dx = dy (Phi(x) + x phi(x))
__global__ void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  }
}
__global__ void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
    dx[i] = dy[i] * (cdf + v * pdf);
  }
}

// LayerScale of a trainable DINOv2 block (timm LayerScale: x * gamma) fused with the residual add:
//   mode 0: out = resid + y * gamma[c]      mode 1: out = y * gamma[c] (dy of the backward)     mode 2: out = y * resid
//   (mode 2 = the elementwise product whose column sums are dgamma)
__global__ void colscale_kernel(const float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ resid,
                                float* __restrict__ out, size_t n, int D, int mode) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (mode == 2) out[i] = y[i] * resid[i];
    else {
      const float v = y[i] * gamma[i % D];
      out[i] = mode == 0 ? resid[i] + v : v;
    }
  }
}

// y = resid + dropout_p(x)  (nn.Dropout + residual add of ResidualConnection, models/utils.py:153-159; plain
// nn.Dropout when resid == null).  Counter-based mask: keep(i) = hash(seed, offset + i) >= p * 2^32, so the
// backward regenerates the identical mask from (seed, offset) and nothing is stored.
__global__ void dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ resid, float* __restrict__ y,
                                   size_t n, uint32_t thresh, float scale, uint64_t seed, uint64_t offset) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = x[i];
    if (thresh != 0u) v = drop_keep(seed, offset, i, thresh) ? v * scale : 0.f;
    y[i] = resid ? resid[i] + v : v;
  }
}

inline int ew_grid(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 256 * 8); }

}  // namespace

extern "C" int mvf_ln_fwd(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd, int rows,
                          int D, float eps, hipStream_t st) {
  MVF_CHECK_ARG(x && g && b && y && mean && rstd && rows > 0 && D > 0);
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, st, x, g, b, y, mean, rstd, rows, D, eps);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// dg/db: accumulate_params != 0 adds into them (flat gradient buffer), else they are overwritten (zeroed here first)
static int ln_bwd_launch(const float* dy, const float* x, const float* g, const float* mean, const float* rstd, const float* dres,
                         float* dx, float* dg, float* db, int rows, int D, int accumulate_params, hipStream_t st) {
  MVF_CHECK_ARG(dy && x && g && mean && rstd && dx && rows > 0 && D > 0 && D <= 1536 && ((dg == nullptr) == (db == nullptr)));
  if (dg && !accumulate_params) {
    if (hipMemsetAsync(dg, 0, (size_t)D * 4, st) != hipSuccess || hipMemsetAsync(db, 0, (size_t)D * 4, st) != hipSuccess)
      return MVF_ERR_ARG;
  }
  const dim3 grid(ceil_div(rows, 4 * LN_ROWS_PER_WAVE));
  if (D <= 512)
    hipLaunchKernelGGL(ln_bwd_kernel<8>, grid, dim3(256), (size_t)8 * D * 4, st, dy, x, g, mean, rstd, dres, dx, dg, db, rows, D);
  else   // ViT widths (trainable backbone blocks): 768 .. 1536
    hipLaunchKernelGGL(ln_bwd_kernel<24>, grid, dim3(256), (size_t)8 * D * 4, st, dy, x, g, mean, rstd, dres, dx, dg, db, rows, D);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_ln_bwd(const float* dy, const float* x, const float* g, const float* mean, const float* rstd, float* dx,
                          float* dg, float* db, int rows, int D, int accumulate_dx, int accumulate_params,
                          hipStream_t st) {
  return ln_bwd_launch(dy, x, g, mean, rstd, accumulate_dx ? dx : nullptr, dx, dg, db, rows, D, accumulate_params, st);
}

// dx = dres + d LN / dx (dy): the pre-LN residual connection x + sub(LN(x)) (models/utils.py:147-159) hands x two gradients --
// the residual path's (dres) and the LayerNorm's -- summed here in the LayerNorm backward's own pass over the rows
extern "C" int mvf_ln_bwd_res(const float* dy, const float* x, const float* g, const float* mean, const float* rstd,
                              const float* dres, float* dx, float* dg, float* db, int rows, int D, int accumulate_params,
                              hipStream_t st) {
  MVF_CHECK_ARG(dres != nullptr);
  return ln_bwd_launch(dy, x, g, mean, rstd, dres, dx, dg, db, rows, D, accumulate_params, st);
}

extern "C" size_t mvf_bn_workspace_floats(int rows, int C) { return (size_t)bn_splits(rows) * 2 * C; }

// mean / biased variance of x over its rows; running_mean / running_var (may be NULL) are updated like nn.BatchNorm1d
// does in training (skip them under SyncBN: the caller updates them from the merged statistics)
extern "C" int mvf_bn_stats(const float* x, int rows, int C, float* mean, float* var, float* running_mean,
                            float* running_var, float momentum, float* ws, size_t ws_floats, hipStream_t st) {
  MVF_CHECK_ARG(x && mean && var && ws && rows > 0 && C > 0 && ((running_mean == nullptr) == (running_var == nullptr)));
  const int RS = bn_splits(rows);
  MVF_CHECK_ARG(ws_floats >= (size_t)RS * 2 * C);
  MVF_CHECK_ARG(ceil_div(C, 64) <= 512);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(ceil_div(C, 64), RS), dim3(256), 0, st, x, rows, C, RS, ws, mean, var, running_mean,
                     running_var, momentum, g_bn_ring.take());
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// SyncBatchNorm (CARL_MVF/train.py:283-286 -> nn.SyncBatchNorm): the ranks' statistics, gathered as rows [mean C | biased var C |
// row count] of `gathered` [W][2C + 1], merged into the statistics of the rank-concatenated batch (Chan; every rank holds
// `count_per_rank` rows in the data-parallel step, so the weights are equal) and the running buffers updated like nn.BatchNorm1d
// does in training (unbiased variance over all W * count rows).  One thread per channel, the ranks summed in rank order: every rank
// computes the same bits from the same gathered block.  Replaces ~10 ATen kernels per BatchNorm and step (cat / stack / sum / pow ...).
__global__ void syncbn_merge_kernel(const float* __restrict__ g, int W, int C, float total, float* __restrict__ mean,
                                    float* __restrict__ var, float* __restrict__ rmean, float* __restrict__ rvar, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const size_t ld = (size_t)2 * C + 1;
  float sm = 0.f;
  for (int r = 0; r < W; ++r) sm += g[r * ld + c];
  const float gm = sm / (float)W;
  float sv = 0.f;
  for (int r = 0; r < W; ++r) {
    const float d = g[r * ld + c] - gm;
    sv += g[r * ld + C + c] + d * d;
  }
  const float gv = sv / (float)W;
  mean[c] = gm;
  var[c] = gv;
  if (rmean != nullptr) {
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * gm;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * gv * (total / fmaxf(total - 1.f, 1.f));
  }
}

extern "C" int mvf_syncbn_merge(const float* gathered, int W, int C, float count_per_rank, float* mean, float* var,
                                float* running_mean, float* running_var, float momentum, hipStream_t st) {
  MVF_CHECK_ARG(gathered && mean && var && W > 0 && C > 0 && count_per_rank > 0.f &&
                ((running_mean == nullptr) == (running_var == nullptr)));
  hipLaunchKernelGGL(syncbn_merge_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, st, gathered, W, C, count_per_rank * (float)W, mean, var,
                     running_mean, running_var, momentum);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_bn_fwd(const float* x, const float* mean, const float* var, const float* g, const float* b, float* y,
                          int rows, int C, float eps, int relu, hipStream_t st) {
  MVF_CHECK_ARG(x && mean && var && g && b && y && rows > 0 && C > 0);
  const size_t n = (size_t)rows * C;
  hipLaunchKernelGGL(bn_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, mean, var, g, b, y, n, C, eps, relu);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// s1 = sum_r dy_eff, s2 = sum_r dy_eff * xhat; dgamma / dbeta (may be NULL): this rank's parameter gradients
// (= s2 / s1 before any cross-rank reduction), added in place when accumulate_params != 0 (flat gradient buffer)
extern "C" int mvf_bn_bwd_reduce(const float* dy, const float* x, const float* mean, const float* var, const float* g,
                                 const float* b, float* s1, float* s2, float* dgamma, float* dbeta, int accumulate_params,
                                 int rows, int C, float eps, int relu, float* ws, size_t ws_floats, hipStream_t st) {
  MVF_CHECK_ARG(dy && x && mean && var && g && b && s1 && s2 && ws && rows > 0 && C > 0);
  MVF_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr));
  const int RS = bn_splits(rows);
  MVF_CHECK_ARG(ws_floats >= (size_t)RS * 2 * C);
  MVF_CHECK_ARG(ceil_div(C, 64) <= 512);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(ceil_div(C, 64), RS), dim3(256), 0, st, dy, x, mean, var, g, b, ws, rows, C,
                     RS, eps, relu, s1, s2, dgamma, dbeta, accumulate_params, g_bn_bwd_ring.take());
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// count = number of rows the statistics were taken over (global count under SyncBN); count <= 0: eval mode
extern "C" int mvf_bn_bwd_apply(const float* dy, const float* x, const float* mean, const float* var, const float* g,
                                const float* b, const float* s1, const float* s2, float* dx, int rows, int C, float eps,
                                int relu, float count, hipStream_t st) {
  MVF_CHECK_ARG(dy && x && mean && var && g && b && s1 && s2 && dx && rows > 0 && C > 0);
  const size_t n = (size_t)rows * C;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n)), dim3(256), 0, st, dy, x, mean, var, g, b, s1, s2, dx, n, C,
                     eps, relu, count > 0.f ? 1.0f / count : 0.f);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_concat_onehot(const float* x, float* out, int rows, int cin, int ntok, int div, hipStream_t st) {
  MVF_CHECK_ARG(x && out && rows > 0 && cin > 0 && ntok > 0 && div > 0);
  const size_t n = (size_t)rows * (cin + ntok);
  hipLaunchKernelGGL(concat_onehot_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, out, rows, cin, ntok, div);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_final_reduce_fwd(const float* x, float* y, int* arg, int B, int ntok, int T, int D, int mode,
                                    hipStream_t st) {
  MVF_CHECK_ARG(x && y && (mode != 2 || arg) && mode >= 0 && mode <= 2 && B > 0 && ntok > 0 && T > 0 && D > 0);
  const size_t n = (size_t)B * T * D;
  hipLaunchKernelGGL(final_reduce_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, y, arg, B, ntok, T, D, mode);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_final_reduce_bwd(const float* dy, const int* arg, float* dx, int B, int ntok, int T, int D, int mode,
                                    hipStream_t st) {
  MVF_CHECK_ARG(dy && dx && (mode != 2 || arg) && mode >= 0 && mode <= 2 && B > 0 && ntok > 0 && T > 0 && D > 0);
  const size_t n = (size_t)B * ntok * T * D;
  hipLaunchKernelGGL(final_reduce_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, st, dy, arg, dx, B, ntok, T, D, mode);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_l2norm_fwd(const float* x, float* y, float* nrm, int rows, int D, float eps, hipStream_t st) {
  MVF_CHECK_ARG(x && y && nrm && rows > 0 && D > 0);
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, st, x, y, nrm, rows, D, eps);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_l2norm_bwd(const float* dy, const float* y, const float* nrm, float* dx, int rows, int D, float eps,
                              hipStream_t st) {
  MVF_CHECK_ARG(dy && y && nrm && dx && rows > 0 && D > 0);
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, st, dy, y, nrm, dx, rows, D, eps);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_relu_bwd(const float* dy, const float* y, float* dx, size_t n, hipStream_t st) {
  MVF_CHECK_ARG(dy && y && dx && n > 0);
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, st, dy, y, dx, n);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// ---- split-K weight gradients of the trainable backbone blocks (ops._LinearTC) ----
// transpose + cast: in fp32 [M, C] -> out bf16 [S, C, Mc] with out[s][c][j] = in[s*Mc + j][c] (0 for rows >= M): both
// operands of dW = dY^T X become K-contiguous along the token axis, one chunk of Mc tokens per batch of mvf_gemm_tc_batched
__global__ __launch_bounds__(256) void transpose_chunks_kernel(const float* __restrict__ in, unsigned short* __restrict__ out,
                                                               int M, int C, int Mc) {
  __shared__ float tile[64][65];
  const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int m = m0 + r, c = c0 + tx;
    tile[r][tx] = (m < M && c < C) ? in[(size_t)m * C + c] : 0.0f;
  }
  __syncthreads();
  const int s = m0 / Mc, j0 = m0 - s * Mc;          // Mc % 64 == 0: a tile never straddles two chunks
  // every thread writes 8 consecutive tokens of one channel: 16-byte stores (2-byte stores ran 10x slower)
  const int jg = (threadIdx.x & 7) * 8;
  for (int cc = threadIdx.x >> 3; cc < 64; cc += 32) {
    const int c = c0 + cc;
    if (c >= C) continue;
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = pack_bf16x2(tile[jg + 2 * q][cc], tile[jg + 2 * q + 1][cc]);
    *reinterpret_cast<uint4*>(out + ((size_t)s * C + c) * Mc + j0 + jg) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// out[i] (+)= sum_s part[s][i]  (fixed order: deterministic)
__global__ void sum_batches_kernel(const float* __restrict__ part, float* __restrict__ out, int S, size_t n, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float acc = accumulate ? out[i] : 0.0f;
    for (int s = 0; s < S; ++s) acc += part[(size_t)s * n + i];
    out[i] = acc;
  }
}

extern "C" int mvf_transpose_chunks(const float* in, void* out_bf16, int M, int C, int Mc, hipStream_t st) {
  MVF_CHECK_ARG(in && out_bf16 && M > 0 && C > 0 && Mc > 0 && Mc % 64 == 0);
  const int S = ceil_div(M, Mc);
  hipLaunchKernelGGL(transpose_chunks_kernel, dim3(S * (Mc / 64), ceil_div(C, 64)), dim3(256), 0, st, in,
                     reinterpret_cast<unsigned short*>(out_bf16), M, C, Mc);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_sum_batches(const float* part, float* out, int S, size_t n, int accumulate, hipStream_t st) {
  MVF_CHECK_ARG(part && out && S > 0 && n > 0);
  hipLaunchKernelGGL(sum_batches_kernel, dim3(ew_grid(n)), dim3(256), 0, st, part, out, S, n, accumulate);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_colscale(const float* y, const float* gamma, const float* resid, float* out, int rows, int D, int mode,
                            hipStream_t st) {
  MVF_CHECK_ARG(y && out && rows > 0 && D > 0 && mode >= 0 && mode <= 2 && (mode == 2 || gamma) && (mode == 1 || resid));
  const size_t n = (size_t)rows * D;
  hipLaunchKernelGGL(colscale_kernel, dim3(ew_grid(n)), dim3(256), 0, st, y, gamma, resid, out, n, D, mode);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_gelu_fwd(const float* x, float* y, size_t n, hipStream_t st) {
  MVF_CHECK_ARG(x && y && n > 0);
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, y, n);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_gelu_bwd(const float* dy, const float* x, float* dx, size_t n, hipStream_t st) {
  MVF_CHECK_ARG(dy && x && dx && n > 0);
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, st, dy, x, dx, n);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// p in [0,1): y = resid + mask * x / (1-p)  (resid may be NULL).  The same call with x = dy, resid = NULL is the backward.
extern "C" int mvf_dropout_add(const float* x, const float* resid, float* y, size_t n, float p, uint64_t seed,
                               uint64_t offset, hipStream_t st) {
  MVF_CHECK_ARG(x && y && n > 0 && p >= 0.f && p < 1.f);
  const uint32_t thresh = p > 0.f ? (uint32_t)std::min<double>(4294967295.0, (double)p * 4294967296.0) : 0u;
  hipLaunchKernelGGL(dropout_add_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, resid, y, n, thresh, 1.0f / (1.0f - p),
                     seed, offset);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}
