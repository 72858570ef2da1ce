// Row-chain kernels for the Linear stages of the trainable head that sit between BatchNorms: the per-entity FC stack and video_emb
// (CARL_MVF/models/mvformer.py:70-86,150-160), the entity reduction + embedding layer (mvformer.py:181-199) and the projection head
// with its normalisation (models/resnet_c2d.py:112-126, models/transformer.py:226-228).  One launch = one Linear with everything
// row-wise around it:
//
//   forward   Y = epi( pro(X) W^T + b ),   pro = [entity reduce] -> [BatchNorm (+ReLU) of the PREVIOUS Linear's output, batch
//             statistics given] -> [append the entity one-hot] -> [dropout];   epi = [+ sin/cos table] -> [dropout] | [L2 normalise];
//             and, when a BatchNorm follows, the batch statistics of Y in the same launch (per-workgroup column mean / M2, merged
//             by the workgroup that arrives last -- common.h last_arriver; Chan's parallel variance, no atomics, fixed order)
//   backward  the same chain reversed: [BatchNorm backward of the NEXT layer applied while loading its dZ] -> epi' -> dXp = g W ->
//             pro' (dropout mask, one-hot columns dropped, ReLU mask) -> dZ and the column sums s1 = sum dZ, s2 = sum dZ xhat the
//             BatchNorm backward of THIS layer's input needs (again finished by the last arriver, with dgamma / dbeta);
//             g is also written as the fragment-major transpose for the weight-gradient launch (mvf_head_dw)
//
// which replaces dropout_add, hgemm, bn_stats, bn_fwd, concat_onehot, final_reduce, l2norm (forward: 21 launches -> 6) and
// l2norm_bwd, hlinear_bwd, bn_bwd_reduce, bn_bwd_apply, dropout_add, final_reduce_bwd (backward: 20 -> 6 + one weight-gradient
// launch).  GEMM operands bf16 (fp32 accumulate), everything else fp32 -- see head_chain.hip; building blocks in head_chain.h.
// Shapes: N <= 512 outputs and <= 512 inputs (the FC widths of CAPACITY_SCALAR 2); wider stacks keep the fp32 kernels.
#include "head_chain.h"

namespace {
using namespace chain;

__device__ unsigned g_rl_ticket_f[TICKET_SLOTS], g_rl_ticket_b[TICKET_SLOTS];
TicketRing g_rl_ring_f, g_rl_ring_b;

// diagnostic (tools/rowlin_probe.py): stage time stamps of workgroup 0, one 16-slot record per launch
long long* g_rl_stamps = nullptr;
int g_rl_stamp_slots = 0, g_rl_stamp_next = 0;
long long* next_stamps() { return g_rl_stamps != nullptr && g_rl_stamp_next < g_rl_stamp_slots ? g_rl_stamps + 16 * g_rl_stamp_next++ : nullptr; }
#define RSTAMP(i) do { if (k.stamps != nullptr && threadIdx.x == 0 && blockIdx.x == 0) k.stamps[i] = wall_clock64(); } while (0)

struct RowLinFwdK {
  int M, Cin, Kin, N, Mp;
  const float* X; long ldx;
  int g_ntok, g_T, g_mode; int* g_arg;
  const float *bn_mean, *bn_var, *bn_g, *bn_b; float bn_eps; int bn_relu;
  int oh_ntok, oh_div;
  Drop drop_in, drop_out;
  const bf16_t* w; const float* bias;
  const float* table; int tab_mod;
  int l2norm; float l2_eps;
  float *Y, *nrm;
  bf16_t* xT;
  float *st_part, *st_mean, *st_var, *st_rmean, *st_rvar; float st_momentum;
  long long* stamps;
  int ticket;
};

// LDS carve-up (bytes): bf16 input panel, fp32 output panel, per-column (scale, shift) of the input BatchNorm
struct RlLds { int lda, ldo; size_t pa, po, sc, total; };
__host__ __device__ inline RlLds rl_lds_fwd(int Kin, int N, int Cin) {
  RlLds l;
  l.lda = fm_steps(Kin) * 32 + 8; l.ldo = N + 4;
  size_t o = 0;
  l.pa = o; o += (size_t)TM * l.lda * 2;
  l.po = o; o += (size_t)TM * l.ldo * 4;
  l.sc = o; o += (size_t)2 * ((Cin + 3) & ~3) * 4;
  l.total = o;
  return l;
}

// F16: IEEE fp16 GEMM operands (MI355X.HEAD_DTYPE fp16: the forward only, head_chain.hip header); xT is written as bf16 either way
template <int NT, bool F16 = false>
__global__ __launch_bounds__(NTH) void rowlin_fwd_kernel(RowLinFwdK k) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const RlLds L = rl_lds_fwd(k.Kin, k.N, k.Cin);
  bf16_t* Pa = reinterpret_cast<bf16_t*>(sm + L.pa);
  float* Po = reinterpret_cast<float*>(sm + L.po);
  float* sc = reinterpret_cast<float*>(sm + L.sc);
  float* sh = sc + ((k.Cin + 3) & ~3);
  const int m0 = blockIdx.x * TM, M = k.M, Cin = k.Cin, Kin = k.Kin, N = k.N;
  const int Kp = fm_steps(Kin) * 32;
  const bool bn = k.bn_mean != nullptr;
  RSTAMP(0);
  if (bn) {
    for (int c = threadIdx.x; c < Cin; c += NTH) {
      const float s = k.bn_g[c] * rsqrtf(k.bn_var[c] + k.bn_eps);
      sc[c] = s;
      sh[c] = k.bn_b[c] - k.bn_mean[c] * s;
    }
    LDS_BARRIER();
  }
  RSTAMP(1);
  // ---- loader: X rows -> pro() -> bf16 panel ----
  {
    const int c4 = Cin >> 2;       // Cin % 4 == 0
    const int total = TM * c4;
    constexpr int U = 8;
    for (int i0 = threadIdx.x; i0 < total; i0 += NTH * U) {
      f32x4_t v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * NTH, r = i / c4, q = i - r * c4, m = m0 + r;
        v[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (i < total && m < M) {
          if (k.g_ntok > 0) {        // entity reduction: rows (b, j, t) -> (b, t); one | avg | max
            const int b = m / k.g_T, t = m - b * k.g_T;
            const float* p = k.X + ((size_t)b * k.g_ntok * k.g_T + t) * k.ldx + 4 * q;
            f32x4_t a = *reinterpret_cast<const f32x4_t*>(p);
            if (k.g_mode != 0) {
              int am[4] = {0, 0, 0, 0};
              for (int j = 1; j < k.g_ntok; ++j) {
                const f32x4_t w = *reinterpret_cast<const f32x4_t*>(p + (size_t)j * k.g_T * k.ldx);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  if (k.g_mode == 1) a[e] += w[e];
                  else if (w[e] > a[e]) { a[e] = w[e]; am[e] = j; }
                }
              }
              if (k.g_mode == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] /= k.g_ntok;
              } else if (k.g_arg != nullptr) {
                *reinterpret_cast<int4*>(k.g_arg + (size_t)m * Cin + 4 * q) = make_int4(am[0], am[1], am[2], am[3]);
              }
            }
            v[u] = a;
          } else {
            v[u] = *reinterpret_cast<const f32x4_t*>(k.X + (size_t)m * k.ldx + 4 * q);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * NTH, r = i / c4, q = i - r * c4, m = m0 + r;
        if (i >= total) continue;
        f32x4_t x = v[u];
        if (m < M) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = 4 * q + e;
            if (bn) { x[e] = x[e] * sc[c] + sh[c]; if (k.bn_relu) x[e] = fmaxf(x[e], 0.f); }
            x[e] = drop_apply(k.drop_in, x[e], (uint64_t)m * Kin + c);
          }
        }
        *reinterpret_cast<u32x2_t*>(Pa + r * L.lda + 4 * q) = (u32x2_t){pack16x2<F16>(x[0], x[1]), pack16x2<F16>(x[2], x[3])};
      }
    }
    // one-hot of the row's entity + zero padding up to the image's K
    for (int i = threadIdx.x; i < TM * (Kp - Cin); i += NTH) {
      const int r = i / (Kp - Cin), c = Cin + i - r * (Kp - Cin), m = m0 + r;
      float x = 0.f;
      if (m < M && c < Kin) {
        x = (m / k.oh_div) % k.oh_ntok == c - Cin ? 1.f : 0.f;
        x = drop_apply(k.drop_in, x, (uint64_t)m * Kin + c);
      }
      Pa[r * L.lda + c] = f32_to_16<F16>(x);
    }
  }
  LDS_BARRIER();
  RSTAMP(2);
  store_T<F16>(Pa, L.lda, Kin, k.xT, k.Mp, m0, M);
  RSTAMP(3);
  // ---- Y = pro(X) W^T + b [+ table] [dropout] ----
  chain_gemm<NT, 4, F16>(Pa, L.lda, Kin, k.w, N, [&](int m, int n) {
    Aux2 a;
    a.b = k.bias != nullptr ? *reinterpret_cast<const float4*>(k.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    a.r = k.table != nullptr ? *reinterpret_cast<const float4*>(k.table + (size_t)((m0 + m) % k.tab_mod) * N + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    return a;
  }, [&](int m, int n, const f32x4_t& v, const Aux2& ax) {
    const int gm = m0 + m;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gm < M) {
      const uint64_t idx = (uint64_t)gm * N + n;
      r.x = drop_apply(k.drop_out, v[0] + ax.b.x + ax.r.x, idx);
      r.y = drop_apply(k.drop_out, v[1] + ax.b.y + ax.r.y, idx + 1);
      r.z = drop_apply(k.drop_out, v[2] + ax.b.z + ax.r.z, idx + 2);
      r.w = drop_apply(k.drop_out, v[3] + ax.b.w + ax.r.w, idx + 3);
    }
    *reinterpret_cast<float4*>(Po + m * L.ldo + n) = r;
  });
  LDS_BARRIER();
  RSTAMP(4);
  if (k.l2norm) {       // F.normalize(dim = -1): 16 lanes per row
    const int r = threadIdx.x >> 4, sub = threadIdx.x & 15;
    float s = 0.f;
    for (int c = sub; c < N; c += 16) { const float y = Po[r * L.ldo + c]; s += y * y; }
    const float nn = fmaxf(sqrtf(sum16(s)), k.l2_eps);
    for (int c = sub; c < N; c += 16) Po[r * L.ldo + c] /= nn;
    if (sub == 0 && m0 + r < M && k.nrm != nullptr) k.nrm[m0 + r] = nn;
    LDS_BARRIER();
  }
  {
    const int c4 = N >> 2;
    for (int i = threadIdx.x; i < TM * c4; i += NTH) {
      const int r = i / c4, q = i - r * c4;
      if (m0 + r < M) *reinterpret_cast<float4*>(k.Y + (size_t)(m0 + r) * N + 4 * q) = *reinterpret_cast<const float4*>(Po + r * L.ldo + 4 * q);
    }
  }
  RSTAMP(5);
  if (k.st_part == nullptr) return;
  // ---- batch statistics of Y for the BatchNorm that follows ----
  const int cnt = min(TM, M - m0);
  for (int c = threadIdx.x; c < N; c += NTH) {
    float a = 0.f;
    for (int r = 0; r < cnt; ++r) a += Po[r * L.ldo + c];
    const float mu = a / cnt;
    float q2 = 0.f;
    for (int r = 0; r < cnt; ++r) { const float d = Po[r * L.ldo + c] - mu; q2 += d * d; }
    k.st_part[((size_t)blockIdx.x * 2 + 0) * N + c] = mu;
    k.st_part[((size_t)blockIdx.x * 2 + 1) * N + c] = q2;
  }
  RSTAMP(6);
  if (!last_arriver(&g_rl_ticket_f[k.ticket], gridDim.x)) { RSTAMP(7); return; }
  for (int c = threadIdx.x; c < N; c += NTH) {
    float n = 0.f, mu = 0.f, m2 = 0.f;
    const int nwg = (int)gridDim.x;
    for (int w0 = 0; w0 < nwg; w0 += 8) {       // Chan's merge, workgroup order; the partials of 8 workgroups requested together
      float mw[8], qw[8];                        // (a load -> use loop pays an L2 round trip per workgroup: 24 of them = 17 us)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int w = min(w0 + u, nwg - 1);
        mw[u] = k.st_part[((size_t)w * 2 + 0) * N + c];
        qw[u] = k.st_part[((size_t)w * 2 + 1) * N + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (w0 + u >= nwg) break;
        const float nb = (float)min(TM, M - (w0 + u) * TM);
        const float d = mw[u] - mu, tot = n + nb;
        mu += d * nb / tot;
        m2 += qw[u] + d * d * n * nb / tot;
        n = tot;
      }
    }
    const float var = m2 / n;
    k.st_mean[c] = mu;
    k.st_var[c] = var;
    if (k.st_rmean != nullptr) {
      k.st_rmean[c] = (1.f - k.st_momentum) * k.st_rmean[c] + k.st_momentum * mu;
      k.st_rvar[c] = (1.f - k.st_momentum) * k.st_rvar[c] + k.st_momentum * var * (n / fmaxf(n - 1.f, 1.f));
    }
  }
}

struct RowLinBwdK {
  int M, Cin, Kin, N, Mp;
  const float* dY;
  const float *nb_Y, *nb_mean, *nb_var, *nb_g, *nb_s1, *nb_s2; float nb_eps, nb_inv_count; int nb_relu_unused;
  Drop drop_out, drop_in;
  int l2norm; const float *l2_y, *l2_nrm; float l2_eps;
  const bf16_t* wT; bf16_t* gT;
  int oh_ntok;
  const float* X; long ldx; const float *bn_mean, *bn_var, *bn_g, *bn_b; float bn_eps; int bn_relu;
  float *st_part, *s1, *s2, *dgamma, *dbeta;
  int g_ntok, g_T, g_mode; const int* g_arg;
  float* dX; long lddx;
  long long* stamps;
  int ticket;
};

struct RlLdsB { int ldg, ldd; size_t pg, pd, cst, total; };
__host__ __device__ inline RlLdsB rl_lds_bwd(int Kin, int N, int Cin) {
  RlLdsB l;
  const int kr = (Kin + 63) & ~63;
  l.ldg = fm_steps(N) * 32 + 8;
  l.ldd = (kr > N ? kr : N) + 4;
  size_t o = 0;
  l.pg = o; o += (size_t)TM * l.ldg * 2;
  l.pd = o; o += (size_t)TM * l.ldd * 4;
  const int cmax = (Cin > N ? Cin : N);
  l.cst = o; o += (size_t)5 * ((cmax + 3) & ~3) * 4;
  l.total = o;
  return l;
}

template <int NT>
__global__ __launch_bounds__(NTH) void rowlin_bwd_kernel(RowLinBwdK k) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const RlLdsB L = rl_lds_bwd(k.Kin, k.N, k.Cin);
  bf16_t* Pg = reinterpret_cast<bf16_t*>(sm + L.pg);
  float* Pd = reinterpret_cast<float*>(sm + L.pd);
  const int cpad = (((k.Cin > k.N ? k.Cin : k.N) + 3) & ~3);
  float* c0 = reinterpret_cast<float*>(sm + L.cst);     // five per-column constant arrays
  float *c1 = c0 + cpad, *c2 = c1 + cpad, *c3 = c2 + cpad, *c4a = c3 + cpad;
  const int m0 = blockIdx.x * TM, M = k.M, Cin = k.Cin, Kin = k.Kin, N = k.N;
  const int Np = fm_steps(N) * 32;
  const bool nb = k.nb_Y != nullptr;
  RSTAMP(0);
  if (nb) {     // BatchNorm backward of the layer that consumes Y: dY = a0 (dZ - a1 - xhat a2), xhat = (Y - mean) rstd
    for (int n = threadIdx.x; n < N; n += NTH) {
      const float rs = rsqrtf(k.nb_var[n] + k.nb_eps);
      c0[n] = k.nb_g[n] * rs; c1[n] = k.nb_s1[n] * k.nb_inv_count; c2[n] = k.nb_s2[n] * k.nb_inv_count; c3[n] = k.nb_mean[n]; c4a[n] = rs;
    }
    LDS_BARRIER();
  }
  RSTAMP(1);
  // ---- dY rows -> fp32 panel (BatchNorm backward applied, dropout mask) ----
  {
    const int q4 = N >> 2, total = TM * q4;
    constexpr int U = 8;
    for (int i0 = threadIdx.x; i0 < total; i0 += NTH * U) {
      f32x4_t v[U], y[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * NTH, r = i / q4, q = i - r * q4, m = m0 + r;
        v[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; y[u] = v[u];
        if (i < total && m < M) {
          v[u] = *reinterpret_cast<const f32x4_t*>(k.dY + (size_t)m * N + 4 * q);
          if (nb) y[u] = *reinterpret_cast<const f32x4_t*>(k.nb_Y + (size_t)m * N + 4 * q);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * NTH, r = i / q4, q = i - r * q4, m = m0 + r;
        if (i >= total) continue;
        f32x4_t d = v[u];
        if (m < M) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = 4 * q + e;
            if (nb) d[e] = c0[n] * (d[e] - c1[n] - (y[u][e] - c3[n]) * c4a[n] * c2[n]);
            d[e] = drop_apply(k.drop_out, d[e], (uint64_t)m * N + n);
          }
        }
        *reinterpret_cast<f32x4_t*>(Pd + r * L.ldd + 4 * q) = d;
      }
    }
  }
  LDS_BARRIER();
  if (k.l2norm) {      // y = x / n, n = max(||x||, eps):  dx = (dy - y (y . dy)) / n   (clamped branch: dx = dy / n)
    const int r = threadIdx.x >> 4, sub = threadIdx.x & 15, m = min(m0 + r, M - 1);
    float s = 0.f;
    for (int c = sub; c < N; c += 16) s += k.l2_y[(size_t)m * N + c] * Pd[r * L.ldd + c];
    s = sum16(s);
    const float nn = k.l2_nrm[m];
    const bool clamped = !(nn > k.l2_eps);
    for (int c = sub; c < N; c += 16) {
      const float dy = Pd[r * L.ldd + c];
      Pd[r * L.ldd + c] = clamped ? dy / nn : (dy - k.l2_y[(size_t)m * N + c] * s) / nn;
    }
    LDS_BARRIER();
  }
  RSTAMP(2);
  // g -> bf16 panel (zero padding up to the image's reduction length), its transpose for the weight gradient
  for (int i = threadIdx.x; i < TM * (Np >> 2); i += NTH) {
    const int r = i / (Np >> 2), q = i - r * (Np >> 2);
    f32x4_t d = {0.f, 0.f, 0.f, 0.f};
    if (4 * q < N && m0 + r < M) d = *reinterpret_cast<const f32x4_t*>(Pd + r * L.ldd + 4 * q);
    *reinterpret_cast<u32x2_t*>(Pg + r * L.ldg + 4 * q) = (u32x2_t){pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3])};
  }
  LDS_BARRIER();
  RSTAMP(3);
  store_T(Pg, L.ldg, N, k.gT, k.Mp, m0, M);
  RSTAMP(4);
  // ---- dXp = g W  (W^T image: rows = input features, padded to 64) ----
  const int Kr = (Kin + 63) & ~63;
  chain_gemm<NT, 4>(Pg, L.ldg, N, k.wT, Kr, [](int, int) { return NoAux{}; }, [&](int m, int n, const f32x4_t& v, const NoAux&) {
    *reinterpret_cast<float4*>(Pd + m * L.ldd + n) = make_float4(v[0], v[1], v[2], v[3]);
  });
  RSTAMP(5);
  const bool bn = k.bn_mean != nullptr;
  if (bn) {     // forward prologue's BatchNorm: scale / shift for the ReLU mask, mean / rstd for xhat (arrays reused: c0..c3)
    LDS_BARRIER();
    for (int c = threadIdx.x; c < Cin; c += NTH) {
      const float rs = rsqrtf(k.bn_var[c] + k.bn_eps), s = k.bn_g[c] * rs;
      c0[c] = s; c1[c] = k.bn_b[c] - k.bn_mean[c] * s; c2[c] = k.bn_mean[c]; c3[c] = rs;
    }
  }
  LDS_BARRIER();
  RSTAMP(6);
  // ---- pro': dropout mask, (one-hot columns dropped), ReLU mask -> dZ (in the panel and to memory) ----
  {
    const int q4 = Cin >> 2, total = TM * q4;
    constexpr int U = 8;
    for (int i0 = threadIdx.x; i0 < total; i0 += NTH * U) {
      f32x4_t x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * NTH, r = i / q4, q = i - r * q4, m = m0 + r;
        x[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (bn && i < total && m < M) x[u] = *reinterpret_cast<const f32x4_t*>(k.X + (size_t)m * k.ldx + 4 * q);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * NTH, r = i / q4, q = i - r * q4, m = m0 + r;
        if (i >= total) continue;
        f32x4_t d = *reinterpret_cast<const f32x4_t*>(Pd + r * L.ldd + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 4 * q + e;
          if (m < M) {
            d[e] = drop_apply(k.drop_in, d[e], (uint64_t)m * Kin + c);
            if (bn && k.bn_relu && !(x[u][e] * c0[c] + c1[c] > 0.f)) d[e] = 0.f;
          } else {
            d[e] = 0.f;
          }
        }
        *reinterpret_cast<f32x4_t*>(Pd + r * L.ldd + 4 * q) = d;
        if (m < M) {
          if (k.g_ntok > 0) {       // entity reduction backward: row (b, t) -> rows (b, j, t)
            const int b = m / k.g_T, t = m - b * k.g_T;
            for (int j = 0; j < k.g_ntok; ++j) {
              f32x4_t o;
#pragma unroll
              for (int e = 0; e < 4; ++e)
                o[e] = k.g_mode == 0 ? (j == 0 ? d[e] : 0.f)
                                     : (k.g_mode == 1 ? d[e] / k.g_ntok : (k.g_arg[(size_t)m * Cin + 4 * q + e] == j ? d[e] : 0.f));
              *reinterpret_cast<f32x4_t*>(k.dX + (((size_t)b * k.g_ntok + j) * k.g_T + t) * k.lddx + 4 * q) = o;
            }
          } else {
            *reinterpret_cast<f32x4_t*>(k.dX + (size_t)m * k.lddx + 4 * q) = d;
          }
        }
      }
    }
  }
  RSTAMP(7);
  if (k.st_part == nullptr) return;
  LDS_BARRIER();
  // ---- s1 = sum dZ, s2 = sum dZ xhat over this workgroup's rows; the last arriver adds the workgroups' partial sums ----
  for (int c = threadIdx.x; c < Cin; c += NTH) {
    float a = 0.f, b2 = 0.f;
    const int cnt = min(TM, M - m0);
    for (int r0 = 0; r0 < cnt; r0 += 8) {       // X rows in batches of 8 loads (a load -> use loop: one L2 round trip per row)
      float xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = k.X[(size_t)(m0 + min(r0 + u, cnt - 1)) * k.ldx + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (r0 + u >= cnt) break;
        const float d = Pd[(r0 + u) * L.ldd + c];
        a += d;
        b2 += d * ((xv[u] - c2[c]) * c3[c]);
      }
    }
    k.st_part[((size_t)blockIdx.x * 2 + 0) * Cin + c] = a;
    k.st_part[((size_t)blockIdx.x * 2 + 1) * Cin + c] = b2;
  }
  RSTAMP(8);
  if (!last_arriver(&g_rl_ticket_b[k.ticket], gridDim.x)) { RSTAMP(9); return; }
  for (int c = threadIdx.x; c < Cin; c += NTH) {
    float a = 0.f, b2 = 0.f;
    const int nwg = (int)gridDim.x;
    for (int w0 = 0; w0 < nwg; w0 += 8) {       // workgroup order, 8 workgroups' partials requested together
      float pa[8], pb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int w = min(w0 + u, nwg - 1);
        pa[u] = k.st_part[((size_t)w * 2 + 0) * Cin + c];
        pb[u] = k.st_part[((size_t)w * 2 + 1) * Cin + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (w0 + u < nwg) { a += pa[u]; b2 += pb[u]; }
    }
    k.s1[c] = a;
    k.s2[c] = b2;
    if (k.dgamma != nullptr) { k.dgamma[c] += b2; k.dbeta[c] += a; }
  }
}

template <typename K>
int launch_fwd(K kern, const RowLinFwdK& k, size_t lds, hipStream_t st, uint64_t& attr) {
  // (the kernels also hold a few bytes of static LDS -- last_arriver's flag -- so the dynamic limit stays below the CU's 160 KB)
  if (mvf_ensure_lds(reinterpret_cast<const void*>(kern), 156 * 1024, attr) != MVF_OK) return MVF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3(ceil_div(k.M, TM)), dim3(NTH), lds, st, k);
  return MVF_OK;
}
template <typename K>
int launch_bwd(K kern, const RowLinBwdK& k, size_t lds, hipStream_t st, uint64_t& attr) {
  // (the kernels also hold a few bytes of static LDS -- last_arriver's flag -- so the dynamic limit stays below the CU's 160 KB)
  if (mvf_ensure_lds(reinterpret_cast<const void*>(kern), 156 * 1024, attr) != MVF_OK) return MVF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3(ceil_div(k.M, TM)), dim3(NTH), lds, st, k);
  return MVF_OK;
}

}  // namespace

// diagnostic: device buffer of slots x 16 int64 (or NULL): the next `slots` row-chain launches of this file record their stage stamps
extern "C" int mvf_rowlin_debug_stamps(long long* buf, int slots) {
  g_rl_stamps = buf; g_rl_stamp_slots = buf != nullptr ? slots : 0; g_rl_stamp_next = 0;
  return MVF_OK;
}

extern "C" int mvf_rowlin_fwd(const MvfRowLinFwd* s, hipStream_t st) {
  MVF_CHECK_ARG(s && s->M > 0 && s->Cin > 0 && s->Cin % 4 == 0 && s->N > 0 && s->N % 128 == 0 && s->X && s->w16 && s->Y && s->ldx >= s->Cin &&
                s->ldx % 4 == 0 && al16(s->X) && al16(s->Y));
  MVF_CHECK_ARG(s->oh_ntok >= 0 && (s->oh_ntok == 0 || s->oh_div > 0) && (s->g_ntok == 0 || (s->g_T > 0 && s->M % s->g_T == 0 && s->g_mode >= 0 && s->g_mode <= 2)));
  MVF_CHECK_ARG((s->bn_mean == nullptr) == (s->bn_var == nullptr) && (s->bn_mean == nullptr || (s->bn_g && s->bn_b)));
  MVF_CHECK_ARG(s->st_part == nullptr || (s->st_mean && s->st_var && (s->st_rmean == nullptr) == (s->st_rvar == nullptr)));
  MVF_CHECK_ARG(s->table == nullptr || s->tab_mod > 0);
  MVF_CHECK_ARG(s->xT == nullptr || (s->Mp % 128 == 0 && s->Mp >= s->M));
  const int Kin = s->Cin + s->oh_ntok;
  if (s->N > 512 || Kin > 512) return MVF_ERR_UNSUPPORTED;
  const RlLds L = rl_lds_fwd(Kin, s->N, s->Cin);
  if (L.total > 156 * 1024) return MVF_ERR_UNSUPPORTED;
  RowLinFwdK k{};
  k.M = s->M; k.Cin = s->Cin; k.Kin = Kin; k.N = s->N; k.Mp = s->Mp; k.X = s->X; k.ldx = s->ldx;
  k.g_ntok = s->g_ntok; k.g_T = s->g_T; k.g_mode = s->g_mode; k.g_arg = s->g_arg;
  k.bn_mean = s->bn_mean; k.bn_var = s->bn_var; k.bn_g = s->bn_g; k.bn_b = s->bn_b; k.bn_eps = s->bn_eps; k.bn_relu = s->bn_relu;
  k.oh_ntok = s->oh_ntok; k.oh_div = s->oh_div > 0 ? s->oh_div : 1;
  k.drop_in = make_drop(s->drop_in); k.drop_out = make_drop(s->drop_out);
  k.w = (const bf16_t*)s->w16; k.bias = s->bias; k.table = s->table; k.tab_mod = s->tab_mod > 0 ? s->tab_mod : 1;
  k.l2norm = s->l2norm; k.l2_eps = s->l2_eps; k.Y = s->Y; k.nrm = s->nrm; k.xT = (bf16_t*)s->xT;
  k.st_part = s->st_part; k.st_mean = s->st_mean; k.st_var = s->st_var; k.st_rmean = s->st_rmean; k.st_rvar = s->st_rvar; k.st_momentum = s->st_momentum;
  k.stamps = next_stamps();
  k.ticket = g_rl_ring_f.take();
  static uint64_t a1 = 0, a2 = 0, a4 = 0, h1 = 0, h2 = 0, h4 = 0;
  int rc;
  if (s->f16) {
    if (s->N >= 512) rc = launch_fwd(rowlin_fwd_kernel<4, true>, k, L.total, st, h4);
    else if (s->N >= 256) rc = launch_fwd(rowlin_fwd_kernel<2, true>, k, L.total, st, h2);
    else rc = launch_fwd(rowlin_fwd_kernel<1, true>, k, L.total, st, h1);
  } else if (s->N >= 512) rc = launch_fwd(rowlin_fwd_kernel<4>, k, L.total, st, a4);
  else if (s->N >= 256) rc = launch_fwd(rowlin_fwd_kernel<2>, k, L.total, st, a2);
  else rc = launch_fwd(rowlin_fwd_kernel<1>, k, L.total, st, a1);
  if (rc != MVF_OK) return rc;
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_rowlin_bwd(const MvfRowLinBwd* s, hipStream_t st) {
  MVF_CHECK_ARG(s && s->M > 0 && s->Cin > 0 && s->Cin % 4 == 0 && s->N > 0 && s->N % 4 == 0 && s->dY && s->w16t && s->dX && al16(s->dY) && al16(s->dX) &&
                s->lddx >= s->Cin && s->lddx % 4 == 0);
  MVF_CHECK_ARG(s->oh_ntok >= 0 && (s->g_ntok == 0 || (s->g_T > 0 && s->M % s->g_T == 0 && s->g_mode >= 0 && s->g_mode <= 2 && (s->g_mode != 2 || s->g_arg))));
  MVF_CHECK_ARG(s->nb_Y == nullptr || (s->nb_mean && s->nb_var && s->nb_g && s->nb_s1 && s->nb_s2 && al16(s->nb_Y)));
  MVF_CHECK_ARG(s->bn_mean == nullptr || (s->bn_var && s->bn_g && s->bn_b && s->X && s->ldx >= s->Cin && s->ldx % 4 == 0 && al16(s->X)));
  MVF_CHECK_ARG(s->st_part == nullptr || (s->bn_mean && s->s1 && s->s2 && (s->dgamma == nullptr) == (s->dbeta == nullptr)));
  MVF_CHECK_ARG(!s->l2norm || (s->l2_y && s->l2_nrm));
  MVF_CHECK_ARG(s->gT == nullptr || (s->Mp % 128 == 0 && s->Mp >= s->M));
  const int Kin = s->Cin + s->oh_ntok;
  if (s->N > 512 || Kin > 512) return MVF_ERR_UNSUPPORTED;
  const RlLdsB L = rl_lds_bwd(Kin, s->N, s->Cin);
  if (L.total > 156 * 1024) return MVF_ERR_UNSUPPORTED;
  RowLinBwdK k{};
  k.M = s->M; k.Cin = s->Cin; k.Kin = Kin; k.N = s->N; k.Mp = s->Mp; k.dY = s->dY;
  k.nb_Y = s->nb_Y; k.nb_mean = s->nb_mean; k.nb_var = s->nb_var; k.nb_g = s->nb_g; k.nb_s1 = s->nb_s1; k.nb_s2 = s->nb_s2;
  k.nb_eps = s->nb_eps; k.nb_inv_count = s->nb_count > 0.f ? 1.0f / s->nb_count : 0.f;
  k.drop_out = make_drop(s->drop_out); k.drop_in = make_drop(s->drop_in);
  k.l2norm = s->l2norm; k.l2_y = s->l2_y; k.l2_nrm = s->l2_nrm; k.l2_eps = s->l2_eps;
  k.wT = (const bf16_t*)s->w16t; k.gT = (bf16_t*)s->gT; k.oh_ntok = s->oh_ntok;
  k.X = s->X; k.ldx = s->ldx; k.bn_mean = s->bn_mean; k.bn_var = s->bn_var; k.bn_g = s->bn_g; k.bn_b = s->bn_b; k.bn_eps = s->bn_eps; k.bn_relu = s->bn_relu;
  k.st_part = s->st_part; k.s1 = s->s1; k.s2 = s->s2; k.dgamma = s->dgamma; k.dbeta = s->dbeta;
  k.g_ntok = s->g_ntok; k.g_T = s->g_T; k.g_mode = s->g_mode; k.g_arg = s->g_arg; k.dX = s->dX; k.lddx = s->lddx;
  const int Kr = (Kin + 63) & ~63;
  k.stamps = next_stamps();
  k.ticket = g_rl_ring_b.take();
  static uint64_t a1 = 0, a2 = 0, a4 = 0;
  int rc;
  if (Kr >= 384) rc = launch_bwd(rowlin_bwd_kernel<4>, k, L.total, st, a4);
  else if (Kr >= 256) rc = launch_bwd(rowlin_bwd_kernel<2>, k, L.total, st, a2);
  else rc = launch_bwd(rowlin_bwd_kernel<1>, k, L.total, st, a1);
  if (rc != MVF_OK) return rc;
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}
