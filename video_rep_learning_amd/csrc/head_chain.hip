// Row-chain kernels of the trainable MV-Former head on the 16-bit matrix cores: a workgroup owns 32 rows of the
// [B*S, d] activation matrix and walks a whole CHAIN of row-wise operators on them -- LayerNorm, Linear (+bias, ReLU,
// dropout, residual), the next LayerNorm, the next Linear ... -- with the activations in LDS and the weights streamed
// from L2 straight into MFMA fragments.  One launch stands for what used to be 5..8 launches of head_gemm.hip /
// head_misc.hip kernels (the reference: one ATen kernel per operator and more).
//
//   mvf_enc_layer_fwd  temporal EncoderLayer (CARL_MVF/models/utils.py:196-226): [attention output -> linear_d2Q + dropout
//                      + residual -> LayerNorm -> fc1 + ReLU -> fc2 + dropout + residual] and, for the NEXT layer,
//                      [LayerNorm -> Q|K|V projection] (utils.py:75-108,147-194); the attention core itself stays in
//                      head_attn_mfma.hip (it mixes the rows of a clip)
//   mvf_enc_layer_bwd  the same chain backwards: [dQKV -> input gradient of the Q|K|V projection -> LayerNorm backward +
//                      residual gradient] and [dropout mask -> fc2^T -> ReLU mask -> fc1^T -> LayerNorm backward + residual
//                      -> dropout mask -> linear_d2Q^T]; every Linear's output gradient is also written TRANSPOSED (bf16)
//                      for the weight-gradient kernel
//   mvf_head_dw        the weight / bias gradients of up to 16 Linears in ONE launch (dW (+)= g^T x over the rows, both
//                      operands row-transposed bf16, so fragments are 16-byte global loads)
//   mvf_head_pack_weights  fp32 master weights -> bf16 [N, K] and bf16 transposed [K, N] copies (once per optimizer step)
//
// Numerics: GEMM operands bf16 (what fp16 autocast does to these layers in the reference, train.py:113-117, with bf16's
// range instead of a loss scaler), fp32 accumulation; MI355X.HEAD_DTYPE fp16 (round 6): the FORWARD GEMMs take IEEE fp16 operands
// (11 significant bits: per-frame embeddings within the north star's 1e-3 of the fp32 path) while every gradient GEMM keeps bf16
// operands -- the forward's saved activations are written as bf16 for the weight-gradient launch -- so no loss scaler is needed; LayerNorm, bias, dropout, residual stream and every saved
// statistic fp32.  oracle/head.py `emulate='bf16'` rounds at the same points.  MI355X.COMPUTE_DTYPE fp32 keeps the fp32
// kernels of head_gemm.hip / head_misc.hip.
//
// gfx950 design: 256 threads = 4 waves, one workgroup per CU (up to 158 KB of LDS panels).  A GEMM stage computes
// out[32, N] = A[32, K] W[N, K]^T with v_mfma_f32_16x16x32_bf16: A fragments by ds_read_b128 from the bf16 LDS panel (row
// pad 16 B), W fragments as 16-byte global loads four k-steps ahead -- a wave owns 64 output columns at a time, so no W
// element is used by two waves and staging it in LDS would only add a barrier.  The operands are swapped in the MFMA
// (D^T = W A^T) so that a lane owns 4 consecutive output columns of one row: float4 epilogue traffic.  M = 768 rows are
// 24 workgroups: latency-bound by design (the step is bound by the frozen backbone beside it; what the head costs it is
// launches), each streaming <= 1.6 MB of bf16 weights from L2.
#include "head_chain.h"

namespace {
using namespace chain;

// LDS carve-up shared by the two encoder kernels (bytes); D, DFF multiples of 64
struct EncLds {
  int ldf, ldb0, ldb1;      // element strides
  size_t pf0, pf1, pb0, pb1, total;
};
__host__ __device__ inline EncLds enc_lds(int D, int DFF, bool bwd) {
  EncLds l;
  l.ldf = D + 4; l.ldb0 = D + 8;
  const int wide = DFF > 3 * D ? DFF : 3 * D;       // the wide bf16 panel also takes the dQKV rows in the backward
  l.ldb1 = (bwd ? wide : DFF) + 8;
  size_t o = 0;
  l.pf0 = o; o += (size_t)TM * l.ldf * 4;
  l.pf1 = o; if (bwd) o += (size_t)TM * l.ldf * 4;
  l.pb0 = o; o += (size_t)TM * l.ldb0 * 2;
  l.pb1 = o; o += (size_t)TM * l.ldb1 * 2;
  l.total = o;
  return l;
}

// measurement knob (tools/chain_probe.py): bit 0 skips the transposed saves, bit 1 cuts every GEMM's k loop to one round, bit 2
// skips the row saves of `a`; results are wrong with any bit set
int g_chain_dbg = 0;
long long* g_chain_stamps = nullptr;   // diagnostic builds of the probe only: wave 0 of workgroup 0 stores s_memrealtime (100 MHz) at stage boundaries
#define STAMP(i) do { if (k.stamps != nullptr && threadIdx.x == 0 && blockIdx.x == 0) k.stamps[i] = wall_clock64(); } while (0)

struct EncFwdK {
  int M, D, DFF, Mp, dbg;
  float eps;
  const float *o, *x_in;
  const bf16_t *wo, *w1, *w2, *wqkv;
  const float *bo, *b1, *b2, *bqkv, *g1, *be1, *g0, *be0;
  Drop da, df;
  float *x1, *mean1, *rstd1, *x2, *qkv, *mean0, *rstd0;
  bf16_t *a, *oT, *h1T, *aT, *h0T;
  long long* stamps;
};

template <bool F16>
__global__ __launch_bounds__(NTH) void enc_fwd_kernel(EncFwdK k) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const EncLds L = enc_lds(k.D, k.DFF, false);
  float* Pf = reinterpret_cast<float*>(sm + L.pf0);
  bf16_t* Pb0 = reinterpret_cast<bf16_t*>(sm + L.pb0);
  bf16_t* Pb1 = reinterpret_cast<bf16_t*>(sm + L.pb1);
  const int m0 = blockIdx.x * TM, M = k.M, D = k.D, DFF = k.DFF;
  if (k.dbg & 1) { k.oT = k.h1T = k.aT = k.h0T = nullptr; }
  if (k.dbg & 4) k.a = nullptr;
  const int KD = (k.dbg & 2) ? 128 : D, KF = (k.dbg & 2) ? 128 : DFF;
  STAMP(0);
  if (k.o != nullptr) {
    // ---- x1 = x + drop(o Wo^T + bo) ----
    load_rows_bf16<F16>(k.o, D, m0, M, D, Pb0, L.ldb0);
    LDS_BARRIER();
    STAMP(1);
    store_T<F16>(Pb0, L.ldb0, D, k.oT, k.Mp, m0, M);
    STAMP(2);
    chain_gemm<2, 4, F16>(Pb0, L.ldb0, KD, k.wo, D, [&](int m, int n) {
      Aux2 a;
      a.b = *reinterpret_cast<const float4*>(k.bo + n);
      a.r = *reinterpret_cast<const float4*>(k.x_in + (size_t)min(m0 + m, M - 1) * D + n);
      return a;
    }, [&](int m, int n, const f32x4_t& v, const Aux2& ax) {
      const int gm = m0 + m;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gm < M) {
        const float4 bb = ax.b, xi = ax.r;
        const uint64_t idx = (uint64_t)gm * D + n;
        r.x = xi.x + drop_apply(k.da, v[0] + bb.x, idx);
        r.y = xi.y + drop_apply(k.da, v[1] + bb.y, idx + 1);
        r.z = xi.z + drop_apply(k.da, v[2] + bb.z, idx + 2);
        r.w = xi.w + drop_apply(k.da, v[3] + bb.w, idx + 3);
        if (k.x1) *reinterpret_cast<float4*>(k.x1 + (size_t)gm * D + n) = r;
      }
      *reinterpret_cast<float4*>(Pf + m * L.ldf + n) = r;
    });
    LDS_BARRIER();
    STAMP(3);
    // ---- h1 = LN(x1);  a = relu(h1 W1^T + b1) ----
    ln_panel<F16>(Pf, L.ldf, D, k.g1, k.be1, k.eps, Pb0, L.ldb0, k.mean1, k.rstd1, m0, M);
    LDS_BARRIER();
    STAMP(4);
    store_T<F16>(Pb0, L.ldb0, D, k.h1T, k.Mp, m0, M);
    STAMP(5);
    chain_gemm<4, 4, F16>(Pb0, L.ldb0, KD, k.w1, DFF, [&](int, int n) {
      Aux1 a;
      a.b = *reinterpret_cast<const float4*>(k.b1 + n);
      return a;
    }, [&](int m, int n, const f32x4_t& v, const Aux1& ax) {
      const float4 bb = ax.b;
      *reinterpret_cast<u32x2_t*>(Pb1 + m * L.ldb1 + n) =
          (u32x2_t){pack16x2<F16>(fmaxf(v[0] + bb.x, 0.f), fmaxf(v[1] + bb.y, 0.f)), pack16x2<F16>(fmaxf(v[2] + bb.z, 0.f), fmaxf(v[3] + bb.w, 0.f))};
    });
    LDS_BARRIER();
    STAMP(6);
    store_rows_bf16(Pb1, L.ldb1, DFF, k.a, m0, M);
    store_T<F16>(Pb1, L.ldb1, DFF, k.aT, k.Mp, m0, M);
    STAMP(7);
    // ---- x2 = x1 + drop(a W2^T + b2) ----
    chain_gemm<2, 4, F16>(Pb1, L.ldb1, KF, k.w2, D, [&](int, int n) {
      Aux1 a;
      a.b = *reinterpret_cast<const float4*>(k.b2 + n);
      return a;
    }, [&](int m, int n, const f32x4_t& v, const Aux1& ax) {
      const int gm = m0 + m;
      float4 r = *reinterpret_cast<const float4*>(Pf + m * L.ldf + n);
      if (gm < M) {
        const float4 bb = ax.b;
        const uint64_t idx = (uint64_t)gm * D + n;
        r.x += drop_apply(k.df, v[0] + bb.x, idx);
        r.y += drop_apply(k.df, v[1] + bb.y, idx + 1);
        r.z += drop_apply(k.df, v[2] + bb.z, idx + 2);
        r.w += drop_apply(k.df, v[3] + bb.w, idx + 3);
        if (k.x2) *reinterpret_cast<float4*>(k.x2 + (size_t)gm * D + n) = r;
      }
      *reinterpret_cast<float4*>(Pf + m * L.ldf + n) = r;
    });
    LDS_BARRIER();
    STAMP(8);
  } else if (k.wqkv != nullptr) {
    load_rows_f32(k.x_in, D, m0, M, D, Pf, L.ldf);
    LDS_BARRIER();
  }
  if (k.wqkv != nullptr) {
    // ---- the next layer's h0 = LN(x);  qkv = h0 Wqkv^T + bqkv ----
    ln_panel<F16>(Pf, L.ldf, D, k.g0, k.be0, k.eps, Pb0, L.ldb0, k.mean0, k.rstd0, m0, M);
    LDS_BARRIER();
    STAMP(9);
    store_T<F16>(Pb0, L.ldb0, D, k.h0T, k.Mp, m0, M);
    STAMP(10);
    chain_gemm<2, 4, F16>(Pb0, L.ldb0, KD, k.wqkv, 3 * D, [&](int, int n) {
      Aux1 a;
      a.b = *reinterpret_cast<const float4*>(k.bqkv + n);
      return a;
    }, [&](int m, int n, const f32x4_t& v, const Aux1& ax) {
      const int gm = m0 + m;
      if (gm < M) {
        const float4 bb = ax.b;
        *reinterpret_cast<float4*>(k.qkv + (size_t)gm * 3 * D + n) = make_float4(v[0] + bb.x, v[1] + bb.y, v[2] + bb.z, v[3] + bb.w);
      }
    });
    STAMP(11);
  }
}

struct EncBwdK {
  int M, D, DFF, Mp;
  const float *dqkv, *x_in, *mean0, *rstd0, *g0, *dres;
  const bf16_t *wqkvT, *w2T, *w1T, *woT, *a;
  float *dg0, *db0, *dx_out;
  bf16_t *dqkvT, *g2T, *duT, *goT;
  Drop df, da;
  const float *x1, *mean1, *rstd1, *g1;
  float *dg1, *db1, *dx1_out, *d_o;
};

__global__ __launch_bounds__(NTH) void enc_bwd_kernel(EncBwdK k) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const EncLds L = enc_lds(k.D, k.DFF, true);
  float* Pf0 = reinterpret_cast<float*>(sm + L.pf0);      // the gradient on the residual stream
  float* Pf1 = reinterpret_cast<float*>(sm + L.pf1);      // a LayerNorm's output gradient
  bf16_t* Pb0 = reinterpret_cast<bf16_t*>(sm + L.pb0);
  bf16_t* Pb1 = reinterpret_cast<bf16_t*>(sm + L.pb1);
  const int m0 = blockIdx.x * TM, M = k.M, D = k.D, DFF = k.DFF;
  load_rows_f32(k.dres, D, m0, M, D, Pf0, L.ldf);
  if (k.dqkv != nullptr) {
    // ---- dh0 = dqkv Wqkv;  dx = dres + LN0'(dh0) ----
    load_rows_bf16(k.dqkv, 3 * D, m0, M, 3 * D, Pb1, L.ldb1);
    LDS_BARRIER();
    store_T(Pb1, L.ldb1, 3 * D, k.dqkvT, k.Mp, m0, M);
    chain_gemm<2, 4>(Pb1, L.ldb1, 3 * D, k.wqkvT, D, [](int, int) { return NoAux{}; }, [&](int m, int n, const f32x4_t& v, const NoAux&) {
      *reinterpret_cast<float4*>(Pf1 + m * L.ldf + n) = make_float4(v[0], v[1], v[2], v[3]);
    });
    LDS_BARRIER();
    ln_bwd_panel(Pf1, L.ldf, Pf0, L.ldf, D, k.x_in, k.mean0, k.rstd0, k.g0, k.dg0, k.db0, m0, M);
    LDS_BARRIER();
    if (k.dx_out != nullptr) {
      const int c4 = D >> 2;
      for (int i = threadIdx.x; i < TM * c4; i += NTH) {
        const int r = i / c4, q = i - r * c4;
        if (m0 + r < M) *reinterpret_cast<float4*>(k.dx_out + (size_t)(m0 + r) * D + 4 * q) = *reinterpret_cast<const float4*>(Pf0 + r * L.ldf + 4 * q);
      }
    }
  } else {
    LDS_BARRIER();
  }
  if (k.w2T == nullptr) return;
  // ---- g2 = mask_f(dy);  du = (g2 W2) [a > 0];  dh1 = du W1;  dx1 = dy + LN1'(dh1) ----
  mask_to_bf16(Pf0, L.ldf, D, k.df, Pb0, L.ldb0, m0, M);
  LDS_BARRIER();
  store_T(Pb0, L.ldb0, D, k.g2T, k.Mp, m0, M);
  struct AuxA { u32x2_t a; };
  chain_gemm<4, 4>(Pb0, L.ldb0, D, k.w2T, DFF, [&](int m, int n) {
    AuxA x;
    x.a = *reinterpret_cast<const u32x2_t*>(k.a + (size_t)min(m0 + m, M - 1) * DFF + n);
    return x;
  }, [&](int m, int n, const f32x4_t& v, const AuxA& ax) {
    const u32x2_t av = ax.a;
    const float d0 = (av[0] & 0xffffu) != 0u && !(av[0] & 0x8000u) ? v[0] : 0.f;
    const float d1 = (av[0] >> 16) != 0u && !(av[0] & 0x80000000u) ? v[1] : 0.f;
    const float d2 = (av[1] & 0xffffu) != 0u && !(av[1] & 0x8000u) ? v[2] : 0.f;
    const float d3 = (av[1] >> 16) != 0u && !(av[1] & 0x80000000u) ? v[3] : 0.f;
    *reinterpret_cast<u32x2_t*>(Pb1 + m * L.ldb1 + n) = (u32x2_t){pack_bf16x2(d0, d1), pack_bf16x2(d2, d3)};
  });
  LDS_BARRIER();
  store_T(Pb1, L.ldb1, DFF, k.duT, k.Mp, m0, M);
  chain_gemm<2, 4>(Pb1, L.ldb1, DFF, k.w1T, D, [](int, int) { return NoAux{}; }, [&](int m, int n, const f32x4_t& v, const NoAux&) {
    *reinterpret_cast<float4*>(Pf1 + m * L.ldf + n) = make_float4(v[0], v[1], v[2], v[3]);
  });
  LDS_BARRIER();
  ln_bwd_panel(Pf1, L.ldf, Pf0, L.ldf, D, k.x1, k.mean1, k.rstd1, k.g1, k.dg1, k.db1, m0, M);
  LDS_BARRIER();
  // ---- dx1 out;  g_o = mask_a(dx1);  d_o = g_o Wo ----
  {
    const int c4 = D >> 2;
    for (int i = threadIdx.x; i < TM * c4; i += NTH) {
      const int r = i / c4, q = i - r * c4;
      if (m0 + r < M) *reinterpret_cast<float4*>(k.dx1_out + (size_t)(m0 + r) * D + 4 * q) = *reinterpret_cast<const float4*>(Pf0 + r * L.ldf + 4 * q);
    }
  }
  mask_to_bf16(Pf0, L.ldf, D, k.da, Pb0, L.ldb0, m0, M);
  LDS_BARRIER();
  store_T(Pb0, L.ldb0, D, k.goT, k.Mp, m0, M);
  chain_gemm<2, 4>(Pb0, L.ldb0, D, k.woT, D, [](int, int) { return NoAux{}; }, [&](int m, int n, const f32x4_t& v, const NoAux&) {
    const int gm = m0 + m;
    if (gm < M) *reinterpret_cast<float4*>(k.d_o + (size_t)gm * D + n) = make_float4(v[0], v[1], v[2], v[3]);
  });
}

// ---------------------------------------------------------------------------------------------------------------
// weight / bias gradients of several Linears in one launch.  Problem p: dW[n][k] (+)= sum_m gT[n][m] xT[k][m],
// db[n] (+)= sum_m gT[n][m]; gT, xT: FM images (rows n resp. k, reduction m) written by store_T.  One workgroup = one
// 64 (n) x 64 (k) tile, a wave = 32 x 32 of it; both operands come straight from global memory as 1 KB fragment loads, 4 m-steps
// ahead (ring refilled in place, see chain_gemm).
// ---------------------------------------------------------------------------------------------------------------
constexpr int DW_MAX = 16;
struct DwProb { const bf16_t* gT; const bf16_t* xT; float* dw; long lddw; float* db; int N, K, tiles_k, tile0; };
struct DwArgs { DwProb p[DW_MAX]; int n, Mp, accumulate; };

__global__ __launch_bounds__(NTH2) void head_dw_kernel(DwArgs a) {
  int pi = 0;
  const int b = blockIdx.x;
#pragma unroll 1
  for (int i = 1; i < a.n; ++i)
    if (b >= a.p[i].tile0) pi = i;
  const DwProb& P = a.p[pi];
  const int t = b - P.tile0, tn = t / P.tiles_k, tk = t - tn * P.tiles_k;
  const int lane = threadIdx.x & 63, wave = wave_id(), c = lane & 15, g = lane >> 4;
  const int n0 = tn * 64 + (wave >> 1) * 32, k0 = tk * 64 + (wave & 1) * 32;
  const int nsteps = a.Mp >> 5;         // Mp % 128 == 0: nsteps % PF == 0
  constexpr int PF = 4;
  // fragment bases of the wave's two n tiles and two k tiles (tile rows past N / K exist in the padded images: computed, dropped)
  const bf16_t* gp[2];
  const bf16_t* xp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    gp[i] = P.gT + ((size_t)((n0 >> 4) + i) * nsteps * 64 + lane) * 8;
    xp[i] = P.xT + ((size_t)((k0 >> 4) + i) * nsteps * 64 + lane) * 8;
  }
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float bs[2] = {0.f, 0.f};
  const bool want_b = P.db != nullptr && tk == 0 && (wave & 1) == 0;
  bf16x8_t gq[PF][2], xq[PF][2];
#pragma unroll
  for (int p = 0; p < PF; ++p) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      gq[p][i] = *reinterpret_cast<const bf16x8_t*>(gp[i] + (size_t)p * 512);
      xq[p][i] = *reinterpret_cast<const bf16x8_t*>(xp[i] + (size_t)p * 512);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  for (int s0 = 0; s0 < nsteps; s0 += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      // D[k-row 4g + r][n-col c] : lane owns dW[n = c][k .. k+3]
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xq[p][j], gq[p][i], acc[i][j], 0, 0, 0);
      if (want_b) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) bs[i] += bf16_to_f32((bf16_t)gq[p][i][e]);
      }
      const int mr = min(s0 + p + PF, nsteps - 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        gq[p][i] = *reinterpret_cast<const bf16x8_t*>(gp[i] + (size_t)mr * 512);
        xq[p][i] = *reinterpret_cast<const bf16x8_t*>(xp[i] + (size_t)mr * 512);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + 16 * i + c;
    if (want_b) {
      float v = bs[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0 && n < P.N) P.db[n] = a.accumulate ? P.db[n] + v : v;
    }
    if (n >= P.N) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kk = k0 + 16 * j + 4 * g;
      float* dp = P.dw + (size_t)n * P.lddw + kk;
      if (kk + 3 < P.K && (P.lddw & 3) == 0 && ((uintptr_t)P.dw & 15) == 0) {
        float4 o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if (a.accumulate) { const float4 q = *reinterpret_cast<const float4*>(dp); o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
        *reinterpret_cast<float4*>(dp) = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kk + r < P.K) dp[r] = a.accumulate ? dp[r] + acc[i][j][r] : acc[i][j][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 master weights -> bf16 FM operands.  Entry e: w [N, K] (row stride ld) -> w16 = FM image of W (rows n, reduction k: the
// forward operand) and w16t = FM image of W^T (rows k, reduction n: the input-gradient operand).  One thread per 16-byte piece.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PACK_MAX = 32;
struct PackEnt { const float* w; long ld; int N, K; unsigned piece0, pieces16; bf16_t* w16; bf16_t* w16t; int f16; };
struct PackArgs { PackEnt e[PACK_MAX]; int n; };

__global__ __launch_bounds__(NTH2) void head_pack_kernel(PackArgs a, unsigned total) {
  for (unsigned pid = blockIdx.x * NTH2 + threadIdx.x; pid < total; pid += gridDim.x * NTH2) {
    int ei = 0;
#pragma unroll 1
    for (int i = 1; i < a.n; ++i)
      if (pid >= a.e[i].piece0) ei = i;
    const PackEnt& E = a.e[ei];
    unsigned q = pid - E.piece0;
    const bool tr = q >= E.pieces16;            // second half of the entry's pieces: the transposed image
    if (tr) q -= E.pieces16;
    bf16_t* dst = tr ? E.w16t : E.w16;
    if (dst == nullptr) continue;
    const int rows = tr ? E.K : E.N, red = tr ? E.N : E.K;
    const int steps = fm_steps(red);
    const int lane = q & 63, blk = q >> 6, rt = blk / steps, stp = blk - rt * steps;
    const int r = rt * 16 + (lane & 15), c0 = stp * 32 + (lane >> 4) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cc = c0 + e;
      const bool in = r < rows && cc < red;
      v[e] = in ? (tr ? E.w[(size_t)cc * E.ld + r] : E.w[(size_t)r * E.ld + cc]) : 0.f;
    }
    // the forward operand (w16) of an fp16-head entry is IEEE fp16; the input-gradient operand (w16t) is bf16 in every mode
    *reinterpret_cast<u32x4_t*>(dst + (size_t)q * 8) = (E.f16 && !tr)
        ? (u32x4_t){pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3]), pack_f16x2(v[4], v[5]), pack_f16x2(v[6], v[7])}
        : (u32x4_t){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  }
}

// Reads `bytes` of read-only data into every XCD's L2: workgroup b takes slice b / 8 of grid / 8 slices (workgroups are dealt
// round-robin over the 8 XCDs, so the 32 workgroups of an XCD cover the whole range; placement is a speed matter only).
__global__ __launch_bounds__(256) void l2_warm_kernel(const u32x4_t* __restrict__ p, size_t n16, unsigned* sink) {
  const size_t slices = gridDim.x / 8, sl = blockIdx.x / 8;
  const size_t per = (n16 + slices - 1) / slices, beg = sl * per, end = beg + per < n16 ? beg + per : n16;
  unsigned acc = 0;
  for (size_t i = beg + threadIdx.x; i < end; i += 256) {
    const u32x4_t v = __builtin_nontemporal_load(p + i);
    acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x9e3779b9u && sink != nullptr) *sink = acc;     // keeps the loads alive; practically never true
}

}  // namespace

extern "C" int mvf_head_chain_debug(int bits) {
  g_chain_dbg = bits;
  return MVF_OK;
}
extern "C" int mvf_head_chain_debug_stamps(long long* stamps16) {
  g_chain_stamps = stamps16;
  return MVF_OK;
}

extern "C" int mvf_head_l2_warm(const void* p, size_t bytes, hipStream_t st) {
  MVF_CHECK_ARG(p && bytes >= 16 && al16(p));
  hipLaunchKernelGGL(l2_warm_kernel, dim3(256), dim3(256), 0, st, reinterpret_cast<const u32x4_t*>(p), bytes / 16, (unsigned*)nullptr);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_head_pack_weights(const MvfPackEntry* entries_host, int n, hipStream_t st) {
  MVF_CHECK_ARG(entries_host && n > 0 && n <= PACK_MAX);
  PackArgs a{};
  size_t pieces = 0;
  for (int i = 0; i < n; ++i) {
    const MvfPackEntry& s = entries_host[i];
    MVF_CHECK_ARG(s.w && s.N > 0 && s.K > 0 && s.ld >= s.K && (s.w16 || s.w16t) && al16(s.w16) && al16(s.w16t));
    PackEnt& e = a.e[i];
    e.w = s.w; e.ld = s.ld; e.N = s.N; e.K = s.K; e.w16 = (bf16_t*)s.w16; e.w16t = (bf16_t*)s.w16t; e.f16 = s.f16 != 0;
    e.piece0 = (unsigned)pieces;
    e.pieces16 = (unsigned)(fm_elems(s.N, s.K) / 8);
    pieces += fm_elems(s.N, s.K) / 8 + fm_elems(s.K, s.N) / 8;
  }
  MVF_CHECK_ARG(pieces < (1ull << 31));
  a.n = n;
  const int grid = (int)std::min<size_t>((pieces + NTH2 - 1) / NTH2, 2048);
  hipLaunchKernelGGL(head_pack_kernel, dim3(grid), dim3(NTH2), 0, st, a, (unsigned)pieces);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" size_t mvf_head_pack_elems(int N, int K, int transposed) {
  return transposed ? fm_elems(K, N) : fm_elems(N, K);
}

extern "C" int mvf_enc_layer_fwd(const MvfEncFwd* s, hipStream_t st) {
  MVF_CHECK_ARG(s && s->M > 0 && s->D > 0 && s->D % 256 == 0 && s->DFF % 256 == 0 && s->D <= 512 && s->x_in);
  MVF_CHECK_ARG(s->o != nullptr || s->wqkv != nullptr);
  MVF_CHECK_ARG(s->Mp % 128 == 0 && s->Mp >= s->M);
  if (s->o) MVF_CHECK_ARG(s->wo && s->w1 && s->w2 && s->bo && s->b1 && s->b2 && s->ln1_g && s->ln1_b && al16(s->o) && al16(s->x_in));
  if (s->wqkv) MVF_CHECK_ARG(s->bqkv && s->ln0_g && s->ln0_b && s->qkv && al16(s->qkv));
  const EncLds L = enc_lds(s->D, s->DFF, false);
  if (L.total > 160 * 1024) return MVF_ERR_UNSUPPORTED;
  EncFwdK k{};
  k.M = s->M; k.D = s->D; k.DFF = s->DFF; k.Mp = s->Mp; k.eps = s->ln_eps; k.dbg = g_chain_dbg; k.stamps = g_chain_stamps;
  k.o = s->o; k.x_in = s->x_in;
  k.wo = (const bf16_t*)s->wo; k.w1 = (const bf16_t*)s->w1; k.w2 = (const bf16_t*)s->w2; k.wqkv = (const bf16_t*)s->wqkv;
  k.bo = s->bo; k.b1 = s->b1; k.b2 = s->b2; k.bqkv = s->bqkv; k.g1 = s->ln1_g; k.be1 = s->ln1_b; k.g0 = s->ln0_g; k.be0 = s->ln0_b;
  k.da = make_drop(s->drop_attn); k.df = make_drop(s->drop_ffn);
  k.x1 = s->x1; k.mean1 = s->mean1; k.rstd1 = s->rstd1; k.x2 = s->x2; k.qkv = s->qkv; k.mean0 = s->mean0; k.rstd0 = s->rstd0;
  k.a = (bf16_t*)s->a; k.oT = (bf16_t*)s->oT; k.h1T = (bf16_t*)s->h1T; k.aT = (bf16_t*)s->aT; k.h0T = (bf16_t*)s->h0T;
  static uint64_t attr[2] = {0, 0};
  if (s->f16) {
    if (mvf_ensure_lds(reinterpret_cast<const void*>(enc_fwd_kernel<true>), 160 * 1024, attr[1]) != MVF_OK) return MVF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(enc_fwd_kernel<true>, dim3(ceil_div(s->M, TM)), dim3(NTH), L.total, st, k);
  } else {
    if (mvf_ensure_lds(reinterpret_cast<const void*>(enc_fwd_kernel<false>), 160 * 1024, attr[0]) != MVF_OK) return MVF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(enc_fwd_kernel<false>, dim3(ceil_div(s->M, TM)), dim3(NTH), L.total, st, k);
  }
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_enc_layer_bwd(const MvfEncBwd* s, hipStream_t st) {
  MVF_CHECK_ARG(s && s->M > 0 && s->D > 0 && s->D % 256 == 0 && s->DFF % 256 == 0 && s->D <= 512 && s->dres && al16(s->dres));
  MVF_CHECK_ARG(s->dqkv != nullptr || s->w2T != nullptr);
  MVF_CHECK_ARG(s->Mp % 128 == 0 && s->Mp >= s->M);
  if (s->dqkv) MVF_CHECK_ARG(s->wqkvT && s->x_in && s->mean0 && s->rstd0 && s->ln0_g && al16(s->dqkv) && (s->dx_out || s->w2T) &&
                             ((s->dln0_g == nullptr) == (s->dln0_b == nullptr)));
  if (s->w2T) MVF_CHECK_ARG(s->w1T && s->woT && s->a && s->x1 && s->mean1 && s->rstd1 && s->ln1_g && s->dx1_out && s->d_o &&
                            al16(s->d_o) && al16(s->dx1_out) && ((s->dln1_g == nullptr) == (s->dln1_b == nullptr)));
  const EncLds L = enc_lds(s->D, s->DFF, true);
  if (L.total > 160 * 1024) return MVF_ERR_UNSUPPORTED;
  EncBwdK k{};
  k.M = s->M; k.D = s->D; k.DFF = s->DFF; k.Mp = s->Mp;
  k.dqkv = s->dqkv; k.x_in = s->x_in; k.mean0 = s->mean0; k.rstd0 = s->rstd0; k.g0 = s->ln0_g; k.dres = s->dres;
  k.wqkvT = (const bf16_t*)s->wqkvT; k.w2T = (const bf16_t*)s->w2T; k.w1T = (const bf16_t*)s->w1T; k.woT = (const bf16_t*)s->woT;
  k.a = (const bf16_t*)s->a;
  k.dg0 = s->dln0_g; k.db0 = s->dln0_b; k.dx_out = s->dx_out;
  k.dqkvT = (bf16_t*)s->dqkvT; k.g2T = (bf16_t*)s->g2T; k.duT = (bf16_t*)s->duT; k.goT = (bf16_t*)s->goT;
  k.df = make_drop(s->drop_ffn); k.da = make_drop(s->drop_attn);
  k.x1 = s->x1; k.mean1 = s->mean1; k.rstd1 = s->rstd1; k.g1 = s->ln1_g; k.dg1 = s->dln1_g; k.db1 = s->dln1_b;
  k.dx1_out = s->dx1_out; k.d_o = s->d_o;
  static uint64_t attr = 0;
  if (mvf_ensure_lds(reinterpret_cast<const void*>(enc_bwd_kernel), 160 * 1024, attr) != MVF_OK) return MVF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(enc_bwd_kernel, dim3(ceil_div(s->M, TM)), dim3(NTH), L.total, st, k);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_head_dw(const MvfDwProblem* probs_host, int n, int Mp, int accumulate, hipStream_t st) {
  MVF_CHECK_ARG(probs_host && n > 0 && n <= DW_MAX && Mp > 0 && Mp % 128 == 0);
  DwArgs a{};
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    const MvfDwProblem& s = probs_host[i];
    MVF_CHECK_ARG(s.gT && s.xT && s.dw && s.N > 0 && s.K > 0 && s.lddw >= s.K && al16(s.gT) && al16(s.xT));
    DwProb& p = a.p[i];
    p.gT = (const bf16_t*)s.gT; p.xT = (const bf16_t*)s.xT; p.dw = s.dw; p.lddw = s.lddw; p.db = s.db; p.N = s.N; p.K = s.K;
    p.tiles_k = ceil_div(s.K, 64); p.tile0 = tiles;
    tiles += ceil_div(s.N, 64) * p.tiles_k;
  }
  a.n = n; a.Mp = Mp; a.accumulate = accumulate;
  hipLaunchKernelGGL(head_dw_kernel, dim3(tiles), dim3(NTH2), 0, st, a);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}
