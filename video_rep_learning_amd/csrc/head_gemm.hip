// fp32 MFMA GEMM with generic strides for the (small, latency-bound) trainable head:
//     C[m, n] (+)= post( act( alpha * sum_k pre(A(m,k)) * B(k,n) + bias[n] + table[idx(m), n] ) )
// A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]: one kernel serves forward (x . W^T), input
// gradient (dy . W) and weight gradient (dy^T . x) of every nn.Linear on the path
// (CARL_MVF/models/mvformer.py:77,86,97; models/utils.py:65-68,182-183; resnet_c2d.py:117-120) by
// passing different strides -- no transposed copies.
//
// Fusions (each removes a launch and an HBM/L2 round trip of a [768, <=1024] fp32 tensor):
//   pre  : A elements can be masked on load -- ReLU backward (dy * [y > 0], models/utils.py:190) or dropout backward
//          (dy * keep / (1-p), the counter-based mask of mvf_dropout_add) -- so the backward of `drop(relu(fc(x)))`
//          chains needs no elementwise kernels;
//   post : y = resid + dropout(v): the residual connection `x + drop(sub(LN(x)))` of models/utils.py:153-159 ends in
//          the epilogue of the sub-layer's last Linear;
//   rowsum: out[m] (+)= sum_k pre(A(m,k)) -- in the weight-gradient problem A = dy^T, so this IS the bias gradient;
//   mvf_hlinear_bwd: dX, dW and db of one Linear in ONE launch (two tile ranges of one grid).
//
// gfx950 design: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).  The matrices are <= a few MB and L2-resident and
// M is 768 rows, so the kernel is latency- not bandwidth-bound: fragments are loaded straight from global/L2 into the
// MFMA operand registers (float4 along k when the operand is k-contiguous, using the k-permutation trick: element s
// of lane (r, g) is k = k0 + 4g + s for BOTH operands), no LDS, no barriers, and the NEXT k-step's fragments are
// requested before the current step's 16 MFMAs are issued (register double buffer) so that L2 latency hides under
// the matrix pipe.  64x64 tile per workgroup (4 waves x 32x32); operands swapped so a lane owns 4 consecutive n.
#include "common.h"
#include "mvf_hip_internal.h"

namespace {

struct HGemmArgs {
  const float* A; long sam, sak;
  const float* B; long sbk, sbn;
  float* C; long ldc;
  const float* bias;
  const float* table; long tab_si, tab_sn; int tab_div, tab_mod;
  int M, N, K;
  int relu, accumulate;
  float alpha;
  // pre-op on A: 0 none, 1 relu mask (amask[same index] > 0), 2 dropout mask
  int a_mode;
  const float* amask;
  // dropout parameters of the A pre-op (a_mode 2) or of the epilogue (post_drop)
  uint32_t d_thresh; float d_scale; uint64_t d_seed, d_offset;
  int post_drop;                      // epilogue: v = dropout(v) before the residual add
  const float* resid; long ldr;       // epilogue: C = resid + dropout(v)
  float* rowsum; int rowsum_acc;      // rowsum[m] (+)= sum_k pre(A(m,k)); written by the blockIdx.x == 0 column of tiles
};

template <bool VEC, int MODE>
__device__ __forceinline__ f32x4_t load_frag(const HGemmArgs& a, const float* base, long row_off, long sk, int k0, int g,
                                             int K) {
  // returns {X(row, k0+4g+0..3)}; zero beyond K.  MODE != 0 only for the A operand.
  f32x4_t v;
  const int k = k0 + 4 * g;
  if constexpr (VEC) {
    v = *reinterpret_cast<const f32x4_t*>(base + row_off + k);
    if constexpr (MODE == 1) {
      const f32x4_t m = *reinterpret_cast<const f32x4_t*>(a.amask + row_off + k);
#pragma unroll
      for (int s = 0; s < 4; ++s) v[s] = m[s] > 0.f ? v[s] : 0.f;
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        v[s] = drop_keep(a.d_seed, a.d_offset, (uint64_t)(row_off + k + s), a.d_thresh) ? v[s] * a.d_scale : 0.f;
    }
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const long idx = row_off + (long)(k + s) * sk;
      float x = (k + s < K) ? base[idx] : 0.f;
      if constexpr (MODE == 1) x = (k + s < K && a.amask[idx] > 0.f) ? x : 0.f;
      if constexpr (MODE == 2) x = (k + s < K && drop_keep(a.d_seed, a.d_offset, (uint64_t)idx, a.d_thresh)) ? x * a.d_scale : 0.f;
      v[s] = x;
    }
  }
  return v;
}

// one 64x64 output tile (bx, by) of the problem `a`
template <bool AVEC, bool BVEC, int MODE>
__device__ __forceinline__ void hgemm_tile(const HGemmArgs& a, int bx, int by) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int m_base = by * 64 + (wave >> 1) * 32;
  const int n_base = bx * 64 + (wave & 1) * 32;
  long aoff[2], boff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    aoff[i] = (long)min(m_base + i * 16 + li, a.M - 1) * a.sam;
    boff[i] = (long)min(n_base + i * 16 + li, a.N - 1) * a.sbn;
  }
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float rs[2] = {0.f, 0.f};
  const bool want_rs = a.rowsum != nullptr && bx == 0 && (wave & 1) == 0;

  f32x4_t af[2], bf[2], afn[2], bfn[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) af[i] = load_frag<AVEC, MODE>(a, a.A, aoff[i], a.sak, 0, g, a.K);
#pragma unroll
  for (int j = 0; j < 2; ++j) bf[j] = load_frag<BVEC, 0>(a, a.B, boff[j], a.sbk, 0, g, a.K);
  for (int k0 = 0; k0 < a.K; k0 += 16) {
    const int kn = k0 + 16;
    if (kn < a.K) {   // request the next step's fragments before this step's MFMAs
#pragma unroll
      for (int i = 0; i < 2; ++i) afn[i] = load_frag<AVEC, MODE>(a, a.A, aoff[i], a.sak, kn, g, a.K);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfn[j] = load_frag<BVEC, 0>(a, a.B, boff[j], a.sbk, kn, g, a.K);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);
    if (want_rs) {
#pragma unroll
      for (int i = 0; i < 2; ++i) rs[i] += af[i][0] + af[i][1] + af[i][2] + af[i][3];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) { af[i] = afn[i]; bf[i] = bfn[i]; }
  }

  if (want_rs) {   // combine the 4 k-groups (lanes li, li+16, li+32, li+48)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float v = rs[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int m = m_base + i * 16 + li;
      if (g == 0 && m < a.M) a.rowsum[m] = a.rowsum_acc ? a.rowsum[m] + v : v;
    }
  }

  const bool vec_out = (a.ldc % 4 == 0) && (a.N % 4 == 0) && (((uintptr_t)a.C & 15) == 0) &&
                       (a.resid == nullptr || (a.ldr % 4 == 0 && ((uintptr_t)a.resid & 15) == 0));
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m_base + i * 16 + li;
    if (m >= a.M) continue;
    const float* trow = a.table ? a.table + (long)((m / a.tab_div) % a.tab_mod) * a.tab_si : nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n_base + j * 16 + 4 * g;
      if (n >= a.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = a.alpha * acc[i][j][r];
        if (n + r < a.N) {
          if (a.bias) v[r] += a.bias[n + r];
          if (trow) v[r] += trow[(long)(n + r) * a.tab_sn];
        }
        if (a.relu) v[r] = fmaxf(v[r], 0.f);
      }
      float* cp = a.C + (long)m * a.ldc + n;
      if (a.post_drop) {          // dropout(v); mask index = element index of the dense [M, ldc] output
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] = drop_keep(a.d_seed, a.d_offset, (uint64_t)((long)m * a.ldc + n + r), a.d_thresh) ? v[r] * a.d_scale : 0.f;
      }
      if (a.resid != nullptr) {   // y = resid + ...
        const float* rp = a.resid + (long)m * a.ldr + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < a.N) v[r] += rp[r];
      }
      if (vec_out) {
        if (a.accumulate) {
          const float4 o = *reinterpret_cast<const float4*>(cp);
          v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
        }
        *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < a.N) cp[r] = a.accumulate ? cp[r] + v[r] : v[r];
      }
    }
  }
}

template <bool AVEC, bool BVEC, int MODE>
__global__ __launch_bounds__(256) void hgemm_kernel(HGemmArgs a) {
  hgemm_tile<AVEC, BVEC, MODE>(a, blockIdx.x, blockIdx.y);
}

// Backward of y = x W^T + b in one launch: tiles [0, nx) compute dX = pre(dy) . W, tiles [nx, nx + nw) compute
// dW (+)= pre(dy)^T . x and, in their first tile column, db (+)= colsum(pre(dy)).
template <bool DXVEC, int MODE>
__global__ __launch_bounds__(256) void hlinear_bwd_kernel(HGemmArgs dx, HGemmArgs dw, int nx, int dx_tiles_n, int dw_tiles_n) {
  const int b = blockIdx.x;
  if (b < nx) hgemm_tile<DXVEC, false, MODE>(dx, b % dx_tiles_n, b / dx_tiles_n);
  else hgemm_tile<false, false, MODE>(dw, (b - nx) % dw_tiles_n, (b - nx) / dw_tiles_n);
}

// out[c] (+)= sum_r x[r*ld + c]   -- bias gradients that are not attached to a weight-gradient GEMM
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ld, int rows, int cols,
                                                     float* __restrict__ out, int accumulate) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < cols)
    for (int r = rl; r < rows; r += 4) s += x[(long)r * ld + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < cols) {
    s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    out[c] = accumulate ? out[c] + s : s;
  }
}

void fill_dropout(HGemmArgs& a, float p, uint64_t seed, uint64_t offset) {
  a.d_thresh = p > 0.f ? (uint32_t)std::min<double>(4294967295.0, (double)p * 4294967296.0) : 0u;
  a.d_scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  a.d_seed = seed;
  a.d_offset = offset;
}

template <int MODE>
void launch_generic(const HGemmArgs& a, bool avec, bool bvec, hipStream_t st) {
  dim3 grid(ceil_div(a.N, 64), ceil_div(a.M, 64));
  if (avec && bvec) hipLaunchKernelGGL((hgemm_kernel<true, true, MODE>), grid, dim3(256), 0, st, a);
  else if (avec) hipLaunchKernelGGL((hgemm_kernel<true, false, MODE>), grid, dim3(256), 0, st, a);
  else if (bvec) hipLaunchKernelGGL((hgemm_kernel<false, true, MODE>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((hgemm_kernel<false, false, MODE>), grid, dim3(256), 0, st, a);
}

}  // namespace

extern "C" int mvf_hgemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
                         const float* bias, const float* table, long tab_si, long tab_sn, int tab_div, int tab_mod,
                         int M, int N, int K, float alpha, int relu, int accumulate, hipStream_t st) {
  return mvf_hgemm_ex(A, sam, sak, B, sbk, sbn, C, ldc, bias, table, tab_si, tab_sn, tab_div, tab_mod, M, N, K, alpha, relu,
                      accumulate, nullptr, 0, 0.f, 0, 0, st);
}

// mvf_hgemm + fused dropout / residual epilogue: C = [resid +] dropout_p(act(...)) (mask index = m*ldc + n, the element
// index of a dense [M, ldc] output -- the same indexing mvf_dropout_add uses, so the backward pre-op matches)
extern "C" int mvf_hgemm_ex(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
                            const float* bias, const float* table, long tab_si, long tab_sn, int tab_div, int tab_mod,
                            int M, int N, int K, float alpha, int relu, int accumulate, const float* resid, long ldr,
                            float drop_p, uint64_t drop_seed, uint64_t drop_offset, hipStream_t st) {
  MVF_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0);
  MVF_CHECK_ARG(table == nullptr || (tab_div > 0 && tab_mod > 0));
  MVF_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && !((resid != nullptr || drop_p > 0.f) && accumulate));
  HGemmArgs a{};
  a.A = A; a.sam = sam; a.sak = sak; a.B = B; a.sbk = sbk; a.sbn = sbn; a.C = C; a.ldc = ldc; a.bias = bias;
  a.table = table; a.tab_si = tab_si; a.tab_sn = tab_sn; a.tab_div = tab_div > 0 ? tab_div : 1;
  a.tab_mod = tab_mod > 0 ? tab_mod : 1; a.M = M; a.N = N; a.K = K; a.relu = relu; a.accumulate = accumulate;
  a.alpha = alpha; a.resid = resid; a.ldr = ldr;
  fill_dropout(a, drop_p, drop_seed, drop_offset);
  a.post_drop = drop_p > 0.f;
  const bool avec = sak == 1 && sam % 4 == 0 && K % 16 == 0 && ((uintptr_t)A & 15) == 0;
  const bool bvec = sbk == 1 && sbn % 4 == 0 && K % 16 == 0 && ((uintptr_t)B & 15) == 0;
  launch_generic<0>(a, avec, bvec, st);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// Backward of y = x W^T + b (x [M,K], W [N,K], dy [M,N] dense rows of stride ldy) in ONE launch:
//   g  = pre(dy):  pre_mode 0 none | 1 ReLU mask (g = dy * [ymask > 0], ymask indexed like dy) |
//                  2 dropout mask (g = dy * keep / (1 - p), element index m*ldy + n, as in the forward)
//   dx = g . W (may be NULL);  dW (+)= g^T . x;  db (+)= colsum(g) (may be NULL)
extern "C" int mvf_hlinear_bwd(const float* dy, long ldy, int pre_mode, const float* ymask, float drop_p,
                               uint64_t drop_seed, uint64_t drop_offset, const float* x, long ldx, const float* W, long ldw,
                               float* dx, long lddx, float* dW, long lddw, float* db, int M, int N, int K,
                               int accumulate_params, hipStream_t st) {
  MVF_CHECK_ARG(dy && x && W && dW && M > 0 && N > 0 && K > 0 && pre_mode >= 0 && pre_mode <= 2);
  MVF_CHECK_ARG(pre_mode != 1 || ymask != nullptr);
  MVF_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f);
  HGemmArgs gx{}, gw{};
  // dX[m, k] = sum_n g[m, n] W[n, k]:   A = dy (rows m, k-index n), B(n, k) = W[n*ldw + k]
  gx.A = dy; gx.sam = ldy; gx.sak = 1; gx.B = W; gx.sbk = ldw; gx.sbn = 1; gx.C = dx; gx.ldc = lddx;
  gx.M = M; gx.N = K; gx.K = N; gx.alpha = 1.f; gx.tab_div = gx.tab_mod = 1;
  gx.a_mode = pre_mode; gx.amask = ymask;
  fill_dropout(gx, pre_mode == 2 ? drop_p : 0.f, drop_seed, drop_offset);
  // dW[n, k] = sum_m g[m, n] x[m, k]:   A(n, m) = dy[m*ldy + n], B(m, k) = x[m*ldx + k]
  gw.A = dy; gw.sam = 1; gw.sak = ldy; gw.B = x; gw.sbk = ldx; gw.sbn = 1; gw.C = dW; gw.ldc = lddw;
  gw.M = N; gw.N = K; gw.K = M; gw.alpha = 1.f; gw.tab_div = gw.tab_mod = 1; gw.accumulate = accumulate_params;
  gw.a_mode = pre_mode; gw.amask = ymask; gw.rowsum = db; gw.rowsum_acc = accumulate_params;
  fill_dropout(gw, pre_mode == 2 ? drop_p : 0.f, drop_seed, drop_offset);
  const int dxn = ceil_div(K, 64), dxm = ceil_div(M, 64), dwn = ceil_div(K, 64), dwm = ceil_div(N, 64);
  const int nx = dx != nullptr ? dxn * dxm : 0, nw = dwn * dwm;
  const bool dxvec = ldy % 4 == 0 && N % 16 == 0 && ((uintptr_t)dy & 15) == 0 &&
                     (pre_mode != 1 || ((uintptr_t)ymask & 15) == 0);
#define LAUNCH(VEC, MODE) \
  hipLaunchKernelGGL((hlinear_bwd_kernel<VEC, MODE>), dim3(nx + nw), dim3(256), 0, st, gx, gw, nx, dxn, dwn)
  if (pre_mode == 0) { if (dxvec) LAUNCH(true, 0); else LAUNCH(false, 0); }
  else if (pre_mode == 1) { if (dxvec) LAUNCH(true, 1); else LAUNCH(false, 1); }
  else { if (dxvec) LAUNCH(true, 2); else LAUNCH(false, 2); }
#undef LAUNCH
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_colsum(const float* x, long ld, int rows, int cols, float* out, int accumulate, hipStream_t st) {
  MVF_CHECK_ARG(x && out && rows > 0 && cols > 0);
  hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(cols, 64)), dim3(256), 0, st, x, ld, rows, cols, out, accumulate);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}
