// fp32 MFMA GEMM with generic strides for the (small, latency-bound) trainable head:
//     C[m, n] (+)= act( alpha * sum_k A(m,k) * B(k,n) + bias[n] + table[idx(m), n] )
// A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]: one kernel serves forward (x . W^T), input
// gradient (dy . W) and weight gradient (dy^T . x) of every nn.Linear on the path
// (CARL_MVF/models/mvformer.py:77,86,97; models/utils.py:65-68,182-183; resnet_c2d.py:117-120) by
// passing different strides -- no transposed copies.
//
// gfx950 design: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).  The matrices here are <= a few MB and
// L2-resident, M is 768 rows, so the kernel is latency- not bandwidth-bound: fragments are loaded straight
// from global/L2 into the MFMA operand registers (float4 along k when the operand is k-contiguous, using
// the k-permutation trick: element s of lane (r, g) is k = k0 + 4g + s for BOTH operands), no LDS, no
// barriers.  64x64 tile per workgroup (4 waves x 32x32); operands swapped so a lane owns 4 consecutive n.
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
};

template <bool VEC>
__device__ __forceinline__ f32x4_t load_frag(const float* base, long row_off, long sk, int k0, int g, int K) {
  // returns {X(row, k0+4g+0..3)}; zero beyond K
  f32x4_t v;
  const int k = k0 + 4 * g;
  if constexpr (VEC) {
    v = *reinterpret_cast<const f32x4_t*>(base + row_off + k);
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = (k + s < K) ? base[row_off + (long)(k + s) * sk] : 0.f;
  }
  return v;
}

template <bool AVEC, bool BVEC>
__global__ __launch_bounds__(256) void hgemm_kernel(HGemmArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int m_base = blockIdx.y * 64 + (wave >> 1) * 32;
  const int n_base = blockIdx.x * 64 + (wave & 1) * 32;
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

  for (int k0 = 0; k0 < a.K; k0 += 16) {
    f32x4_t af[2], bf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[i] = load_frag<AVEC>(a.A, aoff[i], a.sak, k0, g, a.K);
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[j] = load_frag<BVEC>(a.B, boff[j], a.sbk, k0, g, a.K);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);
  }

  const bool vec_out = (a.ldc % 4 == 0) && (a.N % 4 == 0) && (((uintptr_t)a.C & 15) == 0);
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

// out[c] (+)= sum_r x[r*ld + c]   -- bias gradients
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

}  // namespace

extern "C" int mvf_hgemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
                         const float* bias, const float* table, long tab_si, long tab_sn, int tab_div, int tab_mod,
                         int M, int N, int K, float alpha, int relu, int accumulate, hipStream_t st) {
  MVF_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0);
  MVF_CHECK_ARG(table == nullptr || (tab_div > 0 && tab_mod > 0));
  HGemmArgs a;
  a.A = A; a.sam = sam; a.sak = sak; a.B = B; a.sbk = sbk; a.sbn = sbn; a.C = C; a.ldc = ldc; a.bias = bias;
  a.table = table; a.tab_si = tab_si; a.tab_sn = tab_sn; a.tab_div = tab_div > 0 ? tab_div : 1;
  a.tab_mod = tab_mod > 0 ? tab_mod : 1; a.M = M; a.N = N; a.K = K; a.relu = relu; a.accumulate = accumulate;
  a.alpha = alpha;
  const bool avec = sak == 1 && sam % 4 == 0 && K % 16 == 0 && ((uintptr_t)A & 15) == 0;
  const bool bvec = sbk == 1 && sbn % 4 == 0 && K % 16 == 0 && ((uintptr_t)B & 15) == 0;
  dim3 grid(ceil_div(N, 64), ceil_div(M, 64));
  if (avec && bvec) hipLaunchKernelGGL((hgemm_kernel<true, true>), grid, dim3(256), 0, st, a);
  else if (avec) hipLaunchKernelGGL((hgemm_kernel<true, false>), grid, dim3(256), 0, st, a);
  else if (bvec) hipLaunchKernelGGL((hgemm_kernel<false, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((hgemm_kernel<false, false>), grid, dim3(256), 0, st, a);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_colsum(const float* x, long ld, int rows, int cols, float* out, int accumulate, hipStream_t st) {
  MVF_CHECK_ARG(x && out && rows > 0 && cols > 0);
  hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(cols, 64)), dim3(256), 0, st, x, ld, rows, cols, out, accumulate);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}
