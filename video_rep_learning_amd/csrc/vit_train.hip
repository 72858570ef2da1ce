// HBM-bound kernels between the GEMMs of a TRAINABLE backbone block in bf16 mode (ViTBackEnd, reference
// CARL_MVF/models/transformer.py:364-392; timm Block under fp16 autocast there).  The block keeps the residual stream in fp32
// and every GEMM operand in bf16, written once by its producer:
//   forward   LayerNorm -> bf16 (mvf_layernorm_fwd)   GEMM -> bf16 (qkv, fc1 pre-activation)   GELU bf16 -> bf16
//   backward  grad_prep: one pass over a gradient / activation matrix that emits what the two backward GEMMs of a linear
//             layer and its bias gradient need -- the row-major bf16 copy (operand of dX = dY W), the token-major transposed
//             chunks (operands of the split-K dW = dY^T X) and column-sum partials (db)
//             LayerNorm backward with recomputed statistics, the residual-stream gradient added in the same pass
// Each kernel moves every byte once; their roofline is HBM (bytes in the comments are per row of D = 768).
#include "common.h"
#include "mvf_hip_internal.h"

namespace {

constexpr float RSQRT2 = 0.70710678118654752440f;
constexpr float INV_SQRT_2PI = 0.39894228040143267794f;

__device__ __forceinline__ void unpack8(const uint4& u, float (&v)[8]) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}

// ---- exact-erf GELU on bf16 (timm Mlp act_layer = nn.GELU): g = u Phi(u);  du = dg (Phi(u) + u phi(u)).  n % 8 == 0.
__global__ __launch_bounds__(256) void gelu_fwd_bf16_kernel(const uint4* __restrict__ u, uint4* __restrict__ g, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float v[8];
    unpack8(u[i], v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.5f * v[j] * (1.0f + erff(v[j] * RSQRT2));
    g[i] = pack8(v);
  }
}
__global__ __launch_bounds__(256) void gelu_bwd_bf16_kernel(const uint4* __restrict__ dg, const uint4* __restrict__ u,
                                                            uint4* __restrict__ du, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float v[8], d[8];
    unpack8(u[i], v);
    unpack8(dg[i], d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float cdf = 0.5f * (1.0f + erff(v[j] * RSQRT2));
      const float pdf = INV_SQRT_2PI * __expf(-0.5f * v[j] * v[j]);
      d[j] *= cdf + v[j] * pdf;
    }
    du[i] = pack8(d);
  }
}

// ---- grad_prep: in [M, C] (fp32 or bf16, row stride C) ->
//   rm   [M, C] bf16                    row-major copy (fp32 input only; NULL = skip)
//   tr   [S, C, Mc] bf16                tr[s][c][j] = in[s Mc + j][c], 0 beyond M   (NULL = skip)
//   part [gridDim.x][C] fp32            column sums of this workgroup's RT x 64 rows (NULL = skip); mvf_sum_batches adds them
// Workgroup = 64 columns x RT tiles of 64 rows, one tile in LDS at a time.  Mc % (64 RT) == 0, C % 4 == 0.
constexpr int PREP_RT = 4;
template <typename TIN>
__global__ __launch_bounds__(256) void grad_prep_kernel(const TIN* __restrict__ in, bf16_t* __restrict__ rm,
                                                        bf16_t* __restrict__ tr, float* __restrict__ part, int M, int C, int Mc) {
  __shared__ float tile[64][65];
  __shared__ float red[4][64];
  const int c0 = blockIdx.y * 64;
  const int lr = threadIdx.x >> 4, lc = (threadIdx.x & 15) * 4;       // load role: 16 rows x 16 four-column groups per pass
  const int sc = threadIdx.x & 63, sq = threadIdx.x >> 6;              // column-sum role: column sc, rows 16 sq .. + 16
  const int jg = (threadIdx.x & 7) * 8;                                // store role: 8 consecutive tokens of channel cc
  float csum = 0.f;
  for (int t = 0; t < PREP_RT; ++t) {
    const int m0 = (blockIdx.x * PREP_RT + t) * 64;
    if (m0 >= M && tr == nullptr) break;                               // (the transposed chunks are zero-padded to Mc)
    if (t > 0) __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = p * 16 + lr, m = m0 + r, c = c0 + lc;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (m < M && c < C) {
        if constexpr (sizeof(TIN) == 4) {
          const float4 f = *reinterpret_cast<const float4*>(in + (size_t)m * C + c);
          v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
          if (rm != nullptr)
            *reinterpret_cast<uint2*>(rm + (size_t)m * C + c) = make_uint2(pack_bf16x2(f.x, f.y), pack_bf16x2(f.z, f.w));
        } else {
          const uint2 u = *reinterpret_cast<const uint2*>(in + (size_t)m * C + c);
          v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
          v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) tile[r][lc + i] = v[i];
    }
    __syncthreads();
    if (part != nullptr) {
#pragma unroll
      for (int r = 0; r < 16; ++r) csum += tile[sq * 16 + r][sc];
    }
    if (tr != nullptr) {
      const int s = m0 / Mc, j0 = m0 - s * Mc;                         // Mc % 64 == 0: a tile never straddles two chunks
      for (int cc = threadIdx.x >> 3; cc < 64; cc += 32) {
        const int c = c0 + cc;
        if (c >= C) continue;
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = pack_bf16x2(tile[jg + 2 * q][cc], tile[jg + 2 * q + 1][cc]);
        *reinterpret_cast<uint4*>(tr + ((size_t)s * C + c) * Mc + j0 + jg) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
  }
  if (part != nullptr) {
    red[sq][sc] = csum;
    __syncthreads();
    if (sq == 0 && c0 + sc < C)
      part[(size_t)blockIdx.x * C + c0 + sc] = red[0][sc] + red[1][sc] + red[2][sc] + red[3][sc];
  }
}

// ---- LayerNorm backward of a trainable block, one wave per row, the row held in registers (NV float4 per lane):
//   (mean, rstd) recomputed from x exactly as mvf_layernorm_fwd does (two-pass, biased variance)
//   dx = dres + rstd (dh g - mean_D(dh g) - xhat mean_D(dh g xhat))          (dres: the residual stream's gradient; may be NULL)
//   dxb = bf16(dx) (optional: operand of the next backward GEMM),  dg += sum_rows dh xhat,  db += sum_rows dh
// dg / db: per-workgroup partial sums over its 4 x RW rows, one float atomic per column and workgroup.
// Bytes per row: read x, dh, dres (3 x 4 D), write dx (4 D) [+ 2 D]: 12.3 KB at D = 768.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_rc_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                        const float* __restrict__ g, const float* __restrict__ dres,
                                                        float* __restrict__ dx, bf16_t* __restrict__ dxb,
                                                        float* __restrict__ dg, float* __restrict__ db, int rows, int D,
                                                        float eps, int RW) {
  extern __shared__ float red[];   // [4 waves][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = D / 4;
  float4 gg[NV], pg[NV], pb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = lane + i * 64;
    gg[i] = e < nv ? *reinterpret_cast<const float4*>(g + e * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    pg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    pb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float invD = 1.0f / D;
  for (int rr = 0; rr < RW; ++rr) {
    const int row = (blockIdx.x * 4 + wave) * RW + rr;
    if (row >= rows) break;           // whole wave
    const size_t o = (size_t)row * D;
    float4 xv[NV], dv[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = lane + i * 64;
      if (e < nv) {
        xv[i] = *reinterpret_cast<const float4*>(x + o + e * 4);
        dv[i] = *reinterpret_cast<const float4*>(dh + o + e * 4);
        s += xv[i].x + xv[i].y + xv[i].z + xv[i].w;
      }
    }
    const float mean = wave_sum(s) * invD;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = lane + i * 64;
      if (e < nv) {
        xv[i].x -= mean; xv[i].y -= mean; xv[i].z -= mean; xv[i].w -= mean;
        ss += xv[i].x * xv[i].x + xv[i].y * xv[i].y + xv[i].z * xv[i].z + xv[i].w * xv[i].w;
      }
    }
    const float rs = rsqrtf(wave_sum(ss) * invD + eps);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = lane + i * 64;
      if (e < nv) {
        xv[i].x *= rs; xv[i].y *= rs; xv[i].z *= rs; xv[i].w *= rs;                    // xhat
        pg[i].x += dv[i].x * xv[i].x; pg[i].y += dv[i].y * xv[i].y; pg[i].z += dv[i].z * xv[i].z; pg[i].w += dv[i].w * xv[i].w;
        pb[i].x += dv[i].x; pb[i].y += dv[i].y; pb[i].z += dv[i].z; pb[i].w += dv[i].w;
        dv[i].x *= gg[i].x; dv[i].y *= gg[i].y; dv[i].z *= gg[i].z; dv[i].w *= gg[i].w;  // dh g
        c1 += dv[i].x + dv[i].y + dv[i].z + dv[i].w;
        c2 += dv[i].x * xv[i].x + dv[i].y * xv[i].y + dv[i].z * xv[i].z + dv[i].w * xv[i].w;
      }
    }
    c1 = wave_sum(c1) * invD;
    c2 = wave_sum(c2) * invD;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = lane + i * 64;
      if (e < nv) {
        float4 r = make_float4(rs * (dv[i].x - c1 - xv[i].x * c2), rs * (dv[i].y - c1 - xv[i].y * c2),
                               rs * (dv[i].z - c1 - xv[i].z * c2), rs * (dv[i].w - c1 - xv[i].w * c2));
        if (dres != nullptr) {
          const float4 a = *reinterpret_cast<const float4*>(dres + o + e * 4);
          r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
        }
        *reinterpret_cast<float4*>(dx + o + e * 4) = r;
        if (dxb != nullptr) *reinterpret_cast<uint2*>(dxb + o + e * 4) = make_uint2(pack_bf16x2(r.x, r.y), pack_bf16x2(r.z, r.w));
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = lane + i * 64;
    if (e < nv) {
      *reinterpret_cast<float4*>(red + (wave * 2 + 0) * D + e * 4) = pg[i];
      *reinterpret_cast<float4*>(red + (wave * 2 + 1) * D + e * 4) = pb[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    atomicAdd(dg + c, red[0 * D + c] + red[2 * D + c] + red[4 * D + c] + red[6 * D + c]);
    atomicAdd(db + c, red[1 * D + c] + red[3 * D + c] + red[5 * D + c] + red[7 * D + c]);
  }
}

}  // namespace

extern "C" int mvf_gelu_bf16(const void* u, void* g, size_t n, hipStream_t st) {
  MVF_CHECK_ARG(u && g && n > 0 && n % 8 == 0 && ((uintptr_t)u % 16) == 0 && ((uintptr_t)g % 16) == 0);
  const size_t n8 = n / 8;
  const unsigned grid = (unsigned)((n8 + 255) / 256 < 65536 ? (n8 + 255) / 256 : 65536);
  hipLaunchKernelGGL(gelu_fwd_bf16_kernel, dim3(grid), dim3(256), 0, st, (const uint4*)u, (uint4*)g, n8);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_gelu_bwd_bf16(const void* dg, const void* u, void* du, size_t n, hipStream_t st) {
  MVF_CHECK_ARG(dg && u && du && n > 0 && n % 8 == 0 && ((uintptr_t)u % 16) == 0 && ((uintptr_t)dg % 16) == 0 &&
                ((uintptr_t)du % 16) == 0);
  const size_t n8 = n / 8;
  const unsigned grid = (unsigned)((n8 + 255) / 256 < 65536 ? (n8 + 255) / 256 : 65536);
  hipLaunchKernelGGL(gelu_bwd_bf16_kernel, dim3(grid), dim3(256), 0, st, (const uint4*)dg, (const uint4*)u, (uint4*)du, n8);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_grad_prep(int in_dtype, const void* in, void* rowmajor_bf16, void* transposed_bf16, float* colsum_part,
                             int part_rows, int M, int C, int Mc, hipStream_t st) {
  MVF_CHECK_ARG(in && M > 0 && C > 0 && C % 4 == 0 && (in_dtype == MVF_F32 || in_dtype == MVF_BF16));
  MVF_CHECK_ARG(rowmajor_bf16 || transposed_bf16 || colsum_part);
  MVF_CHECK_ARG(!(rowmajor_bf16 && in_dtype == MVF_BF16));
  MVF_CHECK_ARG(!transposed_bf16 || (Mc > 0 && Mc % (64 * PREP_RT) == 0));
  MVF_CHECK_ARG(((uintptr_t)in % 16) == 0 && ((uintptr_t)rowmajor_bf16 % 8) == 0 && ((uintptr_t)transposed_bf16 % 16) == 0);
  // rows covered: the transposed chunks are padded to S * Mc tokens, the other outputs stop at M
  const int rows = transposed_bf16 ? ceil_div(M, Mc) * Mc : M;
  const dim3 grid(ceil_div(rows, 64 * PREP_RT), ceil_div(C, 64));
  MVF_CHECK_ARG(!colsum_part || part_rows == (int)grid.x);     // one partial row per 256 input rows
  if (in_dtype == MVF_F32)
    hipLaunchKernelGGL(grad_prep_kernel<float>, grid, dim3(256), 0, st, (const float*)in, (bf16_t*)rowmajor_bf16,
                       (bf16_t*)transposed_bf16, colsum_part, M, C, Mc);
  else
    hipLaunchKernelGGL(grad_prep_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)rowmajor_bf16,
                       (bf16_t*)transposed_bf16, colsum_part, M, C, Mc);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_ln_bwd_block(const float* dh, const float* x, const float* g, const float* dres, float* dx, void* dx_bf16,
                                float* dg, float* db, int rows, int D, float eps, hipStream_t st) {
  MVF_CHECK_ARG(dh && x && g && dx && dg && db && rows > 0 && D > 0 && D % 4 == 0 && D <= 1536);
  MVF_CHECK_ARG(((uintptr_t)dh % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)dx % 16) == 0 &&
                ((uintptr_t)dres % 16) == 0 && ((uintptr_t)dx_bf16 % 8) == 0);
  constexpr int RW = 8;            // rows per wave: 32 rows per workgroup -> rows / 32 atomic adders per column
  const dim3 grid(ceil_div(rows, 4 * RW));
  const size_t lds = (size_t)8 * D * 4;
  if (D <= 768)
    hipLaunchKernelGGL(ln_bwd_rc_kernel<3>, grid, dim3(256), lds, st, dh, x, g, dres, dx, (bf16_t*)dx_bf16, dg, db, rows, D, eps, RW);
  else if (D <= 1024)
    hipLaunchKernelGGL(ln_bwd_rc_kernel<4>, grid, dim3(256), lds, st, dh, x, g, dres, dx, (bf16_t*)dx_bf16, dg, db, rows, D, eps, RW);
  else
    hipLaunchKernelGGL(ln_bwd_rc_kernel<6>, grid, dim3(256), lds, st, dh, x, g, dres, dx, (bf16_t*)dx_bf16, dg, db, rows, D, eps, RW);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}
