// MX-fp8 producers for the fp8 backbone path (MI355X.COMPUTE_DTYPE fp8; BASELINE.json configs[4]):
//   values: OCP fp8 e4m3 (gfx950 v_cvt_pk_fp8_f32, round to nearest even), one byte per element, row-major [rows, K]
//   scales: one E8M0 exponent byte per 32 consecutive k of a row (the block a lane of v_mfma_scale_f32_16x16x128_f8f6f4
//           multiplies), the four bytes of a 128-wide K tile packed in one dword: scales[K/128][rows] (block b of the tile in
//           byte b) -- the layout gemm_tc256's FP8 variants stage with one 4-byte-per-lane LDS-DMA per wave and K tile.
//   scale rule: the smallest power of two s with amax / s <= 448 (e4m3's largest finite value), so nothing saturates:
//           amax = m * 2^e, m in [1, 2)  ->  s = 2^(e - 8) if m <= 1.75 else 2^(e - 7).
// These stand in for the casts PyTorch autocast inserts in front of every nn.Linear of the timm ViT (reference call site
// CARL_MVF/models/transformer.py:188; the reference's autocast dtype is fp16 -- fp8 is this build's own, configs[4]).
#include "common.h"
#include "mvf_hip_internal.h"
#include "mxfp8.h"

namespace {

// ---- generic quantiser: [rows, K] bf16 / f32 -> fp8 + scales.  Block = 16 rows x one K tile (128 elements); thread
// (r = tid / 16, c = tid % 16) owns 8 consecutive elements, the four lanes 4q .. 4q+3 of a row share one 32-k block.
template <typename TIN>
__global__ __launch_bounds__(256) void quant_mxfp8_kernel(const TIN* __restrict__ x, size_t ldx, unsigned char* __restrict__ q,
                                                          size_t ldq, unsigned* __restrict__ scales, int rows) {
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  const int row = blockIdx.x * 16 + r;
  const int kt = blockIdx.y;
  const bool ok = row < rows;
  float v[8];
  if (ok) {
    const TIN* p = x + (size_t)row * ldx + (size_t)kt * 128 + c * 8;
    if constexpr (sizeof(TIN) == 2) {
      const uint4 u = *reinterpret_cast<const uint4*>(p);
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
      }
    } else {
      const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
  }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  const unsigned sb = mx_scale_byte(amax);
  const float inv = mx_inv_scale(sb);
  if (ok) {
    unsigned char* o = q + (size_t)row * ldq + (size_t)kt * 128 + c * 8;
    *reinterpret_cast<uint2*>(o) = make_uint2(pack_fp8x4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv),
                                              pack_fp8x4(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv));
  }
  // the row's four block scales (lanes c = 0, 4, 8, 12 of its 16 lanes) -> one dword, stored by lane c = 0
  const int base = threadIdx.x & 48;   // first lane of this row's 16 lanes inside the wave
  const unsigned d = __shfl(sb, base, 64) | (__shfl(sb, base + 4, 64) << 8) | (__shfl(sb, base + 8, 64) << 16) |
                     (__shfl(sb, base + 12, 64) << 24);
  if (ok && c == 0) scales[(size_t)kt * rows + row] = d;
}

// ---- LayerNorm with an MX-fp8 result: one wave per row, row cached in registers like layernorm_kernel (vit_misc.hip).
// Lane `lane` of chunk i holds elements 4 (lane + 64 i) .. +4: eight consecutive lanes share a 32-k block.
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_mxfp8_kernel(const float* __restrict__ x, size_t in_stride,
                                                              const float* __restrict__ g, const float* __restrict__ b,
                                                              unsigned char* __restrict__ q, size_t ldq,
                                                              unsigned* __restrict__ scales, int rows, int D, float eps,
                                                              const bf16_t* __restrict__ add, size_t add_stride) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;     // whole wave
  const float* xr = x + (size_t)row * in_stride;
  float4 v[MAXV];
  const int nv = D / 4;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = lane + i * 64;
    if (e < nv) {
      v[i] = *reinterpret_cast<const float4*>(xr + e * 4);
      if (add != nullptr) {     // the deferred attention-branch output (vit_fwd.hip): the row normalised is x + add
        const uint2 u = *reinterpret_cast<const uint2*>(add + (size_t)row * add_stride + e * 4);
        v[i].x += __uint_as_float(u.x << 16); v[i].y += __uint_as_float(u.x & 0xffff0000u);
        v[i].z += __uint_as_float(u.y << 16); v[i].w += __uint_as_float(u.y & 0xffff0000u);
      }
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
  }
  const float mean = wave_sum(s) / D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = lane + i * 64;
    if (e < nv) {
      const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      ss += a * a + bb * bb + c * c + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / D + eps);
  unsigned char* qr = q + (size_t)row * ldq;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = lane + i * 64;           // D % 256 == 0 (checked by the launcher): a chunk is whole or absent
    if (i * 64 < nv) {
      const float4 gg = *reinterpret_cast<const float4*>(g + e * 4);
      const float4 be = *reinterpret_cast<const float4*>(b + e * 4);
      const float o0 = (v[i].x - mean) * rstd * gg.x + be.x, o1 = (v[i].y - mean) * rstd * gg.y + be.y;
      const float o2 = (v[i].z - mean) * rstd * gg.z + be.z, o3 = (v[i].w - mean) * rstd * gg.w + be.w;
      float amax = fmaxf(fmaxf(fabsf(o0), fabsf(o1)), fmaxf(fabsf(o2), fabsf(o3)));
      amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
      amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
      amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
      const unsigned sb = mx_scale_byte(amax);
      const float inv = mx_inv_scale(sb);
      *reinterpret_cast<unsigned*>(qr + e * 4) = pack_fp8x4(o0 * inv, o1 * inv, o2 * inv, o3 * inv);
      // chunk i = 256 elements = K tiles 2i (lanes 0-31) and 2i+1 (lanes 32-63); block leaders are lanes 0, 8, 16, 24 (+32)
      const int half = lane & 32;
      const unsigned d = __shfl(sb, half, 64) | (__shfl(sb, half + 8, 64) << 8) | (__shfl(sb, half + 16, 64) << 16) |
                         (__shfl(sb, half + 24, 64) << 24);
      if ((lane & 31) == 0) scales[(size_t)(2 * i + (half >> 5)) * rows + row] = d;
    }
  }
}

}  // namespace

int mvf_quant_mxfp8_impl(int in_dtype, const void* x, size_t ldx, void* q, size_t ldq, unsigned* scales, int rows, int K,
                         hipStream_t st) {
  MVF_CHECK_ARG(x && q && scales && rows > 0 && K > 0 && K % 128 == 0);
  MVF_CHECK_ARG(in_dtype == MVF_BF16 || in_dtype == MVF_F32);
  MVF_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)q % 8) == 0 && ldq % 8 == 0 &&
                (ldx * (in_dtype == MVF_BF16 ? 2 : 4)) % 16 == 0);
  const dim3 grid((rows + 15) / 16, K / 128);
  if (in_dtype == MVF_BF16)
    hipLaunchKernelGGL(quant_mxfp8_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, ldx, (unsigned char*)q, ldq, scales,
                       rows);
  else
    hipLaunchKernelGGL(quant_mxfp8_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, (unsigned char*)q, ldq, scales,
                       rows);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

int mvf_layernorm_mxfp8_impl(const float* x, size_t in_stride, const float* g, const float* b, void* q, size_t ldq,
                             unsigned* scales, int rows, int D, float eps, hipStream_t st, const void* add_bf16, size_t add_stride) {
  MVF_CHECK_ARG(x && g && b && q && scales && rows > 0 && D > 0 && D % 256 == 0 && D <= 64 * 4 * 8 && ldq % 4 == 0);
  MVF_CHECK_ARG(add_stride % 4 == 0 && ((uintptr_t)add_bf16 % 8) == 0);
  const bf16_t* add = (const bf16_t*)add_bf16;
  const dim3 grid((rows + 3) / 4);
  unsigned char* qq = (unsigned char*)q;
  if (D <= 1024)
    hipLaunchKernelGGL(layernorm_mxfp8_kernel<4>, grid, dim3(256), 0, st, x, in_stride, g, b, qq, ldq, scales, rows, D, eps, add, add_stride);
  else
    hipLaunchKernelGGL(layernorm_mxfp8_kernel<8>, grid, dim3(256), 0, st, x, in_stride, g, b, qq, ldq, scales, rows, D, eps, add, add_stride);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_quant_mxfp8(int in_dtype, const void* x, size_t ldx, void* q, size_t ldq, unsigned* scales, int rows, int K,
                               hipStream_t st) {
  return mvf_quant_mxfp8_impl(in_dtype, x, ldx, q, ldq, scales, rows, K, st);
}
extern "C" int mvf_layernorm_mxfp8(const float* x, size_t in_stride, const float* g, const float* b, void* q, size_t ldq,
                                   unsigned* scales, int rows, int D, float eps, hipStream_t st) {
  return mvf_layernorm_mxfp8_impl(x, in_stride, g, b, q, ldq, scales, rows, D, eps, st, nullptr, 0);
}
