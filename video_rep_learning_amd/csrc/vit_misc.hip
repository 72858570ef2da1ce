// HBM-bound helper kernels of the ViT forward: patch gather (+cast), CLS row init, LayerNorm.
// They stand in for the aten ops timm issues around its GEMMs (reference call site
// CARL_MVF/models/transformer.py:188 -> timm VisionTransformer.forward).
#include "common.h"
#include "mvf_hip_internal.h"

namespace {

// frames [F,3,H,W] fp32 -> patch rows [F*gh*gw, ldk] (T), k = c*P*P + ky*P + kx (Conv2d weight order); columns
// 3*P*P .. ldk-1 are padding (ldk = 3*P*P rounded up to the GEMM's K granule; zero-filled by the caller's memset).
// VEC: one thread moves 4 consecutive kx (16 B read, coalesced along kx then px) -- needs P % 4 == 0; otherwise (patch
// 14 of the DINOv2 family) one element per thread.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, T* __restrict__ out, int F, int H,
                                                     int W, int P, int ldk) {
  const int gh = H / P, gw = W / P;
  constexpr int E = VEC ? 4 : 1;
  const size_t total = (size_t)F * 3 * H * (W / E);
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    // idx walks the image in memory order (f, c, y, x/E) -> fully coalesced reads
    const int xq = (int)(idx % (W / E));
    size_t r = idx / (W / E);
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % 3);
    const int f = (int)(r / 3);
    const int px = (xq * E) / P, kx = (xq * E) % P;
    const int py = y / P, ky = y % P;
    const size_t row = ((size_t)f * gh + py) * gw + px;
    T* dst = out + row * ldk + (size_t)c * P * P + ky * P + kx;
    if constexpr (VEC) {
      const float4 v = *reinterpret_cast<const float4*>(img + idx * 4);
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(dst) = v;
      } else {
        constexpr bool F16 = sizeof(T) == 2 && !__is_same(T, bf16_t);     // f16_t: IEEE half (MVF_F16)
        *reinterpret_cast<uint2*>(dst) = make_uint2(pack16x2<F16>(v.x, v.y), pack16x2<F16>(v.z, v.w));
      }
    } else {
      Elem<T>::st(dst, img[idx]);
    }
  }
}

// x[f, 0, :] = cls_token + pos_embed[0]
__global__ void cls_row_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos,
                               int F, int tpf, int D) {
  const int f = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += blockDim.x) x[(size_t)f * tpf * D + d] = cls[d] + pos[d];
}

// LayerNorm over the last dim, one wave per row, row cached in registers (D <= 64*4*MAXV).
// in: fp32 rows with stride `in_stride` rows apart (lets the final norm touch only CLS rows).
// add != NULL: the row normalised is x + add (bf16 [rows, add_stride]: the attention branch's output that the proj GEMM stored
// instead of read-modifying the fp32 residual; x itself is not updated here -- the fc2 epilogue adds the same values)
template <typename TO, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, size_t in_stride,
                                                        const float* __restrict__ g, const float* __restrict__ b,
                                                        TO* __restrict__ y, size_t out_stride, int rows, int D, float eps,
                                                        const bf16_t* __restrict__ add, size_t add_stride) {
  constexpr bool kF16 = sizeof(TO) == 2 && !__is_same(TO, bf16_t);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * in_stride;
  float4 v[MAXV];
  const int nv = D / 4;  // float4 per row
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int q = lane + i * 64;
    if (q < nv) {
      v[i] = *reinterpret_cast<const float4*>(xr + q * 4);
      if (add != nullptr) {
        const uint2 u = *reinterpret_cast<const uint2*>(add + (size_t)row * add_stride + q * 4);
        float a0, a1, a2, a3;          // (the addend is in the output's 16-bit format: bf16, or fp16 for TO = f16_t)
        unpack16x2<kF16>(u.x, a0, a1);
        unpack16x2<kF16>(u.y, a2, a3);
        v[i].x += a0; v[i].y += a1; v[i].z += a2; v[i].w += a3;
      }
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
  }
  const float mean = wave_sum(s) / D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int q = lane + i * 64;
    if (q < nv) {
      const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      ss += a * a + bb * bb + c * c + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / D + eps);
  TO* yr = y + (size_t)row * out_stride;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int q = lane + i * 64;
    if (q < nv) {
      const float4 gg = *reinterpret_cast<const float4*>(g + q * 4);
      const float4 be = *reinterpret_cast<const float4*>(b + q * 4);
      float o[4] = {(v[i].x - mean) * rstd * gg.x + be.x, (v[i].y - mean) * rstd * gg.y + be.y,
                    (v[i].z - mean) * rstd * gg.z + be.z, (v[i].w - mean) * rstd * gg.w + be.w};
      if constexpr (sizeof(TO) == 4) {
        *reinterpret_cast<float4*>(yr + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        *reinterpret_cast<uint2*>(yr + q * 4) = make_uint2(pack16x2<kF16>(o[0], o[1]), pack16x2<kF16>(o[2], o[3]));
      }
    }
  }
}

// LN fold: per-row partial (sum, sum of squares) written by the residual GEMM epilogues, slice-major [ns][rows][2] ->
// (mean, rstd) [rows][2].  One thread per row (coalesced over rows for every slice); fixed summation order.  Variance as E[x^2] - mean^2 in fp32: its relative error is
// 2^-24 * (1 + mean^2 / var), harmless for LayerNorm inputs (|mean| / std of a ViT residual row is O(1)).
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ part, int ns,
                                                                float* __restrict__ mr, int rows, float inv_d, float eps) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const float2* p = reinterpret_cast<const float2*>(part) + row;
  float s1 = 0.f, s2 = 0.f;
  for (int i = 0; i < ns; ++i) {
    const float2 v = p[(size_t)i * rows];
    s1 += v.x;
    s2 += v.y;
  }
  const float mean = s1 * inv_d;
  const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * inv_d), 0.f);   // (spelled as in gemm_tc256's in-kernel finalize)
  reinterpret_cast<float2*>(mr)[row] = make_float2(mean, 1.0f / sqrtf(var + eps));
}

// fp32 -> bf16 cast (weights pre-pack)
__global__ void cast_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    reinterpret_cast<uint2*>(out)[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
  }
}

// bf16 -> fp32 (attention output of a trainable block handed to the fp32 side of the graph)
__global__ void cast_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const uint2 v = reinterpret_cast<const uint2*>(in)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                                                    __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
  }
}

// diagnostic: which XCD (HW_REG_XCC_ID) and CU each workgroup of a 1-D launch lands on
__global__ void xcc_map_kernel(int* __restrict__ out) {
  if (threadIdx.x == 0) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    out[2 * blockIdx.x] = (int)(xcc & 0xf);
    out[2 * blockIdx.x + 1] = (int)hwid;
  }
  // stay resident for a moment so that consecutive workgroups spread over CUs like a real kernel's do
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < 20000ull) {}
}

}  // namespace

extern "C" int mvf_debug_xcc_map(int* out, int nblocks, int threads, int lds_bytes, hipStream_t st) {
  MVF_CHECK_ARG(out && nblocks > 0 && threads > 0 && threads <= 1024 && lds_bytes >= 0 && lds_bytes <= 160 * 1024);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xcc_map_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(xcc_map_kernel, dim3(nblocks), dim3(threads), lds_bytes, st, out);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

int mvf_im2col_impl(int dtype, const float* img, void* out, int F, int H, int W, int P, int ldk, hipStream_t st) {
  MVF_CHECK_ARG(img && out && F > 0 && H % P == 0 && W % P == 0 && ldk >= 3 * P * P);
  const bool vec = P % 4 == 0 && W % 4 == 0 && ldk % 4 == 0;
  if (ldk > 3 * P * P) {   // zero the padding columns (whole buffer: it is small and the memset is asynchronous)
    const size_t bytes = (size_t)F * (H / P) * (W / P) * ldk * (dtype == MVF_F32 ? 4 : 2);
    if (hipMemsetAsync(out, 0, bytes, st) != hipSuccess) return MVF_ERR_ARG;
  }
  const size_t total = (size_t)F * 3 * H * (vec ? W / 4 : W);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 256 * 16);
#define IM2COL(TT, V) hipLaunchKernelGGL((im2col_kernel<TT, V>), dim3(grid), dim3(256), 0, st, img, (TT*)out, F, H, W, P, ldk)
  if (dtype == MVF_BF16) { if (vec) IM2COL(bf16_t, true); else IM2COL(bf16_t, false); }
  else if (dtype == MVF_F16) { if (vec) IM2COL(f16_t, true); else IM2COL(f16_t, false); }
  else { if (vec) IM2COL(float, true); else IM2COL(float, false); }
#undef IM2COL
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

int mvf_cls_row_impl(float* x, const float* cls, const float* pos, int F, int tpf, int D, hipStream_t st) {
  MVF_CHECK_ARG(x && cls && pos && F > 0);
  hipLaunchKernelGGL(cls_row_kernel, dim3(F), dim3(256), 0, st, x, cls, pos, F, tpf, D);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

int mvf_layernorm_impl(int out_dtype, const float* x, size_t in_stride, const float* g, const float* b, void* y,
                       size_t out_stride, int rows, int D, float eps, hipStream_t st, const void* add_bf16, size_t add_stride) {
  MVF_CHECK_ARG(x && g && b && y && rows > 0 && D % 4 == 0 && D <= 64 * 4 * 8);
  MVF_CHECK_ARG(in_stride % 4 == 0 && out_stride % 4 == 0 && add_stride % 4 == 0 && ((uintptr_t)add_bf16 % 8) == 0);
  const bf16_t* add = (const bf16_t*)add_bf16;
  const int grid = ceil_div(rows, 4);
  const int nv = ceil_div(D / 4, 64);
#define LN_LAUNCH(TO, MV) \
  hipLaunchKernelGGL((layernorm_kernel<TO, MV>), dim3(grid), dim3(256), 0, st, x, in_stride, g, b, (TO*)y, out_stride, rows, D, eps, add, add_stride)
  if (out_dtype == MVF_BF16) {
    if (nv <= 1) LN_LAUNCH(bf16_t, 1); else if (nv <= 2) LN_LAUNCH(bf16_t, 2); else if (nv <= 4) LN_LAUNCH(bf16_t, 4); else LN_LAUNCH(bf16_t, 8);
  } else if (out_dtype == MVF_F16) {
    if (nv <= 1) LN_LAUNCH(f16_t, 1); else if (nv <= 2) LN_LAUNCH(f16_t, 2); else if (nv <= 4) LN_LAUNCH(f16_t, 4); else LN_LAUNCH(f16_t, 8);
  } else {
    if (nv <= 1) LN_LAUNCH(float, 1); else if (nv <= 2) LN_LAUNCH(float, 2); else if (nv <= 4) LN_LAUNCH(float, 4); else LN_LAUNCH(float, 8);
  }
#undef LN_LAUNCH
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

__global__ void cast_f16_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    reinterpret_cast<uint2*>(out)[i] = make_uint2(pack_f16x2(v.x, v.y), pack_f16x2(v.z, v.w));
  }
}

// fp32 -> IEEE fp16, round to nearest even (weights of the fp16 compute mode); n % 4 == 0
extern "C" int mvf_cast_f32_f16(const float* in, void* out, size_t n, hipStream_t st) {
  MVF_CHECK_ARG(in && out && n % 4 == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 8) == 0);
  const size_t n4 = n / 4;
  if (n4 == 0) return MVF_OK;
  const int grid = (int)std::min<size_t>((n4 + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(cast_f16_kernel, dim3(grid), dim3(256), 0, st, in, (uint16_t*)out, n4);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

int mvf_cast_bf16_impl(const float* in, void* out, size_t n, hipStream_t st) {
  MVF_CHECK_ARG(in && out && n % 4 == 0);
  const size_t n4 = n / 4;
  if (n4 == 0) return MVF_OK;
  const int grid = (int)std::min<size_t>((n4 + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid), dim3(256), 0, st, in, (bf16_t*)out, n4);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

int mvf_ln_stats_finalize_impl(const float* part, int ns, float* mr, int rows, int D, float eps, hipStream_t st) {
  MVF_CHECK_ARG(part && mr && rows > 0 && ns > 0 && D > 0);
  hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, part, ns, mr, rows, 1.0f / (float)D,
                     eps);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// LayerNorm(x + add) with a bf16 addend (the LN2 step of the deferred residual); x is not modified
extern "C" int mvf_layernorm_add_fwd(int out_dtype, const float* x, size_t in_stride, const void* add_bf16, size_t add_stride,
                                     const float* g, const float* b, void* y, size_t out_stride, int rows, int D, float eps,
                                     hipStream_t st) {
  MVF_CHECK_ARG(add_bf16 != nullptr && (out_dtype == MVF_F32 || out_dtype == MVF_BF16 || out_dtype == MVF_F16));   // (F16: fp16 addend)
  return mvf_layernorm_impl(out_dtype, x, in_stride, g, b, y, out_stride, rows, D, eps, st, add_bf16, add_stride);
}

extern "C" int mvf_cast_bf16_f32(const void* in, float* out, size_t n, hipStream_t st) {
  MVF_CHECK_ARG(in && out && n > 0 && n % 4 == 0 && ((uintptr_t)in % 8) == 0 && ((uintptr_t)out % 16) == 0);
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(cast_f32_kernel, dim3((unsigned)std::min<size_t>(4096, (n4 + 255) / 256)), dim3(256), 0, st,
                     (const bf16_t*)in, out, n4);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}
