// MX-fp8 device helpers shared by the quantising kernels (mxfp8.hip) and the quantising GEMM epilogue (gemm_tc256.hip).
#pragma once
#include "common.h"

// E8M0 byte of the smallest power-of-two scale s with amax / s <= 448 (e4m3's largest finite value):
// amax = m * 2^e, m in [1, 2)  ->  s = 2^(e - 8) if m <= 1.75 else 2^(e - 7)
__device__ __forceinline__ unsigned mx_scale_byte(float amax) {
  const unsigned b = __float_as_uint(amax);
  const int byte = (int)((b >> 23) & 0xffu) - 8 + ((b & 0x7fffffu) > 0x600000u ? 1 : 0);
  return (unsigned)min(max(byte, 0), 253);
}
__device__ __forceinline__ float mx_inv_scale(unsigned byte) { return __uint_as_float((254u - byte) << 23); }   // 2^(127 - byte)

// four floats -> four OCP e4m3 bytes (element 0 in the low byte), round to nearest even
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (unsigned)v;
}
