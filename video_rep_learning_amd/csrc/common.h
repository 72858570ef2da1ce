// Shared device/host helpers for the MV-Former gfx950 kernels.
// Wavefront = 64 lanes everywhere (CDNA4); no portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MVF_OK 0
#define MVF_ERR_ARG 10001      // bad shape / pointer / alignment handed to an entry point
#define MVF_ERR_UNSUPPORTED 10002

#define MVF_CHECK_ARG(cond)        \
  do {                             \
    if (!(cond)) return MVF_ERR_ARG; \
  } while (0)

#define MVF_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even, NaN stays NaN: the plain cast compiles to v_cvt_pk_bf16_f32 on gfx950
// (one instruction per TWO elements; the integer bit-trick costs ~6 VALU per element and mangles some NaNs).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_native_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const bf16x2_native_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

// ---- fp16 operands (MI355X.COMPUTE_DTYPE fp16: the reference's own autocast dtype, CARL_MVF/train.py:113,301) ----
// Raw IEEE half bits travel as uint16_t like bf16 does; which of the two a buffer holds is a template parameter (`F16`) of the
// kernels that touch it.  Conversions round to nearest even (v_cvt_f16_f32 / v_cvt_pk_f16_f32; NOT the round-toward-zero pkrtz).
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_native_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_native_t;
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  const f16x2_native_t v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float f16_to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ uint16_t f32_to_f16(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
// two fp32 -> one dword of the 16-bit format, and back
template <bool F16>
__device__ __forceinline__ uint32_t pack16x2(float lo, float hi) {
  if constexpr (F16) return pack_f16x2(lo, hi);
  else return pack_bf16x2(lo, hi);
}
template <bool F16>
__device__ __forceinline__ void unpack16x2(uint32_t u, float& lo, float& hi) {
  if constexpr (F16) {
    const f16x2_native_t v = __builtin_bit_cast(f16x2_native_t, u);
    lo = (float)v[0]; hi = (float)v[1];
  } else {
    lo = __uint_as_float(u << 16); hi = __uint_as_float(u & 0xffff0000u);
  }
}
// D = A . B + C on the 16x16x32 matrix instruction of the format (same operand lane maps, same cycles)
template <bool F16>
__device__ __forceinline__ f32x4_t mfma16x16x32(const bf16x8_t& a, const bf16x8_t& b, const f32x4_t& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_native_t, a), __builtin_bit_cast(f16x8_native_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
struct f16_t { uint16_t bits; };   // distinct element type for the kernels templated on their 16-bit / fp32 element (Elem<T>)

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

template <>
struct Elem<f16_t> {
  static __device__ __forceinline__ float ld(const f16_t* p) { return f16_to_f32(p->bits); }
  static __device__ __forceinline__ void st(f16_t* p, float v) { p->bits = f32_to_f16(v); }
};

// ---- 64-lane wave reductions (butterfly over DPP/ds_swizzle via __shfl_xor) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// counter-based dropout mask shared by dropout_add_kernel and the fused GEMM pre-/post-ops:
// keep(i) = hash(seed, offset + i) >= p * 2^32  (splitmix64 finaliser)
__device__ __forceinline__ uint32_t mix32(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 16);
}
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t offset, uint64_t i, uint32_t thresh) {
  return mix32(seed * 0x2545F4914F6CDD1Dull + offset + i) >= thresh;
}

// ---- two-stage reductions in ONE launch: the workgroup that arrives last finishes the job ----
// Every workgroup stores its partial result, then calls last_arriver() (all threads); it returns true -- in every thread -- in exactly
// one workgroup of the grid, the one whose arrival was the n-th, and that workgroup may then read all n partials with plain loads.
// Nobody waits for anybody (no co-residency assumption).  Agent-scope release on the way in, acquire in the last arriver (per-XCD L2s
// are not coherent and a CU's L1 is never refreshed by other CUs' stores: /opt/skills/guides/MI355X_MICROARCH.md, "Valid forms"); the
// explicit s_waitcnt keeps the ticket from overtaking the write-back.  `ticket` is a zero-initialised device word that the last
// arriver resets; launches that could be in flight together (different streams) must not share one: the callers take theirs from a ring
// of TICKET_SLOTS words per kernel family (mvf_hip_internal.h TicketRing).
__device__ __forceinline__ bool last_arriver(unsigned* ticket, unsigned n) {
  __shared__ int s_last_arriver;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == n - 1u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_last_arriver = last;
  }
  __syncthreads();
  return s_last_arriver != 0;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
