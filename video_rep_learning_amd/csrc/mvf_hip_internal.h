// Internal (non-exported) declarations shared between the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <atomic>
#include "../../include/mvf_hip.h"
#include "common.h"

// Raises a kernel's dynamic-LDS limit once per DEVICE (the attribute is per device: a process-wide "done" flag would leave a
// second device at the 64 KB default); a failure is reported so that the caller can decline (MVF_ERR_UNSUPPORTED) instead of
// launching into a generic error.  done_mask: one static uint64_t per kernel instantiation.
static inline int mvf_ensure_lds(const void* fn, size_t bytes, uint64_t& done_mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return MVF_ERR_UNSUPPORTED;
  if (dev < 64 && ((done_mask >> dev) & 1)) return MVF_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return MVF_ERR_UNSUPPORTED;
  if (dev < 64) done_mask |= 1ull << dev;
  return MVF_OK;
}

// Ticket slots for common.h last_arriver(): every launch of a kernel family takes the next of TICKET_SLOTS zero-initialised device
// words, so launches in flight on DIFFERENT streams never count on the same word (the last arriver resets its word; a collision would
// need 16 launches of one family in flight at once).
constexpr int TICKET_SLOTS = 16;
struct TicketRing {
  std::atomic<unsigned> next{0};
  int take() { return (int)(next.fetch_add(1u, std::memory_order_relaxed) % TICKET_SLOTS); }
};

// ---- backbone ----
// LayerNorm folded into the GEMMs (bf16 256x256 kernel only): producer side (epi 2) xb / stats, consumer side (epi 0, 1)
// ln_mr / ln_c -- see gemm_tc_epi.h GemmTcArgs
struct MvfGemmLn {
  void* xb;
  int ldxb;
  float* stats;
  const float* ln_mr;
  const float* ln_c;
  // epi 2: 0 = the addend is `resid` itself (in place), 1 = read it from `addend` [M, ldr], 2 = no addend
  int addend_mode;
  const float* addend;
  // epi 2: a second addend, bf16 [M, ld2] (NULL: none): the attention branch's output stored by the proj GEMM (deferred residual)
  const void* addend2;
  int ld2;
  // epi 0 / 1, instead of ln_mr: the producer's partial sums [ln_ns][M][2] (finalized inside the 256x256 kernel; D = K)
  const float* ln_part;
  int ln_ns;
  float ln_eps;
  // MX-fp8 producer (mvf_gemm_fp8_impl, epi 2): xb receives e4m3 bytes [M, ldxb] of the new residual row, these their block scales
  // [N/128][M] (mxfp8.hip's layout)
  unsigned* xb_scales;
};
int mvf_gemm_tc_impl(int dtype, int epi, const void* A, int lda, const void* W, int ldw, const float* bias, void* C,
                     int ldc, float* resid, int ldr, void* tap, int ldt, const float* pos, const float* ls, int tpf, int M,
                     int N, int K, hipStream_t st, int batch_rows = 0, int w_batch_rows = 0, const MvfGemmLn* ln = nullptr);
// MX-fp8 path (mxfp8.hip, gemm_tc.hip)
int mvf_quant_mxfp8_impl(int in_dtype, const void* x, size_t ldx, void* q, size_t ldq, unsigned* scales, int rows, int K,
                         hipStream_t st);
int mvf_layernorm_mxfp8_impl(const float* x, size_t in_stride, const float* g, const float* b, void* q, size_t ldq,
                             unsigned* scales, int rows, int D, float eps, hipStream_t st, const void* add_bf16 = nullptr, size_t add_stride = 0);
int mvf_gemm_fp8_impl(int epi, const void* A, int lda, const unsigned* sa, const void* W, int ldw, const unsigned* sw,
                      const float* bias, void* C, int ldc, unsigned* c_scales, float* resid, int ldr, void* tap, int ldt,
                      const float* ls, int tpf, int M, int N, int K, hipStream_t st, const void* addend2 = nullptr, int ld2 = 0,
                      const MvfGemmLn* ln = nullptr);
int mvf_ln_stats_finalize_impl(const float* part, int ns, float* mr, int rows, int D, float eps, hipStream_t st);
int mvf_im2col_impl(int dtype, const float* img, void* out, int F, int H, int W, int P, int ldk, hipStream_t st);
// K of the patch-embed GEMM: 3*P*P rounded up to 128 elements (the granule of both GEMM kernels and dtypes)
static inline int mvf_patch_k(int P) { return (3 * P * P + 127) / 128 * 128; }
int mvf_cls_row_impl(float* x, const float* cls, const float* pos, int F, int tpf, int D, hipStream_t st);
int mvf_layernorm_impl(int out_dtype, const float* x, size_t in_stride, const float* g, const float* b, void* y,
                       size_t out_stride, int rows, int D, float eps, hipStream_t st, const void* add_bf16 = nullptr, size_t add_stride = 0);
int mvf_cast_bf16_impl(const float* in, void* out, size_t n, hipStream_t st);
int mvf_vit_attn_impl(int dtype, const void* qkv, void* out, int F, int N, int H, int D, int variant, hipStream_t st);
// vit_attn32.hip: the streamed kernel on 32-query-row tiles (any N; bf16 / fp16); lse != NULL: also the per-query log-sum-exp (bf16)
// q8_scales != NULL: `out` receives MX-fp8 bytes [F*N, D] and q8_scales their block scales [D/128][F*N] (mxfp8.hip's layout) -- bit for bit
// what mvf_quant_mxfp8 makes of the bf16 output (H even; bf16 only)
int mvf_vit_attn32_impl(int dtype, const void* qkv, void* out, float* lse, int F, int N, int H, int D, int form, int nw_force,
                        hipStream_t st, unsigned* q8_scales = nullptr);
// vit_qkv_attn.hip: the qkv projection fused into the attention kernel (N = 193 .. 208); MVF_ERR_UNSUPPORTED outside its shapes
// (folded form: the rows' statistics as ln_mr pairs, or as the producer's partial sums ln_part [ln_ns][F*N][2] finalized in the kernel)
int mvf_qkv_attn_impl(int dtype, const void* A, int lda, const void* W, const float* bias, const float* ln_c, const float* ln_mr,
                      const float* ln_part, int ln_ns, float ln_eps, void* out, int F, int N, int H, int D, hipStream_t st);
// shapes the fused kernel takes (and MVF_FUSE_QKV != 0: the variable keeps the GEMM + attention launches for A/B measurements)
bool mvf_qkv_attn_supported(int dtype, int F, int N, int H, int D, int lda);

// ---- head ----
// MFMA temporal attention (head_attn_mfma.hip); which: 0 forward, 1 backward; MVF_ERR_UNSUPPORTED unless dk in {16,32,64}
int mvf_tattn_mfma(int which, const float* qkv, const float* mask, int mask_len, float* o, float* lse, const float* d_o, float* dqkv,
                   int B, int S, int H, int Dm, hipStream_t st);
