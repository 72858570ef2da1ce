/* mvf_hip.h -- C ABI of libmvf_hip.so: the MI355X (gfx950) kernels of the MV-Former SCL training step.
 *
 * The reference (facebookresearch/video_rep_learning, CARL_MVF/) is pure Python on PyTorch and has no
 * FFI of its own; its hot path reaches native code only through ATen.  This header is the boundary a
 * maintainer binds instead (ctypes stub: INTEGRATION.md).  Every entry point lists the reference code
 * whose arithmetic it replaces (paths relative to /root/reference/CARL_MVF).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no framework types; all pointers are DEVICE pointers unless named *_host
 *   - the caller owns every buffer (incl. workspaces); nothing is allocated, freed or synchronised inside
 *   - work is enqueued on `stream`; returns 0 (MVF_OK) or MVF_ERR_* (bad argument, checked on the host
 *     BEFORE any launch) or a hipError_t value; never throws
 *   - fp32 tensors are `float`; bf16 tensors are raw 16-bit words (`void*` with a dtype code)
 *   - row-major, innermost dimension contiguous unless a stride argument says otherwise
 */
#ifndef MVF_HIP_H_
#define MVF_HIP_H_

#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVF_F32 0
#define MVF_BF16 1
#define MVF_FP8 2   /* mvf_vit_fwd only: MX-fp8 GEMM operands (OCP e4m3 + E8M0 scale per 32 k), bf16 everywhere else */
#define MVF_F16 3   /* frozen backbone (mvf_vit_fwd, mvf_gemm_tc*, mvf_vit_attn_fwd, mvf_layernorm*_fwd, mvf_patchify): IEEE fp16 operands and
                     * activations -- the reference's own autocast dtype (CARL_MVF/train.py:113,301) -- on v_mfma_f32_16x16x32_f16, fp32
                     * accumulation and residual stream; the tapped features are still written as bf16.  K % 128 == 0, N % 32 == 0 (the
                     * 256x256 kernel only); trainable backbone blocks and MX-fp8 do not combine with it. */

/* gemm_tc epilogues */
#define MVF_EPI_STORE 0 /* C = A W^T + b                                  */
#define MVF_EPI_GELU 1  /* C = gelu_erf(A W^T + b)                        */
#define MVF_EPI_RESID 2 /* resid += ls * (A W^T + b) ; optional tap copy   */
#define MVF_EPI_PATCH 3 /* resid[f, 1+p] = A W^T + b + pos[1+p]            */

/* ------------------------------------------------------------------------------------------------
 * Frozen ViT backbone (forward only)
 *   replaces: timm VisionTransformer.forward as called by models/transformer.py:59,188,322-331 and the
 *   FeatureExtractor hook/concat + CLS-drop/movedim/reshape copies of transformer.py:199-214,306-333
 * ---------------------------------------------------------------------------------------------- */
typedef struct MvfVitWeights {
  int depth, dim, heads, patch, img, n_taps;
  int taps[8];                 /* block indices whose OUTPUT is tapped (SMART_FEATS), ascending */
  float ln_eps;                /* 1e-6 for timm ViT */
  const float* cls_token;      /* [dim]            */
  const float* pos_embed;      /* [1+P, dim]       */
  const void* patch_w;         /* [dim, Kp] dtype T, k = c*p*p + ky*p + kx, Kp = 3*p*p rounded up to a multiple of 128 with
                                  zero columns (768 for patch 16, 192 -> 256 for patch 8, 588 -> 640 for patch 14) */
  const float* patch_b;        /* [dim]            */
  const float* norm_w;         /* final norm [dim] */
  const float* norm_b;
  /* per-block arrays of HOST-resident pointer tables (each table has `depth` device pointers) */
  const float* const* ln1_w; const float* const* ln1_b;
  const void* const* qkv_w;  const float* const* qkv_b;   /* [3*dim, dim] T, [3*dim] */
  const void* const* proj_w; const float* const* proj_b;  /* [dim, dim]   T, [dim]   */
  const float* const* ln2_w; const float* const* ln2_b;
  const void* const* fc1_w;  const float* const* fc1_b;   /* [4*dim, dim] T */
  const void* const* fc2_w;  const float* const* fc2_b;   /* [dim, 4*dim] T */
  const float* const* ls1;   const float* const* ls2;     /* LayerScale gamma tables or NULL */
  /* LayerNorm folded into the GEMM that consumes it (bf16 mode, dim % 128 == 0; tables or NULL, entries may be NULL):
   * where qkv_c[l] != NULL, qkv_w[l] holds gamma1 (.) W_qkv (per input column), qkv_b[l] holds b_qkv + W_qkv beta1 and
   * qkv_c[l][n] = sum_k bf16(qkv_w[l][n,k]); the GEMM then reads the un-normalised bf16 residual stream and applies
   * rstd * (acc - mean * c[n]) + b[n] in its epilogue, and no LayerNorm kernel runs for norm1 of block l (timm
   * Block.forward: x + attn(norm1(x)); x + mlp(norm2(x))).  Same for fc1_c / norm2.  qkv_c[0] must be NULL (block 0's
   * norm1 follows the patch embedding). */
  const float* const* qkv_c; const float* const* fc1_c;
  /* dtype MVF_FP8: qkv_w / proj_w / fc1_w / fc2_w hold e4m3 bytes [N, K] and these tables their block scales [K/128][N]
   * dwords (mvf_quant_mxfp8's layout); patch_w stays bf16.  LN fold in fp8 mode: norm1 only -- where qkv_c[l] != NULL (l > 0), qkv_w[l]
   * holds MX-fp8(gamma1 (.) W_qkv), qkv_b[l] = b + W beta1, qkv_c[l][n] = sum_k of the DEQUANTISED qkv_w[l][n, k]; block l - 1's fc2
   * epilogue then leaves MX-fp8(x) and the rows' partial sums for it (mvf_gemm_fp8_ln) and no LayerNorm pass runs for norm1 of block l.
   * fc1_c must be NULL. */
  const unsigned* const* qkv_s; const unsigned* const* proj_s; const unsigned* const* fc1_s; const unsigned* const* fc2_s;
  /* 1: the q rows of every qkv_w[l] / qkv_b[l] (and qkv_c[l]) carry the factor log2(e) / 8 -- timm Attention's `q * self.scale`
   * and the base change of the softmax folded into the frozen weights before their one rounding; the attention kernels then run as
   * variant | MVF_ATTN_Q_PRESCALED.  16-bit / fp8 modes, token counts outside 193 .. 208 only (mvf_vit_attn_q_prescaled). */
  int q_prescaled;
} MvfVitWeights;

size_t mvf_vit_workspace_bytes(int dtype, int frames_per_chunk, int tokens, int dim, int patch);

/* frames [F,3,img,img] fp32 (normalised) -> taps_out[j] [F*(tokens-1), dim] dtype T (block taps[j] output,
 * CLS row dropped), cls_out [F, dim] fp32 (final LN, token 0; may be NULL).  frames_per_chunk <= 0: all. */
int mvf_vit_fwd(const MvfVitWeights* w, int dtype, const float* frames, int F, void* const* taps_out, float* cls_out,
                void* workspace, size_t ws_bytes, int frames_per_chunk, int attn_variant, hipStream_t stream);
/* The same, also handing out the fp32 residual stream x_out [F*tokens, dim] (CLS row included) after the last of the
 * w->depth blocks: the frozen FRONT END of a partially frozen backbone (ViTFrontEnd, models/transformer.py:342-361).
 * depth may be 0 (patch + position embedding only); taps_out / cls_out may be NULL. */
int mvf_vit_fwd_x(const MvfVitWeights* w, int dtype, const float* frames, int F, void* const* taps_out, float* cls_out,
                  float* x_out, void* workspace, size_t ws_bytes, int frames_per_chunk, int attn_variant,
                  hipStream_t stream);
/* Blocks [first_block, first_block + n_blocks) on a residual stream the caller supplies: x [F*tokens, dim] fp32 (CLS row
 * included), updated in place -- timm Block.forward x n_blocks (x += attn(norm1(x)); x += mlp(norm2(x)), LayerScale where the
 * model has it), the loop body of VisionTransformer.forward_features reached from models/transformer.py:188.  For the
 * per-block ("teacher-forced") parity checks, which feed every block the oracle's input.  No taps, no final norm.  bf16:
 * first_block must not be a folded-LayerNorm consumer (pack with the fold off).  workspace: mvf_vit_workspace_bytes(dtype, F, ..). */
int mvf_vit_blocks_fwd(const MvfVitWeights* w, int dtype, float* x, int F, int first_block, int n_blocks, void* workspace,
                       size_t ws_bytes, int attn_variant, hipStream_t stream);

/* measurement hooks (bench.py roofline): when enabled, every GEMM launch of mvf_vit_fwd is bracketed by HIP events
 * on the launch stream; collect() waits for them and returns, per GEMM shape (epilogue kind, N, K) -- group g <
 * *n_groups -- the summed device milliseconds, the summed algorithmic FLOPs (2*M*N*K) and the launch count.  Not for
 * use under graph capture. */
int mvf_prof_enable(int on);
int mvf_prof_collect(double* ms_host, double* flops_host, int* count_host, int* epi_host, int* n_host, int* k_host,
                     int max_groups, int* n_groups_host);

/* pieces of the same path, exported for unit parity tests */
int mvf_gemm_tc(int dtype, int epi, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                float* resid, int ldr, void* tap, int ldt, const float* pos, const float* ls, int tokens_per_frame, int M,
                int N, int K, hipStream_t stream);
/* mvf_gemm_tc (bf16, K % 128 == 0, no pos) with the epilogue extras of the LN fold:
 *   epi 2 (residual): xb [M, ldxb] bf16 = the updated residual row rounded to bf16, stats [N/64][M][2] = per row and 64-column
 *     slice (sum, sum of squares) of the updated residual (either may be NULL);
 *   epi 0 / 1: ln_mr [M][2] = (mean, rstd) per row, ln_c [N]:  C = act(rstd * (A W^T - mean * ln_c) + bias).
 * mvf_ln_stats_finalize turns the partial sums into (mean, rstd = 1/sqrt(var + eps)) with the biased variance over D. */
int mvf_gemm_tc_ln(int dtype, int epi, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                   float* resid, int ldr, void* tap, int ldt, const float* ls, int tokens_per_frame, void* xb, int ldxb,
                   float* stats, const float* ln_mr, const float* ln_c, int M, int N, int K, hipStream_t stream);
int mvf_ln_stats_finalize(const float* part, int ns, float* mean_rstd, int rows, int D, float eps, hipStream_t stream);
/* epi 0 / 1 of mvf_gemm_tc_ln fed with the partial sums themselves (part [ns][M][2], D = K): the persistent 256x256 kernel stages a
 * tile's rows of them during the tile's first K tile and finalizes them in its second (same arithmetic as mvf_ln_stats_finalize,
 * bit for bit) -- no launch between the producing residual GEMM and the consuming one (timm Block.forward: norm1 -> attn.qkv,
 * reached from models/transformer.py:188).  MVF_ERR_UNSUPPORTED: ns > 12, odd M, K < 256, or the 128x128 kernel pinned. */
int mvf_gemm_tc_ln_part(int epi, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                        const float* part, int ns, float eps, const float* ln_c, int M, int N, int K, hipStream_t stream);
/* Deferred residual of the attention branch (bf16 mode, timm Block without LayerScale: x = x + proj(attn); x = x + mlp(norm2(x))):
 * proj stores delta = bf16(A W^T + b) with the plain epilogue (mvf_gemm_tc, epi 0) instead of read-modifying the fp32 residual;
 *   mvf_layernorm_add_fwd   y = LayerNorm(x + delta)          (x untouched)
 *   mvf_gemm_tc_resid2      resid += A W^T + bias + addend2   (fc2: addend2 = delta, bf16 [M, ld2]; tap as in mvf_gemm_tc)
 * -- the branch output is rounded to 16 bits before the add, as under the reference's autocast (models/transformer.py:188) */
int mvf_layernorm_add_fwd(int out_dtype, const float* x, size_t in_stride, const void* add_bf16, size_t add_stride,
                          const float* g, const float* b, void* y, size_t out_stride, int rows, int D, float eps,
                          hipStream_t stream);
int mvf_gemm_tc_resid2(const void* A, int lda, const void* W, int ldw, const float* bias, float* resid, int ldr,
                       const void* addend2, int ld2, void* tap, int ldt, int tokens_per_frame, int M, int N, int K,
                       hipStream_t stream);
/* out [M, ldo] fp32 = A W^T + bias [+ addend [M, ldo]] with bf16 A [M, K] / W [N, K]: the residual epilogue out of place
 * (addend NULL: none) -- nn.Linear forward / input gradient of the TRAINABLE backbone blocks (transformer.py:364-392) */
int mvf_gemm_tc_f32(const void* A, int lda, const void* W, int ldw, const float* bias, float* out, int ldo, const float* addend,
                    int M, int N, int K, hipStream_t stream);
/* MX-fp8 (BASELINE configs[4], MI355X.COMPUTE_DTYPE fp8).  Values: OCP e4m3 bytes, row-major; scales: one E8M0 byte per 32
 * consecutive k of a row, the four bytes of a 128-wide K tile in one dword, laid out scales[K/128][rows] (block b in byte b);
 * scale rule: the smallest power of two with amax / scale <= 448.
 *   mvf_quant_mxfp8      x [rows, K] (bf16 or f32, leading dimension ldx ELEMENTS) -> q [rows, K] (ldq bytes) + scales
 *   mvf_layernorm_mxfp8  LayerNorm(x) (timm norm1 / norm2, fp32 rows `in_stride` floats apart) -> q + scales; D % 256 == 0
 *   mvf_gemm_fp8         C = epi(A W^T + bias) like mvf_gemm_tc (epi 0 / 1: bf16 C, epi 2: fp32 residual + bf16 tap) with
 *                        MX-fp8 A [M, K] / W [N, K] on v_mfma_scale_f32_16x16x128_f8f6f4; K % 256 == 0, N % 32 == 0.
 *                        epi 1 with c_scales != NULL: the GELU output is quantised in the epilogue -- C = e4m3 bytes
 *                        [M, ldc], c_scales [N/128][M] (N % 128 == 0): the next GEMM's A operand, no bf16 round trip */
int mvf_quant_mxfp8(int in_dtype, const void* x, size_t ldx, void* q, size_t ldq, unsigned* scales, int rows, int K,
                    hipStream_t stream);
int mvf_layernorm_mxfp8(const float* x, size_t in_stride, const float* g, const float* b, void* q, size_t ldq, unsigned* scales,
                        int rows, int D, float eps, hipStream_t stream);
/* mvf_gemm_fp8 with LayerNorm 1 folded into the qkv GEMM (fp8 mode's form of mvf_gemm_tc_ln; timm Block.forward `attn(norm1(x))`):
 *   epi 2 (producer, the previous block's fc2): besides the fp32 residual update also xq [M, ldxq] = MX-fp8 of the NEW residual row
 *          (quantised from fp32, un-normalised), xq_scales [N/128][M] and stats [N/64][M][2] = per row and 64-column slice (sum, sum
 *          of squares) -- mvf_ln_stats_finalize turns them into (mean, rstd); N % 128 == 0; addend2 as mvf_gemm_tc_resid2 (or NULL)
 *   epi 0 (consumer, qkv): A = that xq, W = MX-fp8(gamma (.) W_qkv), bias = b + W beta, ln_c[n] = sum_k dequantised W'[n, k]:
 *          C = bf16(rstd * (A W'^T - mean * ln_c) + bias) with (mean, rstd) = ln_mr [M][2] */
int mvf_gemm_fp8_ln(int epi, const void* A, int lda, const unsigned* sa, const void* W, int ldw, const unsigned* sw, const float* bias,
                    void* C, int ldc, float* resid, int ldr, void* tap, int ldt, const float* ls, int tokens_per_frame,
                    const void* addend2, int ld2, void* xq, int ldxq, unsigned* xq_scales, float* stats, const float* ln_mr,
                    const float* ln_c, int M, int N, int K, hipStream_t stream);
/* timm Attention core (as mvf_vit_attn_fwd, bf16 qkv, any N, H even) with the MX-fp8 quantisation of its output in the kernel's
 * epilogue: q [F*N, D] e4m3 bytes + scales [D/128][F*N] -- bit for bit mvf_quant_mxfp8(mvf_vit_attn_fwd(qkv)), the A operand of
 * the fp8 proj GEMM, without the bf16 [F*N, D] tensor's round trip through HBM (reached from models/transformer.py:188 in fp8 mode) */
int mvf_vit_attn_fwd_mxfp8(const void* qkv, void* q, unsigned* scales, int F, int N, int H, int D, hipStream_t stream);
int mvf_gemm_fp8(int epi, const void* A, int lda, const unsigned* sa, const void* W, int ldw, const unsigned* sw,
                 const float* bias, void* C, int ldc, unsigned* c_scales, float* resid, int ldr, void* tap, int ldt,
                 const float* ls, int tokens_per_frame, int M, int N, int K, hipStream_t stream);
/* bf16 only, K % 128 == 0: M/batch_rows independent GEMMs stacked along M (batch_rows % 256 == 0), batch b using rows
 * [b * w_batch_rows, b * w_batch_rows + N) of W: the split-K form of a weight gradient dW = dY^T X over the tokens (each
 * batch one chunk of the token axis, fp32 partial sums with epi = 2 into a zeroed resid; trainable backbone blocks) */
int mvf_gemm_tc_batched(int epi, const void* A, int lda, const void* W, int ldw, void* C, int ldc, float* resid, int ldr,
                        int M, int N, int K, int batch_rows, int w_batch_rows, hipStream_t stream);
/* the same with plain fp32 partial results: `out` [M, ldo] need not be zeroed */
int mvf_gemm_tc_batched_f32(const void* A, int lda, const void* W, int ldw, float* out, int ldo, int M, int N, int K,
                            int batch_rows, int w_batch_rows, hipStream_t stream);
/* kernel choice for mvf_gemm_tc / mvf_vit_fwd (A/B measurements and tests): 0 automatic (bf16 and K % 128 == 0 ->
 * persistent 256x256 8-phase kernel, else 128x128), 1 always 128x128, 2 only 256x256 (MVF_ERR_UNSUPPORTED where it
 * cannot run), 3 the 256x256 kernel with one workgroup per tile instead of one per CU, 4 / 5 / 6 / 7 as 2 with the tile rows
 * pinned to 224 / 256 / 240 / 208 (0, 2, 3: the height that makes ceil(tiles / workgroups) x rows of a launch of three or
 * more rounds smallest, else 256) */
int mvf_gemm_tc_select(int variant);
/* diagnostic: out[2b] = XCD id, out[2b+1] = HW_ID of workgroup b of a 1-D launch (placement study, never on the path) */
int mvf_debug_xcc_map(int* out, int nblocks, int threads, int lds_bytes, hipStream_t stream);
/* CU budget of the persistent 256x256 kernel, which launches one workgroup per CU it may use and keeps it there for the whole
 * launch (its 140 KiB of LDS and 2 x 256 VGPRs per SIMD leave no room for another workgroup): 0 = all CUs of the device
 * (default).  A data-parallel run (utils/distributed.reserve_collective_cus) leaves 8 CUs -- one per XCD -- outside the
 * budget so that RCCL's all-reduce / all-gather kernels (DDP / SyncBN of CARL_MVF/train.py:283-286) never wait for a GEMM
 * launch to drain before they get a CU; also the CU count of a CU-masked stream. */
int mvf_gemm_tc_set_cus(int n);
/* CUs a persistent launch leaves free where that costs it no tile round (default 32; env MVF_GEMM_SPARE): with one workgroup per CU a
 * launch stays open until EVERY workgroup has run, so a workgroup whose CU is held by another queue's kernel -- the trainable head's
 * row-chain launches hold 24 CUs for 30 - 40 us each beside the next batch's backbone -- delays the whole launch, and for the short
 * lane-sized GEMMs (proj: two rounds of 15 us) by more than their own duration.  A lane-sized proj / fc2 (297 tiles, two rounds) runs on
 * 224 workgroups as fast as on 256; fc1 (1 188 tiles, five rounds) keeps 240.  0 = one workgroup per CU of the budget. */
int mvf_gemm_tc_set_spare(int cus);
/* *out = workgroups a persistent launch uses under the current budget (a multiple of 8, at least 8) */
int mvf_gemm_tc_get_wgs(int* out);
/* tile-list order of the persistent 256x256 kernel (speed only, results unchanged): g > 0 = grouped by weight panels -- groups of g
 * column tiles outermost, every row panel inside a group -- where g divides the launch's column-tile count, so that the tiles an
 * XCD works on at one time need g W panels instead of all of them; 0 = row panel major; -1 = chosen per launch (default: 4 where 4
 * divides a column-tile count of 8 or more, else 3 where 3 divides one of 6 or more; MVF_GEMM_NGROUP sets the initial value).  The cuBLAS analogue is a different tile rasterisation of the nn.Linear GEMMs behind models/transformer.py:188 */
int mvf_gemm_tc_set_ngroup(int g);
/* diagnostic build of the 256x256 kernel: per-block s_memtime stamps into buf[blocks][2][8] (NULL = off, the default) */
int mvf_gemm_tc_debug_stamps(unsigned long long* buf);
/* diagnostic, stamped build only: A rows are read as (row & mask), so A's footprint is mask + 1 rows (L2-resident feed rate) */
int mvf_gemm_tc_debug_rowmask(int mask);
/* diagnostic, stamped build only: kt >= 0 also stamps every workgroup barrier of K tile kt of each workgroup's second tile
 * (the stamps buffer then holds [blocks][2][8] + [blocks][2][16] entries) */
int mvf_gemm_tc_debug_ktile(int kt);
/* diagnostic, stamped build only: timing ablations (results are garbage): bit 0 no MFMAs, bit 1 no LDS fragment reads, bit 2 no
 * operand DMAs */
int mvf_gemm_tc_debug_ablate(int bits);
int mvf_patchify(int dtype, const float* frames, void* out, int F, int H, int W, int P, hipStream_t stream);
int mvf_layernorm_fwd(int out_dtype, const float* x, size_t in_stride, const float* g, const float* b, void* y,
                      size_t out_stride, int rows, int D, float eps, hipStream_t stream);
/* variant (bf16 / fp16): 0 = product path (N = 193..208: one key block, two 16-query tiles per wave; every other N: the streamed kernel
 * on 32-query-row tiles, csrc/vit_attn32.hip), 1 = 2-byte gather reads of V (cross-check of the transposing LDS read),
 * 2 = the earlier kernels (one tile per wave / synchronously staged 224-key blocks; what other values fall back to),
 * 4 = streamed 64-key blocks and 7 = streamed 96-key blocks on 16-query tiles (the streamed kernel of rounds 2-5), 6 = the one-block
 * kernel with VALU row sums, 8 + form = a form of the 32-query-row kernel for any N (mvf_vit_attn32_impl), 32 + form + 16 * waves = the
 * same with a forced workgroup size */
/* variant | MVF_ATTN_Q_PRESCALED (bf16 / fp16, variants 0, 6 and 13): the q columns of qkv already carry the factor log2(e) / 8 (the frozen
 * backbone folds it into the q rows of the packed qkv weights and bias, in fp32 before their one rounding): scores are base-2 exponents,
 * out = softmax2(q k^T) v with softmax2(s) = 2^s / sum 2^s */
#define MVF_ATTN_Q_PRESCALED 0x1000
int mvf_vit_attn_fwd(int dtype, const void* qkv, void* out, int F, int N, int H, int D, int variant, hipStream_t stream);
/* softmax normalisation convention of the 16-bit product kernels (variant 0 and the fused qkv + attention kernel): 1 = the row sum is taken
 * over the probabilities AFTER their rounding to bf16 / fp16, on the matrix pipe beside P.V (every N since round 6), 0 = over the fp32
 * values (the fp32 kernel).  Both are softmax to within the operand rounding; the emulating oracle follows this function's answer
 * (tests/test_abi.py). */
int mvf_vit_attn_rowsum_rounded(int dtype, int N);
/* 1 when a frozen backbone of N tokens per frame in `dtype` is packed with pre-scaled q rows (MvfVitWeights.q_prescaled; MVF_ATTN_QS=0 in
 * the environment switches it off): the one statement of that convention -- the packer and the emulating oracle both follow it
 * (tests/test_abi.py) */
int mvf_vit_attn_q_prescaled(int dtype, int N);
/* timm Attention.qkv FUSED into the attention core (reached from models/transformer.py:188): out [F*N, D] = per (frame, head)
 * softmax(q k^T / 8) v with [q | k | v] = A[f] W_h^T + bias (ln_c / ln_mr NULL), or the folded-LayerNorm form
 * rstd (A W'^T - mean ln_c) + bias of mvf_gemm_tc_ln (ln_c [3D]; the rows' statistics as ln_mr [F*N][2], or -- ln_mr NULL -- as the
 * producer's partial sums ln_part [ln_ns][F*N][2] with ln_eps, as mvf_gemm_tc_ln_part takes them).  A [F*N, lda] and W [3D, D] bf16 (MVF_BF16) or
 * fp16 (MVF_F16); A != out.  The [F*N, 3D] qkv tensor never reaches HBM; bit-identical to mvf_gemm_tc(_ln) + mvf_vit_attn_fwd.
 * MVF_ERR_UNSUPPORTED unless N = 193 .. 208, D = 64 H, D % 96 == 0. */
int mvf_vit_qkv_attn_fwd(int dtype, const void* A, int lda, const void* W, const float* bias, const float* ln_c, const float* ln_mr,
                         const float* ln_part, int ln_ns, float ln_eps, void* out, int F, int N, int H, int D, hipStream_t stream);
int mvf_cast_f32_bf16(const float* in, void* out, size_t n, hipStream_t stream);
int mvf_cast_f32_f16(const float* in, void* out, size_t n, hipStream_t stream);      /* IEEE half, round to nearest even; n % 4 == 0 */
int mvf_cast_bf16_f32(const void* in, float* out, size_t n, hipStream_t stream);     /* n % 4 == 0 */
/* ViT attention of a TRAINABLE block in bf16 (timm Attention inside ViTBackEnd, models/transformer.py:364-392; fp16 autocast
 * in the reference): forward = mvf_vit_attn_fwd on the streamed kernel, also writing the per-query log2-domain log-sum-exp
 * lse [F, H, 16 * ceil(N / 16)]; backward: qkv / o / d_o bf16 as in the forward's layout, delta = caller-owned scratch
 * shaped like lse, dqkv [F*N, 3*D] fp32 or bf16 (out_dtype MVF_F32 / MVF_BF16; every element of the q, k, v column blocks is written).  Owner-computes, no atomics. */
int mvf_vit_attn_fwd_lse(const void* qkv, void* out, float* lse, int F, int N, int H, int D, hipStream_t stream);
int mvf_vit_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv, int out_dtype,
                     int F, int N, int H, int D, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Trainable head, fp32, forward + backward
 * ---------------------------------------------------------------------------------------------- */
/* C[m,n] (+)= act(alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] + bias[n] + table[((m/div)%mod)*tab_si + n*tab_sn])
 *   replaces every nn.Linear fwd / input-grad / weight-grad on the path (models/mvformer.py:77,86,97;
 *   models/utils.py:65-68,182-183; models/resnet_c2d.py:117-120) and the PE add of utils.py:136-145 (table) */
int mvf_hgemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
              const float* bias, const float* table, long tab_si, long tab_sn, int tab_div, int tab_mod, int M, int N,
              int K, float alpha, int relu, int accumulate, hipStream_t stream);
/* the same with dropout and the residual connection fused into the epilogue: C = [resid +] dropout_p(act(...))
 *   (ResidualConnection `x + drop(sub(LN(x)))`, models/utils.py:153-159, when `sub` ends in a Linear); the dropout mask
 *   is the counter-based one of mvf_dropout_add with element index m*ldc + n */
int mvf_hgemm_ex(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
                 const float* bias, const float* table, long tab_si, long tab_sn, int tab_div, int tab_mod, int M, int N,
                 int K, float alpha, int relu, int accumulate, const float* resid, long ldr, float drop_p,
                 uint64_t drop_seed, uint64_t drop_offset, hipStream_t stream);
/* backward of y = x W^T + b in ONE launch (x [M,K], W [N,K], dy [M,N], unit inner strides): dx = dy W (may be NULL);
 * dW (+)= dy^T x; db (+)= colsum(dy) (may be NULL).  accumulate_params: add into dW/db (the flat gradient buffer)
 * instead of overwriting.  (ReLU / dropout backward of dy: mvf_relu_bwd / mvf_dropout_add first.) */
int mvf_hlinear_bwd(const float* dy, long ldy, const float* x, long ldx, const float* W, long ldw, float* dx, long lddx,
                    float* dW, long lddw, float* db, int M, int N, int K, int accumulate_params, hipStream_t stream);
/* LSTPCrossAtt's static queries folded through linear_K2d (models/mvformer.py:383, `Q = self.Q_s + self.Q_s_b`; the pooling rewrite
 * of mvf_lstp_fused_fwd takes wq = Q W_K): wq [nq, C] = (qs [nq, d] + qb [d]) . wk [d, C]; backward: gqs, gqb, gwk ACCUMULATE
 * (flat gradient buffer) from dv [nq, C].  nq <= 8.  One launch each way (plain fp32 FMAs, fixed order). */
int mvf_static_query_fwd(const float* qs, const float* qb, const float* wk, long ldw, float* wq, int nq, int d, int C,
                         hipStream_t stream);
int mvf_static_query_bwd(const float* dv, long ldv, const float* qs, const float* qb, const float* wk, long ldw, float* gqs,
                         float* gqb, float* gwk, long ldgw, int nq, int d, int C, hipStream_t stream);
int mvf_colsum(const float* x, long ld, int rows, int cols, float* out, int accumulate, hipStream_t stream);
/* stage 1 of a column sum over very many rows: part[s][c] = sum of row slice s (splits slices); stage 2: mvf_sum_batches */
int mvf_colsum_split(const float* x, long ld, int rows, int cols, int splits, float* part, hipStream_t stream);

/* dx = dy * [y > 0]  (ReLU backward of the FFN, models/utils.py:190) */
int mvf_relu_bwd(const float* dy, const float* y, float* dx, size_t n, hipStream_t stream);
/* exact-erf GELU (timm Mlp of a TRAINABLE ViT block): y = x Phi(x); dx = dy (Phi(x) + x phi(x)) */
/* split-K weight gradient helpers: out bf16 [S, C, Mc], out[s][c][j] = in[s*Mc + j][c] (fp32 in [M, C], zero beyond M;
 * Mc % 64 == 0, S = ceil(M / Mc));  out[i] (+)= sum over the S batches of part[s][i] */
int mvf_transpose_chunks(const float* in, void* out_bf16, int M, int C, int Mc, hipStream_t stream);
int mvf_sum_batches(const float* part, float* out, int S, size_t n, int accumulate, hipStream_t stream);
/* LayerScale of a trainable DINOv2 block: mode 0 out = resid + y * gamma[col]; 1 out = y * gamma[col]; 2 out = y * resid */
int mvf_colscale(const float* y, const float* gamma, const float* resid, float* out, int rows, int D, int mode,
                 hipStream_t stream);
int mvf_gelu_fwd(const float* x, float* y, size_t n, hipStream_t stream);
/* -- the HBM-bound kernels between the bf16 GEMMs of a TRAINABLE backbone block (csrc/vit_train.hip; timm Block inside
 *    ViTBackEnd, models/transformer.py:364-392, under fp16 autocast in the reference) --
 * exact-erf GELU on bf16 (n % 8 == 0): g = u Phi(u);  du = dg (Phi(u) + u phi(u)) */
int mvf_gelu_bf16(const void* u, void* g, size_t n, hipStream_t stream);
int mvf_gelu_bwd_bf16(const void* dg, const void* u, void* du, size_t n, hipStream_t stream);
/* one pass over in [M, C] (MVF_F32 / MVF_BF16, C % 4 == 0) that writes what the backward of a linear layer needs from it:
 *   rowmajor_bf16 [M, C] (fp32 input only), transposed_bf16 [ceil(M / Mc), C, Mc] (tr[s][c][j] = in[s Mc + j][c], zero beyond M;
 *   Mc % 256 == 0) and colsum_part [part_rows, C] fp32 (column sums of each slice of 256 rows: part_rows = ceil(R / 256)
 *   with R = M, or M rounded up to Mc when the transposed output is requested; mvf_sum_batches adds them); any may be NULL */
int mvf_grad_prep(int in_dtype, const void* in, void* rowmajor_bf16, void* transposed_bf16, float* colsum_part, int part_rows,
                  int M, int C, int Mc, hipStream_t stream);
/* LayerNorm backward with the statistics recomputed from x (as mvf_layernorm_fwd computes them):
 *   dx = [dres +] d LN / dx (dh);  dx_bf16 = bf16(dx) (may be NULL);  dg += sum_rows dh xhat;  db += sum_rows dh  (atomics) */
int mvf_ln_bwd_block(const float* dh, const float* x, const float* g, const float* dres, float* dx, void* dx_bf16, float* dg,
                     float* db, int rows, int D, float eps, hipStream_t stream);
int mvf_gelu_bwd(const float* dy, const float* x, float* dx, size_t n, hipStream_t stream);

/* y = resid + dropout_p(x) with a counter-based mask (nn.Dropout + residual add, models/utils.py:153-159;
 * mvformer.py:76; resid may be NULL); backward = same call on dy with resid = NULL */
int mvf_dropout_add(const float* x, const float* resid, float* y, size_t n, float p, uint64_t seed, uint64_t offset,
                    hipStream_t stream);

/* LayerNorm  (models/utils.py:147-159, eps 1e-5) */
int mvf_ln_fwd(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd, int rows, int D,
               float eps, hipStream_t stream);
int mvf_ln_bwd(const float* dy, const float* x, const float* g, const float* mean, const float* rstd, float* dx, float* dg,
               float* db, int rows, int D, int accumulate_dx, int accumulate_params, hipStream_t stream);
/* dx = dres + d LN / dx (dy): x of a pre-LN residual connection x + sub(LN(x)) (ResidualConnection, models/utils.py:147-159)
 * receives two gradients, the residual path's (dres) and the LayerNorm's; summed in the LayerNorm backward's pass */
int mvf_ln_bwd_res(const float* dy, const float* x, const float* g, const float* mean, const float* rstd, const float* dres,
                   float* dx, float* dg, float* db, int rows, int D, int accumulate_params, hipStream_t stream);

/* BatchNorm1d (+fused ReLU)  (models/mvformer.py:78-79, resnet_c2d.py:118-119); SyncBN = caller merges the
 * (mean, var) / (s1, s2) vectors across ranks between the two halves (train.py:283) */
size_t mvf_bn_workspace_floats(int rows, int C);   /* scratch of mvf_bn_stats / mvf_bn_bwd_reduce (two-stage column sums) */
int mvf_bn_stats(const float* x, int rows, int C, float* mean, float* var, float* running_mean, float* running_var,
                 float momentum, float* ws, size_t ws_floats, hipStream_t stream);
/* SyncBatchNorm's merge (train.py:283-286): gathered [W][2C + 1] = every rank's (mean [C], biased var [C], row count) -> mean / var [C]
 * of the rank-concatenated batch (equal row counts: count_per_rank rows on every rank) and, when given, the running buffers updated with
 * the unbiased variance over all W * count_per_rank rows.  The all-gather itself is the caller's (torch.distributed / RCCL). */
int mvf_syncbn_merge(const float* gathered, int W, int C, float count_per_rank, float* mean, float* var, float* running_mean,
                     float* running_var, float momentum, hipStream_t stream);
int mvf_bn_fwd(const float* x, const float* mean, const float* var, const float* g, const float* b, float* y, int rows,
               int C, float eps, int relu, hipStream_t stream);
int mvf_bn_bwd_reduce(const float* dy, const float* x, const float* mean, const float* var, const float* g, const float* b,
                      float* s1, float* s2, float* dgamma, float* dbeta, int accumulate_params, int rows, int C, float eps,
                      int relu, float* ws, size_t ws_floats, hipStream_t stream);
int mvf_bn_bwd_apply(const float* dy, const float* x, const float* mean, const float* var, const float* g, const float* b,
                     const float* s1, const float* s2, float* dx, int rows, int C, float eps, int relu, float count,
                     hipStream_t stream);

/* entity one-hot concat (mvformer.py:144-149), entity reduction (mvformer.py:181-195), F.normalize (transformer.py:228) */
int mvf_concat_onehot(const float* x, float* out, int rows, int cin, int ntok, int div, hipStream_t stream);
int mvf_final_reduce_fwd(const float* x, float* y, int* arg, int B, int ntok, int T, int D, int mode, hipStream_t stream);
int mvf_final_reduce_bwd(const float* dy, const int* arg, float* dx, int B, int ntok, int T, int D, int mode,
                         hipStream_t stream);
int mvf_l2norm_fwd(const float* x, float* y, float* nrm, int rows, int D, float eps, hipStream_t stream);
int mvf_l2norm_bwd(const float* dy, const float* y, const float* nrm, float* dx, int rows, int D, float eps,
                   hipStream_t stream);

/* temporal multi-head self-attention (models/utils.py:11-44,88-104); qkv [B*S, 3*Dm]; mask [B, mask_len] or NULL (1 = keep,
 * 0 = masked key), key s reading column s % mask_len: mask_len = S is a plain key mask, mask_len = T the frame mask of the joint
 * (entity, frame) sequence S = ntok * T, which mvformer.py:171-172 tiles to [B, 1, S] first */
/* 1 = scalar-FMA kernels only (cross-check, A/B), 0 = fp32 matrix-core kernels when dk is 16, 32 or 64 (default) */
int mvf_tattn_select(int scalar_only);
int mvf_tattn_fwd(const float* qkv, const float* mask, int mask_len, float* o, float* lse, int B, int S, int H, int Dm,
                  hipStream_t stream);
int mvf_tattn_bwd(const float* qkv, const float* mask, int mask_len, const float* o, const float* lse, const float* d_o,
                  float* dqkv, int B, int S, int H, int Dm, hipStream_t stream);

/* LSTP learned-query pooling passes over the tap tensors (models/mvformer.py:243-266,352-414) */
int mvf_lstp_scores(const void* const* taps_host, int n_taps, int dtype, int D, int F, int N, int T, int nq,
                    const float* vec, int per_frame, float* scores, hipStream_t stream);
int mvf_lstp_wsum(const void* const* taps_host, int n_taps, int dtype, int D, int F, int N, int T, int nq, const float* w,
                  float* out, hipStream_t stream);
int mvf_lstp_softmax_fwd(const float* scores, float* P, float* Pm, float* rowsum, int F, int N, int nq, float inv_sqrt_d,
                         int disjoint, hipStream_t stream);
int mvf_lstp_softmax_bwd(const float* P, const float* Pm, const float* dP, const float* drow, float* dS, int F, int N,
                         int nq, float inv_sqrt_d, hipStream_t stream);
int mvf_lstp_reduce_frames(const float* G, float* out, int Bc, int nq, int T, int C, hipStream_t stream);
/* late fusion (TransformerEmbModel, models/transformer.py:258-262,287): out[f, :] = max (mode 0) | mean (mode 1) over the N
 * tokens of frame f, taps concatenated along the channels; forward only (frozen backbone) */
int mvf_token_pool(const void* const* taps_host, int n_taps, int dtype, int D, int F, int N, int mode, float* out,
                   hipStream_t stream);
/* gradient of the pooling w.r.t. the tokens (partially frozen backbone):
 * dx[t][f*N+n, :] = sum_j W[f,j,n] dpooled[b,j,t,:] + dS[f,j,n] vec[f|0,j,:]; dx_host: HOST array of n_taps fp32 [F*N, D];
 * either term is skipped when its pair (w, dpooled) / (ds, vec) is NULL */
int mvf_lstp_dx(float* const* dx_host, int n_taps, int D, int F, int N, int T, int nq, const float* w, const float* ds,
                const float* dpooled, const float* vec, int per_frame, hipStream_t stream);

/* One-pass forms of the pooling (static or per-frame queries, no SMART_DISJOINT, frozen taps): the tap tensors are read once
 * forward (online softmax over a frame's tokens: pooled [Bc, nq, T, C] and the normalised weights P [F, nq, N] =
 * LSTPCrossAtt.attn_matrix, mvformer.py:409-411) and once backward (G [Bc, nq, T, C] = d loss / d vec per frame, from dpooled,
 * P and the forward's pooled: sum_n dS_jn x_n = c (sum_n P_jn g_jn x_n - gbar_j pooled_j), g_jn = dpooled_j . x_n).
 * MVF_ERR_UNSUPPORTED outside the kernels' register budget (nq > 3, more than 3 taps, D > 1024 or D % 8 != 0; fp32 taps, or
 * form 1, also 3 queries on 3 taps of 1024 channels -- that shape exists in the matrix-core form only): the caller then runs the
 * mvf_lstp_scores / _softmax_fwd / _wsum chain. */
int mvf_lstp_fused_fwd(const void* const* taps, int n_taps, int dtype, int D, int F, int N, int T, int nq, const float* vec,
                       int per_frame, float inv_sqrt_d, float* P, float* pooled, hipStream_t stream);
int mvf_lstp_fused_bwd(const void* const* taps, int n_taps, int dtype, int D, int F, int N, int T, int nq, const float* dpooled,
                       const float* P, const float* pooled, float inv_sqrt_d, float* G, hipStream_t stream);
/* bf16 taps run the two products on the matrix cores (csrc/lstp_mfma.hip: token tiles of 16 staged once in LDS, the small fp32
 * operands as bf16 hi + lo pairs) where the shape has an instantiation (1 or 3 taps, D = 768 | 1024, nq <= 3).  form = 1 pins the
 * VALU kernels (tests, A/B measurements), 0 restores the automatic choice. */
int mvf_lstp_select(int form);

/* ------------------------------------------------------------------------------------------------
 * Row-chain kernels of the trainable head on the 16-bit matrix cores (csrc/head_chain.hip): one launch walks a chain of
 * row-wise operators over 32-row panels (activations in LDS, bf16 weights streamed from L2).  Used when the head runs in
 * bf16 (MI355X.HEAD_DTYPE, default = bf16 unless COMPUTE_DTYPE is fp32 -- the reference's head runs under fp16 autocast,
 * train.py:113-117); GEMM operands bf16, accumulation / LayerNorm / dropout / residual stream fp32.
 * ---------------------------------------------------------------------------------------------- */
typedef struct MvfDrop {       /* counter-based dropout of mvf_dropout_add (element index = row * width + column); p == 0: identity */
  float p;
  uint64_t seed, offset;
} MvfDrop;

/* FRAGMENT-MAJOR ("FM") image of a bf16 matrix X[rows][red] (red = the reduction index of the GEMM it feeds):
 *     FM[rows / 16][red / 32][64][8]     lane = row % 16 + 16 * ((red % 32) / 8), element = red % 8
 * -- each 16 x 32 block stored as the 64 lanes of v_mfma_f32_16x16x32_bf16 hold it, so a wave fetches a fragment with one coalesced
 * 1 KB load; rows padded to a multiple of 64, red to a multiple of 128 (zeros).
 * bf16 operand copies of nn.Linear weights: w fp32 [N, K] (row stride ld) -> w16 = FM image of W (rows n, reduction k: the forward
 * operand) and w16t = FM image of W^T (rows k, reduction n: the input-gradient operand); either may be NULL.  n <= 32 entries, one
 * launch.  mvf_head_pack_elems: elements of the two images. */
typedef struct MvfPackEntry {
  const float* w;
  long ld;
  int N, K;
  void* w16;
  void* w16t;
  int f16;                     /* != 0: w16 (the FORWARD operand) is written as IEEE fp16 (MI355X.HEAD_DTYPE fp16); w16t stays bf16 */
} MvfPackEntry;
int mvf_head_pack_weights(const MvfPackEntry* entries_host, int n, hipStream_t stream);
/* measurement knob of tools/chain_probe.py (0 = product behaviour): see csrc/head_chain.hip g_chain_dbg */
int mvf_head_chain_debug(int bits);
/* diagnostic: pull `bytes` of read-only data into every XCD's L2 (256 workgroups, slice b / 8 each) */
int mvf_head_l2_warm(const void* p, size_t bytes, hipStream_t stream);
int mvf_head_chain_debug_stamps(long long* stamps16);   /* device buffer of 16 int64, or NULL (default): stage time stamps of workgroup 0 */
int mvf_rowlin_debug_stamps(long long* buf, int slots);   /* slots x 16 int64 (or NULL): stage stamps of the next `slots` mvf_rowlin_* launches */
size_t mvf_head_pack_elems(int N, int K, int transposed);

/* One temporal EncoderLayer minus its attention core (models/utils.py:196-226; ResidualConnection :147-159,
 * MultiheadedAttention :75-108, PositionwiseFeedForward :176-194), rows m of x [M, D]:
 *   segment A (o != NULL):     x1 = x_in + drop_attn(o Wo^T + bo);  h1 = LN1(x1);  a = relu(h1 W1^T + b1);
 *                              x2 = x1 + drop_ffn(a W2^T + b2)                       [o = attention output of this layer]
 *   segment B (wqkv != NULL):  h0 = LN0(x2, or x_in without segment A);  qkv = h0 Wqkv^T + bqkv   [the NEXT layer's Q|K|V]
 * w*: bf16 copies from mvf_head_pack_weights (w16).  Saved for the backward (each may be NULL): x1, mean1 / rstd1, a
 * (bf16 [M, DFF]), mean0 / rstd0, and the operands of the weight gradients oT, h1T, h0T, aT: FM images of the TRANSPOSED
 * activations [D or DFF (padded to 64), Mp] (rows = features, reduction = the row index m; Mp multiple of 128, >= M; m >= M zero).  D % 256 == 0, DFF % 256 == 0, D <= 512; MVF_ERR_UNSUPPORTED when the
 * panels do not fit 160 KB of LDS. */
typedef struct MvfEncFwd {
  int M, D, DFF, Mp;
  float ln_eps;
  const float* o;
  const float* x_in;
  const void *wo, *w1, *w2;
  const float *bo, *b1, *b2, *ln1_g, *ln1_b;
  MvfDrop drop_attn, drop_ffn;
  float *x1, *mean1, *rstd1;
  void* a;
  float* x2;
  void *oT, *h1T, *aT;
  const void* wqkv;
  const float *bqkv, *ln0_g, *ln0_b;
  float *qkv, *mean0, *rstd0;
  void* h0T;
  int f16;                     /* != 0: wo / w1 / w2 / wqkv are IEEE fp16 images and the four GEMMs run on fp16 operands (the saved a and the
                                * transposed images oT / h1T / aT / h0T stay what the bf16 backward reads: a's sign / zero bits, bf16 values) */
} MvfEncFwd;
int mvf_enc_layer_fwd(const MvfEncFwd* args_host, hipStream_t stream);

/* The same chain backwards.  dres [M, D]: the gradient arriving on the residual stream (of the layer's output when only
 * segment A' runs; dx1 of the layer whose Q|K|V gradient dqkv is consumed when segment B' runs).
 *   segment B' (dqkv != NULL):  dh0 = dqkv Wqkv;  dx = dres + LN0'(dh0)   -> dx_out (may be NULL when A' follows)
 *   segment A' (w2T != NULL):   on dy = dx (or dres):  g2 = mask_ffn(dy);  du = (g2 W2) [a > 0];  dh1 = du W1;
 *                               dx1 = dy + LN1'(dh1) -> dx1_out;  go = mask_attn(dx1);  d_o = go Wo
 * w*T: the transposed images (w16t).  dqkvT, g2T / goT, duT: FM images of the transposed output gradients [3D | D | DFF, Mp]
 * for mvf_head_dw (may be NULL).  dln*_g / dln*_b: LayerNorm parameter gradients, ACCUMULATED with float atomics. */
typedef struct MvfEncBwd {
  int M, D, DFF, Mp;
  const float* dqkv;
  const void* wqkvT;
  const float *x_in, *mean0, *rstd0, *ln0_g;
  float *dln0_g, *dln0_b;
  void* dqkvT;
  const float* dres;
  float* dx_out;
  MvfDrop drop_ffn, drop_attn;
  const void *w2T, *w1T, *woT;
  const void* a;
  const float *x1, *mean1, *rstd1, *ln1_g;
  float *dln1_g, *dln1_b;
  void *g2T, *duT, *goT;
  float *dx1_out, *d_o;
} MvfEncBwd;
int mvf_enc_layer_bwd(const MvfEncBwd* args_host, hipStream_t stream);

/* One Linear of the head with everything row-wise around it (csrc/head_rowlin.hip): the per-entity FC stack and video_emb
 * (models/mvformer.py:70-86,150-160), entity reduction + embedding layer (mvformer.py:181-199), projection head + normalisation
 * (models/resnet_c2d.py:112-126, models/transformer.py:226-228).  Rows m < M, 32 per workgroup.
 *   Y [M, N] = epi( pro(X) W^T + bias ),   pro, in this order (each optional):
 *      entity reduction (g_ntok > 0): X is [B, g_ntok, g_T, Cin], row m = (b, t) takes token 0 (g_mode 0) / the mean (1) / the max (2,
 *        arg-max written to g_arg [M, Cin]) over the g_ntok entities;
 *      BatchNorm1d (+ReLU) with the GIVEN batch statistics bn_mean / bn_var (biased), affine bn_g / bn_b;
 *      entity one-hot: oh_ntok columns appended, column j = [(m / oh_div) % oh_ntok == j];
 *      dropout drop_in (element index m * (Cin + oh_ntok) + c);
 *   epi: + table[(m % tab_mod), :] (sin/cos positions), dropout drop_out (index m * N + n), or l2norm != 0: Y / max(||Y||, l2_eps)
 *      (norms to nrm [M]).
 *   st_part != NULL: the batch statistics of Y for the BatchNorm that FOLLOWS: st_mean / st_var [N] (biased variance; Chan's merge
 *      of per-workgroup mean / M2, fixed order, finished by the workgroup that arrives last) and, if given, the running
 *      statistics updated like nn.BatchNorm1d (momentum, unbiased variance).  st_part: 2 * ceil(M / 32) * N floats of scratch.
 *   xT (may be NULL): FM image of pro(X)^T [Cin + oh_ntok (padded to 64), Mp] for mvf_head_dw.
 * w16: mvf_head_pack_weights image.  Cin % 4 == 0, N % 128 == 0, N <= 512, Cin + oh_ntok <= 512 (else MVF_ERR_UNSUPPORTED). */
typedef struct MvfRowLinFwd {
  int M, Cin, N, Mp;
  const float* X;
  long ldx;
  int g_ntok, g_T, g_mode;
  int* g_arg;
  const float *bn_mean, *bn_var, *bn_g, *bn_b;
  float bn_eps;
  int bn_relu;
  int oh_ntok, oh_div;
  MvfDrop drop_in, drop_out;
  const void* w16;
  const float* bias;
  const float* table;
  int tab_mod;
  int l2norm;
  float l2_eps;
  float *Y, *nrm;
  void* xT;
  float *st_part, *st_mean, *st_var, *st_rmean, *st_rvar;
  float st_momentum;
  int f16;                     /* != 0: w16 is an IEEE fp16 image (MvfPackEntry.f16) and the GEMM runs on fp16 operands; xT stays bf16 */
} MvfRowLinFwd;
int mvf_rowlin_fwd(const MvfRowLinFwd* args_host, hipStream_t stream);

/* The same stage backwards.  dY [M, N]: the gradient w.r.t. Y -- or, nb_Y != NULL, the masked gradient dZ w.r.t. the OUTPUT of the
 * BatchNorm (+ReLU) that consumes Y, which is turned into dY here: dY = g rstd (dZ - s1 / count - xhat s2 / count) with that
 * BatchNorm's statistics (nb_mean, nb_var), affine weight nb_g, input nb_Y (= the forward's Y) and column sums nb_s1 / nb_s2.
 * Then epi' (dropout mask drop_out | l2norm backward with the forward's normalised output l2_y and norms l2_nrm), g -> gT (FM
 * transpose for mvf_head_dw, may be NULL), dXp = g W (w16t image), pro' (dropout mask drop_in, one-hot columns dropped, ReLU mask of
 * the forward's BatchNorm recomputed from X and its statistics) -> dX [M, Cin] (row stride lddx; scattered back to
 * [B, g_ntok, g_T, Cin] through the entity reduction).  With bn_mean != NULL what is written to dX is dZ, the masked gradient w.r.t.
 * that BatchNorm's output, and st_part != NULL also gives its column sums s1 = sum dZ, s2 = sum dZ xhat (2 * ceil(M / 32) * Cin
 * floats of scratch; the last workgroup to arrive adds them) and ACCUMULATES dgamma += s2, dbeta += s1 (may be NULL). */
typedef struct MvfRowLinBwd {
  int M, Cin, N, Mp;
  const float* dY;
  const float *nb_Y, *nb_mean, *nb_var, *nb_g, *nb_s1, *nb_s2;
  float nb_eps, nb_count;
  MvfDrop drop_out, drop_in;
  int l2norm;
  const float *l2_y, *l2_nrm;
  float l2_eps;
  const void* w16t;
  void* gT;
  int oh_ntok;
  const float* X;
  long ldx;
  const float *bn_mean, *bn_var, *bn_g, *bn_b;
  float bn_eps;
  int bn_relu;
  float *st_part, *s1, *s2, *dgamma, *dbeta;
  int g_ntok, g_T, g_mode;
  const int* g_arg;
  float* dX;
  long lddx;
} MvfRowLinBwd;
int mvf_rowlin_bwd(const MvfRowLinBwd* args_host, hipStream_t stream);

/* Weight / bias gradients of n <= 16 Linears in one launch: dw[n][k] (+)= sum_m gT[n][m] xT[k][m], db[n] (+)= sum_m gT[n][m]
 * (db may be NULL); gT, xT: FM images of [N, Mp] and [K, Mp] (the transposed saves above; Mp % 128 == 0).  accumulate != 0: add into dw / db (the flat
 * gradient buffer).  Fixed summation order (no atomics). */
typedef struct MvfDwProblem {
  const void* gT;
  const void* xT;
  float* dw;
  long lddw;
  float* db;
  int N, K;
} MvfDwProblem;
int mvf_head_dw(const MvfDwProblem* probs_host, int n, int Mp, int accumulate, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Sequence-contrastive loss (algos/scl.py:52-105), fused forward / backward
 *   negative_flags: bit0 'single' in NEGATIVE_TYPE, bit1 'noself' in NEGATIVE_TYPE
 * ---------------------------------------------------------------------------------------------- */
/* rows[3][M], M = clips * T: chosen_steps [clips, T] int64, seq_lens [clips] int64 (repeated over the clip's frames) and
 * video_masks [clips, T] fp32 (NULL = ones) as per-row floats -- the reshape / expand / float() prologue of
 * compute_sequence_loss (algos/scl.py:52-64) in one launch */
int mvf_scl_rows(const long long* steps, const long long* seq_lens, const float* masks, float* rows, int clips, int T,
                 hipStream_t stream);
int mvf_scl_fwd(const float* emb, const float* step, const float* len, const float* mask, float* S, float* R, float* c,
                float* lossrow, float* loss, int M, int E, int T, int negative_flags, float temperature,
                float label_variance, hipStream_t stream);
int mvf_scl_bwd(const float* emb, const float* step, const float* len, const float* mask, const float* S, const float* R,
                const float* c, const float* gout, float* dE, int M, int E, int T, int row0, int rows, int negative_flags,
                float temperature, float label_variance, hipStream_t stream);
/* E = 64 | 128 | 256 run the pair similarities (and the gradient's coefficient x embedding product) on v_mfma_f32_16x16x4_f32, exact
 * fp32: at the gathered size of 8 ranks (M = 2 048, configs[2]) 0.35 + 4.5 ms -> see profiles/r04/scl_gathered.txt.  form = 1 pins the
 * scalar kernels (any E <= 256; tests, A/B measurements), 0 restores the automatic choice. */
int mvf_scl_select(int form);

/* ------------------------------------------------------------------------------------------------
 * Optimiser: global-norm clip + Adam(L2) on one flat buffer (train.py:124-133,147-149; utils/optimizer.py:60-66)
 * ---------------------------------------------------------------------------------------------- */
/* norm_out: TWO floats. [0] = the norm; [1] (zeroed once by the caller) is incremented whenever the norm is NaN/Inf.
 * mvf_adam_step given that pair skips the update on a non-finite norm and leaves skipped steps out of its bias correction --
 * the GradScaler.step behaviour of the reference's fp16 path (train.py:127-133). */
int mvf_grad_norm(const float* g, size_t n, const float* extra_sq, float* scratch, float* norm_out, hipStream_t stream);
/* zero_grad != 0: g is zeroed as it is consumed (and when the step is skipped): the next iteration's optimizer.zero_grad()
 * (train.py:113) costs no pass of its own */
int mvf_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, float clip, const float* norm, float gscale, int zero_grad,
                  hipStream_t stream);
/* compute units the optimizer's two launches may hold: 0 = wide forms (every CU that falls free), > 0 = that many whole-CU workgroups
 * (beside the persistent backbone GEMM a CU with any optimizer workgroup on it is lost to the GEMM: few CUs for longer cost the pipelined
 * step less than all CUs briefly).  Default: env MVF_OPT_WIDTH, else the library's measured choice. */
int mvf_optim_set_width(int cus);

/* ------------------------------------------------------------------------------------------------
 * GPU-side view augmentation: one clip [T,3,H,W] of floats in [0,1] -> [T,3,S,S], normalised
 *   replaces: train.preproc_views (train.py:39-53) applying the ComposeOp of datasets/data_augment.py:372-413
 *   (random_resized_crop :287-317, flip :10-14, torchvision ColorJitter :340-352, torchvision GaussianBlur :357-365,
 *   grayscale :61-78, resize :16-22, color_normalization :218-238) or of create_data_augment(augment=False) :416-456
 *   (uniform_crop :24-59, resize, color_normalization).  The random draws stay on the host, in the reference's order
 *   (video_rep_learning_amd/datasets/augment.py); a clip's draws arrive here as one MvfAugmentParams.
 *   Steps, in this order: bilinear resize (align_corners = False) of the crop window to S x S; horizontal flip;
 *   n_color colour steps (torchvision float-tensor semantics, each clamped to [0,1]; contrast uses the per-FRAME mean of
 *   the 0.2989/0.587/0.114 gray); blur_kx x blur_ky Gaussian with reflect padding; 0.299/0.587/0.114 grayscale on all
 *   channels; (v - mean) / std.
 * ---------------------------------------------------------------------------------------------- */
typedef struct MvfAugmentParams {
  int crop_top, crop_left, crop_h, crop_w; /* source window inside H x W (whole frame: 0, 0, H, W) */
  int flip;
  int n_color;                             /* 0..4 */
  int color_op[4];                         /* 0 brightness, 1 contrast, 2 saturation, 3 hue; each at most once */
  float color_factor[4];
  int blur_kx, blur_ky;                    /* odd, <= 15 (reference: 5, 9); used when blur_sigma > 0 */
  float blur_sigma;                        /* <= 0: no blur */
  int gray;
  float mean[3], std[3];
} MvfAugmentParams;
size_t mvf_augment_workspace_bytes(int n_clips, int T, int S);
/* in [n_clips,T,3,H,W], out [n_clips,T,3,S,S], params: HOST array of n_clips entries (copied into the launches) */
int mvf_augment_clips(const float* in, float* out, int n_clips, int T, int H, int W, int S, const MvfAugmentParams* params,
                      void* workspace, size_t ws_bytes, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MVF_HIP_H_ */
