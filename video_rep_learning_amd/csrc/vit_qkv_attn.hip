// qkv projection FUSED into the self-attention kernel of the frozen ViT blocks (N = 193 .. 208 tokens, head_dim 64):
//     out[f, n, h*64:(h+1)*64] = softmax(q k^T / 8) v,   [q | k | v] = epi(A[f] . W_h^T)      per (frame f, head h)
// A = the block's LayerNorm-1 output h [F*N, D] (plain bias epilogue) or the un-normalised xb = bf16(x) with the LayerNorm folded
// into the epilogue (rstd * (acc - mean * c) + d, as gemm_tc256's LN-fold consumer).  Stands in for timm Attention.qkv + the
// attention core (reached from CARL_MVF/models/transformer.py:188) without the [F*N, 3*D] qkv tensor ever touching HBM.
//
// Why: the training step runs at the board's power cap -- time = joules -- and a byte of HBM traffic costs
// what ~175 bf16 FLOPs cost.  The qkv tensor is 232 MB written by the GEMM and 232 MB read back by the attention kernel per ViT
// block, 464 MB of the block's 2.24 GB.  Here a workgroup computes one head's Q, K, V for one frame (a 208 x 192 x 768 GEMM,
// 61 MFLOP) into LDS images and runs the attention on them; HBM sees A (shared by the frame's 12 heads through L2) and the output.
// Measured at BASELINE configs[1] (256 frames, profiles/r04/qkv_attn_fused.txt): 218 us / 296 mJ per launch against 175 + 74 us /
// 225 + 98 mJ for the qkv GEMM + attention kernel on the same box.
//
// gfx950 design:
//  * ONE (frame, head) unit per 256-thread workgroup, TWO workgroups per CU: a unit's GEMM phase (~35 k cycles, matrix pipe) and
//    attention phase (~20 k cycles, VALU / latency) are complementary, and the CU's second workgroup drifts half a unit out of phase
//    by itself.  (The first form built -- one persistent 8-wave workgroup per CU with the next unit's K tiles prefetched under the
//    attention phase -- ran the two phases back to back: 264 us per launch at the full 2.39 GHz, below the power cap, i.e.
//    schedule-bound; it is gone, its numbers are in the profile file.)
//  * 4 waves as 1 (all 13 token tiles of 16) x 4 (48 of the 192 q|k|v columns each): 156 accumulator registers per lane;
//    v_mfma_f32_16x16x32 with the W fragment as the A operand, so a lane owns 4 consecutive channels of one token (8-byte LDS
//    stores of the results).  The token fragments stream through three registers sets two tiles ahead of their MFMAs, read and
//    waited for by hand (comment at the loop).
//  * K tile = 32 channels (64-byte LDS rows, one MFMA k-step), a ring of THREE operand buffers filled by LDS-DMA
//    (global_load_lds_dwordx4; 16 rows per 1-KiB piece; chunk index XOR-ed with (-(row >> 2)) & 3 on the source address and on the
//    ds_read_b128 address: the four rows that share a 256-byte bank row of the LDS get four different 16-byte slots); one counted
//    vmcnt wait + one s_barrier per K tile.  No look-ahead into another unit: the other workgroup covers the cold start.
//  * epilogue: LayerNorm fold + bias, bf16 (fp16) rounding, 8-byte stores into the Q / K / V images in the layouts
//    vit_attn_tiles.h reads (K, Q: 16-byte chunks XOR (row & 7); V: 32-byte chunks rotated by (row >> 1) & 3).  The images go ON TOP
//    of the operand ring once its last K tile has been read; the unit's (mean, rstd) pairs arrive by LDS-DMA at kernel start behind
//    them (a plain load per token tile here cost 13 exposed L2 round trips per unit).
//  * attention: the stand-alone kernel's split -- query tiles {w, w + 4} then {w + 8, w + 12} -- with its two-tile body
//    (attn_tiles): the same arithmetic in the same order, so the output is BIT-IDENTICAL to qkv GEMM + attention kernel.
//  * grid 8 x ceil(F / 8) x H; XCD x (blockIdx & 7, speed only) gets the frames f = x mod 8, head-minor, so the 12 heads of a
//    frame meet their A rows in that XCD's L2.
// LDS: images 3 x 26 624 (>= ring 3 x 25 600) + (mean, rstd) 1 664 = 81 536 bytes, two workgroups per CU.
#include "common.h"
#include "mvf_hip_internal.h"
#include "gemm_tc_epi.h"
#include "vit_attn_tiles.h"

#include <cstdlib>

namespace {
using namespace vit_attn;
using gemm_tc::mul_rounded;

constexpr int NTOK = 208;                    // token rows of a unit's GEMM tile (13 tiles of 16; rows >= N repeat row N - 1)
constexpr int NCOL = 192;                    // q | k | v columns of one head
constexpr int KE = 32;                       // K tile, elements (64-byte rows)
constexpr int A_BYTES = NTOK * 64, B_BYTES = NCOL * 64, BUF_BYTES = A_BYTES + B_BYTES;
constexpr int RING = 3;
constexpr int NPA = NTOK / 16;                // 1-KiB pieces per K tile: 13 of A, then 12 of W
constexpr int IMG = NTOK * 128;              // one of the Q / K / V images: [token][64] 16-bit, 128-byte rows

struct QkvAttnArgs {
  const char* A;       // [F*N, lda] 16-bit
  const char* W;       // [3*D, D] 16-bit (row-major nn.Linear weight; gamma-folded when ln_c != NULL)
  const float* bias;   // [3*D] (b, or d = b + W beta when folded)
  const float* ln_c;   // [3*D] or NULL
  const float* ln_mr;  // [F*N][2] (mean, rstd) or NULL
  const float* ln_part;  // folded, ln_mr NULL: the producer's partial sums [ln_ns][F*N][2] (sum, sum of squares per 64-column slice)
  int ln_ns;
  float ln_inv_d, ln_eps;
  char* out;           // [F*N, D] 16-bit
  int lda, F, N, H, D;
  float scale_log2;
};

#define WAIT_VMCNT_IMM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define LDS_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

constexpr int OFF_MR = 3 * IMG;  // the unit's (mean, rstd) pairs behind the images (79 872 >= RING * BUF_BYTES = 76 800)
constexpr int LDS_TOTAL = OFF_MR + NTOK * 8;
static_assert(3 * IMG >= RING * BUF_BYTES && 2 * LDS_TOTAL <= 160 * 1024, "LDS");

template <bool F16>
__global__ __launch_bounds__(256, 2) void vit_qkv_attn_kernel(QkvAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, ui = blockIdx.x >> 3;
  const int f = xcd + 8 * (ui / a.H), h = ui % a.H;
  if (f >= a.F) return;                                // (whole workgroup: the grid is 8 x ceil(F / 8) x H)

  // piece p = wave + 4 i of a K tile (25 = 13 A + 12 W pieces of 16 rows x 64 B): i = 0..2 A; i = 3 A for wave 0, W else; i = 4, 5 W;
  // i = 6 W, wave 0 only
  const int prow = lane >> 2, pch = lane & 3;
  unsigned src[7];
  auto a_src = [&](int p) {
    const int r = p * 16 + prow;
    return (unsigned)(f * a.N + min(r, a.N - 1)) * (unsigned)(a.lda * 2) + (unsigned)((pch ^ ((-(r >> 2)) & 3)) * 16);
  };
  auto w_src = [&](int p) {
    const int n = (p - NPA) * 16 + prow;
    return (unsigned)((n >> 6) * a.D + h * HD + (n & 63)) * (unsigned)(a.D * 2) + (unsigned)((pch ^ ((-(n >> 2)) & 3)) * 16);
  };
#pragma unroll
  for (int i = 0; i < 3; ++i) src[i] = a_src(wave + 4 * i);
  src[3] = wave == 0 ? a_src(12) : w_src(wave + 12);
  src[4] = w_src(wave + 16);
  src[5] = w_src(wave + 20);
  src[6] = w_src(24);
  const char* const g3 = wave == 0 ? a.A : a.W;
  const int d3 = wave == 0 ? 12 * 1024 : A_BYTES + (wave + 12 - NPA) * 1024;
  auto issue = [&](int buf, int kt) {
    char* base = smem + buf * BUF_BYTES;
    const unsigned koff = (unsigned)kt * (KE * 2);
#pragma unroll
    for (int i = 0; i < 3; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(a.A + (size_t)(src[i] + koff)), LDS_PTR(base + (wave + 4 * i) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GLB_PTR(g3 + (size_t)(src[3] + koff)), LDS_PTR(base + d3), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GLB_PTR(a.W + (size_t)(src[4] + koff)), LDS_PTR(base + A_BYTES + (wave + 16 - NPA) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GLB_PTR(a.W + (size_t)(src[5] + koff)), LDS_PTR(base + A_BYTES + (wave + 20 - NPA) * 1024), 16, 0, 0);
    if (wave == 0)
      __builtin_amdgcn_global_load_lds(GLB_PTR(a.W + (size_t)(src[6] + koff)), LDS_PTR(base + A_BYTES + (24 - NPA) * 1024), 16, 0, 0);
  };
  const int frag = li * 64 + ((g ^ ((-(li >> 2)) & 3)) << 4);
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
  const int nk = a.D / KE;

  f32x4_t acc[13][3];
#pragma unroll
  for (int i = 0; i < 13; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  issue(0, 0);
  issue(1, 1);
  // the unit's (mean, rstd) pairs into LDS behind the images: by LDS-DMA, or finalized here from the producer's partial sums (one
  // thread per token, the slices in order: ln_stats_finalize_kernel's arithmetic spelled the same way, so the pairs are the bits that
  // kernel would have written) -- the finalize launch between fc2 and this kernel is gone.  Every K tile's wait + barrier lies between
  // these writes and the epilogue's reads.  (A plain load per token tile in the epilogue cost 13 exposed L2 round trips per unit.)
  if (a.ln_part != nullptr) {
    if (tid < NTOK) {
      const float2* p = reinterpret_cast<const float2*>(a.ln_part) + (size_t)f * a.N + min(tid, a.N - 1);
      const size_t rows = (size_t)a.F * a.N;
      float s1 = 0.f, s2 = 0.f;
      for (int sl = 0; sl < a.ln_ns; ++sl) {
        const float2 v = p[sl * rows];
        s1 += v.x;
        s2 += v.y;
      }
      const float mean = s1 * a.ln_inv_d;
      const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * a.ln_inv_d), 0.f);
      *reinterpret_cast<float2*>(smem + OFF_MR + tid * 8) = make_float2(mean, 1.0f / sqrtf(var + a.ln_eps));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  } else if (a.ln_mr != nullptr) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int fi = (wave * 2 + q) * 64 + lane;          // float index into [208][2]; 7 x 64 = 448 >= 416
      if (fi < NTOK * 2)
        __builtin_amdgcn_global_load_lds(GLB_PTR(a.ln_mr + ((size_t)f * a.N + min(fi >> 1, a.N - 1)) * 2 + (fi & 1)),
                                         LDS_PTR(smem + OFF_MR + (wave * 2 + q) * 256), 4, 0, 0);
    }
  }
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk) { if (wave == 0) WAIT_VMCNT_IMM(7); else WAIT_VMCNT_IMM(6); }   // tile t landed, tile t + 1 may be in flight
    else WAIT_VMCNT_IMM(0);
    __builtin_amdgcn_s_barrier();        // every wave's pieces of tile t; and every wave is done reading tile t - 1 (buffer of t + 2)
    if (t + 2 < nk) issue((t + 2) % RING, t + 2);
    // Fragment reads and their waits by hand: A fragments run two token tiles ahead of their MFMAs and each tile waits with a COUNTED
    // lgkmcnt for its own fragment only.  (Left to itself hipcc reads a pair, waits lgkmcnt(0), issues six MFMAs -- every pair's LDS
    // latency exposed, ~900 cycles per K tile beside 624 of matrix work; with plain loads hoisted in the source it still drains the
    // queue with lgkmcnt(0) every third tile.)  The reads are asm so that the compiler's own wait insertion does not count them.
    const unsigned a_rd = lds0 + (t % RING) * BUF_BYTES + frag;
    const unsigned b_rd = a_rd + A_BYTES + wave * 3072;
    bf16x8_t bf[3], af[3];
    LDS_RD128(bf[0], b_rd, 0); LDS_RD128(bf[1], b_rd, 1024); LDS_RD128(bf[2], b_rd, 2048);
    LDS_RD128(af[0], a_rd, 0); LDS_RD128(af[1], a_rd, 1024);
#pragma unroll
    for (int i = 0; i < 13; ++i) {
      if (i + 2 < 13) LDS_RD128(af[(i + 2) % 3], a_rd, (i + 2) * 1024);
      if (i == 0) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(af[0]));
      else if (i < 11) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(af[i % 3]));
      else if (i == 11) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(af[i % 3]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[i % 3]));
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = mfma16x16x32<F16>(bf[j], af[i % 3], acc[i][j]);
    }
  }
  __builtin_amdgcn_s_barrier();          // the ring is dead: the images go on top of it

  {
    const bool fold = a.ln_c != nullptr;
    // the (mean, rstd) pairs of this lane's 13 tokens out of LDS BEFORE the images overwrite anything (they sit behind the images,
    // but a wave may run ahead into the image stores); bias / c of the wave's three column tiles in one batch of loads
    float2 mrv[13];
#pragma unroll
    for (int i = 0; i < 13; ++i)
      mrv[i] = *reinterpret_cast<const float2*>(smem + OFF_MR + (i * 16 + li) * 8);      // unused garbage when not folded
    float4 bvj[3], cvj[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int n0 = (wave * 3 + j) * 16;
      const int col = (n0 >> 6) * a.D + h * HD + (n0 & 63) + 4 * g;      // column of the [3D] bias / c vectors
      bvj[j] = *reinterpret_cast<const float4*>(a.bias + col);
      cvj[j] = fold ? *reinterpret_cast<const float4*>(a.ln_c + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int n0 = (wave * 3 + j) * 16;
      const int sel = n0 >> 6, d0 = (n0 & 63) + 4 * g;
      const float4 bv = bvj[j], cv = cvj[j];
      char* img = smem + sel * IMG;
#pragma unroll
      for (int i = 0; i < 13; ++i) {
        const int tok = i * 16 + li;
        float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
        if (fold) {
          const float2 mr = mrv[i];
          const float nm = -mr.x;
          v0 = mul_rounded(mr.y, fmaf(nm, cv.x, v0)); v1 = mul_rounded(mr.y, fmaf(nm, cv.y, v1));
          v2 = mul_rounded(mr.y, fmaf(nm, cv.z, v2)); v3 = mul_rounded(mr.y, fmaf(nm, cv.w, v3));
        }
        v0 += bv.x; v1 += bv.y; v2 += bv.z; v3 += bv.w;
        const int chunk = d0 >> 3;
        const int sw = sel == 2 ? (((tok >> 1) & 3) << 1) : (tok & 7);
        *reinterpret_cast<uint2*>(img + tok * 128 + ((chunk ^ sw) << 4) + (d0 & 7) * 2) = make_uint2(pack16x2<F16>(v0, v1), pack16x2<F16>(v2, v3));
      }
    }
  }
  __syncthreads();                       // images complete (no LDS-DMA in flight: a plain barrier)

  {
    AttnArgs aa;
    aa.N = a.N; aa.H = a.H; aa.D = a.D; aa.scale_log2 = a.scale_log2; aa.lse = nullptr; aa.npad = 0; aa.rounds = 0; aa.nblk = 1;
    aa.qkv = nullptr; aa.out = nullptr;
    bf16_t* obase = reinterpret_cast<bf16_t*>(a.out) + (size_t)f * a.N * a.D + h * HD;
    const char* sq = smem;
    const int vsw = ((2 * g + (li >> 3)) & 3) << 5;
    auto load_q = [&](bf16x8_t (&qf)[2][2], const int (&qt)[2]) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int qrow = min(qt[i], 12) * 16 + li;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          qf[i][ks] = *reinterpret_cast<const bf16x8_t*>(sq + qrow * 128 + (((ks * 4 + g) ^ (qrow & 7)) << 4));
      }
    };
    bf16x8_t qf[2][2];
    int qt[2] = {wave, wave + 4};
    load_q(qf, qt);
    attn_tiles<13, 2, F16, true>(aa, smem + IMG, smem + 2 * IMG, obase, qf, qt, li, g, vsw, false);
    qt[0] = wave + 8; qt[1] = wave + 12;
    load_q(qf, qt);
    if (wave == 0) attn_tiles<13, 2, F16, true>(aa, smem + IMG, smem + 2 * IMG, obase, qf, qt, li, g, vsw, false);
    else attn_tiles<13, 1, F16, true>(aa, smem + IMG, smem + 2 * IMG, obase, qf, qt, li, g, vsw, false);
  }
}

// ON unless MVF_FUSE_QKV=0 (which keeps the qkv GEMM + attention launches, for A/B measurements)
const bool g_fuse = [] { const char* e = getenv("MVF_FUSE_QKV"); return e == nullptr || atoi(e) != 0; }();

}  // namespace

bool mvf_qkv_attn_supported(int dtype, int F, int N, int H, int D, int lda) {
  return g_fuse && (dtype == MVF_BF16 || dtype == MVF_F16) && N >= 193 && N <= 208 && H > 0 && D == H * HD && D % (KE * RING) == 0 &&
         F >= 1 && lda >= D && (lda * 2) % 16 == 0 && (size_t)F * N * lda * 2 < (1ull << 32) && (size_t)3 * D * D * 2 < (1ull << 32);
}

// A [F*N, lda] . W_h^T (+ LN fold / bias) -> attention -> out [F*N, D].  MVF_ERR_UNSUPPORTED outside the kernel's shape:
// N = 193 .. 208, D = H * 64, D % 96 == 0 (24 K tiles at D = 768: the ring of three runs through unit boundaries), bf16 / fp16.
int mvf_qkv_attn_impl(int dtype, const void* A, int lda, const void* W, const float* bias, const float* ln_c, const float* ln_mr,
                      const float* ln_part, int ln_ns, float ln_eps, void* out, int F, int N, int H, int D, hipStream_t st) {
  if (!(dtype == MVF_BF16 || dtype == MVF_F16) || N < 193 || N > 208 || D != H * HD || D % (KE * RING) != 0 || F < 1 || lda < D ||
      (lda * 2) % 16 != 0 || (size_t)F * N * lda * 2 >= (1ull << 32) || (size_t)3 * D * D * 2 >= (1ull << 32))      // (32-bit offsets)
    return MVF_ERR_UNSUPPORTED;
  // folded (ln_c): the rows' statistics as (mean, rstd) pairs OR as the producer's partial sums, never both; plain: neither
  MVF_CHECK_ARG(A && W && bias && out && A != out && (ln_c != nullptr) == ((ln_mr != nullptr) != (ln_part != nullptr)) &&
                !(ln_mr != nullptr && ln_part != nullptr) && (ln_part == nullptr || (ln_ns >= 1 && ln_ns <= 64 && ln_ns * 64 == D)));
  MVF_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0);
  QkvAttnArgs a;
  a.A = (const char*)A; a.W = (const char*)W; a.bias = bias; a.ln_c = ln_c; a.ln_mr = ln_mr; a.out = (char*)out;
  a.ln_part = ln_part; a.ln_ns = ln_ns; a.ln_inv_d = 1.0f / (float)D; a.ln_eps = ln_eps;
  a.lda = lda; a.F = F; a.N = N; a.H = H; a.D = D;
  a.scale_log2 = LOG2E / 8.0f;
  const int grid = 8 * ((F + 7) / 8) * H;
  static uint64_t attr[2] = {0, 0};      // per device
  if (dtype == MVF_F16) {
    if (mvf_ensure_lds(reinterpret_cast<const void*>(vit_qkv_attn_kernel<true>), LDS_TOTAL, attr[1]) != MVF_OK) return MVF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(vit_qkv_attn_kernel<true>, dim3(grid), dim3(256), LDS_TOTAL, st, a);
  } else {
    if (mvf_ensure_lds(reinterpret_cast<const void*>(vit_qkv_attn_kernel<false>), LDS_TOTAL, attr[0]) != MVF_OK) return MVF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(vit_qkv_attn_kernel<false>, dim3(grid), dim3(256), LDS_TOTAL, st, a);
  }
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// exported for the unit parity test: the fused kernel against mvf_gemm_tc(_ln) + mvf_vit_attn_fwd on the same operands
extern "C" int mvf_vit_qkv_attn_fwd(int dtype, const void* A, int lda, const void* W, const float* bias, const float* ln_c,
                                    const float* ln_mr, const float* ln_part, int ln_ns, float ln_eps, void* out, int F, int N, int H,
                                    int D, hipStream_t st) {
  return mvf_qkv_attn_impl(dtype, A, lda, W, bias, ln_c, ln_mr, ln_part, ln_ns, ln_eps, out, F, N, H, D, st);
}
