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
//
// gfx950 design:
//  * 512 threads = 8 waves as 2 (token rows: 7 | 6 tiles of 16) x 4 (48 of the 192 q|k|v columns each); v_mfma_f32_16x16x32 with the
//    W fragment as the A operand, so a lane owns 4 consecutive channels of one token (8-byte LDS stores of the results)
//  * K tile = 32 channels (64-byte LDS rows, one MFMA k-step), a ring of THREE operand buffers filled by LDS-DMA
//    (global_load_lds_dwordx4; 16 rows per 1-KiB piece; chunk index XOR-ed with (-(row >> 2)) & 3 on the source address and on the
//    ds_read_b128 address: the four rows that share a 256-byte bank row of the LDS get four different 16-byte slots);
//        per K tile t:  [L: fragments of tile t, issue tile t+2]  s_barrier  [C: MFMAs of tile t]  s_barrier
//    with wave row 1 one barrier behind wave row 0, so each SIMD has one wave in its matrix segment while its partner loads (the
//    first form -- one barrier per K tile, all eight waves in step -- took 1.7 k cycles per K tile against 0.62 k of MFMA work:
//    264 us per launch at the full 2.38 GHz, i.e. not even power-bound).  Hazards: the comment at the loop.  24 K tiles per unit
//    and a ring of 3: the ring runs straight through unit boundaries -- tiles 0 / 1 of the NEXT (frame, head) are in flight while
//    this one's attention runs.
//  * epilogue: LayerNorm fold + bias (the tile's bias / c / (mean, rstd) slices arrive by LDS-DMA at the unit's first K tile: a
//    VGPR load here would make hipcc drain the next unit's DMAs), bf16 (fp16) rounding, 8-byte stores into the Q / K / V images in
//    the layouts vit_attn_tiles.h reads (K, Q: 16-byte chunks XOR (row & 7); V: 32-byte chunks rotated by (row >> 1) & 3)
//  * attention: 13 query tiles on 8 waves -- waves 0-4 a pair (w, w + 8), waves 5-7 one -- with the two-tile body of the unfused
//    kernel (attn_tiles): the same arithmetic in the same order, so the output is BIT-IDENTICAL to qkv GEMM + attention kernel
//  * persistent: one workgroup per CU; XCD x (blockIdx & 7, speed only) walks the (frame, head) units of frames f = x mod 8,
//    frame-major, so the 12 heads of a frame meet their A rows in that XCD's L2.
// LDS: ring 3 x 25 600 + images 3 x 26 624 + bias / c / (mean, rstd) 3 200 = 159 872 bytes (one workgroup per CU).
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
constexpr int IMG = NTOK * 128;              // one of the Q / K / V images: [token][64] 16-bit, 128-byte rows
constexpr int OFF_Q = RING * BUF_BYTES, OFF_K = OFF_Q + IMG, OFF_V = OFF_K + IMG;
constexpr int OFF_BIAS = OFF_V + IMG, OFF_C = OFF_BIAS + NCOL * 4, OFF_MR = OFF_C + NCOL * 4;
constexpr int LDS_TOTAL = OFF_MR + NTOK * 8;
static_assert(LDS_TOTAL <= 160 * 1024, "LDS");
constexpr int NPA = NTOK / 16, NPB = NCOL / 16, NPIECE = NPA + NPB;     // 1-KiB pieces per K tile: 13 + 12

struct QkvAttnArgs {
  const char* A;       // [F*N, lda] 16-bit
  const char* W;       // [3*D, D] 16-bit (row-major nn.Linear weight; gamma-folded when ln_c != NULL)
  const float* bias;   // [3*D] (b, or d = b + W beta when folded)
  const float* ln_c;   // [3*D] or NULL
  const float* ln_mr;  // [F*N][2] (mean, rstd) or NULL
  char* out;           // [F*N, D] 16-bit
  int lda, F, N, H, D;
  float scale_log2;
};

#define WAIT_VMCNT_IMM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <bool F16>
__global__ __launch_bounds__(512, 2) void vit_qkv_attn_kernel(QkvAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int li = lane & 15, g = lane >> 4;
  const int nmt = wr ? 6 : 7;                        // this wave row's token tiles (wave-uniform)
  const int mt0 = wr * 7;

  // ---- unit walk: XCD x owns frames x, x + 8, ...; its workgroups take (frame, head) units slot, slot + nslots, ...
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nslots = (gridDim.x - xcd + 7) >> 3;
  const int nfx = a.F > xcd ? (a.F - xcd + 7) >> 3 : 0;
  const int nunits = nfx * a.H;
  int u = slot;
  if (u >= nunits) return;

  // ---- LDS-DMA sources: piece p = wave, wave + 8, ... (< 25) of a K tile = 16 rows x 64 B; lane -> (row 16 p' + lane / 4, physical
  // chunk lane % 4); logical chunk = physical ^ ((-(row >> 2)) & 3)
  const int prow = lane >> 2, pch = lane & 3;
  unsigned src[4];                                   // byte offsets from a.A (A pieces) / a.W (W pieces), K tile 0
  auto set_sources = [&](int unit) {
    const int f = xcd + 8 * (unit / a.H), h = unit % a.H;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = wave + 8 * i;
      if (p < NPA) {
        const int r = p * 16 + prow;
        const int lc = pch ^ ((-(r >> 2)) & 3);
        src[i] = (unsigned)(f * a.N + min(r, a.N - 1)) * (unsigned)(a.lda * 2) + lc * 16;
      } else if (p < NPIECE) {
        const int n = (p - NPA) * 16 + prow;            // 0 .. 191: q | k | v row of W for head h
        const int lc = pch ^ ((-(n >> 2)) & 3);
        src[i] = (unsigned)((n >> 6) * a.D + h * HD + (n & 63)) * (unsigned)(a.D * 2) + lc * 16;
      } else {
        src[i] = 0;
      }
    }
  };
  auto issue = [&](int buf, int kt) {
    char* base = smem + buf * BUF_BYTES;
    const unsigned koff = (unsigned)kt * (KE * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = wave + 8 * i;
      if (p < NPA)
        __builtin_amdgcn_global_load_lds(GLB_PTR(a.A + (size_t)(src[i] + koff)), LDS_PTR(base + p * 1024), 16, 0, 0);
      else if (p < NPIECE)
        __builtin_amdgcn_global_load_lds(GLB_PTR(a.W + (size_t)(src[i] + koff)), LDS_PTR(base + A_BYTES + (p - NPA) * 1024), 16, 0, 0);
    }
  };
  // the unit's bias / c / (mean, rstd) slices, 4 bytes per lane: waves 0-2 bias, 3-5 c, 0-6 the 416 floats of (mean, rstd)
  auto issue_small = [&](int unit) {
    const int f = xcd + 8 * (unit / a.H), h = unit % a.H;
    if (wave < 3) {
      const int n = wave * 64 + lane;
      __builtin_amdgcn_global_load_lds(GLB_PTR(a.bias + (n >> 6) * a.D + h * HD + (n & 63)), LDS_PTR(smem + OFF_BIAS + wave * 256), 4, 0, 0);
    } else if (wave < 6 && a.ln_c != nullptr) {
      const int n = (wave - 3) * 64 + lane;
      __builtin_amdgcn_global_load_lds(GLB_PTR(a.ln_c + (n >> 6) * a.D + h * HD + (n & 63)), LDS_PTR(smem + OFF_C + (wave - 3) * 256), 4, 0, 0);
    }
    if (wave < 7 && a.ln_mr != nullptr) {
      const int fi = wave * 64 + lane;                  // float index into [208][2]
      const int row = min(fi >> 1, a.N - 1);
      if (fi < NTOK * 2)
        __builtin_amdgcn_global_load_lds(GLB_PTR(a.ln_mr + ((size_t)f * a.N + row) * 2 + (fi & 1)), LDS_PTR(smem + OFF_MR + wave * 256), 4, 0, 0);
    }
  };

  // fragment read offsets inside a buffer: lane (row li of a 16-row tile, k chunk g)
  const int frag = li * 64 + ((g ^ ((-(li >> 2)) & 3)) << 4);
  const int nk = a.D / KE;                            // 24 at D = 768 (a multiple of RING: checked by the launch)

  set_sources(u);
  issue(0, 0);
  issue(1, 1);
  if (wave == 0) WAIT_VMCNT_IMM(4); else WAIT_VMCNT_IMM(3);     // tile 0 landed (this wave's pieces; tile 1 stays in flight)
  __builtin_amdgcn_s_barrier();
  for (;;) {
    const int unext = u + nslots;
    const bool have_next = unext < nunits;
    f32x4_t acc[7][3];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // Two slots per K tile, [L: fragment reads of tile t + DMA issue of tile t + 2] barrier [C: the MFMAs of tile t] barrier, with wave
    // row 1 running ONE barrier behind wave row 0: the two waves of a SIMD alternate, one in its matrix segment while its partner
    // loads.  Every wave waits for its own pieces of tile t + 1 before the barrier that opens wave row 0's L(t + 1) -- wave row 0 at
    // the end of C(t), wave row 1 at the end of L(t) -- so that tile is complete for both rows' reads; tile t + 2 stays in flight.
    // The buffer tile t + 2 lands in held tile t - 1, whose last reads (wave row 1's L(t - 1)) are retired (lgkmcnt(0)) before the
    // barrier in front of wave row 0's L(t), the first slot that issues into it.
    if (wr == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nk; ++t) {
      // ---- L(t)
      const char* ab = smem + (t % RING) * BUF_BYTES;
      const char* bb = ab + A_BYTES;
      bf16x8_t bf[3], af[7];
#pragma unroll
      for (int j = 0; j < 3; ++j) bf[j] = *reinterpret_cast<const bf16x8_t*>(bb + (wc * 3 + j) * 1024 + frag);
#pragma unroll
      for (int i = 0; i < 7; ++i)
        if (i < nmt) af[i] = *reinterpret_cast<const bf16x8_t*>(ab + (mt0 + i) * 1024 + frag);
      if (t == 0) issue_small(u);                     // (every wave is past the previous unit's epilogue, the regions' last reader)
      const bool more = t + 2 < nk || have_next;      // a tile t + 2 exists (this unit's, or tile t + 2 - nk of the next one)
      if (t + 2 < nk) {
        issue((t + 2) % RING, t + 2);
      } else if (have_next) {
        if (t + 2 == nk) set_sources(unext);          // this unit's last tile was issued two iterations ago
        issue((t + 2) % RING, t + 2 - nk);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (wr == 1) {                                  // tile t + 1 (if any) landed; tile t + 2 may stay in flight
        if (more) WAIT_VMCNT_IMM(3); else WAIT_VMCNT_IMM(0);     // (waves 4-7 issue three pieces per K tile)
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- C(t)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 7; ++i)
        if (i < nmt) {
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = mfma16x16x32<F16>(bf[j], af[i], acc[i][j]);
        }
      __builtin_amdgcn_s_setprio(0);
      if (wr == 0) {
        if (more) { if (wave == 0) WAIT_VMCNT_IMM(4); else WAIT_VMCNT_IMM(3); } else WAIT_VMCNT_IMM(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();        // re-align the wave rows: both write the images in the same slot

    // ---- epilogue: LN fold + bias, rounding, Q / K / V images ----
    {
      const float* sbias = reinterpret_cast<const float*>(smem + OFF_BIAS);
      const float* sc = reinterpret_cast<const float*>(smem + OFF_C);
      const float* smr = reinterpret_cast<const float*>(smem + OFF_MR);
      const bool fold = a.ln_c != nullptr;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int n0 = (wc * 3 + j) * 16;               // 0 .. 176: which of q | k | v, and the channel inside it
        const int sel = n0 >> 6, d0 = (n0 & 63) + 4 * g;
        const float4 bv = *reinterpret_cast<const float4*>(sbias + n0 + 4 * g);
        float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fold) cv = *reinterpret_cast<const float4*>(sc + n0 + 4 * g);
        char* img = smem + (sel == 0 ? OFF_Q : (sel == 1 ? OFF_K : OFF_V));
#pragma unroll
        for (int i = 0; i < 7; ++i)
          if (i < nmt) {
            const int tok = (mt0 + i) * 16 + li;
            float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
            if (fold) {
              const float2 mr = *reinterpret_cast<const float2*>(smr + tok * 2);
              const float nm = -mr.x;
              v0 = mul_rounded(mr.y, fmaf(nm, cv.x, v0)); v1 = mul_rounded(mr.y, fmaf(nm, cv.y, v1));
              v2 = mul_rounded(mr.y, fmaf(nm, cv.z, v2)); v3 = mul_rounded(mr.y, fmaf(nm, cv.w, v3));
            }
            v0 += bv.x; v1 += bv.y; v2 += bv.z; v3 += bv.w;
            const int chunk = d0 >> 3;
            const int sw = sel == 2 ? (((tok >> 1) & 3) << 1) : (tok & 7);
            *reinterpret_cast<uint2*>(img + tok * 128 + ((chunk ^ sw) << 4) + (d0 & 7) * 2) =
                make_uint2(pack16x2<F16>(v0, v1), pack16x2<F16>(v2, v3));
          }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                     // images complete

    // ---- attention on the images: waves 0-4 query tiles (w, w + 8), waves 5-7 tile w ----
    {
      AttnArgs aa;
      aa.N = a.N; aa.H = a.H; aa.D = a.D; aa.scale_log2 = a.scale_log2; aa.lse = nullptr; aa.npad = 0; aa.rounds = 0; aa.nblk = 1;
      aa.qkv = nullptr; aa.out = nullptr;
      const int f = xcd + 8 * (u / a.H), h = u % a.H;
      bf16_t* obase = reinterpret_cast<bf16_t*>(a.out) + (size_t)f * a.N * a.D + h * HD;
      const char* sq = smem + OFF_Q;
      const int vsw = ((2 * g + (li >> 3)) & 3) << 5;
      const int qt[2] = {wave, wave + 8};
      bf16x8_t qf[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int qrow = min(qt[i], 12) * 16 + li;      // (waves 5-7 have no second tile: any valid row)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          qf[i][ks] = *reinterpret_cast<const bf16x8_t*>(sq + qrow * 128 + (((ks * 4 + g) ^ (qrow & 7)) << 4));
      }
      if (wave < 5) attn_tiles<13, 2, F16, true>(aa, smem + OFF_K, smem + OFF_V, obase, qf, qt, li, g, vsw, false);
      else attn_tiles<13, 1, F16, true>(aa, smem + OFF_K, smem + OFF_V, obase, qf, qt, li, g, vsw, false);
    }
    if (!have_next) break;
    u = unext;
  }
}

// OFF by default (MVF_FUSE_QKV=1 routes the backbone through it): measured sustained at BASELINE configs[1] (256 frames, profiles/r04/
// qkv_attn_fused.txt) the launch takes 264 us (one barrier per K tile, eight waves in step) / 286 us (this staggered form) against
// 171 + 72 us for the qkv GEMM + attention kernel on the same box, and the training step 11.46 against 11.13 ms.  It is correct (bit for
// bit, tests/test_gpu_kernels.py) and moves 464 MB less per block, but it runs at the full 2.39 GHz BELOW the power cap, i.e. it is
// schedule-bound: a unit's GEMM phase (~35 k cycles, matrix pipe) and attention phase (~20 k cycles, VALU / latency at two waves per
// SIMD) run one after the other in the CU's only workgroup, where the stand-alone attention kernel overlaps three workgroups per CU
// (13 k CU-cycles per unit).  The two phases are complementary -- what would pay is a second co-resident workgroup half a unit out of
// phase, which the 78 KB of Q / K / V images per workgroup do not leave room for.
const bool g_fuse = [] { const char* e = getenv("MVF_FUSE_QKV"); return e != nullptr && e[0] == '1'; }();

}  // namespace

bool mvf_qkv_attn_supported(int dtype, int F, int N, int H, int D, int lda) {
  return g_fuse && (dtype == MVF_BF16 || dtype == MVF_F16) && N >= 193 && N <= 208 && H > 0 && D == H * HD && D % (KE * RING) == 0 &&
         F >= 1 && lda >= D && (lda * 2) % 16 == 0 && (size_t)F * N * lda * 2 < (1ull << 32) && (size_t)3 * D * D * 2 < (1ull << 32);
}

// A [F*N, lda] . W_h^T (+ LN fold / bias) -> attention -> out [F*N, D].  MVF_ERR_UNSUPPORTED outside the kernel's shape:
// N = 193 .. 208, D = H * 64, D % 96 == 0 (24 K tiles at D = 768: the ring of three runs through unit boundaries), bf16 / fp16.
int mvf_qkv_attn_impl(int dtype, const void* A, int lda, const void* W, const float* bias, const float* ln_c, const float* ln_mr,
                      void* out, int F, int N, int H, int D, hipStream_t st) {
  if (!(dtype == MVF_BF16 || dtype == MVF_F16) || N < 193 || N > 208 || D != H * HD || D % (KE * RING) != 0 || F < 1 || lda < D ||
      (lda * 2) % 16 != 0 || (size_t)F * N * lda * 2 >= (1ull << 32) || (size_t)3 * D * D * 2 >= (1ull << 32))      // (32-bit offsets)
    return MVF_ERR_UNSUPPORTED;
  MVF_CHECK_ARG(A && W && bias && out && A != out && (ln_c == nullptr) == (ln_mr == nullptr));
  MVF_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0);
  QkvAttnArgs a;
  a.A = (const char*)A; a.W = (const char*)W; a.bias = bias; a.ln_c = ln_c; a.ln_mr = ln_mr; a.out = (char*)out;
  a.lda = lda; a.F = F; a.N = N; a.H = H; a.D = D;
  a.scale_log2 = LOG2E / 8.0f;
  int wgs = 256;
  (void)mvf_gemm_tc_get_wgs(&wgs);                     // one workgroup per CU of the persistent kernels' budget (RCCL reserve, CU-masked streams)
  const int grid = std::max(8, std::min(wgs & ~7, ((F * H + 7) / 8) * 8));
  static bool attr[2] = {false, false};
  if (dtype == MVF_F16) {
    if (!attr[1]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vit_qkv_attn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL); attr[1] = true; }
    hipLaunchKernelGGL(vit_qkv_attn_kernel<true>, dim3(grid), dim3(512), LDS_TOTAL, st, a);
  } else {
    if (!attr[0]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vit_qkv_attn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL); attr[0] = true; }
    hipLaunchKernelGGL(vit_qkv_attn_kernel<false>, dim3(grid), dim3(512), LDS_TOTAL, st, a);
  }
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// exported for the unit parity test: the fused kernel against mvf_gemm_tc(_ln) + mvf_vit_attn_fwd on the same operands
extern "C" int mvf_vit_qkv_attn_fwd(int dtype, const void* A, int lda, const void* W, const float* bias, const float* ln_c,
                                    const float* ln_mr, void* out, int F, int N, int H, int D, hipStream_t st) {
  return mvf_qkv_attn_impl(dtype, A, lda, W, bias, ln_c, ln_mr, out, F, N, H, D, st);
}
