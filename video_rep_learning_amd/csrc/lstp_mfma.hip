// One-pass LSTP pooling on the matrix cores (bf16 taps; the VALU forms of lstp_pool.hip stay for fp32 taps and other shapes).
// Stands in for LSTPCrossAtt.forward / LearnableTokenPooling (CARL_MVF/models/mvformer.py:243-266, 352-414) by the same exact
// rewrite as lstp_pool.hip: scores = X (W_K^T q), pooled = softmax(scores) X.
//
// Why: the one-pass VALU kernels are instruction-bound, not memory-bound -- a frame's 196 x 2304 tap values meet 3 queries
// twice (score dot products, weighted sums): 2.7 M lane-FMAs per CU and frame plus their bf16 unpacking, 134 us forward for
// 231 MB (1.7 TB/s).  Both products are GEMMs with one skinny side:
//   S  [16 tokens x 16 q]   = X  [16 tokens x C]  . vec^T [C x 16 q]          v_mfma_f32_16x16x32_bf16, K = channels
//   O^T[C x 16 q]          += X^T[C x 16 tokens]  . P     [16 tokens x 16 q]  v_mfma_f32_16x16x16_bf16, K = tokens
// with q padded from nq <= 3 to the 16 columns of a tile (13/16 of the matrix work is padding; it is still 1 / 20 of the VALU
// time).  The small operands (vec, P -- fp32 quantities) enter as bf16 hi + lo pairs, two MFMAs per product, so the result
// carries fp32-level error (2^-17 relative on the small operand); X is used exactly as stored.
//
// One workgroup (8 waves) per frame, token tiles of 16:
//   * the tile's 16 x C bf16 values (72 KB at C = 2304) are loaded into registers one tile ahead (KS 16-byte loads per thread)
//     and written to ONE LDS image with a 16-byte row pad (conflict-free ds_read_b128 fragments)
//   * wave w owns channels [w C/8, (w+1) C/8): its K range of the score product and its rows of O^T (2 KS accumulator tiles);
//     its hi / lo vector fragments are read from their LDS image once and stay in registers (C <= 2304)
//   * the eight partial score tiles meet in LDS (fixed order: deterministic); every wave then runs the same online-softmax
//     update on the same numbers, and its own score accumulator registers ARE the B operand of the second product
//     (accumulator layout D[4g + r][q] = B layout k = 4g + r, n = q): P never goes through LDS
//   * X^T fragments come from the same LDS image through the transposing read ds_read_b64_tr_b16
// Backward (d loss / d vec from dpooled, P, pooled): the same two products with g = X dpooled^T in place of the scores and
// P * g in place of the softmax weights (lstp_pool.hip header, "backward").
//
// Measured at configs[1] size (256 frames x 196 tokens x 2304 channels, 3 queries), inside the training step (serial rocprofv3):
// forward 134 -> 60.6 us, backward 60.9 -> 55.3 us; on cold taps (tools/lstp_bench.py) 134 -> 101 / 108 -> 106 us.  It is still
// not memory-bound (231 MB in 60 us = 3.8 TB/s): one workgroup of two waves per SIMD walks 13 tiles of four barrier-separated
// phases each (PMC: matrix pipe 9 % busy, waves waiting 59 % of their cycles); what the registers leave no room for is a second
// workgroup per CU or two tiles per barrier round.
#include "common.h"
#include "mvf_hip_internal.h"

namespace {

// workgroup barrier for LDS traffic only: __syncthreads() also waits for vmcnt(0), i.e. for the NEXT tile's global loads, which
// are meant to stay in flight across the tile's three barriers
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int LW = 8;             // waves per workgroup
constexpr int QP = 16;            // query columns of a tile

struct LstpMfmaArgs {
  const bf16_t* taps[3];
  int N, T, nq, per_frame;
  const float* vec;       // fwd: query-side vectors [nq, C] or [Bc, nq, T, C]; bwd: dpooled [Bc, nq, T, C]
  float* P;               // [F, nq, N]  (fwd: out, bwd: in)
  float* pooled;          // [Bc, nq, T, C]  (fwd: out, bwd: in)
  float* G;               // bwd: [Bc, nq, T, C]
  float inv_sqrt_d;
};

__device__ __forceinline__ void split_bf16(float v, bf16_t& hi, bf16_t& lo) {
  hi = f32_to_bf16(v);
  lo = f32_to_bf16(v - bf16_to_f32(hi));
}

// VR: rows of the hi / lo vector images = queries + at least one zero row (the lanes of a tile's padding columns read the last one):
// 4 for nq <= 3; 8 for nq <= 7 (6 entities: fg99_mvf.yml, BASELINE configs[2]) -- the 8-row images are as large as the token image and lie
// ON it: they are read into registers once, before tile 0 is stored (C <= 2304 only, where the fragments live in registers)
template <int NT, int DD, int VR = 4>
struct LstpShape {
  static constexpr int C = NT * DD;
  static constexpr int KS = C / 256;                 // 32-channel k-steps per wave = 16-byte chunks per thread and tile
  static constexpr int CW = C / LW;                  // channels per wave
  static constexpr int PITCH = 2 * C + 16;           // bytes per token row of the LDS image
  static constexpr int VP = 2 * C + 16;              // bytes per query row of the hi / lo vector images
  static constexpr int X_BYTES = 16 * PITCH;
  static constexpr int V_BYTES = 2 * VR * VP;        // hi | lo, rows 0 .. VR-1 (rows nq .. VR-1 zero: the padding columns read the last)
  static constexpr bool ALIAS = VR > 4;              // the vector images share the token image's bytes
  static_assert(!ALIAS || (V_BYTES <= X_BYTES && KS <= 9), "aliased vector images: as large as the token image at most, fragments in registers");
  static constexpr int RED_BYTES = LW * QP * 16 * 4;
  static size_t lds_bytes(int nq, int N) { return X_BYTES + (ALIAS ? 0 : V_BYTES) + RED_BYTES + (size_t)nq * N * 4 + 2 * QP * 4; }
};

template <bool BWD, int NT, int DD, int VR = 4>
__global__ __launch_bounds__(512) void lstp_mfma_kernel(LstpMfmaArgs a) {
  using SH = LstpShape<NT, DD, VR>;
  constexpr bool ALIAS = SH::ALIAS;
  constexpr int C = SH::C, KS = SH::KS, CW = SH::CW, PITCH = SH::PITCH, VP = SH::VP;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* xt = sm;                                                   // [16][PITCH]
  char* svh = ALIAS ? sm : sm + SH::X_BYTES;                       // [VR][VP] hi
  char* svl = svh + VR * VP;                                       // [VR][VP] lo
  float* sred = reinterpret_cast<float*>(sm + SH::X_BYTES + (ALIAS ? 0 : SH::V_BYTES));   // [LW][QP][16]
  float* ssc = sred + LW * QP * 16;                                // [nq][N]: fwd raw scores, bwd P
  float* sml = ssc + a.nq * a.N;                                   // [2][QP]: fwd (M, L)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int f = blockIdx.x, b = f / a.T, t = f % a.T, N = a.N, nq = a.nq;

  // ---- tile loads.  Per tap a tile is 16 rows x DD / 8 chunks of 16 bytes = KT x 512: thread tid takes chunks k * 512 + tid of
  // EVERY tap (same row / column for the three taps), so a load is a uniform base (SGPRs: the tap's rows of this frame) + one
  // 32-bit offset per k, and the staging arrays are plain ext-vector locals (as HIP uint4 -- a class type -- hipcc kept them in
  // scratch memory: 208 us instead of 65) ----
  constexpr int KT = DD / 256;
  int rowk[KT], colk[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const int idx = k * 512 + tid;
    rowk[k] = idx / (DD / 8);
    colk[k] = idx - rowk[k] * (DD / 8);
  }
  const char* tapb[NT];
#pragma unroll
  for (int tp = 0; tp < NT; ++tp) tapb[tp] = reinterpret_cast<const char*>(a.taps[tp] + (size_t)f * N * DD);
  // One tile of look-ahead.  Two (a second register set, tiles t + 1 and t + 2 in flight) measured 60.5 / 57.4 us against
  // 63.6 / 58.3: not what the tile time is made of, and it needs the registers the vector fragments now live in.
  constexpr bool VREG = KS <= 9;      // the vector fragments live in registers (read from their LDS image once); C = 3072: in LDS
  u32x4_t stage_a[KS];
// the taps are read once per pass (231 MB): non-temporal, so that the stream does not push the backbone GEMMs' operands (running
  // beside this kernel on the other streams) out of the L2 -- MVF_NT_OFF (build flag): plain loads, for A/B measurements
#ifdef MVF_NT_OFF
#define LSTP_TAP_LOAD(p) (*(p))
#else
#define LSTP_TAP_LOAD(p) __builtin_nontemporal_load(p)
#endif
#define LSTP_LOAD_TILE(TILE, SET)                                                                              \
  _Pragma("unroll") for (int k = 0; k < KT; ++k) {                                                            \
    /* rows past the frame repeat its last token (their weights are 0) */                                     \
    const unsigned voff = (unsigned)(min((TILE) * 16 + rowk[k], N - 1) * DD + colk[k] * 8) * 2u;              \
    _Pragma("unroll") for (int tp = 0; tp < NT; ++tp)                                                         \
      SET[tp * KT + k] = LSTP_TAP_LOAD(reinterpret_cast<const u32x4_t*>(tapb[tp] + voff));                    \
  }
#define LSTP_STORE_TILE(SET)                                                                                   \
  _Pragma("unroll") for (int k = 0; k < KT; ++k)                                                              \
    _Pragma("unroll") for (int tp = 0; tp < NT; ++tp)                                                         \
      *reinterpret_cast<u32x4_t*>(xt + rowk[k] * PITCH + (tp * (DD / 8) + colk[k]) * 16) = SET[tp * KT + k];

  f32x4_t acc[2 * KS];
#pragma unroll
  for (int ct = 0; ct < 2 * KS; ++ct) acc[ct] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m = -1e30f, l = 0.f, gb = 0.f;      // online softmax state of query li (every wave, every g: the same numbers) / gbar
  const int vrow = min(li, VR - 1);         // padding columns read the zero row
  const int ntiles = (N + 15) >> 4;
  LSTP_LOAD_TILE(0, stage_a)
  // ---- this frame's small operand as bf16 hi / lo rows (rows nq .. 3 zero).  A compile-time trip count: all of a thread's
  // loads are in flight together (as a run-time loop this prologue was 18 dependent L2 round trips) ----
  {
    constexpr int NV = VR * C / 512;
    float vv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = k * 512 + tid, j = i / C, c = i - j * C;
      vv[k] = 0.f;
      if (j < nq) vv[k] = (BWD || a.per_frame) ? a.vec[(((size_t)b * nq + j) * a.T + t) * C + c] : a.vec[(size_t)j * C + c];
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = k * 512 + tid, j = i / C, c = i - j * C;
      bf16_t hi, lo;
      split_bf16(vv[k], hi, lo);
      *reinterpret_cast<bf16_t*>(svh + j * VP + c * 2) = hi;
      *reinterpret_cast<bf16_t*>(svl + j * VP + c * 2) = lo;
    }
  }
  if constexpr (BWD)
    for (int i = tid; i < nq * N; i += 512) ssc[i] = a.P[(size_t)f * nq * N + i];

  if constexpr (!ALIAS) LSTP_STORE_TILE(stage_a)
  __syncthreads();
  bf16x8_t vfh[VREG ? KS : 1], vfl[VREG ? KS : 1];
  if constexpr (VREG) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cb = (wave * CW + ks * 32 + 8 * g) * 2;
      vfh[ks] = *reinterpret_cast<const bf16x8_t*>(svh + vrow * VP + cb);
      vfl[ks] = *reinterpret_cast<const bf16x8_t*>(svl + vrow * VP + cb);
    }
  }
  if constexpr (ALIAS) {      // the vector images are in registers now: tile 0 takes their place
    __syncthreads();
    LSTP_STORE_TILE(stage_a)
    __syncthreads();
  }
  // one tile's arithmetic on the LDS image (both products, the exchange barrier in between)
  auto compute = [&](int tile) __attribute__((always_inline)) {
    // ---- partial scores over this wave's channels: D[r] = sum_c X[token 4g + r][c] vec[q = li][c] ----
    f32x4_t d = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cb = (wave * CW + ks * 32 + 8 * g) * 2;
      const bf16x8_t xa = *reinterpret_cast<const bf16x8_t*>(xt + li * PITCH + cb);
      bf16x8_t vh, vl;
      if constexpr (VREG) {
        vh = vfh[ks]; vl = vfl[ks];
      } else {
        vh = *reinterpret_cast<const bf16x8_t*>(svh + vrow * VP + cb);
        vl = *reinterpret_cast<const bf16x8_t*>(svl + vrow * VP + cb);
      }
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, vh, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, vl, d, 0, 0, 0);
    }
    *reinterpret_cast<f32x4_t*>(sred + (wave * QP + li) * 16 + 4 * g) = d;
    LDS_BARRIER();
    f32x4_t s = *reinterpret_cast<const f32x4_t*>(sred + li * 16 + 4 * g);
#pragma unroll
    for (int w = 1; w < LW; ++w) {
      const f32x4_t p = *reinterpret_cast<const f32x4_t*>(sred + (w * QP + li) * 16 + 4 * g);
      s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; s[3] += p[3];
    }
    // ---- weights of the 16 tokens for query li: softmax numerators (forward) / P g (backward) ----
    float wgt[4];
    const int n0 = tile * 16 + 4 * g;
    if constexpr (!BWD) {
      float tm = -1e30f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[r] = (n0 + r < N && li < nq) ? s[r] * a.inv_sqrt_d : -1e30f;
        tm = fmaxf(tm, s[r]);
      }
      if (wave == 0 && li < nq)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n0 + r < N) ssc[li * N + n0 + r] = s[r];
      tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
      const float mn = fmaxf(m, tm);
      const float al = __expf(m - mn);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        wgt[r] = (n0 + r < N && li < nq) ? __expf(s[r] - mn) : 0.f;
        ps += wgt[r];
      }
      ps += __shfl_xor(ps, 16, 64);
      ps += __shfl_xor(ps, 32, 64);
      l = l * al + ps;
      m = mn;
#pragma unroll
      for (int ct = 0; ct < 2 * KS; ++ct) { acc[ct][0] *= al; acc[ct][1] *= al; acc[ct][2] *= al; acc[ct][3] *= al; }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        wgt[r] = (n0 + r < N && li < nq) ? ssc[li * N + n0 + r] * s[r] : 0.f;
        gb += wgt[r];
      }
    }
    bf16_t h[4], lo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) split_bf16(wgt[r], h[r], lo[r]);
    const bf16x4_t bh = {(short)h[0], (short)h[1], (short)h[2], (short)h[3]};
    const bf16x4_t bl = {(short)lo[0], (short)lo[1], (short)lo[2], (short)lo[3]};
    // ---- O^T[c][q] += sum_tokens X[token][c] w[token][q] over this wave's channels ----
#pragma unroll
    for (int ct = 0; ct < 2 * KS; ++ct) {
      const char* p = xt + (4 * g + (li >> 2)) * PITCH + (wave * CW + ct * 16) * 2 + 8 * (li & 3);
      const bf16x4_t xa = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(p));
      acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xa, bh, acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xa, bl, acc[ct], 0, 0, 0);
    }
  };
  // The look-ahead load is issued UNCONDITIONALLY (a tile index past the frame re-reads its last row: the row clamp makes any
  // index safe): behind a branch hipcc's wait-count pass assumes the worst on the merged path.
  for (int tile = 0; tile < ntiles; ++tile) {
    LSTP_LOAD_TILE(tile + 1, stage_a)
    compute(tile);
    LDS_BARRIER();                          // every wave is done with the image (and with the partial-score slots)
    LSTP_STORE_TILE(stage_a)
    LDS_BARRIER();
  }
  // ---- results: lane (li = q < nq, g) holds O^T[c = wave CW + ct 16 + 4g + r][q] ----
  if constexpr (!BWD) {
    if (li < nq) {
      const float inv = 1.0f / l;
      float* ob = a.pooled + (((size_t)b * nq + li) * a.T + t) * C + wave * CW + 4 * g;
#pragma unroll
      for (int ct = 0; ct < 2 * KS; ++ct)
        *reinterpret_cast<float4*>(ob + ct * 16) = make_float4(acc[ct][0] * inv, acc[ct][1] * inv, acc[ct][2] * inv, acc[ct][3] * inv);
      if (wave == 0 && g == 0) { sml[li] = m; sml[QP + li] = l; }
    }
    __syncthreads();
    for (int i = tid; i < nq * N; i += 512) {
      const int j = i / N;
      a.P[(size_t)f * nq * N + i] = __expf(ssc[i] - sml[j]) / sml[QP + j];
    }
  } else {
    gb += __shfl_xor(gb, 16, 64);
    gb += __shfl_xor(gb, 32, 64);
    if (li < nq) {
      const size_t o = (((size_t)b * nq + li) * a.T + t) * C + wave * CW + 4 * g;
#pragma unroll
      for (int ct = 0; ct < 2 * KS; ++ct) {
        const float4 pl = *reinterpret_cast<const float4*>(a.pooled + o + ct * 16);
        *reinterpret_cast<float4*>(a.G + o + ct * 16) =
            make_float4(a.inv_sqrt_d * (acc[ct][0] - gb * pl.x), a.inv_sqrt_d * (acc[ct][1] - gb * pl.y),
                        a.inv_sqrt_d * (acc[ct][2] - gb * pl.z), a.inv_sqrt_d * (acc[ct][3] - gb * pl.w));
      }
    }
  }
}

template <bool BWD, int NT, int DD, int VR = 4>
int go(const LstpMfmaArgs& a, int F, hipStream_t st) {
  using SH = LstpShape<NT, DD, VR>;
  const size_t lds = SH::lds_bytes(a.nq, a.N);
  if (lds > 160 * 1024) return MVF_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(lstp_mfma_kernel<BWD, NT, DD, VR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return MVF_ERR_ARG;
    attr_set = true;
  }
  hipLaunchKernelGGL((lstp_mfma_kernel<BWD, NT, DD, VR>), dim3(F), dim3(512), lds, st, a);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

}  // namespace

// bf16 taps, 1 or 3 of them, D = 768 or 1024, nq <= 7 (nq <= 3 at 3 x 1 024 channels; the lanes of a tile's padding columns read a zero
// row of the 4- or 8-row vector image); MVF_ERR_UNSUPPORTED otherwise -- the caller (lstp_pool.hip) then takes the VALU form
int mvf_lstp_mfma_impl(bool bwd, const void* const* taps, int n_taps, int D, int F, int N, int T, int nq, const float* vec,
                       int per_frame, float inv_sqrt_d, float* P, float* pooled, float* G, hipStream_t st) {
  if (!(n_taps == 1 || n_taps == 3) || !(D == 768 || D == 1024) || nq < 1 || nq > 7 || N < 1) return MVF_ERR_UNSUPPORTED;
  if (nq > 3 && n_taps == 3 && D == 1024) return MVF_ERR_UNSUPPORTED;      // (8-row vector images need the fragments in registers: C <= 2304)
  LstpMfmaArgs a{};
  for (int i = 0; i < n_taps; ++i) a.taps[i] = reinterpret_cast<const bf16_t*>(taps[i]);
  a.N = N; a.T = T; a.nq = nq; a.per_frame = per_frame; a.vec = vec; a.P = P; a.pooled = pooled; a.G = G; a.inv_sqrt_d = inv_sqrt_d;
  if (nq > 3) {       // 4 .. 7 queries: 8-row vector images
    if (n_taps == 3 && D == 768) return bwd ? go<true, 3, 768, 8>(a, F, st) : go<false, 3, 768, 8>(a, F, st);
    if (n_taps == 1 && D == 768) return bwd ? go<true, 1, 768, 8>(a, F, st) : go<false, 1, 768, 8>(a, F, st);
    return bwd ? go<true, 1, 1024, 8>(a, F, st) : go<false, 1, 1024, 8>(a, F, st);
  }
  if (n_taps == 3 && D == 768) return bwd ? go<true, 3, 768>(a, F, st) : go<false, 3, 768>(a, F, st);
  if (n_taps == 1 && D == 768) return bwd ? go<true, 1, 768>(a, F, st) : go<false, 1, 768>(a, F, st);
  if (n_taps == 3 && D == 1024) return bwd ? go<true, 3, 1024>(a, F, st) : go<false, 3, 1024>(a, F, st);
  return bwd ? go<true, 1, 1024>(a, F, st) : go<false, 1, 1024>(a, F, st);
}
