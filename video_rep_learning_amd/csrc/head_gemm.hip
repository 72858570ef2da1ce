// fp32 MFMA GEMM with generic strides for the (small, latency-bound) trainable head:
//     C[m, n] (+)= [resid +] dropout( act( alpha * sum_k A(m,k) * B(k,n) + bias[n] + table[idx(m), n] ) )
// A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]: one kernel serves forward (x . W^T), input
// gradient (dy . W) and weight gradient (dy^T . x) of every nn.Linear on the path
// (CARL_MVF/models/mvformer.py:77,86,97; models/utils.py:65-68,182-183; resnet_c2d.py:117-120) by
// passing different strides -- no transposed copies.
//
// Fusions (each removes a launch and an HBM/L2 round trip of a [768, <=1024] fp32 tensor):
//   epilogue: bias, sin/cos PE table, ReLU, dropout, residual add -- the residual connection
//             `x + drop(sub(LN(x)))` of models/utils.py:153-159 ends in the epilogue of the sub-layer's last Linear;
//   rowsum  : out[m] (+)= sum_k A(m,k) -- in the weight-gradient problem A = dy^T, so this IS the bias gradient;
//   mvf_hlinear_bwd: dX, dW and db of one Linear in ONE launch (two tile ranges of one grid), parameter gradients
//             accumulated in place (flat gradient buffer).
//   (Tried and dropped: applying the ReLU / dropout backward mask to dy while loading it -- every output tile re-hashes
//   or re-loads the mask, 2.5x slower than one 5-us elementwise pass.)
//
// gfx950 design: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain), 64x64 output tile per workgroup (4 waves x 32x32),
// 64-deep k steps.  Whatever the operand's layout -- k contiguous (x, W in the forward), or k strided with the tile
// index contiguous (dy and x in dW = dy^T x, W in dX = dy W) -- its [64 k][64 rows] tile is fetched with coalesced
// 16-byte loads along the CONTIGUOUS direction and put into LDS k-major ([k][row]), from where
// each lane reads the one element per MFMA the 16x16x4 layout wants (lane (r, g): row r, k = 4q + g): scalar LDS
// reads, conflict-free (row stride 80 floats + an XOR swizzle of the 4-float column groups).  Two LDS buffers; the next tile's global loads are issued before the current tile's 64 MFMAs
// and written to the other buffer after them: one barrier per k step, global latency hidden under the matrix pipe.
// The matrices are a few MB and L2-resident and M is 768 rows, so this is latency- not bandwidth-bound work; operands
// swapped in the MFMA so a lane owns 4 consecutive n (16-byte stores).
#include "common.h"
#include "mvf_hip_internal.h"
#include <type_traits>

namespace {

struct HGemmArgs {
  const float* A; long sam, sak;
  const float* B; long sbk, sbn;
  float* C; long ldc;
  const float* bias;
  const float* table; long tab_si, tab_sn; int tab_div, tab_mod;
  int M, N, K;
  int relu, accumulate;
  float alpha;
  uint32_t d_thresh; float d_scale; uint64_t d_seed, d_offset;   // epilogue dropout (d_thresh != 0)
  const float* resid; long ldr;                                   // epilogue: C = resid + ...
  float* rowsum; int rowsum_acc;   // rowsum[m] (+)= sum_k A(m,k); written by the first column of tiles
};

constexpr int TK = 64;            // k step: 64 MFMAs (2048 cycles) per wave and barrier -- covers the L2/MALL latency of the next tile's loads
constexpr int LDT = 80;           // LDS row stride (floats): rows g, g+1 of a fragment read land 16 banks apart
constexpr int TILE_F = TK * LDT;  // floats per operand tile
constexpr int NPASS = TK / 16;    // 16-byte loads per thread and operand tile (256 threads x 4 floats = 16 k-rows of 64)
// element (k, c) of a tile lives at k*LDT + (c ^ SW(k)): the XOR moves whole 4-float groups, so the 16-byte k-major
// stores stay contiguous, fragment reads (16 consecutive c at fixed k) stay conflict-free, and the TRANSPOSING scalar
// stores of a k-contiguous operand (8 lanes = 8 different k-quads, same c) spread over 8 banks instead of 1
__device__ __forceinline__ int lds_at(int k, int c) { return k * LDT + (c ^ (((k >> 2) & 7) << 2)); }

// Global -> registers for one [TK][64] operand tile.  X(r, k) = X[r*sr + k*sk], r = tile row (m or n), exactly one of
// sr, sk is 1.  KMAJOR (sr == 1): thread -> (k = t/16 [+16], rows 4*(t%16)..+3), float4 along the rows.
// else (sk == 1): thread -> (row = t/8 [+32], k = 4*(t%8)..+3), float4 along k.
// GUARD == false: the tile is known to be fully in range and 16-byte loadable (M, N multiples of 64, K of 32, aligned
// strides) -- straight-line code, all loads of a k step issue back to back.  GUARD == true (ragged shapes such as the
// 387-wide first FC layer, tiny query GEMMs) pays per-load range checks.
// ROWS = 64 | 32 rows of the tile (32: the half-height output tiles of small problems; two passes instead of four): KMAJOR
// thread -> (k = t / (ROWS/4) [+ 1024/ROWS per pass], rows 4 (t % (ROWS/4))), else thread -> (row = t/8 [+32], k = 4 (t%8) [+32])
template <int ROWS>
struct TileGeom {
  static constexpr int TPK = ROWS / 4;          // threads along the rows of one k (KMAJOR)
  static constexpr int KPP = 256 / TPK;         // k rows per pass (KMAJOR)
  static constexpr int PASSES = ROWS == 64 ? 4 : 2;
  // pass h of the k-contiguous form: row half (ROWS = 64 only) and k half
  static __device__ __forceinline__ int row_off(int h) { return ROWS == 64 ? 32 * (h & 1) : 0; }
  static __device__ __forceinline__ int k_off(int h) { return ROWS == 64 ? 32 * (h >> 1) : 32 * h; }
};

template <bool KMAJOR, bool GUARD, int ROWS = 64>
__device__ __forceinline__ void tile_load(const float* X, long sr, long sk, int r0, int R, int k0, int K, bool vec,
                                          f32x4_t (&v)[NPASS]) {
  using G = TileGeom<ROWS>;
  const int t = threadIdx.x;
  if constexpr (!GUARD) {
#pragma unroll
    for (int h = 0; h < G::PASSES; ++h) {
      if constexpr (KMAJOR)
        v[h] = *reinterpret_cast<const f32x4_t*>(X + (long)(k0 + t / G::TPK + G::KPP * h) * sk + r0 + (t % G::TPK) * 4);
      else
        v[h] = *reinterpret_cast<const f32x4_t*>(X + (long)(r0 + (t >> 3) + G::row_off(h)) * sr + k0 + (t & 7) * 4 + G::k_off(h));
    }
    return;
  }
#pragma unroll
  for (int h = 0; h < G::PASSES; ++h) {
    f32x4_t x = {0.f, 0.f, 0.f, 0.f};
    if constexpr (KMAJOR) {
      const int k = k0 + t / G::TPK + G::KPP * h, r = r0 + (t % G::TPK) * 4;
      if (k < K) {
        const float* p = X + (long)k * sk + r;
        if (vec && r + 3 < R) x = *reinterpret_cast<const f32x4_t*>(p);
        else {
#pragma unroll
          for (int s = 0; s < 4; ++s) x[s] = r + s < R ? p[s] : 0.f;
        }
      }
    } else {
      const int r = r0 + (t >> 3) + G::row_off(h), k = k0 + (t & 7) * 4 + G::k_off(h);
      if (r < R) {
        const float* p = X + (long)r * sr + k;
        if (vec && k + 3 < K) x = *reinterpret_cast<const f32x4_t*>(p);
        else {
#pragma unroll
          for (int s = 0; s < 4; ++s) x[s] = k + s < K ? p[s] : 0.f;
        }
      }
    }
    v[h] = x;
  }
}

// registers -> LDS tile [k][row]
template <bool KMAJOR, int ROWS = 64>
__device__ __forceinline__ void tile_store(float* T, const f32x4_t (&v)[NPASS]) {
  using G = TileGeom<ROWS>;
  const int t = threadIdx.x;
#pragma unroll
  for (int h = 0; h < G::PASSES; ++h) {
    if constexpr (KMAJOR) {
      *reinterpret_cast<f32x4_t*>(T + lds_at(t / G::TPK + G::KPP * h, (t % G::TPK) * 4)) = v[h];
    } else {
      const int r = (t >> 3) + G::row_off(h), k = (t & 7) * 4 + G::k_off(h);
#pragma unroll
      for (int s = 0; s < 4; ++s) T[lds_at(k + s, r)] = v[h][s];
    }
  }
}

// one TM x 64 output tile (bx, by) of the problem `a` (TM = 64, or 32 for problems that would not cover the chip with 64-row tiles:
// M = 768 rows x N <= 1024 is 48 .. 192 workgroups, each a serial chain of K/4 x 4 fp32 MFMAs per wave); lds: 4 * TILE_F floats
template <bool AKM, bool BKM, bool GUARD, int TM = 64>
__device__ __forceinline__ void hgemm_tile(const HGemmArgs& a, int bx, int by, float* lds, bool avec, bool bvec) {
  constexpr int MI = TM / 32;             // 16-row accumulator tiles per wave along m
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int m0 = by * TM, n0 = bx * 64;
  const int wm = (wave >> 1) * (TM / 2), wn = (wave & 1) * 32;
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float rs[2] = {0.f, 0.f};
  const bool want_rs = a.rowsum != nullptr && bx == 0 && (wave & 1) == 0;

  // Two register sets: tile j travels global -> registers set j & 1 -> LDS buffer j & 1.  While tile t is multiplied, tile t + 1 sits
  // in (or is on its way to) its registers and tile t + 2 is in flight: a request has TWO k steps to come back (one in the first
  // form, which asked for tile t + 1 at the top of step t and needed it at the bottom: 1.17 us per 64-deep step of 32-row tiles
  // against 0.5 us of MFMAs -- the K = 2 304 first-layer GEMM took 42 us, and under the backbone's memory traffic several times that)
  f32x4_t ra[2][NPASS], rb[2][NPASS];
  const int nt = (a.K + TK - 1) / TK;
  tile_load<AKM, GUARD, TM>(a.A, a.sam, a.sak, m0, a.M, 0, a.K, avec, ra[0]);
  tile_load<BKM, GUARD>(a.B, a.sbn, a.sbk, n0, a.N, 0, a.K, bvec, rb[0]);
  if (nt > 1) {
    tile_load<AKM, GUARD, TM>(a.A, a.sam, a.sak, m0, a.M, TK, a.K, avec, ra[1]);
    tile_load<BKM, GUARD>(a.B, a.sbn, a.sbk, n0, a.N, TK, a.K, bvec, rb[1]);
  }
  tile_store<AKM, TM>(lds, ra[0]);
  tile_store<BKM>(lds + TILE_F, rb[0]);
  if (nt > 2) {
    tile_load<AKM, GUARD, TM>(a.A, a.sam, a.sak, m0, a.M, 2 * TK, a.K, avec, ra[0]);
    tile_load<BKM, GUARD>(a.B, a.sbn, a.sbk, n0, a.N, 2 * TK, a.K, bvec, rb[0]);
  }
  __syncthreads();
  // k step t; S = (t + 1) & 1 = the register set of tile t + 1 (a compile-time index: the sets must stay in registers)
  auto kstep = [&](int t, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    const float* As = lds + (t & 1) * 2 * TILE_F;
    const float* Bs = As + TILE_F;
#pragma unroll
    for (int q = 0; q < TK / 4; ++q) {
      float af[2] = {0.f, 0.f}, bf[2];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = As[lds_at(4 * q + g, wm + i * 16 + li)];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = Bs[lds_at(4 * q + g, wn + j * 16 + li)];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);
      if (want_rs) { rs[0] += af[0]; rs[1] += af[1]; }
    }
    if (t + 1 < nt) {
      float* An = lds + ((t + 1) & 1) * 2 * TILE_F;
      tile_store<AKM, TM>(An, ra[S]);
      tile_store<BKM>(An + TILE_F, rb[S]);
    }
    if (t + 3 < nt) {   // the set just emptied takes tile t + 3
      tile_load<AKM, GUARD, TM>(a.A, a.sam, a.sak, m0, a.M, (t + 3) * TK, a.K, avec, ra[S]);
      tile_load<BKM, GUARD>(a.B, a.sbn, a.sbk, n0, a.N, (t + 3) * TK, a.K, bvec, rb[S]);
    }
    __syncthreads();
  };
  for (int t = 0; t < nt; t += 2) {
    kstep(t, std::integral_constant<int, 1>{});
    if (t + 1 < nt) kstep(t + 1, std::integral_constant<int, 0>{});
  }

  const int m_base = m0 + wm, n_base = n0 + wn;
  if (want_rs) {   // combine the 4 k-groups (lanes li, li+16, li+32, li+48)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float v = rs[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int m = m_base + i * 16 + li;
      if (g == 0 && m < a.M) a.rowsum[m] = a.rowsum_acc ? a.rowsum[m] + v : v;
    }
  }

  const bool vec_out = (a.ldc % 4 == 0) && (a.N % 4 == 0) && (((uintptr_t)a.C & 15) == 0) &&
                       (a.resid == nullptr || (a.ldr % 4 == 0 && ((uintptr_t)a.resid & 15) == 0));
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m_base + i * 16 + li;
    if (m >= a.M) continue;
    const float* trow = a.table ? a.table + (long)((m / a.tab_div) % a.tab_mod) * a.tab_si : nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n_base + j * 16 + 4 * g;
      if (n >= a.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = a.alpha * acc[i][j][r];
        if (n + r < a.N) {
          if (a.bias) v[r] += a.bias[n + r];
          if (trow) v[r] += trow[(long)(n + r) * a.tab_sn];
        }
        if (a.relu) v[r] = fmaxf(v[r], 0.f);
      }
      float* cp = a.C + (long)m * a.ldc + n;
      if (a.d_thresh != 0u) {     // dropout(v); mask index = element index of the dense [M, ldc] output
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] = drop_keep(a.d_seed, a.d_offset, (uint64_t)((long)m * a.ldc + n + r), a.d_thresh) ? v[r] * a.d_scale : 0.f;
      }
      if (a.resid != nullptr) {   // y = resid + ...
        const float* rp = a.resid + (long)m * a.ldr + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < a.N) v[r] += rp[r];
      }
      if (vec_out) {
        if (a.accumulate) {
          const float4 o = *reinterpret_cast<const float4*>(cp);
          v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
        }
        *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < a.N) cp[r] = a.accumulate ? cp[r] + v[r] : v[r];
      }
    }
  }
}

// operand forms: A k-major <=> sak != 1 (then sam == 1); B k-major <=> sbk != 1 (then sbn == 1)
template <bool AKM, bool BKM, bool GUARD, int TM = 64>
__global__ __launch_bounds__(256) void hgemm_kernel(HGemmArgs a, int avec, int bvec) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 4 * TILE_F floats (80 KiB)
  hgemm_tile<AKM, BKM, GUARD, TM>(a, blockIdx.x, blockIdx.y, lds, avec != 0, bvec != 0);
}

// Backward of y = x W^T + b in one launch: tiles [0, nx) compute dX = dy . W (A = dy k-contiguous, B = W k-major),
// tiles [nx, nx + nw) compute dW (+)= dy^T . x (both k-major) and, in their first tile column, db (+)= colsum(dy).
template <bool GUARD, int TM = 64>
__global__ __launch_bounds__(256) void hlinear_bwd_kernel(HGemmArgs dx, HGemmArgs dw, int nx, int dx_tiles_n, int dw_tiles_n,
                                                          int dx_avec, int dx_bvec, int dw_avec, int dw_bvec) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 4 * TILE_F floats (80 KiB)
  const int b = blockIdx.x;
  if (b < nx) hgemm_tile<false, true, GUARD, TM>(dx, b % dx_tiles_n, b / dx_tiles_n, lds, dx_avec != 0, dx_bvec != 0);
  else hgemm_tile<true, true, GUARD, TM>(dw, (b - nx) % dw_tiles_n, (b - nx) / dw_tiles_n, lds, dw_avec != 0, dw_bvec != 0);
}

// half-height tiles where 64-row tiles would leave most of the chip idle (MVF_HGEMM_TM=64 | 32 pins the height: A/B measurements)
int g_hgemm_tm = [] { const char* e = getenv("MVF_HGEMM_TM"); return e ? atoi(e) : 0; }();
int pick_tm(long tiles64) { return g_hgemm_tm == 64 || g_hgemm_tm == 32 ? g_hgemm_tm : (tiles64 < 192 ? 32 : 64); }

// out[c] (+)= sum_r x[r*ld + c]   -- bias gradients that are not attached to a weight-gradient GEMM
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ld, int rows, int cols,
                                                     float* __restrict__ out, int accumulate) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < cols)
    for (int r0 = rl; r0 < rows; r0 += 32) {      // batches of 8 loads, same addition order (load -> add loops pay a round trip per row)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[(long)min(r0 + 4 * u, rows - 1) * ld + c];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + 4 * u < rows) s += v[u];
    }
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < cols) {
    s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    out[c] = accumulate ? out[c] + s : s;
  }
}

// the same over a slice of the rows: part[split][c] = sum of rows [split*chunk, (split+1)*chunk) -- stage 1 of a bias gradient
// over tens of thousands of rows (trainable backbone blocks); stage 2 = mvf_sum_batches (fixed order: deterministic)
__global__ __launch_bounds__(256) void colsum_split_kernel(const float* __restrict__ x, long ld, int rows, int cols, int chunk,
                                                           float* __restrict__ part) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
  float s = 0.f;
  if (c < cols)
    for (int ra = r0 + rl; ra < r1; ra += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[(long)min(ra + 4 * u, r1 - 1) * ld + c];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (ra + 4 * u < r1) s += v[u];
    }
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < cols)
    part[(size_t)blockIdx.y * cols + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

constexpr int LDS_B = 4 * TILE_F * 4;   // dynamic LDS per workgroup (above the 64 KiB static limit)

template <typename K>
void allow_lds(K kern) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
}

// 16-byte loads along the contiguous direction are legal when base and the other stride keep 16-B alignment
bool can_vec(const float* p, long other_stride) { return ((uintptr_t)p & 15) == 0 && other_stride % 4 == 0; }

}  // namespace

extern "C" int mvf_hgemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
                         const float* bias, const float* table, long tab_si, long tab_sn, int tab_div, int tab_mod,
                         int M, int N, int K, float alpha, int relu, int accumulate, hipStream_t st) {
  return mvf_hgemm_ex(A, sam, sak, B, sbk, sbn, C, ldc, bias, table, tab_si, tab_sn, tab_div, tab_mod, M, N, K, alpha, relu,
                      accumulate, nullptr, 0, 0.f, 0, 0, st);
}

// mvf_hgemm + fused dropout / residual epilogue: C = [resid +] dropout_p(act(...)) (mask index = m*ldc + n, the element
// index of a dense [M, ldc] output -- the same indexing mvf_dropout_add uses, so its backward is that kernel on dy)
extern "C" int mvf_hgemm_ex(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
                            const float* bias, const float* table, long tab_si, long tab_sn, int tab_div, int tab_mod,
                            int M, int N, int K, float alpha, int relu, int accumulate, const float* resid, long ldr,
                            float drop_p, uint64_t drop_seed, uint64_t drop_offset, hipStream_t st) {
  MVF_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0);
  MVF_CHECK_ARG(table == nullptr || (tab_div > 0 && tab_mod > 0));
  MVF_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && !((resid != nullptr || drop_p > 0.f) && accumulate));
  MVF_CHECK_ARG((sam == 1 || sak == 1) && (sbk == 1 || sbn == 1));   // each operand contiguous along m/n or along k
  HGemmArgs a{};
  a.A = A; a.sam = sam; a.sak = sak; a.B = B; a.sbk = sbk; a.sbn = sbn; a.C = C; a.ldc = ldc; a.bias = bias;
  a.table = table; a.tab_si = tab_si; a.tab_sn = tab_sn; a.tab_div = tab_div > 0 ? tab_div : 1;
  a.tab_mod = tab_mod > 0 ? tab_mod : 1; a.M = M; a.N = N; a.K = K; a.relu = relu; a.accumulate = accumulate;
  a.alpha = alpha; a.resid = resid; a.ldr = ldr;
  a.d_thresh = drop_p > 0.f ? (uint32_t)std::min<double>(4294967295.0, (double)drop_p * 4294967296.0) : 0u;
  a.d_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  a.d_seed = drop_seed; a.d_offset = drop_offset;
  // a [1, K] or [K, 1] operand has both strides "1": prefer the k-contiguous reading
  const bool akm = sak != 1, bkm = sbk != 1;
  const int avec = can_vec(A, akm ? sak : sam), bvec = can_vec(B, bkm ? sbk : sbn);
  const int tm = pick_tm((long)ceil_div(N, 64) * ceil_div(M, 64));
  dim3 grid(ceil_div(N, 64), ceil_div(M, tm));
  const bool full = avec && bvec && M % 64 == 0 && N % 64 == 0 && K % TK == 0;
#define HG2(AK, BK, GD, TMV)                                                                                \
  do {                                                                                                      \
    allow_lds(hgemm_kernel<AK, BK, GD, TMV>);                                                               \
    hipLaunchKernelGGL((hgemm_kernel<AK, BK, GD, TMV>), grid, dim3(256), LDS_B, st, a, avec, bvec);         \
  } while (0)
#define HG(AK, BK)                                                                                          \
  do {                                                                                                      \
    if (tm == 32) { if (full) HG2(AK, BK, false, 32); else HG2(AK, BK, true, 32); }                         \
    else { if (full) HG2(AK, BK, false, 64); else HG2(AK, BK, true, 64); }                                  \
  } while (0)
  if (akm && bkm) HG(true, true);
  else if (akm) HG(true, false);
  else if (bkm) HG(false, true);
  else HG(false, false);
#undef HG
#undef HG2
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// Backward of y = x W^T + b (x [M,K] rows of stride ldx, W [N,K] rows of stride ldw, dy [M,N] rows of stride ldy,
// all with unit inner stride) in ONE launch:  dx = dy . W (may be NULL);  dW (+)= dy^T . x;  db (+)= colsum(dy)
// (may be NULL).  accumulate_params: add into dW / db (the flat gradient buffer) instead of overwriting them.
extern "C" int mvf_hlinear_bwd(const float* dy, long ldy, const float* x, long ldx, const float* W, long ldw, float* dx,
                               long lddx, float* dW, long lddw, float* db, int M, int N, int K, int accumulate_params,
                               hipStream_t st) {
  MVF_CHECK_ARG(dy && x && W && dW && M > 0 && N > 0 && K > 0);
  HGemmArgs gx{}, gw{};
  // dX[m, k] = sum_n dy[m, n] W[n, k]:   A = dy (rows m, k-index n contiguous), B(n, k) = W[n*ldw + k] (k-major)
  gx.A = dy; gx.sam = ldy; gx.sak = 1; gx.B = W; gx.sbk = ldw; gx.sbn = 1; gx.C = dx; gx.ldc = lddx;
  gx.M = M; gx.N = K; gx.K = N; gx.alpha = 1.f; gx.tab_div = gx.tab_mod = 1; gx.d_scale = 1.f;
  // dW[n, k] = sum_m dy[m, n] x[m, k]:   A(n, m) = dy[m*ldy + n], B(m, k) = x[m*ldx + k]  (both k-major)
  gw.A = dy; gw.sam = 1; gw.sak = ldy; gw.B = x; gw.sbk = ldx; gw.sbn = 1; gw.C = dW; gw.ldc = lddw;
  gw.M = N; gw.N = K; gw.K = M; gw.alpha = 1.f; gw.tab_div = gw.tab_mod = 1; gw.d_scale = 1.f;
  gw.accumulate = accumulate_params; gw.rowsum = db; gw.rowsum_acc = accumulate_params;
  const long t64 = (long)(dx != nullptr ? ceil_div(K, 64) * ceil_div(M, 64) : 0) + (long)ceil_div(K, 64) * ceil_div(N, 64);
  const int tm = pick_tm(t64);
  const int dxn = ceil_div(K, 64), dxm = ceil_div(M, tm), dwn = ceil_div(K, 64), dwm = ceil_div(N, tm);
  const int nx = dx != nullptr ? dxn * dxm : 0, nw = dwn * dwm;
  const bool full = can_vec(dy, ldy) && can_vec(W, ldw) && can_vec(x, ldx) && M % 64 == 0 && N % 64 == 0 && K % 64 == 0;
  const int v0 = full ? 1 : (int)can_vec(dy, ldy), v1 = full ? 1 : (int)can_vec(W, ldw), v3 = full ? 1 : (int)can_vec(x, ldx);
#define HB(GD, TMV)                                                                                                         \
  do {                                                                                                                      \
    allow_lds(hlinear_bwd_kernel<GD, TMV>);                                                                                 \
    hipLaunchKernelGGL((hlinear_bwd_kernel<GD, TMV>), dim3(nx + nw), dim3(256), LDS_B, st, gx, gw, nx, dxn, dwn, v0, v1, v0, v3); \
  } while (0)
  if (tm == 32) { if (full) HB(false, 32); else HB(true, 32); }
  else { if (full) HB(false, 64); else HB(true, 64); }
#undef HB
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_colsum_split(const float* x, long ld, int rows, int cols, int splits, float* part, hipStream_t st) {
  MVF_CHECK_ARG(x && part && rows > 0 && cols > 0 && splits > 0);
  hipLaunchKernelGGL(colsum_split_kernel, dim3(ceil_div(cols, 64), splits), dim3(256), 0, st, x, ld, rows, cols,
                     ceil_div(rows, splits), part);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

extern "C" int mvf_colsum(const float* x, long ld, int rows, int cols, float* out, int accumulate, hipStream_t st) {
  MVF_CHECK_ARG(x && out && rows > 0 && cols > 0);
  hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(cols, 64)), dim3(256), 0, st, x, ld, rows, cols, out, accumulate);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}
