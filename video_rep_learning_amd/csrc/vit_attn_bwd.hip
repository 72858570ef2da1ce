// Backward of the ViT self-attention core in bf16 (trainable backbone blocks, MODEL.BASE_MODEL.LAYER < depth, bf16 mode):
//   o = softmax(q k^T / sqrt(64)) v  per (frame, head);  given dO:  dV = P^T dO,  dP = dO V^T,  dS = P o (dP - delta),
//   dQ = dS K / 8,  dK = dS^T Q / 8,  delta = rowsum(dO o O),  P recomputed from the forward's log-sum-exp.
// Stands in for the autograd backward of timm Attention.forward inside the reference's ViTBackEnd blocks
// (CARL_MVF/models/transformer.py:364-392, run under fp16 autocast there).  The fp32 kernels of head_attn_mfma.hip do the
// same job at the fp32 MFMA rate (1/16 of bf16) and remain the parity-mode path.
//
// gfx950 design -- the scheme of head_attn_mfma.hip (owned index on the MFMA column, streamed index on its rows, an
// accumulator tile is directly the B operand of the next product) on v_mfma_f32_16x16x32_bf16:
//   lane (c = lane & 15, g = lane >> 4) holds A[row c][k = 8g + j], B[k = 8g + j][col c], D[row 4g + r][col c].
//   Two stacked D tiles (32 streamed rows) packed to bf16 are a B operand whose k index is the streamed row
//   rho(g, j) = 4g + j (j < 4), 16 + 4g + (j - 4) (j >= 4); the matching A operand (the streamed matrix TRANSPOSED) comes
//   out of the row-major LDS image by two ds_read_b64_tr_b16 -- the pairing vit_attn.hip's forward uses for O^T = V^T P^T.
//   dQ kernel  (owned = 2 query tiles per wave, streamed = keys):
//        S^T  = K . Q^T            A = K rows (LDS),        B = Q rows (registers)
//        dP^T = V . dO^T           A = V rows (LDS),        B = dO rows (registers)
//        dQ^T += K^T . dS^T        A = K transposed (LDS),  B = dS^T tiles as they stand
//   dK/dV kernel (owned = 2 key tiles per wave, streamed = queries):
//        S  = Q . K^T,  dP = dO . V^T        A = Q / dO rows (LDS),          B = K / V rows (registers)
//        dV^T += dO^T . P,  dK^T += Q^T . dS  A = dO / Q transposed (LDS),    B = P / dS tiles as they stand
//   Owner-computes: no atomics, bit-reproducible.  The streamed operands arrive by LDS-DMA (global_load_lds_dwordx4) in
//   128-byte rows with the 32-byte chunks XOR-swizzled by ((row >> 1) & 3) on the SOURCE address: conflict-free for the
//   transposing reads (as measured in the forward), two-way for the 16-byte row reads.
//   lse (log2 domain, from the forward) and delta live in [F][H][Npad] arrays, Npad = 16 * ceil(N / 16).
#include "common.h"
#include "mvf_hip_internal.h"

namespace {

constexpr int HD = 64;
constexpr float LOG2E = 1.4426950408889634f;

struct BwdArgs {
  const bf16_t* qkv;   // [F*N, 3*D]
  const bf16_t* o;     // [F*N, D]   (delta kernel only)
  const bf16_t* d_o;   // [F*N, D]
  const float* lse;    // [F, H, Npad] log2-domain log-sum-exp of the scaled scores
  float* delta;        // [F, H, Npad]
  void* dqkv;          // [F*N, 3*D] fp32 or bf16 (out_bf16)
  int out_bf16;
  int N, H, D, Npad;
  float scale_log2;    // 64^-0.5 * log2(e)
  float scale;         // 64^-0.5
};

// four consecutive gradient values -> dqkv[off .. off + 3]
__device__ __forceinline__ void store4(const BwdArgs& a, size_t off, const f32x4_t& v) {
  if (a.out_bf16)
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.dqkv) + off) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
  else
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.dqkv) + off) = make_float4(v[0], v[1], v[2], v[3]);
}

// byte offset of logical 16-byte chunk ch (0..7) of image row `row` (128-byte rows, 32-byte chunks swizzled)
__device__ __forceinline__ int row_off(int row, int ch) {
  return row * 128 + ((((ch >> 1) ^ ((row >> 1) & 3)) << 5) | ((ch & 1) << 4));
}

// rows [r0, r0 + ROWS) of a [*, ld] bf16 matrix (64 columns from src) -> LDS image by LDS-DMA; rows >= N repeat row N - 1
template <int ROWS>
__device__ __forceinline__ void stage_dma(char* img, const bf16_t* src, size_t ld, int r0, int N, int wave, int lane) {
  const int prow = lane >> 3, pc = lane & 7;
  for (int p = wave; p < ROWS / 8; p += 4) {
    const int r = p * 8 + prow;
    const bf16_t* s = src + (size_t)min(r0 + r, N - 1) * ld + ((pc ^ (((r >> 1) & 3) << 1)) << 3);
    __builtin_amdgcn_global_load_lds(GLB_PTR(s), LDS_PTR(img + p * 1024), 16, 0, 0);
  }
}

// A operand = rows of the streamed image: lane (c, g) <- X[t*16 + c][32 ks + 8g .. +8]
__device__ __forceinline__ bf16x8_t row_frag(const char* img, int t, int ks, int c, int g) {
  return *reinterpret_cast<const bf16x8_t*>(img + row_off(t * 16 + c, ks * 4 + g));
}

// A operand = the streamed image TRANSPOSED: lane (c, g) <- X[rho(g, j) + 32 st][dt*16 + c], j = 0..7
__device__ __forceinline__ bf16x8_t tr_frag(const char* img, int st, int dt, int c, int g) {
  union { bf16x8_t v; bf16x4_t h[2]; } f;
  const int row = st * 32 + 4 * g + (c >> 2);
  const char* p0 = img + row * 128 + ((dt ^ ((row >> 1) & 3)) << 5) + 8 * (c & 3);
  f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(p0));
  f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(p0 + 16 * 128));
  return f.v;
}

__device__ __forceinline__ bf16x8_t pack2(const f32x4_t& a, const f32x4_t& b) {
  union { bf16x8_t v; uint32_t u[4]; } p;
  p.u[0] = pack_bf16x2(a[0], a[1]);
  p.u[1] = pack_bf16x2(a[2], a[3]);
  p.u[2] = pack_bf16x2(b[0], b[1]);
  p.u[3] = pack_bf16x2(b[2], b[3]);
  return p.v;
}

// ------------------------------------------------------------------------------------------------ delta = rowsum(dO o O)
__global__ __launch_bounds__(256) void vattn_delta_kernel(BwdArgs a, int F) {
  // one thread per (frame, head, query): 64 products from two 128-byte rows
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)F * a.H * a.Npad;
  if (idx >= total) return;
  const int q = (int)(idx % a.Npad);
  const int h = (int)((idx / a.Npad) % a.H);
  const size_t f = idx / ((size_t)a.Npad * a.H);
  float s = 0.f;
  if (q < a.N) {
    const bf16_t* po = a.o + (f * a.N + q) * a.D + h * HD;
    const bf16_t* pd = a.d_o + (f * a.N + q) * a.D + h * HD;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint4 x = *reinterpret_cast<const uint4*>(po + i * 8), y = *reinterpret_cast<const uint4*>(pd + i * 8);
      const unsigned xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s = fmaf(__uint_as_float(xs[e] << 16), __uint_as_float(ys[e] << 16), s);
        s = fmaf(__uint_as_float(xs[e] & 0xffff0000u), __uint_as_float(ys[e] & 0xffff0000u), s);
      }
    }
  }
  a.delta[idx] = s;
}

// ------------------------------------------------------------------------------------------------ dQ
// grid (F*H, ceil(qtiles / 8)); a wave owns query tiles qt0, qt0 + 1; KT key tiles per LDS block (even)
template <int KT>
__global__ __launch_bounds__(256, 2) void vattn_dq_kernel(BwdArgs a) {
  constexpr int KROWS = KT * 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * KROWS * 128];
  char* sk = smem;
  char* sv = smem + KROWS * 128;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int f = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const size_t ld = (size_t)3 * a.D;
  const bf16_t* base = a.qkv + (size_t)f * a.N * ld + h * HD;
  const bf16_t* dob = a.d_o + (size_t)f * a.N * a.D + h * HD;
  const int qt0 = (blockIdx.y * 4 + wave) * 2;
  const bool active = qt0 * 16 < a.N;
  bf16x8_t qf[2][2], dof[2][2];
  float lse[2], dl[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = (qt0 + i) * 16 + c;
    const int qr = min(q, a.N - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[i][ks] = *reinterpret_cast<const bf16x8_t*>(base + (size_t)qr * ld + ks * 32 + g * 8);
      dof[i][ks] = *reinterpret_cast<const bf16x8_t*>(dob + (size_t)qr * a.D + ks * 32 + g * 8);
    }
    const size_t li = ((size_t)f * a.H + h) * a.Npad + min(q, a.Npad - 1);
    lse[i] = a.lse[li];
    dl[i] = a.delta[li];
  }
  asm volatile("" : "+v"(qf[0][0]), "+v"(qf[1][1]), "+v"(dof[0][0]), "+v"(dof[1][1]), "+v"(lse[0]), "+v"(dl[1]));
  f32x4_t dq[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[i][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < a.N; k0 += KROWS) {
    if (k0 > 0) __syncthreads();                       // everyone is done with the previous block
    stage_dma<KROWS>(sk, base + a.D, ld, k0, a.N, wave, lane);
    stage_dma<KROWS>(sv, base + 2 * a.D, ld, k0, a.N, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active) continue;
    const int nkeys = a.N - k0;
#pragma unroll
    for (int st = 0; st < KT / 2; ++st) {
      if (st * 32 >= nkeys) break;
      f32x4_t ds[2][2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * st + tt;
        f32x4_t s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dp[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8_t kf = row_frag(sk, t, ks, c, g), vf = row_frag(sv, t, ks, c, g);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[i][ks], s[i], 0, 0, 0);      // S^T[key 4g+r][query c]
            dp[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[i][ks], dp[i], 0, 0, 0);   // dP^T
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = t * 16 + 4 * g + r < nkeys;
            const float p = ok ? __builtin_amdgcn_exp2f(fmaf(s[i][r], a.scale_log2, -lse[i])) : 0.f;
            ds[i][tt][r] = p * (dp[i][r] - dl[i]) * a.scale;
          }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8_t kt = tr_frag(sk, st, dt, c, g);                                        // K^T[d dt*16+c][key rho]
#pragma unroll
        for (int i = 0; i < 2; ++i)
          dq[i][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, pack2(ds[i][0], ds[i][1]), dq[i][dt], 0, 0, 0);
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = (qt0 + i) * 16 + c;
    if (q < a.N) {
      const size_t off = ((size_t)f * a.N + q) * ld + h * HD;           // dQ^T[d = 16 dt + 4g + r][query c]
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store4(a, off + dt * 16 + 4 * g, dq[i][dt]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ dK, dV
// grid (F*H, ceil(ktiles / 8)); a wave owns key tiles kt0, kt0 + 1; QT query tiles per LDS block (even)
template <int QT>
__global__ __launch_bounds__(256, 2) void vattn_dkv_kernel(BwdArgs a) {
  constexpr int QROWS = QT * 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * QROWS * 128];
  char* sq = smem;
  char* sdo = smem + QROWS * 128;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int f = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const size_t ld = (size_t)3 * a.D;
  const bf16_t* base = a.qkv + (size_t)f * a.N * ld + h * HD;
  const bf16_t* dob = a.d_o + (size_t)f * a.N * a.D + h * HD;
  const float* lse_b = a.lse + ((size_t)f * a.H + h) * a.Npad;
  const float* dl_b = a.delta + ((size_t)f * a.H + h) * a.Npad;
  const int kt0 = (blockIdx.y * 4 + wave) * 2;
  const bool active = kt0 * 16 < a.N;
  bf16x8_t kf[2][2], vf[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int kr = min((kt0 + i) * 16 + c, a.N - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[i][ks] = *reinterpret_cast<const bf16x8_t*>(base + a.D + (size_t)kr * ld + ks * 32 + g * 8);
      vf[i][ks] = *reinterpret_cast<const bf16x8_t*>(base + 2 * a.D + (size_t)kr * ld + ks * 32 + g * 8);
    }
  }
  asm volatile("" : "+v"(kf[0][0]), "+v"(kf[1][1]), "+v"(vf[0][0]), "+v"(vf[1][1]));
  f32x4_t dk[2][4], dv[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dk[i][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      dv[i][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }

  for (int q0 = 0; q0 < a.N; q0 += QROWS) {
    if (q0 > 0) __syncthreads();
    stage_dma<QROWS>(sq, base, ld, q0, a.N, wave, lane);
    stage_dma<QROWS>(sdo, dob, (size_t)a.D, q0, a.N, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active) continue;
    const int nq = a.N - q0;
#pragma unroll
    for (int st = 0; st < QT / 2; ++st) {
      if (st * 32 >= nq) break;
      f32x4_t p[2][2], ds[2][2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * st + tt;
        // the four streamed queries of this lane's accumulator rows (Npad: always in range, 16-byte aligned)
        const int qrow = min(q0 + t * 16 + 4 * g, a.Npad - 4);
        const float4 ls = *reinterpret_cast<const float4*>(lse_b + qrow), dl = *reinterpret_cast<const float4*>(dl_b + qrow);
        const float lsr[4] = {ls.x, ls.y, ls.z, ls.w}, dlr[4] = {dl.x, dl.y, dl.z, dl.w};
        f32x4_t s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dp[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8_t qfr = row_frag(sq, t, ks, c, g), dfr = row_frag(sdo, t, ks, c, g);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[i][ks], s[i], 0, 0, 0);     // S[query 4g+r][key c]
            dp[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr, vf[i][ks], dp[i], 0, 0, 0);   // dP
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = t * 16 + 4 * g + r < nq;
            const float pr = ok ? __builtin_amdgcn_exp2f(fmaf(s[i][r], a.scale_log2, -lsr[r])) : 0.f;
            p[i][tt][r] = pr;
            ds[i][tt][r] = pr * (dp[i][r] - dlr[r]) * a.scale;
          }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8_t dot = tr_frag(sdo, st, dt, c, g), qt = tr_frag(sq, st, dt, c, g);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          dv[i][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pack2(p[i][0], p[i][1]), dv[i][dt], 0, 0, 0);
          dk[i][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, pack2(ds[i][0], ds[i][1]), dk[i][dt], 0, 0, 0);
        }
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int key = (kt0 + i) * 16 + c;
    if (key < a.N) {
      const size_t off = ((size_t)f * a.N + key) * ld + h * HD;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        store4(a, off + a.D + dt * 16 + 4 * g, dk[i][dt]);
        store4(a, off + 2 * a.D + dt * 16 + 4 * g, dv[i][dt]);
      }
    }
  }
}

}  // namespace

// workspace-free: delta is caller-owned [F, H, Npad] like lse
extern "C" int mvf_vit_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                                int out_dtype, int F, int N, int H, int D, hipStream_t st) {
  MVF_CHECK_ARG(out_dtype == MVF_F32 || out_dtype == MVF_BF16);
  MVF_CHECK_ARG(qkv && o && d_o && lse && delta && dqkv && F > 0 && N > 1 && H > 0 && D == H * HD);
  MVF_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)o % 16) == 0 && ((uintptr_t)d_o % 16) == 0 &&
                ((uintptr_t)lse % 16) == 0 && ((uintptr_t)delta % 16) == 0 && ((uintptr_t)dqkv % 16) == 0);
  BwdArgs a;
  a.qkv = (const bf16_t*)qkv; a.o = (const bf16_t*)o; a.d_o = (const bf16_t*)d_o; a.lse = lse; a.delta = delta; a.dqkv = dqkv;
  a.out_bf16 = out_dtype == MVF_BF16;
  a.N = N; a.H = H; a.D = D; a.Npad = ceil_div(N, 16) * 16;
  a.scale = 0.125f;
  a.scale_log2 = LOG2E * 0.125f;
  const size_t total = (size_t)F * H * a.Npad;
  hipLaunchKernelGGL(vattn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, F);
  const int tiles = ceil_div(N, 16);
  const dim3 grid(F * H, ceil_div(tiles, 8));
  if (N <= 224) {          // one LDS block holds the whole streamed sequence (ViT-B/16 at 224 px: N = 197)
    hipLaunchKernelGGL((vattn_dq_kernel<14>), grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL((vattn_dkv_kernel<14>), grid, dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL((vattn_dq_kernel<8>), grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL((vattn_dkv_kernel<8>), grid, dim3(256), 0, st, a);
  }
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}
