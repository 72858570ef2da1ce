// 256x256-tile bf16 MFMA GEMM for the frozen ViT backbone (the dominant kernel of the training step):
//     C[M,N] = epi(A[M,K] . W[N,K]^T + bias[N]),  K % 128 == 0, N % 32 == 0
// Same contract and epilogues as gemm_tc.hip (which stays the fp32 parity kernel and the odd-K fallback); replaces
// the ATen/cuBLAS GEMMs behind timm's nn.Linear / Conv2d(patch) calls reached from CARL_MVF/models/transformer.py:188.
//
// gfx950 design (DESIGN.md "gemm_tc256"):
//  * 512 threads = 8 waves as 2(M) x 4(N); a wave owns 128 x 64 outputs = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16
//    (128 accumulator registers); two waves share each SIMD.
//  * K tile = 64 bf16 (128-B rows).  A K tile is FOUR 16-KiB half-tiles in LDS: A0/A1 hold the m-quadrant 0/1 rows
//    of both wave rows, B0/B1 the n-quadrant 0/1 rows of all four wave columns; two K-tile buffers = 128 KiB.
//  * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, lane-linear image, XOR swizzle applied on the
//    per-lane SOURCE address and on the ds_read_b128 address) -- never through VGPRs.
//  * 4 phases per K tile, one output quadrant (64 x 32 per wave, 16 MFMAs) and one half-tile prefetch each
//    (g = running K-tile index of the workgroup):
//        P0: read B0(4) + A0(8)   MFMA (A0,B0)   issue (g+1).A1
//        P1: read B1(4)           MFMA (A0,B1)   issue (g+2).B0
//        P2: read A1(8)           MFMA (A1,B1)   issue (g+2).A0
//        P3: --                   MFMA (A1,B0)   issue (g+2).B1 ; s_waitcnt vmcnt(6)
//    Each phase is  {ds_reads, LDS-DMA issue} s_barrier {MFMAs} s_barrier.  The wave row wr = 1 runs ONE barrier
//    behind wr = 0, so on every SIMD one wave is in its MFMA segment while its partner loads (matrix pipe kept busy).
//  * PERSISTENT: one workgroup per CU walks several output tiles and the staging schedule above never stops at a
//    tile boundary -- K tiles g+1, g+2 simply belong to the NEXT output tile (the issue-side pointers switch to it in
//    the second-to-last K tile).  So only the very first tile of a workgroup pays the cold prologue; every other
//    tile starts with its first two K tiles already in flight while the previous tile's epilogue stores drain
//    (measured on the one-tile-per-workgroup form: prologue + first-tiles stall = 17 %, epilogue = 13 % of a K = 768
//    tile).  At a tile boundary the two wave rows are re-aligned (one extra barrier each side) so that both run their
//    epilogues in the same interval instead of one after the other.
//  * hazards (slots = intervals between workgroup barriers; wr=0 loads in slot 2k, computes in 2k+1; wr=1 one later):
//      RAW  a K tile's last half-tile (A1) is issued 4 phases before the previous K tile's P3 wait; vmcnt(6) leaves
//           exactly the 3 younger half-tiles (2 LDS-DMAs per wave each) in flight; every wave waits BEFORE the first
//           barrier of P3 and the first read of the new K tile comes after that barrier in both wave rows.  Epilogue
//           loads/stores and the bias DMA are issued between a P3 and the next P0, i.e. they are OLDER than every
//           DMA the next vmcnt(6) leaves in flight: the count stays exact (the wait is only more conservative).
//      WAR  B0 is re-staged one phase after its reads: they are retired by lgkmcnt(8) before P0's first barrier
//           (B reads are issued first; order pinned by sched_barrier).  A0, B1, A1 are re-staged two phases after
//           their reads, whose lgkmcnt(0) precedes the reading phase's second barrier in both wave rows.
//           The tile-boundary barriers and the epilogue only add distance.
//  * no vmcnt(0) / __syncthreads inside the loop; all LDS is one dynamic array (a second __shared__ object makes
//    hipcc drain the DMA queue before every ds_read).  The tile's bias slice travels by LDS-DMA too (a VGPR load in
//    the epilogue would make hipcc drain the next tile's DMAs).
//  * operands swapped (W fragment as MFMA-A) so a lane owns 4 consecutive output columns; v_permlane16_swap pairs two
//    tiles into 16-byte stores (gemm_tc_epi.h).
//  * N-grouping (round 1: "no change at all on fc1, 315 us either way" -- measured on 20-launch bursts).  Round 4, sustained at
//    the power cap (seconds of back-to-back launches): the tile list ordered by groups of 3 / 4 weight panels (GemmTcArgs.ngroup)
//    takes qkv from 168 to 155.5 us and fc1 from 253 to 244 us although the fabric-side bytes barely move (qkv 302 -> 315 MB
//    fetched, fc1 487 -> 427): fewer CUs of an XCD pull the SAME A panel / all W panels at the same moment.
//  * tried and rejected: cache-policy hints on the operand DMAs (`nt` on A: 11.59 -> 11.96 ms/step; `nt` on W: 12.56; both:
//    13.05; `sc0` on A: 11.80) -- the default policy is the best of the five.
//  * tried and rejected: non-temporal (`nt`) output stores so that C does not displace A / W lines in L2 -- qkv 193 -> 185 us,
//    fc1 292 -> 286 us in isolation, but the step is unchanged (11.70 ms): the consumers (attention, fc2) then miss.
//  * tried and rejected: a start-phase offset between neighbouring workgroups of the short-K read-modify GEMM (proj) so that
//    their epilogues' residual traffic does not arrive as one chip-wide burst -- 104 -> 104..112 us, 11.6 -> 11.9 ms/step.
//  * tried and rejected (PMC): rotating the K loop per A row-panel to shorten W's L2 re-use distance -- the L2 hits come
//    from workgroups reading the SAME slices at the SAME time; rotation cut the hit rate from 74 % to 47 % (fc2).
//  * tried and rejected (round 2, code removed in round 3): waiting for every half-tile as late as legal (counted vmcnt(10) in
//    P0 / P1 / P3: five half-tiles in flight instead of three) -- K loop 36.5 k -> 37.9 k ticks; the loop is not feed-bound
//    (timing ablations: DESIGN.md section 4, tools/gemm_stamps.py ABL=...).
//  * tried and rejected (round 2): every second workgroup of an XCD sleeping 0.15 .. 0.65 of a tile time before its first tile
//    (fc2, K = 3072) so that the read-modify epilogues of half the chip meet the other half's K loops instead of each other:
//    306.5 -> 307 .. 313 us.
//  * tried and rejected (round 3, tools/corun_probe.sh): capping the kernel at 224 VGPRs (amdgpu_num_vgpr(112): the argument counts
//    half registers on the unified file) with 224-row tiles, so that 64 registers per SIMD lane stay free and a wave of another
//    stream's small kernel (LayerNorm 40 VGPRs, im2col, the head's row kernels) runs INSIDE the GEMM's CU instead of waiting
//    for a workgroup to leave.  The kernels do co-execute (LayerNorm overlapped with the other lane's GEMM 49 % of its time
//    instead of 29 %), but the cap costs 4 - 19 spilled registers: all GEMMs 10.14 -> 10.68 ms one kernel at a time (+0.54 ms),
//    step 11.75 -> 12.40 ms on the same box (+0.65 ms): co-execution returned nothing on top of what the spills cost (the guest
//    wave takes LDS-DMA / VALU issue slots from the SIMD's two GEMM waves).
//  * tried and rejected (round 3, code removed): an L2 touch-prefetch in the epilogue -- wave 0 reading one dword per 128-byte
//    line of K tiles 2..5 (or 2..11) of the A rows of the tile AFTER next, whose ticket has just arrived, a tile time ahead of
//    their staging: qkv 173-178 -> 181-185 (191) us, fc1 254.5 -> 263-265 (277), step 11.00 -> 11.05-11.18 (11.27) ms.  The lines
//    do not survive a tile time in the XCD's 4 MB L2 next to 32 workgroups' operands and outputs; the touches only add traffic.
//  * XCD-aware tile walk: workgroups b, b+8, ... share an XCD (round-robin dispatch, speed only); group x = b & 7 owns a
//    contiguous chunk of the tile list (tiles of one A row-panel are neighbours).  A workgroup's FIRST tile is static
//    (chunk start + b/8); every further tile is a ticket from the group's counter (a.sched, agent-scope atomic), so a
//    workgroup that starts late -- its CU was still running another stream's kernel: the head of the previous batch
//    and the other backbone lane run concurrently -- simply takes fewer tiles instead of stretching the launch by its
//    delay.  Wave 0 requests the ticket for tile i+2 in K tile nk-2 of tile i (the second tile's before the cold
//    prologue), publishes it through a 4-byte LDS slot at the start of tile i's epilogue (a K tile later: no wait), and
//    all waves read it in K tile 1 of tile i+1, whose K tile nk-2 starts staging it.  The counters clean themselves: the last workgroup of a group to
//    leave resets them.
#include "common.h"
#include "mvf_hip_internal.h"
#include "gemm_tc_epi.h"

#include <atomic>
#include <cstdlib>
#include <mutex>

namespace {
using namespace gemm_tc;

constexpr int BM = 256, BN = 256, ROWB = 128;
constexpr int HALF_BYTES = 128 * ROWB;          // 16 KiB: 128 rows x 64 bf16
constexpr int BUF_BYTES = 4 * HALF_BYTES;       // A0 A1 B0 B1
constexpr int BIAS_OFF = 2 * BUF_BYTES;         // 8 waves x 64 floats behind the two K-tile buffers
constexpr int SLOT_OFF = BIAS_OFF + 8 * 256;    // 4 B: the next tile's ticket, wave 0 -> all waves
// LN fold (consumer side): this wave's 64 ln_c values (private, like the bias) and the tile's 256 (mean, rstd) pairs, the
// latter read by ALL waves and therefore double-buffered by tile parity (the next tile's pairs are staged while slower
// waves may still be in this tile's epilogue)
constexpr int LNC_OFF = SLOT_OFF + 16;
constexpr int LNMR_OFF = LNC_OFF + 8 * 256;
constexpr int SC_OFF = LNMR_OFF + 2 * 2048;     // fp8: two K tiles' scales, [buf][A 256 dwords | W 256 dwords]
constexpr int LDS_BYTES = SC_OFF + 2 * 2048;    // 140 KiB -> one workgroup per CU
// LN fold, consumer side with in-kernel finalize (a.ln_part): the (mean, rstd) slot is single (LNMR_OFF: written in K tile 1 of a
// tile, read in that tile's epilogue, next written a whole K loop's barriers later) and the tile's partial sums
// [ln_ns][256 rows][2] sit behind it, over the second parity slot and the fp8 scale region (never used together): ln_ns <= 12
constexpr int LNP_OFF = LNMR_OFF + 2048;
constexpr int LDS_MAX = 160 * 1024;
constexpr int OFF_A0 = 0, OFF_A1 = HALF_BYTES, OFF_B0 = 2 * HALF_BYTES, OFF_B1 = 3 * HALF_BYTES;

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define WG_BARRIER()              \
  do {                            \
    SCHED_FENCE();                \
    __builtin_amdgcn_s_barrier(); \
    SCHED_FENCE();                \
  } while (0)

// the second barrier of a phase; ablation bit 4 of the stamped build drops it (what do the barriers cost beside the MFMAs?)
#define WG_BARRIER_E()                        \
  do {                                        \
    if constexpr ((ABL & 16) == 0) WG_BARRIER(); \
  } while (0)

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
// acc += W_frag . A_frag on v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3, each lane's 32 k scaled by 2^(byte 0 of its
// scale register - 127)).  Inline asm with the accumulator TIED: through the builtin, hipcc (ROCm 7.2) picks the three-address
// form for 105 of the loop's 128 MFMAs (destination != C) and then needs a second set of 128 accumulator registers -- 250
// spilled registers inside the K loop.  Hazards are the caller's: operands come from LDS reads it has waited for, the scales
// from VALU two wait states earlier (SCALE_FIX), and the accumulators are next read a whole K tile later.
__device__ __forceinline__ void mfma_mx(f32x4_t& acc, const i32x8_t& w, const i32x8_t& x, unsigned sw, unsigned sx) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"
               : "+v"(acc)
               : "v"(w), "v"(x), "v"(sw), "v"(sx));
}

// DBG (diagnostic build only, reached through mvf_gemm_tc_debug_stamps; never on the product path): lane 0 of waves 0
// and 4 stamps s_memtime at kernel start, after the prologue wait, after the K loop and the epilogue of the
// workgroup's first two tiles, and at the end, into a.dbg[block][2][8] (a buffer of its own; no output depends on it).
// LN: the LN-fold extras are compiled in (EPI_STORE / EPI_GELU: consumer side, a.ln_mr / a.ln_c; EPI_RESID: producer side,
// a.xb / a.stats).  A template parameter, not a run-time test: the plain variants keep their register budget (no spills).
// FP8: the operands are MX-fp8 (OCP e4m3 bytes + one E8M0 scale per 32 consecutive k of a row: a.sa / a.sw) and the products
// run on v_mfma_scale_f32_16x16x128_f8f6f4 -- one MFMA per 16x16 tile and K tile instead of two, at twice the cycles and four
// times the k: the same matrix-pipe time per K tile for TWICE the FLOPs, from the same bytes (a K tile is 128-byte rows either
// way, so staging, swizzle and hazards are unchanged).  The K tile's scales (256 + 256 dwords) travel by one more LDS-DMA per
// wave, issued with A1 (same distance to its consumer, so the counted vmcnt(6) still leaves exactly three half-tiles in flight).
// ADD2: EPI_RESID with the second, bf16 addend (a.radd2): an instantiation of its own, one accumulator row per fetch batch,
// so that the plain read-modify epilogues keep their register budget (the run-time form spilled two registers in both)
// BMT: rows of a tile, 256 or 224.  224 = the same schedule with three instead of four 16-row fragments in the second m-quadrant
// of each wave row (wave rows of 112 rows: A1 holds 48 live rows per wave row, its P2 / P3 issue 12 MFMAs instead of 16).  The
// launch picks it where ceil(tiles / workgroups) x BMT is smaller: the N = 768 GEMMs of 50 432 rows are 591 tiles = 2.31 rounds at
// 256 rows and 678 tiles = 2.65 rounds at 224 -- three rounds either way, of 12.5 % less work each.
// F16: the 16-bit operands / outputs are IEEE fp16 (v_mfma_f32_16x16x32_f16, v_cvt_pk_f16_f32) instead of bf16 -- the reference's
// own autocast dtype (CARL_MVF/train.py:113,301): same instruction rate, three more mantissa bits, 5-bit exponent
template <int EPI, bool DBG, bool LN, bool FP8, int ABL = 0, bool ADD2 = false, int BMT = 256, bool F16 = false>   // ABL: timing ablations (dbg_abl)
__global__ __launch_bounds__(512, 2) void gemm_tc256_kernel(GemmTcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(BMT == 256 || BMT == 240 || BMT == 224 || BMT == 208, "tile rows");   // (192: correct, slower -- 11.92 vs 11.48 ms/step)
  // 16-row fragments of the second m-quadrant of wave row 0 / 1 (the first quadrant always has four): 4|4, 4|3, 3|3, 3|2.  The two
  // wave rows of a SIMD share its matrix pipe, so unequal rows cost nothing: the pipe sees 64 + 16 (RTA + RTB) / 2 rows' worth
  constexpr int RTA = (BMT == 256 || BMT == 240) ? 4 : 3;
  constexpr int RTB = BMT == 256 ? 4 : (BMT == 208 ? 2 : 3);
  constexpr int WR0 = 64 + 16 * RTA;        // rows of wave row 0 = first row of wave row 1
  static_assert(WR0 + 64 + 16 * RTB == BMT, "tile rows");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int rt1 = wr ? RTB : RTA;           // this wave's quadrant-1 fragments (wave-uniform)
  unsigned long long stamps[8];
  unsigned long long kst[16];   // DBG: barrier-by-barrier stamps of ONE K tile (a.dbg_kt) of the workgroup's second tile
  int nstamp = 0, nk_st = 0, tiles_done = 0;
#define KSTAMP()                                                                             \
  if constexpr (DBG && (ABL & 8) != 0) {                                                     \
    if (t == a.dbg_kt && tiles_done == 1 && nk_st < 16) {                   \
      unsigned long long t_;                                                                 \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
      kst[nk_st++] = t_;                                                                     \
    }                                                                                        \
  }
#define STAMP()                                                                              \
  if constexpr (DBG) {                                                                       \
    if (nstamp < 7) {                                                                        \
      SCHED_FENCE();                                                                         \
      unsigned long long t_;                                                                 \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
      SCHED_FENCE();                                                                         \
      stamps[nstamp++] = t_;                                                                 \
    }                                                                                        \
  }
  STAMP();

  // ---- static tile walk: XCD x (= blockIdx & 7) owns tiles [start, end); its workgroups take them round-robin ----
  const int nbn = (a.N + BN - 1) / BN;
  const int ntiles = ((a.M + BMT - 1) / BMT) * nbn;
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bpx = (nwg - xcd + 7) >> 3;            // workgroups on this XCD
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int end = start + q8 + (xcd < r8 ? 1 : 0);
  int lid = start + slot;                          // tile being computed
  if (lid >= end) return;                          // whole workgroup (only when tiles are unevenly spread over XCDs)
  // list index -> (row panel, column tile).  a.ngroup > 0 (it divides nbn; the launch checks): column groups outermost
  const int nmt = (a.M + BMT - 1) / BMT;
  auto tile_mn = [&](int tile, int& mt, int& nt) {
    if (a.ngroup > 0) {
      const int per = nmt * a.ngroup, grp = tile / per, rem = tile - grp * per;
      mt = rem / a.ngroup;
      nt = grp * a.ngroup + (rem - mt * a.ngroup);
    } else {
      mt = tile / nbn;
      nt = tile - mt * nbn;
    }
  };
  const int nk = FP8 ? a.K >> 7 : a.K >> 6;        // K tiles of 128 bytes per row (even: K % 128 == 0, fp8: K % 256 == 0)
  // dynamic tickets only when this group has more tiles than workgroups (then no workgroup of it returned above)
  const bool dyn = a.sched != nullptr && nk >= 4 && start + bpx < end;
  unsigned* const cnt = a.sched + xcd;
  // The ticket is a compiler-visible atomic (hipcc counts the VMEM operations issued after it and waits with an exact
  // vmcnt(N) before the value is used); an inline-asm atomic would leave a register the compiler believes ready while
  // the hardware has yet to write it.  It is requested where a whole epilogue (or the cold prologue) passes before its use.
  unsigned ticket;   // only ever read by the lane that wrote it
  auto fetch_ticket = [&]() {
    if (wave == 0 && lane == 0) ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto publish_ticket = [&]() {   // before a workgroup barrier that precedes K tile 1 of the tile that needs it
    if (wave == 0 && lane == 0) *(volatile __attribute__((address_space(3))) unsigned*)LDS_PTR(smem + SLOT_OFF) = ticket;
    WAIT_LGKM(0);
  };

  // ---- LDS-DMA source pointers of the tile being STAGED (runs up to two K tiles ahead of the compute side) ----
  // per half-tile every wave issues 2 pieces of 1 KiB = 8 rows x 128 B: piece p = i*8 + wave covers half-tile rows
  // p*8 .. p*8+7; lane -> (row = p*8 + lane/8, physical chunk = lane%8)
  const int prow = lane >> 3;
  const int lchunk = (lane & 7) ^ prow;  // logical 16-B chunk fetched into physical chunk lane%8 (row & 7 == prow)
  // 32-bit byte offsets from a.A / a.W (the launch refuses operands of 4 GiB or more): the DMA then takes the uniform
  // base in SGPRs plus one VGPR per piece -- half the address registers of full pointers
  unsigned asrc[2][2];                   // [half][piece]
  unsigned wsrc[2][2];
  constexpr unsigned ESZ = FP8 ? 1u : 2u;
  // fp8: this wave's 64 scale dwords of a K tile -- waves 0-3: rows tm0 + 64 w .. of A (a.sa[kt][M]), waves 4-7: rows
  // tn0 + 64 (w - 4) .. of W (a.sw[kt][N])
  unsigned ssrc = 0;
  const unsigned* const sbase = wave < 4 ? a.sa : a.sw;
  const unsigned sstride = wave < 4 ? (unsigned)a.M : (unsigned)a.N;
  auto set_sources = [&](int tile) {
    int tmt, tnt;
    tile_mn(tile, tmt, tnt);
    const int tm0 = tmt * BMT, tn0 = tnt * BN;
    if constexpr (FP8)
      ssrc = wave < 4 ? (unsigned)min(tm0 + wave * 64 + lane, a.M - 1) : (unsigned)min(tn0 + (wave - 4) * 64 + lane, a.N - 1);
    // stacked batches (split-K weight gradients): the tile's batch picks its own row block of W
    const int wb0 = a.batch_rows > 0 ? (tm0 / a.batch_rows) * a.w_batch_rows : 0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = (i * 8 + wave) * 8 + prow;                 // row inside the half-tile, 0..127
        // A: wave row r>>6, m-quadrant h (quadrant 1 of a wave row with fewer than four fragments: the rows past its last
        // fragment are not part of the tile; they re-read that fragment's last row)
        const int am = (r >> 6) * WR0 + h * 64 + min(r & 63, h ? ((r >> 6) ? RTB : RTA) * 16 - 1 : 63);
        const int wn = (r >> 5) * 64 + h * 32 + (r & 31);        // W: wave column r>>5, n-quadrant h
        int gm = min(tm0 + am, a.M - 1);
        if constexpr (DBG) gm &= (int)a.dbg_rowmask;
        const int gn = wb0 + min(tn0 + wn, a.N - 1);
        // 24-bit multiply (rows and row bytes < 2^24, checked by the launch): one v_mad_u32_u24 -- a full 32-bit
        // product goes through v_mad_u64_u32, whose don't-care high addend register hipcc shares with the ticket's
        asrc[h][i] = __umul24((unsigned)gm, (unsigned)a.lda * ESZ) + lchunk * 16;
        wsrc[h][i] = __umul24((unsigned)gn, (unsigned)a.ldw * ESZ) + lchunk * 16;
      }
  };
  set_sources(lid);
  const int piece0 = wave * 1024, piece1 = (8 + wave) * 1024;
  auto stage = [&](int buf, int off, const char* gbase, const unsigned (&src)[2], int kt) {
    if constexpr ((ABL & 4) != 0) return;      // ablation: no operand DMAs
    char* dst = smem + buf * BUF_BYTES + off;
    const unsigned koff = (unsigned)kt * ROWB;
    __builtin_amdgcn_global_load_lds(GLB_PTR(gbase + (size_t)(src[0] + koff)), LDS_PTR(dst + piece0), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GLB_PTR(gbase + (size_t)(src[1] + koff)), LDS_PTR(dst + piece1), 16, 0, 0);
  };
  // fp8: the scales of K tile kt -> scale buffer `buf` (A rows first, then W rows; lane-linear, 4 B per lane)
  auto stage_scales = [&](int buf, int kt) {
    if constexpr (FP8)
      __builtin_amdgcn_global_load_lds(GLB_PTR(sbase + (size_t)kt * sstride + ssrc),
                                       LDS_PTR(smem + SC_OFF + buf * 2048 + wave * 256), 4, 0, 0);
  };
  // this wave's 64 bias values -> its private 256 B of LDS, one LDS-DMA (4 B per lane)
  char* sbias = smem + BIAS_OFF + wave * 256;
  char* slnc = smem + LNC_OFF + wave * 256;
  // parity of the workgroup's current tile.  A tile's bias / ln_c / (mean, rstd) are staged right after the previous tile's
  // epilogue of THIS wave, when the parity has just flipped: slot tpar is read in this tile's epilogue only, and was last read
  // two tiles ago (every wave has passed a K loop's barriers since)
  int tpar = 0;
  constexpr bool kLnCons = LN && (EPI == EPI_STORE || EPI == EPI_GELU);
  const bool lnp = kLnCons && a.ln_part != nullptr;    // (mean, rstd) from the partial sums, inside the kernel
  auto stage_bias = [&](int tile) {
    int tmt, tnt;
    tile_mn(tile, tmt, tnt);
    if (a.bias != nullptr) {
      const int n = min(tnt * BN + wc * 64 + lane, a.N - 1);
      __builtin_amdgcn_global_load_lds(GLB_PTR(a.bias + n), LDS_PTR(sbias), 4, 0, 0);
    }
    if constexpr (LN && (EPI == EPI_STORE || EPI == EPI_GELU)) {
      {
        const int n = min(tnt * BN + wc * 64 + lane, a.N - 1);
        __builtin_amdgcn_global_load_lds(GLB_PTR(a.ln_c + n), LDS_PTR(slnc), 4, 0, 0);
        // rows m0 .. m0+255 as 512 consecutive floats; this wave copies floats [64 w, 64 w + 64) = rows m0 + 32 w ..
        if (!lnp) {
          const int fi = wave * 64 + lane;
          const int row = min(tmt * BMT + (fi >> 1), a.M - 1);
          __builtin_amdgcn_global_load_lds(GLB_PTR(a.ln_mr + (size_t)row * 2 + (fi & 1)),
                                           LDS_PTR(smem + LNMR_OFF + tpar * 2048 + wave * 256), 4, 0, 0);
        }
      }
    }
  };
  // in-kernel finalize, step 1 (K tile 0 of a tile, after its first barrier: every wave has left the previous tile's K tile 1,
  // the region's only reader): the tile's partial sums, 2 * ln_ns pieces of 1 KiB = 128 rows x (sum, sum of squares) of one
  // 64-column slice; older than the three half-tiles the K tile's vmcnt(6) leaves in flight, so landed after its P3 barrier
  auto stage_partials = [&](int m0c) {
    for (int p = wave; p < 2 * a.ln_ns; p += 8) {
      const int row = min(m0c + (p & 1) * 128 + lane * 2, a.M - 2);      // (M is even: checked by the launch)
      __builtin_amdgcn_global_load_lds(GLB_PTR(a.ln_part + ((size_t)(p >> 1) * a.M + row) * 2), LDS_PTR(smem + LNP_OFF + p * 1024), 16, 0,
                                       0);
    }
  };
  // step 2 (K tile 1, after its first barrier): wave w turns rows 32 w .. 32 w + 31 into (mean, rstd), one lane per row, slices
  // summed in order -- the arithmetic of ln_stats_finalize_kernel (vit_misc.hip), bit for bit
  auto finalize_partials = [&]() {
    if (lane < 32) {
      const int r = wave * 32 + lane;
      const char* p = smem + LNP_OFF + r * 8;
      float s1 = 0.f, s2 = 0.f;
      for (int sl = 0; sl < a.ln_ns; ++sl) {
        const float2 v = *reinterpret_cast<const float2*>(p + sl * 2048);
        s1 += v.x;
        s2 += v.y;
      }
      const float mean = s1 * a.ln_inv_d;
      const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * a.ln_inv_d), 0.f);
      *reinterpret_cast<float2*>(smem + LNMR_OFF + r * 8) = make_float2(mean, 1.0f / sqrtf(var + a.ln_eps));
    }
  };

  // ---- fragment read offsets: lane (frow, fgrp) reads row frow of a 16-row tile, 16-B chunk (ks*4 + fgrp) ^ (frow&7)
  const int frow = lane & 15, fgrp = lane >> 4;
  const int choff = (fgrp ^ (frow & 7)) << 4;                   // k-step 0; k-step 1 is choff ^ 64
  const int a_rd = (wr * 64 + frow) * ROWB + choff;
  const int b_rd = (wc * 32 + frow) * ROWB + choff;

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- cold prologue (first tile of the workgroup only): K tile 0 complete, K tile 1 except A1 ----
  if (dyn) fetch_ticket();
  stage_bias(lid);
  stage(0, OFF_B0, a.W, wsrc[0], 0);
  stage(0, OFF_A0, a.A, asrc[0], 0);
  stage(0, OFF_B1, a.W, wsrc[1], 0);
  stage(0, OFF_A1, a.A, asrc[1], 0);
  stage_scales(0, 0);
  stage(1, OFF_B0, a.W, wsrc[0], 1);
  stage(1, OFF_A0, a.A, asrc[0], 1);
  stage(1, OFF_B1, a.W, wsrc[1], 1);
  WAIT_VMCNT(6);
  if (dyn) publish_ticket();   // the second tile of this workgroup
  WG_BARRIER();
  STAMP();

  bf16x8_t af[4][2] = {}, bf0[2][2] = {}, bf1[2][2] = {};
  // fp8: a fragment is ONE 32-byte operand (8 consecutive registers), filled by two 16-byte LDS reads into its halves
  i32x8_t afq[4], bq0[2], bq1[2];
  unsigned sfa[4], sf0[2], sf1[2];   // fp8: the E8M0 scale of each fragment's 32-k block, in byte 0
  // fp8: lane (row, g) of v_mfma_scale_f32_16x16x128_f8f6f4 holds k = 16 g .. 16 g + 15 and 64 + 16 g .. 64 + 16 g + 15 in the
  // two halves of its 32-byte operand -- the SAME two 16-byte chunks (g, 4 + g) of the 128-byte row as the bf16 kernel's two
  // k-steps -- while the scale it supplies is that of MX block g (k = 32 g .. 32 g + 31).  Measured (tools/fp8_debug*.py):
  // with 32 consecutive bytes per lane every product is still right (both operands permute k alike) but each block scale
  // lands on another block's data.
  constexpr int KS1 = 64;
  // fp8: one LDS address each for this lane's A-row / W-row scale dwords (K-tile buffer, quadrant and tile are immediates)
  const char* const sa_rd = smem + SC_OFF + (wr * 128 + frow) * 4;
  const char* const sb_rd = smem + SC_OFF + 1024 + (wc * 64 + frow) * 4;
  const unsigned sshift = 8u * fgrp;

// fp8: the scale dwords of an A quadrant's four row tiles.  In P0 they are read BEFORE the scheduling fence, with the B
// operands, so that exactly the eight 16-byte A reads follow it whatever hipcc merges (it pairs the dword reads into
// ds_read2_b32): the lgkmcnt(8) in front of P0's first barrier then retires every B read (WAR rule of B0, file header)
#define LOAD_A_SCALES(OFF)                                                                                   \
  if constexpr (FP8) {                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
      sfa[i] = *reinterpret_cast<const unsigned*>(sa_rd + sbuf_off + (((OFF) == OFF_A1 ? 64 : 0) + i * 16) * 4);  \
  }
#define LOAD_A(OFF)                                                                                          \
  if constexpr ((ABL & 2) == 0)                                                                              \
  _Pragma("unroll") for (int i = 0; i < ((OFF) == OFF_A1 ? RTA : 4); ++i) { /* (wave row 1 may read one fragment it does not use) */ \
    if constexpr (FP8) {                                                                                     \
      afq[i].lo = *reinterpret_cast<const i32x4_t*>(base + (OFF) + a_rd + i * 16 * ROWB);                    \
      afq[i].hi = *reinterpret_cast<const i32x4_t*>(base + (OFF) + (a_rd ^ KS1) + i * 16 * ROWB);            \
    } else {                                                                                                 \
      af[i][0] = *reinterpret_cast<const bf16x8_t*>(base + (OFF) + a_rd + i * 16 * ROWB);                    \
      af[i][1] = *reinterpret_cast<const bf16x8_t*>(base + (OFF) + (a_rd ^ KS1) + i * 16 * ROWB);            \
    }                                                                                                        \
  }
#define LOAD_B(BF, BQ, SF, OFF)                                                                              \
  if constexpr ((ABL & 2) == 0)                                                                              \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                             \
    if constexpr (FP8) {                                                                                     \
      BQ[j].lo = *reinterpret_cast<const i32x4_t*>(base + (OFF) + b_rd + j * 16 * ROWB);                     \
      BQ[j].hi = *reinterpret_cast<const i32x4_t*>(base + (OFF) + (b_rd ^ KS1) + j * 16 * ROWB);             \
      SF[j] = *reinterpret_cast<const unsigned*>(sb_rd + sbuf_off + (((OFF) == OFF_B1 ? 32 : 0) + j * 16) * 4);  \
    } else {                                                                                                 \
      BF[j][0] = *reinterpret_cast<const bf16x8_t*>(base + (OFF) + b_rd + j * 16 * ROWB);                    \
      BF[j][1] = *reinterpret_cast<const bf16x8_t*>(base + (OFF) + (b_rd ^ KS1) + j * 16 * ROWB);            \
    }                                                                                                        \
  }
// fp8: a freshly loaded scale dword holds the K tile's four block scales; this lane's block is fgrp -> byte 0.  Run after the
// phase's lgkmcnt(0); the s_nop covers the VALU-write -> MFMA-operand wait states, which hipcc does not pad for inline asm.
#define SCALE_FIX(ARR, N)                                                                                    \
  if constexpr (FP8) {                                                                                       \
    _Pragma("unroll") for (int q = 0; q < (N); ++q) ARR[q] = __builtin_amdgcn_ubfe(ARR[q], sshift, 8);       \
    asm volatile("s_nop 1");                                                                                 \
  }
#define MFMA_QUAD(MQ, NQ, BF, BQ, SF)                                                                        \
  __builtin_amdgcn_s_setprio(1);                                                                             \
  if constexpr ((ABL & 1) != 0) {                                                                            \
  } else if constexpr (FP8) {                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
      mfma_mx(acc[(MQ) * 4 + i][(NQ) * 2 + j], BQ[j], afq[i], SF[j], sfa[i]);                                \
  } else {                                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                          \
    _Pragma("unroll") for (int i = 0; i < ((MQ) == 1 ? RTA : 4); ++i)                                         \
      if (RTA == RTB || (MQ) == 0 || i < rt1) /* unequal wave rows: wave row 1 skips its missing fragment (wave-uniform) */ \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
      acc[(MQ) * 4 + i][(NQ) * 2 + j] = mfma16x16x32<F16>(BF[j][ks], af[i][ks], acc[(MQ) * 4 + i][(NQ) * 2 + j]);     \
  }                                                                                                          \
  __builtin_amdgcn_s_setprio(0);

// K tile T of the current output tile, read from LDS buffer BUF.  The staged K tile index is T+1 / T+2 of the current
// tile, or -- once `kwrap` = -nk, i.e. the issue-side pointers have moved on -- K tile 0 / 1 of the NEXT output tile.
#define CAN_ISSUE(D) ((t + (D) < nk) || have_next)
#define K_TILE(BUF, T)                                                                                       \
  {                                                                                                          \
    const char* base = smem + (BUF) * BUF_BYTES;                                                             \
    constexpr int sbuf_off = (BUF) * 2048;                                                                   \
    const int t = (T);                                                                                       \
    KSTAMP();                                                                                                \
    /* P0 */                                                                                                 \
    LOAD_B(bf0, bq0, sf0, OFF_B0);                                                                                \
    LOAD_A_SCALES(OFF_A0);                                                                                   \
    SCHED_FENCE();                                                                                           \
    LOAD_A(OFF_A0);                                                                                          \
    if (CAN_ISSUE(1)) {                                                                                      \
      stage((BUF) ^ 1, OFF_A1, a.A, asrc[1], t + 1 + kwrap);                                                      \
      stage_scales((BUF) ^ 1, t + 1 + kwrap);                                                                \
    }                                                                                                        \
    SCHED_FENCE();                                                                                           \
    WAIT_LGKM(8); /* everything in front of the eight A reads is back: the B reads (and the fp8 scales) */   \
    WG_BARRIER();                                                                                            \
    KSTAMP();                                                                                                \
    if constexpr (kLnCons) {                                                                                 \
      if ((BUF) == 0 && lnp && t == 0) stage_partials(m0);                                                   \
      if ((BUF) == 1 && lnp && t == 1) finalize_partials();                                                  \
    }                                                                                                        \
    unsigned tkv = 0;                                                                                        \
    if (dyn && t == 1) tkv = *(volatile __attribute__((address_space(3))) unsigned*)LDS_PTR(smem + SLOT_OFF); \
    WAIT_LGKM(0);                                                                                            \
    if (dyn && t == 1) {                                                                                     \
      lid_next = start + bpx + (int)__builtin_amdgcn_readfirstlane(tkv);                                     \
      have_next = lid_next < end;                                                                            \
    }                                                                                                        \
    SCALE_FIX(sf0, 2);                                                                                       \
    SCALE_FIX(sfa, 4);                                                                                       \
    SCHED_FENCE();                                                                                           \
    MFMA_QUAD(0, 0, bf0, bq0, sf0);                                                                          \
    WG_BARRIER_E();                                                                                          \
    KSTAMP();                                                                                                \
    if ((BUF) == 0 && t == nk - 2 && have_next) { /* from here on the NEXT output tile is staged */          \
      set_sources(lid_next);                                                                                 \
      kwrap = -nk;                                                                                           \
    }                                                                                                        \
    /* P1 */                                                                                                 \
    LOAD_B(bf1, bq1, sf1, OFF_B1);                                                                                \
    if (CAN_ISSUE(2)) stage((BUF), OFF_B0, a.W, wsrc[0], t + 2 + kwrap);                                     \
    WG_BARRIER();                                                                                            \
    KSTAMP();                                                                                                \
    WAIT_LGKM(0);                                                                                            \
    SCALE_FIX(sf1, 2);                                                                                       \
    SCHED_FENCE();                                                                                           \
    MFMA_QUAD(0, 1, bf1, bq1, sf1);                                                                          \
    WG_BARRIER_E();                                                                                          \
    KSTAMP();                                                                                                \
    /* P2 */                                                                                                 \
    LOAD_A_SCALES(OFF_A1);                                                                                   \
    LOAD_A(OFF_A1);                                                                                          \
    if (CAN_ISSUE(2)) stage((BUF), OFF_A0, a.A, asrc[0], t + 2 + kwrap);                                          \
    WG_BARRIER();                                                                                            \
    KSTAMP();                                                                                                \
    WAIT_LGKM(0);                                                                                            \
    SCALE_FIX(sfa, 4);                                                                                       \
    SCHED_FENCE();                                                                                           \
    MFMA_QUAD(1, 1, bf1, bq1, sf1);                                                                               \
    WG_BARRIER_E();                                                                                          \
    KSTAMP();                                                                                                \
    /* P3 */                                                                                                 \
    if (CAN_ISSUE(2)) {                                                                                      \
      stage((BUF), OFF_B1, a.W, wsrc[1], t + 2 + kwrap);                                                          \
      WAIT_VMCNT(6);                                                                                         \
    } else {                                                                                                 \
      WAIT_VMCNT(0);                                                                                         \
    }                                                                                                        \
    if (dyn && (BUF) == 0 && t == nk - 2 && have_next) fetch_ticket(); /* tile after next, see epilogue */   \
    WG_BARRIER();                                                                                            \
    KSTAMP();                                                                                                \
    MFMA_QUAD(1, 0, bf0, bq0, sf0);                                                                          \
    WG_BARRIER_E();                                                                                          \
    KSTAMP();                                                                                                \
  }

  for (;;) {
    int lid_next = dyn ? 0x7fffffff : lid + bpx;   // dyn: known from K tile 1 on (see K_TILE)
    bool have_next = lid_next < end;
    int cmt, cnt_;
    tile_mn(lid, cmt, cnt_);
    const int m0 = cmt * BMT, n0 = cnt_ * BN;   // compute-side tile
    int kwrap = 0;
    if (wr == 1) WG_BARRIER();  // stagger: wave row 1 runs one barrier behind wave row 0

    for (int kt = 0; kt < nk; kt += 2) {
      K_TILE(0, kt);
      K_TILE(1, kt + 1);
    }
    if (wr == 0) WG_BARRIER();  // re-align the wave rows: both run the epilogue in the same interval
    if constexpr (DBG) ++tiles_done;
    if constexpr (FP8) asm volatile("s_nop 15\n\ts_nop 3");   // last asm MFMA's result -> first VALU read (16-pass: 18 states)
    STAMP();

    // ---- epilogue: two adjacent 16x16 tiles at a time (gemm_tc_epi.h epilogue_pair_bf16); bias from LDS ----
    float4 bj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bj[j] = a.bias != nullptr ? *reinterpret_cast<const float4*>(sbias + (j * 16 + fgrp * 4) * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
    // the ticket requested in K tile nk-2 has had a K tile's time; hipcc's vmcnt(0) in front of the bias reads (or in
    // front of this write) finds it done.  Every wave is past its K tile 1 read of the slot.
    if (dyn && have_next) publish_ticket();
    constexpr bool kReadModify = EPI == EPI_RESID || EPI == EPI_PATCH;
    constexpr bool kLnConsumer = kLnCons;
    constexpr bool kLnProducer = LN && EPI == EPI_RESID;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gj[4] = {z4, z4, z4, z4};
    if constexpr (EPI == EPI_RESID) {
      if (a.ls != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + wc * 64 + j * 16 + fgrp * 4;
          if (n < a.N) gj[j] = *reinterpret_cast<const float4*>(a.ls + n);
        }
      }
    }
    // read-modify epilogues: the addends of EB accumulator rows (4*EB 16-byte loads per lane) are in flight together,
    // and batch b+1 is requested before batch b is stored (so the wait for b+1 overlaps b's stores)
#ifndef MVF_EPI_EB
#define MVF_EPI_EB 2
#endif
#ifndef MVF_EPI_PIPE
#define MVF_EPI_PIPE 0
#endif
    constexpr int EB = ADD2 ? 1 : MVF_EPI_EB, NB = 8 / EB;
    float4 add[2][EB][4];
    uint2 add2[2][EB][4];      // (both indexed by the batch parity in the pipelined form, slot 0 otherwise)
    auto prefetch = [&](int b) {
      if constexpr (kReadModify) {
#pragma unroll
        for (int ii = 0; ii < EB; ++ii) {
          const int i = b * EB + ii;
          if (i >= 4 + rt1) continue;            // (a fragment this wave row does not have)
          const int m = m0 + wr * WR0 + (i >> 2) * 64 + (i & 3) * 16 + frow;
          epilogue_prefetch<EPI>(a, m, m < a.M, n0 + wc * 64, fgrp, add[MVF_EPI_PIPE ? (b & 1) : 0][ii]);
          if constexpr (ADD2) epilogue_prefetch2(a, m, m < a.M, n0 + wc * 64, fgrp, add2[MVF_EPI_PIPE ? (b & 1) : 0][ii]);
        }
      }
    };
    if constexpr (kLnConsumer) {
      // LN-fold consumer: acc <- rstd * (acc - mean * c) in place, FIRST and for all 128 accumulators (the row's (mean, rstd)
      // from this tile's LDS slot, the wave's 64 ln_c values from its own slot), then the plain epilogue with bias = d.
      // Done inside the store loop instead, the extra live values (c, mean/rstd) sat on top of the GELU temporaries and the
      // fc1 epilogue took twice as long.
      const char* smr = smem + LNMR_OFF + (lnp ? 0 : tpar * 2048) + (wr * WR0 + frow) * 8;
      // Row outermost: the wave's 16 ln_c values stay live (the K loop's fragments are dead here), a row's (mean, rstd) is read ONCE and
      // the next row's pair is in flight while this row is computed.  (Column tile outermost, as first written -- 4 ln_c values live,
      // the pair re-read per (row, column tile) -- hipcc put a full LDS wait in front of each of the 32 groups of 8 instructions:
      // ~5 k cycles per tile, +11 % on the qkv GEMM; round 6.)  Per element the same two operations in the same order: same bits.
      float4 cj[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) cj[j] = *reinterpret_cast<const float4*>(slnc + (j * 16 + fgrp * 4) * 4);
      float2 mrn = *reinterpret_cast<const float2*>(smr);
#pragma unroll
      for (int i = 0; i < 4 + RTA; ++i) {      // (a fragment wave row 1 lacks: unused accumulators, rows inside the slot)
        const float2 mr = mrn;
        if (i + 1 < 4 + RTA) mrn = *reinterpret_cast<const float2*>(smr + (((i + 1) >> 2) * 64 + ((i + 1) & 3) * 16) * 8);
        const float nm = -mr.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // mul_rounded: a product of its own rounding, never contracted with the bias add of the store loop below -- whether hipcc
          // fused the two depended on the instantiation (tile height, what else the kernel carries: the unequal-wave-row variants
          // stopped fusing the fragment wave row 1 lacks), and with it the bit-for-bit agreement of the kernels (gemm_tc.hip's
          // epilogue4 spells the same two roundings)
          acc[i][j][0] = mul_rounded(mr.y, fmaf(nm, cj[j].x, acc[i][j][0]));
          acc[i][j][1] = mul_rounded(mr.y, fmaf(nm, cj[j].y, acc[i][j][1]));
          acc[i][j][2] = mul_rounded(mr.y, fmaf(nm, cj[j].z, acc[i][j][2]));
          acc[i][j][3] = mul_rounded(mr.y, fmaf(nm, cj[j].w, acc[i][j][3]));
        }
      }
      SCHED_FENCE();   // keep the pre-pass out of the store loop (its values would pile onto the GELU temporaries)
    }
    {
    if (MVF_EPI_PIPE) prefetch(0);
#pragma unroll
    for (int ih = 0; ih < NB; ++ih) {
      if (!MVF_EPI_PIPE) prefetch(ih);
      else if (ih + 1 < NB) prefetch(ih + 1);
      float4(&addb)[EB][4] = add[MVF_EPI_PIPE ? (ih & 1) : 0];
#pragma unroll
      for (int ii = 0; ii < EB; ++ii) {
        const int i = ih * EB + ii;
        if (i >= 4 + rt1) continue;
        const int m = m0 + wr * WR0 + (i >> 2) * 64 + (i & 3) * 16 + frow;
        if constexpr (ADD2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) addb[ii][j] = add_16x4<F16>(addb[ii][j], add2[MVF_EPI_PIPE ? (ih & 1) : 0][ii][j]);
        }
        if constexpr (EPI == EPI_GELU_Q) {
          // the wave's 64 columns of row m = two MX blocks: quantise each, then one 2-byte store of both scales into the
          // row's dword of K tile (n0 + 64 wc) / 128 of the NEXT GEMM (bytes 2 (wc & 1), 2 (wc & 1) + 1)
          const int nb = n0 + wc * 64;
          const bool ok = m < a.M && nb < a.N;
          const unsigned s0 = epilogue_pair_gelu_q(a, m, ok, nb, fgrp, acc[i][0], acc[i][1], bj[0], bj[1]);
          const unsigned s1 = epilogue_pair_gelu_q(a, m, ok, nb + 32, fgrp, acc[i][2], acc[i][3], bj[2], bj[3]);
          if (ok && fgrp == 0)
            *reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(a.csc + (size_t)(nb >> 7) * a.M + m) + (wc & 1) * 2) =
                (unsigned short)(s0 | (s1 << 8));
        } else if constexpr (!kLnProducer) {
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            const int nb = n0 + wc * 64 + jp * 32;
            // N % 32 == 0: a tile pair is in range or out as a whole
            epilogue_pair_bf16<EPI, F16>(a, m, m < a.M && nb < a.N, nb, fgrp, acc[i][2 * jp], acc[i][2 * jp + 1], bj[2 * jp],
                                    bj[2 * jp + 1], kReadModify ? addb[ii][2 * jp] : z4,
                                    kReadModify ? addb[ii][2 * jp + 1] : z4, gj[2 * jp], gj[2 * jp + 1]);
          }
        } else {
          const float2 mr = make_float2(0.f, 1.f);   // producer side (EPI_RESID): xb + row sums
          float s1 = 0.f, s2 = 0.f;
          if constexpr (FP8) {
            // fp8: xb = MX-fp8(x_new), the wave's 64 columns = two blocks; one 2-byte store of both scales as in EPI_GELU_Q
            const int nbw = n0 + wc * 64;
            const bool okw = m < a.M && nbw < a.N;
            const unsigned q0 = epilogue_pair_resid_q(a, m, okw, nbw, fgrp, acc[i][0], acc[i][1], bj[0], bj[1], addb[ii][0], addb[ii][1],
                                                      gj[0], gj[1], s1, s2);
            const unsigned q1 = epilogue_pair_resid_q(a, m, okw, nbw + 32, fgrp, acc[i][2], acc[i][3], bj[2], bj[3], addb[ii][2],
                                                      addb[ii][3], gj[2], gj[3], s1, s2);
            if (okw && fgrp == 0)
              *reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(a.csc + (size_t)(nbw >> 7) * a.M + m) + (wc & 1) * 2) =
                  (unsigned short)(q0 | (q1 << 8));
          } else {
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            const int nb = n0 + wc * 64 + jp * 32;
            const float4 c0 = z4, c1 = z4;
            epilogue_pair_bf16_ln<EPI, F16>(a, m, m < a.M && nb < a.N, nb, fgrp, acc[i][2 * jp], acc[i][2 * jp + 1], bj[2 * jp],
                                       bj[2 * jp + 1], kReadModify ? addb[ii][2 * jp] : z4,
                                       kReadModify ? addb[ii][2 * jp + 1] : z4, gj[2 * jp], gj[2 * jp + 1], mr, c0, c1, s1, s2);
          }
          }
          if constexpr (kLnProducer) {
            if (a.stats != nullptr) {   // kernel argument: every lane runs the cross-lane sums
              s1 = row_quad_sum(s1);
              s2 = row_quad_sum(s2);
              const int slice = (n0 >> 6) + wc;
              // slice-major [N/64][M][2]: the 16 rows of a wave-store are 128 contiguous bytes (row-major, 8-byte stores 96 B
              // apart, cost 40 k cycles per tile: every one its own partial-line write)
              if (fgrp == 0 && m < a.M && slice * 64 < a.N)
                *reinterpret_cast<float2*>(a.stats + ((size_t)slice * a.M + m) * 2) = make_float2(s1, s2);
            }
          }
        }
      }
    }
    }
    STAMP();
    if (!have_next) break;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    lid = lid_next;
    if constexpr (LN) tpar ^= 1;
    WAIT_LGKM(0);        // this wave's bias reads are done before its bias slot is re-staged
    stage_bias(lid);     // older than every DMA the coming vmcnt(6) waits leave in flight
  }
  // every workgroup of a dynamic group ends on one refused ticket, which it has seen: all of the group's ticket
  // requests are complete once all have counted themselves out, and the last one zeroes the pair for the next launch
  if (dyn && wave == 0 && lane == 0) {
    unsigned* const gone = a.sched + 8 + xcd;
    const unsigned old = __hip_atomic_fetch_add(gone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)bpx - 1) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gone, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

#undef K_TILE
#undef CAN_ISSUE
#undef MFMA_QUAD
#undef LOAD_A
#undef LOAD_B

  if constexpr (DBG) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    while (nstamp < 7) stamps[nstamp++] = 0ull;
    {
      SCHED_FENCE();
      unsigned long long t_;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
      SCHED_FENCE();
      stamps[7] = t_;
    }
    if (a.dbg != nullptr && lane == 0 && (wave == 0 || wave == 4)) {
      for (int q = 0; q < 8; ++q) a.dbg[((size_t)blockIdx.x * 2 + (wave >> 2)) * 8 + q] = stamps[q];
      if constexpr ((ABL & 8) != 0) {   // second region of the buffer: [gridDim.x][2][16]
        unsigned long long* k2 = a.dbg + (size_t)gridDim.x * 16;
        for (int q = 0; q < 16; ++q) k2[((size_t)blockIdx.x * 2 + (wave >> 2)) * 16 + q] = q < nk_st ? kst[q] : 0ull;
      }
    }
  }
#undef STAMP
#undef KSTAMP
}

// > 0: column tiles per weight-panel group of the tile list (GemmTcArgs.ngroup); 0 = row panel major
int g_ngroup = [] { const char* e = getenv("MVF_GEMM_NGROUP"); return e ? atoi(e) : -1; }();

// > 0: CUs the persistent launch may take, see mvf_gemm_tc_set_cus (MVF_GEMM_CUS: initial value, A/B measurements)
int g_cu_budget = [] { const char* e = getenv("MVF_GEMM_CUS"); return e ? atoi(e) : 0; }();

int num_cus() {
  if (g_cu_budget > 0) return g_cu_budget;
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// Scheduler counters: a ring of 1024 x 16 zeroed words, one slot per persistent launch (a launch leaves its slot zeroed;
// two launches could only share a slot if 1024 later launches were enqueued while the first was still running).
unsigned* sched_slot() {
  static unsigned* ring = nullptr;
  static std::atomic<unsigned> next{0};
  static std::once_flag once;
  std::call_once(once, [] {
    unsigned* p = nullptr;
    if (hipMalloc(&p, 1024 * 16 * sizeof(unsigned)) == hipSuccess && hipMemset(p, 0, 1024 * 16 * sizeof(unsigned)) == hipSuccess)
      ring = p;
  });
  return ring == nullptr ? nullptr : ring + (size_t)(next.fetch_add(1) & 1023u) * 16;
}

// CUs a persistent launch leaves free where that costs no tile round (see launch_bm); MVF_GEMM_SPARE / mvf_gemm_tc_set_spare
// Measured on the pipelined training step (profiles/r05/gemm_spare.txt, gemm_spare2.txt; the head's row-chain launches are 24 workgroups):
// 0 / 16 / 24 spare CUs no gain, 32: -0.15 .. -0.28 ms per step, 40 .. 256: within 0.05 of that; forwards alone +0.08 .. +0.18 ms.
int g_spare = [] { const char* e = getenv("MVF_GEMM_SPARE"); return e ? atoi(e) : 32; }();

template <int EPI, bool DBG = false, bool LN = false, bool FP8 = false, int ABL = 0, bool ADD2 = false, int BMT = 256, bool F16 = false>
int launch_bm(const GemmTcArgs& a0, bool persistent, hipStream_t st) {
  GemmTcArgs a = a0;
  {   // weight-panel grouping of the tile list (MVF_GEMM_NGROUP / mvf_gemm_tc_set_ngroup; plain, unbatched launches only)
    const int nbn_ = (a.N + BN - 1) / BN;
    // -1 = per launch: groups of 4 column tiles where 4 divides a count of 8 or more, else of 3 where 3 divides a count of 6 or
    // more, else none.  Sustained (seconds of back-to-back launches at the power cap, profiles/r04/ngroup_ab.txt), M = 50 432:
    // qkv (9 column tiles) 167.8 - 169.2 -> 155.5 us with groups of 3; fc1 (12) 253.2 -> 250.7 (3) / 244.1 (4) / 249.0 (6)
    int g = g_ngroup;
    if (g < 0) g = (nbn_ >= 8 && nbn_ % 4 == 0) ? 4 : ((nbn_ >= 6 && nbn_ % 3 == 0) ? 3 : 0);
    a.ngroup = (g > 0 && persistent && a.batch_rows == 0 && nbn_ > g && nbn_ % g == 0) ? g : 0;
  }
  static const bool force_static = getenv("MVF_GEMM_STATIC") != nullptr;   // A/B measurements only
  a.sched = persistent && !force_static ? sched_slot() : nullptr;
  static uint64_t attr_set = 0;      // per device
  constexpr bool lncons = LN && (EPI == EPI_STORE || EPI == EPI_GELU);
  if (mvf_ensure_lds(reinterpret_cast<const void*>(gemm_tc256_kernel<EPI, DBG, LN, FP8, ABL, ADD2, BMT, F16>),
                     lncons ? LDS_MAX : LDS_BYTES, attr_set) != MVF_OK)
    return MVF_ERR_UNSUPPORTED;      // the ln_part form: run_blocks falls back to the ln_stats_finalize launch
  int lds_bytes = LDS_BYTES;
  if (a.ln_part != nullptr) {   // in-kernel finalize: the partial sums of a tile's 256 rows behind the single (mean, rstd) slot
    if (!lncons || !persistent || a.ln_ns < 1 || LNP_OFF + a.ln_ns * 2048 > LDS_MAX || (a.M & 1) || (FP8 ? a.K >> 7 : a.K >> 6) < 4)
      return MVF_ERR_UNSUPPORTED;
    lds_bytes = std::max(LDS_BYTES, LNP_OFF + a.ln_ns * 2048);
  }
  const int ntiles = ((a.M + BMT - 1) / BMT) * ((a.N + BN - 1) / BN);
  // persistent: one workgroup per CU (a multiple of 8 so that every XCD gets the same number); otherwise (A/B
  // measurements) one workgroup per tile -- the same kernel, every workgroup then runs the cold prologue
  int grid = persistent ? std::min(ntiles, std::max(8, num_cus() & ~7)) : ntiles;
  // Spare CUs (MVF_GEMM_SPARE = m, mvf_gemm_tc_set_spare): a persistent launch takes 256 - m workgroups where that needs no more
  // tile rounds than one workgroup per CU (never fewer than the smallest grid with that round count).  A workgroup of a full-width
  // launch that finds its CU held by another queue's kernel (the head's row-chain workgroups hold theirs for 30 - 40 us) keeps the
  // whole launch open until it has run -- for the short lane-sized launches (proj: two rounds of 15 us) that wait is longer than the
  // launch itself.
  if (g_spare > 0 && persistent && grid < ntiles) {
    const int rounds = (ntiles + grid - 1) / grid;
    const int gmin = (((ntiles + rounds - 1) / rounds) + 7) & ~7;
    grid = std::min(grid, std::max(gmin, (grid - g_spare) & ~7));
  }
  hipLaunchKernelGGL((gemm_tc256_kernel<EPI, DBG, LN, FP8, ABL, ADD2, BMT, F16>), dim3(grid), dim3(512), lds_bytes, st, a);
  MVF_LAUNCH_CHECK();
  return MVF_OK;
}

// tile rows: MVF_GEMM_BM = 256 | 240 | 224 | 208 pins them (A/B measurements, tests through mvf_gemm_tc_select 4 .. 7), otherwise
// the height whose ceil(tiles / workgroups) x rows x (1 + penalty) comes out smallest (persistent launches; the fp8, stamped and
// stacked-batch forms keep 256).  g_bm_set: the heights the automatic choice may use (MVF_GEMM_BM_SET, bit 0 = 240, 1 = 224, 2 = 208)
int g_bm_mode = [] { const char* e = getenv("MVF_GEMM_BM"); return e ? atoi(e) : 0; }();
int g_bm_set = [] { const char* e = getenv("MVF_GEMM_BM_SET"); return e ? atoi(e) : 7; }();

// (mirrored by bench.py's tile_rows() -- rocprof kernel names carry the height)
int pick_tile_rows(long M, long N, bool persistent) {
  if (g_bm_mode == 256 || g_bm_mode == 240 || g_bm_mode == 224 || g_bm_mode == 208) return g_bm_mode;
  if (!persistent) return 256;
  const long nwg = std::max(8, num_cus() & ~7), nbn = (N + BN - 1) / BN;
  auto rounds = [&](long bm) { return (((M + bm - 1) / bm) * nbn + nwg - 1) / nwg; };
  const long r256 = rounds(256);
  // only launches of three rounds or more on one to three tile columns: in the pipelined step (lane-sized launches of two
  // rounds, another lane filling the tail) 224-row tiles measured 1 % slower, in a full-batch launch on its own 6 % faster (fc2)
  if (r256 < 3) return 256;
  // a shorter tile re-uses its W fragments over fewer rows and pays its epilogue more often: per-mille penalties
  const int hs[3] = {240, 224, 208}, pen[3] = {1020, 1080, 1110};
  long best = 256, cost = r256 * 256 * 1000;
  for (int i = 0; i < 3; ++i) {
    // 224 / 208: N <= 768 only (measured on the N = 768 shapes).  240: full-batch fc1 only (ten rounds: 258 -> 253-260 us on its
    // own; on the lane-sized launches of the pipelined step, five rounds, 240-row tiles cost the step 1 %)
    if (!(g_bm_set >> i & 1) || (hs[i] == 240 ? r256 < 8 : nbn > 3)) continue;
    const long c = rounds(hs[i]) * hs[i] * pen[i];
    if (c < cost) { cost = c; best = hs[i]; }
  }
  return (int)best;
}

template <int EPI, bool DBG = false, bool LN = false, bool FP8 = false, int ABL = 0, bool ADD2 = false>
int launch(const GemmTcArgs& a, bool persistent, hipStream_t st) {
  if constexpr (!DBG && !FP8) {
    if (a.f16) {    // fp16 operands: the plain forms of the frozen backbone only (no stacked batches, no stamps)
      if (a.batch_rows != 0) return MVF_ERR_UNSUPPORTED;
      switch (pick_tile_rows(a.M, a.N, persistent)) {
        case 240: return launch_bm<EPI, DBG, LN, FP8, ABL, ADD2, 240, true>(a, persistent, st);
        case 224: return launch_bm<EPI, DBG, LN, FP8, ABL, ADD2, 224, true>(a, persistent, st);
        case 208: return launch_bm<EPI, DBG, LN, FP8, ABL, ADD2, 208, true>(a, persistent, st);
        default: return launch_bm<EPI, DBG, LN, FP8, ABL, ADD2, 256, true>(a, persistent, st);
      }
    }
    if (a.batch_rows == 0) switch (pick_tile_rows(a.M, a.N, persistent)) {
      case 240: return launch_bm<EPI, DBG, LN, FP8, ABL, ADD2, 240>(a, persistent, st);
      case 224: return launch_bm<EPI, DBG, LN, FP8, ABL, ADD2, 224>(a, persistent, st);
      case 208: return launch_bm<EPI, DBG, LN, FP8, ABL, ADD2, 208>(a, persistent, st);
      default: break;
    }
  } else {
    if (a.f16) return MVF_ERR_UNSUPPORTED;
  }
  return launch_bm<EPI, DBG, LN, FP8, ABL, ADD2, 256>(a, persistent, st);
}

}  // namespace

// The persistent launch sizes its grid to one workgroup per CU of its budget (0 = what the device reports): a data-parallel
// run keeps 8 CUs out of it for RCCL's kernels (include/mvf_hip.h), a CU-masked stream (hipExtStreamCreateWithCUMask) has fewer.
extern "C" int mvf_gemm_tc_set_cus(int n) {
  MVF_CHECK_ARG(n >= 0 && n <= 4096);
  g_cu_budget = n;
  return MVF_OK;
}

extern "C" int mvf_gemm_tc_set_spare(int cus) {
  MVF_CHECK_ARG(cus >= 0 && cus <= 256);
  g_spare = cus;
  return MVF_OK;
}

extern "C" int mvf_gemm_tc_set_ngroup(int g) {
  MVF_CHECK_ARG(g >= -1 && g <= 64);
  g_ngroup = g;
  return MVF_OK;
}

int mvf_gemm_tc256_num_wgs() { return std::max(8, num_cus() & ~7); }
extern "C" int mvf_gemm_tc_get_wgs(int* out) {
  MVF_CHECK_ARG(out != nullptr);
  *out = mvf_gemm_tc256_num_wgs();
  return MVF_OK;
}
void mvf_gemm_tc256_set_bm(int bm) { g_bm_mode = bm; }

int mvf_gemm_tc256_launch(int epi, const gemm_tc::GemmTcArgs& a, bool persistent, hipStream_t st) {
  const bool fp8 = a.sa != nullptr;
  if (a.K % (fp8 ? 256 : 128) != 0 || a.K < 128 || a.N % 32 != 0) return MVF_ERR_ARG;
  // 32-bit operand offsets inside the kernel
  const size_t wrows = a.batch_rows > 0 ? (size_t)(a.M / a.batch_rows) * a.w_batch_rows : (size_t)a.N;
  if ((size_t)a.M * a.lda * 2 >= (1ull << 32) || wrows * a.ldw * 2 >= (1ull << 32) || a.M >= (1 << 24) ||
      wrows >= (1u << 24) || a.lda >= (1 << 23) || a.ldw >= (1 << 23))
    return MVF_ERR_UNSUPPORTED;
  if (fp8) {   // MX-fp8 operands (validated by mvf_gemm_fp8): no stacked batches, no stamps
    if (a.sw == nullptr || a.batch_rows != 0 || a.dbg != nullptr || a.ln_part != nullptr) return MVF_ERR_ARG;
    // LN fold, consumer side (qkv on the un-normalised MX-fp8 residual stream; (mean, rstd) from the finalize launch: the in-kernel
    // form's LDS region is where this kernel keeps its operand scales) and producer side (fc2: xb = MX-fp8(x_new) + csc + row sums)
    if (a.ln_mr != nullptr)
      return epi == EPI_STORE && a.ln_c != nullptr ? launch<EPI_STORE, false, true, true>(a, persistent, st) : MVF_ERR_ARG;
    if (a.xb != nullptr || a.stats != nullptr) {
      if (epi != EPI_RESID || a.xb == nullptr || a.stats == nullptr || a.csc == nullptr || a.N % 128 != 0) return MVF_ERR_ARG;
      return a.radd2 != nullptr ? launch<EPI_RESID, false, true, true, 0, true>(a, persistent, st)
                                : launch<EPI_RESID, false, true, true>(a, persistent, st);
    }
    switch (epi) {
      case EPI_STORE: return launch<EPI_STORE, false, false, true>(a, persistent, st);
      case EPI_GELU: return launch<EPI_GELU, false, false, true>(a, persistent, st);
      case EPI_GELU_Q: return a.csc != nullptr && a.N % 128 == 0 ? launch<EPI_GELU_Q, false, false, true>(a, persistent, st) : MVF_ERR_ARG;
      case EPI_RESID:
        return a.radd2 != nullptr ? launch<EPI_RESID, false, false, true, 0, true>(a, persistent, st)
                                  : launch<EPI_RESID, false, false, true>(a, persistent, st);
    }
    return MVF_ERR_ARG;
  }
  if (a.dbg != nullptr) {
    if (epi != EPI_STORE) return MVF_ERR_UNSUPPORTED;
    if (a.dbg_kt >= 0) return launch<EPI_STORE, true, false, false, 8>(a, persistent, st);   // per-barrier stamps of one K tile
    switch (a.dbg_abl) {   // timing ablations are separate instantiations: a run-time test inside the loop changes its schedule
      case 0: return launch<EPI_STORE, true>(a, persistent, st);
      case 1: return launch<EPI_STORE, true, false, false, 1>(a, persistent, st);
      case 2: return launch<EPI_STORE, true, false, false, 2>(a, persistent, st);
      case 3: return launch<EPI_STORE, true, false, false, 3>(a, persistent, st);
      case 4: return launch<EPI_STORE, true, false, false, 4>(a, persistent, st);
      case 5: return launch<EPI_STORE, true, false, false, 5>(a, persistent, st);
      case 6: return launch<EPI_STORE, true, false, false, 6>(a, persistent, st);
      case 7: return launch<EPI_STORE, true, false, false, 7>(a, persistent, st);
      case 22: return launch<EPI_STORE, true, false, false, 22>(a, persistent, st);   // MFMAs + every other barrier
      case 23: return launch<EPI_STORE, true, false, false, 23>(a, persistent, st);   // every other barrier alone
    }
    return MVF_ERR_ARG;
  }
  if (a.ln_mr != nullptr || a.ln_part != nullptr || a.xb != nullptr || a.stats != nullptr) {   // LN-fold extras (validated by mvf_gemm_tc_impl)
    switch (epi) {
      case EPI_STORE: return launch<EPI_STORE, false, true>(a, persistent, st);
      case EPI_GELU: return launch<EPI_GELU, false, true>(a, persistent, st);
      case EPI_RESID:
        return a.radd2 != nullptr ? launch<EPI_RESID, false, true, false, 0, true>(a, persistent, st)
                                  : launch<EPI_RESID, false, true>(a, persistent, st);
    }
    return MVF_ERR_ARG;
  }
  switch (epi) {
    case EPI_STORE: return launch<EPI_STORE>(a, persistent, st);
    case EPI_GELU: return launch<EPI_GELU>(a, persistent, st);
    case EPI_RESID:
      return a.radd2 != nullptr ? launch<EPI_RESID, false, false, false, 0, true>(a, persistent, st) : launch<EPI_RESID>(a, persistent, st);
    case EPI_PATCH: return launch<EPI_PATCH>(a, persistent, st);
  }
  return MVF_ERR_ARG;
}
