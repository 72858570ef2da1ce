// Host-side driver of the frozen ViT backbone forward: one C-ABI call enqueues the whole
// patch-embed + depth x (LN, QKV, attention, proj, LN, FC1+GELU, FC2) chain on the caller's stream and
// writes the tapped block outputs (CLS dropped) straight in the layout the LSTP pooling kernels read.
//
// Replaces: timm VisionTransformer.forward + FeatureExtractor hooks/concat + the CLS-drop / movedim /
// reshape copies of CARL_MVF/models/transformer.py:186-214, 306-333.
#include "common.h"
#include "mvf_hip_internal.h"
#include <cstdlib>
#include <vector>

namespace {
enum { EPI_STORE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_PATCH = 3 };

// MVF_LN_INKERNEL=0: keep the ln_stats_finalize launch in front of the folded-LayerNorm GEMMs (A/B measurements)
const bool g_ln_inkernel = []{ const char* e = getenv("MVF_LN_INKERNEL"); return e == nullptr || e[0] != '0'; }();
// MVF_PROJ_DEFER=0: keep the proj GEMM's read-modify epilogue (A/B measurements)
const bool g_proj_defer = []{ const char* e = getenv("MVF_PROJ_DEFER"); return e == nullptr || e[0] != '0'; }();
// fp8 mode: the attention kernel writes the proj GEMM's MX-fp8 operand itself (MVF_ATTN_Q8=0: bf16 output + mvf_quant_mxfp8, for A/B runs)
const bool g_attn_q8 = []{ const char* e = getenv("MVF_ATTN_Q8"); return e == nullptr || e[0] != '0'; }();

struct Ws {
  float* x;     // residual stream  [Mc, D] fp32
  char* h;      // LN out / attention out [Mc, D] T
  char* qkv;    // [Mc, 3D] T
  char* hid;    // [Mc, 4D] T   (also holds the patch rows before the embed GEMM)
  // LN fold (bf16): bf16 copy of the residual stream, row partial sums from the residual epilogues, (mean, rstd) per row
  char* xb;     // [Mc, D] bf16
  char* delta;  // [Mc, D] bf16: the attention branch's output (proj GEMM) while its residual add is deferred to LN2 / fc2
  float* stats; // [D/64, Mc, 2]
  float* mr;    // [Mc, 2]
  // MX-fp8 mode: the A operands of the four GEMMs (LayerNorm output / attention output share hq; fc1+GELU output -> hidq)
  char* hq;         // [Mc, D] e4m3
  unsigned* hs;     // [D/128][Mc] block scales
  char* hidq;       // [Mc, 4D] e4m3
  unsigned* hids;   // [4D/128][Mc]
};

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// ---- optional per-launch timing of the GEMMs with HIP events on the launch stream (bench.py roofline) ----
struct Prof {
  bool on = false;
  std::vector<hipEvent_t> ev;   // pairs
  std::vector<int> epi, n, k;
  std::vector<double> flops;
  size_t used = 0;
} g_prof;

struct Fp8Scales {   // non-null: MX-fp8 operands (mvf_gemm_fp8_impl)
  const unsigned* sa;
  const unsigned* sw;
  unsigned* csc;       // epi 1: quantise the output too (C = e4m3 bytes, csc its block scales), or NULL
  const void* addend2; // epi 2: the deferred attention-branch output (bf16 [M, ld2]), or NULL
  int ld2;
};

int timed_gemm(int dtype, int epi, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
               float* resid, int ldr, void* tap, int ldt, const float* pos, const float* ls, int tpf, int M, int N, int K,
               hipStream_t st, const MvfGemmLn* ln = nullptr, const Fp8Scales* f8 = nullptr) {
  const bool rec = g_prof.on && g_prof.used < 4096;
  size_t slot = 0;
  if (rec) {
    slot = g_prof.used++;
    if (g_prof.ev.size() < 2 * (slot + 1)) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return MVF_ERR_ARG;
      g_prof.ev.push_back(a);
      g_prof.ev.push_back(b);
      g_prof.epi.push_back(0);
      g_prof.n.push_back(0);
      g_prof.k.push_back(0);
      g_prof.flops.push_back(0.0);
    }
    g_prof.epi[slot] = epi;
    g_prof.n[slot] = N;
    g_prof.k[slot] = K;
    g_prof.flops[slot] = 2.0 * M * (double)N * K;
    (void)hipEventRecord(g_prof.ev[2 * slot], st);
  }
  const int rc = f8 != nullptr
                     ? mvf_gemm_fp8_impl(epi, A, lda, f8->sa, W, ldw, f8->sw, bias, C, ldc, f8->csc, resid, ldr, tap, ldt, ls, tpf, M, N, K, st,
                                         f8->addend2, f8->ld2, ln)
                     : mvf_gemm_tc_impl(dtype, epi, A, lda, W, ldw, bias, C, ldc, resid, ldr, tap, ldt, pos, ls, tpf, M, N, K, st, 0,
                                        0, ln);
  if (rec) {
    if (rc == MVF_ERR_UNSUPPORTED) --g_prof.used;     // nothing was launched (the caller takes another form): give the slot back
    else (void)hipEventRecord(g_prof.ev[2 * slot + 1], st);
  }
  return rc;
}

// the fused qkv + attention launch under the same profile (mvf_prof_collect group: epi 5, n = 3 D, k = D; FLOPs = the projection's
// 2 M 3D D + the attention's 4 F H N^2 64)
constexpr int PROF_QKV_ATTN = 5;
int timed_qkv_attn(int dtype, const void* A, const void* W, const float* bias, const float* ln_c, const float* ln_mr,
                   const float* ln_part, int ln_ns, float ln_eps, void* out, int F, int N, int H, int D, hipStream_t st) {
  const bool rec = g_prof.on && g_prof.used < 4096;
  size_t slot = 0;
  if (rec) {
    slot = g_prof.used++;
    if (g_prof.ev.size() < 2 * (slot + 1)) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return MVF_ERR_ARG;
      g_prof.ev.push_back(a);
      g_prof.ev.push_back(b);
      g_prof.epi.push_back(0);
      g_prof.n.push_back(0);
      g_prof.k.push_back(0);
      g_prof.flops.push_back(0.0);
    }
    g_prof.epi[slot] = PROF_QKV_ATTN;
    g_prof.n[slot] = 3 * D;
    g_prof.k[slot] = D;
    g_prof.flops[slot] = 2.0 * F * N * 3.0 * D * D + 4.0 * F * H * (double)N * N * 64.0;
    (void)hipEventRecord(g_prof.ev[2 * slot], st);
  }
  const int rc = mvf_qkv_attn_impl(dtype, A, D, W, bias, ln_c, ln_mr, ln_part, ln_ns, ln_eps, out, F, N, H, D, st);
  if (rec) {
    if (rc == MVF_ERR_UNSUPPORTED) --g_prof.used;
    else (void)hipEventRecord(g_prof.ev[2 * slot + 1], st);
  }
  return rc;
}

size_t carve(int dtype, int fc, int N, int D, int P, Ws* w, char* base) {
  const size_t esz = dtype == MVF_F32 ? 4 : 2;     // MX-fp8 mode keeps bf16 activations between its quantisers
  const size_t Mc = (size_t)fc * N;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align256(bytes);
    return p;
  };
  char* x = take(Mc * D * 4);
  char* h = take(Mc * D * esz);
  char* qkv = take(Mc * 3 * D * esz);
  const size_t patch_bytes = (size_t)fc * (N - 1) * mvf_patch_k(P) * esz;
  char* hid = take(std::max(Mc * 4 * D * esz, patch_bytes));
  char* xb = nullptr;
  char* delta = nullptr;
  char* stats = nullptr;
  char* mr = nullptr;
  if (dtype == MVF_FP8 && D % 64 == 0) {
    delta = take(Mc * D * 2);
    stats = take(Mc * (size_t)(D / 64) * 2 * 4);   // LayerNorm 1 folded into the fp8 qkv GEMM (run_blocks)
    mr = take(Mc * 2 * 4);
  }
  if ((dtype == MVF_BF16 || dtype == MVF_F16) && D % 64 == 0) {
    xb = take(Mc * D * 2);
    delta = take(Mc * D * 2);
    stats = take(Mc * (size_t)(D / 64) * 2 * 4);
    mr = take(Mc * 2 * 4);
  }
  char *hq = nullptr, *hs = nullptr, *hidq = nullptr, *hids = nullptr;
  if (dtype == MVF_FP8) {
    hq = take(Mc * D);
    hs = take((size_t)(D / 128) * Mc * 4);
    hidq = take(Mc * 4 * D);
    hids = take((size_t)(4 * D / 128) * Mc * 4);
  }
  if (w) {
    w->x = (float*)x; w->h = h; w->qkv = qkv; w->hid = hid; w->xb = xb; w->delta = delta; w->stats = (float*)stats; w->mr = (float*)mr;
    w->hq = hq; w->hs = (unsigned*)hs; w->hidq = hidq; w->hids = (unsigned*)hids;
  }
  return off;
}
}  // namespace

extern "C" size_t mvf_vit_workspace_bytes(int dtype, int frames_per_chunk, int tokens, int dim, int patch) {
  return carve(dtype, frames_per_chunk, tokens, dim, patch, nullptr, nullptr);
}

namespace {
#define RUN(call)               \
  do {                          \
    rc = (call);                \
    if (rc != MVF_OK) return rc; \
  } while (0)

// Blocks [l0, l1) on the `fc` frames whose fp32 residual stream is ws.x (updated in place); `dtype` is the activation dtype
// (fp8 mode: bf16), taps_out slices start at frame f0.  Layer l0 must not be a folded-LayerNorm consumer unless the previous
// residual epilogue of THIS call produced its statistics (l0 == first layer of the model: never folded).
int run_blocks(const MvfVitWeights* w, int dtype, bool fp8, const Ws& ws, int fc, int N, int l0, int l1, void* const* taps_out,
               int f0, int attn_variant, hipStream_t st) {
  const int D = w->dim, H = w->heads, np = N - 1;
  const int Mc = fc * N;
  const bool h16 = dtype == MVF_BF16 || dtype == MVF_F16;     // 16-bit activations (bf16, or IEEE fp16: the same data flow)
  const size_t esz = h16 ? 2 : 4;
  int rc;
  // LN fold (bf16): where a layer's table entry qkv_c[l] / fc1_c[l] is set, qkv_w / fc1_w hold gamma (.) W, the bias table
  // holds b + W beta, and the GEMM consumes xb = bf16(x) with the row statistics applied in its epilogue -- no LayerNorm
  // kernel.  xb and the statistics' partial sums come out of the PREVIOUS residual epilogue (proj for LN2, the previous
  // layer's fc2 for LN1); layer 0's LN1 follows the patch embedding and keeps the LayerNorm kernel.
  const bool can_fold = !fp8 && h16 && ws.xb != nullptr && D % 128 == 0;
  auto folded = [&](const float* const* tab, int l) { return can_fold && tab != nullptr && l < w->depth && tab[l] != nullptr; };
  const int ns = D / 64;
  // q rows of the packed qkv weights pre-scaled by log2(e) / 8 (MvfVitWeights.q_prescaled): the attention kernels take q as it comes
  const int qs_bit = w->q_prescaled ? MVF_ATTN_Q_PRESCALED : 0;
  for (int l = l0; l < l1; ++l) {
    int tap = -1;
    for (int j = 0; j < w->n_taps; ++j)
      if (w->taps[j] == l) tap = j;
    if (fp8) {
      // MX-fp8 block: every GEMM operand is quantised by its producer (LayerNorm and the fc1+GELU epilogue write fp8
      // directly; the attention kernel quantises its output in its epilogue -- the streamed 32-row kernel at any token count)
      void* tap_ptr = nullptr;
      if (tap >= 0 && taps_out && taps_out[tap]) tap_ptr = (char*)taps_out[tap] + (size_t)f0 * np * D * 2;
      // deferred residual as in the bf16 path below (LayerScale models: the packer folds gamma_1 into proj's weights and bias
      // and leaves ls1 NULL)
      const bool defer8 = g_proj_defer && ws.delta != nullptr && w->ls1 == nullptr;
      const Fp8Scales sq = {ws.hs, w->qkv_s[l], nullptr, nullptr, 0}, sp = {ws.hs, w->proj_s[l], nullptr, nullptr, 0},
                      s1 = {ws.hs, w->fc1_s[l], ws.hids, nullptr, 0},
                      s2 = {ws.hids, w->fc2_s[l], nullptr, defer8 ? ws.delta : nullptr, D};
      // LayerNorm 1 folded into the qkv GEMM (qkv_c[l] set: qkv_w[l] = MX-fp8(gamma (.) W), qkv_b[l] = b + W beta): the previous
      // block's fc2 epilogue left MX-fp8(x) in hq / hs and the row partial sums in stats -- no layernorm_mxfp8 pass over the stream
      auto fold8 = [&](int ll) { return w->qkv_c != nullptr && ll < w->depth && w->qkv_c[ll] != nullptr && ws.stats != nullptr; };
      if (fold8(l)) {
        if (l == l0) return MVF_ERR_ARG;   // nothing in this call produced the layer's operand and statistics
        RUN(mvf_ln_stats_finalize_impl(ws.stats, ns, ws.mr, Mc, D, w->ln_eps, st));
        const MvfGemmLn lc = {nullptr, 0, nullptr, ws.mr, w->qkv_c[l], 0, nullptr};
        RUN(timed_gemm(dtype, EPI_STORE, ws.hq, D, w->qkv_w[l], D, w->qkv_b[l], ws.qkv, 3 * D, nullptr, 0, nullptr, 0, nullptr,
                       nullptr, N, Mc, 3 * D, D, st, &lc, &sq));
      } else {
        RUN(mvf_layernorm_mxfp8_impl(ws.x, D, w->ln1_w[l], w->ln1_b[l], ws.hq, D, ws.hs, Mc, D, w->ln_eps, st));
        RUN(timed_gemm(dtype, EPI_STORE, ws.hq, D, w->qkv_w[l], D, w->qkv_b[l], ws.qkv, 3 * D, nullptr, 0, nullptr, 0, nullptr,
                       nullptr, N, Mc, 3 * D, D, st, nullptr, &sq));
      }
      if (attn_variant == 0 && H % 2 == 0 && g_attn_q8) {
        // the attention kernel quantises its own output (bit for bit the two launches below; the qkv GEMM is done with hq / hs)
        RUN(mvf_vit_attn32_impl(MVF_BF16, ws.qkv, ws.hq, nullptr, fc, N, H, D, qs_bit ? 5 + 16 : 5, 0, st, ws.hs));
      } else {
        RUN(mvf_vit_attn_impl(dtype, ws.qkv, ws.h, fc, N, H, D, attn_variant | qs_bit, st));
        RUN(mvf_quant_mxfp8_impl(MVF_BF16, ws.h, D, ws.hq, D, ws.hs, Mc, D, st));
      }
      if (defer8)
        RUN(timed_gemm(dtype, EPI_STORE, ws.hq, D, w->proj_w[l], D, w->proj_b[l], ws.delta, D, nullptr, 0, nullptr, 0, nullptr,
                       nullptr, N, Mc, D, D, st, nullptr, &sp));
      else
        RUN(timed_gemm(dtype, EPI_RESID, ws.hq, D, w->proj_w[l], D, w->proj_b[l], nullptr, 0, ws.x, D, nullptr, 0, nullptr,
                       w->ls1 ? w->ls1[l] : nullptr, N, Mc, D, D, st, nullptr, &sp));
      RUN(mvf_layernorm_mxfp8_impl(ws.x, D, w->ln2_w[l], w->ln2_b[l], ws.hq, D, ws.hs, Mc, D, w->ln_eps, st,
                                   defer8 ? ws.delta : nullptr, D));
      // fc1 + GELU with the MX-fp8 quantisation in its epilogue: hidq / hids straight out of the GEMM (no bf16 hid)
      RUN(timed_gemm(dtype, EPI_GELU, ws.hq, D, w->fc1_w[l], D, w->fc1_b[l], ws.hidq, 4 * D, nullptr, 0, nullptr, 0, nullptr,
                     nullptr, N, Mc, 4 * D, D, st, nullptr, &s1));
      // producer side of the next block's fold: hq / hs are free here (fc1 consumed them) and are next read by that block's qkv GEMM
      MvfGemmLn lpr = {ws.hq, D, ws.stats, nullptr, nullptr, 0, nullptr};
      lpr.xb_scales = ws.hs;
      RUN(timed_gemm(dtype, EPI_RESID, ws.hidq, 4 * D, w->fc2_w[l], 4 * D, w->fc2_b[l], nullptr, 0, ws.x, D, tap_ptr, D,
                     nullptr, w->ls2 ? w->ls2[l] : nullptr, N, Mc, D, 4 * D, st, (l + 1 < l1 && fold8(l + 1)) ? &lpr : nullptr, &s2));
      continue;
    }
    // qkv projection fused into the attention kernel (vit_qkv_attn.hip): the [Mc, 3D] qkv tensor never reaches HBM
    const bool fused_attn = attn_variant == 0 && qs_bit == 0 && mvf_qkv_attn_supported(dtype, fc, N, H, D, D);
    if (fused_attn) {
      if (folded(w->qkv_c, l)) {
        if (l == l0) return MVF_ERR_ARG;   // nothing in this call produced the layer's statistics
        // (the kernel turns the producer's partial sums into (mean, rstd) itself: no finalize launch between fc2 and this one)
        if (g_ln_inkernel) {
          RUN(timed_qkv_attn(dtype, ws.xb, w->qkv_w[l], w->qkv_b[l], w->qkv_c[l], nullptr, ws.stats, ns, w->ln_eps, ws.h, fc, N, H, D, st));
        } else {
          RUN(mvf_ln_stats_finalize_impl(ws.stats, ns, ws.mr, Mc, D, w->ln_eps, st));
          RUN(timed_qkv_attn(dtype, ws.xb, w->qkv_w[l], w->qkv_b[l], w->qkv_c[l], ws.mr, nullptr, 0, 0.f, ws.h, fc, N, H, D, st));
        }
      } else {
        // the LayerNorm output is parked in the (otherwise unused) qkv buffer: the kernel's output goes to ws.h, and a unit's
        // output columns must not land in rows another (frame, head) unit of the same launch still reads as its operand
        RUN(mvf_layernorm_impl(dtype, ws.x, D, w->ln1_w[l], w->ln1_b[l], ws.qkv, D, Mc, D, w->ln_eps, st));
        RUN(timed_qkv_attn(dtype, ws.qkv, w->qkv_w[l], w->qkv_b[l], nullptr, nullptr, nullptr, 0, 0.f, ws.h, fc, N, H, D, st));
      }
    }
    if (fused_attn) {
      // done
    } else if (folded(w->qkv_c, l)) {
      if (l == l0) return MVF_ERR_ARG;   // nothing in this call produced the layer's statistics
      // the GEMM turns the producer's partial sums into (mean, rstd) itself (no finalize launch between fc2 and qkv); shapes it
      // cannot stage (more than 12 slices: D > 768, an odd row count) keep the small kernel
      const MvfGemmLn lp = {nullptr, 0, nullptr, nullptr, w->qkv_c[l], 0, nullptr, nullptr, 0, ws.stats, ns, w->ln_eps};
      rc = g_ln_inkernel ? timed_gemm(dtype, EPI_STORE, ws.xb, D, w->qkv_w[l], D, w->qkv_b[l], ws.qkv, 3 * D, nullptr, 0, nullptr, 0,
                                      nullptr, nullptr, N, Mc, 3 * D, D, st, &lp)
                         : MVF_ERR_UNSUPPORTED;
      if (rc == MVF_ERR_UNSUPPORTED) {
        RUN(mvf_ln_stats_finalize_impl(ws.stats, ns, ws.mr, Mc, D, w->ln_eps, st));
        const MvfGemmLn ln = {nullptr, 0, nullptr, ws.mr, w->qkv_c[l], 0, nullptr};
        RUN(timed_gemm(dtype, EPI_STORE, ws.xb, D, w->qkv_w[l], D, w->qkv_b[l], ws.qkv, 3 * D, nullptr, 0, nullptr, 0,
                       nullptr, nullptr, N, Mc, 3 * D, D, st, &ln));
      } else if (rc != MVF_OK) {
        return rc;
      }
    } else {
      RUN(mvf_layernorm_impl(dtype, ws.x, D, w->ln1_w[l], w->ln1_b[l], ws.h, D, Mc, D, w->ln_eps, st));
      RUN(timed_gemm(dtype, EPI_STORE, ws.h, D, w->qkv_w[l], D, w->qkv_b[l], ws.qkv, 3 * D, nullptr, 0, nullptr, 0,
                     nullptr, nullptr, N, Mc, 3 * D, D, st));
    }
    if (!fused_attn) RUN(mvf_vit_attn_impl(dtype, ws.qkv, ws.h, fc, N, H, D, attn_variant | qs_bit, st));
    const bool fold2 = folded(w->fc1_c, l);
    // Deferred residual (bf16, no LayerScale, norm2 not folded): proj stores its result (+ bias) as bf16 with the plain
    // epilogue instead of read-modifying the fp32 residual (310 MB per launch with the matrix cores idle); LayerNorm 2
    // normalises x + delta and the fc2 epilogue adds delta with its own residual update.  The branch output is rounded to
    // bf16 before the add -- what the reference's autocast does to it (fp16 there; transformer.py:188).
    const bool defer = g_proj_defer && h16 && ws.delta != nullptr && !fold2 && w->ls1 == nullptr && D % 128 == 0;
    if (defer) {
      RUN(timed_gemm(dtype, EPI_STORE, ws.h, D, w->proj_w[l], D, w->proj_b[l], ws.delta, D, nullptr, 0, nullptr, 0, nullptr,
                     nullptr, N, Mc, D, D, st));
    } else {
      const MvfGemmLn ln = {fold2 ? ws.xb : nullptr, D, fold2 ? ws.stats : nullptr, nullptr, nullptr, 0, nullptr};
      RUN(timed_gemm(dtype, EPI_RESID, ws.h, D, w->proj_w[l], D, w->proj_b[l], nullptr, 0, ws.x, D, nullptr, 0,
                     nullptr, w->ls1 ? w->ls1[l] : nullptr, N, Mc, D, D, st, fold2 ? &ln : nullptr));
    }
    if (fold2) {
      const MvfGemmLn lp = {nullptr, 0, nullptr, nullptr, w->fc1_c[l], 0, nullptr, nullptr, 0, ws.stats, ns, w->ln_eps};
      rc = g_ln_inkernel ? timed_gemm(dtype, EPI_GELU, ws.xb, D, w->fc1_w[l], D, w->fc1_b[l], ws.hid, 4 * D, nullptr, 0, nullptr, 0,
                                      nullptr, nullptr, N, Mc, 4 * D, D, st, &lp)
                         : MVF_ERR_UNSUPPORTED;
      if (rc == MVF_ERR_UNSUPPORTED) {
        RUN(mvf_ln_stats_finalize_impl(ws.stats, ns, ws.mr, Mc, D, w->ln_eps, st));
        const MvfGemmLn ln = {nullptr, 0, nullptr, ws.mr, w->fc1_c[l], 0, nullptr};
        RUN(timed_gemm(dtype, EPI_GELU, ws.xb, D, w->fc1_w[l], D, w->fc1_b[l], ws.hid, 4 * D, nullptr, 0, nullptr, 0,
                       nullptr, nullptr, N, Mc, 4 * D, D, st, &ln));
      } else if (rc != MVF_OK) {
        return rc;
      }
    } else {
      RUN(mvf_layernorm_impl(dtype, ws.x, D, w->ln2_w[l], w->ln2_b[l], ws.h, D, Mc, D, w->ln_eps, st, defer ? ws.delta : nullptr, D));
      RUN(timed_gemm(dtype, EPI_GELU, ws.h, D, w->fc1_w[l], D, w->fc1_b[l], ws.hid, 4 * D, nullptr, 0, nullptr, 0,
                     nullptr, nullptr, N, Mc, 4 * D, D, st));
    }
    void* tap_ptr = nullptr;
    if (tap >= 0 && taps_out && taps_out[tap]) tap_ptr = (char*)taps_out[tap] + (size_t)f0 * np * D * esz;
    const bool fold_next = l + 1 < l1 && folded(w->qkv_c, l + 1);
    {
      const MvfGemmLn ln = {fold_next ? ws.xb : nullptr, D, fold_next ? ws.stats : nullptr, nullptr, nullptr, 0, nullptr,
                            defer ? ws.delta : nullptr, D};
      RUN(timed_gemm(dtype, EPI_RESID, ws.hid, 4 * D, w->fc2_w[l], 4 * D, w->fc2_b[l], nullptr, 0, ws.x, D, tap_ptr,
                     D, nullptr, w->ls2 ? w->ls2[l] : nullptr, N, Mc, D, 4 * D, st, (fold_next || defer) ? &ln : nullptr));
    }
  }
  return MVF_OK;
}
}  // namespace

extern "C" int mvf_vit_fwd(const MvfVitWeights* w, int dtype, const float* frames, int F, void* const* taps_out,
                           float* cls_out, void* workspace, size_t ws_bytes, int frames_per_chunk, int attn_variant,
                           hipStream_t st) {
  return mvf_vit_fwd_x(w, dtype, frames, F, taps_out, cls_out, nullptr, workspace, ws_bytes, frames_per_chunk, attn_variant, st);
}

// x_out != NULL: also hand out the fp32 residual stream [F*N, dim] (CLS row included) after the last of the w->depth
// blocks -- the frozen FRONT END of a partially frozen backbone (ViTFrontEnd, models/transformer.py:342-361; depth may
// then be 0 = patch embedding + position embedding only)
extern "C" int mvf_vit_fwd_x(const MvfVitWeights* w, int dtype, const float* frames, int F, void* const* taps_out,
                             float* cls_out, float* x_out, void* workspace, size_t ws_bytes, int frames_per_chunk,
                             int attn_variant, hipStream_t st) {
  MVF_CHECK_ARG(w && frames && workspace && F > 0);
  MVF_CHECK_ARG(dtype == MVF_F32 || dtype == MVF_BF16 || dtype == MVF_FP8 || dtype == MVF_F16);
  const bool fp8 = dtype == MVF_FP8;
  if (fp8) {   // MX-fp8 GEMM operands, bf16 everywhere else (patch embedding, attention, taps); LN fold: norm1 only (qkv_c)
    MVF_CHECK_ARG(w->dim % 256 == 0 && w->qkv_s && w->proj_s && w->fc1_s && w->fc2_s && !w->fc1_c && (!w->qkv_c || w->depth == 0 || !w->qkv_c[0]));
    dtype = MVF_BF16;
  }
  const int D = w->dim, H = w->heads, P = w->patch, img = w->img;
  MVF_CHECK_ARG(D == H * 64 && img % P == 0 && (w->depth > 0 || (x_out && w->depth == 0)) && w->n_taps >= 0 && w->n_taps <= 8);
  const int np = (img / P) * (img / P);
  const int N = np + 1;
  const int fc_max = frames_per_chunk > 0 ? std::min(frames_per_chunk, F) : F;
  Ws ws;
  MVF_CHECK_ARG(carve(fp8 ? MVF_FP8 : dtype, fc_max, N, D, P, &ws, (char*)workspace) <= ws_bytes);
  const int kp = mvf_patch_k(P);   // patch_w is [dim, kp] (zero-padded beyond 3*P*P)
  int rc;
  for (int f0 = 0; f0 < F; f0 += fc_max) {
    const int fc = std::min(fc_max, F - f0);
    const int Mc = fc * N;
    // ---- patch embed: gather patches, GEMM with bias + pos_embed fused, rows 1.. of every frame ----
    RUN(mvf_im2col_impl(dtype, frames + (size_t)f0 * 3 * img * img, ws.hid, fc, img, img, P, kp, st));
    RUN(timed_gemm(dtype, EPI_PATCH, ws.hid, kp, w->patch_w, kp, w->patch_b, nullptr, 0, ws.x, D, nullptr, 0,
                         w->pos_embed, nullptr, N, fc * np, D, kp, st));
    RUN(mvf_cls_row_impl(ws.x, w->cls_token, w->pos_embed, fc, N, D, st));
    RUN(run_blocks(w, dtype, fp8, ws, fc, N, 0, w->depth, taps_out, f0, attn_variant, st));
    if (x_out && hipMemcpyAsync(x_out + (size_t)f0 * N * D, ws.x, (size_t)Mc * D * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return MVF_ERR_ARG;
    if (cls_out)  // final LN on the CLS rows only (timm forward_head, global_pool='token')
      RUN(mvf_layernorm_impl(MVF_F32, ws.x, (size_t)N * D, w->norm_w, w->norm_b, cls_out + (size_t)f0 * D, D, fc, D,
                             w->ln_eps, st));
  }
  return MVF_OK;
}

// Blocks [first_block, first_block + n_blocks) of the backbone on a residual stream the CALLER supplies: x [F*N, dim] fp32,
// updated in place (timm Block.forward, reached from models/transformer.py:188 through VisionTransformer.forward_features).
// The per-block ("teacher-forced") parity checks feed every block the oracle's input, so that one block's error is seen on
// its own instead of through the 12 / 24 blocks behind it.  bf16 with the LayerNorm fold: first_block must be a layer whose
// norm1 is NOT folded (its statistics would have come from the previous block's epilogue) -- pack with the fold off.
extern "C" int mvf_vit_blocks_fwd(const MvfVitWeights* w, int dtype, float* x, int F, int first_block, int n_blocks,
                                  void* workspace, size_t ws_bytes, int attn_variant, hipStream_t st) {
  MVF_CHECK_ARG(w && x && workspace && F > 0 && first_block >= 0 && n_blocks > 0 && first_block + n_blocks <= w->depth);
  MVF_CHECK_ARG(dtype == MVF_F32 || dtype == MVF_BF16 || dtype == MVF_FP8 || dtype == MVF_F16);
  const bool fp8 = dtype == MVF_FP8;
  if (fp8) {
    MVF_CHECK_ARG(w->dim % 256 == 0 && w->qkv_s && w->proj_s && w->fc1_s && w->fc2_s && !w->fc1_c);
    dtype = MVF_BF16;
  }
  const int D = w->dim, P = w->patch, img = w->img;
  MVF_CHECK_ARG(D == w->heads * 64 && img % P == 0 && ((uintptr_t)x % 16) == 0);
  const int N = (img / P) * (img / P) + 1;
  Ws ws;
  MVF_CHECK_ARG(carve(fp8 ? MVF_FP8 : dtype, F, N, D, P, &ws, (char*)workspace) <= ws_bytes);
  ws.x = x;
  return run_blocks(w, dtype, fp8, ws, F, N, first_block, first_block + n_blocks, nullptr, 0, attn_variant, st);
}
#undef RUN

// ---- profiling hooks (NOT graph-capturable: mvf_prof_collect synchronises the recorded events) ----
extern "C" int mvf_prof_enable(int on) {
  g_prof.on = on != 0;
  if (on) g_prof.used = 0;
  return MVF_OK;
}
// launches are grouped by (epilogue kind, N, K) = one GEMM shape of the backbone: for group g < *n_groups,
// ms[g] = summed device time, flops[g] = summed 2*M*N*K, count[g] = launches, epi/n/k[g] = the key
extern "C" int mvf_prof_collect(double* ms, double* flops, int* count, int* epi, int* n, int* k, int max_groups,
                                int* n_groups) {
  MVF_CHECK_ARG(ms && flops && count && epi && n && k && n_groups && max_groups > 0);
  int ng = 0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return MVF_ERR_ARG;
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return MVF_ERR_ARG;
    int g = 0;
    while (g < ng && !(epi[g] == g_prof.epi[i] && n[g] == g_prof.n[i] && k[g] == g_prof.k[i])) ++g;
    if (g == ng) {
      if (ng == max_groups) continue;
      epi[g] = g_prof.epi[i]; n[g] = g_prof.n[i]; k[g] = g_prof.k[i];
      ms[g] = 0.0; flops[g] = 0.0; count[g] = 0;
      ++ng;
    }
    ms[g] += t; flops[g] += g_prof.flops[i]; count[g] += 1;
  }
  *n_groups = ng;
  return MVF_OK;
}

// ---- unit-testable pieces of the same path ----
extern "C" int mvf_gemm_tc(int dtype, int epi, const void* A, int lda, const void* W, int ldw, const float* bias, void* C,
                           int ldc, float* resid, int ldr, void* tap, int ldt, const float* pos, const float* ls, int tpf,
                           int M, int N, int K, hipStream_t st) {
  return mvf_gemm_tc_impl(dtype, epi, A, lda, W, ldw, bias, C, ldc, resid, ldr, tap, ldt, pos, ls, tpf, M, N, K, st);
}
// mvf_gemm_tc with the LN-fold extras (bf16, K % 128 == 0): see MvfGemmLn / include/mvf_hip.h
extern "C" int mvf_gemm_tc_ln(int dtype, int epi, const void* A, int lda, const void* W, int ldw, const float* bias, void* C,
                              int ldc, float* resid, int ldr, void* tap, int ldt, const float* ls, int tpf, void* xb, int ldxb,
                              float* stats, const float* ln_mr, const float* ln_c, int M, int N, int K, hipStream_t st) {
  const MvfGemmLn ln = {xb, ldxb, stats, ln_mr, ln_c, 0, nullptr};
  return mvf_gemm_tc_impl(dtype, epi, A, lda, W, ldw, bias, C, ldc, resid, ldr, tap, ldt, nullptr, ls, tpf, M, N, K, st, 0, 0, &ln);
}
// mvf_gemm_tc_ln's consumer side (epi 0 / 1) fed with the producer's PARTIAL sums [ns][M][2] instead of (mean, rstd): the persistent
// 256x256 kernel finalizes them itself; MVF_ERR_UNSUPPORTED where it cannot (ns > 12, odd M, K < 256, pinned 128x128 kernel)
extern "C" int mvf_gemm_tc_ln_part(int epi, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                                   const float* part, int ns, float eps, const float* ln_c, int M, int N, int K, hipStream_t st) {
  MVF_CHECK_ARG(part != nullptr && ns > 0 && ns * 64 == K);      // the partial sums cover the K = D columns in 64-column slices
  const MvfGemmLn ln = {nullptr, 0, nullptr, nullptr, ln_c, 0, nullptr, nullptr, 0, part, ns, eps};
  return mvf_gemm_tc_impl(MVF_BF16, epi, A, lda, W, ldw, bias, C, ldc, nullptr, 0, nullptr, 0, nullptr, nullptr, 1, M, N, K, st, 0, 0, &ln);
}
extern "C" int mvf_gemm_fp8(int epi, const void* A, int lda, const unsigned* sa, const void* W, int ldw, const unsigned* sw,
                            const float* bias, void* C, int ldc, unsigned* c_scales, float* resid, int ldr, void* tap, int ldt,
                            const float* ls, int tpf, int M, int N, int K, hipStream_t st) {
  return mvf_gemm_fp8_impl(epi, A, lda, sa, W, ldw, sw, bias, C, ldc, c_scales, resid, ldr, tap, ldt, ls, tpf, M, N, K, st);
}
// mvf_gemm_fp8 with the LN-fold extras (include/mvf_hip.h): epi 0 consumer (ln_mr, ln_c), epi 2 producer (xq, xq_scales, stats)
extern "C" int mvf_gemm_fp8_ln(int epi, const void* A, int lda, const unsigned* sa, const void* W, int ldw, const unsigned* sw,
                               const float* bias, void* C, int ldc, float* resid, int ldr, void* tap, int ldt, const float* ls, int tpf,
                               const void* addend2, int ld2, void* xq, int ldxq, unsigned* xq_scales, float* stats, const float* ln_mr,
                               const float* ln_c, int M, int N, int K, hipStream_t st) {
  MVF_CHECK_ARG((epi == 0 && ln_mr && ln_c && !xq && !xq_scales && !stats) || (epi == 2 && xq && xq_scales && stats && !ln_mr && !ln_c));
  MvfGemmLn ln = {xq, ldxq, stats, ln_mr, ln_c, 0, nullptr};
  ln.xb_scales = xq_scales;
  return mvf_gemm_fp8_impl(epi, A, lda, sa, W, ldw, sw, bias, C, ldc, nullptr, resid, ldr, tap, ldt, ls, tpf, M, N, K, st, addend2, ld2, &ln);
}
// fp32 result of a bf16 GEMM without the in-place read-modify-write: out = A W^T + bias [+ addend]  (addend NULL: none).
// The trainable backbone blocks' linears (forward, input gradient) -- ops._LinearTC.
extern "C" int mvf_gemm_tc_f32(const void* A, int lda, const void* W, int ldw, const float* bias, float* out, int ldo,
                               const float* addend, int M, int N, int K, hipStream_t st) {
  const MvfGemmLn ln = {nullptr, 0, nullptr, nullptr, nullptr, addend ? 1 : 2, addend};
  return mvf_gemm_tc_impl(MVF_BF16, EPI_RESID, A, lda, W, ldw, bias, nullptr, 0, out, ldo, nullptr, 0, nullptr, nullptr, 1, M, N, K,
                          st, 0, 0, &ln);
}
// the fc2 step of the deferred residual on its own: resid += A W^T + bias + addend2 (bf16 [M, ld2]); tap as in mvf_gemm_tc
extern "C" int mvf_gemm_tc_resid2(const void* A, int lda, const void* W, int ldw, const float* bias, float* resid, int ldr,
                                  const void* addend2, int ld2, void* tap, int ldt, int tpf, int M, int N, int K, hipStream_t st) {
  const MvfGemmLn ln = {nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, addend2, ld2};
  return mvf_gemm_tc_impl(MVF_BF16, EPI_RESID, A, lda, W, ldw, bias, nullptr, 0, resid, ldr, tap, ldt, nullptr, nullptr, tpf, M, N, K,
                          st, 0, 0, &ln);
}
extern "C" int mvf_ln_stats_finalize(const float* part, int ns, float* mean_rstd, int rows, int D, float eps, hipStream_t st) {
  return mvf_ln_stats_finalize_impl(part, ns, mean_rstd, rows, D, eps, st);
}
extern "C" int mvf_patchify(int dtype, const float* frames, void* out, int F, int H, int W, int P, hipStream_t st) {
  return mvf_im2col_impl(dtype, frames, out, F, H, W, P, 3 * P * P, st);
}
extern "C" int mvf_layernorm_fwd(int out_dtype, const float* x, size_t in_stride, const float* g, const float* b, void* y,
                                 size_t out_stride, int rows, int D, float eps, hipStream_t st) {
  return mvf_layernorm_impl(out_dtype, x, in_stride, g, b, y, out_stride, rows, D, eps, st);
}
extern "C" int mvf_vit_attn_fwd(int dtype, const void* qkv, void* out, int F, int N, int H, int D, int variant,
                                hipStream_t st) {
  return mvf_vit_attn_impl(dtype, qkv, out, F, N, H, D, variant, st);
}
extern "C" int mvf_cast_f32_bf16(const float* in, void* out, size_t n, hipStream_t st) {
  return mvf_cast_bf16_impl(in, out, n, st);
}
