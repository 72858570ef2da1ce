// GPU-side view augmentation (include/mvf_hip.h: mvf_augment_clips): two HBM-bound passes per batch of clips
//   K1 resize_color : bilinear crop-resize (+ flip) and the colour steps up to (not including) contrast; writes the
//                     intermediate image and, when a contrast step follows, per-block partial sums of its gray value
//   K2 finish       : per-frame mean from the partials (fixed order: deterministic), contrast and the remaining colour
//                     steps while loading an LDS tile, separable Gaussian with reflect padding, grayscale, normalisation
// (the only dependency that forces two passes is contrast's per-frame mean and the blur's neighbourhood)
// Replaces the per-clip Python loop of train.preproc_views (CARL_MVF/train.py:39-53) over ~8 full-tensor ATen passes per
// op of datasets/data_augment.py:372-413.  Arithmetic follows the reference's / torchvision's float-tensor formulas
// operation by operation (no fused multiply-add contraction), see oracle/augment.py; the divisions of the hue step and of the
// final normalisation are multiplications by v_rcp_f32 / host reciprocals (<= 1 ulp each, inside the 2e-5 gate of the tests):
// as IEEE divisions they were what the finish pass spent its time on (round-2 verdict item 8).
#include "common.h"
#include "mvf_hip_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int MAXC = 8;    // clips per launch (their parameters travel in the kernel arguments)
constexpr int MAXK = 15;   // largest blur kernel extent

struct ClipParams {
  int top, left, ch, cw, flip;
  int n_color, op[4];
  float fac[4];
  int contrast_at;          // index of the contrast step, -1 if none
  int blur, nkx, nky;
  float kx[MAXK], ky[MAXK];
  int gray;
  float mean[3], istd[3];   // 1 / std
  float sy, sx;             // (float)crop_h / S, (float)crop_w / S: PyTorch's area_pixel_compute_scale, once per clip
};

struct AugArgs {
  const float* in;     // [n, T, 3, H, W]
  float* buf;          // [n, T, 3, S, S]
  float* partial;      // [n, T, NB]
  float* out;          // [n, T, 3, S, S]
  int n, T, H, W, S, NB;
  int tile_stride, hrow_stride;   // floats per channel of the finish pass's two LDS images (sized for the launch's largest blur)
  ClipParams c[MAXC];
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ __forceinline__ float tv_gray(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }
__device__ __forceinline__ float blend(float a, float b, float ratio) { return clamp01(ratio * a + (1.0f - ratio) * b); }

// torchvision _rgb2hsv / (h + f) % 1 / _hsv2rgb on one pixel
__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float f) {
  const float maxc = fmaxf(fmaxf(r, g), b), minc = fminf(fminf(r, g), b);
  const bool eqc = maxc == minc;
  const float cr = maxc - minc;
  const float s = cr * __builtin_amdgcn_rcpf(eqc ? 1.0f : maxc);
  const float icrd = __builtin_amdgcn_rcpf(eqc ? 1.0f : cr);
  const float rc = (maxc - r) * icrd, gc = (maxc - g) * icrd, bc = (maxc - b) * icrd;
  const float hr = (maxc == r) ? (bc - gc) : 0.0f;
  const float hg = ((maxc == g) && (maxc != r)) ? (2.0f + rc - bc) : 0.0f;
  const float hb = ((maxc != g) && (maxc != r)) ? (4.0f + gc - rc) : 0.0f;
  float h = (hr + hg + hb) * (1.0f / 6.0f) + 1.0f;   // in [5/6, 11/6): fmod(h, 1) = h - floor(h), exact
  h = h - floorf(h);
  h = h + f;
  h = h - floorf(h);                         // python % 1.0 on [-0.5, 1.5)
  const float v = maxc;
  const float h6 = h * 6.0f;
  const float fl = floorf(h6);
  const float fr = h6 - fl;
  const int i = (int)fl;                     // 0 .. 6 (6 only when h rounds up to 1.0: sector 0 again)
  const float p = clamp01(v * (1.0f - s));
  const float q = clamp01(v * (1.0f - s * fr));
  const float t = clamp01(v * (1.0f - (s * (1.0f - fr))));
  // sector table as selects (r: v q p p t v, g: t v v q p p, b: p p t v v q)
  r = (i == 1) ? q : ((i == 2 || i == 3) ? p : (i == 4 ? t : v));
  g = (i == 0 || i == 6) ? t : ((i == 1 || i == 2) ? v : (i == 3 ? q : p));
  b = (i == 0 || i == 1 || i == 6) ? p : (i == 2 ? t : (i == 5 ? q : v));
}

__device__ __forceinline__ void point_op(int op, float f, float& r, float& g, float& b) {
  if (op == 0) {            // adjust_brightness: blend with black
    r = blend(r, 0.0f, f); g = blend(g, 0.0f, f); b = blend(b, 0.0f, f);
  } else if (op == 2) {     // adjust_saturation: blend with the pixel's gray
    const float gr = tv_gray(r, g, b);
    r = blend(r, gr, f); g = blend(g, gr, f); b = blend(b, gr, f);
  } else if (op == 3) {
    hue_shift(r, g, b, f);
  }
}

// colour steps [lo, hi) of a clip: four fixed slots with constant indices, so that the step codes / factors are loaded into
// SGPRs once (a run-time index into the kernel-argument block is a scalar load and a wait per use, inside the per-pixel loops)
struct ColorSteps {
  int op[4];
  float fac[4];
  __device__ __forceinline__ explicit ColorSteps(const ClipParams& c) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { op[k] = c.op[k]; fac[k] = c.fac[k]; }
  }
  __device__ __forceinline__ void run(int lo, int hi, float& r, float& g, float& b) const {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k >= lo && k < hi) point_op(op[k], fac[k], r, g, b);
  }
};

// PyTorch upsample_bilinear2d, align_corners = False: source index and weight of output position d
__device__ __forceinline__ void src_index(int d, int in, float scale, int& i0, int& i1, float& l0, float& l1) {
  float s = scale * ((float)d + 0.5f) - 0.5f;
  s = s < 0.0f ? 0.0f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.0f - l1;
}

// K1: one thread per output pixel, 32 x 8 pixels per workgroup (S = 224: 7 x 28 full workgroups per frame).  Measured and
// dropped in round 3: four rows per thread (139 us against 135: the kernel queues on the gather loads, not on index
// arithmetic) and one 8-byte load per tap pair (164 us: 8-byte loads at 4-byte alignment are slower than two dwords here)
constexpr int RX = 32, RY = 8;
__global__ __launch_bounds__(256) void resize_color_kernel(AugArgs a) {
  const int ft = blockIdx.z, cidx = ft / a.T;
  const ClipParams& c = a.c[cidx];
  const int S = a.S;
  const int x = blockIdx.x * RX + (threadIdx.x & (RX - 1)), y = blockIdx.y * RY + (threadIdx.x >> 5);
  float gsum = 0.0f;
  if (x < S && y < S) {
    const int xs = c.flip ? S - 1 - x : x;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index(y, c.ch, c.sy, y0, y1, ly0, ly1);
    src_index(xs, c.cw, c.sx, x0, x1, lx0, lx1);
    const size_t plane = (size_t)a.H * a.W;
    const float* fb = a.in + ((size_t)ft * 3) * plane + (size_t)c.top * a.W + c.left;
    const int o00 = y0 * a.W + x0, o01 = y0 * a.W + x1, o10 = y1 * a.W + x0, o11 = y1 * a.W + x1;
    float v[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float* p = fb + ch * plane;
      const float p00 = p[o00], p01 = p[o01], p10 = p[o10], p11 = p[o11];
      v[ch] = ly0 * (lx0 * p00 + lx1 * p01) + ly1 * (lx0 * p10 + lx1 * p11);
    }
    const ColorSteps cs(c);
    cs.run(0, c.contrast_at >= 0 ? c.contrast_at : c.n_color, v[0], v[1], v[2]);
    float* ob = a.buf + ((size_t)ft * 3) * S * S + (size_t)y * S + x;
    ob[0] = v[0]; ob[(size_t)S * S] = v[1]; ob[(size_t)2 * S * S] = v[2];
    gsum = tv_gray(v[0], v[1], v[2]);
  }
  if (c.contrast_at >= 0) {     // block-uniform
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gsum += __shfl_xor(gsum, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gsum;
    __syncthreads();
    if (threadIdx.x == 0)
      a.partial[(size_t)ft * a.NB + blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

constexpr int TX = 64, TY = 16;
__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// The two blur passes with the tap count as a template parameter: the taps are read from the kernel arguments with constant
// indices (SGPRs, once) and the loops unroll -- as run-time loops every tap was a scalar load + wait in front of one LDS read.
// One fused multiply-add per tap (the reference's convolution goes through a vendor library whose summation is not specified
// either; inside the tests' 2e-5).
template <int NK>
__device__ __forceinline__ void blur_h(const float (&kx)[MAXK], const float* tile, float* hrow, int tst, int hst, int tw, int th) {
  float k[NK];
#pragma unroll
  for (int j = 0; j < NK; ++j) k[j] = kx[j];
  for (int i = threadIdx.x; i < th * TX; i += 256) {      // th rows x TX columns
    const int ly = i >> 6, lx = i & (TX - 1);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float* t = tile + ch * tst + ly * tw + lx;
      float acc = 0.0f;
#pragma unroll
      for (int j = 0; j < NK; ++j) acc = __builtin_fmaf(k[j], t[j], acc);
      hrow[ch * hst + i] = acc;
    }
  }
}
template <int NK>
__device__ __forceinline__ void blur_v(const float (&ky)[MAXK], const float* hrow, int hst, int tx, int ty0, float (&v)[4][3]) {
  float k[NK];
#pragma unroll
  for (int j = 0; j < NK; ++j) k[j] = ky[j];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float* hc = hrow + ch * hst + ty0 * TX + tx;
    float w[NK + 3];
#pragma unroll
    for (int j = 0; j < NK + 3; ++j) w[j] = hc[j * TX];
#pragma unroll
    for (int r = 0; r < 4; ++r) {     // output row ty0 + r: taps 0 .. NK-1 of inputs ty0 + r + j, in that order
      float acc = 0.0f;
#pragma unroll
      for (int j = 0; j < NK; ++j) acc = __builtin_fmaf(k[j], w[r + j], acc);
      v[r][ch] = acc;
    }
  }
}
#define MVF_BLUR_CASES(F, ...)                                                                                             \
  switch (nk) {                                                                                                            \
    case 1: F<1>(__VA_ARGS__); break;   case 3: F<3>(__VA_ARGS__); break;   case 5: F<5>(__VA_ARGS__); break;              \
    case 7: F<7>(__VA_ARGS__); break;   case 9: F<9>(__VA_ARGS__); break;   case 11: F<11>(__VA_ARGS__); break;            \
    case 13: F<13>(__VA_ARGS__); break; default: F<15>(__VA_ARGS__); break;                                                \
  }
__device__ __forceinline__ void blur_h_dispatch(const ClipParams& c, const float* tile, float* hrow, int tst, int hst, int tw, int th) {
  const int nk = c.nkx;     // odd, <= MAXK (validated on the host)
  MVF_BLUR_CASES(blur_h, c.kx, tile, hrow, tst, hst, tw, th)
}
__device__ __forceinline__ void blur_v_dispatch(const ClipParams& c, const float* hrow, int hst, int tx, int ty0, float (&v)[4][3]) {
  const int nk = c.nky;
  MVF_BLUR_CASES(blur_v, c.ky, hrow, hst, tx, ty0, v)
}

// K2: per-frame mean from K1's partials (fixed order), contrast + the remaining colour steps applied while the tile
// (with its blur halo, reflect-padded) is loaded into LDS, Gaussian as a horizontal then a vertical pass over LDS,
// grayscale, normalisation.  64 x 16 outputs per workgroup, four per thread (a column of four rows: the vertical pass
// walks its 4 + nky - 1 inputs once).  Halo pixels repeat the pointwise colour work of their owners (1.6 x for a 5 x 9 kernel;
// the 32 x 8 tile of rounds 1-2: 2.25 x) instead of a separate read-modify-write pass over the intermediate image.
__global__ __launch_bounds__(256) void finish_kernel(AugArgs a) {
  const int ft = blockIdx.z;                  // clip * T + frame
  const int cidx = ft / a.T;
  const ClipParams& c = a.c[cidx];
  const int S = a.S;
  extern __shared__ float dyn_lds[];          // tile[3][tile_stride] | hrow[3][hrow_stride]
  float* const tile = dyn_lds;
  float* const hrow = dyn_lds + 3 * a.tile_stride;
  const int tst = a.tile_stride, hst = a.hrow_stride;
  __shared__ float red[4];
  __shared__ int colmap[TX + MAXK - 1], rowoff[TY + MAXK - 1];
  const bool post = c.contrast_at >= 0;       // block-uniform
  float ps = 0.0f, ps_more = 0.0f;
  if (post) {                                 // K1's gray partial sums of this frame: in flight under the tile loads
    const float* pp = a.partial + (size_t)ft * a.NB;
    if ((int)threadIdx.x < a.NB) ps = pp[threadIdx.x];
    for (int i = threadIdx.x + 256; i < a.NB; i += 256) ps_more += pp[i];     // (S > 512 only)
  }
  const int hx = c.blur ? c.nkx >> 1 : 0, hy = c.blur ? c.nky >> 1 : 0;
  const int tw = TX + 2 * hx, th = TY + 2 * hy;
  const float* ib = a.buf + (size_t)ft * 3 * S * S;
  // source column / row offset of every tile column / row, once per workgroup.  Reflect, then clamp: positions beyond the
  // edge of a partial tile only feed pixels that are not stored
  if ((int)threadIdx.x < tw) colmap[threadIdx.x] = min(max(reflect(blockIdx.x * TX + (int)threadIdx.x - hx, S), 0), S - 1);
  if (threadIdx.x >= 128 && (int)threadIdx.x - 128 < th)
    rowoff[threadIdx.x - 128] = min(max(reflect(blockIdx.y * TY + (int)threadIdx.x - 128 - hy, S), 0), S - 1) * S;
  __syncthreads();
  {
    // i = ly * tw + lx walked with stride 256 without a division per element
    const int qstep = 256 / tw, rstep = 256 - qstep * tw;
    int ly = threadIdx.x / tw, lx = threadIdx.x - ly * tw;
    const ColorSteps cs(c);
    const int clo = c.contrast_at + 1, chi = c.n_color;
    float cf = 1.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) cf = (k == c.contrast_at) ? cs.fac[k] : cf;
    // all of a thread's loads first (up to MAXIT x 3 in flight) straight into the LDS tile, then -- a rolled loop, one copy of
    // the colour code -- contrast and the later colour steps on the thread's own elements in place
    constexpr int MAXIT = ((TY + MAXK - 1) * (TX + MAXK - 1) + 255) / 256;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int i = threadIdx.x + it * 256;
      if (i < tw * th) {
        const float* p = ib + rowoff[ly] + colmap[lx];
        tile[i] = p[0]; tile[tst + i] = p[(size_t)S * S]; tile[2 * tst + i] = p[(size_t)2 * S * S];
      }
      lx += rstep; ly += qstep;
      if (lx >= tw) { lx -= tw; ++ly; }
    }
    if (post) {
      // per-frame mean, fixed order (deterministic): every thread forms it from the four wave sums itself
      ps += ps_more;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ps += __shfl_xor(ps, o, 64);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ps;
      __syncthreads();
      const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)(S * S);
#pragma unroll 1
      for (int i = threadIdx.x; i < tw * th; i += 256) {
        float r = tile[i], g = tile[tst + i], b = tile[2 * tst + i];
        r = blend(r, mean, cf); g = blend(g, mean, cf); b = blend(b, mean, cf);
        cs.run(clo, chi, r, g, b);
        tile[i] = r; tile[tst + i] = g; tile[2 * tst + i] = b;
      }
    }
  }
  __syncthreads();
  const int tx = threadIdx.x & (TX - 1), ty0 = (threadIdx.x >> 6) * 4;
  float v[4][3];
  if (c.blur) {
    blur_h_dispatch(c, tile, hrow, tst, hst, tw, th);
    __syncthreads();
    blur_v_dispatch(c, hrow, hst, tx, ty0, v);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) v[r][ch] = tile[ch * tst + (ty0 + r) * tw + tx];
  }
  const int x = blockIdx.x * TX + tx;
  if (x >= S) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = blockIdx.y * TY + ty0 + r;
    if (y >= S) break;
    if (c.gray) {
      const float gch = 0.299f * v[r][0] + 0.587f * v[r][1] + 0.114f * v[r][2];
      v[r][0] = v[r][1] = v[r][2] = gch;
    }
    float* ob = a.out + (size_t)ft * 3 * S * S + (size_t)y * S + x;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) ob[(size_t)ch * S * S] = (v[r][ch] - c.mean[ch]) * c.istd[ch];
  }
}

// torchvision _get_gaussian_kernel1d in fp32: linspace(-half, half, k), exp(-0.5 (x / sigma)^2), / sum
void gaussian1d(int k, float sigma, float* w) {
  const float half = (float)(k - 1) * 0.5f;
  const float step = k > 1 ? (2.0f * half) / (float)(k - 1) : 0.0f;
  float sum = 0.0f;
  for (int i = 0; i < k; ++i) {
    // torch.linspace fills symmetrically from both ends: start + i*step for the first half, end - (k-1-i)*step after it
    const float x = i < k / 2 ? -half + step * (float)i : half - step * (float)(k - 1 - i);
    const float q = x / sigma;
    w[i] = expf(-0.5f * (q * q));
    sum += w[i];
  }
  for (int i = 0; i < k; ++i) w[i] = w[i] / sum;
}

}  // namespace

extern "C" size_t mvf_augment_workspace_bytes(int n_clips, int T, int S) {
  if (n_clips <= 0 || T <= 0 || S <= 0) return 0;
  const size_t nb = (size_t)((S + RX - 1) / RX) * ((S + RY - 1) / RY);
  return ((size_t)n_clips * T * 3 * S * S + (size_t)n_clips * T * nb) * sizeof(float);
}

extern "C" int mvf_augment_clips(const float* in, float* out, int n_clips, int T, int H, int W, int S,
                                 const MvfAugmentParams* params, void* workspace, size_t ws_bytes, hipStream_t st) {
  MVF_CHECK_ARG(in && out && params && workspace && n_clips > 0 && T > 0 && H > 0 && W > 0 && S > 0);
  MVF_CHECK_ARG(ws_bytes >= mvf_augment_workspace_bytes(n_clips, T, S));
  const int gx1 = (S + RX - 1) / RX, gy1 = (S + RY - 1) / RY, NB = gx1 * gy1;
  float* buf = reinterpret_cast<float*>(workspace);
  float* partial = buf + (size_t)n_clips * T * 3 * S * S;
  for (int i = 0; i < n_clips; ++i) {   // validate everything before the first launch
    const MvfAugmentParams& p = params[i];
    MVF_CHECK_ARG(p.crop_h > 0 && p.crop_w > 0 && p.crop_top >= 0 && p.crop_left >= 0 && p.crop_top + p.crop_h <= H &&
                  p.crop_left + p.crop_w <= W);
    MVF_CHECK_ARG(p.n_color >= 0 && p.n_color <= 4);
    int seen = 0;
    for (int k = 0; k < p.n_color; ++k) {
      MVF_CHECK_ARG(p.color_op[k] >= 0 && p.color_op[k] <= 3 && !(seen & (1 << p.color_op[k])));
      seen |= 1 << p.color_op[k];
    }
    if (p.blur_sigma > 0.0f)
      MVF_CHECK_ARG(p.blur_kx > 0 && p.blur_ky > 0 && (p.blur_kx & 1) && (p.blur_ky & 1) && p.blur_kx <= MAXK &&
                    p.blur_ky <= MAXK && p.blur_kx / 2 < S && p.blur_ky / 2 < S);   // reflect padding needs pad < size
    for (int ch = 0; ch < 3; ++ch) MVF_CHECK_ARG(p.std[ch] != 0.0f);
  }
  for (int c0 = 0; c0 < n_clips; c0 += MAXC) {
    const int n = std::min(MAXC, n_clips - c0);
    AugArgs a;
    a.in = in + (size_t)c0 * T * 3 * H * W;
    a.buf = buf + (size_t)c0 * T * 3 * S * S;
    a.partial = partial + (size_t)c0 * T * NB;
    a.out = out + (size_t)c0 * T * 3 * S * S;
    a.n = n; a.T = T; a.H = H; a.W = W; a.S = S; a.NB = NB;
    for (int i = 0; i < n; ++i) {
      const MvfAugmentParams& p = params[c0 + i];
      ClipParams& c = a.c[i];
      c.top = p.crop_top; c.left = p.crop_left; c.ch = p.crop_h; c.cw = p.crop_w; c.flip = p.flip != 0;
      c.n_color = p.n_color; c.contrast_at = -1;
      for (int k = 0; k < 4; ++k) {
        c.op[k] = k < p.n_color ? p.color_op[k] : 0;
        c.fac[k] = k < p.n_color ? p.color_factor[k] : 1.0f;
        if (k < p.n_color && p.color_op[k] == 1) c.contrast_at = k;
      }
      c.blur = p.blur_sigma > 0.0f; c.nkx = c.blur ? p.blur_kx : 1; c.nky = c.blur ? p.blur_ky : 1;
      for (int k = 0; k < MAXK; ++k) c.kx[k] = c.ky[k] = 0.0f;
      if (c.blur) { gaussian1d(c.nkx, p.blur_sigma, c.kx); gaussian1d(c.nky, p.blur_sigma, c.ky); }
      c.gray = p.gray != 0;
      for (int ch = 0; ch < 3; ++ch) { c.mean[ch] = p.mean[ch]; c.istd[ch] = 1.0f / p.std[ch]; }
      c.sy = (float)p.crop_h / (float)S; c.sx = (float)p.crop_w / (float)S;
    }
    hipLaunchKernelGGL(resize_color_kernel, dim3(gx1, gy1, T * n), dim3(256), 0, st, a);
    MVF_LAUNCH_CHECK();
    int mkx = 1, mky = 1;
    for (int i = 0; i < n; ++i) { mkx = std::max(mkx, a.c[i].nkx); mky = std::max(mky, a.c[i].nky); }
    a.tile_stride = (TY + mky - 1) * (TX + mkx - 1);
    a.hrow_stride = (TY + mky - 1) * TX;
    const size_t lds = (size_t)3 * (a.tile_stride + a.hrow_stride) * sizeof(float);   // <= 51 KB at 15 x 15
    hipLaunchKernelGGL(finish_kernel, dim3((S + TX - 1) / TX, (S + TY - 1) / TY, T * n), dim3(256), lds, st, a);
    MVF_LAUNCH_CHECK();
  }
  return MVF_OK;
}
