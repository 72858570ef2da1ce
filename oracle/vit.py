"""Oracle: ViT backbone forward (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Restates the arithmetic of timm==0.9.2 `VisionTransformer.forward` for the
`vit_*_patch{8,16}_224.dino` family, which the reference obtains through
`timm.create_model(name, pretrained=True)` (CARL_MVF/models/transformer.py:59)
and taps with forward hooks on `blocks.{i}` (transformer.py:306-333).  timm is an
un-vendored dependency and is absent offline, so this file follows its
published algorithm:

    x = Conv2d(3, D, k=P, s=P)(img).flatten(2).transpose(1, 2)       # patch_embed
    x = cat([cls_token, x], 1) + pos_embed                            # _pos_embed
    for blk:  x = x + ls1(attn(LN(x)));  x = x + ls2(mlp(LN(x)))      # pre-LN, eps 1e-6
    attn: qkv = Linear(D, 3D); reshape(B, N, 3, H, hd); softmax(q k^T hd^-.5) v; Linear
    mlp : Linear(D, 4D) -> exact (erf) GELU -> Linear(4D, D)
    out = LN(x)[:, 0]                                                  # global_pool='token'

`emulate='bf16'` restates the SAME algorithm with a round-to-nearest-even bf16 rounding at exactly the points where
the product's bf16 mode stores bf16 (csrc/vit_fwd.hip, vit_attn.hip, gemm_tc_epi.h): GEMM weights, the im2col'd
patches, qkv, the softmax probabilities fed to P.V (and to their row sum, accumulated in fp32), the attention output, fc1+GELU output
and the tapped block outputs; accumulation, biases, position embedding, LayerScale and the residual stream stay fp32.
LayerNorm is FOLDED into the GEMM that consumes it where the product folds it (include/mvf_hip.h qkv_c / fc1_c; default
MVF_LN_FOLD=2: norm1 of blocks > 0):   LN(x) W^T + b  ==  rstd * (x W'^T - mean * c) + d   with W' = gamma (.) W,
c[n] = sum_k W'[n,k], d = b + W beta; the rounding points are then xb = bf16(x) and bf16(W') (c is summed over the
ROUNDED W').  Everywhere else (block 0's norm1, every norm2) the LayerNorm output is rounded instead.  The attention branch's
output (proj + bias, LayerScale gamma_1 folded into proj's weights and bias BEFORE their rounding / quantisation) is rounded to
bf16 before the residual add (the product stores it with the plain GEMM epilogue and adds it in LayerNorm 2 and in the fc2
epilogue: csrc/vit_fwd.hip `defer`; D % 128 == 0; also in fp8 mode).  The product's other
settings: `emulate='bf16_fold12'` (MVF_LN_FOLD=1: norm2 folded too), `'bf16_nofold'` (MVF_LN_FOLD=0).
In every emulating mode, at the token counts `q_prescaled(n)` names (the product's streamed attention kernel: all but 193 .. 208), the q
rows of the qkv weights and bias carry log2(e) / 8 BEFORE their rounding (timm Attention's `q * self.scale` and the softmax's base change
folded into the frozen weights: ops.PackedViT), scores are q' k^T and probabilities 2^(s - max).
`emulate='fp8'`: the four GEMMs of a block on MX-fp8 operands (mx_quant), norm1 of blocks > 0 folded into the qkv GEMM on the MX-fp8
UN-normalised residual row (quantised from fp32 by the previous block's fc2 epilogue; W' = MX-fp8(gamma (.) W), c its row sums);
`'fp8_nofold'` (MVF_FP8_LN_FOLD=0): the LayerNorm output quantised in front of every GEMM.  It is the checker for the benchmarked dtype (tests/test_gpu_*: tight gates instead of "bf16 is somewhere
near fp32").

Weights are a flat dict keyed with timm's state-dict names.  PARITY UNPINNED by
the reference for this file (no reference test / vector exists); cross-checked
against HuggingFace `transformers.ViTModel` in tests/test_oracle_vit.py.
"""
import math
import os
import torch
import torch.nn.functional as F


def vit_dims(name):
    """(embed_dim, depth, heads, patch, has_layerscale) for the model names the
    reference accepts (CARL_MVF/models/transformer.py:43-54)."""
    table = {
        'vit_small_patch16_224.dino': (384, 12, 6, 16, False),
        'vit_small_patch8_224.dino': (384, 12, 6, 8, False),
        'vit_base_patch16_224.dino': (768, 12, 12, 16, False),
        'vit_base_patch8_224.dino': (768, 12, 12, 8, False),
        'vit_small_patch14_dinov2.lvd142m': (384, 12, 6, 14, True),
        'vit_base_patch14_dinov2.lvd142m': (768, 12, 12, 14, True),
        'vit_large_patch14_dinov2.lvd142m': (1024, 24, 16, 14, True),
    }
    return table[name]


def init_vit_weights(dim, depth, patch, img=224, seed=0, dtype=torch.float32, layerscale=False):
    """Seeded random ViT weights (trunc-normal 0.02 like timm's init; LN weights
    jittered so that a swapped gamma/beta would be caught)."""
    g = torch.Generator().manual_seed(seed)
    n = (img // patch) ** 2 + 1

    def tn(*shape, std=0.02):
        return (torch.randn(*shape, generator=g, dtype=torch.float64) * std).clamp_(-2 * std, 2 * std).to(dtype)

    w = {
        'cls_token': tn(1, 1, dim),
        'pos_embed': tn(1, n, dim),
        'patch_embed.proj.weight': tn(dim, 3, patch, patch),
        'patch_embed.proj.bias': tn(dim),
        'norm.weight': 1.0 + tn(dim, std=0.1),
        'norm.bias': tn(dim, std=0.1),
    }
    for i in range(depth):
        p = 'blocks.%d.' % i
        w[p + 'norm1.weight'] = 1.0 + tn(dim, std=0.1)
        w[p + 'norm1.bias'] = tn(dim, std=0.1)
        w[p + 'attn.qkv.weight'] = tn(3 * dim, dim)
        w[p + 'attn.qkv.bias'] = tn(3 * dim)
        w[p + 'attn.proj.weight'] = tn(dim, dim)
        w[p + 'attn.proj.bias'] = tn(dim)
        w[p + 'norm2.weight'] = 1.0 + tn(dim, std=0.1)
        w[p + 'norm2.bias'] = tn(dim, std=0.1)
        w[p + 'mlp.fc1.weight'] = tn(4 * dim, dim)
        w[p + 'mlp.fc1.bias'] = tn(4 * dim)
        w[p + 'mlp.fc2.weight'] = tn(dim, 4 * dim)
        w[p + 'mlp.fc2.bias'] = tn(dim)
        if layerscale:
            w[p + 'ls1.gamma'] = 1.0 + tn(dim, std=0.1)
            w[p + 'ls2.gamma'] = 1.0 + tn(dim, std=0.1)
    return w


def patchify(img, patch):
    """[F,3,H,W] -> [F, (H/P)*(W/P), 3*P*P] with k = c*P*P + ky*P + kx, the
    flattening order of a Conv2d weight [D,3,P,P]."""
    f, c, h, w = img.shape
    gh, gw = h // patch, w // patch
    x = img.reshape(f, c, gh, patch, gw, patch)
    x = x.permute(0, 2, 4, 1, 3, 5)  # f, gh, gw, c, ky, kx
    return x.reshape(f, gh * gw, c * patch * patch)


def layer_norm(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def bf16_round(x):
    """Round-to-nearest-even to bf16 and back (what v_cvt_pk_bf16_f32 does to a stored value)."""
    return x.to(torch.bfloat16).to(x.dtype)


def _bf16_round_true(x):
    return x.to(torch.bfloat16).to(x.dtype)


def f16_round(x):
    """Round-to-nearest-even to IEEE fp16 and back (v_cvt_pk_f16_f32); overflows to inf beyond 65504 like the device."""
    return x.to(torch.float16).to(x.dtype)


def _ident(x):
    return x


def mx_quant(t):
    """MX-fp8 quantise-dequantise along the last dimension (the product's fp8 mode, csrc/mxfp8.hip): blocks of 32
    consecutive elements share a power-of-two scale -- the smallest with amax / scale <= 448 -- and the scaled values are
    rounded to OCP e4m3 (round to nearest even)."""
    shp = t.shape
    assert shp[-1] % 32 == 0, shp
    b = t.reshape(*shp[:-1], shp[-1] // 32, 32).float()
    amax = b.abs().amax(-1, keepdim=True)
    m, e = torch.frexp(amax)                       # amax = m * 2^e, m in [0.5, 1)
    exp = (e - 9 + (m > 0.875).to(e.dtype)).clamp(-127, 126)
    scale = torch.ldexp(torch.ones_like(amax), exp)
    q = (b / scale).to(torch.float8_e4m3fn).float() * scale
    return q.reshape(shp).to(t.dtype)


def vit_block(x, w, p, heads, eps=1e-6, emulate=None):
    if emulate in ('fp8', 'fp8_nofold'):
        # (fp8 mode needs D % 256 == 0 anyway.)  'fp8': the product's default -- norm1 of every block but the first folded into the
        # MX-fp8 qkv GEMM (MVF_FP8_LN_FOLD); 'fp8_nofold': a LayerNorm + quantiser pass in front of every GEMM (per-block runs)
        return vit_block_bf16(x, w, p, heads, eps, mx=True, defer_proj=True, fold1=emulate == 'fp8' and p != 'blocks.0.')
    if emulate in ('bf16', 'bf16_fold12', 'bf16_nofold'):
        fold = emulate != 'bf16_nofold' and x.shape[-1] % 128 == 0
        # the product defers the attention branch's residual add (bf16-rounded proj output) unless norm2 is folded / LayerScale
        return vit_block_bf16(x, w, p, heads, eps, fold1=fold and p != 'blocks.0.', fold2=fold and emulate == 'bf16_fold12',
                              defer_proj=emulate != 'bf16_fold12' and x.shape[-1] % 128 == 0)
    f, n, d = x.shape
    hd = d // heads
    h = layer_norm(x, w[p + 'norm1.weight'], w[p + 'norm1.bias'], eps)
    qkv = h @ w[p + 'attn.qkv.weight'].t() + w[p + 'attn.qkv.bias']
    qkv = qkv.reshape(f, n, 3, heads, hd).permute(2, 0, 3, 1, 4)  # 3, f, H, n, hd
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q * hd ** -0.5) @ k.transpose(-1, -2)
    a = torch.softmax(s, dim=-1) @ v  # f, H, n, hd
    a = a.transpose(1, 2).reshape(f, n, d)
    a = a @ w[p + 'attn.proj.weight'].t() + w[p + 'attn.proj.bias']
    if p + 'ls1.gamma' in w:
        a = a * w[p + 'ls1.gamma']
    x = x + a
    h = layer_norm(x, w[p + 'norm2.weight'], w[p + 'norm2.bias'], eps)
    h = gelu_erf(h @ w[p + 'mlp.fc1.weight'].t() + w[p + 'mlp.fc1.bias'])
    h = h @ w[p + 'mlp.fc2.weight'].t() + w[p + 'mlp.fc2.bias']
    if p + 'ls2.gamma' in w:
        h = h * w[p + 'ls2.gamma']
    return x + h


def rowsum_rounded(n):
    """Softmax normalisation of the 16-bit emulation for n tokens: True = the row sum runs over the probabilities after their rounding
    to the operand type (what the device's product kernels do, beside P.V on the matrix pipe: the one-block kernels of 193..208 tokens
    since round 4, the streamed kernel of every other n since round 6), False = over the fp32 values.
    Mirrors `mvf_vit_attn_rowsum_rounded` (include/mvf_hip.h); tests/test_abi.py holds the two together for every n up to 2 048."""
    return n > 0


QS = math.log2(math.e) / 8.0      # head_dim 64: q * 64^-0.5, and the softmax's base changed to 2


def q_prescaled(n):
    """True where the product packs a frozen 16-bit / fp8 backbone of n tokens per frame with the q rows of the qkv weights and bias
    pre-scaled by log2(e) / 8 (in fp64, before their one rounding / quantisation): the attention kernel then takes q k^T as a base-2
    exponent.  Mirrors `mvf_vit_attn_q_prescaled` (include/mvf_hip.h: every token count the streamed kernel serves, i.e. outside
    193 .. 208; MVF_ATTN_QS=0 switches it off on both sides); tests/test_abi.py holds the two together."""
    return n > 0 and not 193 <= n <= 208 and os.environ.get('MVF_ATTN_QS', '1')[:1] != '0'


def ln_linear_bf16(x, g, beta, W, b, eps, fold):
    """Linear(LayerNorm(x)) before the output rounding, bf16 mode.  fold: the product's folded form (module docstring),
    statistics as its kernels take them (sum and sum of squares of the fp32 row, biased variance E[x^2] - mean^2)."""
    r = bf16_round
    if not fold:
        return r(layer_norm(x, g, beta, eps)) @ r(W).t() + b
    mean = x.mean(-1, keepdim=True)
    var = ((x * x).mean(-1, keepdim=True) - mean * mean).clamp_min(0.0)
    rstd = 1.0 / torch.sqrt(var + eps)
    Wp = r(W * g[None, :])
    c = Wp.double().sum(1).to(x.dtype)
    d = (b.double() + W.double() @ beta.double()).to(x.dtype)
    return rstd * (r(x) @ Wp.t() - mean * c) + d


def vit_block_bf16(x, w, p, heads, eps=1e-6, fold1=False, fold2=False, mx=False, defer_proj=False, prescale=None):
    """vit_block with the bf16 mode's rounding points (module docstring); x is the fp32 residual stream.
    mx: the product's fp8 mode -- both operands of the four GEMMs quantised to MX-fp8 (mx_quant along k: LayerNorm outputs,
    attention output and fc1+GELU output after their bf16 rounding, weights once), everything else as in bf16 mode; fold1: norm1 folded
    into the qkv GEMM on the MX-fp8 un-normalised residual row (below), never fold2."""
    r = bf16_round
    f, n, d = x.shape
    hd = d // heads
    prescale = q_prescaled(n) if prescale is None else prescale
    Wqkv, bqkv = w[p + 'attn.qkv.weight'], w[p + 'attn.qkv.bias']
    if prescale:      # the q rows carry log2(e) / 8 before anything is rounded (module docstring)
        assert hd == 64
        rs = torch.ones(3 * d, dtype=torch.float64)
        rs[:d] = QS
        Wqkv, bqkv = (Wqkv.double() * rs[:, None]).to(Wqkv.dtype), (bqkv.double() * rs).to(bqkv.dtype)

    def lin(a, wn, bn, row_scale=None):
        W, b = (Wqkv, bqkv) if wn == 'attn.qkv.weight' else (w[p + wn], w[p + bn])
        if row_scale is not None:         # LayerScale folded into the layer: (gamma (.) W, gamma (.) b), rounded AFTER the fold
            W, b = W * row_scale[:, None], b * row_scale
        if mx:
            return mx_quant(a) @ mx_quant(W).t() + b
        return a @ r(W).t() + b
    if mx and fold1:
        # the fold of ln_linear_bf16 on MX-fp8 operands: the previous block's fc2 epilogue quantised the fp32 residual row itself
        # (un-normalised), the weights are MX-fp8(gamma (.) W), c the row sums of exactly those
        g, beta, W, b = w[p + 'norm1.weight'], w[p + 'norm1.bias'], Wqkv, bqkv
        mean = x.mean(-1, keepdim=True)
        rstd = 1.0 / torch.sqrt(((x * x).mean(-1, keepdim=True) - mean * mean).clamp_min(0.0) + eps)
        Wp = mx_quant(W * g[None, :])
        c = Wp.double().sum(1).to(x.dtype)
        dvec = (b.double() + W.double() @ beta.double()).to(x.dtype)
        qkv = r(rstd * (mx_quant(x) @ Wp.t() - mean * c) + dvec)
    elif mx:
        qkv = r(lin(layer_norm(x, w[p + 'norm1.weight'], w[p + 'norm1.bias'], eps), 'attn.qkv.weight', 'attn.qkv.bias'))
    else:
        qkv = r(ln_linear_bf16(x, w[p + 'norm1.weight'], w[p + 'norm1.bias'], Wqkv, bqkv, eps, fold1))
    qkv = qkv.reshape(f, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    if prescale:
        s = q @ k.transpose(-1, -2)                        # already the base-2 exponent
        pr = torch.exp2(s - s.max(-1, keepdim=True)[0])
    else:
        s = (q @ k.transpose(-1, -2)) * hd ** -0.5
        pr = torch.exp(s - s.max(-1, keepdim=True)[0])
    prr = r(pr)                                            # P in bf16 for P.V
    # the row sum: over the SAME rounded values where the product's kernel takes it on the matrix pipe (a fifth P.V column of ones,
    # fp32 accumulation: the 193..208-token kernels of vit_attn.hip / vit_qkv_attn.hip), over the unrounded fp32 values in its
    # streamed kernel for every other sequence length
    rs = prr.sum(-1, keepdim=True) if rowsum_rounded(n) else pr.sum(-1, keepdim=True)
    a = r((prr @ v) / rs)
    a = a.transpose(1, 2).reshape(f, n, d)
    if defer_proj:
        # deferred residual: the branch output (LayerScale folded into proj's weights and bias) is stored as bf16 before it is
        # added (module docstring)
        x = x + r(lin(a, 'attn.proj.weight', 'attn.proj.bias', w.get(p + 'ls1.gamma')))
    else:
        a = lin(a, 'attn.proj.weight', 'attn.proj.bias')
        if p + 'ls1.gamma' in w:
            a = a * w[p + 'ls1.gamma']
        x = x + a
    if mx:   # quantised straight from the fp32 GELU value in fc1's epilogue (lin() applies mx_quant): no bf16 rounding
        h = gelu_erf(lin(layer_norm(x, w[p + 'norm2.weight'], w[p + 'norm2.bias'], eps), 'mlp.fc1.weight', 'mlp.fc1.bias'))
    else:
        h = r(gelu_erf(ln_linear_bf16(x, w[p + 'norm2.weight'], w[p + 'norm2.bias'], w[p + 'mlp.fc1.weight'],
                                      w[p + 'mlp.fc1.bias'], eps, fold2)))
    h = lin(h, 'mlp.fc2.weight', 'mlp.fc2.bias')
    if p + 'ls2.gamma' in w:
        h = h * w[p + 'ls2.gamma']
    return x + h


def vit_embed(img, w, patch, emulate=None):
    """patch_embed + cls + pos: [F,3,H,W] -> [F, 1+N, D]."""
    dim = w['patch_embed.proj.weight'].shape[0]
    r = bf16_round if emulate else _ident
    x = r(patchify(img, patch)) @ r(w['patch_embed.proj.weight'].reshape(dim, -1)).t() + w['patch_embed.proj.bias']
    cls = w['cls_token'].expand(x.shape[0], -1, -1)
    return torch.cat([cls, x], 1) + w['pos_embed']


def vit_forward(img, w, heads, patch, taps=(3, 7, 11), eps=1e-6, first_block=0, last_block=None, x_in=None,
                emulate=None):
    """Returns (features, cls_out):
      features [F, 1+N, D*len(taps)]: outputs of blocks `taps`, channel-concatenated
          (FeatureExtractor, CARL_MVF/models/transformer.py:322-333; CLS row kept,
          dropped later at transformer.py:204)
      cls_out  [F, D]: final LN, token 0 (timm forward_head with global_pool='token',
          num_classes=0)
    first_block/last_block/x_in restate the ViTFrontEnd/ViTBackEnd split
    (transformer.py:342-392).
    """
    depth = 1 + max(int(k.split('.')[1]) for k in w if k.startswith('blocks.'))
    last_block = depth if last_block is None else last_block
    if emulate == 'fp16':
        # MI355X.COMPUTE_DTYPE fp16 (the reference's own autocast dtype, CARL_MVF/train.py:113,301): the data flow and rounding POINTS
        # of the bf16 mode with IEEE fp16 as the 16-bit format at every one of them -- except the tapped block outputs, which the
        # product still hands to the head as bf16
        global bf16_round
        saved, bf16_round = bf16_round, f16_round
        try:
            return vit_forward(img, w, heads, patch, taps, eps, first_block, last_block, x_in, emulate='bf16')
        finally:
            bf16_round = saved
    assert emulate in (None, 'bf16', 'bf16_fold12', 'bf16_nofold', 'fp8', 'fp8_nofold'), emulate
    x = vit_embed(img, w, patch, emulate) if x_in is None else x_in
    feats = {}
    for i in range(first_block, last_block):
        x = vit_block(x, w, 'blocks.%d.' % i, heads, eps, emulate)
        if i in taps:
            feats[i] = _bf16_round_true(x) if emulate else x      # taps are stored as bf16 in every reduced-precision mode
    out = layer_norm(x, w['norm.weight'], w['norm.bias'], eps)[:, 0] if last_block == depth else x
    features = torch.cat([feats[i] for i in taps], dim=2) if taps else None
    return features, out
