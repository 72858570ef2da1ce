"""Oracle: MV-Former head (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Functional CPU restatement (plain torch ops, differentiable so that
`torch.autograd` gives the reference gradients) of

  attention                     CARL_MVF/models/utils.py:11-44
  MultiheadedAttention          CARL_MVF/models/utils.py:47-108
  generate_sincos_embedding     CARL_MVF/models/utils.py:113-126
  PositionalEncoder             CARL_MVF/models/utils.py:128-145
  ResidualConnection / FFN /    CARL_MVF/models/utils.py:147-242
    EncoderLayer / Encoder
  LSTPCrossAtt                  CARL_MVF/models/mvformer.py:275-414
  LearnableTokenPooling         CARL_MVF/models/mvformer.py:207-266
  MultiEntityTransformerEmbModel CARL_MVF/models/mvformer.py:15-200
  MLPHead                       CARL_MVF/models/resnet_c2d.py:112-126

Parameters are flat dicts keyed with the reference's state-dict names (relative
to the module), so a reference `state_dict()` can be fed in unchanged.  PINNED by
tests/golden/*.npz (generated from the imported reference modules by
tests/golden/gen_golden.py).
"""
import math
import numpy as np
import torch


# ReLU kink bookkeeping for gradient parity checks.  relu'(x) jumps at x = 0, so where a pre-activation lies within
# rounding distance of 0 an fp32 device run and this fp64 run may legitimately sit on different sides: the forward
# values agree to ~1e-6 but that unit's gradient contribution is all-or-nothing.  With KINK['delta'] > 0 every relu()
# below records the elements with |x| < delta as (site, flat index); elements listed in KINK['flip'] are evaluated on
# the OTHER side of the kink.  A test may then accept device gradients that match the oracle for SOME assignment of the
# recorded near-kink units.  Off by default (delta 0, no flips): plain torch.relu.
KINK = {'delta': 0.0, 'near': [], 'flip': frozenset(), 'site': 0}


def kink_reset(delta=0.0, flip=()):
    KINK.update(delta=float(delta), near=[], flip=frozenset(flip), site=0)


def relu(x):
    site = KINK['site']
    KINK['site'] = site + 1
    if KINK['delta'] > 0:
        for i in (x.detach().reshape(-1).abs() < KINK['delta']).nonzero().reshape(-1).tolist():
            KINK['near'].append((site, i))
    flips = [i for (s_, i) in KINK['flip'] if s_ == site]
    if not flips:
        return torch.relu(x)
    keep = (x.detach() > 0).reshape(-1).clone()
    for i in flips:
        keep[i] = ~keep[i]
    return x * keep.reshape(x.shape).to(x.dtype)


class HeadCfg:
    """The cfg.MODEL.EMBEDDER_MODEL / cfg.TRAIN keys the head reads, with the
    reference's defaults for absent keys (mvformer.py:23-58,100-115)."""

    def __init__(self, **kw):
        self.nst = 5                 # SMART_TOKENS
        self.nsdt = 0                # SMART_DYNAMIC_TOKENS
        self.spc = 384               # SMART_POOL_CHANNELS
        self.one_hot = 'none'        # SMART_ONE_HOT
        self.smart_final = 'max'     # SMART_FINAL
        self.num_heads = 8           # NUM_HEADS
        self.num_layers = 3          # NUM_LAYERS
        self.train_len = 32          # TRAIN.NUM_FRAMES
        self.dyn_ctrl = 'separate'   # DYNAMIC_CTRL
        self.ln_keys = False         # SMART_LN_KEYS
        self.val_pass = False        # VAL_PASS
        self.disjoint = False        # SMART_DISJOINT
        self.fwb = False             # FIXED_WIDTH_BASELINE
        self.n_taps = 1              # len(SMART_FEATS.split(','))
        self.bn_eps = 1e-5
        self.bn_momentum = 0.1
        self.ln_eps = 1e-5
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------
# Emulation of the 16-bit head (csrc/head_chain.hip; MI355X.HEAD_DTYPE bf16 / fp16): the SAME restatement with a round-to-nearest-even
# rounding of both GEMM operands of the Linears the device runs on the 16-bit matrix cores -- forward (x, W), input gradient
# (dy, W) and weight gradient (dy, x); the bias gradient sums the rounded dy (the device sums the operand it has).  Everything
# else (accumulation, bias, LayerNorm, BatchNorm, softmax, residual stream) stays in the oracle's precision.
#   mode 'bf16': every operand rounded to bf16.
#   mode 'fp16': the FORWARD operands rounded to IEEE fp16; the gradient products as in bf16 mode -- dy and W rounded to bf16, and the
#                saved activation x is the forward's fp16 value rounded once more to bf16 (the forward kernel writes its transposed
#                operand image that way).
# `prefixes`: the parameter-name prefixes of the Linears concerned (None: all).  Off by default: plain matmul.
EMU = {'mode': None, 'prefixes': None}


def emulate_head(mode=None, prefixes=None):
    assert mode in (None, 'bf16', 'fp16'), mode
    EMU.update(mode=mode, prefixes=None if prefixes is None else tuple(prefixes))


class emulating:
    """with emulating(prefixes[, mode]): ...  -- 16-bit emulation of the Linears under these name prefixes (empty / None: plain oracle)."""

    def __init__(self, prefixes, mode='bf16'):
        self.prefixes = tuple(prefixes) if prefixes else None
        self.mode = mode

    def __enter__(self):
        self.saved = dict(EMU)
        if self.prefixes:
            emulate_head(self.mode, self.prefixes)
        else:
            emulate_head(None)

    def __exit__(self, *exc):
        EMU.update(self.saved)
        return False


def bf16r(t):
    return t.to(torch.bfloat16).to(t.dtype)


def f16r(t):
    return t.to(torch.float16).to(t.dtype)


class _EmuLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, f16):
        if f16:
            xf, wf = f16r(x), f16r(w)
            ctx.save_for_backward(bf16r(xf), bf16r(w))
        else:
            xf, wf = bf16r(x), bf16r(w)
            ctx.save_for_backward(xf, wf)
        ctx.has_b = b is not None
        y = xf @ wf.t()
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        g = bf16r(dy)
        g2, x2 = g.reshape(-1, g.shape[-1]), xr.reshape(-1, xr.shape[-1])
        return g @ wr, g2.t() @ x2, (g2.sum(0) if ctx.has_b else None), None


def linear(x, p, name):
    if EMU['mode'] is not None and (EMU['prefixes'] is None or name.startswith(EMU['prefixes'])):
        return _EmuLinear.apply(x, p[name + '.weight'], p.get(name + '.bias'), EMU['mode'] == 'fp16')
    y = x @ p[name + '.weight'].t()
    if name + '.bias' in p:
        y = y + p[name + '.bias']
    return y


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def batch_norm1d(x, p, name, training, eps=1e-5, momentum=0.1, update_running=False):
    """nn.BatchNorm1d on [rows, C].  Training: batch mean / biased variance;
    running stats get the unbiased variance (torch semantics)."""
    if training:
        mean = x.mean(0)
        var = ((x - mean) ** 2).mean(0)
        if update_running:
            n = x.shape[0]
            with torch.no_grad():
                p[name + '.running_mean'].mul_(1 - momentum).add_(momentum * mean)
                p[name + '.running_var'].mul_(1 - momentum).add_(momentum * var * n / max(n - 1, 1))
                if name + '.num_batches_tracked' in p:
                    p[name + '.num_batches_tracked'] += 1
    else:
        mean, var = p[name + '.running_mean'], p[name + '.running_var']
    return (x - mean) / torch.sqrt(var + eps) * p[name + '.weight'] + p[name + '.bias']


def attention(q, k, v, mask=None, disjoint=False):
    """models/utils.py:11-44 (dropout omitted: parity runs use p=0 / eval)."""
    s = q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1])
    if mask is not None:
        s = s.masked_fill(mask == 0, float('-inf'))
    p = torch.softmax(s, dim=-1)
    if disjoint:
        # argmax over the QUERY dim (dim=2), one-hot mask times softmax (utils.py:26-33)
        win = p.argmax(dim=2, keepdim=True)
        p = p * torch.zeros_like(p).scatter_(2, win, 1.0)
    return p @ v, p


def sincos_table(seq_len, d_model, train_len=None):
    """models/utils.py:113-126.  NOTE the reference's exponent uses the channel
    index itself: even channel i -> sin(pos / 10000^(i/d)); odd channel i ->
    cos(pos / 10000^(i/d)).  float64, like the numpy original."""
    pos = np.arange(seq_len, dtype=np.float64) if train_len is None \
        else np.linspace(0, train_len - 1, num=seq_len)
    ch = np.arange(d_model, dtype=np.float64)
    ang = pos[:, None] / (10000.0 ** (ch[None, :] / d_model))
    tab = np.where((np.arange(d_model) % 2 == 0)[None, :], np.sin(ang), np.cos(ang))
    return torch.from_numpy(tab)


def positional_encode(x, train_len):
    """PositionalEncoder.forward (utils.py:136-145): table depends on S vs seq_len."""
    s, d = x.shape[-2], x.shape[-1]
    tab = sincos_table(s, d) if s == train_len else sincos_table(s, d, train_len)
    return x + tab.to(x.dtype)


def mha(x, mask, p, pre, heads):
    """MultiheadedAttention with Q=K=V=x (utils.py:75-108); mask [B,1,S] or None."""
    b, s, _ = x.shape
    q, k, v = linear(x, p, pre + 'linear_Q2d'), linear(x, p, pre + 'linear_K2d'), linear(x, p, pre + 'linear_V2d')
    d = q.shape[-1]
    dk = d // heads
    q, k, v = [t.view(b, s, heads, dk).transpose(1, 2) for t in (q, k, v)]
    o, _ = attention(q, k, v, None if mask is None else mask.unsqueeze(1))
    o = o.transpose(1, 2).reshape(b, s, d)
    return linear(o, p, pre + 'linear_d2Q')


def encoder(x, mask, p, pre, n_layers, heads, eps=1e-5):
    """Encoder of pre-LN EncoderLayers, no final LN (utils.py:196-242)."""
    for i in range(n_layers):
        lp = '%senc_layers.%d.' % (pre, i)
        h = layer_norm(x, p[lp + 'res_layer0.norm.weight'], p[lp + 'res_layer0.norm.bias'], eps)
        x = x + mha(h, mask, p, lp + 'self_att.', heads)
        h = layer_norm(x, p[lp + 'res_layer1.norm.weight'], p[lp + 'res_layer1.norm.bias'], eps)
        h = relu(linear(h, p, lp + 'feed_forward.fc1'))
        if EMU['mode'] == 'fp16' and (EMU['prefixes'] is None or (lp + 'feed_forward.fc1').startswith(EMU['prefixes'])):
            # the device keeps this activation (and takes the ReLU mask of the backward from it) as IEEE fp16 with subnormals flushed:
            # a positive pre-activation below 2^-14 is a zero there, value and mask (measured: tools/head_fp16_debug2.py)
            h = h * (h.detach() >= 2.0 ** -14).to(h.dtype)
        x = x + linear(h, p, lp + 'feed_forward.fc2')
    return x


# ----------------------------------------------------------------------------
# LSTP pooling
# ----------------------------------------------------------------------------
def lstp_cross_att(x, p, pre, cfg, dyn_in=None):
    """LSTPCrossAtt.forward (mvformer.py:352-414) on one clip.
    x [T, N, C] (keys == values); returns (out [T, nq, d_v], probs [T, nq, N])."""
    t = x.shape[0]
    k = linear(x, p, pre + 'linear_K2d')
    v = x if cfg.val_pass else linear(x, p, pre + 'linear_V2d')
    qs = []
    if cfg.nst > 0:
        qs_static = p[pre + 'Q_s'] + p[pre + 'Q_s_b']  # [1, nst, d]
    if cfg.nsdt > 0:
        if cfg.dyn_ctrl == 'first':
            dyn_in = dyn_in[0:1]
        elif cfg.dyn_ctrl == 'average':
            dyn_in = dyn_in.mean(0, keepdim=True)
        q_d = linear(dyn_in, p, pre + 'in2dynQ').view(dyn_in.shape[0], cfg.nsdt, -1)
    if cfg.nsdt == 0:
        q = qs_static
    elif cfg.nst == 0:
        q = q_d
    else:
        q = torch.cat([qs_static.expand(q_d.shape[0], -1, -1), q_d], 1)
    if cfg.ln_keys:
        k = k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12)  # F.normalize
    out, probs = attention(q.unsqueeze(1), k.unsqueeze(1), v.unsqueeze(1), None, cfg.disjoint)
    if out.shape[0] != t:
        out = out.expand(t, -1, -1, -1)
    return out[:, 0], probs[:, 0]


def token_pooling(feat, p, pre, cfg, n_clips, cls_emb=None):
    """LearnableTokenPooling.forward (mvformer.py:243-266).
    feat [F, N, C] token-major spatial features (the reference holds them as
    [F, C, h, w] and movedim's back; same numbers).  Returns [F, nq, d]."""
    f = feat.shape[0]
    x = feat.view(n_clips, f // n_clips, *feat.shape[1:])
    dyn = None if cfg.nsdt == 0 else cls_emb.view(n_clips, f // n_clips, -1)
    outs, probs = [], []
    for c in range(n_clips):
        o, pr = lstp_cross_att(x[c], p, pre + 'cross_att.', cfg, None if dyn is None else dyn[c])
        outs.append(o)
        probs.append(pr)
    return torch.cat(outs, 0), torch.cat(probs, 0)


# ----------------------------------------------------------------------------
# the head
# ----------------------------------------------------------------------------
def mvf_head(feat, masks, p, cfg, training=False, cls_emb=None, update_running=False, return_aux=False):
    """MultiEntityTransformerEmbModel.forward (mvformer.py:128-200).

    feat  [Bc, T, N, C]  spatial tokens of the tapped blocks, CLS dropped
          (== reference x[Bc,T,C,h,w] with (h,w) flattened and moved last)
    masks [Bc, 1, T] or None
    p     parameters keyed like the reference's `embed.` sub-module state dict
    returns embeddings [Bc, T, E]
    """
    bc, t, n, c = feat.shape
    if cfg.fwb:
        # FWBPooling.forward (mvformer.py:455-463): lin_conv(cls).reshape([F, -1, ntok]) is [F, spc, ntok]; the head
        # then moves the token axis forward (mvformer.py:140-142)
        tt = cfg.nst + cfg.nsdt
        pooled = linear(cls_emb, p, 'pooling.lin_conv').reshape(bc * t, -1, tt).transpose(1, 2)
        probs = None
    else:
        pooled, probs = token_pooling(feat.reshape(bc * t, n, c), p, 'pooling.', cfg, bc, cls_emb)  # [Bc*T, ntok, d]
    ntok = pooled.shape[1]
    x = pooled
    if cfg.one_hot == 'pool':
        eye = torch.eye(ntok, dtype=x.dtype).unsqueeze(0).expand(x.shape[0], -1, -1)
        x = torch.cat([x, eye], 2)
    x = x.reshape(bc * t * ntok, -1)
    i = 0
    while 'fc_layers.%d.weight' % (4 * i + 1) in p:  # [Dropout, Linear, BN, ReLU] blocks
        x = linear(x, p, 'fc_layers.%d' % (4 * i + 1))
        x = batch_norm1d(x, p, 'fc_layers.%d' % (4 * i + 2), training, cfg.bn_eps, cfg.bn_momentum, update_running)
        x = relu(x)
        i += 1
    x = linear(x, p, 'video_emb')
    d = x.shape[1]
    x = x.view(bc, t, ntok, d).transpose(1, 2).reshape(bc * ntok, t, d)
    x = positional_encode(x, cfg.train_len)
    x = x.view(bc, ntok, t, d)
    if cfg.one_hot == 'enc':
        eye = torch.eye(ntok, dtype=x.dtype).view(1, ntok, 1, ntok).expand(bc, -1, t, -1)
        x = torch.cat([x, eye], 3)
    x = x.reshape(bc, ntok * t, -1)
    if cfg.num_layers > 0:
        vm = None if masks is None else masks.unsqueeze(2).expand(bc, 1, ntok, t).reshape(bc, 1, ntok * t)
        x = encoder(x, vm, p, 'video_encoder.', cfg.num_layers, cfg.num_heads, cfg.ln_eps)
    x = x.view(bc, ntok, t, -1)
    if cfg.smart_final == 'max':
        x = x.max(dim=1)[0]
    elif cfg.smart_final == 'one':
        x = x[:, 0]
    elif cfg.smart_final == 'avg':
        x = x.mean(dim=1)
    else:  # 'lin'
        x = linear(x.transpose(1, 2).reshape(bc, t, -1), p, 'lin_final')
    x = linear(x.reshape(bc * t, -1), p, 'embedding_layer').view(bc, t, -1)
    if return_aux:
        return x, {'pooled': pooled, 'probs': probs}
    return x


def late_head(feat, masks, p, cfg, flatten='max_pool', training=False, update_running=False):
    """TransformerEmbModel.forward (late fusion, models/transformer.py:283-300).
    feat [Bc, T, N, C] tokens (N = h*w; N = 1 for LATE_TYPE 'cls'); AdaptiveMax/AvgPool2d(1) over the tokens, then the
    FC/BN/ReLU stack, video_emb, positional encoding, encoder, embedding_layer -- the MV-Former head with one entity and
    no learned pooling."""
    bc, t, n, c = feat.shape
    x = feat.max(dim=2)[0] if flatten == 'max_pool' else feat.mean(dim=2)
    x = x.reshape(bc * t, c)
    i = 0
    while 'fc_layers.%d.weight' % (4 * i + 1) in p:
        x = linear(x, p, 'fc_layers.%d' % (4 * i + 1))
        x = relu(batch_norm1d(x, p, 'fc_layers.%d' % (4 * i + 2), training, cfg.bn_eps, cfg.bn_momentum, update_running))
        i += 1
    x = linear(x, p, 'video_emb').view(bc, t, -1)
    x = positional_encode(x, cfg.train_len)
    if cfg.num_layers > 0:
        x = encoder(x, masks, p, 'video_encoder.', cfg.num_layers, cfg.num_heads, cfg.ln_eps)
    return linear(x.reshape(bc * t, -1), p, 'embedding_layer').view(bc, t, -1)


def mlp_head(x, p, pre='net.', training=False, update_running=False, eps=1e-5, momentum=0.1):
    """MLPHead.forward (resnet_c2d.py:112-126): Linear -> BN1d -> ReLU -> Linear on [B*T, E]."""
    b, l, c = x.shape
    h = linear(x.reshape(-1, c), p, pre + '0')
    h = relu(batch_norm1d(h, p, pre + '1', training, eps, momentum, update_running))
    return linear(h, p, pre + '3').view(b, l, c)


def l2_normalize(x, eps=1e-12):
    """F.normalize(x, dim=-1) (transformer.py:228,230)."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)
