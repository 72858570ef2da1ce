"""Seeded inputs / parameters shared by the golden generator (gen_golden.py, runs
in the build container against the imported reference) and by the parity tests
(which regenerate the same tensors instead of storing them).  torch CPU RNG with
an explicit Generator is deterministic for a fixed torch build, and both sides
run in the same image.

Parameter names and shapes follow the reference's state dict:
  embed.*           MultiEntityTransformerEmbModel  CARL_MVF/models/mvformer.py:15-116
  ssl_projection.*  MLPHead                         CARL_MVF/models/resnet_c2d.py:112-126
(gen_golden.py load_state_dict()s them strictly into the reference modules, which
pins names and shapes.)
"""
import math
import torch


class Dims:
    """Architecture hyper-parameters of one head case."""

    def __init__(self, **kw):
        self.C = 2304          # cfg.MODEL.BASE_MODEL.OUT_CHANNEL (after tap multiplication)
        self.n_taps = 3
        self.spc = 384
        self.nst = 3
        self.nsdt = 0
        self.fc = (512, 512)   # FC_LAYERS widths * CAPACITY_SCALAR
        self.hidden = 256
        self.dff = 1024
        self.heads = 8
        self.layers = 3
        self.E = 128
        self.one_hot = 'pool'
        self.smart_final = 'one'
        self.val_pass = False
        self.ln_keys = False
        self.disjoint = False
        self.dyn_ctrl = 'separate'
        self.fwb = False       # FIXED_WIDTH_BASELINE (FWBPooling, mvformer.py:421-463)
        self.train_len = 32
        self.proj = 128        # cfg.MODEL.PROJECTION_SIZE
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)

    @property
    def ntok(self):
        return self.nst + self.nsdt


def _u(g, shape, bound):
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1).mul_(bound).float()


def _n(g, shape, std, mean=0.0):
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * std + mean).float()


def _linear(p, g, name, fin, fout, gain=1.0):
    b = gain / math.sqrt(fin)
    p[name + '.weight'] = _u(g, (fout, fin), b)
    p[name + '.bias'] = _u(g, (fout,), b)


def _bn(p, g, name, ch):
    p[name + '.weight'] = _n(g, (ch,), 0.1, 1.0)
    p[name + '.bias'] = _n(g, (ch,), 0.1)
    p[name + '.running_mean'] = _n(g, (ch,), 0.1)
    p[name + '.running_var'] = (1.0 + 0.2 * torch.rand(ch, generator=g, dtype=torch.float64)).float()
    p[name + '.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)


def head_params(d, seed):
    """`embed.`-relative parameter dict."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    ca = 'pooling.cross_att.'
    if d.fwb:
        _linear(p, g, 'pooling.lin_conv', d.C // d.n_taps, d.spc * d.ntok, gain=2.0)
    if d.nst > 0 and not d.fwb:
        p[ca + 'Q_s'] = _u(g, (1, d.nst, d.spc), 4.0 / math.sqrt(d.spc))
        p[ca + 'Q_s_b'] = _u(g, (d.spc,), 1.0 / math.sqrt(d.spc))
    if not d.fwb:
        _linear(p, g, ca + 'linear_K2d', d.C, d.spc, gain=4.0)
    if not d.val_pass and not d.fwb:
        _linear(p, g, ca + 'linear_V2d', d.C, d.spc)
    if d.nsdt > 0 and not d.fwb:
        _linear(p, g, ca + 'in2dynQ', d.C // d.n_taps, d.spc * d.nsdt, gain=4.0)
    cin = d.C if d.val_pass else d.spc
    if d.one_hot == 'pool':
        cin += d.ntok
    for i, ch in enumerate(d.fc):
        _linear(p, g, 'fc_layers.%d' % (4 * i + 1), cin, ch)
        _bn(p, g, 'fc_layers.%d' % (4 * i + 2), ch)
        cin = ch
    hid_pe = d.hidden - d.nst if d.one_hot == 'enc' else d.hidden
    _linear(p, g, 'video_emb', cin, hid_pe)
    for i in range(d.layers):
        lp = 'video_encoder.enc_layers.%d.' % i
        for r in ('res_layer0', 'res_layer1'):
            p[lp + r + '.norm.weight'] = _n(g, (d.hidden,), 0.1, 1.0)
            p[lp + r + '.norm.bias'] = _n(g, (d.hidden,), 0.1)
        for nm in ('linear_Q2d', 'linear_K2d', 'linear_V2d', 'linear_d2Q'):
            _linear(p, g, lp + 'self_att.' + nm, d.hidden, d.hidden, gain=2.0)
        _linear(p, g, lp + 'feed_forward.fc1', d.hidden, d.dff)
        _linear(p, g, lp + 'feed_forward.fc2', d.dff, d.hidden)
    _linear(p, g, 'embedding_layer', d.hidden, d.E)
    if d.smart_final == 'lin':
        _linear(p, g, 'lin_final', d.ntok * d.hidden, d.hidden)
    return p


def proj_params(d, seed):
    """`ssl_projection.`-relative parameter dict (MLPHead: hidden = PROJECTION_SIZE)."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    _linear(p, g, 'net.0', d.E, d.proj)
    _bn(p, g, 'net.1', d.proj)
    _linear(p, g, 'net.3', d.proj, d.E)
    return p


def head_inputs(d, bc, t, n, seed, pad=0):
    """feat [Bc,T,N,C] ~ N(0,1); masks [Bc,1,T] (last clip gets `pad` padded frames);
    cls [Bc*T, C/n_taps]."""
    g = torch.Generator().manual_seed(seed)
    feat = torch.randn(bc, t, n, d.C, generator=g)
    cls = torch.randn(bc * t, d.C // d.n_taps, generator=g)
    masks = torch.ones(bc, 1, t)
    if pad:
        masks[-1, 0, t - pad:] = 0
    return feat, masks, cls


def scl_inputs(b, t, e, seed, pad=0, seq_len=100):
    """embs [B,2,T,E] L2-normalised; seq_lens [B,2]; steps [B,2,T] sorted; masks [B*2,1,T].
    `pad`: video 0 is short (seq_len t-pad < T): the last `pad` frames are padding with
    steps clamped to L-1 (datasets/penn_action.py:180-197)."""
    g = torch.Generator().manual_seed(seed)
    embs = torch.randn(b, 2, t, e, generator=g)
    embs = embs / embs.norm(dim=-1, keepdim=True)
    seq_lens = torch.full((b, 2), seq_len, dtype=torch.long)
    steps = torch.sort(torch.randint(0, seq_len, (b, 2, t), generator=g), dim=-1)[0]
    masks = torch.ones(b * 2, 1, t)
    if pad:
        L = t - pad
        seq_lens[0, :] = L
        st = torch.arange(t).clamp(max=L - 1)
        steps[0, 0] = st
        steps[0, 1] = st
        masks[0, 0, L:] = 0
        masks[1, 0, L:] = 0
    return embs, seq_lens, steps, masks


def tensor_digest(x):
    """Size-independent summary for full-size cases: sum, L2, 16 strided samples."""
    x = x.detach().double().reshape(-1)
    idx = torch.linspace(0, x.numel() - 1, 16).long()
    return torch.cat([x.sum().view(1), x.norm().view(1), x[idx]]).numpy()


# ---------------------------------------------------------------------------------------------------------------------
# view augmentation (datasets/data_augment.py)
# ---------------------------------------------------------------------------------------------------------------------
AUG_CROP_CASES = [(360, 480, 3), (480, 270, 4), (40, 52, 5), (224, 224, 6)]          # (height, width, seed)
AUG_CLIP_CASES = [(3, 40, 52, 16, 11), (2, 33, 21, 16, 12), (2, 24, 24, 24, 13)]     # (T, H, W, IMAGE_SIZE, seed)


def aug_clip(t, h, w, seed):
    """A clip as the dataset hands it over (penn_action.py:110-111): uint8 frames / 255 -> [T, 3, H, W] in [0, 1]."""
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (t, 3, h, w), generator=g).float() / 255.0


# ---------------------------------------------------------------------------------------------------------------------
# late fusion (TransformerEmbModel, models/transformer.py:248-300)
# ---------------------------------------------------------------------------------------------------------------------
LATE = dict(C=96, fc=(32, 32), hidden=32, dff=64, heads=4, layers=2, E=16, train_len=8)
# name -> (FLATTEN_METHOD, Bc, T, h = w, padded frames of the last clip, training, seed)
LATE_CASES = {
    'late_max_train': ('max_pool', 3, 8, 4, 3, True, 3001),
    'late_avg_eval':  ('avg_pool', 2, 8, 4, 0, False, 3002),
    'late_cls_train': ('max_pool', 2, 8, 1, 2, True, 3003),       # LATE_TYPE 'cls': a 1 x 1 "feature map"
    'late_interp_pe': ('avg_pool', 2, 12, 2, 0, False, 3004),     # S != TRAIN.NUM_FRAMES
}


def late_params(seed):
    """State dict of the reference's TransformerEmbModel at LATE: the MV-Former head's names without `pooling.*`, first
    FC layer fed by all C channels."""
    d = Dims(one_hot='none', val_pass=True, nst=1, n_taps=1, **LATE)
    return {k: v for k, v in head_params(d, seed).items() if not k.startswith('pooling.')}


def late_inputs(bc, t, hw, seed, pad=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(bc, t, LATE['C'], hw, hw, generator=g)
    masks = torch.ones(bc, 1, t)
    if pad:
        masks[-1, 0, t - pad:] = 0
    return x, masks
