#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the build container, where the
reference is mounted at /root/reference: it imports the reference's own Python
modules (unmodified, by path) and records their outputs for seeded inputs
(tests/golden/_cases.py) into small .npz fixtures next to this file.  The
fixtures are data (inputs are regenerated from seeds; expected outputs are
stored); nothing of the reference's source travels.

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz

Shims (harness side only, the reference is untouched):
  * `models`, `datasets`, `utils`, `algos` are registered as bare namespace
    packages pointing into CARL_MVF/ so `models/__init__.py` (torchvision) is skipped;
  * stub modules for torchvision.models / timm / easydict-free cfg (AttrDict);
  * torch.eye(device=-1) -> cpu: the reference's SMART_ONE_HOT path calls
    `torch.eye(n, device=x.get_device())`, which raises on CPU
    (CARL_MVF/models/mvformer.py:145).
"""
import os
import sys
import types
import importlib
import importlib.util
import logging as pylogging
import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/CARL_MVF'
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import _cases as C  # noqa: E402
from oracle import vit as ovit  # noqa: E402  (only for the seeded ViT weight generator)


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def ad(d):
    return AttrDict({k: ad(v) if isinstance(v, dict) else v for k, v in d.items()})


def install_shims():
    for name in ('models', 'datasets', 'utils', 'algos'):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, name)]
        sys.modules[name] = m
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tv.models = tvm
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = tvm
    lg = types.ModuleType('utils.logging')
    lg.get_logger = pylogging.getLogger
    sys.modules['utils.logging'] = lg
    sys.modules['utils'].logging = lg
    timm = types.ModuleType('timm')
    sys.modules['timm'] = timm
    real_eye = torch.eye

    def eye(*a, **kw):
        if kw.get('device', None) == -1:
            kw['device'] = 'cpu'
        return real_eye(*a, **kw)
    torch.eye = eye
    return timm


def ref_cfg(d, proj=True):
    em = dict(FC_DROPOUT_RATE=0.0, SMART_POOL_CHANNELS=d.spc, SMART_TOKENS=d.nst, SMART_DYNAMIC_TOKENS=d.nsdt,
              SMART_ONE_HOT=d.one_hot, CAPACITY_SCALAR=1, FC_LAYERS=[[w, True] for w in d.fc],
              EMBEDDING_SIZE=d.E, HIDDEN_SIZE=d.hidden, NUM_LAYERS=d.layers, NUM_HEADS=d.heads, D_FF=d.dff,
              SMART_FINAL=d.smart_final, SMART_FEATS=','.join(['3', '7', '11'][:d.n_taps]) if d.n_taps > 1 else '11',
              VAL_PASS=d.val_pass, SMART_DISJOINT=d.disjoint, SMART_LN_KEYS=d.ln_keys, DYNAMIC_CTRL=d.dyn_ctrl,
              FUSION_TYPE='smart')
    if d.fwb:
        em['FIXED_WIDTH_BASELINE'] = True
    return ad(dict(MODEL=dict(EMBEDDER_MODEL=em, BASE_MODEL=dict(OUT_CHANNEL=d.C), PROJECTION=proj,
                              PROJECTION_SIZE=d.proj, L2_NORMALIZE=True),
                   TRAIN=dict(NUM_FRAMES=d.train_len),
                   SCL=dict(POSITIVE_TYPE='gauss', NEGATIVE_TYPE='single_noself', SOFTMAX_TEMPERATURE=0.1,
                            LABEL_VARIENCE=10.0, POSITIVE_WINDOW=5),
                   TRAINING_ALGO='scl'))


def to_ref_x(feat):
    """[Bc,T,N,C] tokens-last -> the reference's [Bc,T,C,h,w]."""
    bc, t, n, c = feat.shape
    h = int(round(n ** 0.5))
    h, w = (h, h) if h * h == n else (n, 1)
    return feat.permute(0, 1, 3, 2).reshape(bc, t, c, h, w).contiguous()


# case name -> (Dims kwargs, Bc, T, N, pad, training)
SMALL = dict(C=96, n_taps=3, spc=24, nst=3, fc=(32, 32), hidden=32, dff=64, heads=4, layers=2, E=16, proj=16,
             train_len=8)
HEAD_CASES = {
    'base_train':   (dict(SMALL), 3, 8, 16, 3, True),
    'base_eval':    (dict(SMALL), 3, 8, 16, 3, False),
    'final_avg':    (dict(SMALL, smart_final='avg'), 2, 8, 16, 2, True),
    'final_max':    (dict(SMALL, smart_final='max'), 2, 8, 16, 2, True),
    'final_lin':    (dict(SMALL, smart_final='lin'), 2, 8, 16, 2, True),
    'onehot_none':  (dict(SMALL, one_hot='none'), 2, 8, 16, 0, True),
    'onehot_enc':   (dict(SMALL, one_hot='enc'), 2, 8, 16, 2, True),
    'nst1':         (dict(SMALL, nst=1), 2, 8, 16, 0, True),
    'nst6':         (dict(SMALL, nst=6), 2, 8, 16, 2, True),
    'interp_pe':    (dict(SMALL), 2, 12, 16, 0, False),          # S != TRAIN.NUM_FRAMES -> linspace positions
    'ln_keys':      (dict(SMALL, ln_keys=True), 2, 8, 16, 0, True),
    'val_pass':     (dict(SMALL, val_pass=True, fc=(32, 32)), 2, 8, 16, 0, True),
    'disjoint':     (dict(SMALL, disjoint=True), 2, 8, 16, 0, True),
    'dyn_separate': (dict(SMALL, nsdt=2, dyn_ctrl='separate'), 2, 8, 16, 0, True),
    'dyn_first':    (dict(SMALL, nsdt=2, dyn_ctrl='first'), 2, 8, 16, 0, True),
    'dyn_average':  (dict(SMALL, nsdt=2, dyn_ctrl='average'), 2, 8, 16, 0, True),
    'dyn_only':     (dict(SMALL, nst=0, nsdt=3, one_hot='none'), 2, 8, 16, 0, True),
    # BASELINE config #2 head at full width (inputs/params regenerated from seeds; outputs stored)
    'cfg2_full':    (dict(), 2, 32, 196, 5, True),
}


# cases added later keep their own seeds and file (the seeds of HEAD_CASES depend on the sorted case list)
HEAD_CASES_FWB = {
    'fwb_train': (dict(SMALL, fwb=True), 3, 8, 16, 3, True, 2001),
    'fwb_eval':  (dict(SMALL, fwb=True, nst=2, nsdt=1, one_hot='none', smart_final='avg'), 2, 8, 16, 0, False, 2002),
    # SMART_LN_KEYS together with dynamic (per-frame) queries (mvformer.py:365-400)
    'ln_keys_dyn':       (dict(SMALL, ln_keys=True, nsdt=2, dyn_ctrl='separate'), 2, 8, 16, 2, True, 2003),
    'ln_keys_dyn_first': (dict(SMALL, ln_keys=True, nst=0, nsdt=3, dyn_ctrl='first', one_hot='none', disjoint=True), 2, 8, 16, 0,
                          True, 2004),
}


def gen_head(mv, cases=None, fname='head.npz'):
    out = {}
    cases = cases or {k: v + (1000 + sorted(HEAD_CASES).index(k),) for k, v in HEAD_CASES.items()}
    for name, (kw, bc, t, n, pad, training, seed) in cases.items():
        d = C.Dims(**kw)
        params = C.head_params(d, seed)
        feat, masks, cls = C.head_inputs(d, bc, t, n, seed + 500, pad)
        mod = mv.MultiEntityTransformerEmbModel(ref_cfg(d))
        mod.load_state_dict(params, strict=True)
        mod.train(training)
        emb = mod(to_ref_x(feat), video_masks=masks, cls_emb=cls)
        rec = {'emb': emb.detach().numpy()}
        if not d.fwb:
            rec['attn'] = mod.pooling.cross_att.attn_matrix.numpy()[: t]  # last clip's [T, nq, N]
        g = torch.Generator().manual_seed(seed + 900)
        gout = torch.randn(emb.shape, generator=g)
        (emb * gout).sum().backward()
        full = d.C <= 256
        for k, p in mod.named_parameters():
            gk = p.grad if p.grad is not None else torch.zeros_like(p)
            rec['grad.' + k] = gk.numpy() if full else C.tensor_digest(gk)
        if training:
            for k, b in mod.named_buffers():
                if 'running' in k:
                    rec['buf.' + k] = b.numpy()
        for k, v in rec.items():
            out['%s/%s' % (name, k)] = v
        print('head', name, tuple(emb.shape), float(emb.abs().mean()))
    np.savez_compressed(os.path.join(HERE, fname), **out)


def gen_primitives(mu):
    out = {}
    g = torch.Generator().manual_seed(7)
    q, k, v = (torch.randn(2, 3, 5, 8, generator=g), torch.randn(2, 3, 7, 8, generator=g),
               torch.randn(2, 3, 7, 8, generator=g))
    mask = torch.ones(2, 1, 1, 7)
    mask[1, 0, 0, 5:] = 0
    for dis in (False, True):
        o, p = mu.attention(q, k, v, mask, None, True, disjoint=dis)
        out['attention/out_disjoint%d' % dis] = o.numpy()
        out['attention/p_disjoint%d' % dis] = p.numpy()
    for s, dm, tl in ((8, 256, None), (32, 256, None), (80, 256, None), (12, 32, 8), (50, 256, 32)):
        out['sincos/%d_%d_%s' % (s, dm, tl)] = mu.generate_sincos_embedding(s, dm, tl)[0].numpy()
    # Encoder at the real width, weights from the seeded generator (checks name/shape + arithmetic)
    for s in (24, 96):
        d = C.Dims()
        params = {k[len('video_encoder.'):]: v for k, v in C.head_params(d, 77).items() if k.startswith('video_encoder.')}
        enc = mu.Encoder(d.hidden, 0.0, d.heads, d.dff, d.layers)
        enc.load_state_dict(params, strict=True)
        x = torch.randn(2, s, d.hidden, generator=g, requires_grad=True)
        m = torch.ones(2, 1, s)
        m[1, 0, s - 5:] = 0
        y = enc(x, m)
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        out['encoder/S%d_x' % s] = x.detach().numpy()
        out['encoder/S%d_gy' % s] = gy.numpy()
        out['encoder/S%d_y' % s] = y.detach().numpy()
        out['encoder/S%d_gx' % s] = x.grad.numpy()
        out['encoder/S%d_gw_digest' % s] = np.stack([C.tensor_digest(p.grad) for p in enc.parameters()])
    np.savez_compressed(os.path.join(HERE, 'primitives.npz'), **out)
    print('primitives done')


def gen_mlp_head(rc):
    out = {}
    d = C.Dims()
    params = C.proj_params(d, 31)
    for training in (True, False):
        mod = rc.MLPHead(ref_cfg(d))
        mod.load_state_dict(params, strict=True)
        mod.train(training)
        g = torch.Generator().manual_seed(32)
        x = torch.randn(4, 16, d.E, generator=g, requires_grad=True)
        y = torch.nn.functional.normalize(mod(x), dim=-1)
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        tag = 'train' if training else 'eval'
        out['%s/y' % tag] = y.detach().numpy()
        out['%s/gx' % tag] = x.grad.numpy()
        for k, p in mod.named_parameters():
            out['%s/grad.%s' % (tag, k)] = p.grad.numpy()
        for k, b in mod.named_buffers():
            if 'running' in k:
                out['%s/buf.%s' % (tag, k)] = b.numpy()
    np.savez_compressed(os.path.join(HERE, 'mlp_head.npz'), **out)
    print('mlp_head done')


SCL_CASES = {
    # name: (B, T, E, pad, negative_type)
    'b1_t8_single_noself':  (1, 8, 128, 0, 'single_noself'),
    'b2_t8_single_noself':  (2, 8, 128, 3, 'single_noself'),
    'b4_t32_single_noself': (4, 32, 128, 12, 'single_noself'),
    'b4_t32_batch_noself':  (4, 32, 128, 12, 'batch_noself'),
    'b2_t8_single':         (2, 8, 128, 0, 'single'),
    'b2_t8_batch':          (2, 8, 128, 3, 'batch'),
    'b8_t32_batch_noself':  (8, 32, 128, 0, 'batch_noself'),     # == 2 ranks x B=4 concatenated (C9)
    'b8_t32_single_noself': (8, 32, 128, 12, 'single_noself'),
}


def gen_scl(scl_mod):
    out = {}
    for name, (b, t, e, pad, neg) in SCL_CASES.items():
        d = C.Dims(E=e)
        cfg = ref_cfg(d)
        cfg.SCL.NEGATIVE_TYPE = neg
        algo = scl_mod.SCL(cfg)
        seed = 2000 + sorted(SCL_CASES).index(name)
        embs, seq_lens, steps, masks = C.scl_inputs(b, t, e, seed, pad, seq_len=100 if t <= 32 else 300)
        embs.requires_grad_(True)
        loss = algo.compute_sequence_loss(embs, seq_lens, steps, masks)['loss']
        loss.backward()
        out[name + '/loss'] = loss.detach().numpy()
        out[name + '/gembs'] = embs.grad.numpy()
        print('scl', name, float(loss))
    np.savez_compressed(os.path.join(HERE, 'scl.npz'), **out)


# ---- stub ViT with the timm attribute surface the reference touches -------------
class _Attn(nn.Module):
    def __init__(s, dim, heads):
        super().__init__()
        s.heads = heads
        s.qkv = nn.Linear(dim, 3 * dim)
        s.proj = nn.Linear(dim, dim)

    def forward(s, x):
        b, n, c = x.shape
        qkv = s.qkv(x).reshape(b, n, 3, s.heads, c // s.heads).permute(2, 0, 3, 1, 4)
        o = torch.nn.functional.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        return s.proj(o.transpose(1, 2).reshape(b, n, c))


class _Mlp(nn.Module):
    def __init__(s, dim):
        super().__init__()
        s.fc1 = nn.Linear(dim, 4 * dim)
        s.act = nn.GELU()
        s.fc2 = nn.Linear(4 * dim, dim)

    def forward(s, x):
        return s.fc2(s.act(s.fc1(x)))


class _Block(nn.Module):
    def __init__(s, dim, heads):
        super().__init__()
        s.norm1 = nn.LayerNorm(dim, eps=1e-6)
        s.attn = _Attn(dim, heads)
        s.norm2 = nn.LayerNorm(dim, eps=1e-6)
        s.mlp = _Mlp(dim)

    def forward(s, x):
        x = x + s.attn(s.norm1(x))
        return x + s.mlp(s.norm2(x))


class _PatchEmbed(nn.Module):
    def __init__(s, dim, patch):
        super().__init__()
        s.proj = nn.Conv2d(3, dim, patch, patch)

    def forward(s, x):
        return s.proj(x).flatten(2).transpose(1, 2)


class StubViT(nn.Module):
    """Independent nn.Module ViT (F.scaled_dot_product_attention / nn.GELU / nn.LayerNorm)."""

    def __init__(s, dim, depth, heads, patch, img):
        super().__init__()
        s.patch_embed = _PatchEmbed(dim, patch)
        s.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        s.pos_embed = nn.Parameter(torch.zeros(1, (img // patch) ** 2 + 1, dim))
        s.patch_drop = nn.Identity()
        s.norm_pre = nn.Identity()
        s.blocks = nn.Sequential(*[_Block(dim, heads) for _ in range(depth)])
        s.norm = nn.LayerNorm(dim, eps=1e-6)
        s.fc_norm = nn.Identity()
        s.head_drop = nn.Identity()
        s.head = nn.Identity()
        s.global_pool = 'token'
        s.num_prefix_tokens = 1

    def _pos_embed(s, x):
        return torch.cat([s.cls_token.expand(x.shape[0], -1, -1), x], 1) + s.pos_embed

    def forward(s, x):
        x = s.norm(s.blocks(s._pos_embed(s.patch_embed(x))))
        return s.head(s.fc_norm(x[:, 0]))


GLUE = dict(name='vit_small_patch16_224.dino', dim=384, depth=12, heads=6, patch=16, img=32, vit_seed=5,
            head_seed=41, proj_seed=42, in_seed=43, bc=2, t=48, fpb=40)


def gen_glue(timm, tr):
    """TransformerModel.forward with a stub timm ViT: pins frame chunking (T=48 > FRAMES_PER_BATCH=40),
    CLS drop, NCHW-ification, projection + normalise (CARL_MVF/models/transformer.py:172-244)."""
    G = GLUE
    w = ovit.init_vit_weights(G['dim'], G['depth'], G['patch'], G['img'], seed=G['vit_seed'])

    def create_model(name, pretrained=True):
        m = StubViT(G['dim'], G['depth'], G['heads'], G['patch'], G['img'])
        m.load_state_dict(w, strict=True)
        return m
    timm.create_model = create_model
    d = C.Dims(C=G['dim'] * 3, n_taps=3, spc=24, fc=(32, 32), hidden=32, dff=64, heads=4, layers=2, E=16, proj=16,
               train_len=G['t'])
    cfg = ref_cfg(d)
    cfg.MODEL.BASE_MODEL = ad(dict(NETWORK='TIMM-' + G['name'], LAYER=12, FRAMES_PER_BATCH=G['fpb']))
    cfg.MODEL.EMBEDDER_MODEL.SMART_FEATS = '3,7,11'
    model = tr.TransformerModel(cfg, 0)
    sd = {'embed.' + k: v for k, v in C.head_params(d, G['head_seed']).items()}
    sd.update({'ssl_projection.' + k: v for k, v in C.proj_params(d, G['proj_seed']).items()})
    sd.update({'backbone.model.' + k: v for k, v in w.items()})
    model.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(G['in_seed'])
    x = torch.randn(G['bc'], G['t'], 3, G['img'], G['img'], generator=g)
    masks = torch.ones(G['bc'], 1, G['t'])
    masks[1, 0, 40:] = 0
    out = {}
    model.eval()
    out['eval_noproj'] = model(x, G['t'], video_masks=masks, project=False).detach().numpy()
    model.train()
    out['train_proj'] = model(x, G['t'], video_masks=masks, project=True).detach().numpy()
    np.savez_compressed(os.path.join(HERE, 'glue.npz'), **out)
    print('glue done')


GLUE2 = dict(name='vit_small_patch16_224.dino', dim=384, depth=12, heads=6, patch=16, img=32, vit_seed=6, back_seed=7,
             head_seed=44, proj_seed=45, in_seed=46, cls_seed=47, bc=2, t=8, layer=10)


def glue2_params(variant):
    """Seeded state dict of the GLUE2 cases, keyed like the reference's TransformerModel.state_dict().
    variant 'split' / 'warmup': LAYER=10, taps 10,11; the trainable back end (res_finetune.model.*) gets its OWN seeded
    weights (not the copies of the front's blocks it starts from) so that a mix-up of the two block sets would show.
    variant 'cls_res': LAYER=12, taps 3,7,11, plus cls_res_res.{weight,bias}."""
    G = GLUE2
    split = variant in ('split', 'warmup')
    d = C.Dims(C=G['dim'] * (2 if split else 3), n_taps=2 if split else 3, spc=24, fc=(32, 32), hidden=32, dff=64, heads=4,
               layers=2, E=16, proj=16, train_len=G['t'])
    w = ovit.init_vit_weights(G['dim'], G['depth'], G['patch'], G['img'], seed=G['vit_seed'])
    sd = {'embed.' + k: v for k, v in C.head_params(d, G['head_seed']).items()}
    sd.update({'ssl_projection.' + k: v for k, v in C.proj_params(d, G['proj_seed']).items()})
    sd.update({'backbone.model.' + k: v for k, v in w.items()})
    if split:
        # ViTFrontEnd registers the front blocks a second time under backbone.blocks.<i> (same tensors)
        for k, v in w.items():
            if k.startswith('blocks.') and int(k.split('.')[1]) < G['layer']:
                sd['backbone.' + k] = v
        wb = ovit.init_vit_weights(G['dim'], G['depth'], G['patch'], G['img'], seed=G['back_seed'])
        for k, v in wb.items():
            if k.startswith('blocks.') and int(k.split('.')[1]) >= G['layer']:
                _, j, rest = k.split('.', 2)
                sd['res_finetune.model.blocks.%d.%s' % (int(j) - G['layer'], rest)] = v
            elif k.startswith('norm.'):
                sd['res_finetune.model.' + k] = v
    else:
        g = torch.Generator().manual_seed(G['cls_seed'])
        sd['cls_res_res.weight'] = torch.randn(d.E, G['dim'], generator=g) / G['dim'] ** 0.5
        sd['cls_res_res.bias'] = torch.randn(d.E, generator=g) * 0.1
    return d, sd


def glue2_inputs():
    G = GLUE2
    g = torch.Generator().manual_seed(G['in_seed'])
    x = torch.randn(G['bc'], G['t'], 3, G['img'], G['img'], generator=g)
    masks = torch.ones(G['bc'], 1, G['t'])
    masks[1, 0, 6:] = 0
    return x, masks


GLUE2_GRADS = ('embed.embedding_layer.weight', 'embed.pooling.cross_att.Q_s', 'res_finetune.model.blocks.0.attn.qkv.weight',
               'res_finetune.model.blocks.1.mlp.fc2.weight', 'res_finetune.model.blocks.0.norm1.bias', 'cls_res_res.weight')


def compact(t):
    """Large tensors are stored as [L2 norm, sum, every 97th element ...] (fixtures stay small); small ones whole."""
    t = t.detach().reshape(-1)
    if t.numel() <= 4096:
        return t
    return torch.cat([t.double().norm().float().view(1), t.double().sum().float().view(1), t[::97]])


def gen_glue_split(timm, tr):
    """TransformerModel with (a) a partially frozen backbone, MODEL.BASE_MODEL.LAYER = 10: ViTFrontEnd / ViTBackEnd, the
    extract ids re-based into the back end (CARL_MVF/models/transformer.py:100-116,342-392); (b) BACKBONE_WARMUP's detach of
    the spatial features (mvformer.py:131-132, train.py:81-91); (c) MODEL.CLS_RES (transformer.py:235-242).  Stored: eval
    embeddings, train-mode projected embeddings, and the gradients of sum(train_proj * probe) w.r.t. a few parameters."""
    G = GLUE2
    out = {}
    real_cuda = nn.Module.cuda
    nn.Module.cuda = lambda self, device=None: self      # ViTBackEnd.__init__ calls .cuda(local_rank) (transformer.py:376-381)
    try:
        for variant in ('split', 'warmup', 'cls_res'):
            d, sd = glue2_params(variant)
            w = {k[len('backbone.model.'):]: v for k, v in sd.items() if k.startswith('backbone.model.')}

            def create_model(name, pretrained=True, w=w):
                m = StubViT(G['dim'], G['depth'], G['heads'], G['patch'], G['img'])
                m.load_state_dict(w, strict=True)
                return m
            timm.create_model = create_model
            cfg = ref_cfg(d)
            split = variant != 'cls_res'
            cfg.MODEL.BASE_MODEL = ad(dict(NETWORK='TIMM-' + G['name'], LAYER=G['layer'] if split else 12, FRAMES_PER_BATCH=40))
            cfg.MODEL.EMBEDDER_MODEL.SMART_FEATS = '10,11' if split else '3,7,11'
            if variant == 'cls_res':
                cfg.MODEL.CLS_RES = True
            model = tr.TransformerModel(cfg, 0)
            model.load_state_dict(sd, strict=True)
            out[variant + '_keys'] = np.array(sorted(model.state_dict().keys()))
            if variant == 'warmup':
                model.embed.set_warmup_status(True)
            x, masks = glue2_inputs()
            model.eval()
            out[variant + '_eval_noproj'] = model(x, G['t'], video_masks=masks, project=False).detach().numpy()
            model.train()
            y = model(x, G['t'], video_masks=masks, project=True)
            out[variant + '_train_proj'] = y.detach().numpy()
            probe = torch.randn(y.shape, generator=torch.Generator().manual_seed(99))
            named = dict(model.named_parameters())
            names = [n for n in GLUE2_GRADS if n in named and named[n].requires_grad]
            grads = torch.autograd.grad((y * probe).sum(), [named[n] for n in names], allow_unused=True)
            for n, gr in zip(names, grads):
                out[variant + '_grad:' + n] = compact(torch.zeros_like(named[n]) if gr is None else gr).numpy()
                out[variant + '_gradnone:' + n] = np.array(gr is None)
            print('glue_split', variant, float(y.abs().mean()), {n: (None if gr is None else float(gr.abs().max()))
                                                                   for n, gr in zip(names, grads)})
    finally:
        nn.Module.cuda = real_cuda
    np.savez_compressed(os.path.join(HERE, 'glue_split.npz'), **out)
    print('glue_split done')


TRAJ = dict(b=2, t=8, n=16, lr=1e-3, steps=3)


def gen_trajectory(mv, rc, scl_mod):
    """3 optimisation steps of (head + MLPHead + SCL + clip + Adam) on fixed backbone features:
    the call order of train.py:108-149 without AMP."""
    d = C.Dims(**SMALL)
    cfg = ref_cfg(d)
    embed = mv.MultiEntityTransformerEmbModel(cfg)
    embed.load_state_dict(C.head_params(d, 51), strict=True)
    proj = rc.MLPHead(cfg)
    proj.load_state_dict(C.proj_params(d, 52), strict=True)
    algo = scl_mod.SCL(cfg)
    params = list(embed.parameters()) + list(proj.parameters())
    opt = torch.optim.Adam(params, lr=TRAJ['lr'], betas=(0.9, 0.999), weight_decay=1e-5)
    b, t = TRAJ['b'], TRAJ['t']
    out = {}
    losses = []
    embed.train()
    proj.train()
    for it in range(TRAJ['steps']):
        feat, _, _ = C.head_inputs(d, b * 2, t, TRAJ['n'], 600 + it, 0)
        _, seq_lens, steps, masks = C.scl_inputs(b, t, d.E, 700 + it, pad=2 if it == 1 else 0, seq_len=30)
        opt.zero_grad()
        e = embed(to_ref_x(feat), video_masks=masks)
        e = torch.nn.functional.normalize(proj(e), dim=-1)
        loss = algo.compute_sequence_loss(e.view(b, 2, t, -1), seq_lens, steps, masks)['loss']
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        opt.step()
        losses.append(float(loss))
    out['losses'] = np.array(losses)
    for k, v in embed.state_dict().items():
        out['embed.' + k] = v.numpy()
    for k, v in proj.state_dict().items():
        out['ssl_projection.' + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, 'trajectory.npz'), **out)
    print('trajectory', losses)


def gen_late(tr):
    """TransformerEmbModel (late fusion, models/transformer.py:248-300) forward + parameter gradients."""
    out = {}
    for name, (flat, bc, t, hw, pad, training, seed) in C.LATE_CASES.items():
        L = C.LATE
        em = dict(FC_DROPOUT_RATE=0.0, CAPACITY_SCALAR=1, FC_LAYERS=[[w, True] for w in L['fc']], EMBEDDING_SIZE=L['E'],
                  HIDDEN_SIZE=L['hidden'], NUM_LAYERS=L['layers'], NUM_HEADS=L['heads'], D_FF=L['dff'], FLATTEN_METHOD=flat)
        cfg = ad(dict(MODEL=dict(EMBEDDER_MODEL=em, BASE_MODEL=dict(OUT_CHANNEL=L['C'])), TRAIN=dict(NUM_FRAMES=L['train_len'])))
        mod = tr.TransformerEmbModel(cfg)
        mod.load_state_dict(C.late_params(seed), strict=True)
        mod.train(training)
        x, masks = C.late_inputs(bc, t, hw, seed + 500, pad)
        emb = mod(x, video_masks=masks)
        rec = {'emb': emb.detach().numpy()}
        gout = torch.randn(emb.shape, generator=torch.Generator().manual_seed(seed + 900))
        (emb * gout).sum().backward()
        for k, p_ in mod.named_parameters():
            rec['grad.' + k] = (p_.grad if p_.grad is not None else torch.zeros_like(p_)).numpy()
        if training:
            for k, b_ in mod.named_buffers():
                if 'running' in k:
                    rec['buf.' + k] = b_.numpy()
        for k, v in rec.items():
            out['%s/%s' % (name, k)] = v
        print('late', name, tuple(emb.shape), float(emb.abs().mean()))
    np.savez_compressed(os.path.join(HERE, 'late.npz'), **out)


def gen_state_keys(mv, rc):
    """Checkpoint layout of BASELINE config #2 (models/__init__.py:17-27): the reference's state-dict keys and shapes for
    `embed.*` / `ssl_projection.*`, and the parameter order of its two optimizer groups (utils/optimizer.py:26-42:
    BatchNorm parameters first, then the rest, in named_modules order) -- the numbering of optimizer_state['state']."""
    import json
    d = C.Dims()
    cfg = ref_cfg(d)
    embed = mv.MultiEntityTransformerEmbModel(cfg)
    proj = rc.MLPHead(cfg)
    top = nn.Module()
    top.embed, top.ssl_projection = embed, proj
    keys = {k: list(v.shape) for k, v in top.state_dict().items()}
    bn, non_bn = [], []
    names = {id(p): n for n, p in top.named_parameters()}
    for n, m in top.named_modules():                       # optimizer.py:26-42 with MODEL.TRAIN_BASE == 'frozen'
        is_bn = isinstance(m, torch.nn.modules.batchnorm._NormBase)
        for p in m.parameters(recurse=False):
            (bn if is_bn else non_bn).append(names[id(p)])
    with open(os.path.join(HERE, 'state_keys.json'), 'w') as f:
        json.dump({'state_dict': keys, 'optimizer_groups': [bn, non_bn]}, f, indent=0)
    print('state keys', len(keys), len(bn), len(non_bn))


def gen_augment():
    """The reference's OWN augmentation functions (datasets/data_augment.py: _get_param_spatial_crop, random_resized_crop,
    flip, grayscale, resize, uniform_crop, color_normalization, and the RandomOp/AugmentOp/ComposeOp plumbing), imported
    with an empty `torchvision.transforms` module: ColorJitterOp / GaussianBlurOp, which construct torchvision objects, are
    NOT exercised (torchvision is absent; see oracle/augment.py)."""
    import random
    tv = sys.modules['torchvision']
    tv.transforms = types.ModuleType('torchvision.transforms')
    sys.modules['torchvision.transforms'] = tv.transforms
    da = importlib.import_module('datasets.data_augment')
    out = {}
    # crop-parameter draws: (height, width, seed) -> 6 consecutive (i, j, h, w)
    for n, (hh, ww, seed) in enumerate(C.AUG_CROP_CASES):
        random.seed(seed)
        out['crop%d' % n] = np.array([da._get_param_spatial_crop((0.8, 1.0), (3.0 / 4.0, 4.0 / 3.0), hh, ww) for _ in range(6)])
    out['crop_fallback'] = np.array([da._get_param_spatial_crop((0.8, 1.0), (3.0, 4.0), 40, 52),       # no draw fits
                                     da._get_param_spatial_crop((0.8, 1.0), (0.1, 0.2), 40, 52)])
    for n, (t, hh, ww, size, seed) in enumerate(C.AUG_CLIP_CASES):
        x = C.aug_clip(t, hh, ww, seed)
        random.seed(seed + 100)
        state = random.getstate()
        i, j, h, w = da._get_param_spatial_crop((0.8, 1.0), (3.0 / 4.0, 4.0 / 3.0), hh, ww)
        random.setstate(state)
        out['rrc%d' % n] = da.random_resized_crop(x, size, size).numpy()           # same draw as (i, j, h, w)
        out['rrc%d_param' % n] = np.array([i, j, h, w])
        out['flip%d' % n] = da.flip(x).numpy()
        out['gray%d' % n] = da.grayscale(x).numpy()
        out['resize%d' % n] = da.resize(x, size).numpy()
        out['norm%d' % n] = da.color_normalization(x).numpy()
        out['ucrop%d' % n] = da.uniform_crop(x, size).numpy()
        # the reference's own ops chained by its ComposeOp, draws from `random` in op order
        pipe = da.ComposeOp([da.AugmentOp(da.random_resized_crop, target_height=size, target_width=size),
                             da.RandomOp(da.flip, 0.5), da.RandomOp(da.grayscale, 0.2),
                             da.AugmentOp(da.resize, size=size),
                             da.AugmentOp(da.color_normalization, mean=[0.485, 0.456, 0.406], stddev=[0.229, 0.224, 0.225])])
        random.seed(seed + 200)
        out['pipe%d' % n] = np.stack([pipe(x).numpy() for _ in range(4)])
        # validation pre-processing exactly as the reference builds it (no torchvision involved)
        cfg = ad(dict(AUGMENTATION=dict(RANDOM_CROP=True), IMAGE_SIZE=size))
        out['val%d' % n] = da.create_data_augment(cfg, augment=False)(x).numpy()
    np.savez_compressed(os.path.join(HERE, 'augment.npz'), **out)
    print('augment done', len(out))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    timm = install_shims()
    mu = importlib.import_module('models.utils')
    mv = importlib.import_module('models.mvformer')
    rc = importlib.import_module('models.resnet_c2d')
    tr = importlib.import_module('models.transformer')
    spec = importlib.util.spec_from_file_location('ref_scl', os.path.join(REF, 'algos', 'scl.py'))
    scl_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scl_mod)
    which = sys.argv[1:] or ['primitives', 'head', 'mlp', 'scl', 'glue', 'glue_split', 'traj', 'keys', 'augment', 'fwb', 'late']
    if 'primitives' in which:
        gen_primitives(mu)
    if 'head' in which:
        gen_head(mv)
    if 'late' in which:
        gen_late(tr)
    if 'fwb' in which:
        gen_head(mv, HEAD_CASES_FWB, 'head_fwb.npz')
    if 'mlp' in which:
        gen_mlp_head(rc)
    if 'scl' in which:
        gen_scl(scl_mod)
    if 'glue' in which:
        gen_glue(timm, tr)
    if 'glue_split' in which:
        gen_glue_split(timm, tr)
    if 'traj' in which:
        gen_trajectory(mv, rc, scl_mod)
    if 'keys' in which:
        gen_state_keys(mv, rc)
    if 'augment' in which:
        gen_augment()


if __name__ == '__main__':
    main()
