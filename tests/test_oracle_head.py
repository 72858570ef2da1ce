"""Oracle (CPU restatement) vs golden vectors recorded from the imported reference
modules (tests/golden/gen_golden.py).  Not a GPU test: this pins the checker."""
import numpy as np
import pytest
import torch

import _cases as C
import gen_golden as G
from oracle import head as OH
from oracle import scl as OS
from oracle import model as OM
from oracle import vit as OV


def hcfg(d):
    return OH.HeadCfg(nst=d.nst, nsdt=d.nsdt, spc=d.spc, one_hot=d.one_hot, smart_final=d.smart_final,
                      num_heads=d.heads, num_layers=d.layers, train_len=d.train_len, dyn_ctrl=d.dyn_ctrl,
                      ln_keys=d.ln_keys, val_pass=d.val_pass, disjoint=d.disjoint, n_taps=d.n_taps, fwb=d.fwb)


def close(a, b, rtol=1e-4, atol=1e-5):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, 'max err %.3e vs ref scale %.3e' % (err, ref)


def test_attention_primitive(golden):
    gp = golden('primitives')
    g = torch.Generator().manual_seed(7)
    q, k, v = (torch.randn(2, 3, 5, 8, generator=g), torch.randn(2, 3, 7, 8, generator=g),
               torch.randn(2, 3, 7, 8, generator=g))
    mask = torch.ones(2, 1, 1, 7)
    mask[1, 0, 0, 5:] = 0
    for dis in (False, True):
        o, p = OH.attention(q, k, v, mask, dis)
        close(o, gp['attention/out_disjoint%d' % dis])
        close(p, gp['attention/p_disjoint%d' % dis])


@pytest.mark.parametrize('s,dm,tl', [(8, 256, None), (32, 256, None), (80, 256, None), (12, 32, 8), (50, 256, 32)])
def test_sincos(golden, s, dm, tl):
    ref = golden('primitives')['sincos/%d_%d_%s' % (s, dm, tl)]
    np.testing.assert_allclose(OH.sincos_table(s, dm, tl).numpy(), ref, rtol=0, atol=1e-12)


@pytest.mark.parametrize('s', [24, 96])
def test_encoder(golden, s):
    gp = golden('primitives')
    d = C.Dims()
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in C.head_params(d, 77).items()
         if k.startswith('video_encoder.')}
    x = torch.tensor(gp['encoder/S%d_x' % s], requires_grad=True)
    m = torch.ones(2, 1, s)
    m[1, 0, s - 5:] = 0
    y = OH.encoder(x, m, p, 'video_encoder.', d.layers, d.heads)
    close(y, gp['encoder/S%d_y' % s])
    (y * torch.tensor(gp['encoder/S%d_gy' % s])).sum().backward()
    close(x.grad, gp['encoder/S%d_gx' % s], rtol=2e-4)


ALL_HEAD_CASES = {k: v + (1000 + sorted(G.HEAD_CASES).index(k), 'head') for k, v in G.HEAD_CASES.items()}
ALL_HEAD_CASES.update({k: v + ('head_fwb',) for k, v in G.HEAD_CASES_FWB.items()})     # FWBPooling ablation


@pytest.mark.parametrize('name', sorted(ALL_HEAD_CASES))
def test_head_forward_backward(golden, name):
    kw, bc, t, n, pad, training, seed, fname = ALL_HEAD_CASES[name]
    gh = golden(fname)
    d = C.Dims(**kw)
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and 'running' not in k else v.clone())
              for k, v in C.head_params(d, seed).items()}
    feat, masks, cls = C.head_inputs(d, bc, t, n, seed + 500, pad)
    emb, aux = OH.mvf_head(feat, masks, params, hcfg(d), training=training, cls_emb=cls, update_running=training,
                           return_aux=True)
    close(emb, gh[name + '/emb'], rtol=2e-4, atol=2e-5)
    if not d.fwb:
        close(aux['probs'][-t:], gh[name + '/attn'], rtol=2e-4, atol=1e-6)
    gout = torch.randn(emb.shape, generator=torch.Generator().manual_seed(seed + 900))
    (emb * gout).sum().backward()
    full = d.C <= 256
    for k, p in params.items():
        key = '%s/grad.%s' % (name, k)
        if key not in gh.files:
            continue
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        if full:       # atol: gradients that are exactly 0 in exact arithmetic (a bias feeding train-mode BN) are fp32 noise ~1e-5
            close(g, gh[key], rtol=1e-3, atol=2e-5)
        else:
            ref = gh[key]
            got = C.tensor_digest(g)
            # (grads that are exactly 0 in exact arithmetic -- e.g. a bias feeding train-mode BN -- are fp32 noise ~1e-4)
            assert abs(got[1] - ref[1]) <= 1e-3 * abs(ref[1]) + 5e-5, (k, got[1], ref[1])
    if training:
        for k in params:
            key = '%s/buf.%s' % (name, k)
            if key in gh.files:
                close(params[k], gh[key], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('name', sorted(C.LATE_CASES))
def test_late_fusion_head(golden, name):
    """oracle late_head vs the imported TransformerEmbModel (tests/golden/late.npz)."""
    gl = golden('late')
    flat, bc, t, hw, pad, training, seed = C.LATE_CASES[name]
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and 'running' not in k else v.clone())
              for k, v in C.late_params(seed).items()}
    x, masks = C.late_inputs(bc, t, hw, seed + 500, pad)
    feat = x.reshape(bc, t, C.LATE['C'], hw * hw).transpose(2, 3)
    cfg = OH.HeadCfg(num_heads=C.LATE['heads'], num_layers=C.LATE['layers'], train_len=C.LATE['train_len'])
    emb = OH.late_head(feat, masks, params, cfg, flatten=flat, training=training, update_running=training)
    close(emb, gl[name + '/emb'], rtol=2e-4, atol=2e-5)
    gout = torch.randn(emb.shape, generator=torch.Generator().manual_seed(seed + 900))
    (emb * gout).sum().backward()
    for k, p in params.items():
        key = '%s/grad.%s' % (name, k)
        if key in gl.files:
            close(p.grad if p.grad is not None else torch.zeros_like(p), gl[key], rtol=1e-3, atol=1e-5)
        key = '%s/buf.%s' % (name, k)
        if training and key in gl.files:
            close(params[k], gl[key], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('training', [True, False])
def test_mlp_head(golden, training):
    gm = golden('mlp_head')
    tag = 'train' if training else 'eval'
    d = C.Dims()
    p = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and 'running' not in k else v.clone())
         for k, v in C.proj_params(d, 31).items()}
    g = torch.Generator().manual_seed(32)
    x = torch.randn(4, 16, d.E, generator=g, requires_grad=True)
    y = OH.l2_normalize(OH.mlp_head(x, p, 'net.', training, update_running=training))
    close(y, gm[tag + '/y'])
    gy = torch.randn(y.shape, generator=g)
    (y * gy).sum().backward()
    close(x.grad, gm[tag + '/gx'], rtol=1e-3)
    for k in ('net.0.weight', 'net.1.weight', 'net.1.bias', 'net.3.weight', 'net.3.bias'):
        close(p[k].grad, gm['%s/grad.%s' % (tag, k)], rtol=1e-3, atol=1e-6)
    if training:
        close(p['net.1.running_var'], gm['train/buf.net.1.running_var'])


@pytest.mark.parametrize('name', sorted(G.SCL_CASES))
def test_scl(golden, name):
    gs = golden('scl')
    b, t, e, pad, neg = G.SCL_CASES[name]
    seed = 2000 + sorted(G.SCL_CASES).index(name)
    embs, seq_lens, steps, masks = C.scl_inputs(b, t, e, seed, pad, seq_len=100 if t <= 32 else 300)
    embs.requires_grad_(True)
    loss = OS.scl_loss(embs, seq_lens, steps, masks, negative_type=neg)
    close(loss, gs[name + '/loss'], rtol=1e-5)
    loss.backward()
    close(embs.grad, gs[name + '/gembs'], rtol=1e-4, atol=1e-7)


def test_scl_gather_equals_concat(golden):
    """F5/C9: with a 'single' negative type the W-rank gathered loss equals the mean of
    per-rank losses; with 'batch*' it does not (cross-video negatives matter)."""
    embs, seq_lens, steps, masks = C.scl_inputs(8, 32, 128, 2000 + sorted(G.SCL_CASES).index('b8_t32_single_noself'), 12)
    full = OS.scl_loss(embs, seq_lens, steps, masks, negative_type='single_noself')
    halves = [OS.scl_loss(embs[i:i + 4], seq_lens[i:i + 4], steps[i:i + 4], masks[2 * i:2 * i + 8],
                          negative_type='single_noself') for i in (0, 4)]
    # mean of per-rank means weighted by valid-frame counts == global mean
    w = [masks[0:8].sum(), masks[8:16].sum()]
    assert abs(float(full) - float((halves[0] * w[0] + halves[1] * w[1]) / (w[0] + w[1]))) < 1e-5


def _glue_params():
    Gd = G.GLUE
    d = C.Dims(C=Gd['dim'] * 3, n_taps=3, spc=24, fc=(32, 32), hidden=32, dff=64, heads=4, layers=2, E=16, proj=16,
               train_len=Gd['t'])
    w = OV.init_vit_weights(Gd['dim'], Gd['depth'], Gd['patch'], Gd['img'], seed=Gd['vit_seed'])
    sd = {'embed.' + k: v for k, v in C.head_params(d, Gd['head_seed']).items()}
    sd.update({'ssl_projection.' + k: v for k, v in C.proj_params(d, Gd['proj_seed']).items()})
    sd.update({'backbone.model.' + k: v for k, v in w.items()})
    return d, sd


def test_model_glue(golden):
    gg = golden('glue')
    Gd = G.GLUE
    d, sd = _glue_params()
    x = torch.randn(Gd['bc'], Gd['t'], 3, Gd['img'], Gd['img'], generator=torch.Generator().manual_seed(Gd['in_seed']))
    masks = torch.ones(Gd['bc'], 1, Gd['t'])
    masks[1, 0, 40:] = 0
    vit_cfg = dict(heads=Gd['heads'], patch=Gd['patch'], taps=(3, 7, 11))
    y = OM.model_forward(x, sd, vit_cfg, hcfg(d), masks, project=False, training=False)
    close(y, gg['eval_noproj'], rtol=1e-3, atol=1e-5)
    y = OM.model_forward(x, sd, vit_cfg, hcfg(d), masks, project=True, training=True)
    close(y, gg['train_proj'], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize('variant', ['split', 'warmup', 'cls_res'])
def test_model_glue_split_warmup_cls_res(golden, variant):
    """The oracle's restatement of the partially frozen backbone (ViTFrontEnd / ViTBackEnd, extract ids re-based:
    transformer.py:100-116,342-392), of BACKBONE_WARMUP's detach (mvformer.py:131-132) and of MODEL.CLS_RES
    (transformer.py:235-242) against the imported reference: outputs and the gradients of a probed sum."""
    gg = golden('glue_split')
    Gd = G.GLUE2
    d, sd = G.glue2_params(variant)
    x, masks = G.glue2_inputs()
    split = variant != 'cls_res'
    vit_cfg = dict(heads=Gd['heads'], patch=Gd['patch'], taps=(10, 11) if split else (3, 7, 11))
    if split:
        vit_cfg['layer'] = Gd['layer']
    if variant == 'warmup':
        vit_cfg['warmup'] = True
    if variant == 'cls_res':
        vit_cfg['cls_res'] = True
    y = OM.model_forward(x, sd, vit_cfg, hcfg(d), masks, project=False, training=False)
    close(y, gg[variant + '_eval_noproj'], rtol=1e-3, atol=1e-5)
    names = [k[len(variant) + 6:] for k in gg.files if k.startswith(variant + '_grad:')]
    assert names
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    p = dict(sd)
    p.update(leaves)
    y = OM.model_forward(x, p, vit_cfg, hcfg(d), masks, project=True, training=True)
    close(y, gg[variant + '_train_proj'], rtol=1e-3, atol=1e-5)
    probe = torch.randn(y.shape, generator=torch.Generator().manual_seed(99))
    grads = torch.autograd.grad((y * probe).sum(), [leaves[n] for n in names], allow_unused=True)
    for n, gr in zip(names, grads):
        none = bool(gg[variant + '_gradnone:' + n])
        assert (gr is None) == none, (n, none)        # warm-up: nothing reaches the trainable blocks
        if gr is not None:
            close(G.compact(gr), gg[variant + '_grad:' + n], rtol=2e-3, atol=1e-6)
    assert variant != 'warmup' or any(bool(gg['warmup_gradnone:' + n]) for n in names)


def test_trajectory(golden):
    """3 steps of head + MLPHead + SCL + clip + Adam (train.py:108-149 order) on fixed features."""
    gt = golden('trajectory')
    d = C.Dims(**G.SMALL)
    params = {'embed.' + k: v.clone() for k, v in C.head_params(d, 51).items()}
    params.update({'ssl_projection.' + k: v.clone() for k, v in C.proj_params(d, 52).items()})
    T = G.TRAJ
    b, t = T['b'], T['t']
    opt_state = {}
    losses = []
    for it in range(T['steps']):
        feat, _, _ = C.head_inputs(d, b * 2, t, T['n'], 600 + it, 0)
        _, seq_lens, steps, masks = C.scl_inputs(b, t, d.E, 700 + it, pad=2 if it == 1 else 0, seq_len=30)
        loss = OM.train_step_features(feat, seq_lens, steps, masks, params, opt_state, hcfg(d),
                                      dict(negative_type='single_noself'), lr=T['lr'])
        losses.append(float(loss))
    np.testing.assert_allclose(losses, gt['losses'], rtol=2e-4)
    # Parameters whose true gradient is identically zero (a bias that only shifts the input of a
    # train-mode BatchNorm, or a key bias under softmax) receive pure fp32 round-off as "gradient";
    # Adam normalises that noise to +-lr steps, so the reference itself is chaotic there.  They (and
    # the BN running means they shift) are only bounded, everything else must match tightly.
    null_grad = ('linear_V2d.bias', 'fc_layers.1.bias', 'fc_layers.5.bias', 'linear_K2d.bias',
                 'enc_layers.1.feed_forward.fc2.bias', 'embedding_layer.bias', 'net.0.bias', 'running_mean')
    for k in params:
        if params[k].dtype.is_floating_point:
            loose = any(k.endswith(n) for n in null_grad)
            close(params[k], gt[k], rtol=1e-4, atol=2.1 * T['steps'] * T['lr'] if loose else 2e-5)
