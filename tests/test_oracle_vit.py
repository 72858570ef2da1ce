"""Oracle ViT (restated timm 0.9.2 VisionTransformer semantics) cross-checked against an
independent implementation: HuggingFace transformers.ViTModel with the same seeded weights.
The reference has no vector at this boundary (timm is un-vendored): 'parity unpinned by the
reference', so this is the strongest available anchor."""
import pytest
import torch

from oracle import vit as OV


def _to_hf(w, depth):
    sd = {'embeddings.cls_token': w['cls_token'], 'embeddings.position_embeddings': w['pos_embed'],
          'embeddings.patch_embeddings.projection.weight': w['patch_embed.proj.weight'],
          'embeddings.patch_embeddings.projection.bias': w['patch_embed.proj.bias'],
          'layernorm.weight': w['norm.weight'], 'layernorm.bias': w['norm.bias']}
    d = w['norm.weight'].shape[0]
    for i in range(depth):
        p, q = 'blocks.%d.' % i, 'layers.%d.' % i
        qkv_w, qkv_b = w[p + 'attn.qkv.weight'], w[p + 'attn.qkv.bias']
        for j, nm in enumerate(('q_proj', 'k_proj', 'v_proj')):
            sd[q + 'attention.%s.weight' % nm] = qkv_w[j * d:(j + 1) * d]
            sd[q + 'attention.%s.bias' % nm] = qkv_b[j * d:(j + 1) * d]
        sd[q + 'attention.o_proj.weight'] = w[p + 'attn.proj.weight']
        sd[q + 'attention.o_proj.bias'] = w[p + 'attn.proj.bias']
        sd[q + 'layernorm_before.weight'] = w[p + 'norm1.weight']
        sd[q + 'layernorm_before.bias'] = w[p + 'norm1.bias']
        sd[q + 'layernorm_after.weight'] = w[p + 'norm2.weight']
        sd[q + 'layernorm_after.bias'] = w[p + 'norm2.bias']
        for nm in ('fc1', 'fc2'):
            sd[q + 'mlp.%s.weight' % nm] = w[p + 'mlp.%s.weight' % nm]
            sd[q + 'mlp.%s.bias' % nm] = w[p + 'mlp.%s.bias' % nm]
    return sd


@pytest.mark.parametrize('dim,depth,heads,patch,img', [(768, 12, 12, 16, 224), (384, 12, 6, 8, 32)])
def test_vit_vs_hf(dim, depth, heads, patch, img):
    transformers = pytest.importorskip('transformers')
    cfg = transformers.ViTConfig(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads,
                                 intermediate_size=4 * dim, image_size=img, patch_size=patch,
                                 layer_norm_eps=1e-6, hidden_act='gelu', attn_implementation='eager')
    hf = transformers.ViTModel(cfg, add_pooling_layer=False).eval()
    w = OV.init_vit_weights(dim, depth, patch, img, seed=3)
    # scale the weights up so that 12 layers actually move the residual stream
    w = {k: (v * 3.0 if ('qkv.weight' in k or 'fc' in k or 'proj.weight' in k) else v) for k, v in w.items()}
    missing, unexpected = hf.load_state_dict(_to_hf(w, depth), strict=False)
    assert not unexpected and all('pooler' in m for m in missing), (missing, unexpected)
    x = torch.randn(2, 3, img, img, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        out = hf(pixel_values=x, output_hidden_states=True)
        feats, cls = OV.vit_forward(x, w, heads, patch, taps=(3, 7, 11))
    for j, blk in enumerate((3, 7, 11)):
        ref = out.hidden_states[blk + 1]          # hidden_states[0] = embeddings
        got = feats[:, :, j * dim:(j + 1) * dim]
        err = (got - ref).abs().max().item()
        assert err <= 2e-4 * ref.abs().max().item() + 1e-5, (blk, err)
    ref = out.last_hidden_state[:, 0]
    assert (cls - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-5


def test_dinov2_layerscale_vs_hf():
    """The DINOv2 family (LayerScale, patch 14: BASELINE configs[4]) against transformers.Dinov2Model with the same seeded
    weights: hidden_states[i + 1] = output of block i (pre final norm), last_hidden_state[:, 0] = final-norm CLS."""
    transformers = pytest.importorskip('transformers')
    dim, depth, heads, patch, img = 384, 12, 6, 14, 56
    cfg = transformers.Dinov2Config(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads, mlp_ratio=4,
                                    image_size=img, patch_size=patch, layer_norm_eps=1e-6, hidden_act='gelu',
                                    layerscale_value=1.0, use_swiglu_ffn=False, attn_implementation='eager')
    hf = transformers.Dinov2Model(cfg).eval()
    w = OV.init_vit_weights(dim, depth, patch, img, seed=5, layerscale=True)
    w = {k: (v * 3.0 if ('qkv.weight' in k or 'fc' in k or 'proj.weight' in k) else v) for k, v in w.items()}
    sd = {'embeddings.cls_token': w['cls_token'], 'embeddings.position_embeddings': w['pos_embed'],
          'embeddings.mask_token': torch.zeros(1, dim),
          'embeddings.patch_embeddings.projection.weight': w['patch_embed.proj.weight'],
          'embeddings.patch_embeddings.projection.bias': w['patch_embed.proj.bias'],
          'layernorm.weight': w['norm.weight'], 'layernorm.bias': w['norm.bias']}
    for i in range(depth):
        p, q = 'blocks.%d.' % i, 'encoder.layer.%d.' % i
        for j, nm in enumerate(('query', 'key', 'value')):
            sd[q + 'attention.attention.%s.weight' % nm] = w[p + 'attn.qkv.weight'][j * dim:(j + 1) * dim]
            sd[q + 'attention.attention.%s.bias' % nm] = w[p + 'attn.qkv.bias'][j * dim:(j + 1) * dim]
        sd[q + 'attention.output.dense.weight'], sd[q + 'attention.output.dense.bias'] = w[p + 'attn.proj.weight'], w[p + 'attn.proj.bias']
        sd[q + 'norm1.weight'], sd[q + 'norm1.bias'] = w[p + 'norm1.weight'], w[p + 'norm1.bias']
        sd[q + 'norm2.weight'], sd[q + 'norm2.bias'] = w[p + 'norm2.weight'], w[p + 'norm2.bias']
        sd[q + 'layer_scale1.lambda1'], sd[q + 'layer_scale2.lambda1'] = w[p + 'ls1.gamma'], w[p + 'ls2.gamma']
        for nm in ('fc1', 'fc2'):
            sd[q + 'mlp.%s.weight' % nm], sd[q + 'mlp.%s.bias' % nm] = w[p + 'mlp.%s.weight' % nm], w[p + 'mlp.%s.bias' % nm]
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    x = torch.randn(2, 3, img, img, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        out = hf(pixel_values=x, output_hidden_states=True)
        feats, cls = OV.vit_forward(x, w, heads, patch, taps=(3, 7, 11))
    for j, blk in enumerate((3, 7, 11)):
        ref = out.hidden_states[blk + 1]
        got = feats[:, :, j * dim:(j + 1) * dim]
        err = (got - ref).abs().max().item()
        assert err <= 2e-4 * ref.abs().max().item() + 1e-5, (blk, err)
    ref = out.last_hidden_state[:, 0]
    assert (cls - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-5


def test_patchify_matches_conv():
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    wt = torch.randn(8, 3, 16, 16, generator=torch.Generator().manual_seed(2))
    ref = torch.nn.functional.conv2d(x, wt, stride=16).flatten(2).transpose(1, 2)
    got = OV.patchify(x, 16) @ wt.reshape(8, -1).t()
    assert (ref - got).abs().max().item() < 1e-4


# ---- the emulating restatements against the plain one: with the roundings switched off, every folded form must BE the plain block ----
@pytest.mark.parametrize('emulate', ['bf16', 'bf16_fold12', 'bf16_nofold', 'fp8', 'fp8_nofold'])
def test_emulating_block_without_roundings_is_the_plain_block(emulate, monkeypatch):
    """oracle/vit.py restates the product's reduced-precision data flow -- LayerNorm folded into the consuming GEMM
    (rstd * (x W'^T - mean * c) + d), the deferred attention-branch residual, LayerScale folded into proj -- around bf16 / MX-fp8
    rounding points.  With those roundings replaced by the identity the restatement must reproduce the plain pre-LN block
    (timm Block.forward) to fp32 rounding: the algebra of every fold, checked on the CPU without a device."""
    dim, depth, heads, patch, img, F = 256, 3, 4, 16, 64, 2
    w = OV.init_vit_weights(dim, depth, patch, img, seed=5, layerscale=True)
    g = torch.Generator().manual_seed(6)
    for k in list(w):                   # non-trivial LayerNorm affines and LayerScales: the folds must carry them
        if k.endswith('norm1.weight') or k.endswith('norm2.weight') or k.endswith('.gamma'):
            w[k] = 1.0 + 0.3 * torch.randn(w[k].shape, generator=g)
        elif k.endswith('norm1.bias') or k.endswith('norm2.bias'):
            w[k] = 0.2 * torch.randn(w[k].shape, generator=g)
    x = torch.randn(F, 3, img, img, generator=g)
    with torch.no_grad():
        ref, cls = OV.vit_forward(x, w, heads, patch, (0, 2))
        monkeypatch.setattr(OV, 'bf16_round', lambda t: t)
        monkeypatch.setattr(OV, '_bf16_round_true', lambda t: t)
        monkeypatch.setattr(OV, 'mx_quant', lambda t: t)
        got, gcls = OV.vit_forward(x, w, heads, patch, (0, 2), emulate=emulate)
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 2e-5
    assert ((gcls - cls).abs().max() / cls.abs().max()).item() < 2e-5


def test_mx_quant_is_idempotent_and_block_local():
    """mx_quant (the MX-fp8 quantise-dequantise the fp8 emulation applies): a second pass changes nothing, and a block's result depends on
    that block alone (the scale is per 32 consecutive elements)."""
    g = torch.Generator().manual_seed(7)
    t = torch.randn(50, 256, generator=g) * torch.exp(2.0 * torch.randn(50, 1, generator=g))
    q = OV.mx_quant(t)
    assert torch.equal(OV.mx_quant(q), q)
    t2 = t.clone()
    t2[:, 32:64] *= 1000.0
    q2 = OV.mx_quant(t2)
    assert torch.equal(q2[:, :32], q[:, :32]) and torch.equal(q2[:, 64:], q[:, 64:])
    assert ((q - t).abs() <= 0.0625 * t.reshape(50, 8, 32).abs().amax(-1, keepdim=True).expand(50, 8, 32).reshape(50, 256) + 1e-30).all()
