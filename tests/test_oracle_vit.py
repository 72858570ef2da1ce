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
