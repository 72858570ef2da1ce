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


def test_patchify_matches_conv():
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    wt = torch.randn(8, 3, 16, 16, generator=torch.Generator().manual_seed(2))
    ref = torch.nn.functional.conv2d(x, wt, stride=16).flatten(2).transpose(1, 2)
    got = OV.patchify(x, 16) @ wt.reshape(8, -1).t()
    assert (ref - got).abs().max().item() < 1e-4
