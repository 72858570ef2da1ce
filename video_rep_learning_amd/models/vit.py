"""ViT backbone module whose forward is the HIP kernel chain (csrc/vit_fwd.hip).

It takes the place of the `timm.create_model(name, pretrained=True)` object the reference builds at
CARL_MVF/models/transformer.py:59 and keeps timm 0.9.2's parameter names (`cls_token`, `pos_embed`,
`patch_embed.proj.*`, `blocks.N.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}.*`, `norm.*`,
DINOv2: `blocks.N.ls{1,2}.gamma`) so timm / reference checkpoints load with `load_state_dict`.  The
nn.Linear / nn.LayerNorm / nn.Conv2d children are parameter HOLDERS only: their torch forward is never called.
There is no network here: weights are seeded trunc-normal(0.02) unless MODEL.BASE_MODEL.WEIGHTS points at a
timm-format state dict.
"""
import torch
import torch.nn as nn

from .. import ops

# name -> (embed_dim, depth, heads, patch, layerscale)   (reference name table: transformer.py:43-54)
VIT_ZOO = {
    'vit_small_patch16_224.dino': (384, 12, 6, 16, False),
    'vit_small_patch8_224.dino': (384, 12, 6, 8, False),
    'vit_base_patch16_224.dino': (768, 12, 12, 16, False),
    'vit_base_patch8_224.dino': (768, 12, 12, 8, False),
    'vit_small_patch14_dinov2.lvd142m': (384, 12, 6, 14, True),
    'vit_base_patch14_dinov2.lvd142m': (768, 12, 12, 14, True),
    'vit_large_patch14_dinov2.lvd142m': (1024, 24, 16, 14, True),
}
# Accepted by the reference's name table (transformer.py:53-55) but NOT built here: timm's DINOv2-giant uses a packed SwiGLU
# MLP (SwiGLUPacked: fc1 -> [x1 | x2], silu(x1) * x2, hidden 4096), not the GELU MLP of every other entry.  Running it through
# the GELU kernels would silently compute a different network, so the name is refused.
VIT_UNSUPPORTED = {
    'vit_giant_patch14_dinov2.lvd142m': 'its MLP is timm SwiGLUPacked (silu(x1) * x2), which the HIP backbone does not '
                                        'implement; use vit_{small,base,large}_patch14_dinov2.lvd142m',
}


class _Holder(nn.Module):
    pass


class _LayerScale(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))


class _Block(nn.Module):
    def __init__(self, dim, layerscale):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Holder()
        self.attn.qkv = nn.Linear(dim, 3 * dim)
        self.attn.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Holder()
        self.mlp.fc1 = nn.Linear(dim, 4 * dim)
        self.mlp.fc2 = nn.Linear(4 * dim, dim)
        if layerscale:
            self.ls1, self.ls2 = _LayerScale(dim), _LayerScale(dim)


class VisionTransformer(nn.Module):
    def __init__(self, embed_dim, depth, num_heads, patch_size, img_size=224, layerscale=False):
        super().__init__()
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.patch_size, self.img_size = patch_size, img_size
        self.num_prefix_tokens = 1
        self.global_pool = 'token'
        n = (img_size // patch_size) ** 2 + 1
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n, embed_dim))
        self.patch_embed = _Holder()
        self.patch_embed.proj = nn.Conv2d(3, embed_dim, patch_size, patch_size)
        self.blocks = nn.Sequential(*[_Block(embed_dim, layerscale) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self._packed = {}
        self._plist = None
        self._pack_epoch = 0
        self.reset_parameters()

    def reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.trunc_normal_(p, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, (nn.Linear, nn.Conv2d)) and m.bias is not None:
                nn.init.zeros_(m.bias)

    def _apply(self, fn, *a, **kw):  # .cuda()/.to() invalidates the packed copy
        self._packed = {}
        self._plist = None
        return super()._apply(fn, *a, **kw)

    def _weights_version(self):
        """Changes whenever a parameter is written in place (load_state_dict's copy_, an optimizer, p.mul_()) or re-homed.
        A checkpoint loaded through a PARENT module (checkpoint.restore, PRETRAINED_CHECKPOINT) never calls this module's
        load_state_dict -- torch recurses with _load_from_state_dict -- so the packed copy is keyed on the parameters
        themselves instead of on an overridden method."""
        if getattr(self, '_plist', None) is None:       # re-homing goes through _apply, which drops this list (15 us per call)
            self._plist = list(self.parameters())
        # data_ptr too: `p.data = ...` and a flat-buffer adoption move the storage without touching _version.  What neither
        # sees is a raw-pointer write into the SAME storage (a ctypes kernel, FusedAdam's flat-buffer step): whoever does that
        # to a packed parameter calls invalidate_packed().
        return tuple((p._version, p.data_ptr()) for p in self._plist) + (self._pack_epoch,)

    def invalidate_packed(self):
        """Forces the packed device copy to be rebuilt at the next forward (explicit hook for writers that bypass torch's
        version counters)."""
        self._pack_epoch += 1

    def _packed_for(self, key, depth, taps, dtype):
        ver = self._weights_version()
        hit = self._packed.get(key)
        if hit is None or hit[0] != ver:
            if hit is not None and torch.cuda.is_available():
                torch.cuda.synchronize()     # a forward on a side stream may still read the copy that is dropped here
            sd = {k: v for k, v in self.state_dict().items()}
            hit = (ver, ops.PackedViT(sd, depth, self.embed_dim, self.num_heads, self.patch_size, self.img_size, taps, dtype))
            self._packed[key] = hit
        return hit[1]

    def packed(self, taps, dtype):
        return self._packed_for((tuple(taps), str(dtype)), self.depth, taps, dtype)

    @torch.no_grad()
    def forward_taps(self, x, taps, dtype='bf16', frames_per_chunk=0):
        """x [F,3,H,W] -> (list of tapped block outputs [F*(N-1), D] (CLS dropped), cls [F, D])."""
        return ops.vit_forward(x, self.packed(taps, dtype), frames_per_chunk=frames_per_chunk)

    def forward(self, x):
        """timm's default forward: final-norm CLS embedding [F, D]."""
        return self.forward_taps(x, (), dtype='fp32')[1]

    @torch.no_grad()
    def forward_front(self, x, nb, dtype='bf16', frames_per_chunk=0):
        """x [F,3,H,W] -> fp32 residual stream [F, N, D] after the first nb (frozen) blocks."""
        pk = self._packed_for(('front', nb, str(dtype)), nb, (), dtype)
        return ops.vit_front(x, pk, frames_per_chunk=frames_per_chunk)


def block_forward(blk, x, heads, fast=False):
    """One TRAINABLE ViT block (timm Block.forward: x + ls1(attn(norm1(x))); x + ls2(mlp(norm2(x)))) on x [F, N, D] fp32,
    composed of the head's fp32 HIP ops, every one with a HIP backward: LayerNorm, GEMM (+ bias, + residual in the epilogue),
    flash-style attention on the fp32 matrix cores, exact-erf GELU.  This is the correctness-first form of SURVEY 8f row 3:
    the frozen blocks keep the bf16 persistent GEMM path."""
    ls = hasattr(blk, 'ls1')          # DINOv2: x + gamma * f(x) instead of the residual fused into the GEMM epilogue
    F, N, D = x.shape
    if fast and not ls and blk.attn.qkv.bias is not None and ops.vit_block_tc_supported(D, heads, blk.mlp.fc1.weight.shape[0]):
        # bf16 mode: the whole block as one autograd node with bf16 GEMM operands end to end (ops._ViTBlockTC)
        return ops.vit_block_tc(x, heads, blk.norm1.eps, blk.norm2.eps,
                                (blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.proj.weight,
                                 blk.attn.proj.bias, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
                                 blk.mlp.fc2.weight, blk.mlp.fc2.bias))
    # bf16 mode: the four GEMMs' forward and input gradient on the bf16 persistent kernel (ops._LinearTC); fp32 mode: exact
    lin = ops.linear_tc if fast else ops.linear
    x2 = x.reshape(F * N, D)
    h = ops.layer_norm(x2, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
    qkv = lin(h, blk.attn.qkv.weight, blk.attn.qkv.bias)
    if fast and D == heads * 64:       # bf16 mode: flash forward + bf16 MFMA backward (the reference's fp16 autocast)
        o = ops.vit_attention_bf16(qkv, F, N, heads)
    else:                              # parity mode: exact fp32 on the fp32 matrix cores
        o = ops.temporal_attention(qkv, None, F, N, heads)             # softmax(q k^T / sqrt(64)) v, no mask
    if ls:
        x2 = ops.layerscale_add(lin(o, blk.attn.proj.weight, blk.attn.proj.bias), blk.ls1.gamma, x2)
    else:
        x2 = lin(o, blk.attn.proj.weight, blk.attn.proj.bias, resid=x2)
    h = ops.layer_norm(x2, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
    h = ops.gelu(lin(h, blk.mlp.fc1.weight, blk.mlp.fc1.bias))
    if ls:
        x2 = ops.layerscale_add(lin(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias), blk.ls2.gamma, x2)
    else:
        x2 = lin(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias, resid=x2)
    return x2.view(F, N, D)


class ViTFrontEnd(nn.Module):
    """Frozen blocks [0, nb) of the backbone (models/transformer.py:342-361).  `self.blocks` aliases the same modules as
    `self.model.blocks[:nb]`, so the state dict carries both key sets like the reference's."""

    def __init__(self, model, nb):
        super().__init__()
        self.model = model
        self.nb = nb
        self.blocks = nn.Sequential(*[model.blocks[i] for i in range(min(nb, len(model.blocks)))])

    def forward(self, x, dtype='bf16', frames_per_chunk=0):
        return self.model.forward_front(x, self.nb, dtype=dtype, frames_per_chunk=frames_per_chunk)


class ViTBackEnd(nn.Module):
    """Trainable deep copies of blocks [nb, depth) and of the final norm (models/transformer.py:364-392).  forward returns
    (tapped block outputs with the CLS row dropped, final-norm CLS embedding) -- what FeatureExtractor(ViTBackEnd) yields in
    the reference (hooks on `blocks.<i - nb>` + the pooled output; fc_norm / head_drop / head are identities for these
    models)."""

    def __init__(self, model, nb):
        super().__init__()
        from copy import deepcopy
        self.global_pool = model.global_pool
        self.num_prefix_tokens = model.num_prefix_tokens
        self.num_heads = model.num_heads
        self.blocks = nn.Sequential(*[deepcopy(model.blocks[i]) for i in range(nb, len(model.blocks))])
        self.norm = deepcopy(model.norm)
        self.fc_norm, self.head_drop, self.head = nn.Identity(), nn.Identity(), nn.Identity()
        for p in self.parameters():
            p.requires_grad_(True)

    def forward(self, x, tap_ids, fast=False):
        F, N, D = x.shape
        taps = {}
        for i, blk in enumerate(self.blocks):
            x = block_forward(blk, x, self.num_heads, fast)
            if i in tap_ids:
                taps[i] = x[:, self.num_prefix_tokens:].reshape(F * (N - self.num_prefix_tokens), D)
        cls = ops.layer_norm(x[:, 0], self.norm.weight, self.norm.bias, self.norm.eps)
        return [taps[i] for i in tap_ids], cls


def resample_abs_pos_embed(posemb, new_tokens, num_prefix_tokens=1):
    """[1, P + g*g, D] -> [1, P + n*n, D]: the grid part bicubically resampled with antialiasing, prefix (CLS) tokens kept --
    what timm does (layers/pos_embed.py: resample_abs_pos_embed) when a checkpoint's image size differs from the model's
    (DINOv2 weights are stored for 518 px = 37 x 37 patches; IMAGE_SIZE 224 / 336 need 16 x 16 / 24 x 24).  One-time host work."""
    import math
    if posemb.shape[1] == new_tokens:
        return posemb
    old = int(math.sqrt(posemb.shape[1] - num_prefix_tokens))
    new = int(math.sqrt(new_tokens - num_prefix_tokens))
    assert old * old + num_prefix_tokens == posemb.shape[1] and new * new + num_prefix_tokens == new_tokens, 'square grids only'
    prefix, grid = posemb[:, :num_prefix_tokens], posemb[:, num_prefix_tokens:]
    dt = grid.dtype
    grid = grid.float().reshape(1, old, old, -1).permute(0, 3, 1, 2)
    grid = torch.nn.functional.interpolate(grid, size=(new, new), mode='bicubic', antialias=True)
    grid = grid.permute(0, 2, 3, 1).reshape(1, new * new, -1).to(dt)
    return torch.cat([prefix, grid], dim=1)


def create_model(name, pretrained=False, weights=None, img_size=224, seed=None):
    """Stand-in for timm.create_model for the names the reference accepts."""
    if name in VIT_UNSUPPORTED:
        raise NotImplementedError('TIMM model %s is not supported on the MI355X path: %s' % (name, VIT_UNSUPPORTED[name]))
    if name not in VIT_ZOO:
        raise ValueError('unknown/unsupported TIMM model: %s' % name)
    dim, depth, heads, patch, ls = VIT_ZOO[name]
    if seed is not None:
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(seed)
            m = VisionTransformer(dim, depth, heads, patch, img_size, ls)
    else:
        m = VisionTransformer(dim, depth, heads, patch, img_size, ls)
    if weights:
        sd = torch.load(weights, map_location='cpu')
        sd = dict(sd.get('model', sd.get('state_dict', sd)))
        if 'pos_embed' in sd and sd['pos_embed'].shape != m.pos_embed.shape:      # stored for another image size
            sd['pos_embed'] = resample_abs_pos_embed(sd['pos_embed'], m.pos_embed.shape[1], m.num_prefix_tokens)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        if missing:
            raise RuntimeError('backbone weights %s lack keys: %s' % (weights, missing[:5]))
    return m
