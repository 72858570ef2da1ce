"""Oracle: model glue + one full training step on CPU (TEST INFRASTRUCTURE ONLY).

Restates
  TransformerModel.forward   CARL_MVF/models/transformer.py:172-244  (timm + 'smart' fusion branch)
  SCL.compute_loss           CARL_MVF/algos/scl.py:28-50
  train.train iteration body CARL_MVF/train.py:108-149 (zero_grad -> loss -> backward ->
                             clip_grad_norm_(GRAD_CLIP) -> Adam step)
on top of oracle.vit / oracle.head / oracle.scl.  State is one flat dict keyed
with the reference's full state-dict names (`backbone.model.*`, `embed.*`,
`ssl_projection.*`).  Used as the checker in tests / smoke and as bench.py's
`cpu_baseline` ("port") leg.
"""
import torch

from . import vit as ovit
from . import head as ohead
from . import scl as oscl


def sub(params, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in params.items() if k.startswith(prefix)}


def backbone_features(frames, params, vit_cfg):
    """[F,3,H,W] -> (spatial [F, N, C_taps] with CLS dropped, cls_emb [F, D]).
    The reference chunks frames by FRAMES_PER_BATCH (transformer.py:180-214); per-frame
    independent, so one pass gives the same numbers."""
    w = sub(params, 'backbone.model.')
    layer = vit_cfg.get('layer', None)
    if layer is None:                      # fully frozen backbone (transformer.py:93-99)
        with torch.no_grad():
            feats, cls = ovit.vit_forward(frames, w, vit_cfg['heads'], vit_cfg['patch'], tuple(vit_cfg['taps']),
                                          emulate=vit_cfg.get('emulate', None))
        return (feats[:, 1:] if feats is not None else None), cls  # drop CLS (transformer.py:204)
    # partially frozen (transformer.py:100-116): ViTFrontEnd = blocks [0, layer) under no_grad; ViTBackEnd = trainable deep
    # copies of blocks [layer, depth) + norm, stored as res_finetune.model.blocks.<i - layer> / res_finetune.model.norm
    with torch.no_grad():
        _, x = ovit.vit_forward(frames, w, vit_cfg['heads'], vit_cfg['patch'], taps=(), last_block=layer)
    back = sub(params, 'res_finetune.model.')
    wb = {}
    for k, v in back.items():
        if k.startswith('blocks.'):
            _, j, rest = k.split('.', 2)
            wb['blocks.%d.%s' % (int(j) + layer, rest)] = v
        else:
            wb[k] = v
    feats, cls = ovit.vit_forward(None, wb, vit_cfg['heads'], vit_cfg['patch'], tuple(vit_cfg['taps']), first_block=layer,
                                  x_in=x)
    return (feats[:, 1:] if feats is not None else None), cls


def model_forward(videos, params, vit_cfg, head_cfg, video_masks=None, project=False, l2_normalize=True,
                  training=False, update_running=False, projection=True):
    """TransformerModel.forward: videos [Bc,T,3,H,W] -> [Bc,T,E]."""
    bc, t = videos.shape[:2]
    feat, cls = backbone_features(videos.reshape(bc * t, *videos.shape[2:]), params, vit_cfg)
    return forward_from_backbone(feat, cls, bc, t, params, vit_cfg, head_cfg, video_masks, project, l2_normalize, training,
                                 update_running, projection)


def forward_from_backbone(feat, cls, bc, t, params, vit_cfg, head_cfg, video_masks=None, project=False, l2_normalize=True,
                          training=False, update_running=False, projection=True):
    """Everything of TransformerModel.forward behind the backbone (transformer.py:218-244), on its outputs
    feat [Bc*T, N, C] / cls [Bc*T, D]: lets a test run the (frozen, expensive) backbone once for several head passes."""
    feat = feat.reshape(bc, t, *feat.shape[1:]) if feat is not None else None
    if vit_cfg.get('warmup', False) and feat is not None:              # BACKBONE_WARMUP epochs (mvformer.py:131-132):
        feat = feat.detach()                                           # spatial features detached, cls_emb is not
    late = vit_cfg.get('late', None)        # None: 'smart' fusion; ('cls' | 'spatial', FLATTEN_METHOD): late fusion
    if late is None:
        x = ohead.mvf_head(feat, video_masks, sub(params, 'embed.'), head_cfg, training=training,
                           cls_emb=cls, update_running=update_running)
    else:
        f_in = cls.reshape(bc, t, 1, -1) if late[0] == 'cls' else feat      # transformer.py:192-196 / :197-207
        x = ohead.late_head(f_in, video_masks, sub(params, 'embed.'), head_cfg, flatten=late[1], training=training,
                            update_running=update_running)
    if projection and project:                                         # transformer.py:226-228
        x = ohead.mlp_head(x, sub(params, 'ssl_projection.'), 'net.', training, update_running)
        x = ohead.l2_normalize(x)
    elif l2_normalize:                                                 # :229-230
        x = ohead.l2_normalize(x)
    if vit_cfg.get('cls_res', False):                                  # MODEL.CLS_RES (transformer.py:235-242)
        w = sub(params, 'cls_res_res.')
        cr = (cls @ w['weight'].t() + w['bias']).reshape(x.shape[0], x.shape[1], -1)
        if l2_normalize:
            cr = ohead.l2_normalize(cr)
        x = x + cr
        if l2_normalize:
            x = ohead.l2_normalize(x)
    return x


def loss_from_backbone(feat, cls, seq_lens, chosen_steps, video_masks, params, vit_cfg, head_cfg, scl_cfg, training=True,
                       update_running=False):
    """SCL.compute_loss behind the backbone: feat [B*2*T, N, C], masks [B,2,T] -> scalar loss."""
    b, v, t = video_masks.shape
    masks = video_masks.reshape(b * v, 1, t)
    embs = forward_from_backbone(feat, cls, b * v, t, params, vit_cfg, head_cfg, masks, project=True, training=training,
                                 update_running=update_running)
    return oscl.scl_loss(embs.reshape(b, v, t, -1), seq_lens.reshape(b, v), chosen_steps, masks, **scl_cfg)


def compute_loss(videos, seq_lens, chosen_steps, video_masks, params, vit_cfg, head_cfg, scl_cfg,
                 training=True, update_running=False):
    """SCL.compute_loss: videos [B,2,T,3,H,W] -> scalar loss."""
    b, v, t = videos.shape[:3]
    masks = video_masks.reshape(b * v, 1, t)
    embs = model_forward(videos.reshape(b * v, t, *videos.shape[3:]), params, vit_cfg, head_cfg, masks,
                         project=True, training=training, update_running=update_running)
    return oscl.scl_loss(embs.reshape(b, v, t, -1), seq_lens.reshape(b, v), chosen_steps, masks, **scl_cfg)


def trainable_names(params):
    return [k for k, v in params.items() if not k.startswith('backbone.') and v.dtype.is_floating_point
            and 'running_' not in k]


def _clip_adam(loss_fn, params, opt_state, lr, betas, weight_decay, grad_clip, adam_eps):
    names = trainable_names(params)
    leaves = {k: params[k].detach().requires_grad_(True) for k in names}
    p = dict(params)
    p.update(leaves)
    loss = loss_fn(p)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    grads = [torch.zeros_like(leaves[k]) if g is None else g for k, g in zip(names, grads)]
    if grad_clip > 0:                                                  # clip_grad_norm_ (train.py:147-148)
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
        coef = torch.clamp(grad_clip / (total + 1e-6), max=1.0)
        grads = [g * coef for g in grads]
    opt_state['step'] = opt_state.get('step', 0) + 1
    st = opt_state['step']
    with torch.no_grad():
        for k, g in zip(names, grads):
            g = g + weight_decay * params[k]
            m = opt_state.setdefault('m.' + k, torch.zeros_like(g))
            v = opt_state.setdefault('v.' + k, torch.zeros_like(g))
            m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
            v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
            denom = (v.sqrt() / (1 - betas[1] ** st) ** 0.5).add_(adam_eps)
            params[k].addcdiv_(m, denom, value=-lr / (1 - betas[0] ** st))
    return loss.detach()


def train_step(batch, params, opt_state, vit_cfg, head_cfg, scl_cfg, lr=1e-4, betas=(0.9, 0.999),
               weight_decay=1e-5, grad_clip=10.0, adam_eps=1e-8):
    """One iteration of train.train (no AMP): returns the loss; params/opt_state updated in place.
    Adam with L2 weight decay == torch.optim.Adam(weight_decay=...) (utils/optimizer.py:60-66)."""
    return _clip_adam(lambda p: compute_loss(*batch, p, vit_cfg, head_cfg, scl_cfg, training=True,
                                             update_running=True),
                      params, opt_state, lr, betas, weight_decay, grad_clip, adam_eps)


def loss_from_features(feat, seq_lens, chosen_steps, video_masks, params, head_cfg, scl_cfg, training=True,
                       update_running=False):
    """Head + projection + SCL on precomputed backbone features feat [B*2, T, N, C]."""
    bc, t = feat.shape[:2]
    masks = video_masks.reshape(bc, 1, t)
    x = ohead.mvf_head(feat, masks, sub(params, 'embed.'), head_cfg, training=training,
                       update_running=update_running)
    x = ohead.l2_normalize(ohead.mlp_head(x, sub(params, 'ssl_projection.'), 'net.', training, update_running))
    return oscl.scl_loss(x.reshape(bc // 2, 2, t, -1), seq_lens, chosen_steps, masks, **scl_cfg)


def train_step_features(feat, seq_lens, chosen_steps, video_masks, params, opt_state, head_cfg, scl_cfg,
                        lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-5, grad_clip=10.0, adam_eps=1e-8):
    return _clip_adam(lambda p: loss_from_features(feat, seq_lens, chosen_steps, video_masks, p, head_cfg, scl_cfg,
                                                   True, True),
                      params, opt_state, lr, betas, weight_decay, grad_clip, adam_eps)
