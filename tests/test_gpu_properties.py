"""Size-independent properties of the hot path at BASELINE configs[1]'s FULL size (ViT-B/16, 224 px, 256 frames, bf16 backbone,
fp32 head) -- checks that do not need the CPU oracle to finish at that size:

  * the frozen backbone treats frames independently: permuting the frames permutes the taps BITWISE (persistent GEMM tile walk,
    ticket scheduling, attention, LayerNorm fold and deferred residual never mix rows of different frames), and the split into
    lanes / chunks is bitwise invisible;
  * the head only sees valid frames: frames beyond a clip's `seq_len` (masked) do not influence the embeddings of valid frames;
  * the SCL loss is symmetric in the two views of every video;
  * one training step is repeatable bitwise when nothing random is left (dropout 0): no atomics-order dependence on the path
    the benchmark times (the LayerNorm dgamma / dbeta atomics of the head are the documented exception: 1e-6)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from video_rep_learning_amd import ops  # noqa: E402
from video_rep_learning_amd.utils import presets  # noqa: E402
from video_rep_learning_amd.models import build_model  # noqa: E402
from video_rep_learning_amd.algos import get_algo  # noqa: E402
from oracle import vit as OV  # noqa: E402  (weights only: the oracle computes nothing here)

DEV = 'cuda'
DIM, DEPTH, HEADS, PATCH, IMG, TAPS = 768, 12, 12, 16, 224, (3, 7, 11)


def _packed(dtype='bf16'):
    # 'fp8-fold': norm1 of blocks > 0 folded into the MX-fp8 qkv GEMM (the default from dim 1024 on, forced here at dim 768): the fc2
    # epilogue's MX-fp8 residual rows, scale dwords and row sums are laid out per CHUNK -- ragged chunks and lanes must not show
    w = OV.init_vit_weights(DIM, DEPTH, PATCH, IMG, seed=21)
    fold = 1 if dtype == 'fp8-fold' else None
    pk = ops.PackedViT({k: v.to(DEV) for k, v in w.items()}, DEPTH, DIM, HEADS, PATCH, IMG, TAPS, dtype.split('-')[0], ln_fold=fold)
    assert dtype != 'fp8-fold' or pk.ln_fold == 2
    return pk


@pytest.mark.parametrize('dtype', ['bf16', 'fp8', 'fp8-fold'])
def test_backbone_is_frame_permutation_equivariant_and_split_invariant_at_256_frames(dtype):
    F = 256
    np_ = (IMG // PATCH) ** 2
    pk = _packed(dtype)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(F, 3, IMG, IMG, generator=g).to(DEV)
    perm = torch.randperm(F, generator=g).to(DEV)
    taps, cls = ops.vit_forward(x, pk)                                  # the product's form: 2 lanes
    taps_p, cls_p = ops.vit_forward(x[perm].contiguous(), pk)
    for t, tp in zip(taps, taps_p):
        assert torch.equal(t.view(F, np_, DIM)[perm], tp.view(F, np_, DIM))
    assert torch.equal(cls[perm], cls_p)
    taps1, cls1 = ops.vit_forward(x, pk, lanes=1)                       # one lane, one chunk
    taps4, cls4 = ops.vit_forward(x, pk, lanes=1, frames_per_chunk=96)  # ragged chunks: 96 + 96 + 64 frames
    for a, b, c in zip(taps, taps1, taps4):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.equal(cls, cls1) and torch.equal(cls, cls4)
    assert all(torch.isfinite(t.float()).all() for t in taps)


def _batch(cfg, seed):
    g = torch.Generator().manual_seed(seed)
    b, t, s = cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.NUM_FRAMES, cfg.IMAGE_SIZE
    videos = torch.randn(b, 2, t, 3, s, s, generator=g).to(DEV)
    seq_lens = torch.full((b, 2), 100, dtype=torch.long, device=DEV)
    steps = torch.sort(torch.randint(0, 100, (b, 2, t), generator=g), dim=-1)[0].to(DEV)
    masks = torch.ones(b, 2, t, device=DEV)
    return videos, seq_lens, steps, masks


def test_masked_frames_do_not_reach_valid_embeddings_full_size():
    cfg = presets.make_cfg(network='TIMM-vit_base_patch16_224.dino', num_frames=32, batch_size=4, compute_dtype='bf16', dropout=0.0)
    torch.manual_seed(3)
    model = build_model(cfg, 0).to(DEV).eval()
    videos, _sl, _st, masks = _batch(cfg, 41)
    b, v, t = videos.shape[:3]
    L = 23                                                   # clips 0 and 5 are 23 frames long, the rest is padding
    masks = masks.clone()
    masks.view(b * v, t)[0, L:] = 0
    masks.view(b * v, t)[5, L:] = 0
    x = videos.view(b * v, t, *videos.shape[3:])
    x2 = x.clone()
    gen = torch.Generator().manual_seed(42)
    x2[0, L:] = torch.randn(t - L, *x.shape[2:], generator=gen).to(DEV) * 3.0        # other content in the padded frames
    x2[5, L:] = 0.0
    with torch.no_grad():
        e1 = model(x, t, video_masks=masks.view(b * v, 1, t))
        e2 = model(x2, t, video_masks=masks.view(b * v, 1, t))
    valid = masks.view(b * v, t).bool()
    assert torch.equal(e1[valid], e2[valid])                 # eval mode: BatchNorm uses running statistics, rows independent


def test_scl_loss_is_symmetric_in_the_two_views_full_size():
    cfg = presets.make_cfg(network='TIMM-vit_base_patch16_224.dino', num_frames=32, batch_size=4, compute_dtype='bf16', dropout=0.0)
    torch.manual_seed(4)
    model = build_model(cfg, 0).to(DEV).eval()               # eval: no batch statistics, the loss depends on the views only
    videos, seq_lens, steps, masks = _batch(cfg, 43)
    algo = get_algo(cfg)
    with torch.no_grad():
        l1 = algo.compute_loss(model, videos, seq_lens, steps, masks)['loss']
        sw = lambda z: z.flip(1).contiguous()                # noqa: E731  swap view 0 and view 1 of every video
        l2 = algo.compute_loss(model, sw(videos), sw(seq_lens), sw(steps), sw(masks))['loss']
    assert abs(l1.item() - l2.item()) <= 2e-6 * abs(l1.item()), (l1.item(), l2.item())


def test_training_step_is_repeatable_full_size():
    """Two identical steps from identical states (dropout 0): loss bitwise equal, every updated parameter equal except the
    head's LayerNorm gamma / beta, whose gradients are float atomics (documented in csrc/head_misc.hip)."""
    from video_rep_learning_amd.train import DataParallelModel
    from video_rep_learning_amd.utils.optimizer import construct_optimizer
    cfg = presets.make_cfg(network='TIMM-vit_base_patch16_224.dino', num_frames=32, batch_size=4, compute_dtype='bf16', dropout=0.0)
    batch = _batch(cfg, 44)
    outs = []
    for _ in range(2):
        torch.manual_seed(5)
        model = build_model(cfg, 0).to(DEV)
        wrapped = DataParallelModel(model)
        opt = construct_optimizer(wrapped, cfg)
        model.train()
        opt.zero_grad()
        loss = get_algo(cfg).compute_loss(wrapped, *batch)['loss']
        loss.backward()
        opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)
        torch.cuda.synchronize()
        outs.append((loss.item(), {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith('backbone')}))
    assert outs[0][0] == outs[1][0]
    for k, v in outs[0][1].items():
        if not v.dtype.is_floating_point:
            assert torch.equal(v, outs[1][1][k]), k
        elif '.norm.' in k or 'norm.weight' in k or 'norm.bias' in k:
            assert torch.allclose(v, outs[1][1][k], rtol=0, atol=2e-6), k        # Adam step of +-lr on a 1e-6-noise gradient...
        else:
            assert torch.equal(v, outs[1][1][k]), k


FULL_BATCHES = {
    # BASELINE configs[3] at its full per-GPU batch: 4 videos x 2 views x 64 frames = 512 frames, temporal sequence S = 192
    'configs[3] T=64 B=4 (512 frames)': dict(network='TIMM-vit_base_patch16_224.dino', num_frames=64, batch_size=4, compute_dtype='bf16'),
    # BASELINE configs[4] at its full per-GPU batch: DINOv2 ViT-L/14 @ 336 px (577 tokens), 4 videos x 2 views x 32 frames = 8 clips
    'configs[4] ViT-L/14 @ 336 B=4 T=32 (8 clips) bf16': dict(network='TIMM-vit_large_patch14_dinov2.lvd142m', num_frames=32, batch_size=4,
                                                              image_size=336, compute_dtype='bf16', SMART_FEATS='7,15,23', LAYER=24),
    'configs[4] ViT-L/14 @ 336 B=4 T=32 (8 clips) fp8': dict(network='TIMM-vit_large_patch14_dinov2.lvd142m', num_frames=32, batch_size=4,
                                                             image_size=336, compute_dtype='fp8', SMART_FEATS='7,15,23', LAYER=24),
}


@pytest.mark.parametrize('tag', list(FULL_BATCHES))
def test_full_per_gpu_batches_of_configs_3_and_4_step_finite_and_repeatable(tag):
    """The parity tests of configs[3] / configs[4] (tests/test_gpu_configs.py) run at sizes the CPU oracle finishes (2 videos / 2 clips);
    the FULL per-GPU batches run here through the size-independent checks: two training steps from identical states (dropout 0) give
    a finite loss, bitwise the same loss, the same updated head (LayerNorm gamma / beta to 2e-6: float atomics), and the loss is
    symmetric in the two views."""
    from video_rep_learning_amd.train import DataParallelModel
    from video_rep_learning_amd.utils.optimizer import construct_optimizer
    kw = dict(FULL_BATCHES[tag])
    layer = kw.pop('LAYER', None)
    cfg = presets.make_cfg(dropout=0.0, **kw)
    if layer is not None:
        cfg.MODEL.BASE_MODEL.LAYER = layer
    batch = _batch(cfg, 61)
    outs = []
    for _ in range(2):
        torch.manual_seed(9)
        model = build_model(cfg, 0).to(DEV)
        wrapped = DataParallelModel(model)
        opt = construct_optimizer(wrapped, cfg)
        model.train()
        opt.zero_grad()
        loss = get_algo(cfg).compute_loss(wrapped, *batch)['loss']
        loss.backward()
        opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)
        torch.cuda.synchronize()
        outs.append((loss.item(), {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith('backbone')}))
        if len(outs) == 2:
            model.eval()
            with torch.no_grad():
                sw = lambda z: z.flip(1).contiguous()            # noqa: E731
                l1 = get_algo(cfg).compute_loss(model, *batch)['loss'].item()
                l2 = get_algo(cfg).compute_loss(model, *[sw(z) for z in batch])['loss'].item()
            assert abs(l1 - l2) <= 2e-6 * abs(l1), (tag, l1, l2)
        del model, wrapped, opt
        torch.cuda.empty_cache()
    assert outs[0][0] == outs[0][0] and 0.0 < outs[0][0] < 1e2, (tag, outs[0][0])
    assert outs[0][0] == outs[1][0], (tag, outs[0][0], outs[1][0])
    for k, v in outs[0][1].items():
        if not v.dtype.is_floating_point:
            assert torch.equal(v, outs[1][1][k]), k
        elif '.norm.' in k or 'norm.weight' in k or 'norm.bias' in k:
            assert torch.allclose(v, outs[1][1][k], rtol=0, atol=2e-6), k
        else:
            assert torch.equal(v, outs[1][1][k]), k


def test_backbone_streams_are_one_set_per_process():
    """ops.backbone_stream: the lookahead stream and the lane streams belong to the process and the device, not to a model -- a second
    model with streams of its own would share hardware queues with the first one's (profiles/r05/order_probe.txt: its backbone forward
    12.5 instead of 10.5 ms).  Identity only (no timing): the same object for the same role, another one for another role."""
    from video_rep_learning_amd import ops
    a, b = ops.backbone_stream('side', 'cuda'), ops.backbone_stream('side', torch.device('cuda', torch.cuda.current_device()))
    assert a is b
    l1, l1b, l2 = ops.backbone_stream('lane1', 'cuda'), ops.backbone_stream('lane1', 'cuda'), ops.backbone_stream('lane2', 'cuda')
    assert l1 is l1b and l1 is not a and l2 is not l1
    assert a.cuda_stream != torch.cuda.default_stream().cuda_stream
