"""One full-size bf16 training step per BASELINE.json config shape on ONE GPU (per-GPU batch of the config): does every
shape run through the product path, and at what rate?  (configs[2..4] are 8-GPU configs; this is their per-GPU work.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd.utils import presets  # noqa: E402
from video_rep_learning_amd.utils.optimizer import construct_optimizer  # noqa: E402
from video_rep_learning_amd.models import build_model  # noqa: E402
from video_rep_learning_amd.algos import get_algo  # noqa: E402
from video_rep_learning_amd.train import DataParallelModel  # noqa: E402

CASES = {
    'cfg1 penn_mvf as shipped: ViT-B/8, T=8, B=1': dict(network='TIMM-vit_base_patch8_224.dino', num_frames=8, batch_size=1),
    'penn_mvf.yml exactly as shipped: ViT-B/8, T=80, B=1': dict(network='TIMM-vit_base_patch8_224.dino', num_frames=80,
                                                                 batch_size=1),
    'cfg2 ViT-B/16, T=32, B=4': dict(network='TIMM-vit_base_patch16_224.dino', num_frames=32, batch_size=4),
    'cfg2 with LAYER=10: blocks 10-11 + norm trainable (bf16 GEMMs in all three directions)': dict(
        network='TIMM-vit_base_patch16_224.dino', num_frames=32, batch_size=4, SMART_FEATS='10,11', LAYER=10),
    'cfg3 fg99 head (6 entities, cap 6, E=256, taps 9,10,11, avg), T=32, B=4': dict(
        network='TIMM-vit_base_patch16_224.dino', num_frames=32, batch_size=4, SMART_TOKENS=6, CAPACITY_SCALAR=6,
        EMBEDDING_SIZE=256, SMART_FEATS='9,10,11', SMART_FINAL='avg'),
    'cfg4 T=64, B=4 (S = 192)': dict(network='TIMM-vit_base_patch16_224.dino', num_frames=64, batch_size=4),
    'cfg5 DINOv2 ViT-L/14 @336, T=32, B=4, bf16': dict(
        network='TIMM-vit_large_patch14_dinov2.lvd142m', num_frames=32, batch_size=4, image_size=336, SMART_FEATS='7,15,23',
        LAYER=24),
    'cfg5 DINOv2 ViT-L/14 @336, T=32, B=4, fp8 (MX-fp8 GEMM operands)': dict(
        network='TIMM-vit_large_patch14_dinov2.lvd142m', num_frames=32, batch_size=4, image_size=336, SMART_FEATS='7,15,23',
        LAYER=24, DTYPE='fp8'),
    'fg99_mvf.yml exactly as shipped: ViT-B/8, T=240, B=1, 6 entities (S = 1440), cap 6, E=256': dict(
        network='TIMM-vit_base_patch8_224.dino', num_frames=240, batch_size=1, SMART_TOKENS=6, CAPACITY_SCALAR=6,
        EMBEDDING_SIZE=256, SMART_FEATS='9,10,11', SMART_FINAL='avg'),
    'pouring_mvf.yml exactly as shipped: ViT-B/8, T=240, B=1, one tap (S = 720)': dict(
        network='TIMM-vit_base_patch8_224.dino', num_frames=240, batch_size=1, SMART_FEATS='11'),
    'cfg2 ViT-B/16, T=32, B=4, fp8 (NOT the headline dtype of this config: configs[1] is quoted at bf16)': dict(
        network='TIMM-vit_base_patch16_224.dino', num_frames=32, batch_size=4, DTYPE='fp8'),
}


# algorithmic TFLOP per step and GPU; None: not tabulated there
TFLOP_PER_STEP = {'cfg1 penn': 2.635, 'penn_mvf.yml exactly': 26.35, 'cfg2 ViT-B/16, T=32, B=4': 9.540, 'cfg3': 9.582, 'cfg4': 19.082,
                  'cfg5': 99.875}


def main():
    dev = torch.device('cuda')
    only = sys.argv[1:] or None
    for name, kw in CASES.items():
        if only and not any(o in name for o in only):
            continue
        kw = dict(kw)
        layer = kw.pop('LAYER', None)
        cfg = presets.make_cfg(compute_dtype=kw.pop('DTYPE', 'bf16'), **kw)
        if layer is not None:
            cfg.MODEL.BASE_MODEL.LAYER = layer        # frozen backbone: LAYER >= block count
        torch.manual_seed(1)
        model = build_model(cfg, 0).to(dev)
        wrapped = DataParallelModel(model)
        opt = construct_optimizer(wrapped, cfg)
        algo = get_algo(cfg)
        model.train()
        b, t, s = cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.NUM_FRAMES, cfg.IMAGE_SIZE
        g = torch.Generator().manual_seed(2)
        videos = torch.randn(b, 2, t, 3, s, s, generator=g).to(dev)
        seq_lens = torch.full((b, 2), 100, dtype=torch.long, device=dev)
        steps = torch.sort(torch.randint(0, 100, (b, 2, t), generator=g), dim=-1)[0].to(dev)
        masks = torch.ones(b, 2, t, device=dev)

        def step():
            wrapped.prefetch(videos)
            opt.zero_grad()
            loss = algo.compute_loss(wrapped, videos, seq_lens, steps, masks)['loss']
            loss.backward()
            opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)
            return loss
        wrapped.prefetch(videos)
        for _ in range(10):
            loss = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        tf = [v for k, v in TFLOP_PER_STEP.items() if name.startswith(k)]
        peak = 5000.0 if cfg.MI355X['COMPUTE_DTYPE'] == 'fp8' else 2500.0
        extra = ''
        if tf and 'LAYER=10' not in name:
            extra = '  %6.0f TFLOP/s algorithmic = %.3f of the dense %s peak' % (tf[0] / dt, tf[0] / dt / peak,
                                                                            'fp8' if peak > 3000 else 'bf16')
        print('%-92s %8.2f ms/step  %7.1f clips/s/GPU  head %s  loss %.4f%s' % (name, dt * 1e3, 2 * b / dt, model.head_dtype, loss.item(), extra),
              flush=True)
        # (the `step` closure holds the model: without dropping it the previous case's buffers live on while the next case allocates)
        del model, wrapped, opt, videos, algo, loss, step
        import gc
        gc.collect()
        torch.cuda.empty_cache()


main()
