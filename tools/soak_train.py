"""Sanity soak (GPU box): a few hundred full training steps of BASELINE configs[1] (bf16 backbone) on four fixed synthetic
batches -- the loss must stay finite and fall (the head memorises the batches), no step may be skipped by the non-finite guard.
    python tools/soak_train.py [steps] [dtype]"""
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


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
    dev = torch.device('cuda')
    cfg = presets.baseline_config_2(compute_dtype=dtype)
    torch.manual_seed(cfg.RNG_SEED)
    model = build_model(cfg, 0).to(dev)
    wrapped = DataParallelModel(model)
    opt = construct_optimizer(wrapped, cfg)
    algo = get_algo(cfg)
    model.train()
    b, t, s = cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.NUM_FRAMES, cfg.IMAGE_SIZE
    g = torch.Generator().manual_seed(7)
    batches = []
    for _ in range(4):
        base = torch.randn(b, 1, t, 3, s, s, generator=g)
        videos = (base + 0.3 * torch.randn(b, 2, t, 3, s, s, generator=g)).to(dev)      # two noisy views of one clip
        st = torch.sort(torch.randint(0, 100, (b, 1, t), generator=g), dim=-1)[0].expand(b, 2, t).contiguous().to(dev)
        batches.append((videos, torch.full((b, 2), 100, dtype=torch.long, device=dev), st, torch.ones(b, 2, t, device=dev)))
    losses = []
    t0 = time.perf_counter()
    wrapped.prefetch(batches[0][0])
    for i in range(steps):
        cur = batches[i % 4]
        wrapped.prefetch(batches[(i + 1) % 4][0])
        opt.zero_grad()
        loss = algo.compute_loss(wrapped, *cur)['loss']
        loss.backward()
        opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)
        losses.append(loss.detach())
        if (i + 1) % 50 == 0:
            vals = torch.stack(losses[-50:]).float().cpu()
            print('steps %4d..%4d: loss mean %.4f min %.4f max %.4f' % (i - 48, i + 1, vals.mean(), vals.min(), vals.max()), flush=True)
    torch.cuda.synchronize()
    all_l = torch.stack(losses).float().cpu()
    skipped = opt.skipped_steps() if hasattr(opt, 'skipped_steps') else 0
    print('%d steps in %.1f s (%.2f ms/step); finite: %s; skipped steps: %s; first 20 mean %.4f -> last 20 mean %.4f' % (
        steps, time.perf_counter() - t0, (time.perf_counter() - t0) / steps * 1e3, bool(torch.isfinite(all_l).all()), skipped,
        all_l[:20].mean(), all_l[-20:].mean()))
    assert torch.isfinite(all_l).all() and int(skipped) == 0 and all_l[-20:].mean() < all_l[:20].mean()


if __name__ == '__main__':
    main()
