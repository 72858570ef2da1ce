"""Diagnostic (GPU box): the head's GEMM launches of ONE training step at BASELINE configs[1] -- entry point, M, N, K, strides form --
and each launch's duration alone on the chip (HIP events around a replay of the same call, 50 repetitions).
    python tools/head_gemm_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib, ops  # noqa: E402
from video_rep_learning_amd.utils import presets  # noqa: E402
from video_rep_learning_amd.utils.optimizer import construct_optimizer  # noqa: E402
from video_rep_learning_amd.models import build_model  # noqa: E402
from video_rep_learning_amd.algos import get_algo  # noqa: E402
from video_rep_learning_amd.train import DataParallelModel  # noqa: E402
from video_rep_learning_amd.datasets import synthetic  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    cfg = presets.baseline_config_2('bf16')
    torch.manual_seed(1)
    model = build_model(cfg, 0).to(dev)
    wrapped = DataParallelModel(model)
    opt = construct_optimizer(wrapped, cfg)
    algo = get_algo(cfg)
    loader = synthetic.SyntheticClips(cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.NUM_FRAMES, cfg.IMAGE_SIZE, iters=1, seed=1234,
                                      device=dev, resident=True)
    (v0, v1), _l, seq_lens, steps, masks, _n = next(iter(loader))
    videos = torch.stack([v0, v1], dim=1)
    seq_lens, steps, masks = seq_lens.to(dev), steps.to(dev), masks.to(dev)
    model.train()

    def step():
        opt.zero_grad()
        loss = algo.compute_loss(wrapped, videos, seq_lens, steps, masks)['loss']
        ops.backward(loss)
        opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    calls = []
    real = _lib.call

    def spy(name, *args):
        if name in ('mvf_hgemm', 'mvf_hgemm_ex', 'mvf_hlinear_bwd'):
            calls.append((name, args))
        return real(name, *args)
    _lib.call = spy
    ops.call = spy if hasattr(ops, 'call') else None
    step()
    torch.cuda.synchronize()
    _lib.call = real
    if hasattr(ops, 'call'):
        ops.call = real
    tot = 0.0
    for name, args in calls:
        for _ in range(5):
            real(name, *args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            real(name, *args)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        tot += us
        if name == 'mvf_hlinear_bwd':
            M, N, K = args[11], args[12], args[13]
            desc = 'dx %s' % ('yes' if args[6] else 'no ')
            gf = 2.0 * M * N * K * (2 if args[6] else 1) / 1e9
        else:
            M, N, K = args[14], args[15], args[16]
            desc = 'A %s B %s' % ('k-major' if args[2] != 1 else 'k-contig', 'k-major' if args[4] != 1 else 'k-contig')
            gf = 2.0 * M * N * K / 1e9
        print('%-16s M %5d N %5d K %5d  %-24s %6.1f us  %5.2f GF  %6.1f TF/s' % (name, M, N, K, desc, us, gf, gf / us * 1e-3 * 1e3))
    print('%d launches, %.0f us back to back' % (len(calls), tot))


if __name__ == '__main__':
    main()
