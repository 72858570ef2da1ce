"""Diagnostic (GPU box): step period of the first steps after a cold start (the driver's bench call times steps 6..25): HIP events at
the end of every step's optimizer, no warm-up beyond the very first step.   python tools/cold_start_probe.py [--steps 80] [--idle 3]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import ops  # noqa: E402
from video_rep_learning_amd.utils import presets  # noqa: E402
from video_rep_learning_amd.utils.optimizer import construct_optimizer  # noqa: E402
from video_rep_learning_amd.models import build_model  # noqa: E402
from video_rep_learning_amd.algos import get_algo  # noqa: E402
from video_rep_learning_amd.train import DataParallelModel  # noqa: E402
from video_rep_learning_amd.datasets import synthetic  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--steps', type=int, default=80)
    p.add_argument('--idle', type=float, default=3.0, help='seconds of idle GPU before the run (as after a model build)')
    a = p.parse_args()
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
        wrapped.prefetch(videos)
        opt.zero_grad()
        loss = algo.compute_loss(wrapped, videos, seq_lens, steps, masks)['loss']
        ops.backward(loss)
        opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)
    wrapped.prefetch(videos)
    step()                       # library load, first-call attributes, allocator growth
    torch.cuda.synchronize()
    for rep in range(2):
        time.sleep(a.idle)
        ev = []
        for _ in range(a.steps):
            step()
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(dev))
            ev.append(e)
        torch.cuda.synchronize()
        per = [ev[i].elapsed_time(ev[i + 1]) for i in range(len(ev) - 1)]
        print('after %.0f s idle: step periods (ms): %s' % (a.idle, ' '.join('%.2f' % v for v in per)))
        print('   steps 6..25 mean %.3f   steps 20..69 mean %.3f   last 20 mean %.3f' %
              (sum(per[5:25]) / 20, sum(per[19:69]) / 50 if len(per) >= 69 else float('nan'), sum(per[-20:]) / 20))


if __name__ == '__main__':
    main()
