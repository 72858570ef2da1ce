"""Diagnostic (GPU box): torch.profiler view of ONE training step's head part -- which ATen ops (copies, adds, fills)
the Python side still issues around the HIP kernels, with shapes and the Python call site."""
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
        loss.backward()
        opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=40,
                                                              max_name_column_width=50, max_shapes_column_width=60))
    print(prof.key_averages(group_by_stack_n=4).table(sort_by='self_cuda_time_total', row_limit=45,
                                                       max_name_column_width=40, max_src_column_width=90))
    ev = [e for e in prof.events() if e.name in ('aten::copy_', 'aten::add_', 'aten::add', 'aten::fill_', 'aten::mul')]
    print('host wall of the step: cpu ops total %.1f ms' % (sum(e.cpu_time_total for e in prof.events() if e.cpu_parent is None) / 1e3))


if __name__ == '__main__':
    main()
