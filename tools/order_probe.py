"""Diagnostic (GPU box): does a config run slower AFTER another one in the same process?  python tools/order_probe.py [first]
Runs (optionally) BASELINE configs[1] for 20 steps, drops it, then configs[2] (the fg99 head): full step and backbone forwards only."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib, ops  # noqa: E402
from video_rep_learning_amd.utils import presets  # noqa: E402
from video_rep_learning_amd.utils.optimizer import construct_optimizer  # noqa: E402
from video_rep_learning_amd.models import build_model  # noqa: E402
from video_rep_learning_amd.algos import get_algo  # noqa: E402
from video_rep_learning_amd.train import DataParallelModel  # noqa: E402

FG99 = dict(network='TIMM-vit_base_patch16_224.dino', num_frames=32, batch_size=4, SMART_TOKENS=6, CAPACITY_SCALAR=6, EMBEDDING_SIZE=256,
            SMART_FEATS='9,10,11', SMART_FINAL='avg')
PENN = dict(network='TIMM-vit_base_patch16_224.dino', num_frames=32, batch_size=4)


def run(kw, steps, tag):
    dev = torch.device('cuda')
    cfg = presets.make_cfg(compute_dtype='bf16', **kw)
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
    stp = torch.sort(torch.randint(0, 100, (b, 2, t), generator=g), dim=-1)[0].to(dev)
    masks = torch.ones(b, 2, t, device=dev)

    def full():
        wrapped.prefetch(videos)
        opt.zero_grad()
        loss = algo.compute_loss(wrapped, videos, seq_lens, stp, masks)['loss']
        loss.backward()
        opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)

    def fwd():
        wrapped.prefetch(videos)
        model._stash.pop(0)

    def timed(fn, n):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    wrapped.prefetch(videos)
    a = timed(full, steps)
    bms = timed(fwd, steps)
    wgs = __import__('ctypes').c_int(0)
    _lib.call('mvf_gemm_tc_get_wgs', __import__('ctypes').byref(wgs))
    print('%s: full step %.3f ms, forwards only %.3f ms; spare set %s, GEMM budget %d workgroups' % (tag, a, bms, ops._SPARE_SET[0], wgs.value), flush=True)
    model._stash.clear()


if len(sys.argv) > 1 and sys.argv[1] == 'first':
    run(PENN, 30, 'configs[1] first')
    gc.collect()
    torch.cuda.empty_cache()
run(FG99, 40, 'configs[2]')
