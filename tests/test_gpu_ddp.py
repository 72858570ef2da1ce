"""Data-parallel equivalence on the device: two ranks (two processes sharing the one GPU of the test box, `gloo`
process group) each train on half of a batch -- SyncBN statistics exchange, bucketed gradient all-reduce through the flat
buffers, fused clip+Adam -- and must end up with the parameters a single process reaches on the whole batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _setup(gather=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from video_rep_learning_amd.utils import presets
    cfg = presets.make_cfg(network='TIMM-vit_small_patch16_224.dino', num_frames=8, batch_size=4, image_size=32,
                           compute_dtype='fp32', dropout=0.0)
    cfg.OPTIMIZER.LR.INITIAL_LR = 1e-3
    if gather:     # cross-GPU embedding all-gather enlarging the negative set
        cfg.SCL.NEGATIVE_TYPE = 'batch_noself'
        cfg.MI355X['GATHER_EMBEDDINGS'] = True
    return cfg


def _batch(cfg):
    g = torch.Generator().manual_seed(77)
    b, t, s = cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.NUM_FRAMES, cfg.IMAGE_SIZE
    videos = torch.randn(b, 2, t, 3, s, s, generator=g)
    seq_lens = torch.full((b, 2), 100, dtype=torch.long)
    steps = torch.sort(torch.randint(0, 100, (b, 2, t), generator=g), dim=-1)[0]
    masks = torch.ones(b, 2, t)
    return videos, seq_lens, steps, masks


def _train_one_step(cfg, batch, sync_bn):
    from video_rep_learning_amd.models import build_model
    from video_rep_learning_amd.algos import get_algo
    from video_rep_learning_amd.train import DataParallelModel
    from video_rep_learning_amd.utils.optimizer import construct_optimizer
    torch.manual_seed(5)
    model = build_model(cfg, 0).to('cuda:0')
    if sync_bn:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    wrapped = DataParallelModel(model)
    opt = construct_optimizer(wrapped, cfg)
    model.train()
    init = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if not k.startswith('backbone')}
    videos, seq_lens, steps, masks = batch
    opt.zero_grad()
    loss = get_algo(cfg).compute_loss(wrapped, videos.to('cuda:0'), seq_lens, steps, masks)['loss']
    loss.backward()
    opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if not k.startswith('backbone')}
    return loss.item(), sd, init


def _worker(rank, world, port, ret, gather):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    cfg = _setup(gather)
    full = _batch(cfg)
    ref_loss, ref_sd, init = _train_one_step(cfg, full, sync_bn=False) if rank == 0 else (None, None, None)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        per = cfg.TRAIN.BATCH_SIZE // world
        mine = tuple(t[rank * per:(rank + 1) * per] for t in full)
        loss, sd, _ = _train_one_step(cfg, mine, sync_bn=True)
        losses = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(losses, torch.tensor([loss]))
        if rank == 0:
            # per tensor: relative L2 of the UPDATE (Adam turns rounding-level gradients into +-lr steps of arbitrary
            # sign, so exactly-null-gradient biases are skipped and the metric is dominated by well-conditioned elements)
            null = ('linear_V2d.bias', 'linear_K2d.bias', 'fc_layers.1.bias', 'fc_layers.5.bias', 'feed_forward.fc2.bias',
                    'embedding_layer.bias', 'net.0.bias', 'num_batches_tracked')
            worst = ('', 0.0)
            for k, v in ref_sd.items():
                if not v.dtype.is_floating_point or any(k.endswith(n) for n in null):
                    continue
                if 'running_' in k:
                    e = (sd[k].double() - v.double()).abs().max().item() / max(v.abs().max().item(), 1e-3)
                else:
                    du_ref, du_got = v.double() - init[k].double(), sd[k].double() - init[k].double()
                    e = ((du_got - du_ref).norm() / du_ref.norm().clamp_min(1e-12)).item()
                if e > worst[1]:
                    worst = (k, e)
            ret['out'] = (ref_loss, float(sum(l.item() for l in losses) / world), worst)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('gather', [False, True])
def test_two_ranks_equal_one_process_on_the_whole_batch(gather):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret, gather), nprocs=2, join=True)
    ref_loss, mean_loss, worst = ret['out']
    # 'single_noself' negatives: the global loss is the mean of the per-rank losses; with gathered embeddings and
    # 'batch_noself' negatives every rank evaluates the SAME global loss, so the mean is that loss again
    assert abs(ref_loss - mean_loss) <= 1e-4 * abs(ref_loss), (ref_loss, mean_loss)
    assert worst[1] <= 2e-2, worst
