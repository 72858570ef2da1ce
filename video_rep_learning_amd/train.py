"""Training entry point with the reference's CLI and loop semantics (CARL_MVF/train.py:57-341):

    python -m torch.distributed.run --nproc-per-node N train.py --cfg_file configs_mvf/penn_mvf.yml \
        --workdir W --logdir L [--opts K V ...]

Per iteration (train.py:94-171): zero_grad -> algo.compute_loss(model, videos, seq_lens, chosen_steps,
video_masks) -> backward -> clip_grad_norm_(GRAD_CLIP) -> optimizer step -> loss all-reduce(avg) for logging.
Differences, all on purpose:
  * device/backend are not hard-wired: HIP device + 'nccl' (= RCCL over xGMI) by default;
  * gradient averaging is done by the optimizer's flat-buffer GradReducer over the 4.8 M TRAINABLE
    parameters (async bucketed RCCL all-reduce overlapped with backward) instead of DDP(find_unused_parameters)
    over all 90 M; SyncBatchNorm statistics are exchanged by the HIP BN op;
  * clip + Adam is one fused HIP kernel pair on the flat buffers;
  * the loss is accumulated on the device and all-reduced / synced to the host once per REPORT_INTERVAL and at
    epoch end (same logged numbers, no per-iteration .item() stall);
  * bf16 compute needs no GradScaler (`algo.scaler` stays None).
"""
import datetime
import json
import os
import pprint
import random
import shutil
import time

# (the host driver of the MI355X boxes only supports dmabuf IPC: RCCL's buffer exchange needs this before the runtime loads)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
# Hardware queues: HIP deals a process's streams over GPU_MAX_HW_QUEUES (default 4) hardware queues, and two streams on one queue run one
# after the other (profiles/r05/order_probe.txt).  A data-parallel rank owns the caller's stream, the lookahead stream, the second backbone
# lane and RCCL's communication stream.  Round 5 asked for 8 queues here, unmeasured; round 6 measured it on the forced one-rank RCCL step
# (profiles/r06/rccl_forced_1rank.txt, three alternations on one box): plain step 10.67-10.73 ms, with the collectives live and the runtime's
# 4 queues 10.85-10.87, with 8 or 16 queues 11.61-11.66 -- the all-reduce then completes at once (0.19 ms exposed instead of 7 ms of queueing
# behind a backbone lane, which the one-batch lookahead hides anyway) but every step pays 0.76 ms for the extra queues.  So the runtime's
# default stays; MVF_HW_QUEUES=<n> asks for n queues (A/B knob).  Read when the runtime loads, hence before torch is imported.
if os.environ.get('MVF_HW_QUEUES', '0') not in ('', '0'):
    os.environ.setdefault('GPU_MAX_HW_QUEUES', os.environ['MVF_HW_QUEUES'])

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from . import ops
from .utils import distributed as du
from .utils import logging
from .utils.parser import parse_args, load_config, setup_train_dir
from .utils.optimizer import construct_optimizer, construct_scheduler, get_lr
from .models import build_model, save_checkpoint, load_checkpoint
from .algos import get_algo
from .datasets import synthetic

logger = logging.get_logger(__name__)


class DataParallelModel(nn.Module):
    """Keeps the `model.module` surface of DistributedDataParallel (train.py:88,316; models/__init__.py:24).
    Gradient synchronisation itself lives in utils.distributed.GradReducer (driven by the optimizer)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **kw):
        return self.module(*a, **kw)

    def prefetch(self, videos):
        """videos [B, V, T, 3, H, W] of the NEXT iteration: start its frozen-backbone forward on the side stream."""
        if hasattr(self.module, 'prefetch') and videos.is_cuda:
            b, v = videos.shape[:2]
            self.module.prefetch(videos.view(b * v, *videos.shape[2:]))


class ScalarWriter:
    """Minimal SummaryWriter stand-in (tensorboard is not a dependency): scalars -> LOGDIR/train_logs/scalars.jsonl."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, 'scalars.jsonl')

    def add_scalar(self, tag, value, step):
        if du.is_root_proc():
            with open(self.path, 'a') as f:
                f.write(json.dumps({'tag': tag, 'value': float(value), 'step': int(step)}) + '\n')


def train(cfg, train_loader, model, optimizer, scheduler, algo, cur_epoch, summary_writer, data_preprocess,
          device='cuda', max_iters=0):
    model.train()
    optimizer.zero_grad()
    data_size = len(train_loader) if not max_iters else min(len(train_loader), max_iters)
    if hasattr(train_loader.sampler, 'set_epoch'):
        train_loader.sampler.set_epoch(cur_epoch)
    if 'BACKBONE_WARMUP' in cfg.TRAIN:
        model.module.embed.set_warmup_status(cur_epoch < cfg.TRAIN.BACKBONE_WARMUP)
    total = {}      # device-side running sums of the per-iteration (NaN -> 0) local losses
    loss = None
    # one-batch lookahead: batch i+1 is fetched + augmented, and its frozen-backbone forward is started on the side
    # stream, BEFORE the head work of batch i is enqueued (models/transformer.py "backbone pipeline")
    def batches():
        for cur_iter, (videos, _labels, seq_lens, chosen_steps, video_masks, names) in enumerate(train_loader):
            if max_iters and cur_iter >= max_iters:
                return
            videos = synthetic.preproc_views(videos[0], videos[1], data_preprocess, device)
            if hasattr(model, 'prefetch'):
                model.prefetch(videos)
            yield cur_iter, videos, seq_lens, chosen_steps, video_masks
    stream_it = batches()
    nxt = next(stream_it, None)
    while nxt is not None:
        cur_iter, videos, seq_lens, chosen_steps, video_masks = nxt
        nxt = next(stream_it, None)
        optimizer.zero_grad()
        loss_dict = algo.compute_loss(model, videos, seq_lens, chosen_steps, video_masks)
        loss = loss_dict['loss']
        ops.backward(loss)                 # loss.backward() with a cached seed gradient
        clip = cfg.OPTIMIZER.GRAD_CLIP
        if hasattr(optimizer, 'reducer'):
            optimizer.step(max_norm=clip if clip > 0 else 0.0)
        else:
            if clip > 0:
                torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
            optimizer.step()
        for key in loss_dict:
            v = torch.nan_to_num(loss_dict[key].detach(), nan=0.0)
            total[key] = v if key not in total else total[key] + v
        if cur_iter % cfg.LOGGING.REPORT_INTERVAL == 0:
            logger.info(f'iter {data_size * cur_epoch + cur_iter}, training loss: {loss.item():.3f}')
    out = {}
    for key in total:
        out[key] = du.all_reduce([total[key].clone()])[0].item() / data_size
    summary_writer.add_scalar('train/learning_rate', get_lr(optimizer)[0], cur_epoch)
    for key in out:
        summary_writer.add_scalar(f'train/{key}', out[key], cur_epoch)
    logger.info('epoch {}, train loss: {:.3f}'.format(cur_epoch, out.get('loss', float('nan'))))
    if cur_epoch != cfg.TRAIN.MAX_EPOCHS - 1:
        scheduler.step()
    return out


def val(cfg, val_loader, model, algo, cur_epoch, summary_writer, data_preprocess, device='cuda', max_iters=0):
    model.eval()
    data_size = len(val_loader) if not max_iters else min(len(val_loader), max_iters)
    total = {}
    with torch.no_grad():
        for cur_iter, (videos, labels, seq_lens, chosen_steps, video_masks, names) in enumerate(val_loader):
            if max_iters and cur_iter >= max_iters:
                break
            videos = synthetic.preproc_views(videos[0], videos[1], data_preprocess, device)
            loss_dict = algo.compute_loss(model, videos, seq_lens, chosen_steps, video_masks, training=False)
            for key in loss_dict:
                v = torch.nan_to_num(loss_dict[key].detach(), nan=0.0)
                total[key] = v if key not in total else total[key] + v
    out = {k: du.all_reduce([v.clone()])[0].item() / data_size for k, v in total.items()}
    for key in out:
        summary_writer.add_scalar(f'val/{key}', out[key], cur_epoch)
    logger.info('epoch {}, val loss: {:.3f}'.format(cur_epoch, out.get('loss', float('nan'))))
    return out


def setup_distributed(args):
    """Process-group init from the launcher's environment (train.py:236-262), backend not hard-wired."""
    world = int(os.getenv('WORLD_SIZE', '1'))
    if os.environ.get('OMPI_COMM_WORLD_SIZE') is None:
        rank = int(os.getenv('RANK', args.local_rank))
    else:
        rank = int(os.getenv('OMPI_COMM_WORLD_RANK')) * max(torch.cuda.device_count(), 1) + args.local_rank
    device = args.device or ('cuda' if torch.cuda.is_available() else 'cpu')
    on_gpu = str(device).startswith('cuda')          # 'cuda', 'cuda:0', torch.device('cuda', 0)
    backend = args.backend or ('nccl' if on_gpu else 'gloo')
    if on_gpu:
        # an explicit index ('cuda:3') names the device; the bare form takes this rank's local one
        dev_index = torch.device(device).index
        torch.cuda.set_device(args.local_rank if dev_index is None else dev_index)
    if world > 1 or 'MASTER_ADDR' in os.environ:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        torch.distributed.init_process_group(backend=backend, init_method='env://', world_size=world, rank=rank,
                                             timeout=datetime.timedelta(seconds=72000))
    return device, world, rank


def plumbing_run(cfg, args, model, optimizer, scheduler, algo, train_loader, data_preprocess, device, start_epoch):
    """`--plumbing` (BASELINE configs[0], "runs without a GPU"): everything of train.py:230-307 AROUND the kernels has already
    run when this is called -- parser, config merge, LOGDIR/config.yml, process group (gloo), build_model, SyncBN conversion,
    construct_optimizer (flat buffers, gradient buckets), loaders, load_checkpoint, scheduler.  What is left: one loader batch,
    a checkpoint round trip, the collectives of an iteration on the flat gradient buffer -- and the first model call, which
    must FAIL on a device without HIP: the product has no CPU compute path and does not grow one here."""
    from ._lib import MvfError
    it = iter(train_loader)
    videos, _labels, seq_lens, chosen_steps, video_masks, names = next(it)
    videos = synthetic.preproc_views(videos[0], videos[1], data_preprocess, device)
    b, t = cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.NUM_FRAMES
    assert videos.shape[:3] == (b, 2, t) and tuple(seq_lens.shape) == (b, 2) and tuple(chosen_steps.shape) == (b, 2, t)
    # checkpoint round trip in the reference's format (models/__init__.py:17-60): parameters, BN buffers, Adam state
    # ... in a scratch directory UNDER the log directory, removed afterwards: a LOGDIR that already holds checkpoint E must not
    # gain a "checkpoint E + 1" with epoch-E weights that a later real run would resume from
    before = {k: v.detach().clone() for k, v in model.module.state_dict().items()}
    logdir = cfg.LOGDIR
    scratch = os.path.join(logdir, 'plumbing_scratch')
    cfg.LOGDIR = scratch
    try:
        if du.is_root_proc():
            shutil.rmtree(scratch, ignore_errors=True)
            save_checkpoint(cfg, model, optimizer, start_epoch)
        du.synchronize()
        with torch.no_grad():
            for p in model.module.parameters():
                if p.requires_grad:
                    p.add_(1.0)
        restored = load_checkpoint(cfg, model, optimizer)
        from .utils import checkpoint as _ckpt
        ck = _ckpt._read(_ckpt.latest(cfg))          # what was on disk, for the caller (the file itself does not survive)
        ck_summary = {'file': os.path.basename(_ckpt.latest(cfg)), 'keys': sorted(ck), 'epoch': ck['epoch'],
                      'model_prefixes': sorted({k.split('.')[0] for k in ck['model_state']})}
        du.synchronize()
    finally:
        cfg.LOGDIR = logdir
        if du.is_root_proc():
            shutil.rmtree(scratch, ignore_errors=True)
    after = model.module.state_dict()
    assert restored == start_epoch + 1 and all(torch.equal(before[k], after[k]) for k in before), 'checkpoint round trip'
    # the iteration's collectives on host tensors (gloo): gradient buckets of the flat buffer, loss all-reduce
    optimizer.zero_grad()
    if hasattr(optimizer, 'reducer'):
        optimizer.flat.flat_g.fill_(float(du.get_rank() + 1))
        optimizer.flat.dirty = True                  # written behind the optimizer's back: the next zero_grad() must fill
        active = optimizer.reducer.active
        gscale = optimizer.reducer.finish()          # launches every bucket's all-reduce (SUM) and waits
        want = float(sum(range(1, du.get_world_size() + 1))) if active else float(du.get_rank() + 1)
        assert gscale == 1.0 / du.get_world_size()
        assert torch.equal(optimizer.flat.flat_g, torch.full_like(optimizer.flat.flat_g, want)), 'gradient all-reduce'
        optimizer.zero_grad()
    loss_log = du.all_reduce([torch.tensor([float(du.get_rank())])])[0].item()
    assert abs(loss_log - (du.get_world_size() - 1) / 2.0) < 1e-6
    assert get_lr(optimizer)[0] == cfg.OPTIMIZER.LR.INITIAL_LR or cfg.OPTIMIZER.LR.DECAY_TYPE == 'cosinewarmup'
    reached = None
    try:
        algo.compute_loss(model, videos, seq_lens, chosen_steps, video_masks)
    except MvfError as e:
        reached = str(e)
    if reached is None:
        raise RuntimeError('--plumbing expects a device without HIP: the model call succeeded, so this is a real run -- drop the flag')
    logger.info('plumbing run complete on %s/%s: config, process group, model (%d parameters), optimizer (%d trainable), loader, '
                'checkpoint round trip, collectives; stopped at the first kernel call: %s' % (
                    device, args.backend or 'gloo', sum(p.numel() for p in model.parameters()),
                    optimizer.flat.numel if hasattr(optimizer, 'flat') else -1, reached))
    du.synchronize()
    if du.is_dist():
        torch.distributed.destroy_process_group()
    return {'plumbing': True, 'stopped_at': reached, 'checkpoint': ck_summary}


def main(argv=None):
    args = parse_args(argv)
    cfg = load_config(args)
    setup_train_dir(cfg, cfg.LOGDIR, args.continue_train, args.tempcfg)
    cfg.PATH_TO_DATASET = os.path.join(args.workdir, cfg.PATH_TO_DATASET)
    device, world, rank = setup_distributed(args)
    cfg.NUM_GPUS = torch.cuda.device_count()
    args.world_size, args.rank = world, rank
    random.seed(cfg.RNG_SEED)
    np.random.seed(cfg.RNG_SEED)
    torch.manual_seed(cfg.RNG_SEED)
    logging.setup_logging(cfg.LOGDIR)
    summary_writer = ScalarWriter(os.path.join(cfg.LOGDIR, 'train_logs'))
    logger.info('Train with config:')
    logger.info(pprint.pformat(cfg))

    model = build_model(cfg, args.local_rank).to(device)
    if du.collectives_active():            # world > 1 (train.py:283), or a forced one-rank group
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
        if str(device).startswith('cuda'):     # 'cuda', 'cuda:0', torch.device('cuda', 0)
            du.reserve_collective_cus()    # RCCL's kernels get CUs of their own beside the persistent GEMM
    model = DataParallelModel(model)
    optimizer = construct_optimizer(model, cfg)
    algo = get_algo(cfg)
    algo.scaler = None

    if not args.synthetic:
        raise NotImplementedError('only --synthetic data is wired in this build: the video decoders / samplers of '
                                  'CARL_MVF/datasets are host-side I/O outside the hot path (SURVEY section 2)')
    raw_hw = tuple(args.synthetic_raw) if getattr(args, 'synthetic_raw', None) else None
    train_loader, _ = synthetic.construct_dataloader(cfg, 'train', device=device, rank=rank, raw_hw=raw_hw)
    val_loader, _ = synthetic.construct_dataloader(cfg, 'val', device=device, rank=rank, iters=4, raw_hw=raw_hw)
    train_preproc = synthetic.get_data_preprocess(cfg, 'train', raw=raw_hw is not None)   # train.py:294-297
    val_preproc = synthetic.get_data_preprocess(cfg, 'val', raw=raw_hw is not None)

    start_epoch = load_checkpoint(cfg, model, optimizer)
    cfg.TRAIN.MAX_ITERS = cfg.TRAIN.MAX_EPOCHS * len(train_loader)
    scheduler = construct_scheduler(optimizer, cfg)
    if args.plumbing:
        return plumbing_run(cfg, args, model, optimizer, scheduler, algo, train_loader, train_preproc, device, start_epoch)
    for cur_epoch in range(start_epoch, cfg.TRAIN.MAX_EPOCHS):
        logger.info(f'Traning epoch {cur_epoch}/{cfg.TRAIN.MAX_EPOCHS}, {len(train_loader)} iters each epoch')
        t0 = time.time()
        train(cfg, train_loader, model, optimizer, scheduler, algo, cur_epoch, summary_writer, train_preproc, device,
              args.max_iters)
        if str(device).startswith('cuda'):
            torch.cuda.synchronize()
        print('train done in (m): ' + str((time.time() - t0) / 60.0))
        if du.is_root_proc() and ((cur_epoch + 1) % cfg.CHECKPOINT.SAVE_INTERVAL == 0 or cur_epoch == cfg.TRAIN.MAX_EPOCHS - 1):
            save_checkpoint(cfg, model, optimizer, cur_epoch)
        if (cur_epoch + 1) % cfg.EVAL.VAL_INTERVAL == 0 or cur_epoch == cfg.TRAIN.MAX_EPOCHS - 1:
            val(cfg, val_loader, model, algo, cur_epoch, summary_writer, val_preproc, device, args.max_iters)
        du.synchronize()
    if du.is_dist():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
