"""BASELINE configs[4] on one GPU (its per-GPU work): DINOv2 ViT-L/14 backbone at 336 px (577 tokens), 32-frame clips, batch 4,
MX-fp8 GEMM operands -- the step rate and a `roofline` object for its dominant kernel against the dense fp8 peak, in bench.py's
format.  `--dtype bf16` gives the same for the bf16 path of the same config.

    python tools/config4_roofline.py [--dtype fp8] [--steps 10] [--serial]

Per-GEMM-shape times: HIP events around every launch in extra one-kernel-at-a-time steps (mvf_prof_enable), as bench.py does;
`--serial` runs the timed steps that way too (the form to put under rocprofv3 --kernel-trace --stats)."""
import argparse
import ctypes
import json
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

TFLOP_PER_STEP = 99.875          # SURVEY.md section 8(d), configs[4] per GPU and step (B = 4)
PEAK = {'fp8': 5000.0, 'bf16': 2500.0}


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--dtype', default='fp8', choices=['fp8', 'bf16'])
    p.add_argument('--steps', type=int, default=10)
    p.add_argument('--serial', action='store_true')
    a = p.parse_args()
    dev = torch.device('cuda', 0)
    cfg = presets.make_cfg(network='TIMM-vit_large_patch14_dinov2.lvd142m', num_frames=32, batch_size=4, image_size=336,
                           SMART_FEATS='7,15,23', compute_dtype=a.dtype)
    cfg.MODEL.BASE_MODEL.LAYER = 24
    torch.manual_seed(1)
    model = build_model(cfg, 0).to(dev)
    wrapped = DataParallelModel(model)
    opt = construct_optimizer(wrapped, cfg)
    algo = get_algo(cfg)
    model.train()
    if a.serial:
        ops.VIT_LANE_MIN_ROWS = 1 << 62
    b, t, s = cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.NUM_FRAMES, cfg.IMAGE_SIZE
    g = torch.Generator().manual_seed(2)
    videos = torch.randn(b, 2, t, 3, s, s, generator=g).to(dev)
    seq_lens = torch.full((b, 2), 100, dtype=torch.long, device=dev)
    steps = torch.sort(torch.randint(0, 100, (b, 2, t), generator=g), dim=-1)[0].to(dev)
    masks = torch.ones(b, 2, t, device=dev)

    def step(lookahead=True):
        if lookahead:
            wrapped.prefetch(videos)
        opt.zero_grad()
        loss = algo.compute_loss(wrapped, videos, seq_lens, steps, masks)['loss']
        loss.backward()
        opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)
        return loss
    look = not a.serial
    if look:
        wrapped.prefetch(videos)
    for _ in range(3):
        loss = step(look)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step(look)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    # one-kernel-at-a-time steps with HIP events around every GEMM launch
    lane_rows, ops.VIT_LANE_MIN_ROWS = ops.VIT_LANE_MIN_ROWS, 1 << 62
    if look:
        step(False)          # drains the primed forward
    torch.cuda.synchronize()
    _lib.call('mvf_prof_enable', 1)
    nprof = 2
    for _ in range(nprof):
        step(False)
    torch.cuda.synchronize()
    _lib.call('mvf_prof_enable', 0)
    ops.VIT_LANE_MIN_ROWS = lane_rows
    G = 16
    ms, fl = (ctypes.c_double * G)(), (ctypes.c_double * G)()
    cnt, epi, nn, kk = (ctypes.c_int * G)(), (ctypes.c_int * G)(), (ctypes.c_int * G)(), (ctypes.c_int * G)()
    ng = ctypes.c_int(0)
    _lib.call('mvf_prof_collect', ms, fl, cnt, epi, nn, kk, G, ctypes.byref(ng))
    names = {(0, 3072, 1024): 'qkv', (0, 1024, 1024): 'proj', (2, 1024, 1024): 'proj + residual', (1, 4096, 1024): 'fc1 + gelu',
             (2, 1024, 4096): 'fc2 + residual', (3, 1024, 640): 'patch embedding'}
    groups = []
    for i in range(ng.value):
        groups.append({'name': names.get((epi[i], nn[i], kk[i]), 'gemm epi%d N=%d K=%d' % (epi[i], nn[i], kk[i])), 'launches': cnt[i],
                       'ms': ms[i], 'flop': fl[i], 'avg_us': round(ms[i] * 1e3 / max(cnt[i], 1), 1),
                       'tflops': round(fl[i] / (ms[i] * 1e-3) / 1e12, 1) if ms[i] > 0 else 0.0})
    dom = max(groups, key=lambda r: r['ms'])
    peak = PEAK[a.dtype]
    ach = dom['flop'] / (dom['ms'] * 1e-3) / 1e12
    tot_ms, tot_fl = sum(r['ms'] for r in groups), sum(r['flop'] for r in groups)
    out = {'metric': 'video-clips/sec/GPU, DINOv2 ViT-L/14 @ 336 px, 32-frame MV-Former (BASELINE configs[4], per-GPU work)',
           'value': round(2 * b / dt, 2), 'unit': 'clips/s', 'n_gpus': 1, 'steps': a.steps, 'ms_per_step': round(dt * 1e3, 2),
           'dtype': a.dtype, 'data': 'synthetic',
           'config': {'workload': 'BASELINE configs[4]: DINOv2 ViT-L/14, 336 px (577 tokens), 32 frames, batch 4/GPU, %s GEMM operands, '
                                  'full train step' % ('MX-fp8' if a.dtype == 'fp8' else 'bf16'),
                      'head_dtype': model.head_dtype, 'tflop_per_step_per_gpu': TFLOP_PER_STEP,
                      'tflops_algorithmic_per_gpu': round(TFLOP_PER_STEP / dt, 1), 'frac_of_peak_whole_step': round(TFLOP_PER_STEP / dt / peak, 4),
                      'last_loss': round(float(loss.item()), 4), 'serial': a.serial},
           'roofline': {'bound': 'mfma', 'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4), 'traffic': None,
                        'kernel': ('gemm_tc256_kernel (v_mfma_scale_f32_16x16x128_f8f6f4) / ' if a.dtype == 'fp8' else 'gemm_tc256_kernel / ') + dom['name'],
                        'launches': dom['launches'], 'avg_launch_us': dom['avg_us'], 'flop_per_launch': dom['flop'] / max(dom['launches'], 1),
                        'timing': 'HIP events around each launch on its stream, kernels serialized',
                        'all_gemm': {'achieved': round(tot_fl / (tot_ms * 1e-3) / 1e12, 1), 'frac': round(tot_fl / (tot_ms * 1e-3) / 1e12 / peak, 4),
                                     'ms_per_step': round(tot_ms / nprof, 3)},
                        'by_kernel': {r['name']: {'launches': r['launches'], 'avg_us': r['avg_us'], 'tflops': r['tflops']} for r in groups}}}
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
