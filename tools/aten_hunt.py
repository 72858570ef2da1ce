"""Which ATen / runtime-copy kernels does one steady-state training step (BASELINE configs[1]) launch, and from where?
torch.profiler with Python stacks over a few steps after warm-up; prints every device kernel that is NOT one of the library's,
with the innermost repo frames of the op that launched it.  Usage: python3 tools/aten_hunt.py [steps]"""
import os
import sys
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

from video_rep_learning_amd.utils import presets  # noqa: E402
from video_rep_learning_amd.utils.optimizer import construct_optimizer  # noqa: E402
from video_rep_learning_amd.models import build_model  # noqa: E402
from video_rep_learning_amd.algos import get_algo  # noqa: E402
from video_rep_learning_amd.train import DataParallelModel  # noqa: E402
from video_rep_learning_amd.datasets import synthetic  # noqa: E402
from video_rep_learning_amd import ops  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device('cuda', 0)
    cfg = presets.baseline_config_2('bf16')
    torch.manual_seed(cfg.RNG_SEED)
    model = build_model(cfg, 0).to(dev)
    wrapped = DataParallelModel(model)
    opt = construct_optimizer(wrapped, cfg)
    algo = get_algo(cfg)
    loader = synthetic.SyntheticClips(cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.NUM_FRAMES, cfg.IMAGE_SIZE, iters=1, seed=1234, device=dev, resident=True)
    (v0, v1), _lab, seq_lens, stp, masks, _n = next(iter(loader))
    videos = torch.stack([v0, v1], dim=1)
    seq_lens, stp, masks = seq_lens.to(dev), stp.to(dev), masks.to(dev)
    model.train()

    def step():
        wrapped.prefetch(videos)
        opt.zero_grad()
        loss = algo.compute_loss(wrapped, videos, seq_lens, stp, masks)['loss']
        ops.backward(loss)
        opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)

    wrapped.prefetch(videos)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    by_site = defaultdict(Counter)
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            continue
        kernels = [k.name for k in getattr(ev, 'kernels', [])]
        if not kernels:
            continue
        for kn in kernels:
            mine = any(t in kn for t in ('gemm_tc', 'vit_attn', 'layernorm_kernel', 'lstp', 'hgemm', 'hlinear', 'tattn', 'scl_', 'adam',
                                         'sqnorm', 'bn_', 'ln_', 'im2col', 'cls_row', 'final_reduce', 'l2norm', 'dropout', 'relu',
                                         'colsum', 'concat_onehot', 'token_pool', 'quant', 'vattn', 'grad_prep', 'gelu', 'enc_fwd', 'enc_bwd', 'rowlin_',
                                         'head_dw', 'head_pack', 'static_query', 'vit_qkv'))
            if mine:
                continue
            frames = [f for f in (ev.stack or []) if 'video_rep_learning_amd' in f or 'bench.py' in f or 'aten_hunt' in f]
            site = ' <- '.join(fr.split('video_rep_learning_amd/')[-1] for fr in frames[:3]) or '(no repo frame: autograd engine / runtime)'
            chain, par = [], getattr(ev, 'cpu_parent', None)
            while par is not None and len(chain) < 4:
                chain.append(par.name[:40])
                par = getattr(par, 'cpu_parent', None)
            site = 'inside ' + ' < '.join(chain) if chain else site
            site += '   shapes ' + str(getattr(ev, 'input_shapes', None))[:100] + ('  thread %s' % ev.thread)
            by_site[(ev.name, kn[:70])][site] += 1
    print('non-library device kernels over %d steps (op, kernel): count per call site' % steps)
    for (op, kn), sites in sorted(by_site.items(), key=lambda kv: -sum(kv[1].values())):
        print('%5.1f/step  %-28s %s' % (sum(sites.values()) / steps, op, kn))
        for s, c in sites.most_common(6):
            print('            %4.1f  %s' % (c / steps, s))


if __name__ == '__main__':
    main()
