"""Diagnostic (GPU box): which PART of the head stretches the pipelined backbone forward?  The frozen-backbone forwards of
BASELINE configs[1] run back to back on their side streams exactly as in training (two lanes, one batch ahead); instead of the
whole head, ONE part of it runs on the caller's stream each step, with real kernels on data of the real shapes:
    none | lstp | linv | fc | enc_chain | enc_fp32 | tail | opt | all (= the real head)
Prints the wall time per step; the difference to `none` is what that part costs the step.
    python tools/stretch_parts.py [--steps 300] [--parts none,lstp,...]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib, ops  # noqa: E402
from video_rep_learning_amd.utils import presets  # noqa: E402
from video_rep_learning_amd.utils.optimizer import construct_optimizer  # noqa: E402
from video_rep_learning_amd.models import build_model  # noqa: E402
from video_rep_learning_amd.models.utils import Encoder  # noqa: E402
from video_rep_learning_amd.algos import get_algo  # noqa: E402
from video_rep_learning_amd.train import DataParallelModel  # noqa: E402
from video_rep_learning_amd.datasets import synthetic  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--steps', type=int, default=300)
    p.add_argument('--parts', default='none,lstp,linv,fc,enc_chain,enc_fp32,tail,opt,all,none')
    p.add_argument('--lanes', type=int, default=0, help='backbone lanes (0 = product default)')
    p.add_argument('--gemm-cus', type=int, default=0, help='CU budget of the persistent backbone GEMM (0 = all)')
    a = p.parse_args()
    dev = torch.device('cuda', 0)
    if a.lanes:
        ops.VIT_LANES = a.lanes
    if a.gemm_cus:
        _lib.call('mvf_gemm_tc_set_cus', a.gemm_cus)
    cfg = presets.baseline_config_2('bf16')
    torch.manual_seed(1)
    model = build_model(cfg, 0).to(dev)
    wrapped = DataParallelModel(model)
    opt = construct_optimizer(wrapped, cfg)
    algo = get_algo(cfg)
    loader = synthetic.SyntheticClips(cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.NUM_FRAMES, cfg.IMAGE_SIZE, iters=1, seed=1234, device=dev, resident=True)
    (v0, v1), _l, seq_lens, steps, masks, _n = next(iter(loader))
    videos = torch.stack([v0, v1], dim=1)
    seq_lens, steps, masks = seq_lens.to(dev), steps.to(dev), masks.to(dev)
    model.train()
    em = model.embed
    g = torch.Generator(device='cpu').manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    # fixed operands of the parts (real shapes: 8 clips x 3 entities x 32 frames = 768 rows)
    x387, x256 = rnd(768, 387).requires_grad_(True), rnd(8, 96, 256).requires_grad_(True)
    pooled = rnd(768, 2304).requires_grad_(True)
    go512, go256, go384 = rnd(768, 256), rnd(8, 96, 256), rnd(768, 384)
    encs = {}
    for name, hd in (('enc_chain', 'bf16'), ('enc_fp32', 'fp32')):
        torch.manual_seed(2)
        e = Encoder(256, 0.1, 8, 1024, 3).to(dev)
        e.head_dtype = hd
        e.train()
        encs[name] = e
    ds = ops.DropoutState(1)
    vm = masks.view(8, 1, 32)
    xt = rnd(8, 3, 32, 256).requires_grad_(True)
    one = torch.ones((), device=dev)
    gpool = None

    def part_lstp(taps):
        nonlocal gpool
        ca = em.pooling.cross_att
        vec = ca.query_vectors(None, taps.n_clips, taps.n_frames)
        pl, _rs = ops.lstp_pool(vec, taps.tensors, taps.n_clips * taps.n_frames, taps.n_tokens, taps.n_frames, 3, ca.d_model)
        if gpool is None:
            gpool = torch.randn_like(pl)
        pl.backward(gpool)

    def part_linv():
        ca = em.pooling.cross_att
        y = ops.linear(pooled, ca.linear_V2d.weight, ca.linear_V2d.bias)
        y.backward(go384)

    def part_fc():
        x = x387
        mods = list(em.fc_layers)
        for i in range(0, len(mods), 4):
            x = ops.dropout_add(x, None, mods[i].p, True, ds)
            x = em._bn(ops.linear(x, mods[i + 1].weight, mods[i + 1].bias), mods[i + 2], relu=True)
        x = ops.linear(x, em.video_emb.weight, em.video_emb.bias, table=em.video_pos_enc.table(32, dev), tab_div=1, tab_mod=32)
        x = ops.dropout_add(x, None, em.video_pos_enc.dout_p, True, ds)
        x.backward(go512)

    def part_enc(name):
        y = encs[name](x256, src_mask=vm, drop_state=ds)
        y.backward(go256)

    def part_tail():
        x = ops.final_reduce(xt, em.smart_final)
        x = ops.linear(x.reshape(256, -1), em.embedding_layer.weight, em.embedding_layer.bias).view(8, 32, -1)
        x = ops.l2_normalize(model.ssl_projection(x))
        loss = algo.compute_sequence_loss(x.view(4, 2, 32, -1), seq_lens, steps, masks)['loss']
        ops.backward(loss)

    def part_opt():
        opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)

    def run(part):
        def step():
            wrapped.prefetch(videos)
            if part == 'all':
                opt.zero_grad()
                loss = algo.compute_loss(wrapped, videos, seq_lens, steps, masks)['loss']
                ops.backward(loss)
                opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)
                return
            taps, cls, done = model._stash.pop(0)[1:]
            torch.cuda.current_stream(dev).wait_event(done)
            if part == 'lstp':
                from video_rep_learning_amd.models.mvformer import Taps
                part_lstp(Taps(taps, 8, 32, 196))
            elif part == 'linv':
                part_linv()
            elif part == 'fc':
                part_fc()
            elif part in encs:
                part_enc(part)
            elif part == 'tail':
                part_tail()
            elif part == 'opt':
                part_opt()
        model._stash = []
        wrapped.prefetch(videos)
        for _ in range(30):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3

    base = None
    for part in a.parts.split(','):
        w = run(part)
        if part == 'none' and base is None:
            base = w
        print('%-10s wall %.3f ms/step   %+.3f ms vs the forwards alone' % (part, w, w - (base if base is not None else w)), flush=True)


if __name__ == '__main__':
    main()
