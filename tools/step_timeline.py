"""Diagnostic (GPU box): where does the pipelined training step spend the time the frozen backbone alone does not need?
HIP events (no profiler: rocprofv3's per-launch cost makes the 300-launch step host-bound) on the backbone's side stream and on the
head's stream, N sustained steps of bench.py's step at BASELINE configs[1]:
    B_start(i), B_end(i): first / last kernel of the backbone forward of batch i (both lanes joined)
    H_end(i):             end of head forward + backward + clip + Adam of batch i
and the host's own time per step.  Prints means over the last 60 % of the steps.
    python tools/step_timeline.py [--steps 300] [--no-head]"""
import argparse
import ctypes
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
from video_rep_learning_amd.datasets import synthetic  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--steps', type=int, default=300)
    p.add_argument('--no-head', action='store_true', help='backbone forwards only (the floor)')
    p.add_argument('--fwd-only', action='store_true', help='head forward + loss only (no backward, no optimizer)')
    p.add_argument('--dummy', default='', help="instead of the head: 'tiny:N' = N one-workgroup elementwise launches per step, "
                                               "'gemm:N' = N head-sized fp32 GEMM launches (768 x 512 x 512) per step")
    p.add_argument('--prof', type=int, default=0, help='after the timed loop: N more steps with the library\'s per-launch HIP events around '
                                                       'every backbone GEMM / fused qkv + attention launch (pipelined, both lanes)')
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

    marks = []            # per backbone launch: (start event, end event)
    inner = model._launch_backbone

    def launch(x, ready_event=None):
        cur = torch.cuda.current_stream(dev)
        if getattr(model, '_side', None) is None:
            return inner(x, ready_event)           # (first call creates the stream)
        if ready_event is None:
            ready_event = torch.cuda.Event()
            ready_event.record(cur)
        model._side.wait_event(ready_event)
        s = torch.cuda.Event(enable_timing=True)
        s.record(model._side)
        out = inner(x, ready_event)
        e = torch.cuda.Event(enable_timing=True)
        e.record(model._side)
        marks.append((s, e))
        return out
    model._launch_backbone = launch

    h_end, host = [], []
    spin = None
    if a.dummy.startswith('spin'):
        spin = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'probes', 'libspin.so'))
    spin_buf = torch.zeros(32 * 1024 * 1024, device=dev, dtype=torch.bfloat16)      # up to 64 MB (default footprint 1.5 MB: one encoder layer's weight images)
    dummy_small = torch.zeros(256, device=dev)
    dummy_x, dummy_w = torch.randn(768, 512, device=dev), torch.randn(512, 512, device=dev)
    dummy_o, dummy_o2 = torch.zeros(3, 384, device=dev), torch.zeros(768, 512, device=dev)
    dummy_q, dummy_wq = torch.randn(3, 2304, device=dev), torch.randn(384, 2304, device=dev)

    def step():
        t0 = time.perf_counter()
        wrapped.prefetch(videos)
        if a.dummy:
            taps, cls, done = model._stash.pop(0)[1:]
            torch.cuda.current_stream(dev).wait_event(done)
            kind, cnt = a.dummy.split(':')
            for _ in range(int(cnt)):
                if kind == 'tiny':
                    dummy_small.add_(1.0)
                elif kind.startswith('spin2'):   # spin2-<grid>-<threads>-<lds bytes>-<us>-<mode>:N   (mode 0 sleep, 1 L2 stream, 2 MFMA, 3 LDS)
                    _f = [int(v) for v in kind.split('-')[1:]]
                    _g, _t, _l, _u, _m = _f[:5]
                    _kb = _f[5] if len(_f) > 5 else 1536      # footprint of the streamed buffer in KB
                    spin.spin_launch2(_g, _t, _l, _u, _m, ctypes.c_void_p(spin_buf.data_ptr()), ctypes.c_size_t(_kb * 1024),
                                      ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
                elif kind.startswith('spin'):    # spin-<grid>-<lds bytes>-<us>:N   (tools/probes/spin_kernel.hip)
                    _g, _l, _u = (int(v) for v in kind.split('-')[1:])
                    spin.spin_launch(_g, _l, _u, None, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
                elif kind == 'rawgemv':   # the same launch without the Python op around it (no output allocation, no autograd node)
                    _lib.call('mvf_hgemm', dummy_q.data_ptr(), 2304, 1, dummy_wq.data_ptr(), 1, 2304, dummy_o.data_ptr(), 384, None, None, 0, 0, 1, 1,
                              3, 384, 2304, 1.0, 0, 0, _lib.stream())
                elif kind == 'rawgemm':
                    _lib.call('mvf_hgemm', dummy_x.data_ptr(), 512, 1, dummy_w.data_ptr(), 1, 512, dummy_o2.data_ptr(), 512, None, None, 0, 0, 1, 1,
                              768, 512, 512, 1.0, 0, 0, _lib.stream())
                elif kind == 'elt':       # elementwise pass over a head-sized activation [768, 512] fp32
                    dummy_x.add_(1.0)
                elif kind == 'gemv':      # 6 workgroups x ~49 us: long and narrow
                    ops.linear(dummy_q, dummy_wq, None)
                elif kind == 'gemm64':    # the same GEMM on 64-row tiles (MVF_HGEMM_TM=64 in the environment)
                    ops.linear(dummy_x, dummy_w, None)
                else:
                    ops.linear(dummy_x, dummy_w, None)
        elif a.no_head:
            model._stash.pop(0)
        elif a.fwd_only:
            with torch.no_grad():
                algo.compute_loss(wrapped, videos, seq_lens, steps, masks)
        else:
            opt.zero_grad()
            loss = algo.compute_loss(wrapped, videos, seq_lens, steps, masks)['loss']
            ops.backward(loss)
            opt.step(max_norm=cfg.OPTIMIZER.GRAD_CLIP)
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(dev))
        h_end.append(e)
        host.append(time.perf_counter() - t0)

    wrapped.prefetch(videos)
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    del marks[:], h_end[:], host[:]
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps * 1e3
    n0 = int(a.steps * 0.4)
    ref = marks[n0][0]
    bs = [ref.elapsed_time(s) for s, e in marks[n0:]]
    be = [ref.elapsed_time(e) for s, e in marks[n0:]]
    n = len(bs)
    period = (be[-1] - be[0]) / (n - 1)
    dur = sum(be[i] - bs[i] for i in range(n)) / n
    gap = sum(bs[i + 1] - be[i] for i in range(n - 1)) / (n - 1)
    print('wall %.3f ms/step (host enqueue %.3f ms/step);  backbone period %.3f ms, forward start-to-end %.3f ms, idle gap between forwards '
          '%.3f ms' % (wall, sum(host[n0:]) / len(host[n0:]) * 1e3, period, dur, gap))
    if a.prof:
        torch.cuda.synchronize()
        _lib.call('mvf_prof_enable', 1)
        for _ in range(a.prof):
            step()
        torch.cuda.synchronize()
        _lib.call('mvf_prof_enable', 0)
        G = 16
        ms, fl = (ctypes.c_double * G)(), (ctypes.c_double * G)()
        cnt, epi, nn, kk = (ctypes.c_int * G)(), (ctypes.c_int * G)(), (ctypes.c_int * G)(), (ctypes.c_int * G)()
        ng = ctypes.c_int(0)
        _lib.call('mvf_prof_collect', ms, fl, cnt, epi, nn, kk, G, ctypes.byref(ng))
        tot = 0.0
        for g in range(ng.value):
            print('   launches epi %d N %4d K %4d: %5.1f per step, avg %7.1f us, %7.3f ms per step (both lanes)' % (
                epi[g], nn[g], kk[g], cnt[g] / a.prof, ms[g] * 1e3 / max(cnt[g], 1), ms[g] / a.prof))
            tot += ms[g] / a.prof
        print('   sum over the GEMM / fused launches: %.3f ms per step = %.3f ms per lane' % (tot, tot / 2))
    if not a.no_head and not a.dummy:
        # the head of batch k (consumes the forward launched one step earlier) against the forward running beside it
        he = [ref.elapsed_time(e) for e in h_end[n0:]]
        lag = [he[i] - be[i] for i in range(min(n, len(he)))]
        print('head end minus end of the forward launched in the same step: mean %.3f ms (min %.3f, max %.3f)' %
              (sum(lag) / len(lag), min(lag), max(lag)))


if __name__ == '__main__':
    main()
