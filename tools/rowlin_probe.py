"""Diagnostic (GPU box): where the time of the row-chain Linear launches goes (csrc/head_rowlin.hip).  Runs the FC stack of configs[1]
(256 + 3 one-hot -> 512 -> BN -> ReLU -> 512 -> BN -> ReLU -> 256 + table) forward and backward through ops.rowlin_chain, prints the
stage stamps of workgroup 0 for every launch and the back-to-back time per launch.
    python tools/rowlin_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib, ops  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    M = 768
    g = torch.Generator(device='cpu').manual_seed(1)
    f = lambda *s: torch.randn(*s, generator=g).to(dev).requires_grad_(True)
    z = lambda n: torch.zeros(n, device=dev)
    w0, b0, g0, be0 = f(512, 259), f(512), f(512), f(512)
    w1, b1, g1, be1 = f(512, 512), f(512), f(512), f(512)
    w2, b2 = f(256, 512), f(256)
    table = torch.randn(32, 256, generator=g).to(dev)
    rm0, rv0, rm1, rv1 = z(512), z(512) + 1, z(512), z(512) + 1
    stages = [ops.RowLinStage(0, 1, onehot=(3, 32), drop_in=(0.1, 5, 0), bn_out=(rm0, rv0, 0.1)),
              ops.RowLinStage(4, 5, bn_in=(2, 3, 1e-5, True), bn_out=(rm1, rv1, 0.1)),
              ops.RowLinStage(8, 9, bn_in=(6, 7, 1e-5, True), table=(table, 32), drop_out=(0.1, 6, 0))]
    params = [w0, b0, g0, be0, w1, b1, g1, be1, w2, b2]
    with torch.no_grad():
        for w in (w0, w1, w2):
            w.mul_(0.05)
    pack = ops.HeadPack()
    x = f(M, 256)

    def step():
        y = ops.rowlin_chain(x, stages, params, True, pack)
        y.backward(torch.ones_like(y))

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    stamps = torch.zeros(8, 16, device=dev, dtype=torch.int64)
    _lib.call('mvf_rowlin_debug_stamps', stamps.data_ptr(), 8)
    step()
    torch.cuda.synchronize()
    _lib.call('mvf_rowlin_debug_stamps', None, 0)
    t = stamps.cpu().tolist()
    tags = ['fwd s0 259->512 +stats', 'fwd s1 bn 512->512 +stats', 'fwd s2 bn 512->256 +table', 'bwd s2', 'bwd s1', 'bwd s0']
    for li, tag in enumerate(tags):
        r = t[li]
        print('%s  (us since kernel start of workgroup 0; 10 ns ticks)' % tag)
        prev = r[0]
        for i in range(1, 12):
            if r[i] == 0:
                continue
            print('   stamp %2d  +%6.2f us   (at %6.2f)' % (i, (r[i] - prev) / 100.0, (r[i] - r[0]) / 100.0))
            prev = r[i]
    # back-to-back times per launch (rocprof-free): forward only, then forward + backward
    def timed(fn, n=100):
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    def fwd_only():
        with torch.no_grad():
            ops.rowlin_chain(x, stages, params, True, pack)
    print('forward (3 launches) %.1f us;  forward + backward (3 + 3 + 1 launches) %.1f us' % (timed(fwd_only), timed(step)))
    print('forward stamps: 0 start, 1 bn scale/shift, 2 X rows -> panel, 3 xT saved, 4 GEMM + epilogue, 5 (l2norm) Y stored, 6 column stats, 7 ticket')
    print('backward stamps: 0 start, 1 nb constants, 2 dY rows -> panel (+ l2norm bwd), 3 g -> bf16, 4 gT saved, 5 GEMM, 6 bn constants, 7 pro\' + dX stored, '
          '8 s1/s2 partials, 9 ticket')


if __name__ == '__main__':
    main()
