"""GEMM-only timing / profiling driver for the backbone GEMM kernels (GPU box; not part of the judged bench).

    python tools/gemm_bench.py [--variant 0|1|2] [--shapes qkv,proj,fc1,fc2] [--frames 256] [--iters 20] [--ln] [--rounds 3]

--ln: the same shapes with the LN-fold epilogue extras (mvf_gemm_tc_ln: consumer side for qkv / fc1, producer side for proj /
fc2), interleaved with the plain form in one process.  --variant 0 is the automatic choice of mvf_gemm_tc.

Run it under `rocprofv3 --kernel-trace --stats` or `rocprofv3 --pmc ...` to get per-kernel durations / counters for
exactly the ViT-B/16 shapes of BASELINE configs[1] (M = frames * 197)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib  # noqa: E402

# the product's forms (bf16, deferred attention-branch residual): proj = plain bf16 store, fc2 = residual epilogue with the bf16
# second addend; 'proj_rm' / 'fc2_rm' = the read-modify forms without the deferral (fp32 / LayerScale / MVF_PROJ_DEFER=0)
SHAPES = {'qkv': (2304, 768, 0), 'proj': (768, 768, 0), 'fc1': (3072, 768, 1), 'fc2': (768, 3072, 2), 'proj_rm': (768, 768, 2),
          'fc2_rm': (768, 3072, 2)}


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--variant', type=int, default=2)      # 2 tile rows per launch, 4 = 224-row tiles, 5 = 256-row tiles, 1 = 128x128 kernel
    p.add_argument('--shapes', default='qkv,proj,fc1,fc2')
    p.add_argument('--frames', type=int, default=256)
    p.add_argument('--iters', type=int, default=20)
    p.add_argument('--warm', type=int, default=3)
    p.add_argument('--ln', nargs='?', const='both', default=None, choices=['both', 'xb', 'stats'])
    p.add_argument('--rounds', type=int, default=1)
    # what happens to the A operand right before every timed launch: nothing (A was evicted by the previous launch's output),
    # a kernel READING it, or a kernel WRITING it (the real pipeline: the producer ran just before) -- is A cache-resident?
    p.add_argument('--pre', default='none', choices=['none', 'read', 'write'])
    # all-zero operands: the matrix pipe toggles no bits, the chip holds a higher clock -- how much of the gap to the nominal peak
    # is the clock under load rather than the schedule?
    p.add_argument('--zeros', action='store_true')
    a = p.parse_args()
    dev = 'cuda'
    M = a.frames * 197
    st = torch.cuda.current_stream().cuda_stream
    _lib.call('mvf_gemm_tc_select', a.variant)
    for name in a.shapes.split(','):
        n, k, epi = SHAPES[name]
        A = torch.randn(M, k, device=dev).to(torch.bfloat16)
        W = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
        if a.zeros:
            A.zero_()
            W.zero_()
        b = torch.randn(n, device=dev)
        C = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
        R = torch.zeros(M, n, device=dev)

        xb = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
        stats = torch.empty(max(n // 64, 1), M, 2, device=dev)
        mr = torch.stack([torch.randn(M, device=dev) * 0.1, 1.0 + 0.1 * torch.rand(M, device=dev)], 1).contiguous()
        c = torch.randn(n, device=dev)

        delta = torch.randn(M, n, device=dev).to(torch.bfloat16) if name == 'fc2' else None

        def plain():
            if name == 'fc2':
                _lib.call('mvf_gemm_tc_resid2', A.data_ptr(), k, W.data_ptr(), k, b.data_ptr(), R.data_ptr(), n, delta.data_ptr(), n,
                          None, 0, 197, M, n, k, st)
                return
            _lib.call('mvf_gemm_tc', _lib.BF16, epi, A.data_ptr(), k, W.data_ptr(), k, b.data_ptr(), C.data_ptr(), n,
                      R.data_ptr(), n, None, 0, None, None, 197, M, n, k, st)

        def fold():
            if epi == 2:
                _lib.call('mvf_gemm_tc_ln', _lib.BF16, epi, A.data_ptr(), k, W.data_ptr(), k, b.data_ptr(), None, 0, R.data_ptr(),
                          n, None, 0, None, 197, None if a.ln == 'stats' else xb.data_ptr(), n,
                          None if a.ln == 'xb' else stats.data_ptr(), None, None, M, n, k, st)
            else:
                _lib.call('mvf_gemm_tc_ln', _lib.BF16, epi, A.data_ptr(), k, W.data_ptr(), k, b.data_ptr(), C.data_ptr(), n, None,
                          0, None, 0, None, 197, None, 0, None, mr.data_ptr(), c.data_ptr(), M, n, k, st)

        A2 = A.clone()

        def timed(fn):
            for _ in range(a.warm):
                fn()
            torch.cuda.synchronize()
            if a.pre != 'none':
                tot = 0.0
                for _ in range(a.iters):
                    if a.pre == 'read':
                        A.view(torch.int16).max()
                    else:
                        A.copy_(A2)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn()
                    e1.record()
                    torch.cuda.synchronize()
                    tot += e0.elapsed_time(e1)
                return tot / a.iters * 1e-3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / a.iters * 1e-3
        for rnd in range(a.rounds):
            for tag, fn in (('plain', plain), ('ln', fold)) if a.ln else (('plain', plain),):
                t = timed(fn)
                print('variant %d %-4s %-5s M=%d N=%d K=%d  %8.1f us  %7.1f TFLOP/s' % (a.variant, name, tag, M, n, k, t * 1e6,
                                                                                      2.0 * M * n * k / t / 1e12), flush=True)
    _lib.call('mvf_gemm_tc_select', 0)


if __name__ == '__main__':
    main()
