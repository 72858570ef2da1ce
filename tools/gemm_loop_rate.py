"""K-loop rate of the product GEMM kernel (GPU box): one round of 256 tiles (M = N = 4096), K long enough that prologue and
epilogue vanish, random and all-zero operands -- the number tools/probes/wave4_probe (four 512-register waves) is compared with.

    python tools/gemm_loop_rate.py [--k 4096,16384] [--variant 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--k', default='768,4096,16384')
    p.add_argument('--mn', type=int, default=4096)
    p.add_argument('--variant', type=int, default=5)
    p.add_argument('--iters', type=int, default=10)
    a = p.parse_args()
    st = torch.cuda.current_stream().cuda_stream
    _lib.call('mvf_gemm_tc_select', a.variant)
    M = N = a.mn
    for K in [int(x) for x in a.k.split(',')]:
        for zeros in (False, True):
            A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
            W = (torch.randn(N, K, device='cuda') * 0.02).to(torch.bfloat16)
            if zeros:
                A.zero_(); W.zero_()
            b = torch.zeros(N, device='cuda')
            C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)

            def run():
                _lib.call('mvf_gemm_tc', _lib.BF16, 0, A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), C.data_ptr(), N, None, 0,
                          None, 0, None, None, 197, M, N, K, st)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            print('M=N=%d K=%6d %-7s %8.1f us  %7.1f TFLOP/s' % (M, K, 'zeros' if zeros else 'random', us, 2.0 * M * N * K / us / 1e6),
                  flush=True)
    _lib.call('mvf_gemm_tc_select', 0)


if __name__ == '__main__':
    main()
