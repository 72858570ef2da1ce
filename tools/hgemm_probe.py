"""Head GEMM (fp32 MFMA, ops.linear forward + backward) timings at the row counts of the shipped configs (B=1, T=80:
480 rows -> guarded tiles) against the benchmark's 768 rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import ops
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (768, 480, 512, 1440):
    for K, N in ((387, 512), (512, 512), (256, 768), (256, 1024), (1024, 256)):
        x = torch.randn(M, K, device='cuda', requires_grad=True)
        w = torch.randn(N, K, device='cuda', requires_grad=True)
        b = torch.randn(N, device='cuda', requires_grad=True)
        gy = torch.randn(M, N, device='cuda')
        fwd = t(lambda: ops.linear(x, w, b))
        def fb():
            y = ops.linear(x, w, b); y.backward(gy)
        both = t(fb)
        print('M=%5d K=%5d N=%5d  fwd %6.1f us   fwd+bwd %6.1f us' % (M, K, N, fwd, both), flush=True)
