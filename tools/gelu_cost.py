import os, sys, torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
from video_rep_learning_amd import _lib
M, n, k = 256 * 197, 3072, 768
A = torch.randn(M, k, device='cuda').to(torch.bfloat16)
W = (torch.randn(n, k, device='cuda') * 0.02).to(torch.bfloat16)
b = torch.randn(n, device='cuda')
C = torch.empty(M, n, device='cuda', dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    for epi in (0, 1):
        fn = lambda: _lib.call('mvf_gemm_tc', _lib.BF16, epi, A.data_ptr(), k, W.data_ptr(), k, b.data_ptr(), C.data_ptr(), n, None, 0, None, 0, None, None, 197, M, n, k, st)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print('fc1 shape epi %d: %.1f us' % (epi, e0.elapsed_time(e1) / 20 * 1e3))
