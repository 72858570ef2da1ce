"""ViT attention kernel timing driver (GPU box): python tools/attn_bench.py [frames] [iters]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N = int(sys.argv[3]) if len(sys.argv) > 3 else 197
D, H = 768, 12
qkv = (torch.randn(F * N, 3 * D, device='cuda') * 1.0).to(torch.bfloat16)
out = torch.empty(F * N, D, device='cuda', dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
for variant in ((0, 2, 1) if N == 197 else (0, 4, 2)):
    fn = lambda: _lib.call('mvf_vit_attn_fwd', _lib.BF16, qkv.data_ptr(), out.data_ptr(), F, N, H, D, variant, st)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    print(('vit_attn bf16 v%d F=%d N=' + str(N) + ': %.1f us  %.1f TFLOP/s  %.0f GB/s') % (variant, F, t * 1e6, 4.0 * F * H * N * N * 64 / t / 1e12,
                                                                   F * N * 4 * D * 2 / t / 1e9))
