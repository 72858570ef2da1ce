"""The two HBM-bound backbone kernels at BASELINE configs[1] size (256 frames x 197 tokens x 768), 4 launches each -- for
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (tools/pmc_hbm_summary.py) and plain timing."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib
F, N, D, H = 256, 197, 768, 12
st = torch.cuda.current_stream().cuda_stream
qkv = torch.randn(F * N, 3 * D, device='cuda').to(torch.bfloat16)
out = torch.empty(F * N, D, device='cuda', dtype=torch.bfloat16)
x = torch.randn(F * N, D, device='cuda')
g, b = torch.ones(D, device='cuda'), torch.zeros(D, device='cuda')
for _ in range(4):
    _lib.call('mvf_vit_attn_fwd', _lib.BF16, qkv.data_ptr(), out.data_ptr(), F, N, H, D, 0, st)
for _ in range(4):
    _lib.call('mvf_layernorm_fwd', _lib.BF16, x.data_ptr(), D, g.data_ptr(), b.data_ptr(), out.data_ptr(), D, F * N, D, 1e-6, st)
torch.cuda.synchronize()
print('done')
