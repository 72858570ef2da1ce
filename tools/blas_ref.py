"""Reference point only (never on the product path): what the vendor BLAS behind torch.matmul reaches on the four ViT-B/16
GEMM shapes, bf16, no epilogue."""
import torch
M = 256 * 197
for name, n, k in (('qkv', 2304, 768), ('proj', 768, 768), ('fc1', 3072, 768), ('fc2', 768, 3072)):
    a = torch.randn(M, k, device='cuda').to(torch.bfloat16)
    w = (torch.randn(n, k, device='cuda') * 0.02).to(torch.bfloat16)
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c = a @ w.t()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print('torch.matmul %-5s M=%d N=%d K=%d  %7.1f us  %7.1f TFLOP/s' % (name, M, n, k, t * 1e6, 2.0 * M * n * k / t / 1e12))
