"""Where does a ViT attention variant go wrong?  python tools/attn_debug.py variant [N ...]  (GPU box)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib  # noqa: E402

variant = int(sys.argv[1])
Ns = [int(a) for a in sys.argv[2:]] or [5, 32, 33, 64, 65, 96, 128, 129, 193, 197, 257]
F, H = 2, 2
D = 64 * H
st = torch.cuda.current_stream().cuda_stream
for N in Ns:
    g = torch.Generator().manual_seed(N)
    qkv = (torch.randn(F * N, 3 * D, generator=g) * 1.5).cuda().to(torch.bfloat16)
    out = torch.full((F * N, D), 7.0, device='cuda', dtype=torch.bfloat16)
    _lib.call('mvf_vit_attn_fwd', _lib.BF16, qkv.data_ptr(), out.data_ptr(), F, N, H, D, variant, st)
    torch.cuda.synchronize()
    q, k, v = qkv.double().view(F, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).transpose(1, 2).reshape(F * N, D)
    err = (out.double() - ref).abs().view(F, N, H, 64)
    bad = (err.amax(-1) > 0.05)          # [F, N, H]
    print('N=%d variant %d: max abs err %.3e, bad rows %d of %d' % (N, variant, err.max().item(), int(bad.sum()), bad.numel()))
    if bad.any():
        for f in range(F):
            for h in range(H):
                rows = bad[f, :, h].nonzero().flatten().tolist()
                if rows:
                    print('   frame %d head %d: bad query rows %s%s' % (f, h, rows[:24], ' ...' if len(rows) > 24 else ''))
        f, n, h = [int(x) for x in bad.nonzero()[0]]
        # which keys would explain the wrong row?  solve for the weights actually applied (least squares over the keys)
        got = out[f * N + n, h * 64:(h + 1) * 64].double().cpu()
        vv = v[f, h].cpu()                               # [N, 64]
        w = torch.linalg.lstsq(vv.t(), got.unsqueeze(1)).solution.flatten() if N <= 64 else None
        p = torch.softmax(q[f, h, n] @ k[f, h].t() / 8.0, -1).cpu()
        if w is not None:
            d = (w - p).abs()
            print('   row %d: keys with wrong weight %s' % (n, (d > 0.02).nonzero().flatten().tolist()))
