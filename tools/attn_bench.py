"""ViT attention kernel timing driver (GPU box): python tools/attn_bench.py [frames] [iters] [N] [heads] [variants...]
Shapes of interest: 256 20 197 12 (configs[1]) | 80 20 785 12 (penn_mvf.yml as shipped, ViT-B/8) | 256 20 577 16 (configs[4], ViT-L/14 @ 336)
variants: 0 product path, 7 the 16-query-tile streamed kernel of rounds 2-5, 8..11 forms of the 32-query-row kernel, 16 + form + 4 * waves.
Checks every variant's output against variant 7 / fp64 on a sample of rows, then times it (bursts of `iters` launches; `--sustain S`: S
seconds of back-to-back launches before the timed burst, the state the training step is in)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith('--')]
sustain = 0.0
for i, a in enumerate(sys.argv):
    if a == '--sustain':
        sustain = float(sys.argv[i + 1])
        argv.remove(sys.argv[i + 1])
F = int(argv[0]) if len(argv) > 0 else 256
iters = int(argv[1]) if len(argv) > 1 else 20
N = int(argv[2]) if len(argv) > 2 else 197
H = int(argv[3]) if len(argv) > 3 else 12
variants = [int(v) for v in argv[4:]] or ([0, 2, 1] if N == 197 else [7, 0, 9, 10, 11])
D = 64 * H
qkv = (torch.randn(F * N, 3 * D, device='cuda') * 1.0).to(torch.bfloat16)
if any(v & 0x1000 for v in variants):   # pre-scaled q: the magnitude a real q has after the factor log2(e) / 8
    qkv.view(F * N, 3, D)[:, 0] *= 0.18
out = torch.empty(F * N, D, device='cuda', dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream


def ref_rows(f, h, prescaled=False):
    q, k, v = qkv[f * N:(f + 1) * N].double().view(N, 3, H, 64)[:, :, h].unbind(1)
    # variant | 0x1000: q carries log2(e) / 8 already -- the kernel computes softmax2(q k^T) = softmax(ln 2 . q k^T)
    return torch.softmax(q @ k.t() * (0.6931471805599453 if prescaled else 0.125), -1) @ v


for variant in variants:
    fn = lambda: _lib.call('mvf_vit_attn_fwd', _lib.BF16, qkv.data_ptr(), out.data_ptr(), F, N, H, D, variant, st)
    out.fill_(7.0)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    err = 0.0
    for f, h in ((0, 0), (F - 1, H - 1), (F // 2, H // 3)):
        got = out[f * N:(f + 1) * N, h * 64:(h + 1) * 64].double()
        rr = ref_rows(f, h, bool(variant & 0x1000))
        err = max(err, ((got - rr).abs().max() / rr.abs().max()).item())
    if sustain > 0:
        t0 = time.time()
        while time.time() - t0 < sustain:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    print('vit_attn bf16 v%-5d F=%d N=%d H=%d: %8.1f us  %7.1f TFLOP/s  %6.0f GB/s   max err vs fp64 %.2e' % (
        variant, F, N, H, t * 1e6, 4.0 * F * H * N * N * 64 / t / 1e12, F * N * 4 * D * 2 / t / 1e9, err), flush=True)
