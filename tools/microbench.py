"""Kernel-level timings on the GPU box (not part of the judged bench): GEMM shapes of ViT-B/16 at F frames,
attention, LayerNorm, whole backbone at several frame-chunk sizes.  Prints TFLOP/s and GB/s."""
import sys
import os
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib, ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vit_streams_probe import random_sd  # noqa: E402   (seeded random ViT-B/16 weights; oracle/ is test infrastructure)

DEV = 'cuda'


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def S():
    return torch.cuda.current_stream().cuda_stream


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    N, D, H = 197, 768, 12
    M = F * N
    print('device', torch.cuda.get_device_name(0), 'F', F, 'M', M)
    for dtype in ('bf16', 'f32'):
        code, tdt = ops._dt(dtype)
        if dtype == 'f32' and F > 64:
            Mx = 64 * N
        else:
            Mx = M
        for (n, k, epi, name) in ((3 * D, D, 0, 'qkv'), (D, D, 2, 'proj'), (4 * D, D, 1, 'fc1'), (D, 4 * D, 2, 'fc2')):
            A = torch.randn(Mx, k, device=DEV).to(tdt)
            W = (torch.randn(n, k, device=DEV) * 0.02).to(tdt)
            b = torch.randn(n, device=DEV)
            Cc = torch.empty(Mx, n, device=DEV, dtype=tdt)
            R = torch.zeros(Mx, n, device=DEV)
            fn = lambda: _lib.call('mvf_gemm_tc', code, epi, A.data_ptr(), k, W.data_ptr(), k, b.data_ptr(), Cc.data_ptr(),
                                   n, R.data_ptr(), n, None, 0, None, None, N, Mx, n, k, S())
            for variant in ((1, 2) if dtype == 'bf16' else (1,)):
                _lib.call('mvf_gemm_tc_select', variant)
                t = timeit(fn)
                _lib.call('mvf_gemm_tc_select', 0)
                print('gemm_tc%s %-5s %-4s M=%6d N=%4d K=%4d  %8.1f us  %7.1f TFLOP/s' % (
                    '256' if variant == 2 else '128', dtype, name, Mx, n, k, t * 1e6, 2.0 * Mx * n * k / t / 1e12))
        qkv = torch.randn(Mx, 3 * D, device=DEV).to(tdt)
        out = torch.empty(Mx, D, device=DEV, dtype=tdt)
        for variant in ((0, 1) if dtype == 'bf16' else (0,)):
            t = timeit(lambda: _lib.call('mvf_vit_attn_fwd', code, qkv.data_ptr(), out.data_ptr(), Mx // N, N, H, D,
                                         variant, S()))
            fl = 4.0 * (Mx // N) * H * N * N * 64
            by = Mx * 4 * D * qkv.element_size()
            print('vit_attn %-5s v%d  %8.1f us  %6.1f TFLOP/s  %6.0f GB/s' % (dtype, variant, t * 1e6, fl / t / 1e12,
                                                                             by / t / 1e9))
        x = torch.randn(Mx, D, device=DEV)
        y = torch.empty(Mx, D, device=DEV, dtype=tdt)
        g = torch.ones(D, device=DEV)
        t = timeit(lambda: _lib.call('mvf_layernorm_fwd', code, x.data_ptr(), D, g.data_ptr(), g.data_ptr(), y.data_ptr(),
                                     D, Mx, D, 1e-6, S()))
        print('layernorm %-5s %8.1f us  %6.0f GB/s' % (dtype, t * 1e6, Mx * D * (4 + y.element_size()) / t / 1e9))
    # whole backbone
    w = random_sd(dev=DEV)
    frames = torch.randn(F, 3, 224, 224, device=DEV)
    pk = ops.PackedViT(w, 12, 768, 12, 16, 224, (3, 7, 11), 'bf16')
    for chunk in (0, 128, 64, 32, 16):
        for variant in (0, 1):
            t = timeit(lambda: ops.vit_forward(frames, pk, frames_per_chunk=chunk, attn_variant=variant), iters=5, warm=2)
            print('vit_fwd bf16 F=%d chunk=%3d attn_v%d: %8.2f ms  %7.1f TFLOP/s (35.13 GF/frame)' %
                  (F, chunk, variant, t * 1e3, 35.13e9 * F / t / 1e12))


if __name__ == '__main__':
    main()
