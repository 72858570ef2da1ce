"""Diagnostic: frozen ViT-B/16 forward of 256 frames as ONE call vs. S concurrent calls of 256/S frames on S streams
(each GEMM's partially filled last round of tiles can then be filled by the other streams' kernels)."""
import ctypes
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib, ops  # noqa: E402


def random_sd(depth=12, dim=768, patch=16, img=224, dev='cuda'):
    g = torch.Generator(device='cpu').manual_seed(0)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.02).to(dev)
    sd = {'cls_token': r(1, 1, dim), 'pos_embed': r(1, (img // patch) ** 2 + 1, dim),
          'patch_embed.proj.weight': r(dim, 3, patch, patch), 'patch_embed.proj.bias': r(dim),
          'norm.weight': torch.ones(dim, device=dev), 'norm.bias': torch.zeros(dim, device=dev)}
    for i in range(depth):
        p = 'blocks.%d.' % i
        sd[p + 'norm1.weight'], sd[p + 'norm1.bias'] = torch.ones(dim, device=dev), torch.zeros(dim, device=dev)
        sd[p + 'norm2.weight'], sd[p + 'norm2.bias'] = torch.ones(dim, device=dev), torch.zeros(dim, device=dev)
        sd[p + 'attn.qkv.weight'], sd[p + 'attn.qkv.bias'] = r(3 * dim, dim), r(3 * dim)
        sd[p + 'attn.proj.weight'], sd[p + 'attn.proj.bias'] = r(dim, dim), r(dim)
        sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias'] = r(4 * dim, dim), r(4 * dim)
        sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'] = r(dim, 4 * dim), r(dim)
    return sd


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device('cuda')
    pk = ops.PackedViT(random_sd(), 12, 768, 12, 16, 224, [5, 7, 9, 11], 'bf16')
    x = torch.randn(F, 3, 224, 224, device=dev)
    np_ = 196
    lib = _lib.load()
    ref = None
    offsets = [int(v) for v in os.environ.get('LANE_OFFSETS_US', '0').split(',')]
    for S, off_us in [(1, 0)] + [(2, o) for o in offsets] + [(1, 0)] + [(2, o) for o in offsets]:
        fs = F // S
        streams = [torch.cuda.Stream() for _ in range(S)]
        nbytes = lib.mvf_vit_workspace_bytes(pk.code, fs, np_ + 1, 768, 16)
        wss = [torch.empty(nbytes, device=dev, dtype=torch.uint8) for _ in range(S)]
        taps = [torch.empty(F * np_, 768, device=dev, dtype=torch.bfloat16) for _ in range(4)]
        cls = torch.empty(F, 768, device=dev)
        torch.cuda.synchronize()

        def run():
            for s in range(S):
                if s > 0 and off_us > 0:
                    with torch.cuda.stream(streams[s]):
                        torch.cuda._sleep(int(off_us * 1e-6 * 2.1e9) * s)
                tab = (ctypes.c_void_p * 4)(*[t.data_ptr() + s * fs * np_ * 768 * 2 for t in taps])
                _lib.call('mvf_vit_fwd', ctypes.byref(pk.struct), pk.code, x.data_ptr() + s * fs * 3 * 224 * 224 * 4, fs,
                          tab, cls.data_ptr() + s * fs * 768 * 4, wss[s].data_ptr(), wss[s].numel(), fs, 0,
                          ctypes.c_void_p(streams[s].cuda_stream))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        if ref is None:
            ref = [t.clone() for t in taps]
        same = all(torch.equal(a, b) for a, b in zip(ref, taps))
        print('offset %4d us  streams %d x %d frames: %.3f ms  (%.1f frames/s)  bitwise same as 1 stream: %s' % (off_us, S, fs, dt * 1e3, F / dt, same),
              flush=True)


if __name__ == "__main__":
    main()
