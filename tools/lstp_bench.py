"""LSTP pooling over the backbone taps at BASELINE configs[1] size (256 frames x 196 tokens x 3 taps x 768 channels, bf16,
3 static queries): the one-pass kernels against the three-launch chain, forward and backward, HIP-event timed.
Algorithmic bytes per pass = the 231 MB of taps read once.  Usage: python3 tools/lstp_bench.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from video_rep_learning_amd import ops  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    F, N, D, T, nq = 256, 196, 768, 32, 3
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    taps = [torch.randn(F * N, D, generator=g).to(dev).to(torch.bfloat16) for _ in range(3)]
    big = torch.empty(512 << 20, device=dev, dtype=torch.uint8)          # evicts the taps from the Infinity Cache between runs
    gy = torch.randn(F // T, nq, T, 3 * D, generator=g).to(dev)
    tap_bytes = 3 * F * N * D * 2
    for one_pass in (False, True):
        ops.LSTP_ONE_PASS = one_pass
        tf = tb = 0.0
        for it in range(iters + 3):
            vec = (0.05 * torch.randn(nq, 3 * D, generator=g)).to(dev).requires_grad_(True)
            big.fill_(it & 255)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
            pooled, _ = ops.lstp_pool(vec, taps, F, N, T, nq, 384)
            e[1].record()
            big.fill_((it + 1) & 255)
            e[2].record()
            pooled.backward(gy)
            e[3].record()
            torch.cuda.synchronize()
            if it >= 3:
                tf += e[0].elapsed_time(e[1])
                tb += e[2].elapsed_time(e[3])
        tf, tb = tf / iters * 1e3, tb / iters * 1e3
        print('%s: forward %.1f us (%.2f TB/s on the %d MB of taps), backward %.1f us (%.2f TB/s)' % (
            'one-pass   ' if one_pass else 'three-launch', tf, tap_bytes / tf / 1e6, tap_bytes >> 20, tb, tap_bytes / tb / 1e6))


if __name__ == '__main__':
    main()
