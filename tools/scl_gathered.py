"""SCL loss at the GATHERED size of BASELINE configs[2] (8 ranks x 4 videos x 2 views x 32 frames: M = 2 048 rows, every rank
evaluates the loss over all rows and the gradient of its own 256): time of mvf_scl_fwd / mvf_scl_bwd with row0 / rows = one
rank's slice, W = 8 emulated on one GPU, next to the local size (M = 256) -- VERDICT r03 item 8.

    python tools/scl_gathered.py [E=128]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import ops  # noqa: E402


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    T = 32
    g = torch.Generator().manual_seed(3)
    for W in (1, 2, 4, 8):
        b = 4 * W
        M = b * 2 * T
        emb = torch.nn.functional.normalize(torch.randn(M, E, generator=g), dim=1).cuda().requires_grad_(True)
        steps = torch.sort(torch.randint(0, 100, (b, 2, T), generator=g), dim=-1)[0].cuda()
        lens = torch.full((b, 2, T), 100).cuda()
        masks = torch.ones(b, 2, T).cuda()
        rows = M // W

        def fwd():
            return ops.scl_loss(emb, steps, lens, masks, T, 'batch_noself', 0.1, 10.0, row0=(W - 1) * rows, rows=rows, grad_scale=float(W))
        for _ in range(3):
            fwd().backward()
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        n = 20
        for _ in range(n):
            e[0].record()
            loss = fwd()
            e[1].record()
            loss.backward()
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1])
            tb += e[1].elapsed_time(e[2])
        print('W = %d  M = %4d rows (own %d), E = %d: forward %7.1f us, backward %7.1f us  (%.2f GFLOP pair work forward)' % (
            W, M, rows, E, tf / n * 1e3, tb / n * 1e3, 2.0 * M * M * E / 1e9), flush=True)


if __name__ == '__main__':
    main()
