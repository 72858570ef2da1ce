"""Frozen ViT-B/16 backbone forward of 256 frames, sustained for seconds per setting: lanes (concurrent slices on their own
streams) x frames per chunk (depth-first: the whole backbone on one chunk, then the next).  Smaller chunks keep a chunk's
activations (x, qkv, hid, ...: 3.6 MB per frame) inside the 256 MiB Infinity Cache between producer and consumer kernels --
does that buy anything at the power cap, against the GEMMs' poorer fill of the chip?

    python tools/chunk_probe.py [--seconds 3] [--settings 2:0,1:0,2:64,2:32,4:32,4:0]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import ops  # noqa: E402
from vit_streams_probe import random_sd  # noqa: E402
from energy_probe import PowerSampler  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--seconds', type=float, default=3.0)
    p.add_argument('--settings', default='2:0,1:0,2:64,2:32,4:32,4:0,1:64,1:32,2:0')
    p.add_argument('--frames', type=int, default=256)
    a = p.parse_args()
    dev = torch.device('cuda')
    F = a.frames
    pk = ops.PackedViT(random_sd(), 12, 768, 12, 16, 224, [3, 7, 11], 'bf16')
    x = torch.randn(F, 3, 224, 224, device=dev)
    sampler = PowerSampler(smi=True)
    sampler.start()
    ref = None
    for s in a.settings.split(','):
        lanes, fpc = [int(v) for v in s.split(':')]
        run = lambda: ops.vit_forward(x, pk, frames_per_chunk=fpc, want_cls=False, lanes=lanes)
        for _ in range(2):
            taps, _ = run()
        torch.cuda.synchronize()
        if ref is None:
            ref = [t.clone() for t in taps]
        same = all(torch.equal(r, t) for r, t in zip(ref, taps))
        t0 = time.time()
        cnt, t_meas, t_acc = 0, None, 0.0
        while time.time() - t0 < a.seconds:
            tb = time.perf_counter()
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            if time.time() - t0 > 0.4 * a.seconds:
                if t_meas is None:
                    t_meas = time.time()
                cnt += 5
                t_acc += time.perf_counter() - tb
        t1 = time.time()
        pw = [q for q in sampler.samples if t_meas is not None and t_meas + 0.2 <= q[0] <= t1]
        w = sum(q[1] for q in pw) / max(len(pw), 1)
        mhz = sum(q[2] for q in pw) / max(len(pw), 1)
        print('lanes %d  frames/chunk %3s : %7.3f ms per forward  %5.0f W %5.0f MHz  bitwise same: %s' % (
            lanes, fpc or 'all', t_acc / max(cnt, 1) * 1e3, w, mhz, same), flush=True)
    sampler.stop_flag = True


if __name__ == '__main__':
    main()
