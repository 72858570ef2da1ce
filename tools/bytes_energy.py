"""What does a byte cost?  Socket power and sustained rate of plain streaming passes (GPU box; ATen kernels: this is a probe,
not the product) over buffers that fit the 256 MiB Infinity Cache and buffers that do not:

    python tools/bytes_energy.py [--seconds 3]

copy  y = x * c      (read n + write n bytes)        read  x.sum()        write  y.fill_(c)
At BASELINE configs[1] the step runs at the power cap (profiles/r04/power_step.txt), so (socket W - base) x seconds / bytes is
what a byte of HBM (or Infinity Cache) traffic costs the step."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from energy_probe import PowerSampler  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--seconds', type=float, default=3.0)
    p.add_argument('--sizes', default='16,48,96,192,512,2048')     # MB per buffer
    a = p.parse_args()
    sampler = PowerSampler(smi=True)
    print('idle %.0f W' % sampler.read()[0], flush=True)
    sampler.start()
    print('%-6s %8s %10s %8s %7s %8s %10s' % ('op', 'MB/buf', 'us/launch', 'GB/s', 'W', 'MHz', 'pJ/B gross'))
    for mb in [int(v) for v in a.sizes.split(',')]:
        n = mb * 1000 * 1000 // 4
        x = torch.randn(n, device='cuda')
        y = torch.empty_like(x)
        for op, fn, nbytes in (('copy', lambda: torch.mul(x, 1.0001, out=y), 8 * n), ('read', lambda: x.sum(), 4 * n),
                               ('write', lambda: y.fill_(1.5), 4 * n)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.time()
            cnt, ms, t_meas = 0, 0.0, None
            batch = max(20, int(0.05 / max(nbytes / 4e12, 2e-5)))
            while time.time() - t0 < a.seconds:
                e0.record()
                for _ in range(batch):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if time.time() - t0 > 0.4 * a.seconds:
                    if t_meas is None:
                        t_meas = time.time()
                    cnt += batch
                    ms += e0.elapsed_time(e1)
            t1 = time.time()
            pw = [s for s in sampler.samples if t_meas is not None and t_meas + 0.2 <= s[0] <= t1]
            w = sum(s[1] for s in pw) / max(len(pw), 1)
            mhz = sum(s[2] for s in pw) / max(len(pw), 1)
            us = ms * 1e3 / max(cnt, 1)
            print('%-6s %8d %10.1f %8.0f %7.0f %8.0f %10.1f' % (op, mb, us, nbytes / us / 1e3, w, mhz, w * us * 1e-6 / nbytes * 1e12), flush=True)
            time.sleep(0.3)
        del x, y
    sampler.stop_flag = True


if __name__ == '__main__':
    main()
