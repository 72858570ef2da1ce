"""Sustained time, socket power, clock and energy per launch of the ViT attention kernel's variants at one shape (GPU box):
    python tools/attn_energy.py F N H [--seconds 3] [--smi] variants...
(variants as tools/attn_bench.py: 0 product, 7 the rounds 2-5 kernel, 8 + form of the 32-query-row kernel, | 0x1000 pre-scaled q).
Is the kernel at the board's power cap (then its time is its joules, and only less WORK makes it faster), and what does each form cost?"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from video_rep_learning_amd import _lib  # noqa: E402
from energy_probe import PowerSampler  # noqa: E402

args = sys.argv[1:]
seconds, smi = 3.0, False
if '--seconds' in args:
    i = args.index('--seconds')
    seconds = float(args[i + 1])
    del args[i:i + 2]
if '--smi' in args:
    smi = True
    args.remove('--smi')
F, N, H = int(args[0]), int(args[1]), int(args[2])
variants = [int(v, 0) for v in args[3:]] or [0, 7]
D = 64 * H
qkv = torch.randn(F * N, 3 * D, device='cuda').to(torch.bfloat16)
qs = qkv.clone()
qs.view(F * N, 3, D)[:, 0] *= 0.18
out = torch.empty(F * N, D, device='cuda', dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
sampler = PowerSampler(smi=smi)
print('power source: %s; idle %.0f W' % ('rocm-smi' if (smi or not sampler.hwmon) else sampler.hwmon, sampler.read()[0]), flush=True)
sampler.start()
print('%-8s %9s %8s %7s %9s %9s %8s' % ('variant', 'us/launch', 'W', 'MHz', 'mJ/launch', 'TFLOP/s', 'pJ/FLOP'))
fl = 4.0 * F * H * N * N * 64
for v in variants:
    src = qs if v & 0x1000 else qkv
    fn = lambda: _lib.call('mvf_vit_attn_fwd', _lib.BF16, src.data_ptr(), out.data_ptr(), F, N, H, D, v, st)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    batch = max(10, int(0.05 / (e0.elapsed_time(e1) / 10 * 1e-3)))
    t_start, n, dev_ms, t0m = time.time(), 0, 0.0, None
    while time.time() - t_start < seconds:
        e0.record()
        for _ in range(batch):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if time.time() - t_start > 0.4 * seconds:
            if t0m is None:
                t0m = time.time()
            n += batch
            dev_ms += e0.elapsed_time(e1)
    t_end = time.time()
    pw = [s for s in sampler.samples if t0m is not None and t0m + 0.15 <= s[0] <= t_end]
    watts = sum(s[1] for s in pw) / max(len(pw), 1)
    mhz = sum(s[2] for s in pw) / max(len(pw), 1)
    us = dev_ms * 1e3 / max(n, 1)
    print('%-8s %9.1f %8.0f %7.0f %9.2f %9.1f %8.3f' % (hex(v) if v > 255 else v, us, watts, mhz, watts * us * 1e-3, fl / us / 1e6,
                                                      watts * us * 1e3 / fl * 1e6 / 1e3), flush=True)
    time.sleep(0.5)
sampler.stop_flag = True
