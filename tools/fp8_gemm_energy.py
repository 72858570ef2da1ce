"""Sustained time, socket power, clock and energy per launch of the MX-fp8 GEMMs at BASELINE configs[4] shapes (ViT-L/14 @ 336: 256 frames x 577
tokens, D = 1024) beside the bf16 kernel on the same shapes (GPU box):   python tools/fp8_gemm_energy.py [--seconds 3] [--frames 256]
Is the fp8 kernel at the board's power cap (then 0.29 - 0.33 of the fp8 peak is its joules), or below it (then it is its schedule)?"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from video_rep_learning_amd import _lib  # noqa: E402
from energy_probe import PowerSampler  # noqa: E402

args = sys.argv[1:]
seconds = float(args[args.index('--seconds') + 1]) if '--seconds' in args else 3.0
F = int(args[args.index('--frames') + 1]) if '--frames' in args else 256
N, D = 577, 1024
M = F * N
dev = 'cuda'
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16


def quant(x):
    rows, K = x.shape
    q = torch.empty(rows, K, device=dev, dtype=torch.uint8)
    s = torch.empty(K // 128, rows, device=dev, dtype=torch.int32)
    _lib.call('mvf_quant_mxfp8', _lib.BF16, x.data_ptr(), K, q.data_ptr(), K, s.data_ptr(), rows, K, st)
    return q, s


h = torch.randn(M, D, device=dev).to(bf)
hid = torch.randn(M, 4 * D, device=dev).to(bf)
hq, hs = quant(h)
hidq, hids = quant(hid)
W = {k: (torch.randn(n, kk, device=dev) * 0.02).to(bf) for k, (n, kk) in dict(qkv=(3 * D, D), proj=(D, D), fc1=(4 * D, D), fc2=(D, 4 * D)).items()}
Wq = {k: quant(v) for k, v in W.items()}
bias = {k: torch.zeros(v.shape[0], device=dev) for k, v in W.items()}
x = torch.randn(M, D, device=dev)
c16 = torch.empty(M, 4 * D, device=dev, dtype=bf)
c8 = torch.empty(M, 4 * D, device=dev, dtype=torch.uint8)
cs = torch.empty(4 * D // 128, M, device=dev, dtype=torch.int32)
delta = torch.randn(M, D, device=dev).to(bf)


def fp8(name, epi, A, As, Nn, K, C=None, csc=None, resid=None):
    q, s = Wq[name]
    return lambda: _lib.call('mvf_gemm_fp8', epi, A.data_ptr(), K, As.data_ptr(), q.data_ptr(), K, s.data_ptr(), bias[name].data_ptr(),
                             None if C is None else C.data_ptr(), Nn, None if csc is None else csc.data_ptr(),
                             None if resid is None else resid.data_ptr(), D, None, 0, None, N, M, Nn, K, st)


def b16(name, epi, A, Nn, K, C=None, resid=None):
    return lambda: _lib.call('mvf_gemm_tc', _lib.BF16, epi, A.data_ptr(), K, W[name].data_ptr(), K, bias[name].data_ptr(),
                             None if C is None else C.data_ptr(), Nn, None if resid is None else resid.data_ptr(), D, None, 0, None, None, N, M, Nn, K, st)


table = [
    ('qkv  fp8', fp8('qkv', 0, hq, hs, 3 * D, D, C=c16), 2.0 * M * 3 * D * D),
    ('qkv  bf16', b16('qkv', 0, h, 3 * D, D, C=c16), 2.0 * M * 3 * D * D),
    ('proj fp8 (store)', fp8('proj', 0, hq, hs, D, D, C=c16), 2.0 * M * D * D),
    ('proj bf16 (store)', b16('proj', 0, h, D, D, C=c16), 2.0 * M * D * D),
    ('fc1  fp8 (+GELU, fp8 out)', fp8('fc1', 1, hq, hs, 4 * D, D, C=c8, csc=cs), 2.0 * M * 4 * D * D),
    ('fc1  fp8 (+GELU, bf16 out)', fp8('fc1', 1, hq, hs, 4 * D, D, C=c16), 2.0 * M * 4 * D * D),
    ('fc1  bf16 (+GELU)', b16('fc1', 1, h, 4 * D, D, C=c16), 2.0 * M * 4 * D * D),
    ('fc2  fp8 (+resid)', fp8('fc2', 2, hidq, hids, D, 4 * D, resid=x), 2.0 * M * 4 * D * D),
    ('fc2  bf16 (+resid)', b16('fc2', 2, hid, D, 4 * D, resid=x), 2.0 * M * 4 * D * D),
]
sampler = PowerSampler()
print('power source: %s; idle %.0f W' % (sampler.hwmon or 'rocm-smi', sampler.read()[0]), flush=True)
sampler.start()
print('%-28s %9s %8s %7s %9s %9s %8s' % ('kernel', 'us/launch', 'W', 'MHz', 'mJ/launch', 'TFLOP/s', 'pJ/FLOP'))
for name, fn, fl in table:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    batch = max(5, int(0.05 / (e0.elapsed_time(e1) / 5 * 1e-3)))
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
    print('%-28s %9.1f %8.0f %7.0f %9.2f %9.1f %8.3f' % (name, us, watts, mhz, watts * us * 1e-3, fl / us / 1e6, watts * us / fl * 1e6), flush=True)
    time.sleep(0.5)
sampler.stop_flag = True
