"""Sustained time, socket power and energy per launch of each backbone kernel at BASELINE configs[1] size (GPU box).

    python tools/energy_probe.py [--seconds 3] [--kernels qkv,proj,fc1,fc2,fc2_plain,attn,ln,ln_add,finalize,im2col]

Every kernel is launched back to back for `--seconds` (batches sized to ~50 ms, one host synchronisation per batch) while a
thread samples the socket power (hwmon power1_average / rocm-smi) and the shader clock.  At BASELINE configs[1] the training
step runs AT the board's power cap (profiles/r04/power_step.txt), so a kernel's sustained time is its share of the step's
energy: joules per launch = power x time is the number that ranks kernels, not the roofline fraction.

fc2_plain = the fc2 shape (N = 768, K = 3072) with the plain bf16 store epilogue: what fc2 would cost if its read-modify
epilogue (fp32 residual + bf16 addend in, fp32 residual + bf16(x) + row sums out) were perfectly hidden AND free -- the upper
bound of any overlap scheme (ping-pong wave rows, second accumulator set) for that kernel."""
import argparse
import glob
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib  # noqa: E402


class PowerSampler(threading.Thread):
    """socket power (W) and sclk (MHz), sampled every `dt` seconds while running"""

    def __init__(self, dt=0.1, smi=False):
        super().__init__(daemon=True)
        self.dt = dt
        self.smi = smi
        self.samples = []
        self.stop_flag = False
        # the hwmon node of THIS process's GPU: the host has eight cards, the box sees one -- match the PCI bus id
        self.hwmon = None
        try:
            bus = torch.cuda.get_device_properties(0).pci_bus_id
            dom = getattr(torch.cuda.get_device_properties(0), 'pci_domain_id', 0)
            dev_id = torch.cuda.get_device_properties(0).pci_device_id
            want = '%04x:%02x:%02x.0' % (dom, bus, dev_id)
        except Exception:
            want = None
        for p in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_average') + \
                glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input'):
            try:
                real = os.path.realpath(os.path.join(os.path.dirname(p), '..', '..'))
                if want is None or os.path.basename(real) != want:
                    continue
                int(open(p).read())
                self.hwmon = p
                break
            except (OSError, ValueError):
                pass
        self.sclk = None
        if self.hwmon:
            f = os.path.join(os.path.dirname(self.hwmon), 'freq1_input')
            if os.path.exists(f):
                self.sclk = f

    def read(self):
        if self.hwmon and not self.smi:
            try:
                w = int(open(self.hwmon).read()) * 1e-6
                mhz = int(open(self.sclk).read()) * 1e-6 if self.sclk else 0.0
                return w, mhz
            except (OSError, ValueError):
                pass
        out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout
        w, mhz = 0.0, 0.0
        for line in out.splitlines():
            if 'Power' in line and ':' in line:
                try:
                    w = float(line.rsplit(':', 1)[1])
                except ValueError:
                    pass
            if 'sclk' in line and '(' in line:
                try:
                    mhz = float(line.split('(')[1].split('Mhz')[0])
                except ValueError:
                    pass
        return w, mhz

    def run(self):
        while not self.stop_flag:
            self.samples.append((time.time(),) + self.read())
            time.sleep(self.dt)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--seconds', type=float, default=3.0)
    p.add_argument('--kernels', default='qkv,proj,fc1,fc2,fc2_plain,attn,ln,ln_add,finalize,im2col')
    p.add_argument('--frames', type=int, default=256)
    p.add_argument('--smi', action='store_true', help='sample with rocm-smi (slower, ~3 samples/s) instead of the hwmon node')
    a = p.parse_args()
    dev = 'cuda'
    F, N, D, H = a.frames, 197, 768, 12
    M = F * N
    st = torch.cuda.current_stream().cuda_stream
    bf = torch.bfloat16

    def rnd(*shape, scale=1.0, dtype=bf):
        return (torch.randn(*shape, device=dev) * scale).to(dtype)

    x = rnd(M, D, dtype=torch.float32)
    xb, h, delta = rnd(M, D), rnd(M, D), rnd(M, D)
    qkv = rnd(M, 3 * D)
    hid = rnd(M, 4 * D)
    Wqkv, Wproj, Wfc1, Wfc2 = rnd(3 * D, D, scale=0.02), rnd(D, D, scale=0.02), rnd(4 * D, D, scale=0.02), rnd(D, 4 * D, scale=0.02)
    b3, b1, b4 = torch.zeros(3 * D, device=dev), torch.zeros(D, device=dev), torch.zeros(4 * D, device=dev)
    stats = torch.empty(D // 64, M, 2, device=dev)
    mr = torch.stack([torch.randn(M, device=dev) * 0.1, 1.0 + 0.1 * torch.rand(M, device=dev)], 1).contiguous()
    c3, c4 = torch.randn(3 * D, device=dev), torch.randn(4 * D, device=dev)
    g, be = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    frames = torch.randn(F, 3, 224, 224, device=dev)
    cplain = torch.empty(M, D, device=dev, dtype=bf)
    call = _lib.call

    def k_qkv():
        call('mvf_gemm_tc_ln', _lib.BF16, 0, xb.data_ptr(), D, Wqkv.data_ptr(), D, b3.data_ptr(), qkv.data_ptr(), 3 * D, None, 0, None, 0,
             None, N, None, 0, None, mr.data_ptr(), c3.data_ptr(), M, 3 * D, D, st)

    def k_proj():
        call('mvf_gemm_tc', _lib.BF16, 0, h.data_ptr(), D, Wproj.data_ptr(), D, b1.data_ptr(), delta.data_ptr(), D, None, 0, None, 0, None,
             None, N, M, D, D, st)

    def k_fc1():
        call('mvf_gemm_tc', _lib.BF16, 1, h.data_ptr(), D, Wfc1.data_ptr(), D, b4.data_ptr(), hid.data_ptr(), 4 * D, None, 0, None, 0, None,
             None, N, M, 4 * D, D, st)

    def k_fc2():
        # the product form: resid += A W^T + b + delta, bf16(x) + row sums out (MvfGemmLn with addend2) -- through the blocks
        # entry point's own GEMM call it is mvf_gemm_tc_impl(..., &ln); exported pieces: resid2 (no xb / stats) and _ln (no
        # addend2).  Time the _ln producer form with the second addend folded in by running resid2 when --no-ln is asked.
        call('mvf_gemm_tc_ln', _lib.BF16, 2, hid.data_ptr(), 4 * D, Wfc2.data_ptr(), 4 * D, b1.data_ptr(), None, 0, x.data_ptr(), D, None, 0,
             None, N, xb.data_ptr(), D, stats.data_ptr(), None, None, M, D, 4 * D, st)

    def k_fc2_resid2():
        call('mvf_gemm_tc_resid2', hid.data_ptr(), 4 * D, Wfc2.data_ptr(), 4 * D, b1.data_ptr(), x.data_ptr(), D, delta.data_ptr(), D,
             None, 0, N, M, D, 4 * D, st)

    def k_fc2_plain():
        call('mvf_gemm_tc', _lib.BF16, 0, hid.data_ptr(), 4 * D, Wfc2.data_ptr(), 4 * D, b1.data_ptr(), cplain.data_ptr(), D, None, 0, None, 0,
             None, None, N, M, D, 4 * D, st)

    def k_attn():
        call('mvf_vit_attn_fwd', _lib.BF16, qkv.data_ptr(), h.data_ptr(), F, N, H, D, 0, st)

    def k_attn_msum():     # variant 6: row sums on the VALU (the form before round 4; variant 0 takes them on the matrix pipe)
        call('mvf_vit_attn_fwd', _lib.BF16, qkv.data_ptr(), h.data_ptr(), F, N, H, D, 6, st)

    def k_qkv_attn():      # qkv projection fused into the attention kernel (folded-LayerNorm form, as blocks 1.. run it)
        call('mvf_vit_qkv_attn_fwd', _lib.BF16, xb.data_ptr(), D, Wqkv.data_ptr(), b3.data_ptr(), c3.data_ptr(), mr.data_ptr(), None, 0, 0.0,
             h.data_ptr(), F, N, H, D, st)

    def k_ln():
        call('mvf_layernorm_fwd', _lib.BF16, x.data_ptr(), D, g.data_ptr(), be.data_ptr(), h.data_ptr(), D, M, D, 1e-6, st)

    def k_ln_add():
        call('mvf_layernorm_add_fwd', _lib.BF16, x.data_ptr(), D, delta.data_ptr(), D, g.data_ptr(), be.data_ptr(), h.data_ptr(), D, M, D,
             1e-6, st)

    def k_finalize():
        call('mvf_ln_stats_finalize', stats.data_ptr(), D // 64, mr.data_ptr(), M, D, 1e-6, st)

    def k_im2col():
        call('mvf_patchify', _lib.BF16, frames.data_ptr(), hid.data_ptr(), F, 224, 224, 16, st)

    flops = {'qkv': 2.0 * M * 3 * D * D, 'proj': 2.0 * M * D * D, 'fc1': 2.0 * M * 4 * D * D, 'fc2': 2.0 * M * 4 * D * D,
             'fc2_resid2': 2.0 * M * 4 * D * D, 'fc2_plain': 2.0 * M * 4 * D * D, 'attn': 4.0 * F * H * N * N * 64, 'attn_msum': 4.0 * F * H * N * N * 64,
             'qkv_attn': 2.0 * M * 3 * D * D + 4.0 * F * H * N * N * 64}
    mbytes = {'qkv': (M * D * 2 + M * 3 * D * 2) / 1e6, 'proj': 2 * M * D * 2 / 1e6, 'fc1': (M * D * 2 + M * 4 * D * 2) / 1e6,
              'fc2': (M * 4 * D * 2 + M * D * (4 + 4 + 2)) / 1e6, 'fc2_resid2': (M * 4 * D * 2 + M * D * (4 + 4 + 2)) / 1e6,
              'fc2_plain': (M * 4 * D * 2 + M * D * 2) / 1e6, 'attn': M * 4 * D * 2 / 1e6, 'attn_msum': M * 4 * D * 2 / 1e6, 'qkv_attn': 2 * M * D * 2 / 1e6, 'ln': M * D * 6 / 1e6,
              'ln_add': M * D * 8 / 1e6, 'finalize': M * (D // 64 + 1) * 8 / 1e6, 'im2col': (F * 3 * 224 * 224 * 4 + F * 196 * 768 * 2) / 1e6}
    table = {'qkv': k_qkv, 'proj': k_proj, 'fc1': k_fc1, 'fc2': k_fc2, 'fc2_resid2': k_fc2_resid2, 'fc2_plain': k_fc2_plain,
             'attn': k_attn, 'attn_msum': k_attn_msum, 'qkv_attn': k_qkv_attn, 'ln': k_ln, 'ln_add': k_ln_add, 'finalize': k_finalize, 'im2col': k_im2col}

    sampler = PowerSampler(smi=a.smi)
    idle = sampler.read()
    print('power source: %s; idle %.0f W' % ('rocm-smi' if (a.smi or not sampler.hwmon) else sampler.hwmon, idle[0]), flush=True)
    sampler.start()
    print('%-11s %9s %8s %7s %9s %9s %9s %8s' % ('kernel', 'us/launch', 'W', 'MHz', 'mJ/launch', 'TFLOP/s', 'GB/s alg', 'pJ/FLOP'))
    for name in a.kernels.split(','):
        fn = table[name]
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        est = e0.elapsed_time(e1) / 20 * 1e-3
        batch = max(10, int(0.05 / est))
        t_start = time.time()
        n, dev_ms, t_meas0 = 0, 0.0, None
        while time.time() - t_start < a.seconds:
            e0.record()
            for _ in range(batch):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if time.time() - t_start > 0.4 * a.seconds:      # steady state: the last 60 % of the run
                if t_meas0 is None:
                    t_meas0 = time.time()
                n += batch
                dev_ms += e0.elapsed_time(e1)
        t_end = time.time()
        pw = [s for s in sampler.samples if t_meas0 is not None and t_meas0 + 0.15 <= s[0] <= t_end]
        watts = sum(s[1] for s in pw) / max(len(pw), 1)
        mhz = sum(s[2] for s in pw) / max(len(pw), 1)
        us = dev_ms * 1e3 / max(n, 1)
        mj = watts * us * 1e-3
        fl = flops.get(name)
        print('%-11s %9.1f %8.0f %7.0f %9.2f %9s %9.0f %8s' % (name, us, watts, mhz, mj, '%.1f' % (fl / us / 1e6) if fl else '-',
                                                             mbytes[name] / us * 1e3, '%.3f' % (mj * 1e9 / fl) if fl else '-'), flush=True)
        time.sleep(0.5)
    sampler.stop_flag = True


if __name__ == '__main__':
    main()
