"""Diagnostic (GPU box): where the time of one row-chain launch goes.  Times mvf_enc_layer_fwd (segments A + B, M = 768, D = 256,
DFF = 1024) alone on the chip with HIP events, with parts of the kernel switched off through mvf_head_chain_debug.
    python tools/chain_probe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib, ops  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    M, D, DFF = 768, 256, 1024
    g = torch.Generator(device='cpu').manual_seed(1)
    f = lambda *s: torch.randn(*s, generator=g).to(dev)
    ws = {'qkv': f(3 * D, D) * 0.05, 'o': f(D, D) * 0.05, 'f1': f(DFF, D) * 0.05, 'f2': f(D, DFF) * 0.05}
    pack = ops.HeadPack()
    W = pack.get(list(ws.items()))
    Mp = (M + 127) // 128 * 128
    o, x = f(M, D), f(M, D)
    vec = lambda n: f(n)
    bo, b1, b2, bq, g1, be1, g0, be0 = vec(D), vec(DFF), vec(D), vec(3 * D), vec(D), vec(D), vec(D), vec(D)
    e32 = lambda *s: torch.empty(*s, device=dev)
    e16 = lambda *s: torch.empty(*s, device=dev, dtype=torch.bfloat16)
    x1, m1, r1, x2, qkv, m0, r0 = e32(M, D), e32(M), e32(M), e32(M, D), e32(M, 3 * D), e32(M), e32(M)
    act, oT, h1T, aT, h0T = e16(M, DFF), e16(D, Mp), e16(D, Mp), e16(DFF, Mp), e16(D, Mp)
    a = _lib.MvfEncFwd()
    a.M, a.D, a.DFF, a.Mp, a.ln_eps = M, D, DFF, Mp, 1e-5
    p = _lib.ptr
    a.o, a.x_in, a.wo, a.w1, a.w2 = p(o), p(x), W['o'][0], W['f1'][0], W['f2'][0]
    a.bo, a.b1, a.b2, a.ln1_g, a.ln1_b = p(bo), p(b1), p(b2), p(g1), p(be1)
    a.drop_attn.p, a.drop_attn.seed, a.drop_ffn.p, a.drop_ffn.seed = 0.1, 3, 0.1, 4
    a.x1, a.mean1, a.rstd1, a.a, a.x2, a.oT, a.h1T, a.aT = p(x1), p(m1), p(r1), p(act), p(x2), p(oT), p(h1T), p(aT)
    a.wqkv, a.bqkv, a.ln0_g, a.ln0_b, a.qkv, a.mean0, a.rstd0, a.h0T = W['qkv'][0], p(bq), p(g0), p(be0), p(qkv), p(m0), p(r0), p(h0T)
    st = torch.cuda.current_stream().cuda_stream

    def run(bits, n=200):
        _lib.call('mvf_head_chain_debug', bits)
        for _ in range(20):
            _lib.call('mvf_enc_layer_fwd', ctypes.byref(a), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            _lib.call('mvf_enc_layer_fwd', ctypes.byref(a), st)
        e1.record()
        torch.cuda.synchronize()
        _lib.call('mvf_head_chain_debug', 0)
        return e0.elapsed_time(e1) / n * 1e3

    stamps = torch.zeros(16, device=dev, dtype=torch.int64)
    _lib.call('mvf_head_chain_debug_stamps', stamps.data_ptr())
    for _ in range(3):
        _lib.call('mvf_enc_layer_fwd', ctypes.byref(a), st)
    torch.cuda.synchronize()
    _lib.call('mvf_head_chain_debug_stamps', None)
    t = stamps.cpu().tolist()
    names = ['start', 'o rows loaded', 'oT saved', 'out-proj GEMM', 'LN1', 'h1T saved', 'fc1 GEMM', 'a / aT saved', 'fc2 GEMM', 'LN0', 'h0T saved',
             'qkv GEMM']
    print('stage stamps of workgroup 0, wave 0 (us since kernel start; s_memrealtime, 10 ns ticks):')
    for i in range(1, 12):
        print('   %-16s +%6.2f us   (at %6.2f)' % (names[i], (t[i] - t[i - 1]) / 100.0, (t[i] - t[0]) / 100.0))
    # is the weight stream bound by cold L2s (one fill per XCD and launch) or by what one CU can keep in flight?
    nb = pack.buf.numel() * 2

    def run_pair(warm, n=200):
        for _ in range(10):
            if warm:
                _lib.call('mvf_head_l2_warm', pack.buf.data_ptr(), nb, st)
            _lib.call('mvf_enc_layer_fwd', ctypes.byref(a), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            if warm:
                _lib.call('mvf_head_l2_warm', pack.buf.data_ptr(), nb, st)
            _lib.call('mvf_enc_layer_fwd', ctypes.byref(a), st)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        _lib.call('mvf_head_l2_warm', pack.buf.data_ptr(), nb, st)
    e1.record()
    torch.cuda.synchronize()
    tw = e0.elapsed_time(e1) / 200 * 1e3
    print('L2 warm kernel alone (%.1f MB) %.1f us;  warm + chain launch %.1f us;  chain launch alone %.1f us' %
          (nb / 1e6, tw, run_pair(True), run_pair(False)))
    for bits, what in ((0, 'product'), (1, 'no transposed saves'), (4, 'no row save of a'), (5, 'no saves'), (2, 'k loops cut to one round'),
                       (7, 'no saves, k loops cut')):
        print('enc_layer_fwd A+B  %-28s %7.1f us' % (what, run(bits)))


if __name__ == '__main__':
    main()
