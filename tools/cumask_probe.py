"""Probe (GPU box): does hipExtStreamCreateWithCUMask work here, and how are mask bits mapped to XCDs?
Times the qkv GEMM (persistent, grid = #CUs assumed 256) and a bandwidth kernel on masked streams."""
import ctypes
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib  # noqa: E402

hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for i in bits:
        words[i // 32] |= (1 << (i % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def bench(stream, fn, iters=10):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn(stream.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            fn(stream.cuda_stream)
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = 'cuda'
    M, n, k = 256 * 197, 2304, 768
    A = torch.randn(M, k, device=dev).to(torch.bfloat16)
    W = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.randn(n, device=dev)
    C = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    x = torch.randn(64 << 20, device=dev)
    y = torch.empty_like(x)

    def gemm(st):
        _lib.call('mvf_gemm_tc', _lib.BF16, 0, A.data_ptr(), k, W.data_ptr(), k, b.data_ptr(), C.data_ptr(), n, None, 0, None, 0,
                  None, None, 197, M, n, k, st)

    def ln(st):
        _lib.call('mvf_layernorm_fwd', _lib.BF16, x.data_ptr(), 768, b.data_ptr(), b.data_ptr(), y.data_ptr(), 768,
                  50432, 768, 1e-6, st)
    torch.cuda.synchronize()
    for ncu in (256, 240, 224, 128):
        _lib.call('mvf_gemm_tc_set_cus', ncu)
        print('unmasked default stream, grid %3d: gemm(qkv) %8.1f us' % (ncu, bench(torch.cuda.current_stream(), gemm)), flush=True)
    drop_lo = set(range(16))
    drop_st = set(range(0, 128, 8))
    drop_st2 = set(list(range(0, 8)) + list(range(128, 136)))
    for name, bits, ncu in (('all 256', range(256), 256), ('bits 0-127', range(128), 128), ('even bits', range(0, 256, 2), 128),
                            ('even bits/256', range(0, 256, 2), 256),
                            ('drop 0-15', [i for i in range(256) if i not in drop_lo], 240),
                            ('drop 0,8,..120', [i for i in range(256) if i not in drop_st], 240),
                            ('drop 0-7,128-135', [i for i in range(256) if i not in drop_st2], 240),
                            ('only 0-15', range(16), 16), ('only 0,8,..120', sorted(drop_st), 16)):
        s = masked_stream(list(bits))
        _lib.call('mvf_gemm_tc_set_cus', ncu)
        print('%-18s grid %3d: gemm(qkv) %8.1f us   layernorm %7.1f us' % (name, ncu, bench(s, gemm), bench(s, ln)), flush=True)
    _lib.call('mvf_gemm_tc_set_cus', 0)
    # placement: XCD of workgroup b under different masks (512-thread workgroups with 130 KiB LDS = one per CU)
    import collections
    for name, bits, nblk in (('all 256', range(256), 256), ('drop 0-15', [i for i in range(256) if i >= 16], 240),
                             ('bits 0-127', range(128), 128), ('only 0-15', range(16), 16)):
        s = masked_stream(list(bits))
        out = torch.full((2 * nblk,), -1, device=dev, dtype=torch.int32)
        with torch.cuda.stream(s):
            _lib.call('mvf_debug_xcc_map', out.data_ptr(), nblk, 512, 130 * 1024, s.cuda_stream)
        torch.cuda.synchronize()
        o = out.cpu().view(nblk, 2)
        xcc = o[:, 0].tolist()
        cus = collections.Counter((x, (h >> 8) & 0xf, (h >> 13) & 0x7) for x, h in o.tolist())
        print('%-12s first 24 XCDs by block id: %s' % (name, xcc[:24]))
        print('             blocks per XCD: %s   distinct (xcc,cu,se): %d   b%%8->xcc consistent: %s' % (
            sorted(collections.Counter(xcc).items()), len(cus), all(xcc[b] == xcc[b % 8] for b in range(nblk))))


if __name__ == '__main__':
    main()
