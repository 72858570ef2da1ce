"""Diagnostic: where a gemm_tc256 workgroup spends its cycles (s_memtime stamps of the DBG build; GPU box only).
Reads SHARES, not absolute run time (the stamped build forbids overlaps the product kernel has)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib  # noqa: E402


def main():
    n, k = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2304, 768)
    M = 256 * 197
    dev = 'cuda'
    st = torch.cuda.current_stream().cuda_stream
    A = torch.randn(M, k, device=dev).to(torch.bfloat16)
    W = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.randn(n, device=dev)
    C = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    nblk = ((M + 255) // 256) * ((n + 255) // 256)
    ntile = nblk
    buf = torch.zeros(nblk * 48, device=dev, dtype=torch.int64)
    kt = int(os.environ.get('KT', '-1'))     # KT=5: also stamp every workgroup barrier of K tile 5 (second tile of each workgroup)

    def fn():
        _lib.call('mvf_gemm_tc', _lib.BF16, 0, A.data_ptr(), k, W.data_ptr(), k, b.data_ptr(), C.data_ptr(), n,
                  None, 0, None, 0, None, None, 197, M, n, k, st)
    for _ in range(3):
        fn()
    _lib.call('mvf_gemm_tc_debug_stamps', buf.data_ptr())
    _lib.call('mvf_gemm_tc_debug_ktile', kt)
    # ABL: timing ablations of the stamped build (bit 0 no MFMAs, bit 1 no LDS fragment reads, bit 2 no operand DMAs)
    _lib.call('mvf_gemm_tc_debug_ablate', int(os.environ.get('ABL', '0')))
    # ROWMASK=2047: A rows read as row & 2047 -> A footprint 2048 rows (L2-resident): the K loop's feed rate without misses
    _lib.call('mvf_gemm_tc_debug_rowmask', int(os.environ.get('ROWMASK', str(0x7fffffff))))
    _lib.call('mvf_gemm_tc_select', int(os.environ.get('VARIANT', '2')))
    for _ in range(2):
        fn()
    _lib.call('mvf_gemm_tc_select', 0)
    torch.cuda.synchronize()
    _lib.call('mvf_gemm_tc_debug_stamps', None)
    _lib.call('mvf_gemm_tc_debug_ktile', -1)
    _lib.call('mvf_gemm_tc_debug_ablate', 0)
    _lib.call('mvf_gemm_tc_debug_rowmask', 0x7fffffff)
    raw = buf.cpu().numpy().astype(np.int64)
    nwg = int(os.environ.get('NWG', '256')) if int(os.environ.get('VARIANT', '2')) != 3 else nblk
    s = raw[:nblk * 16].reshape(nblk, 2, 8)
    if kt >= 0:
        k2 = raw[nwg * 16:nwg * 48].reshape(nwg, 2, 16)
        k2 = k2[k2[:, 0, 0] != 0]
        n_st = int((k2[0, 0] != 0).sum())
        d2 = np.diff(k2[:, :, :n_st], axis=2)
        print('K tile %d, barrier-to-barrier ticks (median over %d workgroups; P0 load|mfma, P1 .., P2 .., P3 ..):' % (kt, k2.shape[0]))
        for wrow in (0, 1):
            print('  wave row %d: %s   sum %d' % (wrow, ' '.join('%5d' % v for v in np.median(d2[:, wrow], axis=0)),
                                                 np.median(d2[:, wrow].sum(axis=1))))
    s = s[s[:, 0, 7] != 0]          # workgroups that ran (persistent launch: one per CU)
    for q in range(1, 7):            # unused stamp slots (fewer than 3 tiles) -> carry forward
        s[:, :, q] = np.where(s[:, :, q] == 0, s[:, :, q - 1], s[:, :, q])
    d = np.diff(s, axis=2)          # [blk, wave-row, 7 segments]
    names = ['cold prologue (14 DMA + wait)', 'tile 0: K loop', 'tile 0: epilogue', 'tile 1: K loop', 'tile 1: epilogue',
             'tile 2: K loop', 'rest (tiles 2.. + drain)']
    tot = (s[:, :, 7] - s[:, :, 0])
    print('N=%d K=%d blocks=%d; s_memtime ticks (100 MHz reference? see guide: tick = shader cycle)' % (n, k, nblk))
    for wrow in (0, 1):
        print(' wave row %d: total median %d  p10 %d  p90 %d' % (wrow, np.median(tot[:, wrow]),
                                                                   np.percentile(tot[:, wrow], 10),
                                                                   np.percentile(tot[:, wrow], 90)))
        for i, nm in enumerate(names):
            v = d[:, wrow, i]
            print('   %-40s median %7d  p10 %7d  p90 %7d  share %5.1f%%' % (nm, np.median(v), np.percentile(v, 10),
                                                                            np.percentile(v, 90),
                                                                            100.0 * v.sum() / tot[:, wrow].sum()))
    # first-round blocks vs later rounds
    span = s[:, :, 7].max() - s[:, :, 0].min()
    print(' kernel span (first start -> last end): %d ticks' % span)


if __name__ == '__main__':
    main()
