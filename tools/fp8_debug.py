"""Structured probes of mvf_gemm_fp8 (GPU box): which of {operand bytes, A scales, W scales, k placement} is wrong when the
fp8 GEMM test fails.  Every case has an exact closed-form answer."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib  # noqa: E402

DEV = 'cuda'
ONE = 0x38     # 1.0 in e4m3
TWO = 0x40     # 2.0


def run(Aq, As, Wq, Ws, M, N, K):
    C = torch.full((M, N), -7.0, device=DEV, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call('mvf_gemm_fp8', 0, Aq.data_ptr(), K, As.data_ptr(), Wq.data_ptr(), K, Ws.data_ptr(), None, C.data_ptr(), N, None, None, 0,
              None, 0, None, 197, M, N, K, st)
    torch.cuda.synchronize()
    return C.float().cpu()


def scales(rows, K, byte_fn):
    """[K/128][rows] dwords, block b of K tile kt in byte b; byte_fn(row, block) -> int tensor [rows, K/32]"""
    r = torch.arange(rows).view(rows, 1).expand(rows, K // 32)
    b = torch.arange(K // 32).view(1, K // 32).expand(rows, K // 32)
    by = byte_fn(r, b).to(torch.int64) & 0xff
    by = by.view(rows, K // 128, 4)
    d = by[..., 0] | (by[..., 1] << 8) | (by[..., 2] << 16) | (by[..., 3] << 24)
    d = d.t().contiguous()                          # [K/128][rows]
    return (d - ((d >> 31) << 32)).to(torch.int32).to(DEV)


def main():
    M, N, K = 256, 256, 256
    ones = lambda r, k: torch.full((r, k), ONE, dtype=torch.uint8, device=DEV)   # noqa: E731
    unit = lambda r, b: torch.full_like(r, 127)    # noqa: E731
    print('1. all ones, unit scales: expect %d everywhere' % K)
    C = run(ones(M, K), scales(M, K, unit), ones(N, K), scales(N, K, unit), M, N, K)
    print('   min %.1f max %.1f  C[0,:4] %s  C[:4,0] %s' % (C.min(), C.max(), C[0, :4].tolist(), C[:4, 0].tolist()))
    print('2. A scale byte = 127 + (m %% 4): expect K * 2^(m %% 4) in row m')
    C = run(ones(M, K), scales(M, K, lambda r, b: 127 + (r % 4)), ones(N, K), scales(N, K, unit), M, N, K)
    print('   C[:8,0] / K = %s   (cols equal: %s)' % ((C[:8, 0] / K).tolist(), bool((C == C[:, :1]).all())))
    print('   C[16:24,0] / K = %s ; C[128:132,0] / K = %s' % ((C[16:24, 0] / K).tolist(), (C[128:132, 0] / K).tolist()))
    print('3. W scale byte = 127 + (n %% 4): expect K * 2^(n %% 4) in column n')
    C = run(ones(M, K), scales(M, K, unit), ones(N, K), scales(N, K, lambda r, b: 127 + (r % 4)), M, N, K)
    print('   C[0,:8] / K = %s   (rows equal: %s)' % ((C[0, :8] / K).tolist(), bool((C == C[:1, :]).all())))
    print('4. A scale byte = 127 + block index (0..%d): expect 32 * sum_b 2^b = %d' % (K // 32 - 1, 32 * (2 ** (K // 32) - 1)))
    C = run(ones(M, K), scales(M, K, lambda r, b: 127 + b), ones(N, K), scales(N, K, unit), M, N, K)
    print('   min %.1f max %.1f' % (C.min(), C.max()))
    print('5. A = 2.0 in k block 1 only (bytes 32..63), 1.0 elsewhere; W scale 2^1 in block 1 only: expect %d' % (K - 32 + 32 * 4))
    A = ones(M, K)
    A[:, 32:64] = TWO
    C = run(A, scales(M, K, unit), ones(N, K), scales(N, K, lambda r, b: 127 + (b == 1).long()), M, N, K)
    print('   min %.1f max %.1f' % (C.min(), C.max()))
    print('5b. A = 2.0 in block ba, W scale 2^1 in block bw: which (ba, bw) pairs give %d (scale meets its data)?' % (K - 32 + 128))
    for ba in range(K // 32):
        hits = []
        for bw in range(K // 32):
            A = ones(M, K)
            A[:, 32 * ba:32 * ba + 32] = TWO
            C = run(A, scales(M, K, unit), ones(N, K), scales(N, K, lambda r, b, bw=bw: 127 + (b == bw).long()), M, N, K)
            if C.min() == C.max() == K - 32 + 128:
                hits.append(bw)
        print('   A block %d <- W scale block(s) %s' % (ba, hits))
    print('5c. the same with the scale on A (A = 1, A scale 2^1 in block bs) and W = 2.0 in block bw')
    for bw in range(K // 32):
        hits = []
        for bs in range(K // 32):
            W = ones(N, K)
            W[:, 32 * bw:32 * bw + 32] = TWO
            C = run(ones(M, K), scales(M, K, lambda r, b, bs=bs: 127 + (b == bs).long()), W, scales(N, K, unit), M, N, K)
            if C.min() == C.max() == K - 32 + 128:
                hits.append(bs)
        print('   W block %d <- A scale block(s) %s' % (bw, hits))
    print('6. A[m, k] = 1.0 only for k == m %% K (one-hot), W[n, k] = 1.0 only for k == n %% K: expect identity pattern C[m,n] = [m%K == n%K]')
    A = torch.zeros(M, K, dtype=torch.uint8, device=DEV)
    A[torch.arange(M), torch.arange(M) % K] = ONE
    W = torch.zeros(N, K, dtype=torch.uint8, device=DEV)
    W[torch.arange(N), torch.arange(N) % K] = ONE
    C = run(A, scales(M, K, unit), W, scales(N, K, unit), M, N, K)
    eye = (torch.arange(M).view(-1, 1) % K == torch.arange(N).view(1, -1) % K).float()
    bad = (C != eye).nonzero()
    print('   mismatches: %d  first: %s' % (bad.shape[0], bad[:6].tolist()))
    if bad.shape[0]:
        nz = C.nonzero()
        print('   nonzeros of C (first 12): %s' % nz[:12].tolist())


if __name__ == "__main__":
    main()
