"""fp8 scaled-MFMA probe 2: where does a W (src0) / A (src1) block scale land?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fp8_debug import run, scales, ONE, TWO, DEV   # noqa
M = N = K = 256
ones = lambda r, k: torch.full((r, k), ONE, dtype=torch.uint8, device=DEV)
unit = lambda r, b: torch.full_like(r, 127)
for which in ('W', 'A'):
    for bs in range(4):
        sc = lambda r, b, bs=bs: 127 + 3 * (b == bs).long()      # x8 on one block of every row
        if which == 'W':
            C = run(ones(M, K), scales(M, K, unit), ones(N, K), scales(N, K, sc), M, N, K)
        else:
            C = run(ones(M, K), scales(M, K, sc), ones(N, K), scales(N, K, unit), M, N, K)
        # expected: 256 - 32 + 256 = 480 everywhere
        vals, cnt = torch.unique(C, return_counts=True)
        print('%s scale x8 on block %d: unique values %s counts %s' % (which, bs, vals.tolist(), cnt.tolist()))
        print('    C[0:4, 0:20] row0 %s' % C[0, :20].tolist())
        print('    C[0:20, 0] col0 %s' % C[:20, 0].tolist())
# per-row-varying block scale: W scale byte = 127 + (n % 16 == 3) in block 2 only
sc = lambda r, b: 127 + 3 * ((b == 2) & (r % 16 == 3)).long()
C = run(ones(M, K), scales(M, K, unit), ones(N, K), scales(N, K, sc), M, N, K)
nz = (C != 256).nonzero()
print('W: x8 on (n %% 16 == 3, block 2): affected columns %s rows %s values %s' % (sorted(set(nz[:, 1].tolist()))[:12], sorted(set(nz[:, 0].tolist()))[:6], torch.unique(C).tolist()))
C = run(ones(M, K), scales(M, K, sc), ones(N, K), scales(N, K, unit), M, N, K)
nz = (C != 256).nonzero()
print('A: x8 on (m %% 16 == 3, block 2): affected rows %s cols %s values %s' % (sorted(set(nz[:, 0].tolist()))[:12], sorted(set(nz[:, 1].tolist()))[:6], torch.unique(C).tolist()))
