"""Reads a rocprofv3 --kernel-trace CSV directory and reports, per kernel-name group, total busy time and how much of it
overlaps (in wall-clock) a running gemm_tc256 dispatch of ANOTHER stream -- evidence for kernels co-executing with the
persistent GEMM.  Usage: python3 tools/trace_overlap.py <dir>"""
import csv
import glob
import sys
from collections import Counter, defaultdict

files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', r.get('Stream_Id', '0'))))
rows.sort()
if not rows:
    print('no rows in', sys.argv[1]); sys.exit(0)
# keep the last 40 % of the trace (steady state)
t0, t1 = rows[0][0], rows[-1][1]
cut = t0 + int((t1 - t0) * 0.6)
rows = [r for r in rows if r[0] >= cut]
gemm = [(s, e, q) for s, e, n, q in rows if 'gemm_tc256' in n]


def group(n):
    n_l = n.lower()
    if 'nccl' in n_l:
        return 'nccl'
    for k in ('nccl', 'gemm_tc256', 'layernorm_kernel', 'vit_attn', 'im2col', 'ln_stats_finalize', 'lstp', 'hgemm', 'hlinear_bwd', 'tattn', 'adam', 'scl_'):
        if k in n:
            return k
    return 'other'


busy = defaultdict(int); ov = defaultdict(int); cnt = defaultdict(int)
gi = 0
for s, e, n, q in rows:
    g = group(n)
    busy[g] += e - s; cnt[g] += 1
    o = 0
    for gs, ge, gq in gemm:
        if ge <= s: continue
        if gs >= e: break
        if gq == q and gs == s: continue
        if gq == q: continue
        o += max(0, min(e, ge) - max(s, gs))
    ov[g] += min(o, e - s)
# every RCCL kernel: when did it run relative to the kernels of the other queues?
nccl = [(s, e, n, q) for s, e, n, q in rows if 'nccl' in n.lower()]
if nccl:
    print('RCCL kernels in the steady-state window: %d' % len(nccl))
    for s, e, n, q in nccl[:40]:
        conc = Counter()
        for s2, e2, n2, q2 in rows:
            if q2 != q and s2 < e and e2 > s:
                conc[group(n2)] += 1
        print('  %-44s %7.1f us  concurrent kernels of other queues: %s' % (n[:44], (e - s) / 1e3, dict(conc) or 'none'))
span = rows[-1][1] - rows[0][0]
print('trace', sys.argv[1], 'steady-state span %.2f ms' % (span / 1e6))
for g in sorted(busy, key=lambda k: -busy[k]):
    print('%-20s calls %6d busy %8.2f ms (%.0f%% of span) overlapped with another queue\'s gemm_tc256: %.0f%%  avg %.1f us' % (
        g, cnt[g], busy[g] / 1e6, 100.0 * busy[g] / span, 100.0 * ov[g] / max(1, busy[g]), busy[g] / 1e3 / cnt[g]))
