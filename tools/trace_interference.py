"""Reads a rocprofv3 --kernel-trace CSV of a pipelined bench.py run and asks: does a backbone kernel take longer while a head kernel is in
flight?  For every backbone kernel instance in the last 60 % of the trace: its duration and the fraction of its interval covered by head
kernels (everything that is not a backbone kernel); per kernel name the mean duration of the instances with < 5 % / > 50 % head cover, and
the least-squares slope of duration against cover.  Also: the idle time of the device (no kernel of any queue running) per step.
Usage: python3 tools/trace_interference.py <dir>"""
import csv
import glob
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
cut = t0 + int((t1 - t0) * 0.4)
rows = [r for r in rows if r[0] >= cut]
BACKBONE = ('gemm_tc256', 'vit_qkv_attn', 'vit_attn', 'layernorm_kernel<unsigned short', 'im2col', 'ln_stats_finalize', 'cls_row', 'cast_')
is_bb = lambda n: any(k in n for k in BACKBONE)
head = sorted((s, e) for s, e, n in rows if not is_bb(n))
# merged head intervals
merged = []
for s, e in head:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
starts = [m[0] for m in merged]
import bisect


def cover(s, e):
    i = max(bisect.bisect_right(starts, s) - 1, 0)
    c = 0
    while i < len(merged) and merged[i][0] < e:
        c += max(0, min(e, merged[i][1]) - max(s, merged[i][0]))
        i += 1
    return c / max(e - s, 1)


by = defaultdict(list)
for s, e, n in rows:
    if is_bb(n):
        by[n.replace('(anonymous namespace)::', '')[:70]].append(((e - s) / 1e3, cover(s, e)))
print('%-72s %6s %9s %9s %7s %9s' % ('backbone kernel', 'n', 'us cov<5%', 'us cov>50%', 'n>50%', 'slope us'))
tot_lo = tot_hi = 0.0
for n, v in sorted(by.items(), key=lambda kv: -sum(d for d, c in kv[1])):
    if len(v) < 20:
        continue
    lo = [d for d, c in v if c < 0.05]
    hi = [d for d, c in v if c > 0.5]
    mx = sum(c for d, c in v) / len(v)
    my = sum(d for d, c in v) / len(v)
    sxx = sum((c - mx) ** 2 for d, c in v)
    slope = sum((c - mx) * (d - my) for d, c in v) / sxx if sxx > 0 else 0.0
    print('%-72s %6d %9.1f %9.1f %7d %9.1f' % (n, len(v), sum(lo) / max(len(lo), 1), sum(hi) / max(len(hi), 1), len(hi), slope))
# device idle time
ev = sorted([(s, 1) for s, e, n in rows] + [(e, -1) for s, e, n in rows])
active, idle, last = 0, 0, rows[0][0]
for t, d in ev:
    if active == 0:
        idle += t - last
    active += d
    last = t
span = rows[-1][1] - rows[0][0]
print('window %.1f ms, device idle (no kernel on any queue) %.2f %%, head kernels in flight %.1f %% of the time' % (
    span / 1e6, 100.0 * idle / span, 100.0 * sum(e - s for s, e in merged) / span))
