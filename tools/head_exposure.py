"""Reads a rocprofv3 --kernel-trace CSV of a pipelined bench.py run and reports, over the last 40 % of the trace: wall time per step,
per-queue busy time, and for the head's kernels (everything that is not a backbone kernel) their summed duration and the wall-clock span
from the first to the last head kernel of a step -- how stretched the head chain is by sharing the chip with the backbone lanes.
Usage: python3 tools/head_exposure.py <dir> [steps-per-trace]"""
import csv
import glob
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
cut = t0 + int((t1 - t0) * 0.6)
rows = [r for r in rows if r[0] >= cut]
BACKBONE = ('gemm_tc256', 'vit_qkv_attn', 'vit_attn', 'layernorm_kernel<unsigned short', 'im2col', 'ln_stats_finalize')
is_bb = lambda n: any(k in n for k in BACKBONE)
# steps: delimited by adam_kernel (2 per step: take every second)
adam = [r for r in rows if 'adam_kernel' in r[2]]
marks = [adam[i][1] for i in range(1, len(adam), 2)]
print('steps in window: %d, wall per step %.3f ms' % (len(marks) - 1, (marks[-1] - marks[0]) / 1e6 / (len(marks) - 1)))
q_busy = defaultdict(int)
for s, e, n, q in rows:
    if marks[0] <= s < marks[-1]:
        q_busy[q] += e - s
for q, b in sorted(q_busy.items()):
    print('queue %s busy %.3f ms/step' % (q, b / 1e6 / (len(marks) - 1)))
spans, sums, cnts = [], [], []
bb_sum = defaultdict(int); bb_cnt = defaultdict(int)
for i in range(len(marks) - 1):
    hk = [(s, e) for s, e, n, q in rows if marks[i] <= e <= marks[i + 1] and not is_bb(n)]
    if not hk:
        continue
    spans.append((max(e for s, e in hk) - min(s for s, e in hk)) / 1e3)
    sums.append(sum(e - s for s, e in hk) / 1e3)
    cnts.append(len(hk))
for s, e, n, q in rows:
    if marks[0] <= s < marks[-1] and is_bb(n):
        k = n.replace('(anonymous namespace)::', '').replace('void ', '')[:60]
        bb_sum[k] += e - s; bb_cnt[k] += 1
print('head kernels per step: %d launches, summed duration %.0f us, first-to-last span %.0f us' %
      (sum(cnts) / len(cnts), sum(sums) / len(sums), sum(spans) / len(spans)))
for k in sorted(bb_sum, key=lambda k: -bb_sum[k]):
    print('  %-62s %6.1f/step  avg %7.1f us' % (k, bb_cnt[k] / (len(marks) - 1), bb_sum[k] / bb_cnt[k] / 1e3))
# idle: wall time in the window during which NO backbone kernel runs
ev = sorted([(s, 1) for s, e, n, q in rows if is_bb(n) and marks[0] <= s < marks[-1]] + [(e, -1) for s, e, n, q in rows if is_bb(n) and marks[0] <= s < marks[-1]])
depth, last, idle = 0, marks[0], 0
for t, d in ev:
    if depth == 0:
        idle += t - last
    depth += d
    last = t
print('wall time with no backbone kernel running: %.3f ms/step' % (idle / 1e6 / (len(marks) - 1)))
