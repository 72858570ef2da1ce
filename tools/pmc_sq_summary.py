"""Mean per launch of every SQ / GRBM counter collected by tools/pmc_sq.sh (separate rocprofv3 --pmc passes over
tools/gemm_bench.py --shapes qkv,proj,fc1,fc2), per kernel name, first launch of each kernel dropped (cold).
Usage: python3 tools/pmc_sq_summary.py gpurun_out/pmc_sq [kernel-name substring, default gemm_tc256_kernel]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else 'gemm_tc256_kernel'
vals = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
    per_disp = defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if not any(w in r['Kernel_Name'] for w in want.split(',')):
            continue
        per_disp[(int(r['Dispatch_Id']), r['Kernel_Name'])][r['Counter_Name']] = float(r['Counter_Value'])
    seen = set()
    for (did, kn), cs in sorted(per_disp.items()):
        short = re.sub(r'\(\(anonymous namespace\)::\w+\).*|\(gemm_tc::GemmTcArgs\).*', '', kn.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', ''))
        if short not in seen:
            seen.add(short)        # drop the first (cold) launch of each kernel in each pass
            continue
        for c, v in cs.items():
            vals[short][c].append(v)
print('tools/pmc_sq.sh: rocprofv3 --pmc passes (4 counters each, no trace) over tools/gemm_bench.py --shapes qkv,proj,fc1,fc2; mean per launch,')
print('summed over the 8 XCDs (GRBM_GUI_ACTIVE / 8 = cycles of the launch).  <0,..,256> = qkv, <0,..,208> = proj, <1,..> = fc1 + GELU, <2,..> = fc2 + residual (+ deferred attention-branch addend).\n')
for kn in sorted(vals):
    print(kn)
    cs = vals[kn]
    for c in cs:
        print('   %-34s %.4g' % (c, sum(cs[c]) / len(cs[c])))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and 'GRBM_GUI_ACTIVE' in cs:
        cyc = sum(cs['GRBM_GUI_ACTIVE']) / len(cs['GRBM_GUI_ACTIVE']) / 8.0
        busy = sum(cs['SQ_VALU_MFMA_BUSY_CYCLES']) / len(cs['SQ_VALU_MFMA_BUSY_CYCLES'])
        print('   => MFMA busy %.1f %% of %d SIMD x %.0f launch cycles' % (100.0 * busy / (1024 * cyc), 1024, cyc))
