# Round 5: configs[4] (DINOv2 ViT-L/14 @ 336, MX-fp8): roofline line + rocprofv3 kernel stats of the same command.
# Usage (GPU box): bash tools/r5_config4.sh     outputs under gpurun_out/r05/
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for d in fp8 bf16; do
  python3 $GRAFT_REPO_ROOT/tools/config4_roofline.py --dtype $d --steps 8 2>/dev/null | tail -1 > $out/config4_$d.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/c4_$d -o run -- python3 $GRAFT_REPO_ROOT/tools/config4_roofline.py --dtype $d --steps 4 --serial > $out/c4_$d.log 2>&1
  cp $out/c4_$d/run_kernel_stats.csv $out/config4_${d}_serial_kernel_stats.csv
  rm -rf $out/c4_$d
done
python3 - <<P
import csv, json
for d in ('fp8', 'bf16'):
    print(d, open('$out/config4_%s.json' % d).read()[:600])
    rows = list(csv.DictReader(open('$out/config4_%s_serial_kernel_stats.csv' % d)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    for r in rows[:14]:
        print('   %-90s calls %5s avg %9.1f us  %5.1f %%' % (r['Name'].replace('(anonymous namespace)::', '')[:90], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
P
