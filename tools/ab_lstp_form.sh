for f in 1 0 1 0; do
  MVF_LSTP_FORM=$f python bench.py --no-cpu-baseline > gpurun_out/ab_lstp_$f.json 2>/dev/null
  python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/ab_lstp_$f.json') if l.startswith('{')][0])
print('form $f', j['ms_per_step'], j['value'])
PY
done
cd /tmp && export TMPDIR=/tmp
MVF_LSTP_FORM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lstp_step -o run -- python3 $GRAFT_REPO_ROOT/bench.py --serial --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/lstp_step/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "lstp" in r["Name"]: print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
