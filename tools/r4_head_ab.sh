# head GEMM tile height A/B: 64-row tiles (MVF_HGEMM_TM=64, the form up to round 3) against the automatic choice (32 rows where
# 64-row tiles give fewer than 192 workgroups); parity tests, then 400 sustained steps each and the serial per-kernel profile
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q -k "linear or static_query or lstp_pool_vs or temporal" 2>&1 | grep -v amdgpu | tail -3
python -m pytest tests/test_gpu_model.py -x -q 2>&1 | grep -v amdgpu | tail -3
for tm in 64 0 64 0; do
  MVF_HGEMM_TM=$tm python bench.py --no-cpu-baseline --steps 400 --warmup 30 2>gpurun_out/ab.err | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('MVF_HGEMM_TM=$tm', j['ms_per_step'], j['value'])"
done
cd /tmp && export TMPDIR=/tmp
for tm in 64 0; do
  MVF_HGEMM_TM=$tm rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/head_tm$tm -o p -- python3 $GRAFT_REPO_ROOT/bench.py --serial --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/head_tm$tm/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows:
    if 'hgemm' in r['Name'] or 'hlinear' in r['Name']:
        print('TM=$tm', r['Name'][:70], r['Calls'], '%.1f us'%(float(r['AverageNs'])/1e3))
PY
done
