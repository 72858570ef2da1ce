set -x
python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or ln_fold or resid2 or tc" 2>&1 | tail -5
for s in 2 7 1 6 2 7; do
  echo "=== BM_SET=$s"
  MVF_GEMM_BM_SET=$s python bench.py --no-cpu-baseline > gpurun_out/abm_$s.json 2> gpurun_out/abm_$s.err
  python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/abm_$s.json') if l.startswith('{')][0])
print(j['ms_per_step'], j['value'], json.dumps(j['roofline'])[:1500])
PY
done
