# A/B: the persistent GEMM's CU budget with two backbone lanes (does leaving CUs to the other lane's memory-bound kernels pay?)
for c in 0 224 192 160 128 0; do
  echo "== MVF_GEMM_CUS=$c"
  MVF_GEMM_CUS=$c python tools/vit_streams_probe.py 256 2>&1 | grep "streams 2" | tail -1
  MVF_GEMM_CUS=$c python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); print('step', j['ms_per_step'])"
done
