# fp16 vs bf16 compute mode on the product step (400 sustained steps each, same box), then the configs[1] parity test with its fp16 leg
cd $GRAFT_REPO_ROOT
for dt in bf16 fp16 bf16 fp16; do
  python bench.py --dtype $dt --no-cpu-baseline --steps 400 --warmup 30 2>gpurun_out/ab.err | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$dt', j['ms_per_step'], j['value'], {k: v['avg_us'] for k, v in j['roofline']['by_kernel'].items()})"
done
python bench.py --dtype fp16 --steps 50 --warmup 20 > gpurun_out/bench_fp16.json 2> gpurun_out/bench_fp16.err; tail -c 1500 gpurun_out/bench_fp16.json
python -m pytest tests/test_gpu_configs.py -x -q -k "test_full_size_fp32_and_bf16_vs_oracle and not fg99 and not T=64" 2>&1 | grep -v amdgpu | tail -5
grep -i "fp16" gpurun_out/parity.txt | tail -3
