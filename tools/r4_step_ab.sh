# same-box A/B of the round-4 GEMM changes on the product step (400 sustained steps each)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or ln_fold or resid2 or tc" 2>&1 | tail -3
for cfg in "MVF_GEMM_NGROUP=0 MVF_LN_INKERNEL=0" "MVF_GEMM_NGROUP=-1 MVF_LN_INKERNEL=0" "MVF_GEMM_NGROUP=-1 MVF_LN_INKERNEL=1" "MVF_GEMM_NGROUP=0 MVF_LN_INKERNEL=0" "MVF_GEMM_NGROUP=-1 MVF_LN_INKERNEL=1"; do
  env $cfg python bench.py --no-cpu-baseline --steps 400 --warmup 30 2>gpurun_out/ab.err | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$cfg', j['ms_per_step'], j['value'], {k: v['avg_us'] for k, v in j['roofline']['by_kernel'].items()})"
done
