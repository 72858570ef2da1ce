set -e
for cfg in "2 0" "1 1" "2 1" "4 0" "1 0"; do
  set -- $cfg
  touch video_rep_learning_amd/csrc/gemm_tc256.hip
  MVF_EXTRA_FLAGS="-DMVF_EPI_EB=$1 -DMVF_EPI_PIPE=$2" python -m video_rep_learning_amd.csrc.build > /dev/null
  echo "=== EB=$1 PIPE=$2" >> gpurun_out/epi_sweep.log
  python tools/gemm_bench.py --shapes proj,fc2 >> gpurun_out/epi_sweep.log 2>&1
done
python -m pytest tests/test_gpu_kernels.py -q -x -k gemm 2>&1 | tail -3 >> gpurun_out/epi_sweep.log
