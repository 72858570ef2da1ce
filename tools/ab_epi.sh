set -e
echo "== HEAD" > gpurun_out/ab.log
MVF_HIP_LIB=$PWD/tmp_ab/libmvf_head.so python tools/gemm_bench.py --shapes proj,fc2 >> gpurun_out/ab.log 2>&1
echo "== NEW" >> gpurun_out/ab.log
python tools/gemm_bench.py --shapes proj,fc2 >> gpurun_out/ab.log 2>&1
MVF_HIP_LIB=$PWD/tmp_ab/libmvf_head.so python bench.py --no-cpu-baseline >> gpurun_out/ab.log 2>&1
python bench.py --no-cpu-baseline >> gpurun_out/ab.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tl -o tl --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/tl.log 2>&1
