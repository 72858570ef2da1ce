cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q -k "scl" 2>&1 | grep -v amdgpu | tail -5
cd /tmp && export TMPDIR=/tmp
for e in 128 256; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/scl_prof$e -o p -- python3 $GRAFT_REPO_ROOT/tools/scl_gathered.py $e > /dev/null 2>&1
done
