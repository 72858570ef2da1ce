cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q -k "fp16" 2>&1 | grep -v amdgpu | tail -25
