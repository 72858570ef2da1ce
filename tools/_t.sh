cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_configs.py -x -q -k "test_full_size and not fg99 and not 192" 2>&1 | grep -v amdgpu | tail -4
grep -E "HIP (bf16|fp16):" gpurun_out/parity.txt | tail -2 | cut -c1-330
bash tools/r4_head_ab.sh
