cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_configs.py -x -q -k "test_full_size and not fg99 and not 192" 2>&1 | grep -v amdgpu | tail -5
grep -i "fp16" gpurun_out/parity.txt | tail -3
python -m pytest tests/test_gpu_kernels.py -x -q -k "test_vit_attention" 2>&1 | grep -v amdgpu | tail -3
python tools/energy_probe.py --smi --seconds 3 --kernels attn,attn_msum,attn,attn_msum 2>&1 | grep -E "^attn"
bash tools/r4_cus_ab.sh
