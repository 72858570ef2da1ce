cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "qkv_projection_fused" 2>&1 | grep -v amdgpu | tail -8
python tools/energy_probe.py --smi --seconds 3 --kernels qkv,attn,qkv_attn,qkv_attn 2>&1 | grep -E "^qkv|^attn|^kernel"
