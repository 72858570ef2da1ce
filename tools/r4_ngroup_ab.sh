cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or ln_fold or resid2 or tc or scl or lstp_one_pass" 2>&1 | tail -4
MVF_GEMM_NGROUP=3 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or ln_fold or resid2 or tc" 2>&1 | tail -3
python tools/scl_gathered.py > gpurun_out/scl_gathered.txt 2>&1; cat gpurun_out/scl_gathered.txt
for g in 0 3 0 3 4 6; do echo "== MVF_GEMM_NGROUP=$g"; MVF_GEMM_NGROUP=$g python tools/energy_probe.py --smi --seconds 3 --kernels qkv,fc1 2>&1 | grep -E "^qkv|^fc1"; done
for g in 0 3; do
  MVF_GEMM_NGROUP=$g bash tools/collect_pmc_traffic.sh > gpurun_out/pmc_traffic_ng$g.log 2>&1
  cp gpurun_out/pmc_traffic.json gpurun_out/pmc_traffic_ng$g.json
done
python - <<'PY'
import json
for g in (0,3):
    j=json.load(open('gpurun_out/pmc_traffic_ng%d.json'%g))
    for k,v in j.items():
        if isinstance(v,dict): print('ngroup',g,k[:20],'fetch MB %.0f write MB %.0f hbm/alg %.3f'%(v['fetch_size_kb']*2*1024/1e6, v['write_size_kb']*1024/1e6, v['ratio']))
PY
bash tools/pmc_sq.sh > /dev/null 2>&1; python3 tools/pmc_sq_summary.py gpurun_out/pmc_sq > gpurun_out/pmc_sq_gemm.txt; grep -E "^gemm|MFMA busy" gpurun_out/pmc_sq_gemm.txt
