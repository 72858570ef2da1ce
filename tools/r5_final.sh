mkdir -p gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/config_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/r05/config_sweep_v2.txt 2>&1
cd $GRAFT_REPO_ROOT
bash tools/r5_config4.sh > gpurun_out/r05/config4_v2.log 2>&1
bash tools/collect_profiles.sh r05 > gpurun_out/r05/collect.log 2>&1
bash tools/r5_head_stats.sh chain_v11 2>&1 | tail -3
grep -v "^Using\|amdgpu.ids" gpurun_out/r05/config_sweep_v2.txt | cut -c1-200
