# Round 6 evidence (GPU box): bash tools/r6_final.sh   -> gpurun_out/r06/ (publish with tools/r6_publish.sh HERE afterwards)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r06
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# 1. the bench line + rocprofv3 kernel stats (pipelined and --serial) + PMC traffic / SQ counters of the GEMMs
bash $R/tools/collect_profiles.sh r06 > $out/collect.log 2>&1
# 2. the accuracy mode (fp16 backbone + fp16-forward head)
python3 $R/bench.py --dtype fp16 2>/dev/null | grep '^{' | tail -1 > $out/bench_fp16.json
# 3. every BASELINE config shape + the shipped configs
python3 $R/tools/config_sweep.py > $out/config_sweep.txt 2>&1
# 4. the streamed attention kernel: rocprofv3 kernel stats (average launch duration) at the shipped token counts, old and new kernel
for s in "256 20 577 16" "80 20 785 12" "256 20 257 12"; do
  set -- $s
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/attn_$3 -o run -- python3 $R/tools/attn_bench.py $s 7 0 4096 0 4096 7 > $out/attn_$3.log 2>&1
  cp $out/attn_$3/run_kernel_stats.csv $out/attn_N$3_kernel_stats.csv
  rm -rf $out/attn_$3
  python3 $R/tools/attn_bench.py $s --sustain 1.0 7 0 4096 8 > $out/attn_N$3_bench.txt 2>&1
done
# 5. configs[4] kernel by kernel
for d in fp8 bf16; do
  python3 $R/tools/config4_roofline.py --dtype $d --steps 8 2>/dev/null | tail -1 > $out/config4_$d.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/c4_$d -o run -- python3 $R/tools/config4_roofline.py --dtype $d --steps 4 --serial > $out/c4_$d.log 2>&1
  cp $out/c4_$d/run_kernel_stats.csv $out/config4_${d}_serial_kernel_stats.csv
  rm -rf $out/c4_$d
done
# 6. soak: 800 training steps with the default head (fp16 forward operands)
python3 $R/tools/soak_train.py 800 bf16 > $out/soak.txt 2>&1
find $out -name "*.db" -delete
tail -3 $out/soak.txt; grep -v "^Using\|amdgpu.ids" $out/config_sweep.txt | cut -c1-190
