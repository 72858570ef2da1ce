# HERE, after `gpurun -- bash tools/r6_final.sh`: copies the judged summaries from gpurun_out/r06 into profiles/r06
set -e
bash tools/publish_profiles.sh r06
src=gpurun_out/r06; dst=profiles/r06
for f in bench_fp16.json config_sweep.txt attn_N577_kernel_stats.csv attn_N785_kernel_stats.csv attn_N257_kernel_stats.csv attn_N577_bench.txt attn_N785_bench.txt \
         attn_N257_bench.txt config4_fp8.json config4_bf16.json config4_fp8_serial_kernel_stats.csv config4_bf16_serial_kernel_stats.csv soak.txt; do
  [ -f $src/$f ] && cp $src/$f $dst/$f
done
[ -f gpurun_out/parity.txt ] && cp gpurun_out/parity.txt $dst/parity.txt
ls $dst
