# Copies what tools/collect_profiles.sh <tag> left under gpurun_out/<tag>/ (merged back from the GPU box) into profiles/<tag>/ under the
# names profiles/README.md lists.  Run HERE after the gpurun call:  bash tools/publish_profiles.sh r04
set -e
tag=${1:-r04}
src=gpurun_out/$tag
dst=profiles/$tag
mkdir -p $dst
grep '^{' $src/bench.log | tail -1 > $dst/bench_n1.json
grep '^{' $src/stats.log | tail -1 > $dst/bench_n1_pipelined_under_rocprof.json
grep '^{' $src/stats_serial.log | tail -1 > $dst/bench_n1_serial.json
cp $src/stats/run_kernel_stats.csv $dst/bench_n1_kernel_stats.csv
cp $src/stats_serial/run_kernel_stats.csv $dst/bench_n1_serial_kernel_stats.csv
cp $src/pmc_traffic.json $dst/pmc_traffic_gemm.json
cp $src/pmc_traffic.json profiles/pmc_traffic.json
cp $src/pmc_sq_gemm.txt $dst/pmc_sq_gemm.txt
cp $src/lstp_bench.txt $dst/lstp_bench.txt
[ -f gpurun_out/parity_full.txt ] && cp gpurun_out/parity_full.txt $dst/parity.txt
[ -f gpurun_out/pmc_sq_qkv_attn.txt ] && grep -v "^\s*$" gpurun_out/pmc_sq_qkv_attn.txt | grep -A40 "vit_qkv_attn_kernel" > $dst/pmc_sq_qkv_attn.txt
ls -la $dst | tail -20
