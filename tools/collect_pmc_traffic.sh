# Re-collects ONLY profiles/pmc_traffic.json (needed after any edit of gemm_tc256.hip / gemm_tc_epi.h: the file carries their sha).
# Usage on the GPU box: bash tools/collect_pmc_traffic.sh ; then copy gpurun_out/pmc_traffic.json to profiles/pmc_traffic.json
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_only
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $R/tools/gemm_bench.py --iters 3 --warm 1 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $R/tools/gemm_bench.py --iters 3 --warm 1 > $out/pmc_write.log 2>&1
python3 $R/tools/pmc_summary.py $out/pmc_fetch $out/pmc_write 4 > $R/gpurun_out/pmc_traffic.json
find $out -name "*.db" -delete; find $out -name "*counter_collection.csv" -delete
cat $R/gpurun_out/pmc_traffic.json | head -40
