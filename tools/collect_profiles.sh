# Round-end evidence (GPU box): bench line, rocprofv3 kernel stats of the same command, PMC traffic of the GEMM shapes.
# Usage: bash tools/collect_profiles.sh <tag>      outputs under gpurun_out/<tag>/
set -e
tag=${1:-final}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py > $out/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $out/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_serial -o run -- python3 $GRAFT_REPO_ROOT/bench.py --serial --no-cpu-baseline > $out/stats_serial.log 2>&1
# PMC passes on their own (never together with a trace): the four GEMM shapes, plain and with the LN-fold extras (--ln)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 3 --warm 1 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 3 --warm 1 > $out/pmc_write.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/pmc_fetch $out/pmc_write 4 > $out/pmc_traffic.json
# SQ counters of the qkv / fc1 launches (MFMA busy, LDS, waits): five separate --pmc passes (tools/pmc_sq.sh)
bash $GRAFT_REPO_ROOT/tools/pmc_sq.sh > $out/pmc_sq.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_sq_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_sq > $out/pmc_sq_gemm.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/lstp_bench.py > $out/lstp_bench.txt 2>&1
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete
ls -R $out | head -40
