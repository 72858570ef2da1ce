# Evidence for the N > 1 path on a ONE-GPU box: the forced one-rank `nccl` (= RCCL) step issues every collective of the
# data-parallel step (bucketed async gradient all-reduce, SyncBN all-gather / all-reduce, loss all-reduce).  A kernel trace shows
# whether RCCL's kernels run beside the head-backward / backbone-lane kernels, with and without the 8-CU reserve.
# Usage: bash tools/rccl_overlap.sh <tag>
tag=${1:-rccl}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 MVF_FORCE_REDUCER=1
B="--no-cpu-baseline --steps 30 --warmup 10"
python3 $R/bench.py $B > $out/bench_reserve8.log 2>&1
MVF_RCCL_CUS=0 python3 $R/bench.py $B > $out/bench_reserve0.log 2>&1
env -u MVF_FORCE_REDUCER python3 $R/bench.py $B > $out/bench_plain.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/trace_reserve8 -o t -- python3 $R/bench.py --no-cpu-baseline --steps 8 --warmup 4 --profile-steps 0 > $out/trace_reserve8.log 2>&1
export MVF_RCCL_CUS=0
rocprofv3 --kernel-trace --output-format csv -d $out/trace_reserve0 -o t -- python3 $R/bench.py --no-cpu-baseline --steps 8 --warmup 4 --profile-steps 0 > $out/trace_reserve0.log 2>&1
python3 $R/tools/trace_overlap.py $out/trace_reserve8 > $out/overlap_reserve8.txt 2>&1
python3 $R/tools/trace_overlap.py $out/trace_reserve0 > $out/overlap_reserve0.txt 2>&1
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete
for f in $out/bench_*.log; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; grep -o '"gemm_cu_budget": [^}]*' $f; done
head -60 $out/overlap_reserve8.txt; head -60 $out/overlap_reserve0.txt
