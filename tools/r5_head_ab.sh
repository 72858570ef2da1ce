# Round 5: the head's cost to the pipelined step with and without the row-chain kernels (csrc/head_chain.hip).
# Usage (GPU box): bash tools/r5_head_ab.sh        outputs under gpurun_out/r05/
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in 0 1; do
  echo "== MVF_HEAD_CHAIN=$c: step timeline" >> $out/head_ab.txt
  MVF_HEAD_CHAIN=$c python3 $GRAFT_REPO_ROOT/tools/step_timeline.py --steps 300 2>/dev/null | tail -12 >> $out/head_ab.txt
done
echo "== backbone forwards only" >> $out/head_ab.txt
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py --steps 300 --no-head 2>/dev/null | tail -6 >> $out/head_ab.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_serial -o run -- python3 $GRAFT_REPO_ROOT/bench.py --serial --no-cpu-baseline --steps 30 --warmup 10 > $out/stats_serial.log 2>&1
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete
cat $out/head_ab.txt
