# Round 5: the head's cost to the pipelined step with and without the row-chain kernels (csrc/head_chain.hip), one box.
# Usage (GPU box): bash tools/r5_head_ab.sh [tag]       output: gpurun_out/r05/head_ab_<tag>.txt
tag=${1:-x}
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
f=$out/head_ab_$tag.txt
: > $f
for rep in 1 2; do
for c in 0 1; do
  echo "== MVF_HEAD_CHAIN=$c: step timeline (run $rep)" >> $f
  MVF_HEAD_CHAIN=$c python3 $GRAFT_REPO_ROOT/tools/step_timeline.py --steps 400 2>/dev/null | tail -2 >> $f
done
done
echo "== backbone forwards only" >> $f
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py --steps 400 --no-head 2>/dev/null | tail -2 >> $f
cat $f
