# Round 5: finer sweep of MVF_GEMM_SPARE and its interplay with the optimizer's width (full pipelined step, 300 steps per line)
cd /tmp
T=$GRAFT_REPO_ROOT/tools/step_timeline.py
for rep in 1 2; do
for cfg in "0 64" "24 64" "32 64" "40 64" "48 64" "32 32" "32 24" "32 0" "0 64"; do
  set -- $cfg
  a=$(MVF_GEMM_SPARE=$1 MVF_OPT_WIDTH=$2 python3 $T --steps 300 2>/dev/null | grep "^wall" | cut -c6-12)
  echo "MVF_GEMM_SPARE=$1 MVF_OPT_WIDTH=$2 (run $rep): step $a ms"
done
done
