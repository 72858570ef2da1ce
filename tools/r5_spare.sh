# Round 5: CUs the persistent backbone GEMM leaves free where that costs no tile round (MVF_GEMM_SPARE), full step and forwards only.
cd /tmp
T=$GRAFT_REPO_ROOT/tools/step_timeline.py
for rep in 1 2; do
for m in 0 16 32 64 104 256; do
  a=$(MVF_GEMM_SPARE=$m python3 $T --steps 300 2>/dev/null | grep "^wall" | cut -c6-12)
  b=$(MVF_GEMM_SPARE=$m python3 $T --steps 300 --no-head 2>/dev/null | grep "^wall" | cut -c6-12)
  echo "MVF_GEMM_SPARE=$m (run $rep): step $a ms, forwards only $b ms"
done
done
