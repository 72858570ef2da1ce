# Round 5: A/B of two builds of the library on the pipelined step and on the serial kernel stats, one box.
# Usage (GPU box): bash tools/r5_lib_ab.sh <variant> [tag]    (variant: tools/probes/libmvf_<variant>.so from tools/build_variant.sh)
# output: gpurun_out/r05/lib_ab_<tag>.txt
v=$1; tag=${2:-$1}
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
f=$out/lib_ab_$tag.txt
: > $f
for rep in 1 2 3; do
  echo "== product build: step timeline (run $rep)" >> $f
  python3 $GRAFT_REPO_ROOT/tools/step_timeline.py --steps 400 2>/dev/null | tail -2 >> $f
  echo "== $v build: step timeline (run $rep)" >> $f
  MVF_HIP_LIB=$GRAFT_REPO_ROOT/tools/probes/libmvf_$v.so python3 $GRAFT_REPO_ROOT/tools/step_timeline.py --steps 400 2>/dev/null | tail -2 >> $f
done
cat $f
