# Round 5: the non-headline configs on the final sources + the LayerNorm-fold A/B.
# Usage (GPU box): bash tools/r5_sweep.sh     outputs under gpurun_out/r05/
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/config_sweep.py > $out/config_sweep.txt 2>&1
cat $out/config_sweep.txt | grep -v "^Using\|amdgpu.ids"
: > $out/ln_fold_ab.txt
for rep in 1 2; do for f in 2 1; do
  echo -n "MVF_LN_FOLD=$f (2 = norm1 folded, default; 1 = norm1 and norm2 folded, proj's residual epilogue is the producer): " >> $out/ln_fold_ab.txt
  MVF_LN_FOLD=$f python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 300 --warmup 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms/step')" >> $out/ln_fold_ab.txt
done; done
cat $out/ln_fold_ab.txt
