# Socket power and clocks while (a) the product training step and (b) each backbone GEMM shape on its own run for seconds
# (GPU box; rocm-smi read-only).  Is the STEP at the power cap on average, or only a back-to-back GEMM loop?
#   bash tools/power_step.sh > gpurun_out/power_step.txt
cd $GRAFT_REPO_ROOT
sample() {   # sample <pid>: one line per 0.5 s while the process lives
  while kill -0 $1 2>/dev/null; do
    echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks 2>&1 | grep -E 'Power|sclk' | sed -e 's/.*: //' | tr '\n' ' ')"
    sleep 0.4
  done
}
echo "== idle"; rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk"
echo "== training step, 1500 timed steps"
python bench.py --no-cpu-baseline --steps 1500 --warmup 50 > gpurun_out/power_step_bench.json 2> gpurun_out/power_step_bench.err &
pid=$!
sample $pid
wait $pid
python - <<'PY'
import json
j = json.loads([l for l in open('gpurun_out/power_step_bench.json') if l.startswith('{')][0])
print('bench: %.3f ms/step, %.1f clips/s; fc2 %s' % (j['ms_per_step'], j['value'], j['roofline']['avg_launch_us']))
PY
for shape in fc2 fc1 qkv proj; do
  echo "== $shape alone, product form (LN fold extras), back to back"
  python tools/gemm_bench.py --variant 0 --shapes $shape --ln --iters 6000 --rounds 2 &
  pid=$!
  sample $pid
  wait $pid
done
