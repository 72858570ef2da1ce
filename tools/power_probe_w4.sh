# Socket power and clock while the four-wave probe loop runs for seconds (GPU box): bash tools/power_probe_w4.sh
cd $GRAFT_REPO_ROOT
for mode in 2 1; do
  echo "== wave4 full loop, operand mode $mode (2 = N(0,1), 1 = zeros)"
  tools/probes/wave4_probe 2 256 256 $mode 40 &
  pid=$!
  sleep 2.0
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.5; done
  wait $pid
done
