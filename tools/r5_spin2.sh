# Round 5: cost of a narrow memory-streaming kernel beside the pipelined forwards as a function of the footprint it streams
# (19 launches per step, 24 workgroups x 512 threads x 156 KB LDS, 33 us each; tools/step_timeline.py --dummy spin2-...-<KB>)
cd /tmp
T=$GRAFT_REPO_ROOT/tools/step_timeline.py
run() { echo -n "$1: "; python3 $T --steps 300 $2 2>/dev/null | grep "^wall" | cut -c1-28; }
run "forwards only              " "--no-head"
for kb in 16 256 1536 6144 24576 65536; do run "L2 stream, footprint $kb KB" "--dummy spin2-24-512-159744-33-1-$kb:19"; done
run "forwards only              " "--no-head"
run "sleep                      " "--dummy spin2-24-512-159744-33-0:19"
run "sleep, no LDS              " "--dummy spin2-24-512-0-33-0:19"
run "MFMA                       " "--dummy spin2-24-512-159744-33-2:19"
run "forwards only              " "--no-head"
