# Round 5: what about a narrow head kernel stretches the backbone?  19 launches per step (the encoder part's count) of 24 workgroups x 512
# threads x 156 KB LDS, busy for 33 us each in four ways, beside the pipelined forwards (tools/step_timeline.py --dummy spin2-...).
cd /tmp
T=$GRAFT_REPO_ROOT/tools/step_timeline.py
run() { echo -n "$1: "; python3 $T --steps 300 $2 2>/dev/null | grep "^wall" | cut -c1-28; }
run "forwards only                 " "--no-head"
run "sleep    24 WG x 33 us x 19   " "--dummy spin2-24-512-159744-33-0:19"
run "L2 stream 24 WG x 33 us x 19  " "--dummy spin2-24-512-159744-33-1:19"
run "MFMA     24 WG x 33 us x 19   " "--dummy spin2-24-512-159744-33-2:19"
run "LDS      24 WG x 33 us x 19   " "--dummy spin2-24-512-159744-33-3:19"
run "forwards only                 " "--no-head"
run "L2 stream 24 WG x 33 us x 38  " "--dummy spin2-24-512-159744-33-1:38"
run "L2 stream 24 WG x 66 us x 19  " "--dummy spin2-24-512-159744-66-1:19"
run "L2 stream 6 WG x 33 us x 19   " "--dummy spin2-6-512-159744-33-1:19"
run "L2 stream 96 WG x 33 us x 19  " "--dummy spin2-96-512-159744-33-1:19"
run "sleep    96 WG x 33 us x 19   " "--dummy spin2-96-512-159744-33-0:19"
run "forwards only                 " "--no-head"
