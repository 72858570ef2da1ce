# Round 5: would row-chain workgroups that leave half a CU free cost the step less?  19 launches per step, 33 us each, sleeping:
# 24 workgroups x 512 threads x 156 KB LDS (today's shape) against 48 x 256 threads x 78 KB (16-row panels) and smaller ones.
cd /tmp
T=$GRAFT_REPO_ROOT/tools/step_timeline.py
run() { echo -n "$1: "; python3 $T --steps 300 $2 2>/dev/null | grep "^wall" | cut -c1-28; }
for rep in 1 2; do
run "forwards only                  " "--no-head"
run "24 x 512 thr x 156 KB          " "--dummy spin2-24-512-159744-33-0:19"
run "48 x 256 thr x  78 KB          " "--dummy spin2-48-256-79872-33-0:19"
run "48 x 256 thr x  64 KB          " "--dummy spin2-48-256-65536-33-0:19"
run "96 x 128 thr x  39 KB          " "--dummy spin2-96-128-39936-33-0:19"
run "24 x 512 thr x   0 KB          " "--dummy spin2-24-512-0-33-0:19"
done
