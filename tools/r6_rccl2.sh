# Round 6: which hardware queue does RCCL's stream land on?  (4 queues, the runtime default.)  The backbone's streams created before /
# after the process group, with filler streams in between: step time and the exposed all-reduce wait of the forced one-rank RCCL step.
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r06
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 MVF_FORCE_REDUCER=1 MVF_HW_QUEUES=0
B="--no-cpu-baseline --steps 200 --warmup 30 --profile-steps 1 --min-sustain-s 0"
f=$out/rccl_stream_order.txt
: > $f
line() { python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
c = d['config']
print('%-70s %7.3f ms/step  exposed all-reduce %s ms' % (sys.argv[1], d['ms_per_step'], c['comm']['exposed_allreduce_ms_per_step']))
" "$1"; }
for rep in 1 2; do
  python3 $R/bench.py $B 2>/dev/null | line "streams created lazily (RCCL's first)" >> $f
  MVF_STREAMS_FIRST=side,lane1 python3 $R/bench.py $B 2>/dev/null | line "side, lane1 before the process group" >> $f
  MVF_STREAMS_FIRST=side,lane1,dummy python3 $R/bench.py $B 2>/dev/null | line "side, lane1, one filler stream before the process group" >> $f
  MVF_STREAMS_FIRST=side,lane1,dummy,dummy python3 $R/bench.py $B 2>/dev/null | line "side, lane1, two filler streams before the process group" >> $f
  MVF_STREAMS_FIRST=dummy python3 $R/bench.py $B 2>/dev/null | line "one filler stream before the process group" >> $f
  MVF_STREAMS_FIRST=dummy,dummy python3 $R/bench.py $B 2>/dev/null | line "two filler streams before the process group" >> $f
done
cat $f
