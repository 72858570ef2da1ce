# Round 6: the data-parallel step on the ONE GPU of a test box -- a forced one-rank `nccl` (= RCCL) group issues every
# collective of the step (bucketed async gradient all-reduce, SyncBatchNorm all-gather + mvf_syncbn_merge, backward column-sum all-reduce).
#   (a) what the collectives + the 8-CU reserve cost against the plain step, alternating on one box;
#   (b) GPU_MAX_HW_QUEUES with the reducer live: MVF_HW_QUEUES=0 (runtime default, 4) / 8 / 16.
# Usage (GPU box): bash tools/r6_rccl.sh      -> gpurun_out/r06/rccl_forced_1rank.txt
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r06
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
B="--no-cpu-baseline --steps 200 --warmup 30 --profile-steps 1 --min-sustain-s 0"
f=$out/rccl_forced_1rank.txt
: > $f
line() { python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
c = d['config']
print('%-58s %7.3f ms/step  %6.1f clips/s  hw queues %s  gemm CUs %s  comm %s' % (sys.argv[1], d['ms_per_step'], d['value'], c['gpu_max_hw_queues'], c['gemm_cu_budget'], c.get('comm')))
" "$1"; }
for rep in 1 2 3; do
  env -u MVF_FORCE_REDUCER python3 $R/bench.py $B 2>/dev/null | line "plain step (no process group)" >> $f
  MVF_FORCE_REDUCER=1 MVF_HW_QUEUES=0 python3 $R/bench.py $B 2>/dev/null | line "forced one-rank RCCL, runtime-default hardware queues" >> $f
  MVF_FORCE_REDUCER=1 MVF_HW_QUEUES=8 python3 $R/bench.py $B 2>/dev/null | line "forced one-rank RCCL, 8 hardware queues" >> $f
  MVF_FORCE_REDUCER=1 MVF_HW_QUEUES=16 python3 $R/bench.py $B 2>/dev/null | line "forced one-rank RCCL, 16 hardware queues" >> $f
done
MVF_FORCE_REDUCER=1 MVF_HW_QUEUES=8 MVF_RCCL_CUS=0 python3 $R/bench.py $B 2>/dev/null | line "forced one-rank RCCL, 8 queues, NO CU reserve" >> $f
cat $f
