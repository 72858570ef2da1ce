# does leaving a few CUs outside the persistent GEMM's budget (so that the head's small kernels never wait for a GEMM workgroup to
# leave) pay at the power cap?  400 sustained steps each, same box
cd $GRAFT_REPO_ROOT
for c in 0 248 240 224 0 248; do
  MVF_GEMM_CUS=$c python bench.py --no-cpu-baseline --steps 400 --warmup 30 2>gpurun_out/ab.err | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('MVF_GEMM_CUS=$c', j['ms_per_step'], j['value'])"
done
