mkdir -p gpurun_out/r06
o=gpurun_out/r06/fc2_epi_pipe_ab.txt
: > $o
for rep in 1 2 3; do
  for lib in "" tools/probes/libmvf_epipipe.so; do
    echo "== MVF_HIP_LIB=$lib" >> $o
    MVF_HIP_LIB=$lib python tools/energy_probe.py --kernels fc2_resid2,fc2_plain --seconds 3 2>&1 | grep "^fc2" >> $o
  done
done
for rep in 1 2; do
  for lib in "" tools/probes/libmvf_epipipe.so; do
    echo "== MVF_HIP_LIB=$lib  bench.py --steps 200 --warmup 50" >> $o
    MVF_HIP_LIB=$lib python bench.py --steps 200 --warmup 50 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity']['ok'])" >> $o
  done
done
cat $o
