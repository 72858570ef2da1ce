# Round 5: per-kernel durations of the head inside the training step, kernels one at a time (rocprofv3 --kernel-trace --stats of
# bench.py --serial).  Usage (GPU box): bash tools/r5_head_stats.sh [tag]     output: gpurun_out/r05/<tag>_kernel_stats.csv
tag=${1:-head}
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$tag -o run -- python3 $GRAFT_REPO_ROOT/bench.py --serial --no-cpu-baseline --steps 30 --warmup 10 > $out/st_$tag.log 2>&1
cp $out/st_$tag/run_kernel_stats.csv $out/${tag}_kernel_stats.csv
rm -rf $out/st_$tag
python3 - <<P
import csv
rows=list(csv.DictReader(open('$out/${tag}_kernel_stats.csv')))
steps=[int(r['Calls']) for r in rows if 'scl_grad' in r['Name']][0]
tot=0; nl=0
for r in rows:
    n=r['Name']
    if 'gemm_tc256' in n or 'vit_qkv' in n or 'layernorm_kernel<unsigned' in n or 'im2col' in n or 'at::' in n or 'rocblas' in n or 'cls_row' in n or 'cast_bf16' in n or 'layernorm_kernel<float' in n: continue
    c=int(r['Calls']); tot+=float(r['TotalDurationNs'])/1e3/steps; nl+=c/steps
    print('%-62s %5.1f /step  avg %7.1f us  %7.1f us/step' % (n.replace('(anonymous namespace)::','')[:62], c/steps, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3/steps))
print('head: %.1f launches/step, %.1f us/step of kernels (serial)' % (nl, tot))
P
