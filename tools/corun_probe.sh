# A/B on ONE box: does capping the persistent GEMM at 224 VGPRs (64 left per SIMD lane) let the other streams' small-register
# kernels (LayerNorm, im2col, head row kernels) run inside the GEMM's CUs?  Usage: bash tools/corun_probe.sh <tag>
tag=${1:-corun}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
V=$R/video_rep_learning_amd/csrc/libmvf_hip_v224.so
B="python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10"
for rep in 1 2; do
  $B > $out/base_$rep.log 2>&1
  MVF_GEMM_BM=224 $B > $out/base_bm224_$rep.log 2>&1
  MVF_HIP_LIB=$V MVF_GEMM_BM=224 $B > $out/v224_bm224_$rep.log 2>&1
done
MVF_HIP_LIB=$V MVF_GEMM_BM=224 rocprofv3 --kernel-trace --output-format csv -d $out/trace_v224 -o t -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 3 --profile-steps 0 > $out/trace_v224.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/trace_base -o t -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 3 --profile-steps 0 > $out/trace_base.log 2>&1
python3 $R/tools/trace_overlap.py $out/trace_v224 > $out/overlap_v224.txt 2>&1
python3 $R/tools/trace_overlap.py $out/trace_base > $out/overlap_base.txt 2>&1
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -size +20M -delete
grep -h '"value"' $out/*.log | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print(d['ms_per_step'], d['value'])
    except Exception as e: print('?', l[:80])
"
for f in $out/base_*.log $out/v224_*.log; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f; done
cat $out/overlap_v224.txt $out/overlap_base.txt
