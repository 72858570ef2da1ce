# fp8 mode: norm1 folded into the MX-fp8 qkv GEMM against the layernorm_mxfp8 pass (MVF_FP8_LN_FOLD=0), same box (GPU box):
#   bash tools/r6_fp8_fold.sh
mkdir -p gpurun_out/r06
o=gpurun_out/r06/fp8_ln_fold_ab.txt
: > $o
for rep in 1 2; do
  for f in 0 1; do
    echo "== MVF_FP8_LN_FOLD=$f (run $rep)" >> $o
    MVF_FP8_LN_FOLD=$f python tools/config_sweep.py "cfg5 DINOv2 ViT-L/14 @336, T=32, B=4, fp8" "cfg2 ViT-B/16, T=32, B=4, fp8" 2>&1 | grep "ms/step" >> $o
  done
done
cat $o
