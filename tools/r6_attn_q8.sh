# fp8 mode: the attention kernel's MX-fp8 epilogue against bf16 output + mvf_quant_mxfp8 (MVF_ATTN_Q8=0), same box (GPU box):
#   bash tools/r6_attn_q8.sh
mkdir -p gpurun_out/r06
o=gpurun_out/r06/attn_q8_ab.txt
: > $o
for rep in 1 2; do
  for q8 in 0 1; do
    echo "== MVF_ATTN_Q8=$q8 (run $rep)" >> $o
    MVF_ATTN_Q8=$q8 python tools/config_sweep.py "cfg5 DINOv2 ViT-L/14 @336, T=32, B=4, fp8" "cfg2 ViT-B/16, T=32, B=4, fp8" 2>&1 | grep "ms/step" >> $o
  done
done
cat $o
