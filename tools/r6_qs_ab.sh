# frozen-backbone q rows pre-scaled by log2(e) / 8 (product; the streamed attention kernel's QS form) against plain q (MVF_ATTN_QS=0), same box:
#   bash tools/r6_qs_ab.sh
mkdir -p gpurun_out/r06
o=gpurun_out/r06/attn_qs_ab.txt
: > $o
for rep in 1 2; do
  for q in 0 1; do
    echo "== MVF_ATTN_QS=$q (run $rep)" >> $o
    MVF_ATTN_QS=$q python tools/config_sweep.py "penn_mvf.yml exactly" "cfg5 DINOv2" "pouring_mvf.yml" 2>&1 | grep "ms/step" | cut -c1-170 >> $o
  done
done
cat $o
