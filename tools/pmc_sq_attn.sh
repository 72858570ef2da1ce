# SQ counter passes over the ViT attention kernels (GPU box): bash tools/pmc_sq_attn.sh
cd /tmp && export TMPDIR=/tmp
o=$GRAFT_REPO_ROOT/gpurun_out/pmc_attn
mkdir -p $o
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA" "SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $o/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/attn_bench.py 256 3 > $o/p$i.log 2>&1
done
find $o -name "*.db" -delete
