# SQ / traffic counter passes over the two augmentation kernels (GPU box): bash tools/pmc_sq_augment.sh
cd /tmp && export TMPDIR=/tmp
o=$GRAFT_REPO_ROOT/gpurun_out/pmc_aug
mkdir -p $o
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM" "SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $o/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/augment_bench.py > $o/p$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_sq_summary.py $o finish_kernel,resize_color_kernel > $GRAFT_REPO_ROOT/gpurun_out/pmc_aug.txt 2>&1
find $o -name "*.db" -delete; find $o -name "*counter_collection.csv" -delete
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_aug.txt
