# SQ counter passes over the one-pass LSTP kernels (GPU box): bash tools/pmc_sq_lstp.sh
cd /tmp && export TMPDIR=/tmp
o=$GRAFT_REPO_ROOT/gpurun_out/pmc_lstp
mkdir -p $o
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $o/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/lstp_bench.py > $o/p$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_sq_summary.py $o lstp_mfma_kernel,lstp_fused > $GRAFT_REPO_ROOT/gpurun_out/pmc_lstp.txt 2>&1
find $o -name "*.db" -delete; find $o -name "*counter_collection.csv" -delete
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_lstp.txt
