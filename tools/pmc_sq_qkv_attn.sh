# SQ counters of the fused qkv + attention kernel (vit_qkv_attn.hip) at BASELINE configs[1] size: five separate rocprofv3 --pmc passes
# (never combined with a trace) over a short tools/energy_probe.py run.  GPU box: bash tools/pmc_sq_qkv_attn.sh
cd /tmp && export TMPDIR=/tmp
o=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_qkv_attn
rm -rf $o; mkdir -p $o
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $o/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/energy_probe.py --kernels qkv_attn --seconds 0.02 > $o/p$i.log 2>&1
done
find $o -name "*.db" -delete
python3 $GRAFT_REPO_ROOT/tools/pmc_sq_summary.py $o vit_qkv_attn_kernel
