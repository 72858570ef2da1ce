# SQ counter passes over the streamed 32-query-row attention kernel (GPU box): bash tools/pmc_sq_attn32.sh [F N H variant tag]
F=${1:-256}; N=${2:-577}; H=${3:-16}; V=${4:-0}; TAG=${5:-pmc_attn32}
cd /tmp && export TMPDIR=/tmp
o=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $o
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA" "SQ_LEVEL_WAVES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $o/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/attn_bench.py $F 3 $N $H $V > $o/p$i.log 2>&1
done
find $o -name "*.db" -delete
python3 $GRAFT_REPO_ROOT/tools/pmc_sq_summary.py $o vit_attn32,vit_attn_bf16 > $o/summary.txt 2>&1
cat $o/summary.txt
