set -o pipefail
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${PMC_OUT:-blk_pmc}
mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/blk_pmc_$i -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/${PMC_SCRIPT:-blk_pmc.py} ${PMC_ARGS:-} > /dev/null 2>&1; echo "pass $i rc $?"
  python3 $GRAFT_REPO_ROOT/tools/pmc_sum.py /tmp/blk_pmc_$i ${PMC_FILTER:-blk_attn} > $O/pass$i.json
done
cat $O/pass*.json
