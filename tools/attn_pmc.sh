# FETCH_SIZE / WRITE_SIZE passes (separate) of the window-attention kernels at the config-2 (fp32) and config-3 (bf16) stage-0 shape
set -o pipefail
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/attn_pmc
mkdir -p $O
for dt in f32 bf16; do for c in FETCH_SIZE WRITE_SIZE; do
  ATTN_PMC_DTYPE=$dt rocprofv3 --kernel-trace --pmc $c -d /tmp/attn_${dt}_$c -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/attn_pmc.py > $O/run_${dt}_$c.log 2>&1; echo "pass $dt $c rc $?"
  python3 $GRAFT_REPO_ROOT/tools/pmc_sum.py /tmp/attn_${dt}_$c win_attn merge_k > $O/${dt}_$c.json
done; done
