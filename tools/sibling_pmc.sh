# FETCH_SIZE / WRITE_SIZE passes (separate, as the guide prescribes) of the sibling attention kernels, bf16 storage
set -o pipefail
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/sibling_pmc
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/sib_$c -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/sibling_pmc.py bf16 > $O/run_$c.log 2>&1; echo "pass $c rc $?"
  python3 $GRAFT_REPO_ROOT/tools/pmc_sum.py /tmp/sib_$c blk_ band_ merge_k > $O/$c.json
done
cat $O/FETCH_SIZE.json $O/WRITE_SIZE.json
