# round 4: the four-wave fp32 block attention kernels (blk_attn_f32.hip): HGATE tests, then the kernels alone
set -o pipefail
O=gpurun_out/r04l
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build.log 2>&1; echo "build rc $?"
timeout -k 10 600 python -m pytest tests/test_gpu_hgate.py -m gpu -q -x > $O/pytest_hgate.log 2>&1; rc=$?; echo "hgate rc $rc"; tail -15 $O/pytest_hgate.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python tools/blk_one.py > $O/blk_one_f32.txt 2>&1 && cat $O/blk_one_f32.txt
PMC_OUT=r04l PMC_ARGS=f32 PMC_FILTER=blk_ bash tools/blk_pmc.sh
