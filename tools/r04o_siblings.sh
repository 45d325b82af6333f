# round 4: sibling models after the fp32 attention kernels of blk_attn_f32.hip / band_attn_f32.hip: tests
set -o pipefail
O=gpurun_out/r04o
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build.log 2>&1; echo "build rc $?"
timeout -k 10 900 python -m pytest tests/test_gpu_hgate.py tests/test_gpu_wgate.py -m gpu -q > $O/pytest_siblings.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -15 $O/pytest_siblings.log
