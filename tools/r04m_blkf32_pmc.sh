python sl-hwgat_amd/build.py > /dev/null 2>&1; echo "build rc $?"
python sl-hwgat_amd/build.py --lab > /dev/null 2>&1; echo "lab build rc $?"
mkdir -p gpurun_out/r04m
timeout -k 10 600 python -m pytest tests/test_gpu_hgate.py -m gpu -q -x > gpurun_out/r04m/pytest_hgate.log 2>&1; rc=$?; echo "hgate rc $rc"; tail -5 gpurun_out/r04m/pytest_hgate.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python tools/blk_one.py > gpurun_out/r04m/blk_one_f32.txt 2>&1 && cat gpurun_out/r04m/blk_one_f32.txt
timeout -k 10 120 python tools/blk_one.py > gpurun_out/r04m/blk_one_f32.txt 2>&1 && cat gpurun_out/r04m/blk_one_f32.txt
