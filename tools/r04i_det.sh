set -o pipefail
O=gpurun_out/r04i
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build.log 2>&1; echo "build rc $?"
timeout -k 10 900 python -m pytest tests/test_gpu_determinism.py -m gpu -q -x > $O/pytest_det.log 2>&1; echo "det rc $?"; tail -25 $O/pytest_det.log
