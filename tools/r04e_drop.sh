# round 4: sibling attention dropout tests first, then the whole GPU suite
set -o pipefail
O=gpurun_out/r04e
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build.log 2>&1; echo "build rc $?"
timeout -k 10 600 python -m pytest tests/test_gpu_hgate.py tests/test_gpu_wgate.py -m gpu -q -k "dropout" > $O/pytest_drop.log 2>&1; echo "drop rc $?"; tail -15 $O/pytest_drop.log
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest_gpu.log
