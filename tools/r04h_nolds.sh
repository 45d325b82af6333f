# round 4: A/B of counted (compiler-placed) lgkmcnt waits instead of lgkmcnt(0) in front of every MFMA cluster of the eight-wave kernels
set -o pipefail
O=gpurun_out/r04h
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build.log 2>&1; echo "build rc $?"
run() {  # tag
  NT_LAB_DTYPE=bf16 NT_LAB_STATS=1 python tools/nt_lab.py > $O/ntlab_$1.txt 2>&1; echo "ntlab $1 rc $?"
  NT_LAB_DTYPE=bf16 python tools/tn_lab.py > $O/tnlab_$1.txt 2>&1; echo "tnlab $1 rc $?"
  python bench.py --config 3 --no-cpu-baseline > $O/bench_c3_$1.json 2> $O/bench_c3_$1.err; echo "c3 $1 rc $?"
}
run base
cp tools/lab/nt8w_nolds.hip.txt sl-hwgat_amd/csrc/gemm_bf16_nt8w.hip
cp tools/lab/tn8w_nolds.hip.txt sl-hwgat_amd/csrc/gemm_bf16_tn8w.hip
python sl-hwgat_amd/build.py > $O/build_nolds.log 2>&1; echo "build nolds rc $?"
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -k "bf16" > $O/pytest_nolds.log 2>&1; echo "pytest nolds rc $?"; tail -2 $O/pytest_nolds.log
run nolds
cp tools/lab/tn8w_nolds_bfirst.hip.txt sl-hwgat_amd/csrc/gemm_bf16_tn8w.hip
python sl-hwgat_amd/build.py > $O/build_bfirst.log 2>&1; echo "build bfirst rc $?"
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -k "tn8w" > $O/pytest_bfirst.log 2>&1; echo "pytest bfirst rc $?"; tail -2 $O/pytest_bfirst.log
run bfirst
python - <<'PY'
import json
for v in ("base", "nolds", "bfirst"):
    try:
        d = json.load(open(f"gpurun_out/r04h/bench_c3_{v}.json"))
        print(v, d["value"], d["value_without_kernel_timers"], d["kernels"]["hwgat_linear_nt_bf16"]["ms_per_step"], d["kernels"]["hwgat_linear_tn_bf16"]["ms_per_step"])
    except Exception as e:
        print(v, "ERR", e)
PY
for v in base nolds bfirst; do echo "== $v"; tail -12 $O/tnlab_$v.txt; grep "per step" $O/ntlab_$v.txt; done
