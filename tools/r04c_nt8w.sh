# round 4, third GPU pass: full GPU suite; A/B of the eight-wave NT kernel with store-aware DMA waits (v1: resumed tiles only, v2: uniform)
set -o pipefail
O=gpurun_out/r04c
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build.log 2>&1; echo "build rc $?"
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log
NT_LAB_DTYPE=bf16 NT_LAB_STATS=1 python tools/nt_lab.py > $O/ntlab_base.txt 2>&1; echo "lab base rc $?"
python bench.py --config 3 --no-cpu-baseline > $O/bench_c3_base.json 2> $O/bench_c3_base.err; echo "c3 base rc $?"
for v in v1 v2; do
  cp tools/lab/nt8w_$v.hip.txt sl-hwgat_amd/csrc/gemm_bf16_nt8w.hip
  python sl-hwgat_amd/build.py > $O/build_$v.log 2>&1; echo "build $v rc $?"
  timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -k "nt8w or bf16" > $O/pytest_$v.log 2>&1; echo "pytest $v rc $?"; tail -2 $O/pytest_$v.log
  NT_LAB_DTYPE=bf16 NT_LAB_STATS=1 python tools/nt_lab.py > $O/ntlab_$v.txt 2>&1; echo "lab $v rc $?"
  python bench.py --config 3 --no-cpu-baseline > $O/bench_c3_$v.json 2> $O/bench_c3_$v.err; echo "c3 $v rc $?"
done
python - <<'PY'
import json
for v in ("base", "v1", "v2"):
    try:
        d = json.load(open(f"gpurun_out/r04c/bench_c3_{v}.json"))
        print(v, d["value"], d["value_without_kernel_timers"], d["kernels"]["hwgat_linear_nt_bf16"]["ms_per_step"], d["kernels"]["hwgat_linear_tn_bf16"]["ms_per_step"])
    except Exception as e:
        print(v, "ERR", e)
PY
paste -d'|' <(cut -c1-60,60-120 $O/ntlab_base.txt) <(cut -c60-110 $O/ntlab_v1.txt) <(cut -c60-110 $O/ntlab_v2.txt) | tail -40
