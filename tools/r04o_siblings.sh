# round 4: sibling models: tests and the four bench lines
set -o pipefail
O=gpurun_out/r04o
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build.log 2>&1; echo "build rc $?"
timeout -k 10 900 python -m pytest tests/test_gpu_hgate.py tests/test_gpu_wgate.py -m gpu -q > $O/pytest_siblings.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 $O/pytest_siblings.log
[ $rc -eq 0 ] || exit $rc
for m in hgate wgate; do for dt in f32 bf16; do
  timeout -k 10 300 python bench.py --model $m --dtype $dt --no-cpu-baseline --no-secondary > $O/bench_${m}_${dt}.json 2> $O/bench_${m}_${dt}.err; echo "$m $dt rc $?"
done; done
python - <<'PY'
import json
for m in ("hgate", "wgate"):
    for dt in ("f32", "bf16"):
        d = json.loads(open(f"gpurun_out/r04o/bench_{m}_{dt}.json").read().strip().splitlines()[-1])
        print(m, dt, d["value"], d["ms_per_step"], {k: (v.get("frac"), v.get("avg_us")) for k, v in d["kernels"].items() if isinstance(v, dict) and "attn" in k})
PY
