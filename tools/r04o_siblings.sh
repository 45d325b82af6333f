# round 4: sibling models after the fp32 attention kernels of blk_attn_f32.hip / band_attn_f32.hip: tests, bench lines
set -o pipefail
O=gpurun_out/r04o
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build.log 2>&1; echo "build rc $?"
timeout -k 10 900 python -m pytest tests/test_gpu_hgate.py tests/test_gpu_wgate.py tests/test_gpu_determinism.py tests/test_gpu_graph.py -m gpu -q > $O/pytest_siblings.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 $O/pytest_siblings.log
[ $rc -eq 0 ] || exit $rc
for m in hgate wgate; do
  timeout -k 10 300 python bench.py --model $m --no-cpu-baseline --no-secondary > $O/bench_${m}_f32.json 2> $O/bench_${m}_f32.err; echo "$m rc $?"
done
python - <<'PY'
import json
for m in ("hgate", "wgate"):
    d = json.loads(open(f"gpurun_out/r04o/bench_{m}_f32.json").read().strip().splitlines()[-1])
    print(m, d["value"], d["ms_per_step"], {k: (v["frac"], v["avg_us"]) for k, v in d["kernels"].items()})
PY
