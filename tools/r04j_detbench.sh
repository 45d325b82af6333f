set -o pipefail
O=gpurun_out/r04j
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build.log 2>&1; echo "build rc $?"
python bench.py --no-cpu-baseline --no-secondary --deterministic > $O/bench_f32_det.json 2> $O/bench_f32_det.err; echo "f32 det rc $?"
python bench.py --config 3 --no-cpu-baseline --deterministic > $O/bench_c3_det.json 2> $O/bench_c3_det.err; echo "c3 det rc $?"
python bench.py --model hgate --dtype bf16 --no-cpu-baseline --deterministic > $O/bench_hgate_bf16_det.json 2> $O/bench_hgate_bf16_det.err; echo "hgate det rc $?"
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_gpu.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04j/bench_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], {k: v.get("ms_per_step") for k, v in d["kernels"].items() if "linear" in k})
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
python tools/tn8w_cache_probe.py > $O/tn8w_cache_probe.txt 2>&1; echo "probe rc $?"; cat $O/tn8w_cache_probe.txt
