# round 4, fourth GPU pass: full GPU suite on the current tree, default bench WITH the cpu baseline, config-3 kernel stats
set -o pipefail
O=gpurun_out/r04d
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build.log 2>&1; echo "build rc $?"
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04d/bench_default.json"))
print(d["value"], d["ms_per_step"], {k: v.get("value", v.get("error")) for k, v in d["secondary"].items()})
c = d["cpu_baseline"]; print(c["value"], c["cores"], c["host_saturated"], c["thread_sweep_clips_per_s"])
PY
