# round 4: one-off A/B of a build against the headline / config 3 (+ the attention kernel tests)
python sl-hwgat_amd/build.py > /dev/null 2>&1; echo "build rc $?"
mkdir -p gpurun_out/r04p
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > gpurun_out/r04p/pytest_kernels.log 2>&1; echo "kernels rc $?"; tail -3 gpurun_out/r04p/pytest_kernels.log
for a in "" "--config 3"; do
python bench.py --no-cpu-baseline --no-secondary --steps 20 $a > gpurun_out/r04p/bench_try.json 2> gpurun_out/r04p/bench_try.err; echo "rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04p/bench_try.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("value_without_kernel_timers"), {k: (v.get("frac"), v.get("avg_us"), v.get("ms_per_step")) for k, v in d["kernels"].items() if isinstance(v, dict)})
PY
done
