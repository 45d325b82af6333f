# round 4: one-off A/B of a build against the headline
python sl-hwgat_amd/build.py > /dev/null 2>&1; echo "build rc $?"
mkdir -p gpurun_out/r04p
for a in "" ""; do
python bench.py --no-cpu-baseline --no-secondary --steps 20 $a > gpurun_out/r04p/bench_try.json 2> gpurun_out/r04p/bench_try.err; echo "rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04p/bench_try.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("value_without_kernel_timers"), {k: (v.get("frac"), v.get("avg_us"), v.get("ms_per_step")) for k, v in d["kernels"].items() if isinstance(v, dict)})
PY
done
