# round 4: one-off A/B of a build against the headline / config 3 (+ the attention / LayerNorm kernel tests)
python sl-hwgat_amd/build.py > /dev/null 2>&1; echo "build rc $?"
mkdir -p gpurun_out/r04p
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > gpurun_out/r04p/pytest_kernels.log 2>&1; echo "kernels rc $?"; tail -3 gpurun_out/r04p/pytest_kernels.log
python bench.py --no-cpu-baseline --no-secondary --steps 20 > gpurun_out/r04p/bench_f32.json 2> gpurun_out/r04p/bench_f32.err; echo "f32 rc $?"
python bench.py --no-cpu-baseline --no-secondary --steps 20 --config 3 > gpurun_out/r04p/bench_c3.json 2> gpurun_out/r04p/bench_c3.err; echo "c3 rc $?"
python - <<'PY'
import json
for f in ("f32", "c3"):
    d = json.loads(open(f"gpurun_out/r04p/bench_{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("value_without_kernel_timers"), {k: v.get("frac") for k, v in d["kernels"].items() if isinstance(v, dict)})
    print("  ", {k: v["total_ms"] for k, v in d["other_hip_entry_points"].items() if "ln" in k})
PY
