# round 4: fp32 band attention, workgroup-staged (band_attn_f32.hip): tests, lab sweep, the WGATE bench line
python sl-hwgat_amd/build.py > /dev/null 2>&1; echo "build rc $?"
python sl-hwgat_amd/build.py --lab > /dev/null 2>&1; echo "lab build rc $?"
mkdir -p gpurun_out/r04n
timeout -k 10 600 python -m pytest tests/test_gpu_wgate.py -m gpu -q -x > gpurun_out/r04n/pytest_wgate.log 2>&1; rc=$?; echo "wgate rc $rc"; tail -5 gpurun_out/r04n/pytest_wgate.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/band_f32_lab.py > gpurun_out/r04n/band_f32_lab.txt 2>&1; echo "rc $?"; cat gpurun_out/r04n/band_f32_lab.txt
timeout -k 10 300 python bench.py --model wgate --no-cpu-baseline --no-secondary > gpurun_out/r04n/bench_wgate_f32.json 2> gpurun_out/r04n/bench_wgate_f32.err; echo "wgate bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04n/bench_wgate_f32.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: (v.get("frac"), v.get("avg_us")) for k, v in d["kernels"].items() if isinstance(v, dict)})
PY
