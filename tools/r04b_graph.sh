# round 4, second GPU pass: full GPU suite on the ABI-4000 library (device-resident seeds), graphed train step A/B
set -o pipefail
O=gpurun_out/r04b
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build.log 2>&1; echo "build rc $?"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_gpu.log
for cfg in "--config 3" "--model hgate --dtype bf16" "--model wgate --dtype bf16" "--model hgate" "--model wgate"; do
  tag=$(echo $cfg | tr -d ' -')
  python bench.py $cfg --no-cpu-baseline --no-secondary > $O/bench_${tag}_eager.json 2> $O/bench_${tag}_eager.err; echo "$tag eager rc $?"
  python bench.py $cfg --no-cpu-baseline --no-secondary --graph > $O/bench_${tag}_graph.json 2> $O/bench_${tag}_graph.err; echo "$tag graph rc $?"
done
python bench.py --no-cpu-baseline --no-secondary --graph > $O/bench_default_graph.json 2> $O/bench_default_graph.err; echo "f32 graph rc $?"
python bench.py --no-cpu-baseline > $O/bench_default_nocpu.json 2> $O/bench_default_nocpu.err; echo "default rc $?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04b/bench_*.json")):
    try:
        d = json.load(open(f))
        sec = d.get("secondary") or {}
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("value_without_kernel_timers"),
              {k: v.get("value", v.get("error")) for k, v in sec.items()})
    except Exception as e:
        print(f, "ERR", e)
PY
