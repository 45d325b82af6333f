# round 4, end: the measurement pass behind profiles/r04p_* -- part 1: the whole GPU suite and the bench lines
# (headline with cpu_baseline + secondary, config 3 / 5, host-fed, process-group rehearsal with pinning, sibling models, graphed steps)
set -o pipefail
O=gpurun_out/r04p
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build.log 2>&1; echo "build rc $?"
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc $?"
python bench.py --config 3 --no-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err; echo "c3 rc $?"
HWGAT_FORCE_DIST=1 HWGAT_FORCE_PIN=1 python bench.py --no-cpu-baseline --no-secondary > $O/bench_force_dist.json 2> $O/bench_force_dist.err; echo "force_dist rc $?"
python bench.py --from-host --no-cpu-baseline --no-secondary > $O/bench_from_host.json 2> $O/bench_from_host.err; echo "from_host rc $?"
for m in hgate wgate; do for dt in f32 bf16; do
  python bench.py --model $m --dtype $dt --no-cpu-baseline > $O/bench_${m}_${dt}.json 2> $O/bench_${m}_${dt}.err; echo "$m $dt rc $?"
done; done
python bench.py --graph --no-cpu-baseline --no-secondary > $O/bench_default_graph.json 2> $O/bench_default_graph.err; echo "graph f32 rc $?"
python bench.py --graph --config 3 --no-cpu-baseline --no-secondary > $O/bench_config3_graph.json 2> $O/bench_config3_graph.err; echo "graph c3 rc $?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04p/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); sec = d.get("secondary") or {}
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("value_without_kernel_timers"), d["roofline"].get("frac"),
              {k: v.get("value", v.get("error")) for k, v in sec.items()}, (d.get("cpu_baseline") or {}).get("value"),
              {k: v.get("frac") for k, v in (d.get("kernels") or {}).items() if isinstance(v, dict) and "attn" in k})
    except Exception as e:
        print(f, "ERR", e)
PY
