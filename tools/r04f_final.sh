# round 4: the measurement pass behind profiles/r04f_* (headline, config 3 / 5, host-fed, process-group rehearsal, sibling
# models, rocprofv3 kernel stats, PMC: whole-step bytes and the attention kernels' traffic, fp32 + bf16)
set -o pipefail
O=gpurun_out/r04f
R=$GRAFT_REPO_ROOT
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build.log 2>&1; echo "build rc $?"
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc $?"
python bench.py --config 3 --no-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err; echo "c3 rc $?"
python bench.py --config 5 --dtype bf16 --steps 3 --warmup 1 > $O/bench_config5_bf16.json 2> $O/bench_config5_bf16.err; echo "c5 bf16 rc $?"
python bench.py --config 5 --dtype f32 --steps 2 --warmup 1 > $O/bench_config5_f32.json 2> $O/bench_config5_f32.err; echo "c5 f32 rc $?"
HWGAT_FORCE_DIST=1 HWGAT_FORCE_PIN=1 python bench.py --no-cpu-baseline --no-secondary > $O/bench_force_dist.json 2> $O/bench_force_dist.err; echo "force_dist rc $?"
python bench.py --from-host --no-cpu-baseline --no-secondary > $O/bench_from_host.json 2> $O/bench_from_host.err; echo "from_host rc $?"
for m in hgate wgate; do for dt in f32 bf16; do
  python bench.py --model $m --dtype $dt > $O/bench_${m}_${dt}.json 2> $O/bench_${m}_${dt}.err; echo "$m $dt rc $?"
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_f32 -o f32 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 10 > $R/$O/bench_under_rocprof_f32.json 2>/dev/null; echo "rocprof f32 rc $?"
cp /tmp/prof_f32/f32_kernel_stats.csv $R/$O/f32_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python3 $R/bench.py --config 3 --no-cpu-baseline --steps 10 > $R/$O/bench_under_rocprof_config3.json 2>/dev/null; echo "rocprof c3 rc $?"
cp /tmp/prof_c3/c3_kernel_stats.csv $R/$O/config3_bf16_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/sb_$c -o p --output-format csv -- python3 $R/bench.py --config 3 --steps 4 --warmup 1 --no-kernel-timers --no-cpu-baseline > /dev/null 2>&1; echo "step pmc $c rc $?"
  for dt in f32 bf16; do
    ATTN_PMC_DTYPE=$dt rocprofv3 --kernel-trace --pmc $c -d /tmp/attn_${dt}_$c -o p --output-format csv -- python3 $R/tools/attn_pmc.py > /dev/null 2>&1; echo "attn pmc $dt $c rc $?"
    python3 $R/tools/pmc_sum.py /tmp/attn_${dt}_$c win_attn merge_k > $R/$O/attn_${dt}_$c.json
    rocprofv3 --kernel-trace --pmc $c -d /tmp/sib_${dt}_$c -o p --output-format csv -- python3 $R/tools/sibling_pmc.py $dt > /dev/null 2>&1; echo "sibling pmc $dt $c rc $?"
    python3 $R/tools/pmc_sum.py /tmp/sib_${dt}_$c blk_ band_ merge_k > $R/$O/sib_${dt}_$c.json
  done
done
cd $R
python tools/step_bytes.py /tmp/sb_FETCH_SIZE /tmp/sb_WRITE_SIZE --steps 5 --itemsize 2 --label "config 3 (bf16), round-4 kernels" > $O/step_bytes_c3.json; echo "step_bytes rc $?"
python tools/pmc_traffic_build.py win $O/attn_f32_FETCH_SIZE.json $O/attn_f32_WRITE_SIZE.json $O/attn_bf16_FETCH_SIZE.json $O/attn_bf16_WRITE_SIZE.json > $O/attn_pmc_traffic.json; echo "attn traffic rc $?"
python tools/pmc_traffic_build.py sibling $O/sib_f32_FETCH_SIZE.json $O/sib_f32_WRITE_SIZE.json $O/sib_bf16_FETCH_SIZE.json $O/sib_bf16_WRITE_SIZE.json > $O/sibling_attn_pmc_traffic.json; echo "sibling traffic rc $?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04f/bench_*.json")):
    try:
        d = json.load(open(f)); sec = d.get("secondary") or {}
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("value_without_kernel_timers"), d["roofline"].get("frac"),
              {k: v.get("value", v.get("error")) for k, v in sec.items()}, (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
