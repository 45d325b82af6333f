# round 4, end: the measurement pass behind profiles/r04p_* -- part 2: rocprofv3 kernel stats of the bench commands and the PMC
# passes (attention kernels' HBM traffic fp32 + bf16, whole-step bytes of config 3)
set -o pipefail
O=gpurun_out/r04p
R=$GRAFT_REPO_ROOT
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build2.log 2>&1; echo "build rc $?"
python bench.py --config 5 --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_config5_bf16.json 2> $O/bench_config5_bf16.err; echo "c5 bf16 rc $?"
python bench.py --config 5 --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config5_f32.json 2> $O/bench_config5_f32.err; echo "c5 f32 rc $?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_f32 -o f32 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 10 > $R/$O/bench_under_rocprof_f32.json 2>/dev/null; echo "rocprof f32 rc $?"
cp /tmp/prof_f32/f32_kernel_stats.csv $R/$O/f32_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python3 $R/bench.py --config 3 --no-cpu-baseline --steps 10 > $R/$O/bench_under_rocprof_config3.json 2>/dev/null; echo "rocprof c3 rc $?"
cp /tmp/prof_c3/c3_kernel_stats.csv $R/$O/config3_bf16_kernel_stats.csv
for m in hgate wgate; do
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -o $m --output-format csv -- python3 $R/bench.py --model $m --no-cpu-baseline --no-secondary --steps 10 > $R/$O/bench_under_rocprof_${m}_f32.json 2>/dev/null; echo "rocprof $m rc $?"
  cp /tmp/prof_$m/${m}_kernel_stats.csv $R/$O/${m}_f32_kernel_stats.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/sb_$c -o p --output-format csv -- python3 $R/bench.py --config 3 --steps 4 --warmup 1 --no-kernel-timers --no-cpu-baseline --no-secondary > /dev/null 2>&1; echo "step pmc $c rc $?"
  for dt in f32 bf16; do
    ATTN_PMC_DTYPE=$dt rocprofv3 --kernel-trace --pmc $c -d /tmp/attn_${dt}_$c -o p --output-format csv -- python3 $R/tools/attn_pmc.py > /dev/null 2>&1; echo "attn pmc $dt $c rc $?"
    python3 $R/tools/pmc_sum.py /tmp/attn_${dt}_$c win_attn merge_k > $R/$O/attn_${dt}_$c.json
    rocprofv3 --kernel-trace --pmc $c -d /tmp/sib_${dt}_$c -o p --output-format csv -- python3 $R/tools/sibling_pmc.py $dt > /dev/null 2>&1; echo "sibling pmc $dt $c rc $?"
    python3 $R/tools/pmc_sum.py /tmp/sib_${dt}_$c blk_ band_ merge_k > $R/$O/sib_${dt}_$c.json
  done
done
cd $R
python tools/step_bytes.py /tmp/sb_FETCH_SIZE /tmp/sb_WRITE_SIZE --steps 5 --itemsize 2 --label "config 3 (bf16), end of round 4" > $O/step_bytes_c3.json; echo "step_bytes rc $?"
python tools/pmc_traffic_build.py win $O/attn_f32_FETCH_SIZE.json $O/attn_f32_WRITE_SIZE.json $O/attn_bf16_FETCH_SIZE.json $O/attn_bf16_WRITE_SIZE.json > $O/attn_pmc_traffic.json; echo "attn traffic rc $?"
python tools/pmc_traffic_build.py sibling $O/sib_f32_FETCH_SIZE.json $O/sib_f32_WRITE_SIZE.json $O/sib_bf16_FETCH_SIZE.json $O/sib_bf16_WRITE_SIZE.json > $O/sibling_attn_pmc_traffic.json; echo "sibling traffic rc $?"
python - <<'PY'
import json
for f in ("attn_pmc_traffic", "sibling_attn_pmc_traffic"):
    d = json.load(open(f"gpurun_out/r04p/{f}.json"))
    for k, v in d.items():
        if isinstance(v, dict) and "traffic_over_algorithmic" in v: print(f, k, v["kernel"][:40], v["traffic_over_algorithmic"])
    for k, v in (d.get("bf16") or {}).items():
        if isinstance(v, dict) and "traffic_over_algorithmic" in v: print(f, "bf16", k, v["kernel"][:40], v["traffic_over_algorithmic"])
PY
