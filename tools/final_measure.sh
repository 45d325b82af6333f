set -o pipefail
mkdir -p gpurun_out/r03fin
O=gpurun_out/r03fin
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc $?"
HWGAT_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-secondary > $O/bench_force_dist.json 2> $O/bench_force_dist.err; echo "force_dist rc $?"
python bench.py --config 3 --no-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err; echo "c3 rc $?"
python bench.py --from-host --no-cpu-baseline --no-secondary > $O/bench_from_host.json 2> $O/bench_from_host.err; echo "from_host rc $?"
for m in wgate hgate; do for dt in bf16 f32; do
  python bench.py --model $m --dtype $dt --no-cpu-baseline > $O/bench_${m}_${dt}.json 2> $O/bench_${m}_${dt}.err; echo "$m $dt rc $?"
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_f32 -o f32 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 10 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof_f32.json 2>/dev/null; echo "rocprof f32 rc $?"
cp /tmp/prof_f32/f32_kernel_stats.csv $GRAFT_REPO_ROOT/$O/f32_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config 3 --no-cpu-baseline --steps 10 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof_config3.json 2>/dev/null; echo "rocprof c3 rc $?"
cp /tmp/prof_c3/c3_kernel_stats.csv $GRAFT_REPO_ROOT/$O/config3_bf16_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_fetch -o fetch --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/nt8w_pmc.py > /dev/null 2>&1; echo "pmc fetch rc $?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_write -o write --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/nt8w_pmc.py > /dev/null 2>&1; echo "pmc write rc $?"
cd $GRAFT_REPO_ROOT
python tools/pmc_sum.py /tmp/pmc_fetch > $O/pmc_fetch.json
python tools/pmc_sum.py /tmp/pmc_write > $O/pmc_write.json
ls -la $O
