cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p3 -o c3 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config 3 --no-cpu-baseline --no-kernel-timers --steps 10 --warmup 5 > /dev/null 2>&1
cp /tmp/p3/c3_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/c3_stats_now.csv
rocprofv3 --kernel-trace --stats -d /tmp/ph -o hg --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --model hgate --dtype bf16 --no-cpu-baseline --no-kernel-timers --steps 10 --warmup 5 > /dev/null 2>&1
cp /tmp/ph/hg_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/hg_stats_now.csv
