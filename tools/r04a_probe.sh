# round 4, first GPU pass: Infinity-Cache probe, micro-batch A/B at config 3, whole-step PMC byte table
set -o pipefail
O=gpurun_out/r04a
mkdir -p $O
python sl-hwgat_amd/build.py > $O/build.log 2>&1; echo "build rc $?"
python bench.py --no-cpu-baseline > $O/bench_default_nocpu.json 2> $O/bench_default.err; echo "default rc $?"
for mb in 32 16 8; do
  python bench.py --config 3 --no-cpu-baseline --micro-batch $mb > $O/bench_c3_mb$mb.json 2> $O/bench_c3_mb$mb.err; echo "c3 mb$mb rc $?"
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/sb_$c -o p --output-format csv -- python3 $R/bench.py --config 3 --steps 2 --warmup 1 --no-kernel-timers --no-cpu-baseline > $R/$O/sb_$c.json 2> $R/$O/sb_$c.err; echo "pmc $c rc $?"
done
cd $R
python tools/step_bytes.py /tmp/sb_FETCH_SIZE /tmp/sb_WRITE_SIZE --steps 3 --itemsize 2 --label "config 3 (bf16), round-3 kernels" > $O/step_bytes_c3.json; echo "step_bytes rc $?"
ls -la $O
