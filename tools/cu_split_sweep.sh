#!/bin/bash
# CU-partition sweep of the dW side stream: one bench line per setting -> gpurun_out/cu_split/*.json
mkdir -p gpurun_out/cu_split
run() { name=$1; shift; python bench.py --no-cpu-baseline --no-secondary --steps 12 --warmup 4 "$@" > gpurun_out/cu_split/$name.json 2> gpurun_out/cu_split/$name.err || echo "$name failed"; }
run base --no-kernel-timers
run side_unmasked --dw-side-cus 0
run side32 --dw-side-cus 32
run side64 --dw-side-cus 64
run side96 --dw-side-cus 96
run side32_rest --dw-side-cus 32 --main-cus-rest
run side64_rest --dw-side-cus 64 --main-cus-rest
run c3_base --config 3 --no-kernel-timers
run c3_side64 --config 3 --dw-side-cus 64
run c3_side32_rest --config 3 --dw-side-cus 32 --main-cus-rest
python - <<'PY'
import json, glob, os
out = {}
for f in sorted(glob.glob("gpurun_out/cu_split/*.json")):
    try:
        d = json.load(open(f))
        out[os.path.basename(f)[:-5]] = {"clips_per_s": d["value"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"], "dw_side_stream": d.get("dw_side_stream")}
    except Exception as e:
        out[os.path.basename(f)[:-5]] = {"error": str(e), "stderr_tail": open(f[:-5] + ".err").read()[-400:]}
json.dump(out, open("gpurun_out/cu_split/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
