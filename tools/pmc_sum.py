#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter values per kernel name from the counter_collection CSV(s) under a directory:
    python tools/pmc_sum.py gpurun_out/pmc_dir [name-substring ...]
prints {kernel: {counter: sum, "dispatches": n}} as JSON (kernels filtered by the substrings, if given)."""
import csv, glob, json, os, re, sys

root, subs = sys.argv[1], sys.argv[2:]
out = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            if subs and not any(s in name for s in subs):
                continue
            short = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
            short = re.sub(r"\(.*$", "", short).strip() or name
            k = out.setdefault(short, {"dispatches": 0})
            k[r["Counter_Name"]] = k.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if (f, r["Dispatch_Id"]) not in seen:
                seen.add((f, r["Dispatch_Id"]))
                k["dispatches"] += 1
print(json.dumps(out, indent=1))
