#!/usr/bin/env python3
"""Bytes moved by EVERY kernel of one train step, from two rocprofv3 PMC passes of the same bench command:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/sb_fetch -o p --output-format csv -- python3 bench.py --config 3 \
        --steps 2 --warmup 1 --no-kernel-timers --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/sb_write -o p --output-format csv -- python3 bench.py ... (same)
    python tools/step_bytes.py /tmp/sb_fetch /tmp/sb_write --steps 3 --itemsize 2 > profiles/r04_step_bytes.json

(separate passes: FETCH_SIZE and WRITE_SIZE do not fit the TCC counters together).  Conversion as MI355X_MICROARCH.md
prescribes for gfx950: both counters are in KiB; FETCH_SIZE reports half of the bytes of wide streaming reads, so it is
doubled; WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics.  Infinity-Cache hits are counted (the
counters sit on the L2's memory side), so this is L2 <-> fabric traffic, an upper bound of the HBM bytes.

`--steps` = the number of train steps the profiled command ran in all (warm-up + timed).  Output: per kernel (template
instantiation) dispatches and bytes per step, its share, and the total against SURVEY 8d's plan (45 E s per block x 8
blocks + 4 E s) with E = B*T*K*d0 elements.
"""
import argparse
import csv
import glob
import json
import os
import re


def sums(root, counter):
    out = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                name = r["Kernel_Name"]
                short = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
                short = re.sub(r"\(.*$", "", short).strip() or name
                k = out.setdefault(short, [0, 0.0])
                k[1] += float(r["Counter_Value"])
                if (f, r["Dispatch_Id"]) not in seen:
                    seen.add((f, r["Dispatch_Id"]))
                    k[0] += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--itemsize", type=int, default=2)
    ap.add_argument("--E", type=int, default=64 * 128 * 80 * 128, help="elements of one activation tensor (B*T*K*d0)")
    ap.add_argument("--blocks", type=int, default=8)
    ap.add_argument("--label", default="")
    a = ap.parse_args()
    fe, wr = sums(a.fetch_dir, "FETCH_SIZE"), sums(a.write_dir, "WRITE_SIZE")
    Es = a.E * a.itemsize
    rows, tot_f, tot_w = {}, 0.0, 0.0
    for name in sorted(set(fe) | set(wr)):
        nf, f = fe.get(name, (0, 0.0))
        nw, w = wr.get(name, (0, 0.0))
        fb, wb = 2.0 * f * 1024 / a.steps, w * 1024 / a.steps
        tot_f, tot_w = tot_f + fb, tot_w + wb
        rows[name] = {"dispatches_per_step": round(max(nf, nw) / a.steps, 2), "read_bytes_per_step": int(fb),
                      "write_bytes_per_step": int(wb), "bytes_per_step": int(fb + wb),
                      "in_units_of_E_s": round((fb + wb) / Es, 3)}
    total = tot_f + tot_w
    for r in rows.values():
        r["share"] = round(r["bytes_per_step"] / total, 4) if total else 0.0
    rows = dict(sorted(rows.items(), key=lambda kv: -kv[1]["bytes_per_step"]))
    plan = (45 * a.blocks + 4) * Es
    print(json.dumps({
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py; "
                  "FETCH_SIZE x 2 (gfx950), KiB -> bytes; tools/step_bytes.py", "label": a.label,
        "steps_profiled": a.steps, "E_elements": a.E, "itemsize": a.itemsize, "E_s_bytes": Es,
        "total_bytes_per_step": int(total), "read_bytes_per_step": int(tot_f), "write_bytes_per_step": int(tot_w),
        "total_in_units_of_E_s": round(total / Es, 2), "per_block_in_units_of_E_s": round((total / Es - 4) / a.blocks, 2),
        "plan_survey_8d_bytes_per_step": plan, "total_over_plan": round(total / plan, 3), "kernels": rows}, indent=1))


if __name__ == "__main__":
    main()
