#!/usr/bin/env python3
"""Turn the per-kernel FETCH_SIZE / WRITE_SIZE sums of tools/pmc_sum.py into the traffic records bench.py reads
(`traffic`, `traffic_source` of the roofline objects):
    python tools/pmc_traffic_build.py win     f32_FETCH.json f32_WRITE.json bf16_FETCH.json bf16_WRITE.json > profiles/r04_attn_pmc_traffic.json
    python tools/pmc_traffic_build.py sibling f32_FETCH.json f32_WRITE.json bf16_FETCH.json bf16_WRITE.json > profiles/r04_sibling_attn_pmc_traffic.json
Conversion as MI355X_MICROARCH.md prescribes for gfx950: counters in KiB, FETCH_SIZE x 2, WRITE_SIZE exact; the
calibration kernel (merge_k: reads E, writes E) is reported next to them.  fp32 records at the top level, bf16 under "bf16"."""
import json
import sys

kind = sys.argv[1]
files = sys.argv[2:6]
SHAPES = {   # E elements of the launches tools/attn_pmc.py / tools/sibling_pmc.py make
    "win": {"hwgat_win_attn_fwd": ("win_attn_fwd_k", 64 * 128 * 80 * 128), "hwgat_win_attn_bwd": ("win_attn_bwd_k", 64 * 128 * 80 * 128)},
    "sibling": {"hwgat_blk_attn_fwd": (("blk_attn_fwd_k", "blk_fwd_b16_k", "blk_fwd_f32_k"), 64 * 128 * 29 * 128),
                "hwgat_blk_attn_bwd": (("blk_attn_bwd_k", "blk_bwd_b16_k", "blk_bwd_f32_k"), 64 * 128 * 29 * 128),
                "hwgat_band_attn_fwd": (("band_attn_fwd_k", "band_fwd_st_k", "band_fwd_f32st_k"), 64 * 128 * 64 * 128),
                "hwgat_band_attn_bwd": (("band_attn_bwd_k", "band_bwd_st_k", "band_bwd_f32st_k"), 64 * 128 * 64 * 128)},
}[kind]
CAL_E = {"win": 64 * 128 * 80 * 128, "sibling": 64 * 128 * 64 * 128}[kind]


def pick(d, subs):
    subs = (subs,) if isinstance(subs, str) else subs
    hits = {k: v for k, v in d.items() if any(s in k for s in subs)}
    assert len(hits) == 1, (subs, list(d))
    return next(iter(hits.items()))


def records(fetch, write, itemsize):
    out = {}
    for entry, (subs, E) in SHAPES.items():
        kf, f = pick(fetch, subs)
        kw, w = pick(write, subs)
        n = f["dispatches"]
        assert n == w["dispatches"] and kf == kw
        rd, wr = 2.0 * f["FETCH_SIZE"] * 1024 / n, w["WRITE_SIZE"] * 1024 / n
        alg = (7 if entry.endswith("bwd") else 4) * E * itemsize
        out[entry] = {"kernel": kf, "launches": n, "E_bytes": E * itemsize, "FETCH_SIZE_KiB": f["FETCH_SIZE"] / n,
                      "WRITE_SIZE_KiB": w["WRITE_SIZE"] / n, "traffic_bytes_per_launch": int(rd + wr),
                      "algorithmic_bytes": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 5)}
    kf, f = pick(fetch, "merge_k")
    _, w = pick(write, "merge_k")
    n = f["dispatches"]
    out["calibration_merge_k"] = {"kernel": kf, "launches": n, "known_bytes_each_way": CAL_E * itemsize,
                                  "fetch_ratio_to_known": round(f["FETCH_SIZE"] * 1024 / n / (CAL_E * itemsize), 5),
                                  "write_ratio_to_known": round(w["WRITE_SIZE"] * 1024 / n / (CAL_E * itemsize), 5)}
    return out


f32 = records(json.load(open(files[0])), json.load(open(files[1])), 4)
b16 = records(json.load(open(files[2])), json.load(open(files[3])), 2)
doc = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/"
                 + ("attn_pmc.py" if kind == "win" else "sibling_pmc.py") + " on 1x MI355X, round-4 binaries (ABI 4000); "
                 "tools/pmc_sum.py + tools/pmc_traffic_build.py; per launch",
       "note": "FETCH_SIZE reads 1/2 of the streamed bytes on gfx950 (MI355X_MICROARCH.md): x 2; WRITE_SIZE exact; calibration = merge_k"}
if kind == "win":
    doc["E_bytes"] = f32["hwgat_win_attn_fwd"]["E_bytes"]
    b16["E_bytes"] = b16["hwgat_win_attn_fwd"]["E_bytes"]
doc.update(f32)
doc["bf16"] = b16
print(json.dumps(doc, indent=1))
