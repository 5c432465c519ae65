#!/usr/bin/env python3
"""Condense tools/profile_gather.sh's passes into profiles/<round>_gather_pmc.json: per variant (take / put × direct /
bucketed, 2^28 uniformly random 4-byte rows) the time and the HBM bytes PER ROW, kernel by kernel and in total.
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes (MI355X_MICROARCH.md §HBM), so
the read side is doubled — for scattered 4-byte loads that is the line-fetch volume, which is what bounds them."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
round_ = sys.argv[1] if len(sys.argv) > 1 else "r03"
base = os.path.join(ROOT, "gpurun_out", f"gather_{round_}")
N, LAUNCHES = 1 << 28, 3
SKIP = ("synth_", "__amd_rocclr", "trig16_build", "pow_build", "lut8_build")


def short(name):
    return name.replace("void ", "").split("(")[0]


def counters(which, variant, counter):
    per = collections.defaultdict(float)
    for f in glob.glob(os.path.join(base, f"{which}_{variant}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                per[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    return per


def durations(variant):
    per = collections.defaultdict(float)
    for f in glob.glob(os.path.join(base, f"trace_{variant}", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            per[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return per


out = {"rows": N, "launches_per_pass": LAUNCHES, "index_distribution": "uniform random over 2^28, 4-byte values",
       "variants_explained": "take_bucketed = merge-back pipeline (tk2_* kernels), take_pairs = pair pipeline (bkt_* kernels, gather_bucket = 3), put_bucketed = pair pipeline with range starts from the column scan, takebits_* = Boolean take (bucketed: the merge-back pipeline over the bitmap's words), putbits_* = Boolean put (bucketed: entries by destination region, applied in LDS)",
       "note": "per-row figures = totals of the pass / (launches x rows); read bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950), write bytes = WRITE_SIZE KiB x 1024",
       "variants": {}}
for v in ("take_direct", "take_bucketed", "take_pairs", "put_direct", "put_bucketed", "takebits_direct", "takebits_bucketed", "putbits_direct", "putbits_bucketed"):
    fetch, write, dur = counters("fetch", v, "FETCH_SIZE"), counters("write", v, "WRITE_SIZE"), durations(v)
    kernels = {}
    for k in sorted(set(fetch) | set(write) | set(dur)):
        if k.startswith(SKIP):
            continue
        kernels[k] = {"ms_per_launch": round(dur.get(k, 0.0) / LAUNCHES, 4),
                      "read_B_per_row": round(fetch.get(k, 0.0) * 1024 * 2 / LAUNCHES / N, 2),
                      "write_B_per_row": round(write.get(k, 0.0) * 1024 / LAUNCHES / N, 2)}
    tot_ms = sum(k["ms_per_launch"] for k in kernels.values())
    out["variants"][v] = {"kernels": kernels, "ms_per_launch": round(tot_ms, 4),
                          "G_rows_per_s": round(N / tot_ms / 1e6, 1) if tot_ms else None,
                          "read_B_per_row": round(sum(k["read_B_per_row"] for k in kernels.values()), 2),
                          "write_B_per_row": round(sum(k["write_B_per_row"] for k in kernels.values()), 2)}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
path = os.path.join(ROOT, "profiles", f"{round_}_gather_pmc.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "kernels"} for k, v in out["variants"].items()}, indent=1))
