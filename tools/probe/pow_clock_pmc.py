#!/usr/bin/env python3
"""Effective shader clock per launch from a `rocprofv3 --pmc GRBM_GUI_ACTIVE` pass over tools/probe/pow_clock.py:
busy cycles ÷ (End − Start) ns = GHz.  Usage: pow_clock_pmc.py <counter_collection.csv> → gpurun_out/pow_clock_pmc.json"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
per = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    name = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if dur > 200_000:  # the 1e9-row launches only
        per[name].append({"ms": round(dur / 1e6, 4), "busy_cycles": float(r["Counter_Value"]), "GHz": round(float(r["Counter_Value"]) / dur, 4)})
out = {}
for k, v in per.items():
    ghz = [x["GHz"] for x in v]
    ms = [x["ms"] for x in v]
    out[k] = {"launches": len(v), "GHz_min": min(ghz), "GHz_max": max(ghz), "ms_min": min(ms), "ms_max": max(ms), "per_launch": v}
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "pow_clock_pmc.json"), "w"), indent=1)
for k, v in out.items():
    print(k, {kk: vv for kk, vv in v.items() if kk != "per_launch"})
