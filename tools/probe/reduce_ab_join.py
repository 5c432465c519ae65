#!/usr/bin/env python3
"""DEV TOOL: from a rocprofv3 --kernel-trace CSV of reduce_ab.py: per reduction call (a big kernel followed by its small ones) the
durations and the idle gaps, medians over the calls."""
import csv
import statistics
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
calls, cur = {}, None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 200_000:  # a main kernel opens a call
        cur = {"main": name, "t0": s, "parts": [(name, s, e)]}
        calls.setdefault(name, []).append(cur)
    elif cur is not None and s - cur["parts"][-1][2] < 50_000:
        cur["parts"].append((name, s, e))
for main, cs in calls.items():
    cs = cs[2:]  # warm-up
    if not cs:
        continue
    total = statistics.median(c["parts"][-1][2] - c["t0"] for c in cs)
    print(f"{main}: {len(cs)} calls, first start → last end {total / 1e3:.1f} us")
    for k in range(max(len(c["parts"]) for c in cs)):
        ds = [c["parts"][k][2] - c["parts"][k][1] for c in cs if len(c["parts"]) > k]
        gs = [c["parts"][k][1] - c["parts"][k - 1][2] for c in cs if len(c["parts"]) > k] if k else [0]
        nm = next(c["parts"][k][0] for c in cs if len(c["parts"]) > k)
        print(f"    {nm[:70]:70s} gap {statistics.median(gs) / 1e3:6.2f} us  dur {statistics.median(ds) / 1e3:8.2f} us")
