#!/usr/bin/env python3
"""Per-call durations (ms) of kernels whose name contains a pattern, from a rocprofv3 --kernel-trace CSV."""
import collections
import csv
import sys

path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
d = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if pat in name:
        d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in d.items():
    print(k, len(v), "calls; last:", [round(x, 3) for x in v[-6:]])
