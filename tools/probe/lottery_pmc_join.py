#!/usr/bin/env python3
"""Join tools/probe/lottery_pmc.py's per-allocation fractions with a rocprofv3 --pmc pass over the same process: the compare dispatches come
11 per allocation, in order; counters are averaged over the last 9 of each 11.  usage: lottery_pmc_join.py <fractions.json> <counter_collection.csv>"""
import csv
import json
import sys

rows = json.load(open(sys.argv[1]))
per = {}  # counter → [values in dispatch order] of the compare kernel
order = {}
with open(sys.argv[2]) as fh:
    for r in csv.DictReader(fh):
        if "cmp_ballot_kernel" not in r.get("Kernel_Name", ""):  # the main kernel only (a ragged end has its own little launch)
            continue
        per.setdefault(r["Counter_Name"], []).append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
out = []
for a in rows:
    rec = {"allocation": a["allocation"], "frac": a["frac"]}
    for c, vals in per.items():
        vals = [v for _, v in sorted(vals)]
        mine = vals[a["allocation"] * 11 + 2: a["allocation"] * 11 + 11]
        rec[c] = round(sum(mine) / max(len(mine), 1), 1)
    out.append(rec)
n_disp = {c: len(v) for c, v in per.items()}
assert all(v == 11 * len(rows) for v in n_disp.values()), n_disp
print(json.dumps(out))
