#!/usr/bin/env python3
"""Condense tools/profile_table.sh's three passes into profiles/:
  <round>_kernel_table.json        HIP-event table written by tools/kernel_table.py INSIDE the rocprofv3 --kernel-trace run
  <round>_kernel_table_stats.csv   rocprofv3's own per-kernel averages of that same process
  <round>_pmc_kernel_table.json    HBM bytes per launch of every kernel (FETCH_SIZE x 1024 x 2 [gfx950], WRITE_SIZE x 1024)
and print the DESIGN.md §4 table (markdown) built from the first file only."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
round_ = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = os.path.join(ROOT, "gpurun_out")
prof = os.path.join(ROOT, "profiles")
os.makedirs(prof, exist_ok=True)

src = os.path.join(out, f"kernel_table_{round_}.json")
if os.path.exists(src):
    shutil.copy(src, os.path.join(prof, f"{round_}_kernel_table.json"))
stats = glob.glob(os.path.join(out, f"tab_trace_{round_}", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(prof, f"{round_}_kernel_table_stats.csv"))
log = os.path.join(out, f"tab_trace_{round_}.log")
if os.path.exists(log):
    keep = [ln for ln in open(log, errors="replace") if ln.startswith(("{'kernel'", "H2D", "D2H", "config-1", "|"))]
    open(os.path.join(prof, f"{round_}_kernel_table_under_rocprof.log"), "w").writelines(keep)


def pmc(which, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, f"tab_{which}_{round_}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                per[r["Kernel_Name"].replace("void ", "")[:150]].append(float(r["Counter_Value"]))
    return per


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
summary = {}
for k in sorted(set(fetch) | set(write)):
    f, w = sorted(fetch.get(k, [])), sorted(write.get(k, []))
    fm, wm = (f[len(f) // 2] if f else None), (w[len(w) // 2] if w else None)
    summary[k] = {"launches": max(len(f), len(w)), "read_GB_per_launch_x2": None if fm is None else round(fm * 2048 / 1e9, 4),
                  "write_GB_per_launch": None if wm is None else round(wm * 1024 / 1e9, 4)}
if summary:
    json.dump({"note": "median per launch; read = FETCH_SIZE KiB x 1024 x 2 (gfx950: 128-B requests tallied at 64 B), write = WRITE_SIZE KiB x 1024",
               "kernels": summary}, open(os.path.join(prof, f"{round_}_pmc_kernel_table.json"), "w"), indent=1)
path = os.path.join(prof, f"{round_}_kernel_table.json")
if os.path.exists(path):
    d = json.load(open(path))
    print(f"\n(source: profiles/{round_}_kernel_table.json — one process, {d['device']})\n")
    print("| kernel | alg. B/row | ms | GB/s | frac of 8 TB/s | note |\n|---|---|---|---|---|---|")
    for r in d["kernels"]:
        print(f"| {r['kernel']} | {r['alg_B_per_row']:.4g} | {r['ms']} | {r['GBps']} | {r['frac_8TBs']} | {r.get('note', '')} |")
    for r in d.get("reference_bench_shapes", []):
        print(f"| {r['kernel']} | – | {r['us']} us | {r['GBps']} | – | {r['note']} |")
