#!/usr/bin/env python3
"""Condense the rocprofv3 passes written by tools/profile_bench.sh into profiles/:

  profiles/<round>_kernel_stats.csv     per-kernel call count / average duration (from --kernel-trace --stats)
  profiles/<round>_pmc_summary.json     per-kernel HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes
  profiles/hbm_traffic.json             {"add_f32_bytes_per_launch": …} read by bench.py for roofline.traffic

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB
(bytes = value × 1024); on gfx950 FETCH_SIZE reports exactly ½ of the bytes of a wide coalesced streaming read
(16 B/lane) so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  The compare kernel
reads with 4-B-per-lane loads, a width the guide calls uncalibrated: its corrected figure is reported next to the
raw one and next to the algorithmic byte count so the calibration is visible.
"""
from __future__ import annotations

import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(pattern):
    """Newest matching file only: gpurun merges every call's outputs into gpurun_out/, so older passes pile up there."""
    hits = glob.glob(os.path.join(ROOT, "gpurun_out", pattern), recursive=True)
    return [max(hits, key=os.path.getmtime)] if hits else []


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name[:140]


def kernel_stats(round_, which="trace"):
    rows = []
    for f in find(f"prof_{which}_{round_}/**/*kernel_stats.csv"):
        with open(f) as fh:
            rows.extend(list(csv.DictReader(fh)))
    return rows


def pmc(round_, which, counter):
    """→ {kernel_name: [values per dispatch]}"""
    per = {}
    for f in find(f"prof_{which}_{round_}/**/*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                per.setdefault(r.get("Kernel_Name", "?"), []).append(float(r.get("Counter_Value", 0.0)))
    return per


def main():
    round_ = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    stats = kernel_stats(round_)
    if stats:
        keys = list(stats[0].keys())
        with open(os.path.join(ROOT, "profiles", f"{round_}_kernel_stats.csv"), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=keys)
            w.writeheader()
            for r in stats:
                w.writerow(r)
    head = kernel_stats(round_, "trace_headline")  # the headline step alone (tools/profile_bench.sh pass 1b)
    if head:
        with open(os.path.join(ROOT, "profiles", f"{round_}_kernel_stats_headline.csv"), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(head[0].keys()))
            w.writeheader()
            for r in head:
                w.writerow(r)
    fetch = pmc(round_, "fetch", "FETCH_SIZE")
    write = pmc(round_, "write", "WRITE_SIZE")
    summary = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, [])
        wv = write.get(k, [])
        # steady state: drop nothing, the kernels of interest are launched many times with identical shapes → median
        fm = sorted(f)[len(f) // 2] if f else None
        wm = sorted(wv)[len(wv) // 2] if wv else None
        summary[short(k)] = {
            "launches_fetch_pass": len(f), "launches_write_pass": len(wv),
            "FETCH_SIZE_KiB_per_launch_raw": fm, "WRITE_SIZE_KiB_per_launch_raw": wm,
            "read_bytes_per_launch_corrected_x2": None if fm is None else fm * 1024 * 2,
            "write_bytes_per_launch": None if wm is None else wm * 1024,
        }
    for k, v in summary.items():
        r, w_ = v["read_bytes_per_launch_corrected_x2"], v["write_bytes_per_launch"]
        v["hbm_bytes_per_launch"] = None if (r is None or w_ is None) else r + w_
    with open(os.path.join(ROOT, "profiles", f"{round_}_pmc_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    add = [v for k, v in summary.items() if "ew_kernel" in k and "OpAdd" in k and "float" in k]
    # the bench step's second kernel: the i32 compare WITH the fused validity AND (template argument `true`); the un-fused and f32 compares of
    # extra.configs / extra.fused are other instantiations
    cmpk = [v for k, v in summary.items() if "cmp_ballot_kernel<int, 4, true>" in k] or \
           [v for k, v in summary.items() if "cmp_ballot_kernel" in k or "cmp_vec_kernel" in k]
    out = {}
    if add and add[0]["hbm_bytes_per_launch"]:
        out["add_f32_bytes_per_launch"] = add[0]["hbm_bytes_per_launch"]
    # fused launches only: the same instantiation also runs the un-fused compares of extra.configs (no validity pointers: 8.125 B/row,
    # one bitmap written).  The two passes replay the same program, so launch i of one is launch i of the other: pair them and keep the
    # launches that wrote TWO bitmaps.
    fk = [k for k in fetch if "cmp_ballot_kernel<int, 4, true>" in k]
    if fk and fk[0] in write and len(fetch[fk[0]]) == len(write[fk[0]]):
        pairs = [(f_ * 2048 + w_ * 1024, w_ * 1024) for f_, w_ in zip(fetch[fk[0]], write[fk[0]])]
        wmax = max(w_ for _, w_ in pairs)
        fused = sorted(b_ for b_, w_ in pairs if w_ > 0.75 * wmax)
        if fused:
            out["eq_i32_bytes_per_launch"] = fused[len(fused) // 2]
            out["eq_i32_fused_launches"] = len(fused)
    elif cmpk and cmpk[0]["hbm_bytes_per_launch"]:
        out["eq_i32_bytes_per_launch"] = cmpk[0]["hbm_bytes_per_launch"]
    if out:
        out["source"] = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, round {round_}; read side doubled (gfx950)"
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w") as fh:
            json.dump(out, fh, indent=1)
    print(json.dumps({"kernels_with_stats": len(stats), "kernels_with_pmc": len(summary), "traffic": out}, indent=1))


if __name__ == "__main__":
    main()
