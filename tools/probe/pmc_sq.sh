#!/bin/bash
# DEV TOOL (GPU box): shader-side counters of the take / put pipeline passes — where do the waves of P2 / G2 / F2 and P / G / F
# spend their cycles?  One rocprofv3 --pmc pass per counter group over tools/probe/take_passes.py (2 timed iterations).
#   bash tools/probe/pmc_sq.sh   → gpurun_out/r03_pmc_sq.json
set -u
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_sq
mkdir -p "$OUT"
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -o t -- python3 "$REPO/tools/probe/take_passes.py" 268435456 2 > "$OUT/g$i.log" 2>&1
  echo "group $i ($grp) rc=$?"
done
cd "$REPO"
python3 - <<'PY'
import collections, csv, glob, json, os
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if k.startswith(("tk2_", "bkt_")):
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(per.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = {c: round(v, 1) for c, v in m.items()}
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU"):
            if c in m:
                d[c + "/WAVE_CYCLES"] = round(m[c] / wc, 3)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        d["LDS_BANK_CONFLICT/IDX_ACTIVE"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"], 3)
    out[k] = d
json.dump(out, open("gpurun_out/r03_pmc_sq.json", "w"), indent=1)
for k, d in out.items():
    print(k, {c: v for c, v in d.items() if "/" in c})
PY
