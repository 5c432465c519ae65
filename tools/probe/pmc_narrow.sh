#!/bin/bash
# DEV TOOL (GPU box): shader-side counters of the config-4 kernels (VERDICT r3 #4a: is the LDS the bound of lut8_kernel /
# trig16_kernel?).  One rocprofv3 --pmc pass per counter group over tools/probe/narrow_run.py.
#   bash tools/probe/pmc_narrow.sh [tag]  → gpurun_out/<tag>_pmc_narrow.json
set -u
TAG=${1:-r04}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_narrow
mkdir -p "$OUT"
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -o t -- python3 "$REPO/tools/probe/narrow_run.py" 1000000000 2 > "$OUT/g$i.log" 2>&1
  echo "group $i ($grp) rc=$?"
done
cd "$REPO"
python3 - "$TAG" <<'PY'
import collections, csv, glob, json, sys
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_narrow/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if k.startswith(("lut8_kernel", "trig16_kernel", "cvt_wide_kernel", "cast_chain_kernel", "ew_kernel<float, UnSin", "ew_kernel<float, UnCos", "ew_kernel<float, UnSinh", "cmp_")):
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
json.dump(out, open(f"gpurun_out/{sys.argv[1]}_pmc_narrow.json", "w"), indent=1)
for k, d in out.items():
    print(k, {c: v for c, v in d.items() if "/" in c})
PY
