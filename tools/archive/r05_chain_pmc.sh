#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/chain_pmc; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -o t -- python3 "$REPO/tools/probe/chain_pmc.py" > "$OUT/g$i.log" 2>&1; echo "group $i rc=$?"
done
cd "$REPO"
python3 - <<'PY'
import collections, csv, glob
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/chain_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if k.startswith(("chain_kernel", "ew_kernel<float, UnSin", "ew_kernel<float, OpMul")):
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = 1 << 28
for k, cs in sorted(per.items()):
    print(k)
    for c, v in sorted(cs.items()):
        vals = sorted(v)
        # chain_kernel<float,true,0,false> runs two different chains: print all distinct medians
        print(f"   {c:24s} per row: " + " ".join(f"{x / rows:.3f}" for x in vals))
PY
