#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/cap_pmc; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for grp in "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VALU SQ_WAVES" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_TA_BUSY_sum TD_TD_BUSY_sum"; do
  i=$((i+1)); timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -o t -- python3 "$REPO/tools/probe/cap_pmc.py" > "$OUT/g$i.log" 2>&1; echo "group $i rc=$?"
done
cd "$REPO"
python3 - <<'PY'
import collections, csv, glob, json
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/cap_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if k.startswith(("ew_kernel<float, UnSin", "cvt_wide_kernel<unsigned short, float", "lut8_kernel")):
            capped = int(r.get("LDS_Block_Size", 0) or 0) >= 4096
            per[(k[:44], "capped" if capped else "uncapped")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for (k, c), cs in sorted(per.items()):
    out[f"{k} [{c}]"] = {n: round(sorted(v)[len(v) // 2], 1) for n, v in sorted(cs.items())}
json.dump(out, open("gpurun_out/r05_cap_pmc.json", "w"), indent=1)
names = sorted({n for d in out.values() for n in d})
keys = sorted(out)
for i in range(0, len(keys), 2):
    a, b = out[keys[i]], out[keys[i + 1]] if i + 1 < len(keys) else {}
    print(keys[i], "|", keys[i + 1] if i + 1 < len(keys) else "")
    for n in names:
        if n in a and n in b and b[n]:
            print(f"   {n:42s} {a[n]:16.0f} {b[n]:16.0f}   capped/uncapped {a[n] / b[n]:.3f}")
PY
