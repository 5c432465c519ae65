#!/bin/bash
# DEV TOOL (GPU box): memory-side counters of the narrow kernels next to the f32 add and a plain fill — where do the requests of
# cast u8→f32 (0.78 of the roof) wait that the add's (0.835) and the fill's (0.85) do not?  One rocprofv3 --pmc pass per group over
# tools/probe/narrow_run.py.   bash tools/probe/pmc_memside.sh [tag]  → gpurun_out/<tag>_pmc_memside.json
set -u
TAG=${1:-r04}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_memside
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
export NARROW_ONLY=cast_u8_f32,add_f32,fill_f32,cast_f32_u8,sin_u8,cast_u16_f32,sin_f32
i=0
for grp in "GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_NORMAL_WRITEBACK_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TD_TD_BUSY_sum" \
           "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM" \
           "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -o t -- python3 "$REPO/tools/probe/narrow_run.py" 1000000000 3 > "$OUT/g$i.log" 2>&1
  echo "group $i ($grp) rc=$? $(tail -c 600 $OUT/g$i.log | grep -o '"[a-z0-9_]*": {"ms": [0-9.]*' | tr '\n' ' ')"
done
cd "$REPO"
python3 - "$TAG" <<'PY'
import collections, csv, glob, json, sys
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_memside/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if k.startswith(("lut8_kernel", "cvt_wide_kernel", "cvt_narrow", "ew_kernel<float, OpAdd", "ew_kernel<float, UnSin", "fill_kernel", "cvt_")):
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(per.items()):
    m = {c: sorted(v)[len(v) // 2] for c, v in cs.items()}  # median launch
    out[k] = {c: round(v, 1) for c, v in sorted(m.items())}
json.dump(out, open(f"gpurun_out/{sys.argv[1]}_pmc_memside.json", "w"), indent=1)
cols = sorted({c for d in out.values() for c in d})
for c in cols:
    print(f"{c:48s}", "  ".join(f"{k[:28]:28s}={out[k].get(c, float('nan')):>16.1f}" for k in out))
PY
