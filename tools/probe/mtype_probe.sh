#!/bin/bash
# DEV TOOL (GPU box): tools/probe/mtype_probe plain, then under rocprofv3 --pmc for the size classes of the L2's fabric requests.
#   bash tools/probe/mtype_probe.sh [tag]  → gpurun_out/<tag>_mtype_probe.jsonl, gpurun_out/<tag>_mtype_pmc.json
set -u
TAG=${1:-r05}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/mtype_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
timeout 600 "$REPO/tools/probe/mtype_probe" 28 28 gsc 5 > "$REPO/gpurun_out/${TAG}_mtype_probe.jsonl" 2>&1
echo "plain rc=$?"; cat "$REPO/gpurun_out/${TAG}_mtype_probe.jsonl"
timeout 300 "$REPO/tools/probe/mtype_probe" 28 25 g 5 > "$REPO/gpurun_out/${TAG}_mtype_probe_128MiB_source.jsonl" 2>&1
echo "128 MiB source rc=$?"; cat "$REPO/gpurun_out/${TAG}_mtype_probe_128MiB_source.jsonl"
cd /tmp
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_WR_UNCACHED_32B_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -o t -- "$REPO/tools/probe/mtype_probe" 28 28 gs 1 > "$OUT/g$i.log" 2>&1
  echo "group $i ($grp) rc=$?"
done
cd "$REPO"
python3 - "$TAG" <<'PY'
import collections, csv, glob, json, sys
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/mtype_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if k.startswith(("gather_k", "scatter_k")):
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
pol = ["plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"]
mem = ["hipMalloc", "uncached", "finegrained", "contiguous"]
out = {}
for k, cs in sorted(per.items()):
    a = k[k.index("<") + 1:k.index(">")].split(",")
    name = f"{k.split('<')[0]} mem={mem[int(a[1])]} policy={pol[int(a[0])]}"
    out[name] = {c: round(sorted(v)[len(v) // 2] / 2**28, 4) for c, v in sorted(cs.items())}  # per row
json.dump(out, open(f"gpurun_out/{sys.argv[1]}_mtype_pmc.json", "w"), indent=1)
for k, v in out.items(): print(k, v)
PY
