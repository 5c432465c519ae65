#!/bin/bash
# Run ON THE GPU BOX from the repo root: two_streams_pmc.py under one rocprofv3 --pmc pass per counter group; per group the mean counter values
# of the sequential launches and of the two-stream launches.  usage: two_streams_pmc.sh "<group1>" "<group2>" …
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i + 1))
  rm -rf /tmp/tp_$i
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/tp_$i -o t -- python3 "$REPO/tools/probe/two_streams_pmc.py" 6 > /tmp/tp_$i.log 2>&1
  f=$(find /tmp/tp_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$grp" <<'P' | tee -a "$OUT/r06b_two_streams_pmc.jsonl"
import collections, csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "UnNeg" in r["Kernel_Name"] and "ew_kernel" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = by[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(by)
half = len(ids) // 2
out = {"group": sys.argv[2], "launches_per_order": half}
for name, sel in (("seq", ids[2::2]), ("two", ids[3::2])):  # alternating launches; the first pair is the warm-up
    acc = collections.defaultdict(list)
    for d in sel:
        for k, v in by[d].items():
            acc[k].append(v)
    out[name] = {k: round(sum(v) / len(v), 1) for k, v in acc.items()}
print(json.dumps(out))
P
  else
    echo "group $grp: no counter file"; tail -5 /tmp/tp_$i.log
  fi
done
