#!/bin/bash
# Run ON THE GPU BOX from the repo root: tools/probe/lottery_pmc.py under one rocprofv3 --pmc pass per counter group (never together with
# tracing); prints one JSON line per pass = per-allocation {frac, counters}.  usage: lottery_pmc.sh "<group1>" "<group2>" …
REPO=$(pwd)
OUT=$REPO/gpurun_out/lottery_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i + 1))
  rm -rf /tmp/lp_$i
  LOTTERY_OUT=/tmp/lp_$i.json timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/lp_$i -o t -- python3 "$REPO/tools/probe/lottery_pmc.py" ${LOTTERY_N:-6} > /tmp/lp_$i.log 2>&1
  f=$(find /tmp/lp_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    echo "{\"group\": \"$grp\", \"rows\": $(python3 "$REPO/tools/probe/lottery_pmc_join.py" /tmp/lp_$i.json "$f")}" | tee -a "$OUT/passes.jsonl"
  else
    echo "group $grp: no counter file"; tail -5 /tmp/lp_$i.log
  fi
done
