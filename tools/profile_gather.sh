#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: HBM traffic and per-kernel time of take / put at 2^28 uniformly
# random rows, direct vs bucketed — one rocprofv3 pass per (variant, counter), never --pmc together with tracing.
# tools/pmc_gather.py condenses the passes into profiles/<round>_gather_pmc.json.
set -u
ROUND=${1:-r03}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/gather_$ROUND
mkdir -p "$OUT"
cd /tmp
for v in take_direct take_bucketed take_pairs put_direct put_bucketed takebits_direct takebits_bucketed putbits_direct putbits_bucketed; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$v" -o t -- python3 "$REPO/tools/probe/gather_pmc.py" $v > "$OUT/trace_$v.log" 2>&1
  echo "$v trace rc=$?"
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_$v" -o t -- python3 "$REPO/tools/probe/gather_pmc.py" $v > "$OUT/fetch_$v.log" 2>&1
  echo "$v fetch rc=$?"
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_$v" -o t -- python3 "$REPO/tools/probe/gather_pmc.py" $v > "$OUT/write_$v.log" 2>&1
  echo "$v write rc=$?"
done
cd "$REPO"
python3 tools/pmc_gather.py "$ROUND"
