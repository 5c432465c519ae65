#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: HBM traffic of EVERY kernel family at 1e9 rows —
# tools/kernel_table.py under two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes, no tracing domains).
# tools/pmc_traffic.py --table condenses them into profiles/<round>_pmc_kernel_table.json.
set -u
ROUND=${1:-r01}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/tab_fetch_$ROUND" -- python3 "$REPO/tools/kernel_table.py" --iters 3 --tag pmc_fetch > "$OUT/tab_fetch_$ROUND.log" 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/tab_write_$ROUND" -- python3 "$REPO/tools/kernel_table.py" --iters 3 --tag pmc_write > "$OUT/tab_write_$ROUND.log" 2>&1
echo "write rc=$?"
cd "$REPO"
find "$OUT" -name "*counter_collection.csv" | head
