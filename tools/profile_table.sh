#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: the ONE-RUN evidence for DESIGN.md §4.
#   pass 1  rocprofv3 --kernel-trace --stats over tools/kernel_table.py — the HIP-event table (kernel_table_<round>.json)
#           and rocprofv3's per-kernel averages come from the SAME process;
#   pass 2/3  --pmc FETCH_SIZE / WRITE_SIZE over the same script on the same box (PMC cannot be combined with tracing).
# tools/condense_table.py then writes profiles/<round>_kernel_table.json, _kernel_table_stats.csv, _pmc_kernel_table.json.
set -u
ROUND=${1:-r02}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tab_trace_$ROUND" -o t -- python3 "$REPO/tools/kernel_table.py" --iters 9 --tag "$ROUND" > "$OUT/tab_trace_$ROUND.log" 2>&1
echo "trace rc=$?"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/tab_fetch_$ROUND" -o t -- python3 "$REPO/tools/kernel_table.py" --iters 3 --tag pmc_fetch > "$OUT/tab_fetch_$ROUND.log" 2>&1
echo "fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/tab_write_$ROUND" -o t -- python3 "$REPO/tools/kernel_table.py" --iters 3 --tag pmc_write > "$OUT/tab_write_$ROUND.log" 2>&1
echo "write rc=$?"
cd "$REPO"
python3 tools/condense_table.py "$ROUND"
