#!/bin/bash
# round 6, second session, call 3: the reductions' launches under rocprofv3 --kernel-trace (durations + gaps)
set -u
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp
rm -rf "$OUT/red_trace"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/red_trace" -o t -- python3 "$REPO/tools/probe/reduce_ab.py" > "$OUT/red_trace.log" 2>&1
echo "trace rc=$?"
cd "$REPO"
python3 tools/probe/reduce_ab_join.py $(find gpurun_out/red_trace -name "*kernel_trace.csv" | head -1) | tee gpurun_out/r06b_reduce_ab.txt
find gpurun_out/red_trace -name "*kernel_trace.csv" -size +8M -delete
