#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 passes over bench.py.
#   1. --kernel-trace --stats      → per-kernel average duration (must agree with bench.py's HIP-event numbers)
#   2. --pmc FETCH_SIZE            → HBM read traffic   (separate pass; TCC slots do not fit both)
#   3. --pmc WRITE_SIZE            → HBM write traffic
# Outputs land in gpurun_out/prof_*; tools/pmc_traffic.py condenses them into profiles/.
set -u
ROUND=${1:-r01}
STEPS=${2:-5}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_trace_$ROUND" -- python3 "$REPO/bench.py" --steps "$STEPS" --warmup 2 --no-cpu-baseline --no-traffic > "$OUT/prof_trace_$ROUND.log" 2>&1
echo "trace rc=$?"
# 1b. the same trace of the HEADLINE STEP ALONE (--no-extra-configs): the extras launch the headline's kernels again on buffers of their own
#     (extra.fused's unfused baseline adds IN PLACE at 1.93 ms, extra.layout_pool over ordinary pool blocks), so the full run's per-kernel
#     average is not the timed loop's; this pass's is
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_trace_headline_$ROUND" -- python3 "$REPO/bench.py" --steps "$STEPS" --warmup 2 --no-cpu-baseline --no-traffic --no-extra-configs > "$OUT/prof_trace_headline_$ROUND.log" 2>&1
echo "headline trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch_$ROUND" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > "$OUT/prof_fetch_$ROUND.log" 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_write_$ROUND" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > "$OUT/prof_write_$ROUND.log" 2>&1
echo "write rc=$?"
cd "$REPO"
python3 tools/pmc_traffic.py "$ROUND" || true
find "$OUT" -name "*.csv" | head -20
