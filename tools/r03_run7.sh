set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bucketed.py -x -q -m gpu 2>&1 | tail -5
export TMPDIR=/tmp; REPO=$(pwd); cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_take_r03b" -- python3 "$REPO/tools/probe/take_passes.py" 268435456 7 > "$REPO/gpurun_out/r03_take_passes.log" 2>&1
cd "$REPO"; grep "take_\|put_" gpurun_out/r03_take_passes.log | head
f=$(find gpurun_out/prof_take_r03b -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:18]:
    print(r['Name'][:44].ljust(46), r['Calls'].rjust(4), f"{float(r['AverageNs'])/1e6:8.4f} ms")
PY
