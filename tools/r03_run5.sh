set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bucketed.py tests/test_gpu_pools.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r03_pytest5.log
echo "pytest rc=$?"; tail -8 gpurun_out/r03_pytest5.log
export TMPDIR=/tmp; REPO=$(pwd); cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_take_r03" -- python3 "$REPO/tools/probe/take_passes.py" > "$REPO/gpurun_out/r03_take_passes.log" 2>&1
echo "take rc=$?"; cd "$REPO"; grep "take_\|put_" gpurun_out/r03_take_passes.log | head
f=$(find gpurun_out/prof_take_r03 -name "*kernel_stats.csv" | head -1); echo $f; head -25 "$f" | cut -c1-160
