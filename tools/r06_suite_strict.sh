#!/bin/bash
# round 6: the strict suite alone (timing expectations asserted)
cd "$(dirname "$0")/.."
export AGPU_PERF_STRICT=1
timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=10 > gpurun_out/r06_gpu_suite.log 2>&1
echo "gpu suite rc=$?" | tee -a gpurun_out/r06_gpu_suite.log
grep -E "passed|failed|FAILED|^E  " gpurun_out/r06_gpu_suite.log | head
grep "^\[perf\]" gpurun_out/r06_gpu_suite.log | cut -c1-300
