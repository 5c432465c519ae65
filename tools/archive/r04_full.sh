#!/bin/bash
# round 4: the whole GPU suite + the bench under the profiler passes that back the committed numbers
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export AGPU_PERF_STRICT=${AGPU_PERF_STRICT:-1}  # the runs that produce committed evidence assert the timing expectations
timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r04_gpu_suite.log 2>&1
echo "gpu suite rc=$?" >> gpurun_out/r04_gpu_suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r04_smoke.log
timeout 900 python bench.py > gpurun_out/r04_bench_plain.json 2> gpurun_out/r04_bench_plain.err
echo "bench rc=$?" >> gpurun_out/r04_bench_plain.err
tail -n 5 gpurun_out/r04_gpu_suite.log gpurun_out/r04_smoke.log gpurun_out/r04_bench_plain.err
