#!/bin/bash
# round 4: ONE box, one call — the whole GPU suite (timing expectations asserted), smoke, the plain bench line, the bench and the kernel table
# under rocprofv3 (trace + the two PMC passes each), the take / put passes.  Outputs under gpurun_out/; the summaries are copied into profiles/.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export AGPU_PERF_STRICT=1
timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=10 > gpurun_out/r04_gpu_suite.log 2>&1
echo "gpu suite rc=$?" | tee -a gpurun_out/r04_gpu_suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1
echo "smoke rc=$?" | tee -a gpurun_out/r04_smoke.log
timeout 900 python bench.py > gpurun_out/r04_bench_plain.json 2> gpurun_out/r04_bench_plain.err
echo "bench rc=$?"
timeout 1500 bash tools/profile_bench.sh r04 5 > gpurun_out/r04_profile_bench.log 2>&1
echo "profile_bench rc=$?"
timeout 2400 bash tools/profile_table.sh r04 > gpurun_out/r04_profile_table.log 2>&1
echo "profile_table rc=$?"
timeout 1500 bash tools/profile_gather.sh r04 > gpurun_out/r04_profile_gather.log 2>&1
echo "profile_gather rc=$?"
grep -E "passed|failed" gpurun_out/r04_gpu_suite.log | tail -2
