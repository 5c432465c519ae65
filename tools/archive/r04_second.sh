#!/bin/bash
# round 4, second GPU contact: cast-headed chains, the one-outer-step sin / cos, the LDS counters of the narrow trig kernels
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_sc1.py tests/test_cpp_host.py -x -q -m gpu > gpurun_out/r04_s1.log 2>&1
echo "fused+sc1+cpp rc=$?" >> gpurun_out/r04_s1.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -x -q -m gpu --durations=8 > gpurun_out/r04_s2.log 2>&1
echo "parity+golden+fullsize rc=$?" >> gpurun_out/r04_s2.log
timeout 600 python tools/probe/trig16_vs_pair.py > gpurun_out/r04_trig16_vs_pair.json 2> gpurun_out/r04_trig16_vs_pair.err
timeout 600 python tools/probe/narrow_run.py 1000000000 9 > gpurun_out/r04_narrow_run.json 2> gpurun_out/r04_narrow_run.err
timeout 1500 bash tools/probe/pmc_narrow.sh r04 > gpurun_out/r04_pmc_narrow.log 2>&1
timeout 600 python bench.py > gpurun_out/r04_bench_second.json 2> gpurun_out/r04_bench_second.err
echo "bench rc=$?" >> gpurun_out/r04_bench_second.err
tail -n 3 gpurun_out/r04_s1.log gpurun_out/r04_s2.log; cat gpurun_out/r04_trig16_vs_pair.json gpurun_out/r04_narrow_run.json
