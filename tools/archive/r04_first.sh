#!/bin/bash
# round 4, first GPU contact: the new tests (arena reuse, comm proof / poison, swizzle at BASELINE scale) + a bench line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export AGPU_PERF_STRICT=0
timeout 1500 python -m pytest tests/test_gpu_pools.py tests/test_gpu_comm.py -x -q -m gpu > gpurun_out/r04_t1.log 2>&1
echo "pools+comm rc=$?" >> gpurun_out/r04_t1.log
timeout 1800 python -m pytest tests/test_gpu_swizzle_fullsize.py -q -m gpu --durations=20 > gpurun_out/r04_t2.log 2>&1
echo "swizzle fullsize rc=$?" >> gpurun_out/r04_t2.log
timeout 900 python -m pytest tests/test_gpu_bucketed.py -x -q -m gpu --durations=10 > gpurun_out/r04_t3.log 2>&1
echo "bucketed rc=$?" >> gpurun_out/r04_t3.log
timeout 600 python bench.py > gpurun_out/r04_bench_first.json 2> gpurun_out/r04_bench_first.err
echo "bench rc=$?" >> gpurun_out/r04_bench_first.err
tail -3 gpurun_out/r04_t1.log gpurun_out/r04_t2.log gpurun_out/r04_t3.log
