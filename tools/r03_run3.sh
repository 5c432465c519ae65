set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_null_counts.py tests/test_gpu_pools.py tests/test_gpu_runtime.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03_pytest3.log
echo "pytest rc=$?"; tail -4 gpurun_out/r03_pytest3.log
python tools/probe/pool_placement.py 1000000000 3 > gpurun_out/r03_pool_placement.log 2>&1
echo "placement rc=$?"; grep "^{" gpurun_out/r03_pool_placement.log
python tools/kernel_table.py --tag r03b > gpurun_out/r03b_kernel_table.log 2>&1
grep -i "popcount\|validity\|bitmap not" gpurun_out/r03b_kernel_table.log | grep "^{" | cut -c1-200
