set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pools.py tests/test_gpu_runtime.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03_pytest4.log
echo "pytest rc=$?"; tail -4 gpurun_out/r03_pytest4.log
python tools/probe/pool_placement.py 1000000000 4 > gpurun_out/r03_pool_placement.log 2>&1
echo "placement rc=$?"; grep "^{" gpurun_out/r03_pool_placement.log
