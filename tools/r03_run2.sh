set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_null_counts.py tests/test_gpu_pools.py tests/test_gpu_by_name.py tests/test_gpu_sc1.py tests/test_gpu_host_copies.py tests/test_gpu_comm.py tests/test_gpu_arrow_cdata.py tests/test_cpp_host.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r03_pytest2.log
echo "pytest rc=$?"
tail -6 gpurun_out/r03_pytest2.log
python tools/kernel_table.py --tag r03a > gpurun_out/r03a_kernel_table.log 2>&1
echo "table rc=$?"
grep -i "popcount\|validity\|bitmap not" gpurun_out/r03a_kernel_table.log | head -20
