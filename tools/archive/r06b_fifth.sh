#!/bin/bash
# round 6, second session: two lock-step streams as the default — parity (new test file + the files that touch the kernels) and the kernel table
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_two_streams.py -x -q -m gpu > gpurun_out/r06b_t3.log 2>&1; echo "two_streams rc=$?"; grep -E "passed|failed|^E " gpurun_out/r06b_t3.log | head
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz_abi.py tests/test_gpu_fuzz_dyn.py tests/test_gpu_tile_tunings.py tests/test_gpu_sc1.py tests/test_gpu_runtime.py tests/test_gpu_by_name.py -x -q -m gpu > gpurun_out/r06b_t4.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^E " gpurun_out/r06b_t4.log | head
timeout 900 python tools/kernel_table.py --tag r06b3 > gpurun_out/r06b3_kernel_table.log 2>&1
echo "kernel_table rc=$?"
grep "^{" gpurun_out/r06b3_kernel_table.log | cut -c1-150 | head -70
