#!/bin/bash
# round 6, second session: reductions with ONE finishing launch (parity at every size class, fullsize + comm + fuzz tests, the launches under the tracer)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sum or reduce" > gpurun_out/r06b_t1.log 2>&1; echo "parity rc=$?"; grep -E "passed|failed" gpurun_out/r06b_t1.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_null_counts.py tests/test_gpu_comm.py tests/test_gpu_fuzz_abi.py tests/test_gpu_by_name.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/r06b_t2.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^E " gpurun_out/r06b_t2.log | head
bash tools/r06b_third.sh
