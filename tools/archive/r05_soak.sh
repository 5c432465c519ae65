#!/bin/bash
# Round 5 soak: ABI + dyn fuzz on fresh seeds (the adaptive tile policy and the occupancy cap drawn at random), then the GPU suite twice
set -u
mkdir -p gpurun_out
AGPU_FUZZ_BASE=${1:-50000} AGPU_FUZZ_SEEDS=6000 timeout 1500 python -m pytest tests/test_gpu_fuzz_abi.py tests/test_gpu_fuzz_dyn.py -x -q 2>&1 | grep -E 'passed|failed' | tee gpurun_out/r05_soak.txt
for i in 1 2; do timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E 'passed|failed' | tee -a gpurun_out/r05_soak.txt; done
