#!/bin/bash
# Round 6 soak on the final code: ABI + dyn fuzz on fresh seeds (every one of the eight tuning keys drawn per iteration; the pruned put pipeline, the packed-f32
# sin / cos / log, the chunk-wise chain interpreter underneath), then the GPU suite twice
cd "$(dirname "$0")/.."
set -u
mkdir -p gpurun_out
AGPU_FUZZ_BASE=${1:-70000} AGPU_FUZZ_SEEDS=8000 timeout 2400 python -m pytest tests/test_gpu_fuzz_abi.py tests/test_gpu_fuzz_dyn.py -x -q 2>&1 | grep -E 'passed|failed|Error|assert' | tee gpurun_out/r06_soak.txt
for i in 1 2; do timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E 'passed|failed' | tee -a gpurun_out/r06_soak.txt; done
