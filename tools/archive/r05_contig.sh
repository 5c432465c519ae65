#!/bin/bash
# Round 5, GPU box: (1) the GPU suite on the round's first fixes; (2) does PHYSICALLY CONTIGUOUS backing (hipDeviceMallocContiguous
# through the dev switch AGPU_DEVICE_MALLOC_FLAGS=4) remove the allocation lottery of the compare and of tiles-per-block?
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05_pytest_gpu_a.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_pytest_gpu_a.log
for flags in 0 4 0 4; do
  echo "== placement_lottery AGPU_DEVICE_MALLOC_FLAGS=$flags"
  AGPU_DEVICE_MALLOC_FLAGS=$flags timeout 600 python tools/probe/placement_lottery.py 6 2>&1 | tee -a gpurun_out/r05_lottery_flags$flags.txt
done
for flags in 0 4 0 4; do
  echo "== prefetch_sweep AGPU_DEVICE_MALLOC_FLAGS=$flags"
  AGPU_DEVICE_MALLOC_FLAGS=$flags timeout 600 python tools/probe/prefetch_sweep.py 2>&1 >> gpurun_out/r05_prefetch_flags$flags.jsonl | tee -a gpurun_out/r05_prefetch_flags$flags.txt
done
