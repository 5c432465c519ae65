#!/bin/bash
# Round 5, GPU box: the adaptive tiles-per-block policy — its tests, the tile-tuning invariance tests, the ABI fuzz, then what it decides and
# what that is worth in two fresh processes (the kernel families at 1e9 rows with tile_auto off / on, alternating).
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_tile_auto.py tests/test_gpu_tile_tunings.py tests/test_gpu_fuzz_abi.py tests/test_gpu_fused.py -x -q > gpurun_out/r05_tile_auto_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r05_tile_auto_tests.log
for run in a b; do
  timeout 600 python tools/probe/tile_auto_ab.py > gpurun_out/r05_tile_auto_ab_$run.json 2> gpurun_out/r05_tile_auto_ab_$run.txt; echo "ab $run rc=$?"; cat gpurun_out/r05_tile_auto_ab_$run.txt
done
