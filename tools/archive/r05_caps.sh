#!/bin/bash
# Round 5, GPU box: the occupancy caps as product defaults — tests, then default vs tuning wave_lds = -1 (no cap) on the kernels that have one
set -u
mkdir -p gpurun_out
TAG=${1:-a}
bash tools/probe/box_fingerprint.sh > gpurun_out/r05_box_caps_$TAG.txt 2>&1; tail -1 gpurun_out/r05_box_caps_$TAG.txt
timeout 1500 python -m pytest tests/test_gpu_tile_tunings.py tests/test_gpu_tile_auto.py tests/test_gpu_fuzz_abi.py tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_zz_gpu_perf.py -x -q 2>&1 | tail -4
timeout 600 python tools/probe/caps_ab.py > gpurun_out/r05_caps_ab_$TAG.json 2> gpurun_out/r05_caps_ab_$TAG.txt; echo "caps rc=$?"; cat gpurun_out/r05_caps_ab_$TAG.txt
