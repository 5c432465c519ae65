#!/bin/bash
# Round 5, GPU box: tiles per block with the k tiles of a block S tiles apart inside one window (AGPU_TILE_NEAR=S) instead of a grid apart.
set -u
mkdir -p gpurun_out
bash tools/probe/box_fingerprint.sh > gpurun_out/r05_box_near.txt 2>&1; tail -1 gpurun_out/r05_box_near.txt
AGPU_TILE_NEAR=8 timeout 900 python -m pytest tests/test_gpu_tile_tunings.py tests/test_gpu_tile_auto.py -x -q 2>&1 | tail -3
for S in 0 8 64 512 4096 0 64; do
  echo "== AGPU_TILE_NEAR=$S"
  AGPU_TILE_NEAR=$S PREFETCH_KS=0,1,2,3,4,8,1,2 timeout 600 python tools/probe/prefetch_sweep.py 2>&1 >> gpurun_out/r05_near_$S.jsonl | tee -a gpurun_out/r05_near_$S.txt
done
