#!/bin/bash
# Round 5, GPU box: tiles per block far beyond 8 — grid = tiles / K shrinks to a few blocks per CU, i.e. PERSISTENT kernels whose blocks walk
# the column in lockstep with the next tile prefetched: ONE moving front instead of K fronts a grid apart (the form that lost 7-9 % in
# unlucky allocations, R4.1).  Two processes, because the allocation class follows the process.
set -u
mkdir -p gpurun_out
for run in a b; do
  echo "== process $run"
  PREFETCH_KS=0,1,2,16,32,64,128,256,512,1024,0,1,64,256 timeout 900 python tools/probe/prefetch_sweep.py 2>&1 >> gpurun_out/r05_persistent_$run.jsonl | tee gpurun_out/r05_persistent_$run.txt
done
