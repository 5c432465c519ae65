#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1800 python -m pytest tests/ -x -q -m gpu --durations=10 > $O/fourth_suite.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/fourth_suite.log
grep -E "passed|failed|FAILED|^E  " $O/fourth_suite.log | head -20
for shape in 0 64 0 64; do
  echo "== AGPU_BITMAP_BLOCK=$shape"
  AGPU_BITMAP_BLOCK=$shape timeout 900 python tools/kernel_table.py --tag r06b_$shape 2>/dev/null | grep -E "validity AND|popcount|bitmap not|merge" | grep "^|"
done 2>&1 | tee $O/fourth_bitmap_ab.txt
