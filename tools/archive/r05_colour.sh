#!/bin/bash
# Round 5, GPU box: the compare against the colour of its second column (k × 4 KiB), fresh process each time: is the best colour the same in every process?
set -u
mkdir -p gpurun_out
for i in 1 2 3 4 5; do
  python tools/probe/cmp_colour.py --quick 2>/dev/null | python3 -c "
import sys, json
rows = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('process $i:', ' '.join('%d:%.3f' % (r['k1'], r['frac']) for r in rows))" | tee -a gpurun_out/r05_colour_by_process.txt
done
