#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1800 python -m pytest tests/ -x -q -m gpu --durations=5 > $O/fifth_suite.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/fifth_suite.log
grep -E "passed|failed|FAILED|^E  " $O/fifth_suite.log | head -20
timeout 1500 python tests/tools/exhaustive_vs_oracle.py log sin cos > $O/fifth_exhaustive.log 2>&1
echo "exhaustive rc=$?"; tail -5 $O/fifth_exhaustive.log
cp gpurun_out/r03_exhaustive_vs_oracle.json $O/exhaustive_log_sin_cos.json 2>/dev/null
CAPS=-1,0,6800,10240 TILES=0,1,2 timeout 900 python tools/probe/r06_sincos_sweep.py 2>&1 | tee $O/sincos_sweep5.txt | head -4
