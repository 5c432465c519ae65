#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=10 > $O/third_suite.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/third_suite.log
tail -5 $O/third_suite.log
timeout 1500 python tests/tools/exhaustive_vs_oracle.py sin cos > $O/third_exhaustive.log 2>&1
echo "exhaustive rc=$?"; tail -4 $O/third_exhaustive.log
cp gpurun_out/r03_exhaustive_vs_oracle.json $O/exhaustive_sincos.json 2>/dev/null
CAPS=-1,0 TILES=0,1,2,4 timeout 900 python tools/probe/r06_sincos_sweep.py 2>&1 | tee $O/sincos_sweep3.txt
