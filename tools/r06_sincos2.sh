#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
for i in 1 2; do
CAPS=-1,0,4200,6800 TILES=1,2,4,8 timeout 900 python tools/probe/r06_sincos_sweep.py 2>&1 | tee -a gpurun_out/r06/sincos_sweep2.txt
done
