#!/bin/bash
set -u
mkdir -p gpurun_out
for L in 0 4200 5600 6800 10240 20480 0 6800; do
  AGPU_EXP_LDS=$L timeout 300 python tools/probe/occ_probe2.py 2>&1 | tail -1 | tee -a gpurun_out/r05_occ_probe2.jsonl
done
