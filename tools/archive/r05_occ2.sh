#!/bin/bash
set -u
mkdir -p gpurun_out
TAG=${1:-x}
bash tools/probe/box_fingerprint.sh > gpurun_out/r05_box_occ_$TAG.txt 2>&1; tail -1 gpurun_out/r05_box_occ_$TAG.txt
for L in 0 5600 6800 8000 10240 0 6800; do
  AGPU_DYN_LDS=$L timeout 300 python tools/probe/occ_probe.py 2>&1 | tail -1 | tee -a gpurun_out/r05_occ_probe_$TAG.jsonl
done
