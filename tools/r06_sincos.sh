#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
for lib in "" tools/probe/variants/libagpu_sincos_u1.so tools/probe/variants/libagpu_sincos_u4.so ""; do
  AGPU_LIB=${lib:+$PWD/$lib} timeout 900 python tools/probe/r06_sincos_sweep.py 2>&1 | tee -a gpurun_out/r06/sincos_sweep.txt
done
