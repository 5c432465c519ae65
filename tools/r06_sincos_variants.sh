#!/bin/bash
# DEV TOOL (runs here, no GPU): A/B builds of the library that differ in the packs per lane of the f32 sin / cos tile kernel (AGPU_SINCOS_U);
# → tools/probe/variants/libagpu_sincos_u<N>.so, picked up on the GPU box through AGPU_LIB (tools/probe/r06_sincos_sweep.py)
cd "$(dirname "$0")/../arrow_gpu_amd/csrc" || exit 1
mkdir -p ../../tools/probe/variants build
for u in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DAGPU_SINCOS_U=$u -c elementwise.hip -o build/elementwise_su$u.o &
done
wait
for u in "$@"; do
  objs=$(ls build/*.o | grep -v "elementwise" | tr '\n' ' ')
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/probe/variants/libagpu_sincos_u$u.so $objs build/elementwise_su$u.o -L/opt/rocm/lib -lrccl -lrocprofiler-sdk-roctx
done
ls -la ../../tools/probe/variants/
