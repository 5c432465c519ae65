#!/bin/bash
# DEV TOOL: A/B of the sin / cos f32 formulations — builds four variants of elementwise.hip (AGPU_SINCOS_FORM x AGPU_SINCOS_PACK) here (no GPU
# needed), `bash tools/probe/sincos_ab.sh run` times them alternately on the GPU box.
cd "$(dirname "$0")/../.."
V=tools/probe/variants
if [ "${1:-build}" = build ]; then
  mkdir -p $V
  OBJS=$(ls arrow_gpu_amd/csrc/build/*.o | grep -v "elementwise")
  for form in 0 1; do for pack in 0 1; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DAGPU_SINCOS_FORM=$form -DAGPU_SINCOS_PACK=$pack \
        -c arrow_gpu_amd/csrc/elementwise.hip -o /tmp/ew_f${form}p${pack}.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $V/libagpu_sincos_f${form}p${pack}.so $OBJS /tmp/ew_f${form}p${pack}.o -L/opt/rocm/lib -lrccl -lrocprofiler-sdk-roctx &
  done; done; wait; ls -la $V/*.so
else
  for round in 1 2 3; do for v in f0p0 f0p1 f1p0 f1p1; do
    echo -n "$v "; AGPU_LIB=$PWD/$V/libagpu_sincos_$v.so NARROW_ONLY=sin_f32,cos_f32 python tools/probe/narrow_run.py 1000000000 9
  done; done
fi
