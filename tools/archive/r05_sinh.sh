#!/bin/bash
# Round 5, GPU box: sinh f32 — packs per lane (U = 4 product, 2, 1) × occupancy cap (tuning wave_lds), build variants through AGPU_LIB
set -u
mkdir -p gpurun_out
for u in 4 2 1 4 2; do
  echo "== sinh U=$u"
  AGPU_LIB=$PWD/tools/probe/variants/libagpu_sinh$u.so python - <<'PY' 2>&1 | tee -a gpurun_out/r05_sinh.txt
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "sinh"); q = CmpQuery(dev); h = p._handle
p.set_tuning("tile_auto", 1)
f, g = dev.create_table_buffers([4 * n, 4 * n])
capi.call("agpu_synth_f32", h, C.c_void_p(f.ptr), n, 1, 0, C.c_float(-30.0), C.c_float(30.0)); p.sync()
def med():
    fn = lambda: capi.call("agpu_unary", h, capi.UN_SINH, capi.F32, C.c_void_p(f.ptr), C.c_void_p(g.ptr), n)
    for _ in range(4): fn()
    p.sync(); ts = []
    for _ in range(9):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return 8.0 * n / sorted(ts)[4] / 1e6 / 8000
row = []
for cap in (-1, 4200, 5600, 6800, 8000, 10240, 13600, 20480, -1, 6800):
    p.set_tuning("wave_lds", cap); row.append(f"{cap}:{med():.3f}")
print(" ".join(row), flush=True)
PY
done
