#!/bin/bash
# Round 5, GPU box: sin / cos f32 — packs per lane (U = 2 product, 1, 3, 4: build variants through AGPU_LIB) × occupancy cap (tuning wave_lds)
set -u
mkdir -p gpurun_out
for lib in "" tools/probe/variants/libagpu_sc1.so tools/probe/variants/libagpu_sc3.so tools/probe/variants/libagpu_sc4.so "" tools/probe/variants/libagpu_sc4.so; do
  echo "== AGPU_LIB=${lib:-product (U = 2)}"
  AGPU_LIB=${lib:+$PWD/$lib} python - <<'PY' 2>&1 | tee -a gpurun_out/r05_sincos_u.txt
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, CmpQuery, GpuDevice
n = 1_000_000_000
dev = GpuDevice(0); p = ArrowComputePipeline(dev, "sc"); q = CmpQuery(dev); h = p._handle
p.set_tuning("tile_auto", 1)
f, g = dev.create_table_buffers([4 * n, 4 * n])
capi.call("agpu_synth_f32", h, C.c_void_p(f.ptr), n, 1, 0, C.c_float(-1000.0), C.c_float(1000.0)); p.sync()
def med(op):
    fn = lambda: capi.call("agpu_unary", h, op, capi.F32, C.c_void_p(f.ptr), C.c_void_p(g.ptr), n)
    for _ in range(4): fn()
    p.sync(); ts = []
    for _ in range(9):
        q.begin(p); fn(); q.end(p); ts.append(q.wait_for_results())
    return 8.0 * n / sorted(ts)[4] / 1e6 / 8000
for name, op in (("sin", capi.UN_SIN), ("cos", capi.UN_COS)):
    row = []
    for cap in (-1, 4200, 5600, 6800, 8000, 10240, 13600, 20480, -1, 6800):
        p.set_tuning("wave_lds", cap); row.append(f"{cap}:{med(op):.3f}")
    print(name, " ".join(row), flush=True)
PY
done
